/*
 * objnerf_hip.h -- C ABI of libobjnerf_hip.so, the MI355X (gfx950) implementation of OpenObj's
 * vectorised object-NeRF hot path.
 *
 * The reference (BIT-DYN/OpenObj, objnerf/) has no FFI / operator layer of its own: its hot path
 * is a Python call sequence (train.py:394-474).  This header is the boundary a maintainer binds
 * that call sequence to; each entry point cites the reference interface it replaces.  All
 * pointers are DEVICE pointers unless named host_*; all tensors are contiguous row-major fp32
 * unless stated; outputs and workspaces are caller-allocated; nothing here allocates, frees or
 * synchronises, and every launch goes to `stream` (a hipStream_t passed as void*) -- with ONE explicit
 * exception: an objnerf_context (below) holds helper streams and events the caller creates and hands to
 * objnerf_train_step so that the layer-wise path can run independent GEMMs side by side; its work is
 * forked from and joined back into `stream` with events, so `stream` order still describes completion.
 * Kernel attributes (the opt-in to > 64 KB LDS) are set once per device, thread-safely, at first use.  Return value:
 * 0 on success, a negative OBJNERF_E* code otherwise (the reference's assert / exit(-1) paths).
 *
 * Symbols (paths relative to the reference's objnerf/):
 *   K objects, R rays per object, S samples per ray, H hidden width, C feature width (512),
 *   E1=87 / E2=42 embedding split (trainer.py:20-21).
 *
 * Parameter arena.  The reference stacks the K networks into 18+1 tensors with
 * functorch.combine_state_for_ensemble (utils.py:55-62).  Here the same values live in ONE
 * object-major fp32 arena [K][P_stride]: object k's 19 tensors back to back in the order of
 * OccupancyMap.parameters() (model.py:29-56) followed by UniDirsEmbed.B_layer.weight
 * (embedding.py:39-40).  objnerf_param_layout() returns the offsets; the stacked tensors the
 * reference API exposes are strided views of the arena.  Gradients and Adam moments use the same
 * layout.
 */
#ifndef OBJNERF_HIP_H
#define OBJNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBJNERF_ABI_VERSION 7

#define OBJNERF_OK 0
#define OBJNERF_EINVAL (-22)       /* bad shape / null pointer / unsupported size        */
#define OBJNERF_ENOTSUP (-95)      /* shape valid for the reference but not built here   */
#define OBJNERF_ELAUNCH (-5)       /* HIP launch error (hipGetLastError != hipSuccess)   */

#define OBJNERF_N_TENSORS 19       /* 18 OccupancyMap tensors + B_layer.weight           */

/* Index of each tensor inside an object's block of the arena (model.py:29-56 order). */
enum objnerf_tensor {
  OBJNERF_T_IN_W = 0, OBJNERF_T_IN_B, OBJNERF_T_M1_W, OBJNERF_T_M1_B, OBJNERF_T_CAT_W,
  OBJNERF_T_CAT_B, OBJNERF_T_M2_W, OBJNERF_T_M2_B, OBJNERF_T_ALPHA_W, OBJNERF_T_ALPHA_B,
  OBJNERF_T_CL_W, OBJNERF_T_CL_B, OBJNERF_T_OC_W, OBJNERF_T_OC_B, OBJNERF_T_FL_W, OBJNERF_T_FL_B,
  OBJNERF_T_OF_W, OBJNERF_T_OF_B, OBJNERF_T_PE_B
};

/* Network shape: trainer.py:15-21,36-44. */
typedef struct objnerf_net {
  int32_t hidden;        /* cfg.hidden_feature_size (32 objects / 128 background)         */
  int32_t feat_dim;      /* cfg.clip_point_feature_size (512)                             */
  int32_t n_freqs;       /* cfg.n_unidir_funcs + 1 (6)                                    */
  int32_t reserved;
} objnerf_net;

/* offsets[i] = start of tensor i inside an object's block, offsets[19] = parameter count P;
 * returns P_stride (P rounded up to 64 floats), or a negative error. */
int64_t objnerf_param_layout(const objnerf_net* net, int64_t offsets[OBJNERF_N_TENSORS + 1]);

int objnerf_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * A1  cameraInfo.get_rays_dirs (vmap.py:701-720): out [W][H][3], un-normalised, transposed image.
 */
int objnerf_rays_dirs(int32_t W, int32_t H, float fx, float fy, float cx, float cy, float* out,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * A2+A3+A4  sceneObject.get_training_samples + sample_3d_points (vmap.py:386-554,
 * utils.py:324-397) for ONE object: gather pixels from the keyframe buffers and place the
 * depth-guided z-values.
 * Randomness, INJECTED form (exact parity with the reference's torch.rand / normal_ draws):
 *   kf_ids [n_frames] int64, u_w,u_h [n_frames][n_px] in [0,1),
 *   u [n_frames*n_px][N+M] uniforms, g [n_frames*n_px][M] draws of N(0,(eps/3)^2).
 * Randomness, SEEDED form (u_w = u_h = u = g = NULL): every draw is Philox4x32-10 of
 *   (seed; purpose, draw; object index, ray, bin) computed in place -- no random number is ever stored.  `draw` is
 *   the caller's call counter (a new value = new draws, 29 bits), the object index identifies the object's random
 *   stream (kf_meta[3]; without kf_meta: obj_index, + k for object k of a stacked call), so that objects,
 *   iterations and GPUs holding different objects never share a draw.
 *   kf_ids may then be NULL too: kf_meta [4] int32 = {n_keyframes, slot of the second-latest keyframe, slot of the
 *   latest (-1, -1 while n_keyframes <= 2), object index} selects the keyframes as vmap.py:390-401 does (uniform
 *   over the stored keyframes, the latest two always included, LAST).  out_kf [n_frames] int64 / out_px [n][2] int32 (pixel
 *   (w, h) of each ray) are optional records of what was drawn (the part-feature gather of vmap.py:437-452 needs
 *   them).
 * Keyframe buffers: rgbs [F][W][H][4] u8 (rgb+state), depth [F][W][H], t_wc [F][4][4],
 * bbox [F][4] = [u lo,u hi,v lo,v hi], rays_dir_cache [W][H][3].
 * Outputs: rgb [n][3] u8, gt_depth [n], valid [n] u8, labels [n] u8, z [n][N+M], pts [n][N+M][3]
 * (n = n_frames*n_px).  out_pts may be NULL when out_origins / out_dirs [n][3] are given: the world-frame ray
 * origins and directions go there and objnerf_train_step (pts == NULL form) forms the points in registers --
 * the [n][N+M][3] point tensor is then never written or read.
 * max_depth_ws: device scratch of 1 + 6*n floats ([0] receives the batch depth
 * maximum of vmap.py:489, the rest holds the world-frame origins / directions between the passes).
 */
typedef struct objnerf_sample_args {
  int32_t F, W, H, n_frames, n_px, n_cam2surf, n_bins, obj_index;
  float surface_eps, stop_eps, min_bound, obj_center;
  const uint8_t* rgbs; const float* depth; const float* t_wc; const float* bbox;
  const float* rays_dir_cache;
  const int64_t* kf_ids; const float* u_w; const float* u_h; const float* u; const float* g;
  uint8_t* out_rgb; float* out_depth; uint8_t* out_valid; uint8_t* out_labels;
  float* out_z; float* out_pts; float* max_depth_ws;
  /* ABI 3 */
  uint64_t seed; uint32_t draw; uint32_t reserved;
  const int32_t* kf_meta; int64_t* out_kf; int32_t* out_px;
  float* out_origins; float* out_dirs;
  /* ABI 6 -- the part-level feature gather of vmap.py:437-452 inside the same launch (out_partfeat == NULL: off).
   * global_partfeat [pf_frames][pf_w][pf_h][pf_c] fp32: the scene's part-feature maps (train.py:378), one per used
   * dataset frame; use_frame [F] int32: dataset frame id of every keyframe slot (vmap.py:108,199-231; stacked call:
   * [K][F]); the ray's source row is frame trunc(use_frame[kf] / pf_stride) (float64 quotient, :440), pixel
   * (floor(idx_w / part_down), floor(idx_h / part_down)) with the FLOAT pixel index divided in fp32 (:441-442).
   * out_partfeat [n][pf_c] (stacked: [K][n][pf_c]) is written once, a wave per ray, 16-byte lanes.  Indices outside
   * the map are clamped (the reference raises IndexError; the host wrapper checks the frame range). */
  const float* global_partfeat; const int32_t* use_frame; float* out_partfeat;
  int32_t pf_frames, pf_w, pf_h, pf_c, pf_stride; float part_down;
} objnerf_sample_args;
int objnerf_sample_rays(const objnerf_sample_args* a, void* stream);

/* The same for K objects in ONE launch chain (train.py:316-330 calls the sampler once per object and stacks the
 * results, train.py:368-388).  `table`: DEVICE array of K keyframe-store descriptors (the objects share F, W, H and
 * every scalar of `a`; a->rgbs / depth / t_wc / bbox are ignored).  Every draw and output array of `a` (kf_meta,
 * out_kf, out_px, out_origins, out_dirs included) is stacked [K][...] with the per-object layouts above -- the outputs ARE the stacked batch tensors of the training step --
 * and max_depth_ws holds K x (1 + 6 n) floats. */
typedef struct objnerf_kf_store {
  const uint8_t* rgbs; const float* depth; const float* t_wc; const float* bbox;
} objnerf_kf_store;
int objnerf_sample_rays_stacked(const objnerf_sample_args* a, int32_t K, const objnerf_kf_store* table, void* stream);

/* A4 alone (ABI 7): sceneObject.sample_3d_points(sampled_rgbs, sampled_depth, origins, dirs_w) (vmap.py:456-554) on
 * pixels the caller has already gathered: sampled_rgbs [n_frames][n_px][4] u8 (rgb + state, vmap.py:421),
 * sampled_depth [n_frames][n_px], origins [n_frames][3], dirs_w [n_frames][n_px][3] (utils.origin_dirs_W).  Of `a` the
 * scalars n_frames, n_px, n_cam2surf, n_bins, surface_eps, stop_eps, min_bound, obj_center, obj_index, seed, draw, the
 * draws u / g (both or neither: neither = the seeded form) and the outputs out_rgb, out_depth, out_valid, out_labels,
 * out_z, out_pts (or out_origins + out_dirs [n][3]) and max_depth_ws (1 + 6 n floats) are used; everything that
 * describes the keyframe store is ignored.  sampled_partfeat is passed through by the host (the reference returns its
 * argument, vmap.py:554). */
int objnerf_sample_points(const objnerf_sample_args* a, const uint8_t* sampled_rgbs, const float* sampled_depth,
                          const float* origins, const float* dirs_w, void* stream);

/* ------------------------------------------------------------------------------------------
 * f-3  Frame ingestion (train.py:196-256 + sceneObject.__init__ / append_keyframe, vmap.py:29-257): one new frame is
 * written into a keyframe slot of EVERY object that is visible in it, in one launch.  Per object the reference
 * builds a state map from the instance image (1 = this object, 2 = unknown (-1), 0 = other; train.py:201-203) and
 * copies rgb + state, depth, the camera pose and the 2-D box into the slot.
 *   rgb [W][H][3] u8, depth [W][H], inst [W][H] int32, t_wc [16]: the frame (device, images stored transposed)
 *   items: DEVICE array of K records: the object's four store tensors (layouts of objnerf_sample_rays), the slot
 *   to write, the instance id that means "this object" and its 2-D box.  Slot selection (keyframe / live frame /
 *   pruning, vmap.py:166-257) is host bookkeeping and stays with the caller.
 */
typedef struct objnerf_ingest_item {
  uint8_t* rgbs; float* depth; float* t_wc; float* bbox;
  int32_t slot, obj_id;
  float box[4];
} objnerf_ingest_item;
int objnerf_ingest_frame(int32_t W, int32_t H, const uint8_t* rgb, const float* depth, const int32_t* inst,
                         const float* t_wc, int32_t K, const objnerf_ingest_item* items, void* stream);

/* ------------------------------------------------------------------------------------------
 * f-1  Trainer.sample_points_bbox (trainer.py:130-198), the sampler of render_2D_syn (vmap.py:604-685).
 * objnerf_box_rays: P camera rays dirs_C [P][3] (un-normalised, rays_dir_cache[pixels]) of ONE view against an
 *   oriented box: T_WC, T_OC = inverse(T_WO) @ T_WC (both [4][4] row-major, computed by the caller as
 *   trainer.py:152-157 does), half = extent / 2.  Outputs: dirs_W [P][3] (origin_dirs_W, utils.py:324-336),
 *   near [P] (clipped at 0), far [P] (+0.2, :166-167), hit [P] u8 (ray_box_intersection, utils.py:309-319).
 * objnerf_box_points: for the n hit rays (compacted by the caller): z_vals [n][n_bins-1] = mid-points of
 *   stratified_bins(near, far, n_bins) with the injected draw u [n][n_bins] (utils.py:342-379, trainer.py:171-175;
 *   u == NULL: Philox draws of (seed; draw; ray, bin) generated in place) and pts [n][n_bins-1][3] = origin +
 *   dirs_W * z (:176).
 */
int objnerf_box_rays(int64_t P, const float* T_WC, const float* T_OC, const float* half_extent,
                     const float* dirs_C, float* out_dirs_W, float* out_near, float* out_far,
                     uint8_t* out_hit, void* stream);
int objnerf_box_points(int64_t n, int32_t n_bins, const float* origin /* [3] */, const float* dirs_W,
                       const float* near, const float* far, const float* u, uint64_t seed, uint32_t draw,
                       float* out_z, float* out_pts, void* stream);

/* f-1 in ONE launch (ABI 6): sceneObject.render_2D_syn's per-object chain (vmap.py:644-676) for the n hit rays of
 * objnerf_box_rays -- the 149 mid-points of Trainer.sample_points_bbox (trainer.py:171-176; u [n][n_bins] injected, or
 * NULL: the Philox draws of objnerf_box_points under (seed, draw)), UniDirsEmbed + OccupancyMap (embedding.py:46-55,
 * model.py:61-103), occupancy_activation / occupancy_to_termination / render (render_rays.py:6-63) -- with a lane per
 * ray and nothing per sample in HBM.  params: ONE object's block, scale [1].  Outputs per ray: depth, opacity, rgb [3]
 * and (out_hfeat != NULL) the composited hidden of the feature branch [n][H]: the 512-d map is objnerf_feature_head of
 * it with weight = opacity (exact: that head is linear).  out_z [n][n_bins-1] optional (the z_vals of the call).
 * Hidden 32 only (OBJNERF_ENOTSUP otherwise: wider networks keep objnerf_box_points -> objnerf_eval_points_ws ->
 * objnerf_composite).  mode: 0 = the reference's fp32 arithmetic (1e-4 parity, fixture G11); OBJNERF_TRAIN_BF16 = opt-in:
 * the operands of every hidden nn.Linear rounded to bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulation), sin / cos /
 * sigmoid on the transcendental unit -- the arithmetic of the bf16 training mode, fp32 compositing. */
int objnerf_render_fwd(const objnerf_net* net, int64_t n, int32_t n_bins, const float* params, const float* scale,
                       const float* origin /* [3] */, const float* dirs_W, const float* near, const float* far,
                       const float* u, uint64_t seed, uint32_t draw, float* out_depth, float* out_opacity,
                       float* out_rgb, float* out_hfeat, float* out_z, int32_t mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * A6+A7  UniDirsEmbed.forward + OccupancyMap.forward (embedding.py:46-55, model.py:61-103),
 * inference: K objects x N points each.
 *   params [K][P_stride], scale [K] (pe buffer `scale`), pts [K][N][3]
 *   out_alpha [K][N] (= 10*raw, model.py:88), out_color [K][N][3] (sigmoid applied),
 *   out_hfeat [K][N][H] = relu(clip_linear(.)) or NULL, out_clip [K][N][C] = out_clip(.) or NULL.
 * This is what Trainer.eval_points (trainer.py:105-128) and render_2D_syn (vmap.py:644-655) run.
 */
int objnerf_eval_points(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                        int64_t p_stride, const float* scale, const float* pts, float* out_alpha,
                        float* out_color, float* out_hfeat, float* out_clip, void* stream);

/* The same for ANY hidden width that is a multiple of 32 (the background network of trainer.py:15-17 is 128
 * wide and is rendered by the same render_2D_syn, vmap.py:644-655): hidden 32 runs the fused kernel above and
 * needs no workspace; wider networks run layer by layer with activations in the caller's workspace of
 * objnerf_eval_workspace_bytes(net, K, N) bytes (256-byte aligned). */
size_t objnerf_eval_workspace_bytes(const objnerf_net* net, int32_t K, int64_t N);
int objnerf_eval_points_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                           int64_t p_stride, const float* scale, const float* pts, float* out_alpha,
                           float* out_color, float* out_hfeat, float* out_clip, void* workspace,
                           size_t workspace_bytes, void* stream);

/* A7 alone: OccupancyMap.forward on a caller-supplied embedding emb [K][N][129] (model.py:61-103, the
 * call vmap(fc_model)(fc_param, fc_buffer, batch_embedding) of train.py:425).  Outputs as above. */
int objnerf_mlp_forward(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                        int64_t p_stride, const float* emb, float* out_alpha, float* out_color,
                        float* out_hfeat, float* out_clip, void* stream);

/* The same for any hidden width that is a multiple of 32 (the reference calls scene_bg.trainer.fc_occ_map(bg_embedding)
 * on the 128-wide background network, train.py:449-450): hidden 32 runs the fused kernel and needs no workspace, wider
 * networks the layer-wise chain with objnerf_eval_workspace_bytes(net, K, N) bytes of caller workspace. */
int objnerf_mlp_forward_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                           int64_t p_stride, const float* emb, float* out_alpha, float* out_color,
                           float* out_hfeat, float* out_clip, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ABI 5 -- the BACKWARD halves of the two mirrored modules, for a caller that keeps the reference's loop body
 * (train.py:424-436: vmap(pe_model) -> vmap(fc_model) -> step_batch_loss -> loss.backward()) instead of the fused
 * objnerf_train_step: what autograd runs through model.py:61-103 and embedding.py:46-55.  fp32, any hidden width that
 * is a multiple of 32.
 *   emb      [K][N][129]  the embedding the forward entry was given (activations are recomputed from it),
 *   d_alpha  [K][N]       dL / d alpha   (alpha = the 10x-scaled occupancy logit objnerf_mlp_forward returns),
 *   d_color  [K][N][3]    dL / d color   (post-sigmoid),
 *   d_clip   [K][N][C]    dL / d out_clip, or NULL: the feature branch (tensors 14..17) then receives no gradient
 *                         and its slots of `grads` are left untouched,
 *   grads    [K][p_stride]  gradient arena in objnerf_param_layout order, tensors 0..13 (0..17 with d_clip) overwritten,
 *   d_emb    [K][N][129]  dL / d emb, overwritten (feed it to objnerf_embed_bwd).
 * workspace: objnerf_mlp_backward_workspace_bytes(net, K, N, d_clip != NULL) bytes, 256-byte aligned. */
size_t objnerf_mlp_backward_workspace_bytes(const objnerf_net* net, int32_t K, int64_t N, int32_t with_clip);
int objnerf_mlp_backward_ws(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                            int64_t p_stride, const float* emb, const float* d_alpha, const float* d_color,
                            const float* d_clip, float* grads, float* d_emb, void* workspace,
                            size_t workspace_bytes, void* stream);

/* d B [K][21][3] of UniDirsEmbed.B_layer.weight (embedding.py:36-38) from d_emb [K][N][129] and the inputs of
 * objnerf_embed; scratch: K * 64 floats.  (The reference differentiates neither the points nor the scale.) */
int objnerf_embed_bwd(const objnerf_net* net, int32_t K, int64_t N, const float* params, int64_t p_stride,
                      const float* scale, const float* pts, const float* d_emb, float* d_B, float* scratch,
                      void* stream);

/* A6 alone: emb [K][N][3+21*n_freqs]. */
int objnerf_embed(const objnerf_net* net, int32_t K, int64_t N, const float* params,
                  int64_t p_stride, const float* scale, const float* pts, float* out_emb,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * A8+A9+A10  occupancy_activation -> occupancy_to_termination -> render (render_rays.py:6-63,
 * loss.py:27-35,82) for n_rays rays of S samples:
 *   alpha [n][S], color [n][S][3], z [n][S], vals [n][S][V] or NULL (any per-sample vector to
 *   composite: the H-wide feature hidden, or the reference's C-wide clip tensor).
 *   out_term [n][S] or NULL, out_depth/out_var/out_opacity [n], out_rgb [n][3], out_vals [n][V].
 */
#define OBJNERF_COMPOSITE_INPUT_IS_OCCUPANCY 1   /* `alpha` already holds sigmoid(alpha) (render_rays.py:32) */
int objnerf_composite(int64_t n_rays, int32_t S, int32_t flags, const float* alpha, const float* color,
                      const float* z, const float* vals, int32_t V, float* out_term,
                      float* out_depth, float* out_var, float* out_rgb, float* out_opacity,
                      float* out_vals, void* stream);

/* Measurement support (bench.py `roofline.peak_measured`; nothing in the reference): a saturated-MFMA loop, n_wg
 * workgroups of 4 waves x iters x 4 MFMAs on two accumulator chains per wave; dtype 0: v_mfma_f32_32x32x2_f32 (4096 FLOP
 * each), 1: v_mfma_f32_32x32x16_bf16 (32768 FLOP each) -- the instructions of the training kernels.  sink: n_wg * 256
 * floats. */
int objnerf_mfma_peak(int32_t dtype, int32_t iters, int32_t n_wg, float* sink, void* stream);

/* A8 alone: occupancy_activation(alpha) = sigmoid(alpha), n elements (render_rays.py:6-14). */
int objnerf_occupancy(int64_t n, const float* alpha, float* out, void* stream);

/* out_clip head applied after compositing (exact: the head is linear):
 * out [K][n][C] = of_w[k] . hfeat[k][n] + of_b[k] * weight[k][n]   (weight = opacity, or NULL=1). */
int objnerf_feature_head(const objnerf_net* net, int32_t K, int64_t n, const float* params,
                         int64_t p_stride, const float* hfeat, const float* weight, float* out,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * A11  loss.step_batch_loss (loss.py:5-103) on materialised network outputs:
 *   alpha [K][R][S], color [K][R][S][3], z [K][R][S], gt_depth [K][R], gt_rgb [K][R][3],
 *   labels [K][R] u8 (0 other / 1 this / 2 unknown), optional pred_feat [K][R][S][C] + gt_feat
 *   [K][R][C].  Scalings as loss.py:6.
 * Outputs: loss_terms [K][4] = per-object (depth, colour, opacity, feature) means, total [1],
 *   optional d_alpha / d_color / d_pred_feat (gradients of `total`), status [1] int32: bit 0 when a
 *   per-object term exceeds 1e5 (render_rays.py:109-111 "loss explode": the reference exits), bit 1 when a term is
 *   not finite (the reference's `> 100000` test is False for NaN and it carries on).  The same word, with the same
 *   two bits, is written by every entry point that has a `status` argument.
 * The cross-object early return (render_rays.py:89-94) is applied: if ANY object has no label-1
 * ray the depth / colour / feature terms are zero for ALL objects, likewise label!=2 for opacity.
 * counts: int32 workspace of 2*K + 2 entries; receives [K][2] = (n_label1, n_label_not2) and the two
 * early-return flags.
 * flags_in: optional [2] int32 "some object elsewhere (other GPU) had an empty mask".
 */
typedef struct objnerf_loss_args {
  int32_t K, R, S, C;
  float color_scaling, opacity_scaling, feat_scaling, reserved;
  const float* alpha; const float* color; const float* z; const float* gt_depth;
  const float* gt_rgb; const uint8_t* labels; const float* pred_feat; const float* gt_feat;
  const int32_t* flags_in;
  const int32_t* counts_in;   /* optional [K][2]: use these (e.g. all-reduced over GPUs that split one object's rays)
                                 instead of counting this call's labels */
  float* loss_terms; float* total; float* d_alpha; float* d_color; float* d_pred_feat;
  int32_t* counts; int32_t* status;
} objnerf_loss_args;
int objnerf_step_batch_loss(const objnerf_loss_args* a, void* stream);

/* Label statistics alone: counts [K][2], flags_out [2] (1 = some object has an empty mask). */
int objnerf_label_counts(int32_t K, int32_t R, const uint8_t* labels, int32_t* counts,
                         int32_t* flags_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * A12  one training iteration, fused: train.py:424-472 (vmap(pe) -> vmap(fc) -> step_batch_loss
 * -> backward) for K stacked objects in ONE pass over the rays.
 *
 * Inputs: params/scale as above; either pts [K][R][S][3] (train.py:397 batch_input_pcs) or, when
 * pts == NULL, origins [K][R][3] + dirs [K][R][3] + z (pts = o + d*z - center, vmap.py:548-551);
 * z [K][R][S]; gt_depth, gt_rgb, labels as objnerf_step_batch_loss; gt_feat [K][R][C] or NULL
 * (NULL = cfg.part_mode off: the feature branch gets no gradient, train.py:435-438).
 * counts [K][2] + flags [2]: from objnerf_label_counts (flags may have been max-reduced across
 * GPUs first -- the early return spans every object of the batch).
 * Outputs: grads [K][P_stride] (same layout as params; entries of tensors without gradient are
 * left untouched), loss_terms [K][4], status [1] (bits as objnerf_step_batch_loss: 0 = a term above 1e5, 1 = a term
 * that is not finite; every path of this entry point -- fused hidden 32, one-launch hidden 128, layer-wise, fused
 * hidden 256 -- writes both).
 * workspace: objnerf_train_workspace_bytes() bytes, 256-byte aligned (its with_feat argument: bit 0 = feature
 * loss, bit 1 = size for OBJNERF_TRAIN_LAYERWISE, bit 2 = size for the 16-bit modes only -- at hidden 256 they keep
 * the five activation and five gradient buffers in the operand type, a third less; a workspace sized without bit 2 serves every mode).
 */
#define OBJNERF_TRAIN_BF16 1   /* mode bit: MFMA operands rounded to bf16 (fp32 accumulate, fp32 master weights,
                                * fp32 embedding/compositing/losses).  NOT the reference's arithmetic (fp32,
                                * train.py:74) -- an opt-in throughput mode gated by PSNR.  Fused kernel: hidden 32,
                                * S <= 64, no feature loss (else OBJNERF_ENOTSUP); layer-wise path (other widths,
                                * longer rays): bf16 GEMM operands, any configuration. */
#define OBJNERF_TRAIN_FP16 4   /* mode bit: the layer-wise path with fp16 GEMM operands (v_mfma_f32_16x16x32_f16), fp32
                                * accumulate, fp32 master weights / activations / losses (BASELINE configs[4] names
                                * fp16).  Any width and ray length; implies the layer-wise path.  Operands saturate at
                                * +-65504 and the backward GEMMs scale their gradient operand by ~8 R (a power of two)
                                * so that it stays in fp16's normal range.  PSNR-gated like OBJNERF_TRAIN_BF16; not
                                * together with it (OBJNERF_EINVAL). */
#define OBJNERF_TRAIN_LAYERWISE 2   /* mode bit: take the layer-wise (any width) path even for hidden 32 / S <= 64 --
                                     * a second, independent implementation of the same iteration; the tests use
                                     * it to cross-check the fused kernel at sizes no CPU oracle reaches. */
#define OBJNERF_TRAIN_SELF_COUNTS 8   /* mode bit (ABI 7): the step derives the label statistics itself -- `counts` [K][2] and
                                       * `flags` [2] are then OUTPUTS (n(label == 1), n(label != 2) per object; the two
                                       * early-return flags of render_rays.py:89-94 over the K objects of THIS call), written
                                       * before anything of the step reads them: no objnerf_label_counts launch by the caller.
                                       * For an un-sharded batch only (under object sharding the flags span every rank's
                                       * objects and the background's counts every rank's rays: reduce them and pass them in). */
/* Helper streams + events for objnerf_train_step (see the preamble): create on the device that will run the steps,
 * destroy when no step using it is in flight.  The only entries of the library that create or free anything. */
struct objnerf_context;
int objnerf_context_create(struct objnerf_context** out);
int objnerf_context_destroy(struct objnerf_context* ctx);

/* ABI 7 -- the optimiser step of the iteration (train.py:472-473: optimiser.step() right after loss.backward()) inside
 * objnerf_train_step: the launch that reduces the partial gradients applies torch.optim.AdamW to the element it has
 * just summed (objnerf_adamw_step_flags' arithmetic, the same flag-dependent skipping and per-group step counters), so
 * an iteration has no separate optimiser launch and the gradient is not read back from HBM.  `grads` is still written.
 * group_steps / bank as in objnerf_adamw_step_flags (the caller alternates `bank`).  Tensors that receive no gradient
 * in the call's configuration (the feature branch without gt_feat) are skipped like a None .grad. */
typedef struct objnerf_adamw_args {
  float* exp_avg; float* exp_avg_sq;      /* [K][p_stride] like params */
  int32_t* group_steps; int32_t bank; int32_t reserved;
  float lr, beta1, beta2, eps, weight_decay, reserved_f;
} objnerf_adamw_args;

typedef struct objnerf_train_args {
  int32_t K, R, S, mode;
  float color_scaling, opacity_scaling, feat_scaling, obj_center;
  const float* params; int64_t p_stride; const float* scale;
  const float* pts; const float* origins; const float* dirs; const float* z;
  const float* gt_depth; const float* gt_rgb; const uint8_t* labels; const float* gt_feat;
  const int32_t* counts; const int32_t* flags;
  float* grads; float* loss_terms; int32_t* status;
  void* workspace; size_t workspace_bytes;
  uint8_t* relu_masks;   /* NULL in production.  Test hook, [K][R][S][6][hidden / 8] bytes: bit (f & 7) of byte
                          * f >> 3 of layer l (h1, h2, h3, h4, colour hidden, feature hidden -- the last only with
                          * gt_feat) is set iff that ReLU passed its input for the sample.  The parity tests evaluate
                          * the oracle on the SAME branches (an fp32 implementation and the reference legitimately
                          * disagree about inputs within rounding of zero; tests/parity_util.py).  Not with
                          * OBJNERF_TRAIN_BF16. */
  struct objnerf_context* context;   /* NULL: everything on `stream`.  Else the layer-wise path (hidden != 32,
                                      * S > 64, OBJNERF_TRAIN_LAYERWISE) forks its weight-gradient GEMMs and the feature
                                      * preparation onto the context's streams; one context serves one call at a time. */
  /* ABI 4 */
  float* emb_debug;      /* NULL in production.  Test hook, [K][R][S][129]: the fused fp32 kernel (hidden 32, S <= 64)
                          * writes the embedding rows its tiles computed IN REGISTERS (embedding.py:46-55 in the
                          * reference's column order) -- the kernel never materialises them otherwise, so this is the
                          * only way to compare its positional encoding with fixture G1.  Every other path refuses it
                          * (OBJNERF_ENOTSUP): their embedding is objnerf_embed's output. */
  /* ABI 7 */
  const objnerf_adamw_args* optim;   /* NULL: gradients only.  Else `params` (declared const for the gradient-only
                          * form) is UPDATED in place by the step's last launch, after every kernel that reads it. */
} objnerf_train_args;
size_t objnerf_train_workspace_bytes(const objnerf_net* net, int32_t K, int32_t R, int32_t S,
                                     int32_t with_feat);
int objnerf_train_step(const objnerf_net* net, const objnerf_train_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * A12  torch.optim.AdamW.step as train.py:78 configures it (lr, weight_decay; betas .9/.999,
 * eps 1e-8), over the arena.  has_grad [P] u8: 0 for tensors that received no gradient this
 * iteration (they get NO update and NO decay, like a None .grad).  step = 1-based step count.
 */
int objnerf_adamw_step(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads,
                       float* exp_avg, float* exp_avg_sq, const uint8_t* has_grad, int32_t step,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       void* stream);

/* The same step with the early-return quirk of render_rays.py:89-94 carried through to the optimiser: `flags` is the
 * iteration's device flag pair (objnerf_label_counts / the all-reduced flags).  flags[0] set = the depth, colour and
 * feature terms were constants this iteration: the colour branch [colour_lo, feature_lo) and the feature branch
 * [feature_lo, feature_hi) of every object had .grad = None in the reference, and torch.optim.AdamW skips such
 * parameters entirely (no decay, no moment update, no step increment); both flags set = nothing is updated.
 * group_steps: device int32[2][3]: two banks of per-group step counters (trunk + density head + B | colour | feature),
 * zero-initialised by the caller.  The call reads bank `bank` (0 / 1) and writes the advanced counters to the other
 * one: the caller alternates `bank` from call to call, starting with 0 (one launch, no counter is read after it moved).
 * has_grad still masks what a configuration never differentiates. */
int objnerf_adamw_step_flags(int32_t K, int64_t P, int64_t p_stride, float* params, const float* grads, float* exp_avg,
                             float* exp_avg_sq, const uint8_t* has_grad, const int32_t* flags, int32_t* group_steps,
                             int32_t bank, int64_t colour_lo, int64_t feature_lo, int64_t feature_hi, float lr,
                             float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------
 * Helper functions of the reference's call surface that a caller may use outside the fused iteration
 * (objnerf_helpers.hip).  All take / return device pointers; small HBM-bound kernels.
 */
/* render_rays.render (render_rays.py:56-63): out [n_rays][C] = sum over the S samples of termination [n_rays][S] *
 * vals [n_rays][S][C] (C = 1 for depth-like quantities). */
int objnerf_render(int64_t n_rays, int32_t S, int32_t C, const float* termination, const float* vals, float* out,
                   void* stream);

/* render_rays.render_loss (render_rays.py:65-83).  mode 0 "L1": |render - gt|, 1 "L2": squared, n elements;
 * mode 2 "cos": 1 - cosine_similarity over rows of C entries (norms clamped at 1e-8), n rows.  normalise: / gt. */
int objnerf_render_loss(int64_t n, int32_t C, int32_t mode, int32_t normalise, const float* render, const float* gt,
                        float* out, void* stream);
/* render_rays.reduce_batch_loss (render_rays.py:85-117): loss_mat [K][R], var [K][R] or NULL (information weight
 * 1 / (sqrt(var) + 1e-4), or 1 / (var + 1e-4) with l2), mask [K][R] u8.  avg: out [K] = masked means, all zero when
 * ANY object's mask is empty (:89-94); status bit 0 set when a mean exceeds 1e5 (:109-111, the reference exits), bit 1
 * when a mean is not finite.
 * avg == 0: out [K][R] = the weighted matrix.  counts_ws: K + 1 ints of scratch. */
int objnerf_reduce_batch_loss(int32_t K, int32_t R, const float* loss_mat, const float* var, const uint8_t* mask,
                              int32_t l2, int32_t avg, int32_t* counts_ws, float* out, int32_t* status, void* stream);
/* render_rays.make_3D_grid (render_rays.py:119-146): out [dim][dim][dim][3] = lattice on [lo, hi]^3, optionally
 * scaled per axis (scale [3]) and moved by transform [4][4] (row-major). */
int objnerf_make_grid(int32_t dim, float lo, float hi, const float* scale, const float* transform, float* out,
                      void* stream);
/* utils.ray_box_intersection (utils.py:309-319): n rays against the axis-aligned box [bmin, bmax] ([3] each). */
int objnerf_ray_box(int64_t n, const float* origins, const float* dirs, const float* bmin, const float* bmax,
                    float* out_near, float* out_far, uint8_t* out_hit, void* stream);
/* utils.origin_dirs_W (utils.py:324-336): out_dirs_W [F][P][3] = R_wc[f] dirs_C[f][p]  (T_WC [F][4][4]; the
 * origins are T_WC[:, :3, 3], a view on the caller's side). */
int objnerf_dirs_w(int64_t F, int64_t P, const float* T_WC, const float* dirs_C, float* out_dirs_W, void* stream);
/* utils.stratified_bins (utils.py:342-379): out [n_rays][n_bins] = lo + range * i / n + U * range / n; lo / hi per
 * ray or NULL (then the scalar); U = u [n_rays][n_bins] (injected draws) or, with u == NULL, the counter-based
 * generator of objnerf_philox.h under (seed, offset). */
int objnerf_stratified_bins(int64_t n_rays, int32_t n_bins, const float* lo, float lo_scalar, const float* hi,
                            float hi_scalar, const float* u, uint64_t seed, uint64_t offset, float* out, void* stream);
/* utils.normal_bins_sampling (utils.py:382-397): out [n_rays][n_bins] = depth + clip(sort(N(0, (delta/3)^2)), +-delta);
 * g = injected draws (already scaled) or NULL for the counter-based generator. */
int objnerf_normal_bins(int64_t n_rays, int32_t n_bins, const float* depth, float delta, const float* g, uint64_t seed,
                        uint64_t offset, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OBJNERF_HIP_H */
