"""Helpers of the GPU parity tests: run the oracle iteration (fp32 = the reference's arithmetic, or anchored in
fp64) and compare gradients at the stated 1e-4 bound.

Why an fp64 anchor and matched ReLU branches exist.  BASELINE.json's bar is 1e-4 (relative to a tensor's largest
entry).  An fp32 HIP kernel and the fp32 oracle both carry rounding noise of a few 1e-7 per operation; where a ReLU
input of some sample lies within that noise of zero the two sides take different branches.  That is not a 1/N effect:
the gradient of a layer is a sum over N samples with heavy cancellation, so ONE sample whose unit flipped moves the
gradients below it by 1e-4 .. 1e-2 of their maximum (measured, tools/parity_report.py on MI355X,
profiles/r02_parity_report.txt: at hidden 256, 8192 rays x 64 samples torch's own fp32 run is 5.8e-4 .. 3e-3 from the
fp64 result while the HIP path is at 1e-6; at 8192 x 128 it is the other way round; one flipped colour unit out of
4e8 costs 1.5e-3 of an object's colour-bias gradient at 4096 x 64).  fp32 against fp32, and fp32 against fp64, are
ill-posed wherever a flip occurred.  So:

* `oracle_step(..., dtype=torch.float64)` is the ANCHOR: the embedding formed in fp32 exactly as the reference does
  (the fp32 rounding of the sin argument is part of the function), everything after it in fp64.
* The training entry point has a test hook, `objnerf_train_args.relu_masks`: the ReLU branch of every unit and
  sample of the iteration.  `oracle_step(..., masks=unpack_masks(...))` evaluates the oracle on THOSE branches
  (x * mask instead of relu(x)), and `check_flips` proves the substitution is legitimate: every unit whose branch
  differs from the fp64 run's own has an input within 2e-5 of zero (fp32 rounding of an O(1) sum).  With the branches
  matched the full-size tests hold the plain 1e-4.
* `assert_grads` accepts a tensor when it is within 1e-4 of the anchor or of the fp32 oracle (parity with the
  reference's arithmetic as it runs here) -- or, only where the reference's own fp32 run is further than that from
  the anchor (compositing: 1 - occupancy is quantised to 6e-8 in fp32, which both fp32 sides share), no further
  from the anchor than twice the fp32 oracle is.
  Every reference-generated fixture (G5, G6, G10) and every small batch is held to the same rule without the hook.
"""
import torch

from oracle import objnerf_oracle as O


def _t(x, device=None, dtype=None):
    t = torch.as_tensor(x)
    if device is not None:
        t = t.to(device)
    if dtype is not None and t.is_floating_point():
        t = t.to(dtype)
    return t


FLIP_TOL = 4e-6        # (measured: the largest input of a flipped unit is 1.8e-6, profiles/r03_flip_report.txt)
# Flip COUNT.  check_flips used to bound only the magnitude of a flipped unit's input; an implementation whose
# pre-activations carried, say, 1e-5 of error everywhere would pass that and flip ten times more branches than fp32
# rounding does.  The count is bounded against the data itself: BANDS are half-widths around zero, `near[l][j]` the
# number of units of layer l whose anchor (fp64) pre-activation lies inside band j.  A unit flips only if the
# implementation's error at that unit exceeds |x|, so an implementation with error <= e everywhere flips at most
# near(e) units -- and about near(e) / 2 if its error is uniform in [-e, e].  FLIP_COUNT_BAND is the band the flip
# count must stay under (plus FLIP_COUNT_SLACK units for tiny batches).  Measured on MI355X (tools/flip_report.py,
# profiles/r03_flip_report.txt; 8.4e7 units per layer): torch's own fp32 arithmetic flips 2 - 7 units per layer, the
# layer-wise HIP path 2 - 6, the fused kernel 10 - 29 (its angle-doubled embedding octaves carry up to ~4e-6 of error),
# against 73 - 124 units within 1e-6 of zero and 21 - 44 within 3e-7.
BANDS = (3e-7, 1e-6, 3e-6, 1e-5)
FLIP_COUNT_BAND = 1e-6
FLIP_COUNT_SLACK = 4


def unpack_masks(masks_u8, hidden):
    """objnerf_train_args.relu_masks [K,R,S,6,hidden/8] uint8 -> bool [6][K,R,S,hidden] (bit f & 7 of byte f >> 3)."""
    bits = (masks_u8[..., None] >> torch.arange(8, device=masks_u8.device, dtype=torch.uint8)) & 1
    m = bits.reshape(*masks_u8.shape[:4], hidden).bool()
    return [m[:, :, :, l] for l in range(6)]


def oracle_step(fc, B, scale, b, feat, dtype=None, device="cpu", k_chunk=None, do_clip=None, masks=None):
    """One iteration of train.py:424-472 through the oracle.  fc: 18 stacked tensors, B [K,21,3], scale float or
    [K]; b: batch dict (pts, z, gt_depth, gt_rgb, labels, gt_feat).  Returns dict(loss, terms [K,4], grads [19]).
    k_chunk: run the objects in chunks (they are independent; only the early-return flags span the batch, so a
    chunked run refuses batches that would trigger them).
    masks: forced ReLU branches (unpack_masks); the result then also has "flips": per layer the number of units whose
    branch differs from this run's own x > 0 and the largest |x| among them (check_flips)."""
    K = B.shape[0]
    dev = torch.device(device)
    fcr = [_t(p, dev).clone().requires_grad_(True) for p in fc]
    Br = _t(B, dev).clone().requires_grad_(True)
    sc = torch.full((K,), float(scale), device=dev) if not torch.is_tensor(scale) else scale.to(dev).float()
    keys = ["pts", "gt_depth", "gt_rgb", "labels", "z"] + (["gt_feat"] if feat else [])
    tb = {k: _t(b[k], dev) for k in keys}
    if do_clip is None:
        do_clip = bool(feat)
    k_chunk = k_chunk or K
    if k_chunk < K:
        lab = tb["labels"]
        assert bool(((lab == 1).sum(1) > 0).all()) and bool(((lab != 2).sum(1) > 0).all()), \
            "chunked oracle run needs a batch without early return"
    terms = torch.zeros(K, 4, dtype=torch.float64)
    total = 0.0
    n_layers = 6 if (feat or do_clip) else 5
    flips = [[0, 0.0] for _ in range(n_layers)]
    near = [[0] * len(BANDS) for _ in range(n_layers)]
    for k0 in range(0, K, k_chunk):
        sl = slice(k0, min(K, k0 + k_chunk))
        mk = None if masks is None else [m[sl].to(dev) for m in masks[:n_layers]]
        loss, t = O.train_forward_loss([p[sl] for p in fcr], Br[sl], sc[sl], tb["pts"][sl], tb["gt_depth"][sl],
                                       tb["gt_rgb"][sl], tb["labels"][sl], tb["z"][sl],
                                       gt_feat=tb["gt_feat"][sl] if feat else None, return_terms=True,
                                       mlp_dtype=dtype, do_clip=do_clip, masks=mk)
        loss.backward()
        total += float(loss.item())
        for j, name in enumerate(["depth", "color", "opacity", "feat"]):
            if t[name] is not None:
                terms[sl, j] = t[name].detach().double().cpu()
        if mk is not None:
            for l, pre in enumerate(t["pre"]):
                diff = mk[l] != (pre > 0)
                flips[l][0] += int(diff.sum())
                if bool(diff.any()):
                    flips[l][1] = max(flips[l][1], float(pre[diff].abs().max()))
                ap = pre.abs()
                for j, wd in enumerate(BANDS):
                    near[l][j] += int((ap < wd).sum())
        del loss, t
    grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu() for p in fcr + [Br]]
    none_grad = [p.grad is None for p in fcr + [Br]]
    return dict(loss=total, terms=terms, grads=grads, none_grad=none_grad, flips=flips if masks is not None else None,
                near=near if masks is not None else None)


def oracle_step_16(fc, B, scale, b, feat, operand_dtype, act16, grad_scale=1.0, round_head_weights=False,
                   device="cpu", round_head_grads=False):
    """One iteration through the SPECIFICATION of the opt-in 16-bit modes (oracle.mlp_forward_stacked_16: operands of
    every hidden nn.Linear rounded to the operand type, fp32 accumulation).  Same result dict as oracle_step."""
    K = B.shape[0]
    dev = torch.device(device)
    fcr = [_t(p, dev).clone().requires_grad_(True) for p in fc]
    Br = _t(B, dev).clone().requires_grad_(True)
    sc = torch.full((K,), float(scale), device=dev) if not torch.is_tensor(scale) else scale.to(dev).float()
    keys = ["pts", "gt_depth", "gt_rgb", "labels", "z"] + (["gt_feat"] if feat else [])
    tb = {k: _t(b[k], dev) for k in keys}
    loss, t = O.train_forward_loss(fcr, Br, sc, tb["pts"], tb["gt_depth"], tb["gt_rgb"], tb["labels"], tb["z"],
                                   gt_feat=tb["gt_feat"] if feat else None, return_terms=True, do_clip=bool(feat),
                                   operand_dtype=operand_dtype, act16=act16, grad_scale=grad_scale,
                                   round_head_weights=round_head_weights, round_head_grads=round_head_grads)
    loss.backward()
    terms = torch.zeros(K, 4, dtype=torch.float64)
    for j, name in enumerate(["depth", "color", "opacity", "feat"]):
        if t[name] is not None:
            terms[:, j] = t[name].detach().double().cpu()
    grads = [(p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu() for p in fcr + [Br]]
    return dict(loss=float(loss.item()), terms=terms, grads=grads, none_grad=[p.grad is None for p in fcr + [Br]])


def rel_norm(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_flips(o, tol=FLIP_TOL, count_band=FLIP_COUNT_BAND, slack=FLIP_COUNT_SLACK):
    """Every unit whose forced branch differs from the oracle run's own had an input within `tol` of zero, AND there
    are no more such units per layer than units whose anchor input lies within `count_band` of zero (see BANDS)."""
    jb = BANDS.index(count_band)
    for l, (n, worst) in enumerate(o["flips"]):
        assert worst < tol, ("relu branch differs for an input far from zero", "layer", l, "units", n, "max |x|", worst)
        assert n <= o["near"][l][jb] + slack, ("too many relu branch flips", "layer", l, "flips", n, "units within",
                                               count_band, "of zero", o["near"][l][jb], "bands", BANDS, o["near"][l])
    return [n for n, _ in o["flips"]]


def maxerr(a, b):
    return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def assert_grads(hip, o64, o32=None, names=None, skip=(), tol=1e-4, floor=1e-3):
    """hip: list of 19 gradient tensors (device or host); o64 / o32: oracle_step results (anchor, reference
    arithmetic).  A tensor passes when |hip - fp32 oracle| < tol * max|anchor| or |hip - anchor| < tol * max|anchor|
    -- or, only where the fp32 oracle itself is further than tol / 2 from the anchor, |hip - anchor| < 2 |fp32
    oracle - anchor|.  Returns the list of (tensor, hip error, fp32 error) that needed more than the plain bound."""
    loose = []
    for i in range(len(hip)):
        if i in skip or o64["none_grad"][i]:
            continue
        ref = o64["grads"][i]
        scale = max(floor, float(ref.abs().max()))
        e_h = maxerr(hip[i], ref)
        if e_h < tol * scale:
            continue
        name = names[i] if names else i
        assert o32 is not None, (i, name, "hip-anchor", e_h / scale)
        e_o = maxerr(o32["grads"][i], ref) if o32 is not None else 0.0
        e_ho = maxerr(hip[i], o32["grads"][i]) if o32 is not None else float("inf")
        bound = tol * scale
        if e_o > 0.5 * tol * scale:
            bound = max(bound, 2.0 * e_o)
        loose.append((name, e_h / scale, e_o / scale))
        assert e_ho < tol * scale or e_h < bound, (i, name, "hip-anchor", e_h / scale, "fp32-anchor", e_o / scale,
                                                    "hip-fp32", e_ho / scale, "bound", bound / scale)
    return loose


def assert_terms(loss_terms, o64, o32=None, feat=False, tol=1e-4):
    """Per-object loss terms [K,4] (depth, colour, opacity, feature) against the anchor at tol of the largest term.
    The depth term divides by sqrt(var) + 1e-4 (render_rays.py:96-100): a ray whose weight sits on one sample has
    var ~ 1e-8 from cancelling fp32 roundings, and its term moves by per cent between fp32 and fp64 -- there the
    reference's own fp32 arithmetic (o32) sets the floor, as in assert_grads."""
    t = torch.as_tensor(loss_terms).double().cpu()
    for j in range(4 if feat else 3):
        ref = o64["terms"][:, j]
        bound = tol * max(1.0, float(ref.abs().max()))
        if o32 is not None:
            bound = max(bound, 2.0 * float((o32["terms"][:, j] - ref).abs().max()))
        err = float((t[:, j] - ref).abs().max())
        assert err < bound, ("loss term", j, err, bound)
