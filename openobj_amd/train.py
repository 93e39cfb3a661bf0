"""The training loop of the reference's train.py:272-276,394-485 for training_strategy == "hip":
stack the object networks into the arena, run n_iter_per_frame fused iterations over slices of the
per-frame sample pool, copy the stacked parameters back.  Object creation, dataset reading, labelling
and visualisation (the rest of train.py) stay with the caller."""
import contextlib
from typing import Dict, List, Optional

import torch

from . import dist as odist
from . import ops, optim
from .render_rays import LossExplode, check_status as _check_status  # noqa: F401


class BackgroundLoop:
    """The separate background network (obj_id 0, hidden_feature_size_bg = 128, train.py:447-463).

    It is NOT in the vmap stack.  Under object sharding it is replicated on every rank: each rank trains
    on its slice of the iteration's n_per_optim_bg rays, the per-ray loss is normalised by the GLOBAL
    mask counts (a 2-int SUM all-reduce before the step), the gradient (182 339 floats at hidden 128) with the
    four loss terms appended is SUM all-reduced over RCCL in ONE collective and every rank applies the same
    AdamW update.  `begin` / `finish` split the step around that collective so that a caller can run other work
    (the object kernel) while it is in flight -- ShardedIteration does."""

    def __init__(self, cfg, bg_trainer, with_feat: bool = False, group=None, bf16: bool = False):
        """bf16: opt-in OBJNERF_TRAIN_BF16 mode (bf16 GEMM operands, fp32 accumulation and everything else)."""
        self.cfg, self.trainer, self.with_feat, self.group = cfg, bg_trainer, with_feat, group
        self.bf16 = bf16
        self.arena = bg_trainer.arena           # K = 1: trained in place, no copy-back needed
        self.arena.scale.fill_(float(bg_trainer.obj_scale))
        self.opt = optim.ArenaAdamW(self.arena, lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
        self.mask = self.arena.has_grad_mask(with_feat)
        self.ws = None

    def _workspace(self, batch):
        K, R, S = batch["z"].shape
        if self.ws is None or self.ws.key != ops.TrainWorkspace.make_key(K, R, S, self.with_feat, self.bf16):
            self.ws = ops.TrainWorkspace(self.arena, K, R, S, self.with_feat, precision=self.bf16)
            # gradient and loss terms in ONE buffer: one collective moves both
            self.flat = torch.zeros(self.arena.p_stride + 4, device=self.arena.params.device)
            self.ws.grads = self.flat[:self.arena.p_stride].view(1, -1)
            self.ws.loss_terms = self.flat[self.arena.p_stride:].view(1, 4)
        return self.ws

    def local_counts(self, batch):
        """n(label == 1), n(label != 2) of this rank's ray slice (int32 [1,2], device)."""
        return ops.label_counts(batch["labels"])[0]

    def local_counts_flags(self, batch):
        """(counts, early-return flags) of the rays at hand (objnerf_label_counts: both from one launch)."""
        return ops.label_counts(batch["labels"])

    def begin(self, batch: Dict[str, torch.Tensor], counts: Optional[torch.Tensor], flags: Optional[torch.Tensor],
              loss_out: Optional[torch.Tensor] = None):
        """Launch the step with the GLOBAL mask counts / flags and start the gradient all-reduce; returns its handle.
        counts = flags = None (one rank holds all of the iteration's background rays): the step counts the labels
        itself and, with no collective between the gradient and the optimiser, applies AdamW in its last launch
        (ops.train_step(optim=)) -- finish() then has nothing left to do."""
        ws = self._workspace(batch)
        sharded = odist._active(self.group)
        # loss_out (float32 [1, 4], un-sharded callers): this iteration's loss terms go there instead of the tail of the
        # collective's buffer -- a frame's terms land in one [n_iter, 1, 4] tensor without a copy per iteration
        ws.loss_terms = loss_out if (loss_out is not None and not sharded) else self.flat[self.arena.p_stride:].view(1, 4)
        self._fused_opt = not sharded
        self._flags = flags if flags is not None else ws.flags
        ops.train_step(self.arena, ws, batch, with_feat=self.with_feat, global_flags=flags, global_counts=counts,
                       bf16=self.bf16, optim=self.opt if self._fused_opt else None)
        return odist.allreduce_sum_async(self.flat, self.group)    # the one data-path collective

    def finish(self, work) -> torch.Tensor:
        if work is not None:
            work.wait()
        if not self._fused_opt:
            self.opt.step(self.ws.grads, self.mask, flags=self._flags)
        return self.ws.loss_terms

    def step(self, batch: Dict[str, torch.Tensor], counts: Optional[torch.Tensor] = None,
             flags: Optional[torch.Tensor] = None, loss_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """batch: THIS rank's slice of the background rays, tensors shaped [1, R_local, ...].
        counts / flags (un-sharded): label statistics the caller already has (one launch per frame, mapping.train_frame);
        None = the step counts the labels itself."""
        if not odist._active(self.group):
            return self.finish(self.begin(batch, counts, flags, loss_out=loss_out))
        counts, flags = self.local_counts_flags(batch)
        odist.allreduce_sum_(counts, self.group)              # global n(label==1), n(label!=2)
        flags = ((counts.reshape(-1, 2) == 0).any(dim=0)).to(torch.int32)
        return self.finish(self.begin(batch, counts, flags))


class ShardedIteration:
    """One iteration of the object-sharded step (train.py:424-474 over several GPUs) with TWO collectives:

      1. pre-step, ONE int32[4] SUM: the early-return flags of the stacked objects (render_rays.py:89-94 spans every
         object of the batch, wherever it lives) and the background's global mask counts (dist.pack_pre);
      2. post-step, ONE fp32 SUM of the background gradient with its loss terms appended, started right after the
         background kernels and in flight while the object kernel and its AdamW run (RCCL's stream); the
         background AdamW waits for it.

    obj_loop / bg_loop may be None (a rank without foreground objects, do_bg = 0)."""

    def __init__(self, obj_loop=None, bg_loop=None, group=None, overlap: bool = True, resident: bool = False,
                 device=None, pipelined: bool = False):
        """device: where this rank's collectives' buffers live (cfg.training_device).  Needed by a rank that is handed
        NO batch in some iteration (no foreground object yet, do_bg = 0): it still has to join both exchanges, or the
        other ranks block in them.  Defaults to the device of the first batch seen.
        overlap: the pre-step exchange and the whole background chain (its kernels, its collective's wait, its AdamW)
        run on a second HIP stream beside the object kernel -- the two are independent (own parameters, moments and
        batches; the reference only adds their losses before ONE backward, train.py:463).  The object stream waits
        for that stream twice: for the global flags before its kernel, and before step() returns.
        resident: the caller guarantees that the batches handed to step() are complete before the call (resident
        pools, as in bench.py / mapping.IncrementalMapper); the second stream then never waits for the object
        stream, so the background chain of iteration i + 1 fills the tail of object kernel i.  Otherwise it waits
        for the work queued on the caller's stream at every step().
        pipelined (with resident): step() does not make the caller's stream wait for the background stream before it
        returns.  The two chains only share the iteration's label statistics, so with resident batches the background
        chain of iteration i may still be running under the object kernel of iteration i + 1 (each chain stays in order
        on its own stream; both need whole compute units -- 160 KB of LDS per workgroup -- so what one leaves idle at
        its ragged end the other fills).  The caller calls join() before it reads the returned loss terms or the
        background parameters on its own stream."""
        self.obj_loop, self.bg_loop, self.group = obj_loop, bg_loop, group
        self.overlap, self.resident = overlap, resident
        self.pipelined = bool(pipelined and resident)
        self.device = torch.device(device) if device is not None else None
        self._side = None

    def _bg_stream(self, dev):
        if not self.overlap or dev.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def frame_pre(self, obj_labels=None, bg_labels=None, n_iter: Optional[int] = None) -> Optional[torch.Tensor]:
        """Collective 1 of ALL iterations of a frame in ONE exchange (train.py:394-404: the sample pool of a frame is
        complete before its first iteration, so every iteration's labels are known up front).
        obj_labels u8 [n_iter, K, R] / bg_labels u8 [n_iter, 1, R_bg] (either may be None; n_iter then has to be
        given): one objnerf_label_counts launch per tensor, ONE int32[n_iter, 4] SUM all-reduce (rows as
        dist.pack_pre).  Returns None without sharding (the
        kernels then derive flags and counts from the batch itself).  The result is a list: element `it` is
        (object flags int32[2], background counts int32[1,2], background flags int32[2]) for step(..., pre=)."""
        if not odist._active(self.group):
            return None
        ref = obj_labels if obj_labels is not None else bg_labels
        n_iter = ref.shape[0] if ref is not None else int(n_iter)
        dev = ref.device if ref is not None else self.device
        if dev is None:
            raise ValueError("ShardedIteration.frame_pre without labels needs device= at construction")
        pre = torch.zeros(n_iter, 4, dtype=torch.int32, device=dev)
        if obj_labels is not None and self.obj_loop is not None:
            K = obj_labels.shape[1]
            counts = ops.label_counts(obj_labels.reshape(n_iter * K, -1))[0]            # [n_iter * K, 2]
            pre[:, 0:2] = (counts.reshape(n_iter, K, 2) == 0).sum(dim=1).to(torch.int32)   # objects with an empty mask
        if bg_labels is not None and self.bg_loop is not None:
            pre[:, 2:4] = ops.label_counts(bg_labels.reshape(n_iter, -1))[0]
        odist.allreduce_sum_(pre, self.group)                     # collective 1, once per frame
        # unpacked once for every iteration (dist.unpack_pre row-wise): step() only takes views -- no kernel and no
        # cross-stream hand-over per iteration; the background stream picks the buffers up after this point
        rows = ((pre[:, 0:2] > 0).to(torch.int32), pre[:, 2:4].reshape(n_iter, 1, 2).contiguous(),
                (pre[:, 2:4] == 0).to(torch.int32))
        side = self._bg_stream(dev)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            for t in rows:
                t.record_stream(side)
        return [tuple(t[it] for t in rows) for it in range(n_iter)]

    def step(self, obj_batch=None, bg_batch=None, pre: Optional[torch.Tensor] = None):
        """-> (object loss terms [K,4] | None, background loss terms [1,4] | None)
        pre: this iteration's element of frame_pre() (already reduced over the ranks): the pre-step exchange is skipped
        and the iteration has ONE collective, the background gradient's."""
        ref = obj_batch if obj_batch is not None else bg_batch
        if ref is not None and self.device is None:
            self.device = ref["z"].device
        dev = self.device
        sharded = odist._active(self.group)
        if ref is None and pre is not None:
            return None, None                   # (the exchange this rank would have had to join happened in frame_pre)
        if ref is None:
            # nothing to train on this rank in this iteration (no foreground object yet and no background batch).
            # Unsharded that is a no-op; sharded, the rank still joins collective 1 with zeros (flags and counts are
            # SUMs, so the other ranks' values pass through) -- otherwise they would block in it.  Collective 2 is
            # issued by the ranks that were handed background rays: every rank or none (the caller splits ONE
            # replicated batch over all ranks).
            if not sharded:
                return None, None
            if dev is None:
                raise ValueError("ShardedIteration.step(None, None) under sharding needs device= at construction")
        do_obj = obj_batch is not None and self.obj_loop is not None
        do_bg = bg_batch is not None and self.bg_loop is not None
        side = self._bg_stream(dev) if do_obj and do_bg else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        if side is not None and not self.resident:
            side.wait_stream(main)
        obj_flags = bg_counts = bg_flags = gflags = work = None
        with on_side():
            if pre is not None:                     # reduced for the whole frame by frame_pre()
                gflags, bg_counts, bg_flags = pre
            elif do_obj and sharded:
                obj_flags = ops.label_counts(obj_batch["labels"])[1]
            if do_bg and pre is None and sharded:
                bg_counts = self.bg_loop.local_counts(bg_batch)
            # (un-sharded: bg_counts = bg_flags = None -- the background step counts its labels itself)
            if sharded and pre is None:
                pre1 = odist.pack_pre(obj_flags, bg_counts, dev)
                odist.allreduce_sum_(pre1, self.group)             # collective 1
                gflags, bg_counts, bg_flags = odist.unpack_pre(pre1)
                if side is not None:
                    main.wait_stream(side)                         # (the object kernel needs the global flags)
                    gflags.record_stream(main)                     # (allocated on the second stream, read on this one)
            if do_bg:
                work = self.bg_loop.begin(bg_batch, bg_counts, bg_flags)      # collective 2 starts here
        obj_terms = bg_terms = None
        if do_obj:
            obj_terms = self.obj_loop.step(obj_batch, global_flags=gflags)     # beside the background chain
        if do_bg:
            with on_side():
                bg_terms = self.bg_loop.finish(work)
        if side is not None and not self.pipelined:
            main.wait_stream(side)
        return obj_terms, bg_terms

    def join(self):
        """The caller's stream waits for everything queued on the background stream (pipelined mode)."""
        if self._side is not None and self.device is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)


class HipTrainLoop:
    def __init__(self, cfg, trainers: List, with_feat: bool = False, bf16: bool = False):
        """trainers: the per-object Trainer instances in obj_dict order (train.py:255-256).
        bf16: opt-in bf16-operand MFMA mode of the fused kernel (ops.train_step); default = reference fp32.

        cfg.training_strategy (cfg.py:25, train.py:405-429) selects how the K objects are driven:
          "vmap" / "hip"  ONE fused launch over the stacked copies of the parameters (copied back after the frame's
                          iterations, train.py:478-485) -- the reference's vmap path;
          "forloop"       the reference's fallback: every object's OWN modules are trained in place, one
                          objnerf_train_step (K = 1) and one AdamW per object and iteration, under the batch's
                          cross-object early-return flags (the outputs are stacked before the loss, train.py:417-419);
          anything else   the reference prints and exits (train.py:427-429): ValueError here."""
        self.cfg, self.trainers, self.with_feat, self.bf16 = cfg, list(trainers), with_feat, bf16
        self.strategy = getattr(cfg, "training_strategy", "hip")
        if self.strategy not in ("hip", "vmap", "forloop"):
            raise ValueError("training strategy {} is not implemented ".format(self.strategy))
        self.arena = None
        self.opt = None
        self.ws = None
        self.rebuild()

    def rebuild(self):
        """utils.update_vmap for both model lists (train.py:272-276): gather the K per-object arena
        blocks; on the vmap path the Adam moments restart because the reference adds a fresh param group of the freshly
        stacked tensors."""
        K = len(self.trainers)
        t0 = self.trainers[0]
        if self.strategy == "forloop":
            # per-object param groups (train.py:240-251): each object's parameters enter the optimiser ONCE, when the
            # object is created, and keep their Adam moments and step count when later objects arrive -- the optimiser
            # state therefore lives on the Trainer, not on this (rebuilt) loop
            self.arena = None
            self.opts = []
            for t in self.trainers:
                if getattr(t, "hip_opt", None) is None:
                    t.hip_opt = optim.ArenaAdamW(t.arena, lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay)
                self.opts.append(t.hip_opt)
            for t in self.trainers:
                t.arena.scale.fill_(float(t.obj_scale))
            self.mask = t0.arena.has_grad_mask(self.with_feat)
            self.wss = None
            self.ws = None
            return
        self.arena = ops.ParamArena(K, t0.arena.net, t0.arena.params.device)
        with torch.no_grad():
            for k, t in enumerate(self.trainers):
                self.arena.params[k].copy_(t.arena.params[0])
                self.arena.scale[k] = float(t.obj_scale)
        self.opt = optim.ArenaAdamW(self.arena, lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay)
        self.mask = self.arena.has_grad_mask(self.with_feat)
        self.ws = None

    def _step_forloop(self, batch, global_flags):
        K, R, S = batch["z"].shape
        if self.wss is None or self.wss[0].key != ops.TrainWorkspace.make_key(1, R, S, self.with_feat, self.bf16):
            self.wss = []
            for t in self.trainers:             # one set of helper streams for the K one-object workspaces
                self.wss.append(ops.TrainWorkspace(t.arena, 1, R, S, self.with_feat, precision=self.bf16,
                                                   context=self.wss[0].context if self.wss else None))
            self.loss_terms = torch.zeros(K, 4, device=batch["z"].device)
            self.status = torch.zeros(1, dtype=torch.int32, device=batch["z"].device)
            self.ws = self                      # (status holder for train_frame)
        counts, flags = ops.label_counts(batch["labels"])           # over the WHOLE stacked batch
        if global_flags is not None:
            flags = global_flags
        self.status.zero_()
        for k, t in enumerate(self.trainers):
            bk = {key: v[k:k + 1] for key, v in batch.items()}
            ops.train_step(t.arena, self.wss[k], bk, with_feat=self.with_feat, global_flags=flags,
                           global_counts=counts[k:k + 1], bf16=self.bf16)
            self.opts[k].step(self.wss[k].grads, self.mask, flags=flags)
            self.loss_terms[k] = self.wss[k].loss_terms[0]
            torch.maximum(self.status, self.wss[k].status, out=self.status)
        return self.loss_terms

    def step(self, batch: Dict[str, torch.Tensor], global_flags: Optional[torch.Tensor] = None,
             global_counts: Optional[torch.Tensor] = None, loss_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One iteration (train.py:424-474).  Returns the per-object loss terms [K,4] (device tensor).
        global_counts (with global_flags): the batch's label counts [K, 2] when the caller already has them (one launch
        for all iterations of a frame, mapping.train_frame) -- the step then launches no label pass of its own.
        loss_out: float32 [K, 4] that receives this iteration's loss terms (default: the workspace's own tensor)."""
        if self.strategy == "forloop":
            return self._step_forloop(batch, global_flags)
        K, R, S = batch["z"].shape
        if self.ws is None or self.ws.key != ops.TrainWorkspace.make_key(K, R, S, self.with_feat, self.bf16):
            self.ws = ops.TrainWorkspace(self.arena, K, R, S, self.with_feat, precision=self.bf16)
            self._own_terms = self.ws.loss_terms
        self.ws.loss_terms = loss_out if loss_out is not None else self._own_terms
        # (optim=: the iteration's optimiser.step() runs in the step's last launch; the flags it skips tensor groups by are
        # the global pair, or the batch's own -- ws.flags -- when none was supplied)
        ops.train_step(self.arena, self.ws, batch, with_feat=self.with_feat, global_flags=global_flags,
                       global_counts=global_counts, bf16=self.bf16, optim=self.opt)
        return self.ws.loss_terms

    def train_frame(self, pool: Dict[str, torch.Tensor], n_iter: Optional[int] = None,
                    n_per_optim: Optional[int] = None, check_status: bool = True) -> List[torch.Tensor]:
        """pool: the stacked per-frame sample tensors of train.py:368-388 ([K, n_iter*n_per_optim, ...]);
        iteration i trains on rays [i*n_per_optim, (i+1)*n_per_optim) (train.py:396-404)."""
        n_iter = n_iter or self.cfg.n_iter_per_frame
        npo = n_per_optim or self.cfg.n_per_optim
        out = []
        for it in range(n_iter):
            sl = slice(it * npo, (it + 1) * npo)
            batch = {k: v[:, sl].contiguous() for k, v in pool.items()}
            out.append(self.step(batch).clone())
        if check_status:
            _check_status(self.ws.status)
        return out

    def copy_back(self):
        """train.py:478-485: stacked parameters -> each object's modules (their arena block)."""
        if self.strategy == "forloop":
            return                              # the modules themselves were trained
        with torch.no_grad():
            for k, t in enumerate(self.trainers):
                t.arena.params[0].copy_(self.arena.params[k])
