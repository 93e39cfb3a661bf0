"""The training loop of the reference's train.py:272-276,394-485 for training_strategy == "hip":
stack the object networks into the arena, run n_iter_per_frame fused iterations over slices of the
per-frame sample pool, copy the stacked parameters back.  Object creation, dataset reading, labelling
and visualisation (the rest of train.py) stay with the caller."""
from typing import Dict, List, Optional

import torch

from . import dist as odist
from . import ops, optim
from .render_rays import LossExplode


class BackgroundLoop:
    """The separate background network (obj_id 0, hidden_feature_size_bg = 128, train.py:447-463).

    It is NOT in the vmap stack.  Under object sharding it is replicated on every rank: each rank trains
    on its slice of the iteration's n_per_optim_bg rays, the per-ray loss is normalised by the GLOBAL
    mask counts (an 2-int SUM all-reduce), the gradient (182 339 floats at hidden 128) is SUM
    all-reduced over RCCL and every rank applies the same AdamW update."""

    def __init__(self, cfg, bg_trainer, with_feat: bool = False, group=None, bf16: bool = False):
        """bf16: opt-in OBJNERF_TRAIN_BF16 mode (bf16 GEMM operands, fp32 accumulation and everything else)."""
        self.cfg, self.trainer, self.with_feat, self.group = cfg, bg_trainer, with_feat, group
        self.bf16 = bf16
        self.arena = bg_trainer.arena           # K = 1: trained in place, no copy-back needed
        self.arena.scale.fill_(float(bg_trainer.obj_scale))
        self.opt = optim.ArenaAdamW(self.arena, lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
        self.mask = self.arena.has_grad_mask(with_feat)
        self.ws = None

    def step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """batch: THIS rank's slice of the background rays, tensors shaped [1, R_local, ...]."""
        K, R, S = batch["z"].shape
        if self.ws is None or self.ws.key != (K, R, S, self.with_feat):
            self.ws = ops.TrainWorkspace(self.arena, K, R, S, self.with_feat)
        counts, flags = ops.label_counts(batch["labels"])
        if odist._active(self.group):
            odist.allreduce_sum_(counts, self.group)              # global n(label==1), n(label!=2)
            flags = ((counts.reshape(-1, 2) == 0).any(dim=0)).to(torch.int32)
        ops.train_step(self.arena, self.ws, batch, with_feat=self.with_feat, global_flags=flags,
                       global_counts=counts, bf16=self.bf16)
        odist.allreduce_sum_(self.ws.grads, self.group)           # the one data-path collective
        odist.allreduce_sum_(self.ws.loss_terms, self.group)
        self.opt.step(self.ws.grads, self.mask)
        return self.ws.loss_terms


class HipTrainLoop:
    def __init__(self, cfg, trainers: List, with_feat: bool = False, bf16: bool = False):
        """trainers: the per-object Trainer instances in obj_dict order (train.py:255-256).
        bf16: opt-in bf16-operand MFMA mode of the fused kernel (ops.train_step); default = reference fp32."""
        self.cfg, self.trainers, self.with_feat, self.bf16 = cfg, list(trainers), with_feat, bf16
        self.arena = None
        self.opt = None
        self.ws = None
        self.rebuild()

    def rebuild(self):
        """utils.update_vmap for both model lists (train.py:272-276): gather the K per-object arena
        blocks; Adam moments restart because the reference adds a fresh param group."""
        K = len(self.trainers)
        t0 = self.trainers[0]
        self.arena = ops.ParamArena(K, t0.arena.net, t0.arena.params.device)
        with torch.no_grad():
            for k, t in enumerate(self.trainers):
                self.arena.params[k].copy_(t.arena.params[0])
                self.arena.scale[k] = float(t.obj_scale)
        self.opt = optim.ArenaAdamW(self.arena, lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay)
        self.mask = self.arena.has_grad_mask(self.with_feat)
        self.ws = None

    def step(self, batch: Dict[str, torch.Tensor], global_flags: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One iteration (train.py:424-474).  Returns the per-object loss terms [K,4] (device tensor)."""
        K, R, S = batch["z"].shape
        if self.ws is None or self.ws.key != (K, R, S, self.with_feat):
            self.ws = ops.TrainWorkspace(self.arena, K, R, S, self.with_feat)
        ops.train_step(self.arena, self.ws, batch, with_feat=self.with_feat, global_flags=global_flags,
                       bf16=self.bf16)
        self.opt.step(self.ws.grads, self.mask)
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
        if check_status and int(self.ws.status.item()) != 0:
            raise LossExplode("loss explode")
        return out

    def copy_back(self):
        """train.py:478-485: stacked parameters -> each object's modules (their arena block)."""
        with torch.no_grad():
            for k, t in enumerate(self.trainers):
                t.arena.params[0].copy_(self.arena.params[k])
