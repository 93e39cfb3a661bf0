"""CPU: host-side mirror of the reference interface (no kernels run here)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import T
from openobj_amd import cfg as ocfg
from openobj_amd import dist as odist
from openobj_amd import ops, trainer, utils


def make_cfg(**kw):
    c = ocfg.Config(ocfg.replica_room0_config(train_device="cpu", **kw))
    c.obj_id = 1
    return c


def test_config_matches_room0_json():
    """Attribute names and derived values of cfg.py:16-114 on the shipped room_0 hyper-parameters."""
    c = make_cfg()
    assert (c.n_per_optim, c.win_size, c.n_samples_per_frame) == (120, 5, 24)
    assert (c.n_per_optim_bg, c.win_size_bg, c.n_samples_per_frame_bg) == (1200, 10, 120)
    assert (c.n_bins, c.n_bins_cam2surface, c.n_bins_cam2surface_bg) == (9, 1, 5)
    assert (c.hidden_feature_size, c.hidden_feature_size_bg, c.clip_point_feature_size) == (32, 128, 512)
    assert (c.W, c.H, c.fx, c.cx, c.cy) == (1200, 680, 600.0, 599.5, 339.5)
    assert (c.learning_rate, c.weight_decay) == (0.001, 0.013)
    assert c.keyframe_step == 2.5 and c.keyframe_step_bg == 5.0 and c.part_mode and c.part_down == 5
    assert (c.surface_eps, c.stop_eps, c.obj_scale, c.bg_scale) == (0.1, 0.05, 2.0, 5.0)


def test_config_from_json_file(tmp_path):
    p = tmp_path / "c.json"
    p.write_text(json.dumps(ocfg.replica_room0_config(train_device="cpu", strategy="vmap")))
    assert ocfg.Config(str(p)).training_strategy == "vmap"


def test_trainer_init_reproduces_reference(golden):
    """Same seed -> same initial parameters as the reference's Trainer (trainer.py:36-44, model.py:4-6):
    the G5 fixture holds the reference's stack for torch.manual_seed(50)."""
    g = golden("g5_step_s10_nofeat")
    torch.manual_seed(50)
    ts = [trainer.Trainer(make_cfg()) for _ in range(3)]
    for k, t in enumerate(ts):
        for i, p in enumerate(t.fc_occ_map.parameters()):
            assert torch.equal(p.detach(), T(g[f"fc0_{i}"])[k]), (k, i)
    keys = list(ts[0].fc_occ_map.state_dict().keys())
    assert keys == [n for n in ops.TENSOR_NAMES[:18]]
    assert list(ts[0].pe.state_dict().keys()) == ["scale", "B_layer.weight"]


def test_update_vmap_stacks_like_combine_state_for_ensemble(golden):
    g = golden("g5_step_s10_nofeat")
    torch.manual_seed(50)
    ts = [trainer.Trainer(make_cfg()) for _ in range(3)]
    fmodel, params, buffers = utils.update_vmap([t.fc_occ_map for t in ts])
    pe_model, pe_params, pe_buffers = utils.update_vmap([t.pe for t in ts], arena=fmodel.arena)
    assert len(params) == 18 and len(pe_params) == 1
    for i, p in enumerate(params):
        assert tuple(p.shape) == tuple(g[f"fc0_{i}"].shape)
        assert torch.equal(p, T(g[f"fc0_{i}"]))
    assert tuple(pe_params[0].shape) == (3, 21, 3)
    assert tuple(pe_buffers[0].shape) == (3, 6) and tuple(pe_buffers[1].shape) == (3,)


def test_shard_objects_partition():
    for K in (1, 7, 50, 120, 512):
        for world in (1, 2, 4, 8):
            blocks = [odist.shard_objects(K, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == K
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 1 owns an object with an empty label-1 mask: the flag must become global
    flags = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)
    odist.global_flags(flags)
    terms = torch.full((2, 4), float(rank + 1))
    tot = odist.total_loss(terms)
    # background network: rays split over ranks, mask counts and gradients SUM-reduced (train.BackgroundLoop)
    lo, hi = odist.shard_rays(1200, world, rank)
    counts = torch.tensor([[hi - lo, 7 * (rank + 1)]], dtype=torch.int32)
    odist.allreduce_sum_(counts)
    grads = torch.full((1, 8), float(rank + 1))
    odist.allreduce_sum_(grads)
    # replicated background network of the sharded mapping loop: identical weights from rank 0
    w = torch.full((1, 5), float(10 + rank))
    odist.broadcast_(w, 0)
    # the iteration's two collectives (train.ShardedIteration): packed pre-step SUM, then gradient + loss terms in one
    pre = odist.pack_pre(torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32),
                         torch.tensor([[hi - lo, 7 * (rank + 1)]], dtype=torch.int32), "cpu")
    odist.allreduce_sum_(pre)
    gflags, bg_counts, bg_flags = odist.unpack_pre(pre)
    flat = torch.cat([torch.full((8,), float(rank + 1)), torch.full((4,), 0.5 * (rank + 1))])
    work = odist.allreduce_sum_async(flat)
    work.wait()
    # ... and the whole orchestration with stand-in loops (the kernels need a GPU; the ORDER of operations does not)
    from openobj_amd import ops as oops, train as otrain

    class ObjLoop:
        def step(self, batch, global_flags=None):
            self.seen_flags = global_flags.tolist()
            return torch.full((2, 4), float(rank))

    class BgLoop:
        def local_counts(self, batch):
            return torch.tensor([[hi - lo, 0 if rank == 0 else 5]], dtype=torch.int32)

        def begin(self, batch, counts, flags):
            self.seen = (counts.tolist(), flags.tolist())
            self.flat = torch.full((6,), float(rank + 1))
            return odist.allreduce_sum_async(self.flat)

        def finish(self, work):
            work.wait()
            return self.flat[-4:].view(1, 4)

    _lc = oops.label_counts
    oops.label_counts = lambda labels: (None, torch.tensor([0, 1 if rank == 0 else 0], dtype=torch.int32))
    try:
        ol, bl = ObjLoop(), BgLoop()
        ot, bt = otrain.ShardedIteration(ol, bl).step({"z": torch.zeros(2, 3, 4), "labels": None},
                                                      {"z": torch.zeros(1, 3, 4), "labels": None})
        # one foreground object in the whole job and do_bg = 0 (mapping.train_frame on a rank that owns nothing):
        # rank 1 is handed NO batch and must still join collective 1, or rank 0 blocks in it
        ol2 = ObjLoop()
        it2 = otrain.ShardedIteration(ol2 if rank == 0 else None, None, device="cpu")
        r2 = it2.step({"z": torch.zeros(1, 3, 4), "labels": None} if rank == 0 else None, None)
        lone = (r2[0].tolist() if r2[0] is not None else None, r2[1], getattr(ol2, "seen_flags", None))
    finally:
        oops.label_counts = _lc
    q.put((rank, flags.tolist(), float(tot), counts.tolist(), grads[0, 0].item(), odist.rank_world(), w[0, 0].item(),
           (gflags.tolist(), bg_counts.tolist(), bg_flags.tolist(), flat.tolist()),
           (ol.seen_flags, bl.seen, bt.tolist()), lone))
    dist.destroy_process_group()


def test_two_rank_flags_and_loss_gloo():
    """world_size-2 gloo run of the object-sharding host logic (the N>1 path of bench.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    expect = 2 * (1 + 5 + 10 + 5) * 1.0 + 2 * (1 + 5 + 10 + 5) * 2.0
    assert odist.rank_world() == (0, 1)              # no process group in this process
    for rank, flags, tot, counts, g0, rw, w0, packed, orch, lone in res:
        assert rw == (rank, 2) and w0 == 10.0
        assert flags == [1, 0]
        assert abs(tot - expect) < 1e-4
        assert counts == [[1200, 21]] and g0 == 3.0
        # packed pre-step exchange: flags as counts of empty-mask objects, background counts summed
        assert packed[0] == [1, 0] and packed[1] == [[1200, 21]] and packed[2] == [0, 0]
        assert packed[3] == [3.0] * 8 + [1.5] * 4                      # gradient and loss terms in ONE all-reduce
        # orchestration: both ranks see rank 0's empty mask, the background's GLOBAL counts, and the summed buffer
        assert orch[0] == [0, 1] and orch[1] == ([[1200, 5]], [0, 0]) and orch[2] == [[3.0, 3.0, 3.0, 3.0]]
        # the rank without a batch returned (None, None) after joining the exchange; the owner saw ITS flags only
        assert lone == (([[0.0] * 4] * 2, None, [0, 1]) if rank == 0 else (None, None, None))


def _frame_worker(rank, world, port, q):
    """A frame of n_iter iterations under object sharding: ONE pre-step exchange for the whole frame
    (ShardedIteration.frame_pre) + one collective per iteration (the background gradient)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openobj_amd import ops as oops, train as otrain
    n_calls = [0]
    _ar = dist.all_reduce

    def counting(*a, **k):
        n_calls[0] += 1
        return _ar(*a, **k)

    dist.all_reduce = counting

    def cpu_label_counts(labels):                    # objnerf_label_counts restated (the kernel needs a GPU)
        c = torch.stack([(labels == 1).sum(1), (labels != 2).sum(1)], dim=1).to(torch.int32)
        return c, (c == 0).any(dim=0).to(torch.int32)

    _lc = oops.label_counts
    oops.label_counts = cpu_label_counts
    n_iter, K, R, Rb = 5, 2, 6, 4
    gen = torch.Generator().manual_seed(100 + rank)
    obj_labels = torch.randint(0, 3, (n_iter, K, R), generator=gen, dtype=torch.uint8)
    obj_labels[:, :, 0] = 1
    if rank == 1:
        obj_labels[2, 1] = 0                         # iteration 2: rank 1's second object has no label-1 ray
        obj_labels[4, 0] = 2                         # iteration 4: all label 2 -> both masks empty
    bg_labels = torch.randint(0, 3, (n_iter, 1, Rb), generator=gen, dtype=torch.uint8)
    seen = []

    class ObjLoop:
        def step(self, batch, global_flags=None):
            seen.append(("obj", global_flags.tolist()))
            return torch.zeros(K, 4)

    class BgLoop:
        def begin(self, batch, counts, flags):
            seen.append(("bg", counts.tolist(), flags.tolist()))
            self.flat = torch.full((6,), float(rank + 1))
            return odist.allreduce_sum_async(self.flat)

        def finish(self, work):
            work.wait()
            return self.flat[-4:].view(1, 4)

    try:
        it = otrain.ShardedIteration(ObjLoop(), BgLoop(), device="cpu", resident=True)
        pre = it.frame_pre(obj_labels, bg_labels)
        after_pre = n_calls[0]
        sums = []
        for i in range(n_iter):
            _, bt = it.step({"z": torch.zeros(K, R, 3), "labels": obj_labels[i]},
                            {"z": torch.zeros(1, Rb, 3), "labels": bg_labels[i]}, pre=pre[i])
            sums.append(bt[0, 0].item())
        total = n_calls[0]
        # a rank that owns nothing in this frame joins the ONE exchange and nothing else
        it2 = otrain.ShardedIteration(ObjLoop() if rank == 0 else None, None, device="cpu")
        pre2 = it2.frame_pre(obj_labels if rank == 0 else None, None, n_iter)
        r2 = [it2.step({"z": torch.zeros(K, R, 3), "labels": obj_labels[i]} if rank == 0 else None, None, pre=pre2[i])
              for i in range(n_iter)]
        lone_calls = n_calls[0] - total
    finally:
        oops.label_counts = _lc
        dist.all_reduce = _ar
    bg_local = torch.stack([(bg_labels[:, 0] == 1).sum(1), (bg_labels[:, 0] != 2).sum(1)], dim=1)
    q.put((rank, after_pre, total, lone_calls, seen, sums, bg_local.tolist(), [x[0] is None for x in r2]))
    dist.destroy_process_group()


def test_one_pre_step_collective_per_frame_gloo():
    """mapping.train_frame under sharding: 1 + n_iter collectives per frame (the early-return flags and background mask
    counts of ALL iterations in one int32[n_iter, 4] all-reduce before the loop -- render_rays.py:89-94,
    train.py:447-463 -- then the background gradient's all-reduce per iteration), with the flags every rank's object
    kernel sees equal to the batch-wide ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    n_iter = 5
    bg_global = (np.array(res[0][6]) + np.array(res[1][6])).tolist()
    for rank, after_pre, total, lone_calls, seen, sums, _, lone_none in res:
        assert after_pre == 1 and total == 1 + n_iter, (after_pre, total)
        assert lone_calls == 1
        obj_seen = [s[1] for s in seen if s[0] == "obj"][:n_iter]
        assert obj_seen == [[0, 0], [0, 0], [1, 0], [0, 0], [1, 1]], obj_seen      # rank 1's empty masks, on BOTH ranks
        bg_seen = [s for s in seen if s[0] == "bg"]
        assert [s[1] for s in bg_seen] == [[c] for c in bg_global]                 # the GLOBAL background counts
        assert [s[2] for s in bg_seen] == [[int(c[0] == 0), int(c[1] == 0)] for c in bg_global]
        assert sums == [3.0] * n_iter                                               # collective 2 of every iteration
        assert lone_none == ([False] * n_iter if rank == 0 else [True] * n_iter)


def test_sharded_iteration_without_batches_unsharded_is_a_noop():
    from openobj_amd import train as otrain
    assert otrain.ShardedIteration(None, None).step(None, None) == (None, None)


def test_view_buffers_zbuffer_merge():
    """train.py:581-598: nearer object wins a pixel; a background object paints colour but leaves the depth
    buffer alone, so a later foreground object still takes the pixel."""
    from openobj_amd.render_view import ViewBuffers
    buf = ViewBuffers(3, 2)
    m_bg = np.ones((3, 2), dtype=bool)
    ok = buf.merge(m_bg, np.full(6, 5.0, np.float32), np.full((6, 3), 10, np.uint8), class_id=7, is_background=True)
    assert ok.all() and (buf.depth == 100).all() and (buf.rgb == 10).all() and (buf.maskid == 7).all()
    m1 = np.zeros((3, 2), dtype=bool)
    m1[0, 0] = m1[1, 1] = True
    buf.merge(m1, np.array([2.0, 3.0], np.float32), np.full((2, 3), 50, np.uint8), class_id=1, is_background=False)
    assert buf.depth[0, 0] == 2.0 and buf.depth[1, 1] == 3.0 and buf.maskid[0, 0] == 1 and buf.rgb[1, 1, 0] == 50
    m2 = np.zeros((3, 2), dtype=bool)
    m2[0, 0] = m2[2, 1] = True
    buf.merge(m2, np.array([2.5, 4.0], np.float32), np.full((2, 3), 90, np.uint8), class_id=2, is_background=False)
    assert buf.maskid[0, 0] == 1 and buf.rgb[0, 0, 0] == 50          # farther: loses
    assert buf.maskid[2, 1] == 2 and buf.depth[2, 1] == 4.0          # free pixel: wins
    assert buf.maskid[1, 0] == 7                                     # untouched background pixel
