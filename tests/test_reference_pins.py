"""Pins of the reference-facing state machines either side of the hot path against
fixtures produced by RUNNING the reference (tests/golden/make_golden.py g12 / g13):

* G12: sceneObject.__init__ / append_keyframe / prune_keyframe (vmap.py:29-257) and the per-object state map of
  train.py:197-205, replayed frame by frame: kf_id_dict order, lastest_kf_queue, kf_pointer, n_keyframes, use_frame
  after every frame, the candidates offered to random.choice, and the final keyframe buffers.
* G13: the checkpoint file the reference writes (vmap.py:556-576) loads here; and -- in the build container only,
  where /root/reference exists -- a file written here loads through the reference's own load_checkpoints
  (vmap.py:579-602, load_state_dict(strict)).
"""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T
from openobj_amd import cfg as ocfg
from openobj_amd import vmap as ovmap

SCENARIOS = ["fg_step2p5_buf6", "fg_step1_buf5", "bg_step5_buf20"]


def _map_cfg(g, tag, device="cpu"):
    obj_id, Fb, W, H, n_frames = [int(x) for x in g[f"{tag}_cfg"]]
    c = ocfg.Config(ocfg.replica_room0_config(train_device=device))
    c.keyframe_buffer_size = Fb
    c.W, c.H = W, H
    c.part_mode, c.part_down, c.stride = True, 2, 10
    c.do_bg = 1
    step = float(g[f"{tag}_steps"][0])
    if obj_id == 0:
        c.keyframe_step_bg = step
    else:
        c.keyframe_step = step
    return c, obj_id, n_frames


def _state_map(inst, obj_id):
    """train.py:201-203 (host restatement for the CPU replay; the GPU replay lets objnerf_ingest_frame build it)."""
    st = torch.zeros(inst.shape, dtype=torch.uint8)
    st[inst == obj_id] = 1
    st[inst == -1] = 2
    return st


class _ReplayChoice:
    """random.choice of prune_keyframe (vmap.py:253): must be offered as many candidates as the reference was, and
    returns the pick the reference made."""

    def __init__(self, g, tag):
        self.picks, self.n_cands, self.i = g[f"{tag}_picks"], g[f"{tag}_n_cands"], 0

    def __call__(self, seq):
        assert len(seq) == int(self.n_cands[self.i]), (self.i, len(seq), int(self.n_cands[self.i]))
        pick = tuple(int(x) for x in self.picks[self.i])
        assert pick in [tuple(int(y) for y in x) for x in seq], (self.i, pick, seq)
        self.i += 1
        return pick


def _check_trace(so, g, tag, i):
    items = np.array(list(so.kf_id_dict.items()), np.int64).reshape(-1, 2)
    want = g[f"{tag}_items"][i]
    want = want[want[:, 0] >= 0]
    assert np.array_equal(items, want), (tag, i, items.tolist(), want.tolist())
    q = g[f"{tag}_queue"][i]
    assert list(so.lastest_kf_queue) == [int(x) for x in q if x >= 0], (tag, i)
    n_kf, ptr, cnt, full = [int(x) for x in g[f"{tag}_meta"][i]]
    assert (so.n_keyframes, -1 if so.kf_pointer is None else so.kf_pointer, so.frame_cnt, int(so.kf_buffer_full)) == \
        (n_kf, ptr, cnt, full), (tag, i)
    assert np.array_equal(np.asarray(so.use_frame), g[f"{tag}_use_frame"][i]), (tag, i)


def _check_buffers(so, g, tag):
    live = [int(x) for x in g[f"{tag}_live_slots"]]
    assert torch.equal(so.rgbs_batch[live].cpu(), T(g[f"{tag}_rgbs_batch"]))          # rgb AND the state channel
    assert torch.equal(so.depth_batch[live].cpu(), T(g[f"{tag}_depth_batch"]))
    assert torch.equal(so.t_wc_batch[live].cpu(), T(g[f"{tag}_t_wc_batch"]))
    assert torch.equal(so.bbox[live].cpu(), T(g[f"{tag}_bbox_batch"]))


@pytest.mark.parametrize("tag", SCENARIOS)
def test_keyframe_trace_g12_host(golden, monkeypatch, tag):
    """The host bookkeeping (openobj_amd.vmap.sceneObject, direct slot writes) replays the reference's trace."""
    g = golden("g12_keyframes")
    cfg, obj_id, n_frames = _map_cfg(g, tag)
    replay = _ReplayChoice(g, tag)
    monkeypatch.setattr(ovmap.random, "choice", replay)
    so = None
    for i in range(n_frames):
        args = (T(g[f"{tag}_rgb"][i]), T(g[f"{tag}_depth"][i]), _state_map(T(g[f"{tag}_inst"][i]), obj_id),
                T(g[f"{tag}_bbox"][i]), T(g[f"{tag}_t_wc"][i]), int(g[f"{tag}_frame_ids"][i]))
        if so is None:
            so = ovmap.sceneObject(cfg, obj_id, *args)
        else:
            so.append_keyframe(*args)
        _check_trace(so, g, tag, i)
    assert replay.i == len(g[f"{tag}_picks"])
    _check_buffers(so, g, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", SCENARIOS)
def test_keyframe_trace_g12_ingest_kernel(golden, monkeypatch, dev, tag):
    """The production path: slots deferred and written by objnerf_ingest_frame, which also builds the state map from
    the instance image (train.py:201-203) -- same trace, same buffers, bit for bit."""
    from openobj_amd import ops
    g = golden("g12_keyframes")
    cfg, obj_id, n_frames = _map_cfg(g, tag, device=str(dev))
    replay = _ReplayChoice(g, tag)
    monkeypatch.setattr(ovmap.random, "choice", replay)
    so = None
    for i in range(n_frames):
        rgb, depth = T(g[f"{tag}_rgb"][i]).to(dev), T(g[f"{tag}_depth"][i]).to(dev)
        inst, twc = T(g[f"{tag}_inst"][i]).to(dev), T(g[f"{tag}_t_wc"][i]).to(dev)
        bbox, fid = T(g[f"{tag}_bbox"][i]), int(g[f"{tag}_frame_ids"][i])
        writes = []
        if so is None:
            so = ovmap.sceneObject(cfg, obj_id, rgb, depth, None, bbox, twc, fid, defer=writes)
        else:
            so._defer = writes
            so.append_keyframe(rgb, depth, None, bbox, twc, fid)
        ops.ingest_frame(rgb, depth, inst, twc, [(s.keyframe_store(), slot, s.obj_id, box.tolist())
                                                 for s, slot, box in writes])
        so._defer = None
        _check_trace(so, g, tag, i)
    torch.cuda.synchronize()
    _check_buffers(so, g, tag)


# ------------------------------------------------------------------------------------------------ G13
def _obj_cfg(device="cpu"):
    c = ocfg.Config(ocfg.replica_room0_config(train_device=device))
    c.W, c.H = 8, 6
    c.part_mode = False
    return c


def _blank_object(device="cpu", obj_id=5):
    c = _obj_cfg(device)
    W, H = c.W, c.H
    return ovmap.sceneObject(c, obj_id, torch.zeros(W, H, 3, dtype=torch.uint8), torch.zeros(W, H),
                             torch.zeros(W, H, dtype=torch.uint8), torch.zeros(4), torch.eye(4), 0)


def test_reference_written_checkpoint_loads_g13(golden):
    """tests/golden/g13_ref_obj_5.pth was written by the reference's save_checkpoints; loading it here reproduces the
    reference's parameters bit for bit and its metadata."""
    g = golden("g13_ckpt")
    so = _blank_object()
    assert so.load_checkpoints(os.path.join(GOLDEN, "g13_ref_obj_5.pth")) is True
    for i, p in enumerate(so.trainer.fc_occ_map.parameters()):
        assert torch.equal(p.detach().cpu(), T(g[f"p{i}"])), i
    assert torch.equal(so.trainer.pe.B_layer.weight.detach().cpu(), T(g["B"]))
    assert (so.obj_id, so.semantic_id, float(so.trainer.obj_scale)) == (5, 4, 2.0)
    assert np.array_equal(so.clip_feat, g["clip_feat"]) and np.array_equal(so.caption_feat, g["caption_feat"])


@pytest.mark.gpu
def test_reference_written_checkpoint_renders_like_reference_g13(golden, dev):
    """... and the loaded networks evaluate to the reference's outputs on the fixture's points."""
    g = golden("g13_ckpt")
    so = _blank_object(str(dev))
    assert so.load_checkpoints(os.path.join(GOLDEN, "g13_ref_obj_5.pth")) is True
    pts = T(g["pts"]).to(dev)
    emb = so.trainer.pe(pts)
    alpha, color, clip = so.trainer.fc_occ_map(emb)
    assert float((emb.detach().cpu() - T(g["emb"])).abs().max()) < 2e-5
    assert float((alpha.detach().cpu() - T(g["alpha"])).abs().max()) < 1e-4
    assert float((color.cpu() - T(g["color"])).abs().max()) < 1e-5
    assert float((clip.cpu() - T(g["clip"])).abs().max()) < 1e-4


REF = "/root/reference/objnerf"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_checkpoint_written_here_loads_in_reference(tmp_path, monkeypatch):
    """The other direction, executed with the reference's own code: sceneObject.save_checkpoints here ->
    the reference's sceneObject.load_checkpoints (strict load_state_dict on its own OccupancyMap / UniDirsEmbed)."""
    sys.path.insert(0, os.path.join(GOLDEN))
    import make_golden as MG                       # imports the reference modules with the GUI packages stubbed
    torch.manual_seed(77)
    so = _blank_object(obj_id=9)
    with torch.no_grad():
        so.trainer.pe.B_layer.weight.add_(0.03 * torch.randn(21, 3))
    so.clip_feat = np.random.RandomState(1).randn(2, 512).astype(np.float32)
    so.caption_feat = np.random.RandomState(2).randn(2, 384).astype(np.float32)
    so.set_semantic(11)
    so.save_checkpoints(str(tmp_path), epoch=3)
    path = os.path.join(str(tmp_path), "obj_9.pth")
    assert os.path.exists(path)
    # the reference runs torch 2.0.1, where torch.load unpickles numpy arrays by default
    _load = torch.load
    monkeypatch.setattr(MG.ref_vmap.torch, "load", lambda f, *a, **k: _load(f, *a, weights_only=False, **k))
    cfg = MG.make_cfg(hidden=32, scale=2.0)
    import types
    ref_self = types.SimpleNamespace(trainer=MG.ref_trainer.Trainer(cfg), training_device="cpu")
    assert MG.ref_vmap.sceneObject.load_checkpoints(ref_self, path) is True
    for p_ref, p in zip(ref_self.trainer.fc_occ_map.parameters(), so.trainer.fc_occ_map.parameters()):
        assert torch.equal(p_ref.detach(), p.detach().cpu())
    assert torch.equal(ref_self.trainer.pe.B_layer.weight.detach(), so.trainer.pe.B_layer.weight.detach().cpu())
    assert list(ref_self.trainer.fc_occ_map.state_dict().keys()) == list(so.trainer.fc_occ_map.state_dict().keys())
    assert (ref_self.obj_id, ref_self.semantic_id, float(ref_self.trainer.obj_scale)) == (9, 11, 2.0)
    assert np.array_equal(ref_self.clip_feat, so.clip_feat)
