"""Dataset adapters (openobj_amd/dataset.py) against a scene written in the reference's on-disk layouts."""
import numpy as np
import pytest
import torch

from openobj_amd import cfg as ocfg
from openobj_amd import dataset as ods
try:
    from tests import scene_files as SF
except ImportError:          # plain `pytest tests/` puts tests/ itself, not the repository root, on sys.path
    import scene_files as SF


def make_cfg(root, fmt, **kw):
    over = {"dataset.path": str(root), "dataset.format": fmt, "trainer.part_mode": 0, "camera.w": SF.W, "camera.h": SF.H,
            "camera.fx": SF.FX, "camera.fy": SF.FY, "camera.cx": SF.CX, "camera.cy": SF.CY}
    over.update(kw)
    return ocfg.Config(ocfg.replica_room0_config(train_device="cpu", **over))


def expected_obj_and_boxes(inst_hw):
    """Independent per-pixel restatement of dataset.py:112-173 for the helper scene."""
    inst = inst_hw.astype(np.int32).T.copy()            # [W, H]
    inst[inst == 0] = -1
    obj = np.full_like(inst, -1)
    boxes = {}
    for k in (4, 7, 5):
        ws, hs = np.nonzero(inst == k)
        w0, w1, h0, h1 = ws.min(), ws.max() + 1, hs.min(), hs.max() + 1
        if w1 - w0 <= 10 or h1 - h0 <= 10:
            continue
        mw, mh = int(0.1 * (w1 - w0)), int(0.1 * (h1 - h0))
        boxes[k] = [max(w0 - mw, 0), min(w1 + mw, SF.W - 1), max(h0 - mh, 0), min(h1 + mh, SF.H - 1)]
        obj[inst == k] = k
    obj[inst == 1] = 0
    return obj, boxes


@pytest.mark.parametrize("fmt", ["Replica", "ScanNet"])
def test_frame_samples(tmp_path, fmt):
    SF.write_scene(str(tmp_path), fmt, n_frames=30)
    c = make_cfg(tmp_path, fmt)
    ds = ods.Replica(c) if fmt == "Replica" else ods.ScanNet(c)
    assert len(ds) == 3
    for i in range(3):
        s = ds[i]
        rgb, depth_mm, inst = SF._frame(i, None)
        assert s["frame_id"] == 10 * i
        assert s["image"].shape == (SF.W, SF.H, 3) and s["image"].dtype == np.uint8
        if fmt == "Replica":
            assert np.array_equal(s["image"], rgb.transpose(1, 0, 2))
        else:           # JPEG at twice the resolution, resized back: close, not equal
            assert np.abs(s["image"].astype(int) - rgb.transpose(1, 0, 2).astype(int)).mean() < 3.0
        assert s["depth"].dtype == np.float32 and np.allclose(s["depth"], depth_mm.T / 1000.0)
        obj, boxes = expected_obj_and_boxes(inst)
        assert s["obj"].dtype == np.int32 and np.array_equal(s["obj"], obj)
        assert sorted(s["bbox_dict"]) == [0, 4, 7]                 # the speck (5) is dropped, 0 = background
        assert s["bbox_dict"][0].tolist() == [0, SF.W, 0, SF.H]
        for k, b in boxes.items():
            assert s["bbox_dict"][k].dtype == torch.int64 and s["bbox_dict"][k].tolist() == b
        assert sorted(s["obj_clip"]) == [0, 4, 7] and s["obj_clip"][4].shape == (1, 16) and s["obj_cap"][7].shape == (12,)
        assert np.array_equal(s["obj_clip"][0], ds.obj_clipfeat[i][1])          # background takes label 1's feature
        assert np.allclose(s["T"], np.eye(4) + np.outer([1, 0, 0, 0], [0, 0, 0, 0.02 * i]))
        assert np.array_equal(s["T_obj"], np.eye(4))


def test_depth_filter_and_imap_mode(tmp_path):
    SF.write_scene(str(tmp_path), "Replica", n_frames=10)
    c = make_cfg(tmp_path, "Replica", **{"render.depth_range": [0.0, 2.5], "trainer.imap_mode": 1})
    s = ods.Replica(c)[0]
    assert float(s["depth"].max()) == 2.0                       # the 3 m wall is beyond max_depth -> 0
    assert (s["depth"] == 0).sum() > 0
    assert not s["obj"].any() and sorted(s["bbox_dict"]) == [0]  # one map for the whole frame


def test_part_features_and_loader(tmp_path):
    SF.write_scene(str(tmp_path), "ScanNet", n_frames=20, part_dim=6, part_down=2)
    c = make_cfg(tmp_path, "ScanNet", **{"trainer.part_mode": 1, "trainer.part_down": 10})
    dl = ods.init_loader(c, multi_worker=False)
    samples = list(dl)
    assert len(samples) == 2
    s = samples[1]
    assert torch.is_tensor(s["image"]) and s["image"].dtype == torch.uint8 and s["depth"].dtype == torch.float32
    assert s["part_feat"].shape == (SF.W // 4, SF.H // 4, 6)      # [W', H', C], halved once more (part_down 10)
    raw = np.load(str(tmp_path / "partlevel" / "10.npy")).transpose(1, 0, 2)
    assert np.allclose(s["part_feat"][0, 0].numpy(), raw[:2, :2].mean((0, 1)), atol=1e-6)
    assert int(s["frame_id"]) == 10 and sorted(s["bbox_dict"]) == [0, 4, 7]


def test_resize_linear_matches_half_pixel_rule():
    img = np.arange(4 * 6 * 1, dtype=np.uint8).reshape(4, 6, 1) * 10
    out = ods.resize_linear(img, 3, 2)                            # exact 2x decimation: mean of the 2x2 blocks
    want = img.reshape(2, 2, 3, 2, 1).astype(float).mean((1, 3))
    assert np.array_equal(out, np.rint(want).astype(np.uint8))
    assert ods.resize_linear(img, 6, 4) is img
    up = ods.resize_linear(img, 12, 8)
    assert up.shape == (8, 12, 1) and up[0, 0, 0] == img[0, 0, 0] and up[-1, -1, 0] == img[-1, -1, 0]


def test_enlarge_bbox_rules():
    assert ods.enlarge_bbox([10, 20, 30, 60], 0.2, 100, 100) == [8, 16, 32, 64]
    assert ods.enlarge_bbox([0, 0, 99, 99], 0.2, 100, 100) == [0, 0, 99, 99]
    assert ods.enlarge_bbox([10, 10, 15, 60], 0.2, 100, 100) is None          # margin rounds to 0 (utils.py:73-74)


def test_majority_cluster_mean_and_imports():
    """utils.py:138-155 (DBSCAN majority cluster) and that the mapping module imports without a GPU."""
    from openobj_amd import mapping
    rs = np.random.RandomState(0)
    a = np.tile(np.eye(8)[1], (6, 1)) + 0.01 * rs.randn(6, 8)        # the majority cluster
    b = np.tile(np.eye(8)[5], (3, 1)) + 0.01 * rs.randn(3, 8)
    v = np.vstack([a[:3], b, a[3:]])
    m = mapping.get_majority_cluster_mean(v, eps=0.2, min_samples=2)
    assert np.allclose(m, a.mean(0), atol=1e-12)
    assert mapping._first(np.ones((1, 4))).shape == (4,) and mapping._first(np.ones(4)).shape == (4,)


G14_CASES = [("Replica", dict(n_frames=30, part_dim=4, part_down=4), 4),
             ("ScanNet", dict(n_frames=30, part_dim=4, part_down=4, color_scale=1), 4),
             ("ScanNet", dict(n_frames=20, part_dim=6, part_down=2, color_scale=1), 10)]


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_samples_equal_the_reference_classes_g14(tmp_path, golden, ci):
    """Fixture G14 = the reference's own Replica / ScanNet classes run on this helper scene (tests/golden/
    make_g14_dataset.py; image decoding through stand-ins, see its header): every field of every sample, exactly."""
    g = golden("g14_dataset")
    fmt, kw, cfg_pd = G14_CASES[ci]
    SF.write_scene(str(tmp_path), fmt, seed=3 + ci, **kw)
    c = make_cfg(tmp_path, fmt, **{"trainer.part_mode": 1, "trainer.part_down": cfg_pd})
    assert (c.stride, c.start, c.depth_scale, c.max_depth) == (10, 0, 1 / 1000.0, 8.0)      # the generator's cfg
    ds = ods.Replica(c) if fmt == "Replica" else ods.ScanNet(c)
    assert len(ds) == int(g[f"c{ci}_len"])
    for i in range(len(ds)):
        s, pre = ds[i], f"c{ci}_s{i}_"
        for k in ("image", "depth", "T", "T_obj", "obj"):
            ref = g[pre + k]
            assert s[k].dtype == ref.dtype and s[k].shape == ref.shape and np.array_equal(s[k], ref), (i, k)
        assert s["frame_id"] == int(g[pre + "frame_id"])
        keys = sorted(s["bbox_dict"])
        assert keys == g[pre + "keys"].tolist() == sorted(s["obj_clip"]) == sorted(s["obj_cap"])
        assert np.array_equal(np.stack([s["bbox_dict"][k].numpy() for k in keys]), g[pre + "boxes"])
        assert np.array_equal(np.stack([np.asarray(s["obj_clip"][k]).reshape(-1) for k in keys]), g[pre + "clip"])
        assert np.array_equal(np.stack([np.asarray(s["obj_cap"][k]).reshape(-1) for k in keys]), g[pre + "cap"])
        pf = s["part_feat"].numpy()
        assert pf.shape == g[pre + "part_feat"].shape and np.array_equal(pf, g[pre + "part_feat"])
