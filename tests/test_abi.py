"""CPU: the C-ABI library builds, loads and exports every symbol include/objnerf_hip.h declares
(no compute calls -- there is no GPU here), and the host-side layout helpers agree with it."""
import os
import re

import pytest

from conftest import ROOT
from openobj_amd import _lib
from oracle import objnerf_oracle as O

HEADER = os.path.join(ROOT, "include", "objnerf_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(objnerf_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = declared_functions()
    assert len(names) >= 13
    l = _lib.lib()
    for n in names:
        assert hasattr(l, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert l.objnerf_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("H,P", [(32, 30659), (128, 182339), (256, 527939)])
def test_param_layout_matches_reference_counts(H, P):
    """Parameter counts of SURVEY.md section 7 (probed on the reference): 30596 / 182276 / 527876
    OccupancyMap parameters + 63 for B_layer.weight."""
    offs, stride = _lib.param_layout(H, 512, 6)
    assert offs[-1] == P
    assert stride % 64 == 0 and stride >= P
    sizes = [offs[i + 1] - offs[i] for i in range(19)]
    specs = O.param_specs(H, 512)
    for (name, shape), n in zip(specs, sizes[:18]):
        m = 1
        for s in shape:
            m *= s
        assert m == n, name
    assert sizes[18] == 63


def test_train_workspace_sizes():
    """objnerf_train_workspace_bytes is pure host arithmetic: its with_feat bit field (bit 0 feature loss, bit 1
    layer-wise sizing, bit 2 sizing for the 16-bit modes) -- bit 2 only changes the hidden-256 / >= 4096-sample shape,
    where activations and back-propagated gradients live in the operand type."""
    import ctypes as C
    from openobj_amd import ops
    l = _lib.lib()
    def size(H, K, R, S, wf):
        net = ops.NetShape(H, 512, 6).c()
        return int(l.objnerf_train_workspace_bytes(C.byref(net), K, R, S, wf))
    full, half = size(256, 2, 64, 64, 0), size(256, 2, 64, 64, 4)
    assert 0 < half < 0.75 * full
    assert size(256, 2, 64, 64, 1) > full and size(256, 2, 64, 64, 5) < size(256, 2, 64, 64, 1)
    assert size(256, 2, 32, 64, 4) == size(256, 2, 32, 64, 0)          # 2048 samples per object: fp32 storage
    assert size(128, 1, 1200, 64, 4) == size(128, 1, 1200, 64, 0)      # hidden 128: fp32 storage
    assert size(32, 50, 4096, 64, 4) == size(32, 50, 4096, 64, 0)      # fused kernel: slabs only
    assert size(32, 50, 4096, 64, 2) > size(32, 50, 4096, 64, 0)       # bit 1: the layer-wise path's activations


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.ObjnerfError):
        _lib.lib()


def test_cpu_tensors_are_rejected():
    import torch
    from openobj_amd import ops
    with pytest.raises(_lib.ObjnerfError):
        ops._req(torch.zeros(3), torch.float32, "x")
