"""GPU: the PSNR half of BASELINE.json's metric ("... PSNR within 0.1 dB of reference") on the scene SURVEY.md 8(d)
specifies (G9b: 8 analytic ellipsoids, 4096 held-out rays per object), against what the reference's own modules
produced for the same weight seeds and batches in the build container (tests/golden/g9b_ensemble_*.npz,
make_g9b_ensemble.py).

Two statements, because training is chaotic:
  * after 50 iterations the trajectories of two correct implementations have not diverged: PER SEED |delta| < 0.1 dB in
    fp32 (the reference's arithmetic), the MEAN within 0.1 dB in the 16-bit modes;
  * after 300 iterations only ensemble means compare: 320 seeds give a 95 % interval of +-0.06 dB on the difference of
    means (sigma 0.39 dB), and the difference itself must stay inside 0.1 dB.
All seeds train side by side in one arena (psnr_scene.EnsembleRun): one fused launch per iteration."""
import numpy as np
import pytest

from openobj_amd import psnr_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def runs(dev):
    cache = {}

    def get(feat, mode):
        if (feat, mode) not in cache:
            ref = psnr_scene.reference_ensemble_b(feat)
            assert ref is not None, "tests/golden/g9b_ensemble_*.npz missing"
            seeds = [int(x) for x in ref["seeds"]]
            run = psnr_scene.EnsembleRun(dev, with_feat=feat).run(seeds, psnr_scene.MODES[mode])
            cache[(feat, mode)] = (psnr_scene.compare(run, ref, len(seeds)), run, ref)
        return cache[(feat, mode)]

    return get


def test_fp32_per_seed_after_50_iterations(runs):
    c, run, ref = runs(False, "f32")
    print("fp32, 50 iterations, per seed:", c["iter50"])
    assert c["iter50"]["n"] == 320
    assert c["iter50"]["max_abs_delta_db"] < 0.1, c["iter50"]              # EVERY seed within 0.1 dB
    assert abs(c["iter50"]["mean_delta_db"]) < 0.01, c["iter50"]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_ensemble_mean_after_300_iterations(runs, mode):
    c, run, ref = runs(False, mode)
    print(mode, "300 iterations, ensemble:", c["iter300"], " 50 iterations:", c["iter50"])
    assert c["iter300"]["ci95_db"] < 0.1, c["iter300"]                        # the measurement resolves 0.1 dB
    assert abs(c["iter300"]["delta_db"]) < 0.1, c["iter300"]
    assert abs(c["iter300"]["hip_std_db"] - c["iter300"]["ref_std_db"]) < 0.1, c["iter300"]
    assert abs(c["iter50"]["mean_delta_db"]) < 0.1, c["iter50"]               # 16-bit modes: the mean, not every seed
    assert run["psnr300"].min() > ref["psnr300"].min() - 1.0


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_feature_loss_variant(runs, mode):
    """The same scene trained WITH the 512-d feature loss (cfg.part_mode; BASELINE configs[2]): 128 seeds."""
    c, run, ref = runs(True, mode)
    print(mode, "with the feature loss:", c)
    if mode == "f32":
        # per seed: one seed in 128 can take another ReLU branch inside the first 50 iterations (measured: 127 seeds
        # within 0.05 dB, one -- seed 9092 -- at 0.136); the spread of the differences is what is bounded.  That seed was
        # traced (tools/psnr_seed_trace.py: HIP and the oracle agree to 4e-5 in every parameter for four iterations, then
        # ONE AdamW step moves cat_layer.weight by 0.6 lr) and the REFERENCE ITSELF shows the same discrete jumps on it:
        # under 1e-6 / 1e-5 relative perturbations of its initial weights its PSNR after 50 iterations is unchanged to
        # 1e-4 dB in 9 of 16 draws and moves by +0.031, +0.049 (3x), -0.011 (2x), -0.106 dB in the others
        # (profiles/r04_feat_seed9092_sensitivity.txt, tools/h256_sensitivity.py VARIANT=feat); its neighbour 9091
        # never moves by more than 5e-4 dB.
        d = np.abs(run["psnr50"] - ref["psnr50"][:len(run["psnr50"])])
        assert (d >= 0.1).sum() <= 1 and d.max() < 0.2 and c["iter50"]["std_delta_db"] < 0.03, c["iter50"]
    assert abs(c["iter50"]["mean_delta_db"]) < 0.1, c["iter50"]
    assert abs(c["iter300"]["delta_db"]) < max(0.1, c["iter300"]["ci95_db"]), c["iter300"]
    assert abs(c["featcos300"]["hip_mean"] - c["featcos300"]["ref_mean"]) < 2e-3, c["featcos300"]
    assert c["featcos300"]["hip_mean"] > 0.99


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_hidden_256_network_psnr(dev, mode):
    """16-bit configurations are judged by PSNR.  BASELINE configs[4]'s network (hidden 256) on
    the G9 scene at 32 samples per ray (fixture G9C: the reference's own modules, 33 weight seeds): bf16 / fp16 run the
    two FUSED hidden-256 kernels (objnerf_train256.hip: fwd256_kernel + wgrad256_kernel), fp32 the layer-wise chain.
    At this width training is chaotic from the start: the REFERENCE ITSELF moves by 0.3 .. 2.7 dB after 50 iterations
    when its initial weights are perturbed by 1e-7 (profiles/r04_h256_sensitivity.txt, tools/h256_sensitivity.py), so no
    per-seed statement is well-posed (at hidden 32 it is: test_fp32_per_seed_after_50_iterations).  Both comparisons
    are therefore between ensembles: the difference of the 33-seed means after 50 and after 300 iterations within
    max(0.1 dB, its 95 % interval), the spread of the ensemble like the reference's, and no seed collapsing."""
    ref = psnr_scene.reference_ensemble_c()
    assert ref is not None, "tests/golden/g9c_ensemble_h256.npz missing"
    seeds = [int(x) for x in ref["seeds"]]
    er = psnr_scene.EnsembleRun(dev, with_feat=False, spec=psnr_scene.G9C)
    assert er.cfg.hidden_feature_size == 256
    run = er.run(seeds, psnr_scene.MODES[mode])
    c = psnr_scene.compare(run, ref, len(seeds))
    print(mode, "hidden 256:", c)
    assert c["iter50"]["n"] >= 32
    early = psnr_scene.delta_report(run["psnr50"], ref["psnr50"][:len(seeds)])
    assert abs(early["delta_db"]) < max(0.1, early["ci95_db"]), early
    assert abs(c["iter300"]["delta_db"]) < max(0.1, c["iter300"]["ci95_db"]), c["iter300"]
    assert abs(c["iter300"]["hip_std_db"] - c["iter300"]["ref_std_db"]) < 0.6, c["iter300"]
    assert run["psnr300"].min() > ref["psnr300"].min() - 1.5 and run["psnr50"].min() > ref["psnr50"].min() - 1.5


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_hidden_256_network_psnr_paired_early(dev, mode):
    """The gate of test_hidden_256_network_psnr resolves +-0.5 .. 0.7 dB (33-seed ensembles, sigma 1.1 - 1.4 dB): a 0.4 dB
    regression of the fused hidden-256 kernels would pass it.  This one is PAIRED.  At hidden 256 training is chaotic
    by iteration 50, but not yet after 10 or 20: a 1e-7 relative perturbation of its initial weights moves the
    REFERENCE's own PSNR by <= 0.003 dB after 10 iterations and <= 0.05 dB after 20 (tools/h256_early.py,
    profiles/r05_h256_early_sensitivity.txt) while the PSNR has risen from ~10 to ~20 / ~27 dB -- so the same seed on the same
    batches is a well-posed comparison there.  Fixture G9D (tests/golden/make_g9d_early.py): the reference's own modules
    on the 33 seeds of G9C, PSNR after 10 and 20 iterations.
      fp32 (layer-wise chain, the reference's arithmetic): PER SEED within 0.1 dB after 10 iterations (measured max 0.065,
      mean -0.004 +- 0.005) and 0.4 dB after 20 (max 0.21, mean 0.000 +- 0.027);
      fp16 / bf16 (the two fused hidden-256 kernels): the MEAN of the 33 per-seed differences after 10 iterations within
      0.1 / 0.15 dB (measured -0.014 +- 0.051 / -0.046 +- 0.108, 95 % intervals); after 20 iterations within 0.15 / 0.3 dB
      (+0.016 +- 0.108 / +0.053 +- 0.22); no seed off by more than 1 / 1.5 dB after 10 iterations (measured 0.58 / 1.02) and
      1.5 / 3.15 dB after 20 (0.85 / 2.42; the bf16 bound is 1.3 x its measurement).
    What it detects is DEMONSTRATED, not asserted: tools/h256_handicap.py degrades the gradients of the same runs (seeded
    noise relative to each tensor's rms, between the step and AdamW) and profiles/r06_h256_handicap.txt records the
    strength at which the trained PSNR drops by ~0.3 dB and that this gate fails there.
    (profiles/r06_h256_paired_psnr.txt: the green run these bounds are read from.)"""
    ref = psnr_scene.reference_early_d()
    assert ref is not None, "tests/golden/g9d_early_h256.npz missing"
    seeds = [int(x) for x in ref["seeds"]]
    er = psnr_scene.EnsembleRun(dev, with_feat=False, spec=dict(psnr_scene.G9C, steps=20, early=10))
    assert er.cfg.hidden_feature_size == 256
    run = er.run(seeds, psnr_scene.MODES[mode])
    r10 = psnr_scene.paired_report(run["psnr50"], ref["psnr10"])          # ("psnr50" = after spec["early"] iterations)
    r20 = psnr_scene.paired_report(run["psnr300"], ref["psnr20"])         # ("psnr300" = after spec["steps"])
    print(mode, "hidden 256, paired, 10 iterations:", r10)
    print(mode, "hidden 256, paired, 20 iterations:", r20)
    assert r10["n"] >= 32 and 15.0 < r10["ref_mean_db"] < 25.0 and r20["ref_mean_db"] > r10["ref_mean_db"] + 3.0
    # the bounds: psnr_scene.PAIRED_GATE (one table for this test and for tools/h256_handicap.py, which records the gate
    # FAILING under a degraded gradient: profiles/r06_h256_handicap.txt)
    bad = psnr_scene.paired_gate_failures(mode, r10, r20)
    assert not bad, (bad, r10, r20)
