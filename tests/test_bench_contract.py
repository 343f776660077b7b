"""CPU: the bench line committed under profiles/ (produced by bench.py on an MI355X) carries every field of the driver's
contract, with consistent arithmetic.  Guards the JSON schema without needing a GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r01_bench_1gpu.json", "r01_bench_1gpu_shipped.json", "r02_bench_1gpu.json", "r02_bench_1gpu_shipped.json",
                                  "r03_bench_1gpu.json", "r04_bench_1gpu.json", "r04_bench_1gpu_shipped.json", "r05_bench_1gpu.json",
                                  "r05_bench_1gpu_shipped.json", "r06_bench_1gpu.json", "r06_bench_1gpu_shipped.json"])
def test_committed_bench_line(name):
    r = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "ref-views/s" and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    # `dtype` = the arithmetic the path computes in.  Rounds 1-4 wrote "f32 (...)"; VERDICT r4: the default's operands ARE fp16 - round 5 says so
    assert r["dtype"].startswith("fp16 operands/storage, fp32 accumulate" if name[:3] in ("r05", "r06") else "f32")
    assert r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - r["n_gpus"] * r["config"]["global_batch"] / r["n_gpus"] * 1e3 / r["ms_per_step"]) <= 1e-6 * r["value"]
    ro = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] in ("hbm", "mfma") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    assert r["parity"]["refined_depth_rel_l1_vs_oracle"] <= r["parity"]["bar"] == 1e-3
    if name == "r05_bench_1gpu.json":
        # the round-5 record: graph replay, the shipped mix and the uniform fp16 format in the same line, the roofline
        # ranked by kernel function with its instantiations listed, five head / range launches
        assert r["config"]["issue"].startswith("one hipGraph replay") and r["config"]["conv_precision"].startswith("stagemix (product default")
        assert r["shipped"]["value"] > 0 and r["shipped"]["parity"]["refined_depth_rel_l1_vs_oracle"] <= 1e-3
        assert r["uniform_f16mix_mode"]["value"] > 0 and r["fp32_equivalent_mode"]["value"] > 0
        assert len(ro["instantiations"]) > 1 and ro["kernel"].endswith("_kernel")
        assert r["families"]["heads_and_ranges"]["launches_per_ref_view"] <= 6
        assert "feature_emitter" in r and "stages" in r["feature_emitter"]
        # last session: the same policy fed with fp16 octet tiles at the fine stages (direct gather) - same arithmetic, reported beside the headline
        ft = r["fp16_tiles_handoff_mode"]
        assert ft["value"] > r["value"] and ft["default_vs_this_refined_depth_rel_l1"] <= 1e-5
    if name == "r06_bench_1gpu.json":
        # the round-6 record: the default policy names what it chose, its other branch and the fp32-equivalent format ride in
        # the same line, and the reference's composite PyTorch path on the same GPU stands beside the CPU baseline
        assert r["config"]["precision_policy"] == "auto -> f16mix" and r["config"]["conv_precision"].startswith("auto (the cascade's default policy")
        assert r["config"]["issue"].startswith("one hipGraph replay")
        assert 0 < r["exact_coarse_mode"]["value"] < r["value"] and 0 < r["fp32_equivalent_mode"]["value"] < r["exact_coarse_mode"]["value"]
        tc = r["torch_rocm_composite"]
        assert tc["unit"] == "ref-views/s" and 0 < tc["value"] < r["value"] and tc["hip_path_vs_this_refined_depth_rel_l1"] <= 1e-3
        assert r["shipped"]["value"] > 0 and r["shipped"]["parity"]["refined_depth_rel_l1_vs_oracle"] <= 1e-3
        assert r["fp16_tiles_handoff_mode"]["value"] > r["value"]
        assert len(ro["instantiations"]) > 1 and ro["kernel"].endswith("_kernel")
    if name[:3] in ("r04", "r05", "r06"):
        # the round-4 record: counter traffic present, the fraction against the guide's dense MFMA peak, no per-kernel
        # bandwidth above the 8 TB/s roof, the whole path with its three fractions
        assert ro["traffic"] is not None and ro["traffic"] > 0 and ro["peak"] == 2500.0 and ro["unit"] == "TFLOP/s"
        wp = r["whole_path"]
        assert all(0 < wp[k] < 1 for k in ("frac", "frac_as_built", "frac_pmc"))
        if not wp["pmc_symbols_without_counters"]:          # every launch counted: the counter bytes cannot be below the bytes the tensors have
            assert wp["frac_as_built"] <= wp["frac_pmc"]
        assert r["kernels"] and all(v.get("gbs", 0) <= 8000.0 for v in r["kernels"].values())
