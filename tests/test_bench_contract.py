"""CPU: the bench line committed under profiles/ (produced by bench.py on an MI355X) carries every field of the driver's
contract, with consistent arithmetic.  Guards the JSON schema without needing a GPU."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r01_bench_1gpu.json", "r01_bench_1gpu_shipped.json", "r02_bench_1gpu.json", "r02_bench_1gpu_shipped.json",
                                  "r03_bench_1gpu.json", "r04_bench_1gpu.json", "r04_bench_1gpu_shipped.json"])
def test_committed_bench_line(name):
    r = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "ref-views/s" and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["data"] == "synthetic" and r["dtype"].startswith("f32") and "workload" in r["config"] and "model" not in r["config"]
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
    if name.startswith("r04"):
        # the round-4 record: counter traffic present, the fraction against the guide's dense MFMA peak, no per-kernel
        # bandwidth above the 8 TB/s roof, the whole path with its three fractions
        assert ro["traffic"] is not None and ro["traffic"] > 0 and ro["peak"] == 2500.0 and ro["unit"] == "TFLOP/s"
        wp = r["whole_path"]
        assert all(0 < wp[k] < 1 for k in ("frac", "frac_as_built", "frac_pmc"))
        if not wp["pmc_symbols_without_counters"]:          # every launch counted: the counter bytes cannot be below the bytes the tensors have
            assert wp["frac_as_built"] <= wp["frac_pmc"]
        assert r["kernels"] and all(v.get("gbs", 0) <= 8000.0 for v in r["kernels"].values())
