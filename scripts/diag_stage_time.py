"""Diagnostic (GPU): wall time of each cascade stage on one stream, per precision policy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from mvsformerplusplus_amd import synth, ops
dev = torch.device("cuda:0")
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev)
for pol in (sys.argv[1:] or ["stagemix", "f16mix", "bf16x3"]):
    head = bench.build_head(dev, conv_precision=pol)
    n = len(head.ndepths)
    with torch.no_grad():
        out = head(feats, projs, dv)
        hyps = [out["stage%d" % (s + 1)]["depth_values"] for s in range(n)]
        from mvsformerplusplus_amd import cost_volume as _cv
        _cv.F16_SATURATION_CHECK_EVERY = 0                  # no synchronising self-check inside the timed loops
        ts = []
        for s in range(n):
            st = head.fusions[s]
            f, p = feats["stage%d" % (s + 1)], projs["stage%d" % (s + 1)]
            for _ in range(3): st(f, p, hyps[s], 1.0)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): st(f, p, hyps[s], 1.0)
            torch.cuda.synchronize(); ts.append((time.time() - t0) / 20 * 1e3)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): head(feats, projs, dv)
        torch.cuda.synchronize(); tw = (time.time() - t0) / 10 * 1e3
    print("%-9s stages %s  sum %.3f  cascade %.3f ms" % (pol, " ".join("%.3f" % t for t in ts), sum(ts), tw), [st.conv_precision for st in head.fusions], flush=True)
