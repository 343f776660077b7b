"""Per-stage, per-kernel HIP-event times of one cascade pass (which launches are small-grid / latency-bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import profiling, synth
dev = torch.device("cuda:0")
head = bench.build_head(dev)
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev)
for _ in range(3):
    _, L = profiling.profile_cascade(head, feats, projs, dv, bench.TMP)
acc = {}
for _ in range(5):
    _, L = profiling.profile_cascade(head, feats, projs, dv, bench.TMP)
    for l in L:
        acc.setdefault((l.stage, l.kernel), []).append(l.ms)
tot = {}
for (st, k), v in sorted(acc.items()):
    ms = sum(v) / 5
    tot[st] = tot.get(st, 0) + ms
    print("stage %d  %-36s %7.3f ms" % (st + 1, k, ms))
print({k + 1: round(v, 3) for k, v in tot.items()})
