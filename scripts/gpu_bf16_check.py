"""Where does the time go with 2-byte feature inputs?  wall vs device time of the cascade, per feature dtype."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvsformerplusplus_amd import synth

dev = torch.device("cuda:0")
head = bench.build_head(dev)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev, feat_dtype=dt)
    print(dt, {k: (v.dtype, v.is_contiguous()) for k, v in feats.items()})
    with torch.no_grad():
        for _ in range(3):
            head(feats, projs, dv)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); s.record()
        for _ in range(10):
            head(feats, projs, dv)
        e.record(); t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("  wall %.2f ms  device span %.2f ms  host enqueue %.2f ms" % ((time.perf_counter() - t0) * 100, s.elapsed_time(e) / 10, t_host * 100))
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            head(feats, projs, dv); torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:6]
        for r in rows:
            print("   %-80s %8.3f ms x%d" % (r.key[:80], r.device_time_total / 1e3, r.count))
