"""Randomised differential test of the depth-map filters on the GPU box (HIP vs oracle/fusion_ref.py):
    gpurun -- 'python scripts/fuzz_fusion_gpu.py 200'"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import fusion as Fu, synth
from oracle import fusion_ref as FR

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for case in range(n_cases):
    h, w, v, n = rnd.randint(6, 80), rnd.randint(6, 100), rnd.randint(2, 10), 1          # the reference driver logic (test.py:472-474) only broadcasts correctly for n = 1
    rot, base = rnd.uniform(0, 4), rnd.uniform(5, 60)
    g = torch.Generator().manual_seed(case)
    cams = synth.make_cameras(v + 1, h, w, baseline=base, rot_deg=rot, seed=case, batch=n)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    surf = 600 + 50 * torch.sin(2 * xx + case) * torch.cos(1.5 * yy)
    d = surf[None, None].repeat(n, v + 1, 1, 1) * (1 + rnd.choice([0.0005, 0.002, 0.01]) * torch.randn(n, v + 1, h, w, generator=g))
    d[:, 1:, :, : rnd.randint(0, 3)] = 0.0
    conf = torch.rand(n, v + 1, h, w, generator=g)
    rd, sd, rc, sc = d[:, :1].contiguous(), d[:, 1:, None].contiguous(), cams[:, 0].contiguous(), cams[:, 1:].contiguous()
    ct, td, tv = rnd.uniform(0.1, 0.6), rnd.choice([0.5, 1.0, 2.0]), rnd.randint(1, v)
    a = FR.filter_depth(rd, conf[:, 0], sd, conf[:, 1:], rc, sc, conf_thresh=ct, thres_disp=td, thres_view=tv)
    b = Fu.filter_depth(rd.to(dev), conf[:, 0].to(dev), sd.to(dev), conf[:, 1:].to(dev), rc.to(dev), sc.to(dev), conf_thresh=ct, thres_disp=td, thres_view=tv)
    c = FR.dynamic_filter_depth(rd, conf[:, 0], sd, rc, sc, conf_thresh=ct)
    e = Fu.dynamic_filter_depth(rd.to(dev), conf[:, 0].to(dev), sd.to(dev), rc.to(dev), sc.to(dev), conf_thresh=ct)
    res = []
    for ref, got in ((a, b), (c, e)):
        mm = float((ref["mask"] != got["mask"].cpu()).float().mean())
        same = (ref["geo_mask"] == got["geo_mask"].cpu()) & ((ref["depth"] - got["depth"].cpu()).abs() < 0.5)
        # a per-view mask that flips at its threshold moves the averaged depth of that pixel by a fraction of the noise: count them
        dd = float(((ref["depth"] - got["depth"].cpu()).abs()[same] > 5e-3).float().mean()) if same.any() else 0.0
        res.append((mm, 1 - float(same.float().mean()), dd))
    ok = all(m <= 1e-2 and s <= 2e-2 and dd <= 2e-3 for m, s, dd in res)
    if not ok or case < 5:
        print("%s  %dx%d v=%d n=%d rot=%.1f  static: mask flips %.4f, depth outliers %.1e | dynamic: %.4f, %.1e" % ("ok  " if ok else "FAIL", h, w, v, n, rot, res[0][0], res[0][2], res[1][0], res[1][2]), flush=True)
    bad += 0 if ok else 1
print("%d cases, %d bad" % (n_cases, bad))
sys.exit(1 if bad else 0)
