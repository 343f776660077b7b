import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet
dev = torch.device("cuda:0")
for B in (1, 2, 8):
    stage, C, D, H, W, V = 0, 64, 32, 64, 80, 5
    net = StageNet({"base_ch": [8] * 4, "depth_type": ["ce"] * 4}, D, stage).to(dev).train()
    cams = synth.make_cameras(V, H, W, baseline=30.0, seed=1, batch=B).to(dev)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, V, C, H, W, generator=g).to(dev).requires_grad_(True)
    hyp = (torch.linspace(900, 450, D)[None, :, None, None] * (1 + 0.02 * torch.rand(B, D, H, W, generator=g))).to(dev).contiguous()
    def step():
        out = net(feats, cams, hyp, 1.0); out["prob_volume_pre"].square().mean().backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    t_issue = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 10
    print("B=%d: host issue %.2f ms / step, with completion %.2f ms / step" % (B, t_issue * 1e3, t_all * 1e3))
