"""Training-path timing on one GPU: forward + backward of one cascade stage (HIP gather forward / backward, PyTorch-ROCm autograd for
the conv / BatchNorm layers) at a DTU-training-like size, with the share of the two HIP gather kernels and the peak memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from train_torch_route import stage_forward_train_torch
from mvsformerplusplus_amd import ops, synth
from mvsformerplusplus_amd.cost_volume import StageNet
dev = torch.device("cuda:0")
ARGS = {"base_ch": [8] * 4, "depth_type": ["ce"] * 4}
for stage, C, D, H, W in ((3, 8, 4, 512, 640), (2, 16, 8, 256, 320), (1, 32, 16, 128, 160), (0, 64, 32, 64, 80)):
    B, V = 2, 5
    net = StageNet(dict(ARGS), D, stage).to(dev).train()
    cams = synth.make_cameras(V, H, W, baseline=30.0, seed=1, batch=B).to(dev)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(B, V, C, H, W, generator=g).to(dev).requires_grad_(True)
    hyp = (torch.linspace(900, 450, D)[None, :, None, None] * (1 + 0.02 * torch.rand(B, D, H, W, generator=g))).to(dev).contiguous()

    def step():
        out = net(feats, cams, hyp, 1.0) if mode == "hip" else stage_forward_train_torch(net, feats, cams, hyp, 1.0)
        out["prob_volume_pre"].square().mean().backward()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    for mode in (("hip",) if os.environ.get("PROF_TRAIN_HIP_ONLY") else ("hip", "torch")):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        e[0].record()
        for _ in range(5):
            step()
        e[1].record()
        torch.cuda.synchronize()
        res[mode] = (e[0].elapsed_time(e[1]) / 5, torch.cuda.max_memory_allocated() / 2 ** 20)
    total = res["hip"][0]
    # the two HIP gather kernels on their own
    with torch.no_grad():
        f, code = ops._feat(feats.detach())
        hom = ops.compose_homography(cams)
        vis = torch.rand(B, V - 1, H, W, device=dev)
        vol, _ = ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8)
        gv = torch.randn_like(vol)
        vs = vis.sum(1).contiguous()
        t = []
        for fn in (lambda: ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8), lambda: ops.warp_corr_aggregate_bwd(f, code, hom, hyp, vis, vs, vol, gv, 8)):
            fn(); torch.cuda.synchronize()
            e[0].record()
            for _ in range(5):
                fn()
            e[1].record(); torch.cuda.synchronize()
            t.append(e[0].elapsed_time(e[1]) / 5)
    print("stage %d  B=%d V=%d C=%d D=%d %dx%d: step (fwd+bwd) %.2f ms, peak %.0f MB (U-Net through PyTorch autograd: %.2f ms, %.0f MB) | HIP aggregate fwd %.3f ms, bwd %.3f ms"
          % (stage + 1, B, V, C, D, H, W, total, res["hip"][1], res.get("torch", (0, 0))[0], res.get("torch", (0, 0))[1], t[0], t[1]))
