"""Driver for counter collection on the warp kernels: runs warp_corr_entropy / warp_corr_aggregate of stage 1 (C=64, D=32)
and stage 4 (C=8, D=4) of the bench workload, 3 times each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import ops, synth

dev = "cuda"
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev)
for stage, D in ((1, 32), (4, 4)):
    f, code = ops._feat(feats["stage%d" % stage])
    B, V, C, H, W = f.shape
    hom = ops.compose_homography(projs["stage%d" % stage])
    hyp = ops.init_range(dv, D, H, W, inverse=True)
    if stage == 4:      # narrow hypothesis window like the real stage 4 (~0.8 % of depth)
        hyp = (600.0 * (1 + 0.002 * torch.arange(D, device=dev).float()))[None, :, None, None].expand(1, D, H, W).contiguous()
    vis = torch.rand(B, V - 1, H, W, device=dev)
    for rep in range(3):
        ops.warp_corr_entropy(f, code, hom, hyp, 8)
        ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8)
torch.cuda.synchronize()
print("done")
