"""Counter-collection driver: the two gather passes at every stage of the bench workload with the cascade's real hypotheses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import ops, synth
dev = torch.device("cuda:0")
head = bench.build_head(dev)
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev)
with torch.no_grad():
    out = head(feats, projs, dv, tmp=bench.TMP)
torch.cuda.synchronize()
for s in range(4):
    key = "stage%d" % (s + 1)
    f, code = ops._feat(feats[key])
    hyp = out[key]["depth_values"].contiguous()
    hom = ops.compose_homography(projs[key])
    B, V, C, H, W = f.shape
    vis = torch.rand(B, V - 1, H, W, device=dev)
    for rep in range(3):
        ops.warp_corr_entropy(f, code, hom, hyp, 8)
        ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8)
torch.cuda.synchronize()
print("done")
