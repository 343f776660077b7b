"""SURVEY section 8d Track S on the MI355X: ONE StageNet (stage_idx 3: C = G = 8, CostRegNet since D > 8) at a large literal D -
cfg1 (640x512, V = 3, D = 48: BASELINE configs[0]'s shape) and the optional D = 192 at 1152x1536 (340 Mvoxel, possible since round 5:
no 2 GB ceiling).  Wall time per forward call on one stream, SURVEY 8d's algorithmic bytes (fp32 volume model) / time vs 8 TB/s.
    gpurun -- 'python scripts/track_s_bench.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import cost_volume, synth
from mvsformerplusplus_amd.cost_volume import StageNet

cost_volume.F16_SATURATION_CHECK_EVERY = 0          # no synchronising self-check (8th call) inside the timed loops

dev = torch.device("cuda:0")
ARGS = {"base_ch": [8, 8, 8, 8], "depth_type": ["ce"] * 4, "fusion_type": "cnn", "cost_reg_type": ["Normal"] * 4}
for name, H, W, V, D, reps in (("cfg1  640x512  V=3 D=48 ", 512, 640, 3, 48, 20), ("TrackS 1152x1536 V=3 D=192", 1152, 1536, 3, 192, 5),
                               ("TrackS 1152x1536 V=5 D=192", 1152, 1536, 5, 192, 5)):
    for prec in (None, "final_stage", "f16mix"):       # the default policy; the same with args["final_stage"] (round 6); the explicit fp16 format
        st = StageNet(dict(ARGS, **({"final_stage": True} if prec == "final_stage" else {"conv_precision": prec} if prec else {})), D, 3)
        st.load_state_dict(synth.seeded_state_dict(synth.state_dict_manifest(st.state_dict()), 5), strict=True)
        st = st.eval().to(dev)
        st.return_prob_volumes = True
        cams = synth.make_cameras(V, H, W, baseline=20.0, seed=0)
        proj = synth.stage_proj_matrices(cams, 1)["stage1"].to(dev)
        feats = torch.randn(1, V, 8, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
        hyp = torch.linspace(425.0, 935.0, D, device=dev).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
        with torch.no_grad():
            for _ in range(2):
                st(feats, proj, hyp, tmp=1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                st(feats, proj, hyp, tmp=1.0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
        nvox = D * H * W
        # SURVEY 8d byte model: features once + hypotheses + volume write + regulariser R = 51 floats / voxel (CostRegNet) + head 2 reads + outputs
        algo = V * 8 * H * W * 4 + nvox * 4 + 8 * nvox * 4 + 51 * nvox * 4 + 2 * nvox * 4 + 2 * H * W * 4
        print("%s %-8s %8.2f ms per StageNet call   %.2f GB algorithmic (SURVEY 8d)  ->  %.0f GB/s = %.1f %% of 8 TB/s   peak memory %.1f GB" % (
            name, (st.precision_policy if prec is None else "final_st" if prec == "final_stage" else prec), ms, algo / 1e9, algo / ms / 1e6, algo / ms / 1e6 / 80.0, torch.cuda.max_memory_allocated() / 1e9), flush=True)
        del st, feats, hyp
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
