"""Randomised differential test on the GPU box: StageNet (HIP) vs the oracle on random shapes, view counts, batch sizes, feature
dtypes, regulariser kinds and camera rigs.  Not part of the pytest suites (run time is unbounded by design):
    gpurun -- 'python scripts/fuzz_gpu.py 60'      # number of cases
Prints one line per case and a summary; exits non-zero when a case breaks the 1e-3 depth bar or raises."""
import os, sys, random, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet
from oracle import ref_path as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
torch.set_num_threads(16)
TCFG = {"base_channel": 8, "mid_channel": 64, "num_heads": 4, "down_rate": [2, 4, 4], "mlp_ratio": 4, "layer_num": 2, "drop": 0.0, "attn_drop": 0.0,
        "position_encoding": True, "attention_type": "FLASH2", "softmax_scale": "entropy_invariance", "train_avg_length": 12185, "use_pe_proj": True}
bad = 0
for case in range(n_cases):
    kind = rnd.choice(["unet", "unet", "unet3d", "unet3d", "transformer"])
    D = {"unet": rnd.choice([16, 32, 48]), "unet3d": rnd.choice([4, 8]), "transformer": rnd.choice([8, 16])}[kind]
    H, W = 8 * rnd.randint(2, 10), 8 * rnd.randint(2, 14)
    if kind == "transformer":
        H, W = 4 * rnd.randint(4, 12), 4 * rnd.randint(4, 16)
    V, B = rnd.randint(2, 5), rnd.choice([1, 1, 2])
    C = rnd.choice([8, 16, 32, 64])
    dt = rnd.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
    depth_type = rnd.choice(["ce", "ce", "reg"])
    rot, base = rnd.uniform(0, 6), rnd.uniform(10, 80)
    desc = "%-11s D=%-2d %3dx%-3d V=%d B=%d C=%-2d %-8s %s rot=%.1f base=%.0f" % (kind, D, H, W, V, B, C, str(dt).split('.')[-1], depth_type, rot, base)
    try:
        args = {"base_ch": [8] * 4, "depth_type": [depth_type] * 4, "fusion_type": "cnn", "model_th": 8,
                "cost_reg_type": (["PureTransformerCostReg"] + ["Normal"] * 3) if kind == "transformer" else ["Normal"] * 4,
                "transformer_config": [dict(TCFG)]}
        st = StageNet(args, D, 0)
        st.load_state_dict(synth.seeded_state_dict(synth.state_dict_manifest(st.state_dict()), 1000 + case), strict=True)
        st = st.eval().to(dev)
        cams = synth.make_cameras(V, H, W, baseline=base, rot_deg=rot, seed=case, batch=B)
        feats = synth.make_features(cams, C, H, W, dmin=480.0, dmax=880.0, seed=case).to(dt)
        g = torch.Generator().manual_seed(case)
        hyp = ((1.0 / torch.linspace(1 / 900.0, 1 / 440.0, D))[None, :, None, None] * (1 + 0.03 * torch.rand(B, D, H, W, generator=g))).contiguous()
        pos = None
        if kind == "transformer" and rnd.random() < 0.7:
            dv = torch.linspace(425.0, 935.0, 192)[None].repeat(B, 1)
            pos = O.get_position_3d(H, W, cams[:, 0, 1, :3, :3], hyp, dv.min(), dv.max())[0]
        tmp = rnd.choice([1.0, 5.0])
        sd = {k: v.cpu() for k, v in st.state_dict().items()}
        with torch.no_grad():
            ref = O.stage_forward(feats, cams, hyp, tmp, sd, G=8, depth_type=depth_type, position3d=pos, transformer_config=TCFG)
            out = st(feats.to(dev), cams.to(dev), hyp.to(dev), tmp=tmp, position3d=None if pos is None else pos.to(dev))
        r = float(((out["depth"].cpu() - ref["depth"]).abs() / ref["depth"].abs()).mean())
        c = float((out["photometric_confidence"].cpu() - ref["photometric_confidence"]).abs().mean())
        ok = r <= 1e-3 and c <= 2e-3 and bool(torch.isfinite(out["depth"]).all())
        print("%s  %s  depth rel-L1 %.2e  conf mean-abs %.2e" % ("ok  " if ok else "FAIL", desc, r, c), flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        bad += 1
        print("EXC   %s  %s" % (desc, repr(e)[:200]), flush=True)
        traceback.print_exc()
print("%d cases, %d bad" % (n_cases, bad))
sys.exit(1 if bad else 0)
