"""Numerics study (CPU, oracle only; VERDICT r3 item 2 "error study on the oracle first"): how far does the final depth of the SHIPPED
regulariser mix (stage-1 transformer + PE3D) move when the attention core runs like the reference's flash-attn
(dino/layers/attention.py:141-170: q, k, v and the probabilities in ONE 16-bit term, fp32 accumulation) instead of fp32?
    python scripts/study_attention_precision.py [H W]
Emulated exactly as the kernel computes: q pre-multiplied by scale * log2(e) and rounded, k rounded, scores fp32, p = exp2(s - m)
rounded for the p.v product (the row sum l uses the unrounded fp32 p), v rounded, fp32 accumulation.  `lazy` = the running maximum
lags the true one by up to 2^8 (the kernel raises it only when a score exceeds it by more than 8): p <= 256 before rounding."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from conftest import rel_l1
from oracle import ref_path as O
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 512)
TCFG = {"base_channel": 8, "mid_channel": 64, "num_heads": 4, "down_rate": [2, 4, 4], "mlp_ratio": 4, "layer_num": 6, "drop": 0.0, "attn_drop": 0.0,
        "position_encoding": True, "attention_type": "FLASH2", "softmax_scale": "entropy_invariance", "train_avg_length": 12185, "use_pe_proj": True}
ARGS = dict(P.ARGS, cost_reg_type=["PureTransformerCostReg", "Normal", "Normal", "Normal"], use_pe3d=True, transformer_config=[TCFG])
NDEPTHS, RATIO = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0]
RND = {"bf16": lambda x: x.bfloat16().float(), "fp16": lambda x: x.half().float(), "fp32": lambda x: x}
_sdpa = F.scaled_dot_product_attention
STATS = {}


def make_sdpa(fmt_qk, fmt_pv, lazy):
    rq, rp = RND[fmt_qk], RND[fmt_pv]

    def sdpa(q, k, v, scale=None, **kw):
        B, Hh, N, hd = q.shape
        q2 = rq(q * (scale * 1.4426950408889634))
        k2, v2 = rq(k), rp(v)
        STATS["max|q|"] = max(STATS.get("max|q|", 0.0), float(q2.abs().max()))
        STATS["max|k|"] = max(STATS.get("max|k|", 0.0), float(k2.abs().max()))
        STATS["max|v|"] = max(STATS.get("max|v|", 0.0), float(v2.abs().max()))
        out = torch.empty_like(q)
        for i in range(0, N, 2048):
            s = q2[:, :, i:i + 2048] @ k2.transpose(-2, -1)
            m = s.max(-1, keepdim=True).values
            STATS["max|s|"] = max(STATS.get("max|s|", 0.0), float(s.abs().max()))
            if lazy:
                m = m - 8.0 * torch.rand_like(m)                # the stale maximum the lazy rule allows
            p = torch.exp2(s - m)
            out[:, :, i:i + 2048] = (rp(p) @ v2) / p.sum(-1, keepdim=True)
        return out
    return sdpa


def state_dicts(peaky, seed=11):
    sds = []
    for i in range(4):
        net = StageNet(json.loads(json.dumps(ARGS)), NDEPTHS[i], i)
        sd = synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), seed + i)
        if peaky:
            sd["cost_reg.prob.weight"] = sd["cost_reg.prob.weight"] * 30.0
        sds.append(sd)
    return sds


MODES = [("fp16 q,k / bf16 p,v (round 4 default)", "fp16", "bf16", False), ("fp16 q,k / bf16 p,v, lazy max", "fp16", "bf16", True),
         ("fp32 q,k / bf16 p,v (round 3 'bf16p')", "fp32", "bf16", False), ("bf16 q,k,p,v (reference flash-attn)", "bf16", "bf16", False),
         ("bf16 q,k,p,v, lazy max", "bf16", "bf16", True), ("fp16 q,k,p,v", "fp16", "fp16", False), ("fp16 q,k,p,v, lazy max", "fp16", "fp16", True)]
for peaky in (False, True):
    sds = state_dicts(peaky)
    for seed in (2, 5):
        feats, projs, dv = synth.make_cascade_inputs(H, W, 5, seed=seed, rot_deg=1.0)
        run = lambda: O.cascade_forward(feats, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"],
                                        use_pe3d=True, transformer_config=[TCFG])
        with torch.no_grad():
            ref = run()
            for name, fqk, fpv, lazy in MODES:
                F.scaled_dot_product_attention = make_sdpa(fqk, fpv, lazy)
                try:
                    res = run()
                finally:
                    F.scaled_dot_product_attention = _sdpa
                errs = [rel_l1(res["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]) for s in range(1, 5)]
                lg = float((res["stage1"]["prob_volume_pre"] - ref["stage1"]["prob_volume_pre"]).abs().max())
                print("peaky=%d seed=%d  %-40s refined depth rel-L1 %.2e  stages %s  stage-1 logits max abs %.1e  conf mean abs %.1e" % (
                    peaky, seed, name, rel_l1(res["refined_depth"], ref["refined_depth"]), " ".join("%.1e" % e for e in errs), lg,
                    float((res["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())), flush=True)
print("operand magnitudes seen (fp16 range 65504):", {k: round(v, 2) for k, v in STATS.items()})
