"""Randomised differential test of the whole cascade on the GPU box (CascadeDepthHead in the PRODUCT DEFAULT regulariser format vs
oracle.cascade_forward), both regulariser mixes, a third of the cases with the x30-logits stress weights:
    gpurun -- 'python scripts/fuzz_cascade_gpu.py 30 [seed] [conv_precision]'"""
import os, sys, random, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cascade import CascadeDepthHead
from oracle import ref_path as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
prec = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
torch.set_num_threads(16)
bad = 0
for case in range(n_cases):
    shipped = rnd.random() < 0.5
    H, W, V, B = 64 * rnd.randint(1, 4), 64 * rnd.randint(1, 5), rnd.randint(2, 6), rnd.choice([1, 1, 2])
    dt = rnd.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
    peaky = rnd.random() < 0.33
    args = json.loads(json.dumps(dict(bench.ARGS, **bench.SHIPPED))) if shipped else dict(bench.ARGS)
    if prec:
        args["conv_precision"] = prec
    head = CascadeDepthHead(args)
    for i, st in enumerate(head.fusions):
        sd = synth.seeded_state_dict(synth.state_dict_manifest(st.state_dict()), 500 + 7 * case + i)
        if peaky:
            sd["cost_reg.prob.weight"] = sd["cost_reg.prob.weight"] * 30.0
        st.load_state_dict(sd, strict=True)
    head = head.eval().to(dev)
    feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=case, rot_deg=rnd.uniform(0, 3), baseline=rnd.uniform(15, 60), batch=B, feat_dtype=dt)
    sds = [{k: v.cpu() for k, v in st.state_dict().items()} for st in head.fusions]
    with torch.no_grad():
        ref = O.cascade_forward({k: v.float() for k, v in feats.items()}, projs, dv, sds, ndepths=args["ndepths"], depth_interals_ratio=args["depth_interals_ratio"],
                                base_ch=args["base_ch"], use_pe3d=shipped, transformer_config=bench.SHIPPED["transformer_config"] if shipped else None)
        out = head({k: v.to(dev) for k, v in feats.items()}, {k: v.to(dev) for k, v in projs.items()}, dv.to(dev))
    r = float(((out["refined_depth"].cpu() - ref["refined_depth"]).abs() / ref["refined_depth"].abs()).mean())
    c = float((out["photometric_confidence"].cpu() - ref["photometric_confidence"]).abs().mean())
    ok = r <= 1e-3 and (peaky or c <= 2e-3)              # the bar is on depth; the confidence of the stress cases is only reported
    print("%s  %-7s %-5s %3dx%-3d V=%d B=%d %-8s %s depth rel-L1 %.2e conf mean-abs %.2e" % ("ok  " if ok else "FAIL", "shipped" if shipped else "normal", "x30" if peaky else "", H, W, V, B,
                                                                                             str(dt).split('.')[-1], head.fusions[-1].conv_precision, r, c), flush=True)
    bad += 0 if ok else 1
print("%d cases, %d bad" % (n_cases, bad))
sys.exit(1 if bad else 0)
