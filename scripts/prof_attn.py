"""A/B driver for the transformer's attention core alone: qkv projection + attention at the cfg2 stage-1 token count (27 648).
    python scripts/prof_attn.py [n]
Times the split-bf16 forms of rounds 1-3 and every tile-shape variant of the fp16 kernel (MVS_ATTN_VARIANT, csrc/attention_f16_kernels.hip),
and checks each against float64 attention on a 2 048-token prefix problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import _lib, ops, packing

only = None
if "--only" in sys.argv:                      # PMC passes: just one fp16 variant, a few launches
    i = sys.argv.index("--only"); only = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 27648
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, n, 64, generator=g)
w0 = torch.randn(192, 64, generator=g) * 0.125
w = packing.pack_linear_bf16x3(w0).to(dev)
xd = x.to(dev)
ns = 2048
qkv = (x[:, :ns].double() @ w0.double().t()).reshape(1, ns, 3, 4, 16).permute(2, 0, 3, 1, 4)
ref = (torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * 0.27, -1) @ qkv[2]).transpose(1, 2).reshape(1, ns, 64).float()


def timeit(code, reps=6):
    for _ in range(2):
        y = ops.tr_attention(xd, w, 4, 0.27, _lib.PREC_BF16X3, code)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        y = ops.tr_attention(xd, w, 4, 0.27, _lib.PREC_BF16X3, code)
    e.record()
    torch.cuda.synchronize()
    err = float((ops.tr_attention(xd[:, :ns].contiguous(), w, 4, 0.27, _lib.PREC_BF16X3, code).cpu() - ref).abs().max())
    return s.elapsed_time(e) / reps, err


if only is not None:
    os.environ["MVS_ATTN_VARIANT"] = str(only)
    print("variant %d: %.3f ms" % ((only,) + timeit(_lib.PREC_ATTN16, reps=3)[:1]))
    sys.exit(0)
print("n = %d tokens, 4 heads x 16; qkv projection + attention per layer; error = max abs vs float64 on a %d-token problem (|out| <= %.2f)" % (n, ns, float(ref.abs().max())))
for name, code in (("bf16x3 (rounds 1-3)", None), ("bf16p", _lib.PREC_BF16P)):
    t, err = timeit(code)
    print("%-44s %.3f ms   err %.2e" % (name, t, err))
names = {0: "default", 1: "QT=1 KB=128", 2: "QT=2 KB=128", 3: "QT=4 KB=128", 4: "QT=2 KB=256", 5: "QT=1 KB=256", 6: "QT=4 KB=256"}
for v in sorted(names):
    os.environ["MVS_ATTN_VARIANT"] = str(v)
    t, err = timeit(_lib.PREC_ATTN16)
    print("attn16 variant %d  %-30s %.3f ms   err %.2e" % (v, names[v], t, err))
abl = {18: "no exponentials", 20: "no staging after block 0", 52: "no staging, no barrier", 24: "no p.v / row-sum MFMA", 32: "no score MFMA",
       40: "no MFMA at all", 80: "row sums by VALU adds (correct results)"}
print("ablations of variant 2 (QT=2 KB=128); results are wrong by construction, time only - they exist only in a library built with\n"
      "-DMVS_ATTN_ABLATIONS (build.build(extra_flags=['-DMVS_ATTN_ABLATIONS'], out=...) + MVS_HIP_LIB); the shipped library runs the default for these numbers:")
for v, name in abl.items():
    os.environ["MVS_ATTN_VARIANT"] = str(v)
    t, _ = timeit(_lib.PREC_ATTN16)
    print("  %-34s %.3f ms" % (name, t))
os.environ.pop("MVS_ATTN_VARIANT", None)
