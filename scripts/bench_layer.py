"""U-Net layers on their own at the cfg2 stage-4 / stage-3 shapes, HIP-event timing, best of three interleaved runs: the split-bf16
activation format (MVS_PREC_BF16X3_SPLIT) against the fp16 activation format (MVS_PREC_F16X2).  MVS_HIP_LIB selects a variant library
(A/B of kernel changes; profiles/r03_conv_march_ab.txt was measured with an earlier form of this script)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import _lib, ops, packing

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
LAYERS = [  # (cin, cout, stride, [B, D, H, W] input shape)
    (16, 16, (1, 1, 1), (1, 4, 576, 768)), (16, 16, (1, 1, 1), (1, 8, 288, 384)),
    (32, 32, (1, 1, 1), (1, 4, 288, 384)), (64, 64, (1, 1, 1), (1, 4, 144, 192)),
    (16, 32, (1, 2, 2), (1, 4, 576, 768)), (32, 64, (1, 2, 2), (1, 4, 288, 384)), (8, 16, (1, 2, 2), (1, 4, 1152, 1536)),
]
for ci, co, stride, shape in LAYERS:
    w = torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1
    bias = torch.randn(64, generator=g).to(dev)
    x32 = torch.randn(*shape, ci, generator=g)
    forms = {"split": (_lib.PREC_BF16X3_SPLIT, ops.to_split(x32).to(dev), packing.pack_conv_weights_bf16x3(w, packing.conv_chunk(ci, stride)).to(dev)),
             "f16x2": (_lib.PREC_F16X2, x32.half().to(dev), packing.f16x2(packing.pack_conv_weights_bf16x3, w, packing.conv_chunk(ci, stride)).to(dev)),
             "f16 (one term)": (_lib.PREC_F16, x32.half().to(dev), packing.f16x2(packing.pack_conv_weights_bf16x3, w, packing.conv_chunk(ci, stride)).to(dev))}
    res = {}
    for rep in range(3):
        for name, (prec, x, wp) in forms.items():
            try:
                for _ in range(3):
                    y = ops.conv3d_bn_relu(x, wp, bias, co, 3, stride, True, prec)
            except _lib.MvsHipError as e:
                res[name] = None
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = ops.conv3d_bn_relu(x, wp, bias, co, 3, stride, True, prec)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    nvox_out = y.numel() // co
    flop = 2.0 * 27 * ci * co * nvox_out
    print("conv %2d->%2d s%d%d%d %-18s " % (ci, co, *stride, shape) + "   ".join(
        "%s %6.1f us (%5.0f TF)" % (n, min(t), flop / min(t) * 1e-6) if t else "%s n/a" % n for n, t in res.items()), flush=True)
