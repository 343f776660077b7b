"""Every layer of CostRegNet3D at cfg2's stage-4 / stage-3 shapes in the product format of the fine stages ("f16mix": fp16 activations, two fp16
weight terms on the 8- / 16-channel layers, one on conv4 .. conv7), HIP-event timing, best of three interleaved runs.  MVS_HIP_LIB selects a variant
library for A/B runs (round 6: -DMVS_ZSKIP=0).  Prints us per launch and the stage sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import _lib, ops, packing

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


def prec_of(cin, cout):
    return _lib.PREC_F16 if (min(cin, cout) >= 32 or max(cin, cout) >= 64) else _lib.PREC_F16X2      # module.py: the "f16mix" rule


for stage, (D, H, W) in (("stage 4", (4, 1152, 1536)), ("stage 3", (8, 576, 768))):
    tot = 0.0
    convs = [(8, 16, (1, 2, 2), 1), (16, 16, (1, 1, 1), 2), (16, 32, (1, 2, 2), 2), (32, 32, (1, 1, 1), 4), (32, 64, (1, 2, 2), 4), (64, 64, (1, 1, 1), 8)]
    for ci, co, stride, down in convs:
        w = torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1
        bias = torch.randn(64, generator=g).to(dev)
        x = torch.randn(1, D, H // down, W // down, ci, generator=g).half().to(dev)
        wp = packing.f16x2(packing.pack_conv_weights_bf16x3, w, packing.conv_chunk(ci, stride)).to(dev)
        p = prec_of(ci, co)
        t = timed(lambda: ops.conv3d_bn_relu(x, wp, bias, co, 3, stride, True, p))
        tot += t
        print("%s conv   %2d->%2d s%d%d%d  %6.1f us" % (stage, ci, co, *stride, t), flush=True)
        del x
    for ci, co, down in ((64, 32, 8), (32, 16, 4), (16, 8, 2)):
        w = torch.randn(ci, co, 3, 3, 3, generator=g) * 0.1
        bias = torch.randn(64, generator=g).to(dev)
        x = torch.randn(1, D, H // down, W // down, ci, generator=g).half().to(dev)
        skip = torch.randn(1, D, 2 * (H // down), 2 * (W // down), co, generator=g).half().to(dev)
        wp = packing.f16x2(packing.pack_deconv_weights_bf16x3, w, 1).to(dev)
        p = prec_of(ci, co)
        if co == 8:
            pw, pb = torch.randn(8, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
            t = timed(lambda: ops.deconv3d_prob(x, wp, bias, 1, skip, pw, pb, p))
        else:
            t = timed(lambda: ops.deconv3d_bn_relu_add(x, wp, bias, co, 1, skip, p))
        tot += t
        print("%s deconv %2d->%2d s122  %6.1f us%s" % (stage, ci, co, t, " (+ prob)" if co == 8 else ""), flush=True)
        del x, skip
    print("%s U-Net sum %7.1f us" % (stage, tot), flush=True)
