"""One U-Net layer on its own at the cfg2 stage shapes: 16 -> 16 stride 1 in the split activation format, HIP-event timing, best of three
interleaved runs.  MVS_HIP_LIB selects a variant library (A/B of kernel changes: profiles/r03_conv_march_ab.txt was measured with this script,
its second row then timed the row-marching kernel of commit dc3a8dc through MVS_MARCH_MIN_VOXELS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import _lib, ops, packing

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
P3 = _lib.PREC_BF16X3_SPLIT
w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1
wp = packing.pack_conv_weights_bf16x3(w, packing.conv_chunk(16, (1, 1, 1))).to(dev)
bias = torch.randn(64, generator=g).to(dev)
for shape in ((1, 4, 576, 768), (1, 8, 288, 384), (1, 8, 144, 192), (1, 16, 72, 96)):
    x = ops.to_split(torch.randn(*shape, 16, generator=g)).to(dev)
    res = {}
    for rep in range(3):
        for name, thr in (("tile", "1000000000"),):
            os.environ["MVS_MARCH_MIN_VOXELS"] = thr
            for _ in range(3):
                y = ops.conv3d_bn_relu(x, wp, bias, 16, 3, (1, 1, 1), True, P3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = ops.conv3d_bn_relu(x, wp, bias, 16, 3, (1, 1, 1), True, P3)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    nvox = shape[0] * shape[1] * shape[2] * shape[3]
    flop = 2.0 * 27 * 16 * 16 * nvox
    for name in res:
        t = min(res[name])
        print("%-18s %-6s %7.1f us  (runs %s)  %6.1f TFLOP/s  %5.0f GB/s" % (shape, name, t, " ".join("%.1f" % v for v in res[name]), flop / t * 1e-6, nvox * 128 / t * 1e-3))
