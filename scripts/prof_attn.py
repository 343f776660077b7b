"""Driver for profiling the transformer kernels alone: qkv + attention at the cfg2 stage-1 token count (27 648)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import _lib, ops, packing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 27648
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, n, 64, generator=g).to(dev)
w = packing.pack_linear_bf16x3(torch.randn(192, 64, generator=g) * 0.125).to(dev)
for _ in range(3):
    y = ops.tr_attention(x, w, 4, 0.27, _lib.PREC_BF16X3)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    y = ops.tr_attention(x, w, 4, 0.27, _lib.PREC_BF16X3)
e.record()
torch.cuda.synchronize()
print("qkv+attention n=%d: %.3f ms" % (n, s.elapsed_time(e) / 5))
