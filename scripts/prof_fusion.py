"""Timing of the depth-map filters at the Tanks-and-Temples filter setting of test.py (10 source views) on 1152x1536 maps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsformerplusplus_amd import fusion as Fu, synth

dev = torch.device("cuda:0")
h, w, v = 1152, 1536, 10
cams = synth.make_cameras(v + 1, h, w, baseline=30.0, rot_deg=1.0, seed=1).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
depth = 600 + 40 * torch.rand(1, v + 1, h, w, generator=g, device=dev)
conf = torch.rand(1, v + 1, h, w, generator=g, device=dev)
rd, sd = depth[:, :1].contiguous(), depth[:, 1:, None].contiguous()
args = (rd, conf[:, 0].contiguous(), sd, cams[:, 0].contiguous(), cams[:, 1:].contiguous())
for name, fn in (("dynamic", lambda: Fu.dynamic_filter_depth(*args, conf_thresh=0.5)),
                 ("static", lambda: Fu.filter_depth(args[0], args[1], args[2], conf[:, 1:].contiguous(), args[3], args[4], conf_thresh=0.5, thres_disp=1.0, thres_view=3))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    nbytes = h * w * (4 + 4 + v * 4 + 4 + 1 + 1 + 12 + (v * 4 if name == "static" else 0))
    print("%s filter, %dx%d, v=%d: %.3f ms / reference view  (%.0f GB/s algorithmic)" % (name, h, w, v, ms, nbytes / ms / 1e6))
