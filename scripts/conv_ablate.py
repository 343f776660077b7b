"""Experiment: what is the time of the MFMA convolution kernels made of?  Builds variants of the library with parts of the conv
kernels disabled (-DMVS_ABL=k: 1 no activation loads, 2 no weight loads after step 1, 3 no LDS operand reads after step 1,
4 one MFMA term of three, 5 no output stores, 6 one contraction step only) and times the stage-3/4 layers with each.
  python scripts/conv_ablate.py build      (build container)
  python scripts/conv_ablate.py run        (GPU box)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scripts", "abl")
VARIANTS = [v for v in os.environ.get("ABL_VARIANTS", "0,1,2,3,4,5,6").split(",")]      # "k" -> -DMVS_ABL=k;  "name:-DX=1:-DY=2" -> those flags


def _flags(v):
    return ["-DMVS_ABL=%s" % v] if v.isdigit() else v.split(":")[1:]


def _tag(v):
    return v if v.isdigit() else v.split(":")[0]
FILE = os.environ.get("ABL_FILE", "conv_bf16x3_kernels.hip")


def build():
    from mvsformerplusplus_amd import build as b
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(b.CSRC, "*.hip")))
    objs, procs = [], []
    for s in srcs:
        if os.path.basename(s) == FILE:
            continue
        o = os.path.join("/tmp", "abl_" + os.path.basename(s)[:-4] + ".o")
        procs.append(subprocess.Popen([b.HIPCC] + b.FLAGS + b.FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]))
        objs.append(o)
    vobjs = []
    for k in VARIANTS:
        o = "/tmp/abl_conv_%s.o" % _tag(k)
        procs.append(subprocess.Popen([b.HIPCC] + b.FLAGS + _flags(k) + ["-c", os.path.join(b.CSRC, FILE), "-o", o]))
        vobjs.append(o)
    for p in procs:
        assert p.wait() == 0
    for k, o in zip(VARIANTS, vobjs):
        subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libmvs_abl%s.so" % _tag(k))] + objs + [o])
    print("built", OUT)


def run():
    import torch
    from mvsformerplusplus_amd import _lib, ops, packing
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    # (name, kind, cin, cout, stride, D, H, W) at the input of the layer, cfg2 stages 4 and 3
    layers = [("conv 8->16 s122  st4", "c", 8, 16, (1, 2, 2), 4, 1152, 1536), ("conv 16->16 s111 st4", "c", 16, 16, (1, 1, 1), 4, 576, 768),
              ("conv 16->32 s122 st4", "c", 16, 32, (1, 2, 2), 4, 576, 768), ("conv 32->32 s111 st4", "c", 32, 32, (1, 1, 1), 4, 288, 384),
              ("conv 32->64 s122 st4", "c", 32, 64, (1, 2, 2), 4, 288, 384), ("conv 64->64 s111 st4", "c", 64, 64, (1, 1, 1), 4, 144, 192),
              ("deconv 64->32 s122 st4", "d", 64, 32, 1, 4, 144, 192), ("deconv 32->16 s122 st4", "d", 32, 16, 1, 4, 288, 384),
              ("deconv 16->8 s122 st4", "d", 16, 8, 1, 4, 576, 768), ("conv 16->16 s111 st3", "c", 16, 16, (1, 1, 1), 8, 288, 384),
              ("conv 16->16 s222 st2", "c", 16, 16, (1, 1, 1), 8, 72, 96), ("conv 8->16 s122 st3", "c", 8, 16, (1, 2, 2), 8, 576, 768),
              ("conv 8->16 s222 st2", "c", 8, 16, (2, 2, 2), 16, 288, 384), ("conv 8->16 s222 st1", "c", 8, 16, (2, 2, 2), 32, 144, 192),
              ("deconv 16->8 s122 st3", "d", 16, 8, 1, 8, 288, 384), ("deconv 16->8 s222 st2", "d", 16, 8, 2, 8, 144, 192),
              ("deconv 16->8+prob st4", "p", 16, 8, 1, 4, 576, 768), ("deconv 16->8+prob st3", "p", 16, 8, 1, 8, 288, 384),
              ("prob head 8->1 st2", "h", 8, 1, (1, 1, 1), 16, 288, 384), ("prob head 8->1 st1", "h", 8, 1, (1, 1, 1), 32, 144, 192),
              ("conv 64->64 s111 st1", "c", 64, 64, (1, 1, 1), 4, 18, 24), ("conv 32->64 s222 st1", "c", 32, 64, (2, 2, 2), 8, 36, 48),
              ("deconv 64->32 s222 st1", "d", 64, 32, 2, 4, 18, 24), ("conv 32->32 s111 st1", "c", 32, 32, (1, 1, 1), 8, 36, 48)]
    PREC, conv_in = 1, (lambda t: t.to(dev))
    if os.environ.get("ABL_SET") == "coarse":
        # round 5: the ten U-Net launches of the COARSE stages (stage 1: [32,144,192], stage 2: [16,288,384]) in the format the default
        # policy runs them in - split bf16 (MVS_PREC_BF16X3_SPLIT): what are these latency-bound launches made of?
        layers = []
        for st, (D0, H0, W0) in (("st1", (32, 144, 192)), ("st2", (16, 288, 384))):
            layers += [("conv1 8->16 s222 " + st, "c", 8, 16, (2, 2, 2), D0, H0, W0), ("conv2 16->16 " + st, "c", 16, 16, (1, 1, 1), D0 // 2, H0 // 2, W0 // 2),
                       ("conv3 16->32 s222 " + st, "c", 16, 32, (2, 2, 2), D0 // 2, H0 // 2, W0 // 2), ("conv4 32->32 " + st, "c", 32, 32, (1, 1, 1), D0 // 4, H0 // 4, W0 // 4),
                       ("conv5 32->64 s222 " + st, "c", 32, 64, (2, 2, 2), D0 // 4, H0 // 4, W0 // 4), ("conv6 64->64 " + st, "c", 64, 64, (1, 1, 1), D0 // 8, H0 // 8, W0 // 8),
                       ("conv7 deconv 64->32 " + st, "d", 64, 32, 2, D0 // 8, H0 // 8, W0 // 8), ("conv9 deconv 32->16 " + st, "d", 32, 16, 2, D0 // 4, H0 // 4, W0 // 4),
                       ("conv11 deconv 16->8 " + st, "d", 16, 8, 2, D0 // 2, H0 // 2, W0 // 2), ("prob head 8->1 " + st, "h", 8, 1, (1, 1, 1), D0, H0, W0)]
        PREC, conv_in = _lib.PREC_BF16X3_SPLIT, (lambda t: ops.to_split(t.to(dev)))
    res = {}
    for k in VARIANTS:
        _lib._LIB = _lib.bind(os.path.join(OUT, "libmvs_abl%s.so" % _tag(k)))
        for name, kind, cin, cout, stride, D, H, W in layers:
            x = conv_in(torch.randn(1, D, H, W, cin, generator=g))
            if kind == "h":
                w16 = torch.zeros(16, 8, 3, 3, 3)
                w16[0] = torch.randn(8, 3, 3, 3, generator=g) * 0.05
                wp = packing.pack_conv_weights_bf16x3(w16, 8).to(dev)
                bias = torch.zeros(16, device=dev)
                f = lambda: ops.conv3d_logits(x, wp, bias, PREC)
            elif kind == "c":
                w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05
                wp = packing.pack_conv_weights_bf16x3(w, _ch_of(cin, cout, stride)).to(dev)
                bias = torch.zeros(cout, device=dev)
                f = lambda: ops.conv3d_bn_relu(x, wp, bias, cout, 3, stride, True, PREC)
            else:
                w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.05
                wp = packing.pack_deconv_weights_bf16x3(w, stride).to(dev)
                bias = torch.zeros(cout, device=dev)
                skip = conv_in(torch.randn(1, D * stride, 2 * H, 2 * W, cout, generator=g))
                f = lambda: ops.deconv3d_bn_relu_add(x, wp, bias, cout, stride, skip, PREC)
                if kind == "p":
                    pw, pb = torch.randn(8, device=dev), torch.zeros(1, device=dev)
                    f = lambda: ops.deconv3d_prob(x, wp, bias, stride, skip, pw, pb, PREC)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[(name, k)] = e0.elapsed_time(e1) / n * 1e3
            del x
    print("%-26s" % "us per launch" + "".join("%9s" % _tag(k)[:8] for k in VARIANTS))
    for name, *_ in layers:
        print("%-26s" % name + "".join("%9.1f" % res[(name, k)] for k in VARIANTS))


def _ch_of(cin, cout, stride):
    return 16 if stride == (1, 1, 1) else 8


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
