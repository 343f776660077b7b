"""Build libmvs_hip.so (gfx950) in-tree with hipcc.  `python -m mvsformerplusplus_amd.build`.

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(mvsformerplusplus_amd/csrc/libmvs_hip.so) so that it travels with a repo snapshot to the GPU box; it is
git-ignored.  No CPU fallback exists: the package refuses to work without this library.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmvs_hip.so")
STAMP = os.path.join(CSRC, ".libmvs_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value"]
# per-file extras.  transformer_kernels: MFMA results are consumed by VALU code right away (scores -> softmax), so the
# accumulators must live in VGPRs (the default AGPR form costs a v_accvgpr move per value and direction); no NaN can
# occur in the softmax (masked scores are -inf, never inf - inf), which lets max chains fold into v_max3_f32.
FILE_FLAGS = {"transformer_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
              # attention_f16_kernels: as above; -fno-slp-vectorize keeps the row-sum adds single instructions (MI355X_MICROARCH.md: packed f32 VALU
              # beside MFMAs costs more than the two scalar ops it replaces)
              "attention_f16_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans", "-fno-slp-vectorize"],
              # vis_kernels: every MFMA result of the row-streaming CNN goes straight into a VALU epilogue (bias, ReLU, bf16 split, 1x1 +
              # sigmoid): VGPR accumulators save 40 v_accvgpr moves per row and wave (18 % of the kernel's non-MFMA VALU instructions)
              # -fno-honor-nans: ReLU on an MFMA result is then ONE v_max_f32 (with NaNs honoured the compiler first canonicalises the
              # accumulator, a second v_max per value: 8 of 63 VALU per row and wave)
              "vis_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]}


def _digest():
    h = hashlib.sha1((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False, extra_flags=(), out=None):
    """out: build a VARIANT of the library (measurement scripts: other -D switches) next to the product library; never stamped."""
    dig = _digest()
    if out is None and not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs, procs = [], []
    tag = "" if out is None else "." + os.path.basename(out)
    for s in srcs:
        o = s[:-4] + tag + ".o"
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + list(extra_flags) + ["-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        log = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), log))
        if verbose and log.strip():
            print(log)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs)
    for o in objs:
        os.remove(o)
    if out is None:
        with open(STAMP, "w") as f:
            f.write(dig)
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
