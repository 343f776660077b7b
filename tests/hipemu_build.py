"""TEST INFRASTRUCTURE: build the host-emulation twin of libmvs_hip.so (tests/hipemu/README in hip_runtime.h).

Compiles the unchanged csrc/*.hip sources for x86 against tests/hipemu/hip/hip_runtime.h.  The product package
never loads this library; tests opt in through the `emu_lib` fixture."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvsformerplusplus_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "hipemu")
OUT = os.path.join(EMU, "_build")
CLANG = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(EMU, "hipemu.cpp")]
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.join(EMU, "hip", "hip_runtime.h")]
    h = hashlib.sha1()
    for d in sorted(deps):
        h.update(open(d, "rb").read())
    tag = h.hexdigest()[:16]
    lib = os.path.join(OUT, "libmvs_hip_emu_%s.so" % tag)
    if os.path.exists(lib):
        return lib
    # one builder at a time (pytest -n N: every worker's session fixture lands here at once and they would delete each other's objects)
    import fcntl
    with open(os.path.join(OUT, ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return lib if os.path.exists(lib) else _build_locked(srcs, tag, lib, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(srcs, tag, lib, verbose):
    for old in glob.glob(os.path.join(OUT, "libmvs_hip_emu_*.so")):
        os.remove(old)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + "." + tag + ".o")
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-g0", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
               "-I", EMU, "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError("hipemu compile failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out:
            print(out)
    subprocess.check_call([CLANG, "-shared", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
