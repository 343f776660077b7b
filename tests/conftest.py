import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# No session-wide precision override.  A stage built without an explicit conv_precision runs the PRODUCT DEFAULT
# POLICY ("stagemix" since round 5: coarse stages fp32-equivalent, fine stages "f16mix"; a bare regulariser / layer wrapper: "f16mix"); the
# golden / oracle cases are parametrised over PRECS = [None (= the default), "bf16x3" (the fp32-equivalent mode)] and assert per-mode
# bounds (parity_cases.tol).
from mvsformerplusplus_amd import cost_volume as _cv  # noqa: E402
PRODUCT_DEFAULT_PRECISION = _cv.STAGE_DEFAULT_PRECISION
PRECS = [None, "bf16x3"]
PRECS_ALL = [None, "bf16x3", "f16mix", "f16x2", "f16"]   # + the uniform fp16 formats (opt-in since round 5: "f16mix" on every stage was round 4's default)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_golden(name):
    """-> dict of torch tensors / python scalars from tests/golden/<name> (plus "__name__" = the file name)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {"__name__": name}
    for k in z.files:
        a = z[k]
        if a.dtype.kind in "US":
            out[k] = [str(s) for s in a.tolist()] if a.ndim else str(a)
        elif a.ndim == 0:
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a)
    return out


_WEIGHT_SUMS = None


def golden_weights(fx, prefix="w.", name=None):
    """Rebuild the deterministic weights a fixture was generated with (manifest + seed stored in it).  The fixtures do not carry the
    weight arrays; so that they do not silently depend on the generator's reproducibility, the SHA-256 of every regenerated state
    dict is pinned in tests/golden/weights_sha256.json (written when the fixtures were made): a torch RNG change fails HERE, loudly,
    instead of as a numeric mismatch somewhere downstream.  `name` = the fixture file (checked when given)."""
    import hashlib
    from mvsformerplusplus_amd import synth
    global _WEIGHT_SUMS
    name = name or fx.get("__name__")
    keys = fx[prefix + "keys"]
    shapes = [tuple(json.loads(s)) for s in fx[prefix + "shapes"]]
    sd = synth.seeded_state_dict(dict(zip(keys, shapes)), int(fx[prefix + "seed"]))
    if name is not None:
        if _WEIGHT_SUMS is None:
            _WEIGHT_SUMS = json.load(open(os.path.join(GOLDEN, "weights_sha256.json")))
        h = hashlib.sha256()
        for k in sorted(sd):
            h.update(k.encode())
            h.update(sd[k].contiguous().numpy().tobytes())
        want = _WEIGHT_SUMS[name + ":" + prefix]
        assert h.hexdigest() == want, ("the regenerated weights of %s (%s) differ from the ones the fixture was generated with: "
                                       "torch's CPU generator changed - regenerate the fixtures (tests/golden/make_golden.py)" % (name, prefix))
    return sd


def rel_l1(a, b):
    return float((a - b).abs().div(b.abs().clamp_min(1e-12)).mean())


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def emu_lib():
    """Host-emulated twin of libmvs_hip.so (tests/hipemu): the same kernel sources compiled for x86."""
    import hipemu_build
    from mvsformerplusplus_amd import _lib
    return _lib.bind(hipemu_build.build())


@pytest.fixture
def emu(monkeypatch, emu_lib):
    """Route the package's C-ABI calls to the emulated library for one test (CPU tensors allowed)."""
    from mvsformerplusplus_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", emu_lib)
    monkeypatch.setattr(_lib, "_REQUIRE_DEVICE", False)
    return "cpu"
