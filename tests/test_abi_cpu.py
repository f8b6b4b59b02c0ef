"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/cmlhip.h declares; ctypes mirrors agree with the header; the product never routes through the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cmlhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmlhip_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from libcml_amd import build, device
    build.build()
    L = C.CDLL(device.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.cmlhip_abi_version() == 1
    # every declared symbol also has a ctypes prototype (tests call through those)
    unproto = [n for n in names if n not in device.PROTOTYPES]
    assert not unproto, unproto


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from libcml_amd import device
    assert device.lib().cmlhip_device_count() == 0
    with pytest.raises(device.CmlHipError):
        device.Ctx()


def test_struct_layouts_match_header_arithmetic():
    from libcml_amd import abi
    assert C.sizeof(abi.BAPair) == 26 * 8
    assert C.sizeof(abi.BAPoint) == 4 + 4 + 8 + 4 + 4 + 32 + 32 + 4 + 4      # trailing pad to 8
    assert C.sizeof(abi.BAResidual) == 16 and C.sizeof(abi.BAFrame) == 16
    assert C.sizeof(abi.TrackerResult) == 4 * 5 + 12 + 64 * 8 + 8 * 8 + 81 * 4 + 4   # pad to 8 at the end
    assert abi.RJ_FLOATS == 74


def test_product_does_not_touch_the_oracle():
    """libcml_amd/ (product) must not import, link or execute anything under oracle/ or tests/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libcml_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_lib|cml_oracle|libcml_oracle|from tests|import tests|orc_", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "libcml_amd", "libcmlhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_synthetic_window_is_deterministic_and_consistent():
    from libcml_amd import synth
    W1 = synth.make_window("tiny", seed=5)
    W2 = synth.make_window("tiny", seed=5)
    assert np.array_equal(W1.gray[1], W2.gray[1]) and np.array_equal(W1.pts, W2.pts)
    W3 = synth.make_window("tiny", seed=5, shard=1)
    assert not np.array_equal(W1.gray[1], W3.gray[1])
    assert W1.gray[0].min() >= -6 and W1.gray[0].max() <= 275 and W1.gray[0].std() > 5
    res = synth.residual_list(W1, W1.R_eval, W1.t_eval)
    assert len(res) == W1.P * (W1.N - 1)
    assert set(np.unique(res["state"])) <= {0, 1}


def test_reference_side_adapter_compiles_against_the_reference_headers():
    """tools/refcheck: the flat records of include/cmlhip.h filled from DSOPoint / DSOResidual / DSOFrame / DSOFramePrecomputed /
    DSOTracker::Residual by the adapter of INTEGRATION.md §4, compiled (-fsyntax-only) against the reference's own headers.  Build
    container only: skipped where /root/reference does not exist (the GPU box)."""
    import os
    import subprocess
    if not os.path.isdir("/root/reference/src/cml"):
        pytest.skip("reference tree absent")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["sh", os.path.join(root, "tools", "refcheck", "check.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "passed" in r.stdout


def test_no_shared_mutable_statics():
    """include/cmlhip.h promises independent, re-entrant contexts (tracker ctx and BA ctx on two host threads): the device
    layer may hold no mutable file-scope or function-scope state.  Every `static` VARIABLE in libcml_amd/csrc must be one of: a const
    cache of getenv() (written once under C++11's thread-safe static initialisation, read-only afterwards), a const table / constant,
    or a __constant__ device table.  (tests/test_threads_gpu.py runs the two-thread scenario itself on the GPU.)"""
    bad = []
    csrc = os.path.join(ROOT, "libcml_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".inc")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, f), errors="ignore"), 1):
            code = line.split("//")[0]
            for m in re.finditer(r"\bstatic\s+(?!inline\b|__device__|__host__|__global__|__forceinline__)([^;(){}]*?)\b([A-Za-z_]\w*)\s*(=|;|\[)", code):
                decl = m.group(1)
                if "(" in code[m.end():].split("=")[0] and m.group(3) != "=":
                    continue
                if re.search(r"\bconst(expr)?\b", decl):
                    continue                                   # static const T x = ... / static const char* e = getenv(...)
                bad.append("%s:%d: %s" % (f, ln, line.strip()))
    assert not bad, "mutable static state in the device layer:\n" + "\n".join(bad)
    # no global / namespace-scope mutable variables either: __device__ globals would be shared by every context of the process
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".inc")):
            for ln, line in enumerate(open(os.path.join(csrc, f), errors="ignore"), 1):
                if re.match(r"^(__device__|__managed__)\s+(?!__forceinline__|inline|static\s+(inline|__forceinline__))[\w:<> ]+\s+\w+\s*(=|;|\[)", line):
                    bad.append("%s:%d: %s" % (f, ln, line.strip()))
    assert not bad, bad
