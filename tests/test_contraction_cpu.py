"""The reading the parity tests pin (oracle compiled -ffp-contract=off) against the arithmetic of the reference's Release build
(-O3 -march=native, fused multiply-adds at the compiler's discretion): same oracle sources, both builds, one linearization + system
of the 320x240 window.  Holds the first-pass gap to what tools/contraction_sensitivity.py reports (profiles/round2_contraction_sensitivity.txt):
decisions identical, last-bit differences in a few per cent of the residual energies, 1e-7-level differences in the accumulated system."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "contraction_sensitivity.py")


def test_first_pass_gap_between_contraction_off_and_release_arithmetic(tmp_path):
    outs = []
    for so, tgt in (("libcml_oracle.so", "libcml_oracle.so"), ("libcml_oracle_contract.so", "contract")):
        out = str(tmp_path / (tgt.replace(".", "_") + ".npz"))
        subprocess.check_call([sys.executable, TOOL, "--child", so, tgt, "small", "1", out], cwd=ROOT)
        outs.append(np.load(out))
    A, B = outs
    assert (A["state_0"] == B["state_0"]).all()
    ea, eb = A["energy_0"].astype(np.float64), B["energy_0"].astype(np.float64)
    assert np.max(np.abs(ea - eb) / np.maximum(np.abs(ea), 1e-30)) < 1e-3
    for k, tol in (("HA_0", 1e-6), ("bA_0", 1e-5), ("Hsc_0", 1e-6), ("bsc_0", 1e-5)):
        assert np.linalg.norm(A[k] - B[k]) / np.linalg.norm(A[k]) < tol, k
