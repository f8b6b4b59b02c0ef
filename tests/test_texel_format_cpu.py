"""CPU side of the config-E statement (tests/e_texel_check.py): what rounding the level-0 texels to half precision does to the reference's arithmetic,
measured by the oracle alone (oracle on rounded images against oracle on the fp32 images of the same window).  The GPU test
(tests/test_config_e_gpu.py::test_config_e_against_fp32_texels) makes the same statement for the device's fp16-texel run at full size; per residual the
device equals the oracle on rounded images in every bit, so the two statements differ by fp32 accumulation order only."""
import numpy as np

from tests import ba_setup as S
from tests import e_texel_check as T


def test_compare_is_zero_on_identical_runs_and_small_on_fp16_texels():
    I = S.make_inputs("small")
    st32, Hs32 = T.oracle_side(I)
    same = T.compare(I, st32, Hs32, st32, Hs32)
    assert same["class_flips"] == 0 and same["H_A_rel"] == 0.0 and same["H_sc_jacobi_rel"] == 0.0 and same["x_gauge_free_rel"] == 0.0
    fp32 = [I.grads[k][0] for k in range(I.N)]
    try:
        for k in range(I.N):
            I.grads[k][0] = fp32[k].astype(np.float16).astype(np.float32)
        st16, Hs16 = T.oracle_side(I)
    finally:
        for k in range(I.N):
            I.grads[k][0] = fp32[k]
    rep = T.compare(I, st16, Hs16, st32, Hs32)
    # half-precision texels: intensities up to 255 carry an ulp of 0.125 grey levels — a few residuals at a threshold change class, the
    # matrices move by 1e-3 of their largest entry (config E at full size: 6.5e-4 / 1.4e-3 Jacobi-scaled, tests/test_config_e_gpu.py)
    assert rep["class_flips"] <= max(4, rep["R"] // 300), rep
    assert 0.0 < rep["H_A_rel"] < 5e-3 and 0.0 < rep["H_sc_rel"] < 5e-3, rep
    assert rep["H_A_jacobi_rel"] < 1e-2 and rep["energy_rel_median"] < 2e-2, rep
    assert np.isfinite(rep["x_gauge_free_rel"]) and rep["x_gauge_free_rel"] < 0.2, rep
