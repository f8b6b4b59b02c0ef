"""The residual kernel of the device-resident loop (k_ba_lin_rs: pair-sorted residuals, 4 lanes per residual, Jacobians kept in
reduced form) against the record-writing kernel k_ba_linearize — which tests/test_ba_parity_gpu.py and tests/test_config_e_gpu.py
hold bit-exact against the oracle — on identical device state:
  * per-residual outputs (states, energies, JpJdF, centre projection): BIT-EXACT (same expressions in the same order);
  * the pair blocks summed from the kernel's matrix-core tiles against the blocks summed from the records, and everything
    downstream (H_A, b_A, H_sc, b_sc): fp32 accumulation-order tolerance;
  * the 74-float records re-created on demand (cml_materialize_records): BIT-EXACT against the records the other kernel wrote."""
import os

import numpy as np
import pytest

from libcml_amd import abi
from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


def _make(I, use_rs, texel_format, tile):
    if use_rs:
        os.environ.pop("CMLHIP_NO_RS", None)
    else:
        os.environ["CMLHIP_NO_RS"] = "1"
    os.environ["CMLHIP_RS_TILE"] = str(tile)          # both contexts: the tile size also fixes the summation order of the record path
    try:
        ctx = D.make_ctx(I, texel_format=texel_format)
    finally:
        os.environ.pop("CMLHIP_NO_RS", None)
        os.environ.pop("CMLHIP_RS_TILE", None)
    return ctx


ODD = (4, 300, 323, 241, 3, 260.0, 260.0, 161.0, 120.0)       # a window on images whose sides are not multiples of 4
MID = (8, 5000, 640, 480, 3, 525.0, 525.0, 319.5, 239.5)      # R = 35 000: with 16-residual tiles more than two workgroups per CU, the register-bounded instantiation of k_ba_lin_rs4


# tile 16 = k_ba_lin_rs4 (4 lanes per residual, small windows), tile 64 = k_ba_lin_rs (one lane per residual, large windows);
# the library picks by window size, the test forces each on every window
@pytest.mark.parametrize("tile", [16, 64])
# (B with fp16 texels: 1241 is not a multiple of the 4-texel tile width of the tiled level 0 the lane-per-residual kernel gathers from)
#  and an image whose width AND height are not multiples of the 4 x 4 tile: the clamped border tiles)
@pytest.mark.parametrize("config,half", [("tiny", False), ("small", False), ("small", True), ("B", False), ("B", True), (ODD, True), (MID, False)])
def test_resident_kernel_matches_record_kernel(config, half, tile):
    I = S.make_inputs(config)
    if half:
        for k in range(I.N):
            for lvl in range(len(I.grads[k])):
                I.grads[k][lvl] = I.grads[k][lvl].astype(np.float16).astype(np.float32)
    fmt = abi.TEXEL_F16 if half else abi.TEXEL_F32
    a, b = _make(I, False, fmt, tile), _make(I, True, fmt, tile)          # a: record-writing kernel in the loop, b: resident kernel
    try:
        for c in (a, b):
            c.ba_linearize(); c.ba_apply(1)
            D.accumulate(c, I)                                 # leaves adjoints / priors on the device
            c.ba_iteration_async(1e-5)                         # K3..K6 identical in both (records path), then the residual kernel under test
            c.sync()
        sa, sb = a.ba_states(), b.ba_states()
        for k in ("state", "new_state", "good"):
            assert np.array_equal(sa[k], sb[k]), k
        for k in ("energy", "new_energy", "new_energy_wo"):
            assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k
        g = sa["good"] == 1
        assert g.sum() > 10
        assert np.array_equal(a.ba_get_idepth().view(np.uint64), b.ba_get_idepth().view(np.uint64))
        assert np.array_equal(a.ba_jpjdf()[g].view(np.uint32), b.ba_jpjdf()[g].view(np.uint32)), "JpJdF differs"
        IN = sa["new_state"] == 0
        assert np.array_equal(a.ba_center()[IN].view(np.uint32), b.ba_center()[IN].view(np.uint32))
        # accumulate: records (a) vs matrix-core tiles of the resident kernel (b)
        Ha, Hb = D.accumulate(a, I), D.accumulate(b, I)
        pa, pb = a.ba_pair_acc(0), b.ba_pair_acc(0)
        for q in range(I.N * I.N):
            assert np.abs(pa[q] - pb[q]).max() <= 2e-5 * max(np.abs(pa[q]).max(), 1e-30), q
        # ... and in fact in every bit: the tiles are accumulated in the slot order of the record path, and the 256-thread accumulate
        # launch of the resident loop (k_ba_acc_rs: 4 waves carry the 16 running sums) adds them in the order of the 1024-thread one
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), "pair blocks: records / k_ba_acc vs tiles / k_ba_acc_rs"
        assert D.rel(Hb[0], Ha[0]) < 2e-5 and D.rel(Hb[1], Ha[1]) < 2e-5
        assert D.rel(Hb[4], Ha[4]) < 5e-5 and D.rel(Hb[5], Ha[5]) < 5e-5
        qa, qb = a.ba_point_acc(), b.ba_point_acc()
        assert np.abs(qa - qb).max() <= 2e-5 * np.abs(qa).max()
        # the records the resident kernel did not write, re-created on demand
        assert np.array_equal(a.ba_rj(1)[g].view(np.uint32), b.ba_rj(1)[g].view(np.uint32)), "re-materialised records differ"
        # and the loop goes on from there identically in structure: one more iteration, same classification
        for c in (a, b):
            c.ba_iteration_async(1e-5); c.sync()
        sa2, sb2 = a.ba_states(), b.ba_states()
        assert (sa2["new_state"] != sb2["new_state"]).sum() <= max(2, I.R // 2000)      # inputs now differ by fp32 accumulation noise
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("config", ["small", "B"])
def test_two_launch_forms_of_the_4_lane_kernel_agree(config):
    """k_ba_lin_rs4 is launched over (tile group, pair) with the pair table in the kernel arguments when the window has at most 128
    pairs, and over a tile table in memory otherwise (CMLHIP_RS4_1D=1 forces that form): same body, same bits — states, energies,
    JpJdF, centre projections, pair blocks and the state after two more iterations."""
    I = S.make_inputs(config)
    out = []
    for one_d in (False, True):
        if one_d:
            os.environ["CMLHIP_RS4_1D"] = "1"
        try:
            c = _make(I, True, abi.TEXEL_F32, 16)
            try:
                c.ba_linearize(); c.ba_apply(1)
                D.accumulate(c, I)
                for _ in range(3):
                    c.ba_iteration_async(1e-5)
                c.sync()
                st = c.ba_states()
                D.accumulate(c, I)
                out.append((st, c.ba_jpjdf().copy(), c.ba_center().copy(), c.ba_get_idepth().copy(), c.ba_pair_acc(0).copy()))
            finally:
                c.close()
        finally:
            os.environ.pop("CMLHIP_RS4_1D", None)
    a, b = out
    for k in ("state", "new_state", "good"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for k in ("energy", "new_energy", "new_energy_wo"):
        assert np.array_equal(a[0][k].view(np.uint32), b[0][k].view(np.uint32)), k
    g = a[0]["good"] == 1
    assert g.sum() > 10
    assert np.array_equal(a[1][g].view(np.uint32), b[1][g].view(np.uint32))
    IN = a[0]["new_state"] == 0
    assert np.array_equal(a[2][IN].view(np.uint32), b[2][IN].view(np.uint32))
    assert np.array_equal(a[3].view(np.uint64), b[3].view(np.uint64))
    assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))
