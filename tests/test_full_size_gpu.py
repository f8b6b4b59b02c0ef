"""BASELINE.json's full-size window (config B: 8 keyframes x 2000 points, R = 14000) on the device, checked through
size-independent properties (the oracle comparison at this size is tests/test_ba_parity_gpu.py[B]):
  * index maps are exact partitions of the residual set;
  * H_A, H_sc symmetric, H_sc positive semi-definite, the solve's x satisfies the assembled system (backward error);
  * backup -> step -> restore is a bitwise round trip of the inverse depths;
  * the device-resident Gauss-Newton loop lowers the photometric energy and is bit-reproducible."""
import numpy as np
import pytest

from libcml_amd import device, host, synth
from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def window_b():
    I = S.make_inputs("B")
    assert (I.N, I.P, I.R) == (8, 2000, 14000)
    return I


def test_full_size_accumulate_solve_properties(window_b):
    I = window_b
    ctx = D.make_ctx(I)
    try:
        m = ctx.ba_index_maps()
        assert np.array_equal(np.sort(m["by_point"]), np.arange(I.R)) and np.array_equal(np.sort(m["by_pair"]), np.arange(I.R))
        assert m["by_point_off"][-1] == I.R and m["by_pair_off"][-1] == I.R and np.all(np.diff(m["by_point_off"]) >= 0)
        host_of = I.points["host"][I.residuals["point"]]
        assert np.array_equal(m["pair_of"], host_of + I.residuals["target"] * I.N)          # htIDX, BA.cpp:1677
        r = ctx.ba_linearize()
        assert r.n_in + r.n_oob + r.n_outlier == I.R and r.n_in > I.R // 3 and np.isfinite(r.energy)
        ctx.ba_apply(1)
        HA, bA, HL, bL, Hsc, bsc = D.accumulate(ctx, I)
        assert np.abs(HA - HA.T).max() <= 1e-9 * np.abs(HA).max()
        assert np.array_equal(Hsc, Hsc.T)
        ev = np.linalg.eigvalsh(Hsc[4:, 4:])
        assert ev.min() > -1e-7 * ev.max()
        # H_A - H_sc is the marginal of a Gram matrix: positive semi-definite up to fp32 accumulation noise
        evm = np.linalg.eigvalsh((HA - Hsc)[4:, 4:])
        assert evm.min() > -1e-4 * evm.max()
        lam = 1e-5
        x, rc = ctx.ba_solve(lam)
        assert rc == 0 and np.all(x[:4] == 0)
        n = 8 * I.N + 4
        H = HL + HA
        H[np.diag_indices(n)] *= (1 + lam)
        H = H - Hsc / (1 + lam)
        b = bL + bA - bsc
        Sv = 1.0 / np.sqrt(np.diag(H) + 10.0)
        res = Sv[4:] * (H[4:, 4:] @ x[4:] - b[4:])
        assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(Sv[4:] * b[4:])
        # point steps: finite, and the round trip backup -> step -> restore is exact
        before = ctx.ba_get_idepth().copy()
        ctx.ba_backup_points()
        step, rcb = ctx.ba_backsub(x)
        assert rcb == 0 and np.all(np.isfinite(step))
        sums = ctx.ba_step_points()
        after = ctx.ba_get_idepth()
        moved = after != before
        assert moved.sum() > I.P // 2 and sums[2] >= moved.sum()              # numID counts every point whose new inverse depth was accepted
        ctx.ck(ctx.L.cmlhip_ba_restore_points(ctx.h))
        back = ctx.ba_get_idepth()
        assert np.array_equal(back.astype(np.float32), before.astype(np.float32))            # the backup is fp32 (BA.cpp:919-922)
    finally:
        ctx.close()


def test_full_size_resident_loop_descends_and_repeats(window_b):
    I = window_b
    outs = []
    for rep in range(2):
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 6)
        assert ba.run_resident(), ba.last_error()
        e = ba.energies(16)
        idp, alive, ng = ba.points()
        outs.append((e.copy(), idp.copy(), np.concatenate([ba.frame(k)["state"] for k in range(I.N)])))
        ba.close(); ctx.close()
    e = outs[0][0]
    assert len(e) >= 2 and np.all(np.isfinite(e)) and e[-1] < e[0] * I.R             # e[0] is energy / #residuals (BA.cpp:798)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
