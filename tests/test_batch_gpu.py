"""cmlhip_ba_iteration_batch (several windows per launch): S independent windows stepped by five launches
must end in exactly the state S solo loops (cmlhip_ba_iteration_async per window) end in — same kernel bodies, same arguments, so
every bit — and every window's residual pass must replay bit for bit on the oracle (tests/resident_check.py)."""
import numpy as np
import pytest

from libcml_amd import abi, device, host, synth
from tests import resident_check as RC

pytestmark = pytest.mark.gpu


def _window(config, shard, seed=0xC0FFEE, texel_format=abi.TEXEL_F32):
    W = synth.make_window(config, seed=seed, shard=shard)
    ctx = device.Ctx(max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N, texel_format=texel_format)
    ba = host.window_to_host_ba(ctx, W, image_id_base=1000 * (shard + 1), levels=1)
    ba.set_param("iterations", 1)
    assert ba.run(), ba.last_error()
    ctx.refresh_window_size()
    assert ba.begin_resident(), ba.last_error()
    return W, ctx, ba


def _state(ctx, N):
    st = ctx.ba_states()
    fs = (abi.BAFrameState * N)()
    ctx.ck(ctx.L.cmlhip_ba_get_resident_state(ctx.h, fs, None, None))
    return st, ctx.ba_get_idepth().copy(), ctx.ba_jpjdf().copy(), np.frombuffer(bytes(fs), np.uint8).copy()


# windows of different shapes in one batch (they must share the point-slice class of the Schur SYRK: all P <= 512 here)
MIXED = [("small", 0), (("M0", (5, 400, 480, 360, 4, 390.0, 390.0, 239.5, 179.5)), 1), ("small", 2), (("M1", (6, 500, 400, 300, 3, 330.0, 330.0, 199.5, 149.5)), 3)]


# two windows in the THROUGHPUT regime of the residual kernel (R = 42 000 >= 36 k: tiles of 64, lane per residual; batched through
# k_ba_lin_rs_batch): refused until round 3
THROUGHPUT = [(("T%d" % k, (8, 6000, 1241, 376, 4, 718.856, 718.856, 606.69, 184.72)), k) for k in range(2)]


@pytest.mark.parametrize("spec", ["mixed", "B4", "T2", "T2h"])
def test_batched_iterations_equal_solo_iterations_bit_for_bit(spec):
    wins = MIXED if spec == "mixed" else (THROUGHPUT if spec in ("T2", "T2h") else [("B", k) for k in range(4)])
    its = 3 if spec in ("T2", "T2h") else 7
    fmt = abi.TEXEL_F16 if spec == "T2h" else abi.TEXEL_F32              # T2h: fp16 texels, gathered from the tiled level 0
    solo, batch = [], []
    for cfg, shard in wins:
        cfg = cfg[1] if isinstance(cfg, tuple) else cfg
        solo.append(_window(cfg, shard, texel_format=fmt)); batch.append(_window(cfg, shard, texel_format=fmt))
    try:
        for W, ctx, ba in solo:
            for _ in range(its):
                ctx.ba_iteration_async(1e-5)
            ctx.sync()
        bctx = [b[1] for b in batch]
        for _ in range(its):
            device.ba_iteration_batch(bctx, 1e-5)
        bctx[0].sync()
        for (W, cs, _), (_, cb, _) in zip(solo, batch):
            a, b = _state(cs, W.N), _state(cb, W.N)
            for k in ("state", "new_state", "good"):
                assert np.array_equal(a[0][k], b[0][k]), k
            for k in ("energy", "new_energy", "new_energy_wo"):
                assert np.array_equal(a[0][k].view(np.uint32), b[0][k].view(np.uint32)), k
            assert np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)), "inverse depths"
            assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), "JpJdF"
            assert np.array_equal(a[3], b[3]), "frame states"
            assert (a[0]["good"] == 1).sum() > 0.3 * len(a[0]["good"])
        # and the batched pass itself against the oracle
        replays = [RC.make_replay(cb, bb, W) for (W, cb, bb) in batch]
        reps = RC.check_one_batched_pass(bctx, replays, 1e-5, with_records=True)
        for r in reps:
            assert r["ok"], r
        for r in replays:
            r.close()
    finally:
        for W, ctx, ba in solo + batch:
            ba.close(); ctx.close()


def test_batch_refuses_what_it_does_not_take():
    W, ctx, ba = _window("small", 0)
    try:
        c2 = device.Ctx(max_frames=4, max_points=10, max_residuals=10)
        with pytest.raises(device.CmlHipError):                  # a context without a window
            device.ba_iteration_batch([ctx, c2], 1e-5)
        with pytest.raises(device.CmlHipError):                  # the same context twice
            device.ba_iteration_batch([ctx, ctx], 1e-5)
        c2.close()
        W2, ctx2, ba2 = _window("medium", 1)                     # 600 points: another point-slice class than "small" (300)
        try:
            with pytest.raises(device.CmlHipError):
                device.ba_iteration_batch([ctx, ctx2], 1e-5)
        finally:
            ba2.close(); ctx2.close()
    finally:
        ba.close(); ctx.close()
