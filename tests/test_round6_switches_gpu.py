"""Round-6 additions to the boundary, each held to what include/cmlhip.h says of it:
  * cmlhip_ba_set_resident_outputs(LEAN): everything the next pass, the accumulation and the host mirror read is bit-identical to FULL; centerProjectedTo and the
    returned energy are not stored, state_NewEnergyWithOutlier only for residuals into the newest frame;
  * cmlhip_set_device_share: a tracker batch sized for 1 / S of the device (fewer workgroups per hypothesis) gives every hypothesis the same bits;
  * cmlhip_ba_window_generation: changes with every reset / upload, not with appends or commits."""
import ctypes as C
import os

import numpy as np
import pytest

from libcml_amd import abi, device
from tests import ba_setup as S
from tests import dev_setup as D
from tests import trk_opt_setup as TO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", [16, 64])
@pytest.mark.parametrize("config,half", [("small", False), ("B", False), ("B", True)])
def test_lean_outputs_change_nothing_that_is_read(config, half, tile):
    I = S.make_inputs(config)
    if half:
        for k in range(I.N):
            for lvl in range(len(I.grads[k])):
                I.grads[k][lvl] = I.grads[k][lvl].astype(np.float16).astype(np.float32)
    fmt = abi.TEXEL_F16 if half else abi.TEXEL_F32
    os.environ["CMLHIP_RS_TILE"] = str(tile)
    try:
        full, lean = D.make_ctx(I, texel_format=fmt), D.make_ctx(I, texel_format=fmt)
    finally:
        os.environ.pop("CMLHIP_RS_TILE", None)
    try:
        lean.ba_set_resident_outputs(True)
        assert lean.ba_resident_outputs_lean() and not full.ba_resident_outputs_lean()
        for c in (full, lean):
            c.ba_linearize(); c.ba_apply(1)
            D.accumulate(c, I)
        c0 = lean.ba_center().copy(); s0 = lean.ba_states()
        for _ in range(3):
            for c in (full, lean):
                c.ba_iteration_async(1e-5)
        full.sync(); lean.sync()
        sf, sl = full.ba_states(), lean.ba_states()
        for k in ("state", "new_state", "good"):
            assert np.array_equal(sf[k], sl[k]), k
        for k in ("energy", "new_energy"):
            assert np.array_equal(sf[k].view(np.uint32), sl[k].view(np.uint32)), k
        assert np.array_equal(full.ba_jpjdf().view(np.uint32), lean.ba_jpjdf().view(np.uint32))
        assert np.array_equal(full.ba_get_idepth().view(np.uint64), lean.ba_get_idepth().view(np.uint64))
        newest = I.residuals["target"] == I.N - 1
        assert np.array_equal(sf["new_energy_wo"][newest].view(np.uint32), sl["new_energy_wo"][newest].view(np.uint32))      # setNewFrameEnergyTH's input
        other = ~newest
        assert np.array_equal(sl["new_energy_wo"][other].view(np.uint32), s0["new_energy_wo"][other].view(np.uint32))        # ... and nothing else of it was touched
        assert np.array_equal(lean.ba_center().view(np.uint32), c0.view(np.uint32))                                          # centerProjectedTo: as the record pass left it
        moved = (full.ba_center() != c0).any()
        assert moved                                                                                                         # (FULL does maintain it)
        pf, thf, _ = full.ba_pairs(); pl, thl, _ = lean.ba_pairs()
        assert np.array_equal(thf.view(np.uint32), thl.view(np.uint32)) and pf.tobytes() == pl.tobytes()                     # frameEnergyTH, DSOFramePrecomputed of the loop
    finally:
        full.close(); lean.close()


def test_tracker_batch_bits_do_not_depend_on_the_device_share():
    P = TO.make_problem("B")
    hyps = [TO.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025)), TO.perturbed(P, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
            TO.perturbed(P, (-0.01, 0.004, 0.0), (0.05, 0.0, -0.04)), TO.perturbed(P, (0.002, 0.002, -0.006), (-0.02, 0.03, 0.01))]
    out = []
    for share in (1, 4, 64):
        ctx = device.Ctx(max_frames=8)
        try:
            ctx.set_device_share(share)
            ctx.pyramid_build(501, P.W.gray[P.s.new], P.levels)
            for l in range(P.levels):
                ctx.tracker_set_reference(l, P.uvic[l])
            out.append(ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps))
        finally:
            ctx.close()
    for other in out[1:]:
        for x, y in zip(out[0], other):
            assert bytes(bytearray(np.array(x.R[:]).tobytes())) == np.array(y.R[:]).tobytes() and np.array(x.t[:]).tobytes() == np.array(y.t[:]).tobytes()
            assert x.a == y.a and x.b == y.b and x.n_steps == y.n_steps and list(x.E[:]) == list(y.E[:]) and list(x.step_accept[:x.n_steps]) == list(y.step_accept[:y.n_steps])


def test_window_generation_is_the_owner_token():
    I = S.make_inputs("tiny")
    ctx = D.make_ctx(I)
    try:
        L = ctx.L
        g = C.c_uint(0)

        def gen():
            ctx.ck(L.cmlhip_ba_window_generation(ctx.h, C.byref(g)))
            return g.value
        g0 = gen()
        ctx.ba_linearize()
        assert gen() == g0                                    # passes, commits and getters leave it alone
        ctx.ba_upload_window(I.frames_dev, I.points, I.residuals)
        g1 = gen()
        assert g1 != g0                                       # the same window again, the same sizes: still another window
        ctx.ck(L.cmlhip_ba_window_reset(ctx.h))
        assert gen() != g1
    finally:
        ctx.close()
