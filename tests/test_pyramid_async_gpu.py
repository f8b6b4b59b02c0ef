"""cmlhip_pyramid_build_async (include/cmlhip.h): the pyramid of the NEXT frame built by the context's image worker — a host thread and a
stream of its own — while the context tracks the current one, as the reference's capture thread builds pyramids ahead of the SLAM thread
(capture/CaptureImage.cpp:39-78).  Every level must hold the bits of the synchronous build; consumers must order themselves behind a build
that is still in flight; ids may be dropped and reused at any point."""
import numpy as np
import pytest

from libcml_amd import abi, device, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("texel", [abi.TEXEL_F32, abi.TEXEL_F16])
def test_async_build_holds_the_bits_of_the_synchronous_one(texel):
    W = synth.make_window("small")
    ctx = device.Ctx(max_frames=8, texel_format=texel)
    try:
        imgs = [np.ascontiguousarray(W.gray[k % W.N] + 0.25 * k, np.float32) for k in range(6)]
        for k, g in enumerate(imgs):
            ctx.pyramid_build(100 + k, g, 4)
        ref = [[ctx.pyramid_get(100 + k, l) for l in range(4)] for k in range(len(imgs))]
        for k, g in enumerate(imgs):                             # all six handed over back to back: the worker's two staging buffers take turns
            ctx.pyramid_build_async(200 + k, g, 4)
        for k in reversed(range(len(imgs))):                     # the consumer of the LAST one asks first
            for l in range(4):
                assert np.array_equal(ctx.pyramid_get(200 + k, l), ref[k][l]), (k, l)
        # drop an image whose build nobody has waited for, reuse the id at once with another image
        ctx.pyramid_build_async(300, imgs[0], 4)
        assert ctx.pyramid_drop(300) == abi.OK
        ctx.pyramid_build_async(300, imgs[3], 4)
        assert np.array_equal(ctx.pyramid_get(300, 2), ref[3][2])
        # an id that is in the cache: rebuilt through the synchronous path, same result
        ctx.pyramid_build_async(300, imgs[5], 4)
        assert np.array_equal(ctx.pyramid_get(300, 0), ref[5][0])
    finally:
        ctx.close()


def test_tracker_consumes_a_pyramid_that_is_still_being_built():
    """the first call that names the image orders itself behind the build: same tracker evaluation as with the synchronous build"""
    from tests import trk_opt_setup as TS
    P = TS.make_problem("B")
    out = []
    for use_async in (False, True):
        ctx = device.Ctx(max_frames=8)
        try:
            for l in range(P.levels):
                ctx.tracker_set_reference(l, P.uvic[l])
            g = np.ascontiguousarray(P.W.gray[P.s.new], np.float32)
            (ctx.pyramid_build_async if use_async else ctx.pyramid_build)(501, g, P.levels)
            hyps = [TS.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025))]
            r = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)[0]
            out.append((bytes(bytearray(bytes(r))[:8 * 14 + 8]), bytes(r.E), r.n_steps))
        finally:
            ctx.close()
    assert out[0] == out[1]
