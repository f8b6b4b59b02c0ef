"""SURVEY §8 f3 (initializer part) on the device: DSOInitializer::calcResAndGS through the C ABI against the oracle.
Bar: every per-point output (isGood, isGood_new, energy_new, maxstep, lastHessian_new, the JbBuffer row) BIT-EXACT — the
8 residuals of a point are summed in pattern order in fp32 on both sides, fp contraction off; the 9x9 system, its Schur
complement and the energy are sums over the points in a different order (matrix cores vs the tiered SSE accumulators):
relative 2e-5 of the matrix scale."""
import numpy as np
import pytest

from libcml_amd import abi, device
from tests import initializer_setup as IS

pytestmark = pytest.mark.gpu

POINT_FIELDS = ("is_good", "is_good_new", "energy_new", "maxstep", "last_hessian_new", "jb")


def _same(a, b, name):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        bad = a.view(np.uint32) != b.view(np.uint32)
        bad &= ~(np.isnan(a) & np.isnan(b))
        assert not bad.any(), (name, int(bad.sum()), a[bad][:4], b[bad][:4])
    else:
        assert np.array_equal(a, b), name


def _close(a, b, name, rel=2e-5):
    scale = max(float(np.abs(b).max()), 1e-30)
    assert float(np.abs(a - b).max()) <= rel * scale, (name, float(np.abs(a - b).max()), scale)


def _run(level, trans_scale, step, texel=abi.TEXEL_F32):
    W, g0, g1, R, t, ratio, tlog = IS.scene(level=level, trans_scale=trans_scale)
    pts = IS.make_points(g0, step=step)
    prm = IS.make_params(W.K, level, R, t, ratio, tlog)
    ctx = device.Ctx(max_frames=2, texel_format=texel)
    try:
        ctx.pyramid_put(77, level, g1)
        d = pts.copy()
        Hd, bd, Hscd, bscd, resd = ctx.initializer_calc_res_and_gs(77, level, prm, d)
    finally:
        ctx.close()
    return pts, prm, g1, d, (Hd, bd, Hscd, bscd, resd)


@pytest.mark.parametrize("level,trans_scale,step", [(0, 1.0, 5), (1, 1.0, 3), (2, 1.0, 2), (1, 0.0, 3), (1, 1e-3, 3)])
def test_calc_res_and_gs_matches_oracle(level, trans_scale, step):
    """trans_scale 1: alphaEnergy > alphaK * npts (alphaOpt = 0, coupling branch); 0 / 1e-3: alphaOpt = alphaW branch."""
    pts, prm, g1, d, (Hd, bd, Hscd, bscd, resd) = _run(level, trans_scale, step)
    o, Ho, bo, Hsco, bsco, reso = IS.oracle_calc(g1, prm, pts)
    for f in POINT_FIELDS:
        _same(o[f], d[f], f)
    n = len(pts)
    assert 0 < int(o["is_good_new"].sum()) < n
    if trans_scale == 1.0:                                               # some pattern pixels leave the image: isGood is cleared for good
        assert int((o["is_good"] == 0).sum()) > int((pts["is_good"] == 0).sum())
    _close(Hd, Ho, "H"); _close(bd, bo, "b"); _close(Hscd, Hsco, "Hsc"); _close(bscd, bsco, "bsc")
    assert abs(resd[0] - reso[0]) <= 2e-5 * abs(reso[0]) and resd[1] == reso[1] and resd[2] == reso[2] == 2 * n
    branch_alpha_w = reso[1] < prm.alpha_k * n
    assert branch_alpha_w == (trans_scale < 0.5)


def test_fp16_texels_and_empty_list():
    pts, prm, g1, d, out = _run(1, 1.0, 3, texel=abi.TEXEL_F16)
    g16 = g1.astype(np.float16).astype(np.float32)
    o, Ho, bo, Hsco, bsco, reso = IS.oracle_calc(g16, prm, pts)
    for f in POINT_FIELDS:
        _same(o[f], d[f], f)
    _close(out[0], Ho, "H")
    W, g0, g1, R, t, ratio, tlog = IS.scene(level=1)
    prm = IS.make_params(W.K, 1, R, t, ratio, tlog)
    ctx = device.Ctx(max_frames=2)
    try:
        ctx.pyramid_put(5, 1, g1)
        e = np.zeros(0, abi.INIT_POINT_DTYPE)
        H, b, Hsc, bsc, res = ctx.initializer_calc_res_and_gs(5, 1, prm, e)
        assert not H.any() and not b.any() and not Hsc.any() and res[0] == 0 and res[2] == 0
        with pytest.raises(device.CmlHipError):
            ctx.initializer_calc_res_and_gs(6, 1, prm, e)            # unknown image
    finally:
        ctx.close()
