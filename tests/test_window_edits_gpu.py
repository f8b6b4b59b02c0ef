"""The window kept across keyframes (cmlhip_ba_window_*, include/cmlhip.h): a window arrived at by EDITS — append points / residuals, drop entries and
renumber (compact), removeFrame's renumbering of the frame ids (retire_frame), per-point refresh and resetOOB at the commit — is the window a fresh
cmlhip_ba_upload_window of the same lists builds: same sizes, same index maps (pair_of, CSR by point / by pair: bit-exact bookkeeping),
same thresholds / b0, and a residual pass over it returns the same bits (states, energies, the 74-float Jacobian records, centre projections) — every
derived device table is read by that pass.  Reference: BA::addPoints / addNewFrame / removePoint / removeFrame (BA.cpp:382-462, DSOContext.h:94-111,154-174)."""
import ctypes as C

import numpy as np
import pytest

from libcml_amd import abi, device
from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu
_P = C.POINTER


def _fresh(I):
    return D.make_ctx(I)


def _snapshot(ctx, I):
    """everything observable about the uploaded window + one residual pass over it"""
    maps = ctx.ba_index_maps()
    rd = ctx.ba_linearize()
    st = ctx.ba_states()
    out = {"size": ctx.refresh_window_size(), "lin": (np.float64(rd.energy).view(np.uint64), rd.n_in, rd.n_oob, rd.n_outlier, np.float32(rd.new_frame_energy_th).view(np.uint32)),
           "rj": ctx.ba_rj(0).view(np.uint32), "center": ctx.ba_center().view(np.uint32), "idepth": ctx.ba_get_idepth().view(np.uint64)}
    for k, v in maps.items():
        out["map_" + k] = np.asarray(v)
    for k in ("state", "new_state", "good"):
        out["st_" + k] = np.asarray(st[k])
    for k in ("energy", "new_energy", "new_energy_wo"):
        out["st_" + k] = np.asarray(st[k]).view(np.uint32)
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


def _edited(I, rng, extra_points=40, retire_extra_frame=True):
    """the same final lists as I's, reached through edits: start from a window with (a) a random subset of the final points missing, (b) junk points /
    residuals that get dropped, (c) one extra frame in the middle that is retired, (d) stale per-point values and residual states"""
    N, P, R = I.N, I.P, I.R
    ctx = device.Ctx(max_frames=N + 1, max_points=P + extra_points + 8, max_residuals=R + 8 * extra_points + 64)
    for k in range(N):
        ctx.pyramid_put(int(I.frames_dev["image_id"][k]), 0, I.grads[k][0])
    ctx.ba_set_params(I.prm)
    L, h = ctx.L, ctx.h
    ck = ctx.ck
    f_extra = N // 2 if retire_extra_frame else None          # ids >= f_extra are shifted up by one in the starting window

    def up(fid):
        return fid + 1 if (f_extra is not None and fid >= f_extra) else fid
    late = rng.random(P) < 0.3                                # points (and their residuals) that arrive with the last edit
    first_pts = np.flatnonzero(~late)
    # --- starting window: early points interleaved with junk points hosted by the extra frame / by ordinary frames
    pts0, tag0 = [], []                                       # tag: final point index or -1 (junk)
    for p in first_pts:
        if rng.random() < extra_points / max(len(first_pts), 1):
            j = I.points[rng.integers(P)].copy(); j["host"] = f_extra if (f_extra is not None and rng.random() < 0.5) else up(int(j["host"]))
            pts0.append(j); tag0.append(-1)
        q = I.points[p].copy(); q["host"] = up(int(q["host"])); q["idepth"] *= 1.5; q["idepth_zero"] = 0.25; q["prior"] = 7.0      # stale dynamic values
        pts0.append(q); tag0.append(int(p))
    pts0 = np.array(pts0, dtype=I.points.dtype); tag0 = np.array(tag0)
    slot_of = {int(t): i for i, t in enumerate(tag0) if t >= 0}
    res0, rtag0 = [], []
    for r in range(R):
        p = int(I.residuals["point"][r])
        if late[p]:
            continue
        if rng.random() < 0.02:                               # junk residual: targets the extra frame (or is simply dropped)
            jr = np.zeros((), I.residuals.dtype); jr["point"] = slot_of[p]; jr["target"] = f_extra if f_extra is not None else up(int(I.residuals["target"][r])); jr["state"] = 1
            if f_extra is None or int(pts0[slot_of[p]]["host"]) != f_extra:
                res0.append(jr); rtag0.append(-1)
        q = np.zeros((), I.residuals.dtype); q["point"] = slot_of[p]; q["target"] = up(int(I.residuals["target"][r])); q["state"] = 2; q["is_linearized"] = 0
        res0.append(q); rtag0.append(r)
    for i, t in enumerate(tag0):                              # junk points get a residual each
        if t < 0:
            jr = np.zeros((), I.residuals.dtype); jr["point"] = i; jr["target"] = (int(pts0[i]["host"]) + 1) % (N + (1 if f_extra is not None else 0)); res0.append(jr); rtag0.append(-1)
    res0 = np.array(res0, dtype=I.residuals.dtype); rtag0 = np.array(rtag0)
    ck(L.cmlhip_ba_window_reset(h))
    ck(L.cmlhip_ba_window_append_points(h, len(pts0), pts0.ctypes.data_as(C.c_void_p)))
    ck(L.cmlhip_ba_window_append_residuals(h, len(res0), res0.ctypes.data_as(C.c_void_p)))
    # --- removeFrame of the extra frame: renumbering, then drop what named it + the junk
    pa = (tag0 >= 0).astype(np.uint8); ra = (rtag0 >= 0).astype(np.uint8)
    if f_extra is not None:
        ck(L.cmlhip_ba_window_retire_frame(h, f_extra))
    ck(L.cmlhip_ba_window_compact(h, len(pa), pa.ctypes.data_as(_P(C.c_ubyte)), len(ra), ra.ctypes.data_as(_P(C.c_ubyte))))
    # --- the late points and residuals arrive (BA::addPoints): appended in final-list order
    kept_pts = list(tag0[tag0 >= 0]); kept_res = list(rtag0[rtag0 >= 0])
    new_pts = np.flatnonzero(late)
    final_slot = {int(t): i for i, t in enumerate(kept_pts)}
    for i, p in enumerate(new_pts):
        final_slot[int(p)] = len(kept_pts) + i
    lp = I.points[new_pts].copy()
    ck(L.cmlhip_ba_window_append_points(h, len(lp), lp.ctypes.data_as(C.c_void_p)))
    new_res = np.array([r for r in range(R) if late[int(I.residuals["point"][r])]], dtype=int)
    lr = I.residuals[new_res].copy()
    lr["point"] = [final_slot[int(p)] for p in I.residuals["point"][new_res]]
    ck(L.cmlhip_ba_window_append_residuals(h, len(lr), lr.ctypes.data_as(C.c_void_p)))
    order_p = np.array(kept_pts + [int(p) for p in new_pts]); order_r = np.array(kept_res + [int(r) for r in new_res])
    wp, wr = C.c_int(), C.c_int()
    ck(L.cmlhip_ba_window_counts(h, C.byref(wp), C.byref(wr)))
    assert (wp.value, wr.value) == (P, R)
    # --- commit: refresh of the per-point values; states as the final lists have them (residuals not IN are handed over as the exceptions)
    idp = np.ascontiguousarray(I.points["idepth"][order_p], np.float64); idz = np.ascontiguousarray(I.points["idepth_zero"][order_p], np.float32)
    pri = np.ascontiguousarray(I.points["prior"][order_p], np.float32)
    exc = np.flatnonzero((I.residuals["state"][order_r] != 0) | (I.residuals["is_linearized"][order_r] != 0)).astype(np.int32)
    exs = np.ascontiguousarray(I.residuals["state"][order_r][exc], np.int32)
    ck(L.cmlhip_ba_window_commit(h, N, I.frames_dev.ctypes.data_as(C.c_void_p), idp.ctypes.data_as(_P(C.c_double)), idz.ctypes.data_as(_P(C.c_float)),
                                 pri.ctypes.data_as(_P(C.c_float)), 1, len(exc), exc.ctypes.data_as(_P(C.c_int)), exs.ctypes.data_as(_P(C.c_int))))
    ctx.N, ctx.P, ctx.R = N, P, R
    ctx.ba_set_pairs(I.pairs)
    return ctx, order_p, order_r


def _permuted_inputs(I, order_p, order_r):
    """I with its point / residual lists in the order the edited window ended up with (a fresh upload of THOSE lists is the comparison)"""
    import copy
    J = copy.copy(I)
    inv = np.empty(I.P, int); inv[order_p] = np.arange(I.P)
    J.points = np.ascontiguousarray(I.points[order_p])
    rs = I.residuals[order_r].copy(); rs["point"] = inv[rs["point"]]
    J.residuals = np.ascontiguousarray(rs)
    return J


@pytest.mark.parametrize("config,seed,retire", [("small", 1, True), ("small", 2, False), ("medium", 3, True), ("tiny", 4, True)])
def test_edited_window_equals_fresh_upload(config, seed, retire):
    I = S.make_inputs(config, seed=seed)
    I.residuals["state"][:] = 0            # the commit's resetOOB (BA.cpp:766-779) hands every residual that is not LINEARIZED over as IN: the fresh upload gets the same lists
    rng = np.random.default_rng(100 + seed)
    ctx_e, order_p, order_r = _edited(I, rng, retire_extra_frame=retire)
    J = _permuted_inputs(I, order_p, order_r)
    # LINEARIZED residuals would need their res_toZero (set by a relinearisation pass, not by an upload): the lists of these windows have none
    assert not np.any(J.residuals["is_linearized"] != 0)
    ctx_f = _fresh(J)
    try:
        _same(_snapshot(ctx_e, J), _snapshot(ctx_f, J))
        # ... and a second keyframe's worth of edits on top of the committed window: drop every 7th point with its residuals, commit again
        L, h = ctx_e.L, ctx_e.h
        pa = np.ones(J.P, np.uint8); pa[::7] = 0
        ra = pa[J.residuals["point"]].copy()
        ctx_e.ck(L.cmlhip_ba_window_compact(h, J.P, pa.ctypes.data_as(_P(C.c_ubyte)), J.R, ra.ctypes.data_as(_P(C.c_ubyte))))
        keep_p = np.flatnonzero(pa); keep_r = np.flatnonzero(ra)
        K = _permuted_inputs(J, np.concatenate([keep_p, np.flatnonzero(pa == 0)]), np.arange(J.R))
        import copy
        K = copy.copy(J)
        inv = -np.ones(J.P, int); inv[keep_p] = np.arange(len(keep_p))
        K.points = np.ascontiguousarray(J.points[keep_p]); rs = J.residuals[keep_r].copy(); rs["point"] = inv[rs["point"]]; K.residuals = np.ascontiguousarray(rs)
        K.P, K.R = len(keep_p), len(keep_r)
        st = np.ascontiguousarray(K.residuals["state"], np.int32); exc = np.flatnonzero(st != 0).astype(np.int32); exs = np.ascontiguousarray(st[exc])
        ctx_e.ck(L.cmlhip_ba_window_commit(h, K.N, K.frames_dev.ctypes.data_as(C.c_void_p), None, None, None, 1, len(exc), exc.ctypes.data_as(_P(C.c_int)), exs.ctypes.data_as(_P(C.c_int))))
        ctx_e.N, ctx_e.P, ctx_e.R = K.N, K.P, K.R
        ctx_e.ba_set_pairs(K.pairs)
        ctx_g = _fresh(K)
        try:
            _same(_snapshot(ctx_e, K), _snapshot(ctx_g, K))
        finally:
            ctx_g.close()
    finally:
        ctx_e.close(); ctx_f.close()


def test_window_edit_errors_are_reported_not_ignored():
    I = S.make_inputs("tiny")
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
    try:
        L, h = ctx.L, ctx.h
        for k in range(I.N):
            ctx.pyramid_put(int(I.frames_dev["image_id"][k]), 0, I.grads[k][0])
        ctx.ba_set_params(I.prm)
        assert L.cmlhip_ba_window_reset(h) == 0
        assert L.cmlhip_ba_window_append_points(h, I.P, I.points.ctypes.data_as(C.c_void_p)) == 0
        bad = I.residuals[:4].copy(); bad["point"][2] = I.P + 5
        assert L.cmlhip_ba_window_append_residuals(h, 4, bad.ctypes.data_as(C.c_void_p)) == abi.ERR_INVALID          # residual names a point that is not there
        assert L.cmlhip_ba_window_append_points(h, 1, I.points.ctypes.data_as(C.c_void_p)) == abi.ERR_INVALID           # beyond max_points given at create
        assert L.cmlhip_ba_window_append_residuals(h, I.R, I.residuals.ctypes.data_as(C.c_void_p)) == 0
        fl = np.ones(I.P - 1, np.uint8); rl = np.ones(I.R, np.uint8)
        assert L.cmlhip_ba_window_compact(h, I.P - 1, fl.ctypes.data_as(_P(C.c_ubyte)), I.R, rl.ctypes.data_as(_P(C.c_ubyte))) == abi.ERR_STATE   # lists differ in length
        fl = np.ones(I.P, np.uint8); fl[int(I.residuals["point"][0])] = 0
        assert L.cmlhip_ba_window_compact(h, I.P, fl.ctypes.data_as(_P(C.c_ubyte)), I.R, rl.ctypes.data_as(_P(C.c_ubyte))) == abi.ERR_INVALID     # survivor names a dropped point
    finally:
        ctx.close()


def test_a_refused_commit_inside_an_upload_scope_leaves_nothing_behind():
    """cmlhip_ba_window_commit failing half-way inside an open upload scope (here: a frame whose image is not in the pyramid cache): the scope's block and the
    kernels waiting for it are dropped, the context holds NO window (CMLHIP_ERR_STATE from what needs one), and the same lists commit fine afterwards."""
    I = S.make_inputs("tiny")
    I.residuals["state"][:] = 0
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
    try:
        L, h = ctx.L, ctx.h
        for k in range(I.N):
            ctx.pyramid_put(int(I.frames_dev["image_id"][k]), 0, I.grads[k][0])
        ctx.ba_set_params(I.prm)
        assert L.cmlhip_ba_window_reset(h) == 0
        assert L.cmlhip_ba_window_append_points(h, I.P, I.points.ctypes.data_as(C.c_void_p)) == 0
        assert L.cmlhip_ba_window_append_residuals(h, I.R, I.residuals.ctypes.data_as(C.c_void_p)) == 0
        bad = I.frames_dev.copy(); bad["image_id"][I.N - 1] = 987654
        assert L.cmlhip_upload_scope_begin(h) == 0
        assert L.cmlhip_ba_window_commit(h, I.N, bad.ctypes.data_as(C.c_void_p), None, None, None, 0, 0, None, None) == abi.ERR_NOT_FOUND
        assert L.cmlhip_upload_scope_end(h) == 0                       # nothing left to flush
        lr = abi.BALinResult()
        assert L.cmlhip_ba_linearize(h, C.byref(lr)) == abi.ERR_STATE  # no window
        assert L.cmlhip_upload_scope_begin(h) == 0
        assert L.cmlhip_ba_window_commit(h, I.N, I.frames_dev.ctypes.data_as(C.c_void_p), None, None, None, 0, 0, None, None) == 0
        ctx.N, ctx.P, ctx.R = I.N, I.P, I.R
        ctx.ba_set_pairs(I.pairs)                                       # (staged into the same scope)
        assert L.cmlhip_upload_scope_end(h) == 0
        ref = _fresh(I)
        try:
            _same(_snapshot(ctx, I), _snapshot(ref, I))
        finally:
            ref.close()
    finally:
        ctx.close()
