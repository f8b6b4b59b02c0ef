"""SURVEY §8(b) "Threading": the reference drives the photometric tracker (per frame, Hybrid.cpp:103-106 -> trackWithDso) and the
photometric mapper (per keyframe, direct/Mapping.cpp:3-41 directMappingLoop) from two host threads.  include/cmlhip.h promises that
contexts are independent and re-entrant: here ONE tracker context and ONE BA context are driven concurrently from two host threads
(ctypes releases the GIL for the duration of every call) for a few hundred calls each, and every result must be bit-identical to the
same call sequence run serially — no shared mutable state between contexts (the only statics in libcml_amd/csrc are getenv caches and
__constant__ tables; tests/test_abi_cpu.py::test_no_shared_mutable_statics greps for exactly that)."""
import threading

import numpy as np
import pytest

from libcml_amd import abi, device, host, synth

pytestmark = pytest.mark.gpu

N_TRACK = 120        # tracker calls per run: each = 5 level evaluations + one device-resident optimize of 3 hypotheses
N_BA = 300           # resident Gauss-Newton iterations per run, read back every 25


class _Tracker:
    def __init__(self, W):
        self.W = W
        fx, fy, cx, cy = W.K
        self.L = min(W.levels + 1, 5)
        self.ref, self.new = W.N - 2, W.N - 1
        self.ctx = device.Ctx(max_frames=4)
        self.ctx.pyramid_build(1, W.gray[self.ref], self.L); self.ctx.pyramid_build(2, W.gray[self.new], self.L)
        pts = []
        for i in range(W.P):
            h = int(W.pts["host"][i]); x, y, idp = float(W.pts["x"][i]), float(W.pts["y"][i]), float(W.pts["idepth"][i])
            Rht = W.R_eval[self.ref] @ W.R_eval[h].T; tht = W.t_eval[self.ref] - Rht @ W.t_eval[h]
            q = Rht @ np.array([(x - cx) / fx, (y - cy) / fy, 1.0]) + tht * idp
            pts.append(((q[0] / q[2]) * fx + cx, (q[1] / q[2]) * fy + cy, idp / q[2], 1.0))
        self.nout = self.ctx.tracker_make_coarse_depth(1, self.L, np.array(pts))
        self.Rt = W.R_true[self.new] @ W.R_true[self.ref].T; self.tt = W.t_true[self.new] - self.Rt @ W.t_true[self.ref]
        self.prm = abi.default_tracker_params()
        a, b = W.aff_true[self.ref]
        self.ref_exp = [a, b, 1.0]; self.init_exp = [a, b, 1.0]

    def run(self, n, out):
        W = self.W; fx, fy, cx, cy = W.K
        so3 = synth.so3_exp
        for it in range(n):
            rec = []
            for lvl in range(self.L):
                d = float(1 << lvl)
                K = np.array([fx / d, fy / d, (cx + 0.5) / d - 0.5, (cy + 0.5) / d - 0.5])
                r, _ = self.ctx.tracker_eval(2, lvl, self.Rt, self.tt + 0.001 * (it % 7), K, np.array([1.0, 0.0]), 0.0, self.prm, 1)
                rec.append(bytes(r))
            hyps = [(so3(np.array([0.003 + 0.0002 * i + 1e-5 * (it % 5), -0.002, 0.001])) @ self.Rt, self.tt + np.array([0.02, -0.01 + 0.001 * i, 0.015])) for i in range(3)]
            res = self.ctx.tracker_optimize_batch(2, self.L, W.K, self.ref_exp, self.init_exp, self.prm, hyps)
            for q in res:
                b = bytearray(bytes(q))
                off = abi.TrackerOptResult.eval_us.offset                       # the two in-kernel clock readings are timings, not results
                b[off:off + 16] = bytes(16)
                rec.append(bytes(b))
            out.append(rec)

    def close(self):
        self.ctx.close()


class _Mapper:
    def __init__(self, W):
        self.W = W
        self.ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
        self.ba = host.window_to_host_ba(self.ctx, W, image_id_base=5000, levels=1)
        self.ba.set_param("iterations", 1)
        assert self.ba.run(), self.ba.last_error()
        self.ctx.refresh_window_size()
        assert self.ba.begin_resident(), self.ba.last_error()

    def run(self, n, out):
        for it in range(n):
            self.ctx.ba_iteration_async(1e-5)
            if it % 25 == 24:
                self.ctx.sync()
                st = self.ctx.ba_states()
                out.append((st["state"].tobytes(), st["energy"].tobytes(), self.ctx.ba_get_idepth().tobytes(), self.ctx.ba_jpjdf().tobytes()))
        self.ctx.sync()

    def close(self):
        self.ba.close(); self.ctx.close()


def _fresh(W):
    return _Tracker(W), _Mapper(W)


def test_tracker_and_mapper_contexts_from_two_threads_match_serial_runs():
    W = synth.make_window("medium")
    # serial reference: the tracker's calls, then the mapper's, one thread
    trk, mp = _fresh(W)
    try:
        t_ser, m_ser = [], []
        trk.run(N_TRACK, t_ser); mp.run(N_BA, m_ser)
    finally:
        trk.close(); mp.close()
    # concurrent: the same call sequences on fresh contexts, one host thread each
    trk, mp = _fresh(W)
    try:
        t_par, m_par, errs = [], [], []

        def guard(fn, n, out):
            try:
                fn(n, out)
            except Exception as e:          # noqa: BLE001 — reported below, from the main thread
                errs.append(repr(e))

        ta = threading.Thread(target=guard, args=(trk.run, N_TRACK, t_par))
        tb = threading.Thread(target=guard, args=(mp.run, N_BA, m_par))
        ta.start(); tb.start(); ta.join(); tb.join()
        assert not errs, errs
    finally:
        trk.close(); mp.close()
    assert len(t_par) == len(t_ser) == N_TRACK and len(m_par) == len(m_ser) == N_BA // 25
    for k, (a, b) in enumerate(zip(t_ser, t_par)):
        assert a == b, "tracker call %d differs between the serial and the two-thread run" % k
    for k, (a, b) in enumerate(zip(m_ser, m_par)):
        assert a == b, "BA readback %d differs between the serial and the two-thread run" % k


def test_two_ba_contexts_from_two_threads():
    """two mappers (two sequence shards on one GPU, one host thread each): same bits as their serial runs"""
    Ws = [synth.make_window("small", shard=k) for k in range(2)]
    ser = []
    for W in Ws:
        m = _Mapper(W)
        try:
            o = []; m.run(150, o); ser.append(o)
        finally:
            m.close()
    ms = [_Mapper(W) for W in Ws]
    try:
        par = [[], []]
        th = [threading.Thread(target=ms[k].run, args=(150, par[k])) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        for m in ms:
            m.close()
    assert par == ser
