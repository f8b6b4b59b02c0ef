"""DSOBundleAdjustment::run (BA.cpp:744-910) composed from ORACLE primitives — the checker for the host mirror's run()."""
import ctypes as C

import numpy as np

from tests import ba_setup as S
from tests import oracle_lib as O


def oracle_run(I, iterations=4, fixed_lambda=1e-5, th_opt=1.2, force_accept=True, fix_lambda=True):
    ob = S.OracleBA(I)
    N, P = I.N, I.P
    n = 8 * N + 4
    lib = O.lib()
    log = dict(energy=[], x=[], accepted=[], lam=[])
    r = ob.linearize()
    last_e = r.energy
    last_l = 0.0 if force_accept else ob.l_energy()[0]        # calcLEnergy (+ calcMEnergy = 0: no prior), BA.cpp:783-784
    ob.apply(1)
    log["energy"].append(r.energy)
    lam = fixed_lambda
    I.frames[N - 1].frame_energy_th = 0  # placeholder (oracle window keeps its own thresholds)
    for it in range(iterations):
        backup = [np.array(I.frames[k].state[:]) for k in range(N)]
        lib.orc_ba_backup_points(ob.w)
        HA, bA, HL, bL, Hsc, bsc = ob.accumulate()
        x, rc = ob.solve(fixed_lambda if fix_lambda else lam, HA, bA, HL, bL, Hsc, bsc)
        if it >= 2:
            ns = np.zeros(7 * n)
            lib.orc_ba_nullspaces(I.frames, N, C.byref(I.scales), O.ptr(ns, C.c_double))
            x = O.orthogonalize(x, ns.reshape(7, n), 1e-5)
        log["x"].append(x.copy())
        step, rc = ob.backsub(x)
        assert rc == 0
        sums = dict(A=np.float32(0), B=np.float32(0), T=np.float32(0), R=np.float32(0))
        for k in range(N):
            st = backup[k].copy(); stp = np.zeros(10); stp[:8] = -x[4 + 8 * k:12 + 8 * k]
            st += stp
            lib.orc_frame_set_state(C.byref(I.frames[k]), O.ptr(st, C.c_double), C.byref(I.scales))
            sums["A"] += np.float32(stp[6] ** 2); sums["B"] += np.float32(stp[7] ** 2)
            sums["T"] += np.float32((stp[:3] ** 2).sum()); sums["R"] += np.float32((stp[3:6] ** 2).sum())
        ps = np.zeros(3, np.float32)
        lib.orc_ba_step_points(ob.w, O.ptr(ps, C.c_float))
        sumNID = ps[1] / ps[2]
        canbreak = (np.sqrt(sums["A"] / N) < 0.0005 * th_opt and np.sqrt(sums["B"] / N) < 0.00005 * th_opt and
                    np.sqrt(sums["R"] / N) < 0.00005 * th_opt and np.sqrt(sums["T"] / N) * sumNID < 0.00005 * th_opt)
        I.pairs = S.frame_pairs(I.frames, N); ob.set_pairs(I.pairs)
        I.adH, I.adT, I.adHTd, I.prior, I.dprior = S.adjoints_and_delta(I.frames, N, I.scales)
        r = ob.linearize()
        new_l = 0.0 if force_accept else ob.l_energy()[0]
        if r.energy + new_l < last_e + last_l or force_accept:                 # BA.cpp:830-853
            ob.apply(1)
            log["energy"].append(r.energy); log["accepted"].append(True)
            last_e, last_l = r.energy, new_l
            lam *= 0.25
        else:                                                                   # loadSateBackup, BA.cpp:871-875
            for k in range(N):
                lib.orc_frame_set_state(C.byref(I.frames[k]), O.ptr(backup[k].copy(), C.c_double), C.byref(I.scales))
            lib.orc_ba_restore_points(ob.w)
            I.pairs = S.frame_pairs(I.frames, N); ob.set_pairs(I.pairs)
            I.adH, I.adT, I.adHTd, I.prior, I.dprior = S.adjoints_and_delta(I.frames, N, I.scales)
            r = ob.linearize()
            last_e = r.energy; last_l = ob.l_energy()[0]
            log["accepted"].append(False)
            lam *= 1e2
        log["lam"].append(lam)
        if canbreak and it >= 1:
            break
    # re-anchor the newest frame (BA.cpp:885-894): setEvalPT(PRE_worldToCam, [0.., a, b])
    fb = I.frames[N - 1]
    nz = np.zeros(10); nz[6] = fb.state[6]; nz[7] = fb.state[7]
    fb.w2c_eval = fb.PRE_w2c
    lib.orc_frame_set_state(C.byref(fb), O.ptr(nz, C.c_double), C.byref(I.scales))
    lib.orc_frame_set_state_zero(C.byref(fb), O.ptr(nz, C.c_double), C.byref(I.scales))
    ob.w.contents.b0[N - 1] = float(np.float32(fb.state_zero[7] * np.float32(I.scales.b)))      # getB0 follows state_zero (DSOFrame.h:197-199)
    I.pairs = S.frame_pairs(I.frames, N); ob.set_pairs(I.pairs)
    I.adH, I.adT, I.adHTd, I.prior, I.dprior = S.adjoints_and_delta(I.frames, N, I.scales)
    r = ob.linearize(); ob.apply(1)        # linearizeAll(true)
    log["energy"].append(r.energy)
    st = ob.states()
    good = st["good"] == 1
    # residual / point bookkeeping of BA.cpp:1568-1642
    nres = np.zeros(P, np.int32)
    np.add.at(nres, I.residuals["point"][good], 1)
    outliers = np.nonzero(nres == 0)[0]
    poses = []
    for k in range(N):
        Rm, t = O.se3_matrix(I.frames[k].PRE_w2c)
        poses.append((Rm, t, I.frames[k].state_scaled[6], I.frames[k].state_scaled[7]))
    idepth = np.array([ob.w.contents.points[i].idepth for i in range(P)])
    return dict(poses=poses, idepth=idepth, good=good, outliers=outliers, log=log, th=float(ob.w.contents.frame_energy_th[N - 1]), ob=ob)


def oracle_run_release_rounding(I, **kw):
    """The same procedure on the SAME oracle sources built with the reference's Release flags (-O3 -march=native, fused multiply-adds at the
    compiler's discretion: oracle/Makefile `contract`, built on the box it runs on): a second correct rounding of the path.  The distance
    between the two oracle runs is what this window does to rounding noise over the iterations — the yardstick a device-vs-oracle distance
    is held against (tests/test_host_mirror_gpu.py)."""
    import os
    import subprocess
    subprocess.check_call(["make", "-C", O.ORACLE_DIR, "contract"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(O.ORACLE_DIR, "libcml_oracle_contract.so"))
    L.orc_ba_create.restype = C.POINTER(O.OrcBAWindow)
    L.orc_ba_linearize_one.restype = C.c_double
    L.orc_ba_calc_l_energy.restype = C.c_double
    L.orc_ba_calc_m_energy.restype = C.c_double
    keep = O._lib
    O._lib = L
    try:
        return oracle_run(I, **kw)
    finally:
        O._lib = keep
