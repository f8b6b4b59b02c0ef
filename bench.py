#!/usr/bin/env python
"""bench.py — point-residuals/s (+ Schur-reduce+solve ms) of the sliding-window photometric BA hot path on MI355X.

Workload (BASELINE.json configs[1]): 8-keyframe window, 2000 active points (R = 14 000 point-residuals), 1241x376
KITTI-shape level-0 gradient images, fp32.  One STEP = one Gauss-Newton iteration of DSOBundleAdjustment::run on the
window, entirely on the device, enqueued back to back on the context stream:
    backup -> accumulate (13x13 pair blocks, fp64 stitch) -> point Schur rows + fp64 SYRK -> dense 8N solve ->
    back-substitution -> point step -> residual/Jacobian evaluation of all R residuals -> energy threshold -> apply.
value = (point-residuals evaluated by all ranks) / (max-over-ranks wall time), inputs resident in HBM.
N > 1: one process per GPU (torchrun), one independent window (sequence shard) per rank, RCCL used only as the
start/stop barrier — weak scaling, no data-path collective.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 8 px x 4 taps x 12 B (fp32 texel: I, dI/dx, dI/dy) + colours 32 + weights 32 + (x,y,idepth) 12 + ids 8 = 468 B;
# with fp16 pyramids (config E) the taps are 6 B each: 192 + 84 = 276 B
READ_BYTES_PER_RESIDUAL = {"fp32": 468, "fp16": 276}
HBM_PEAK_GBS = 8000.0                  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _baseline_metric():
    """BASELINE.json's metric string, verbatim (it travels with the repo)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "point-residuals/sec + Schur-reduce+solve ms, 8 KF \u00d7 2000 pts window"


def _one_socket_cores():
    """Logical CPU ids of the physical cores (one hardware thread each) of socket 0 that this process may run on."""
    allowed = os.sched_getaffinity(0)
    seen, cpus = set(), []
    cur = {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip() and cur:
                cpu = int(cur.get("processor", -1)); key = (cur.get("physical id", "0"), cur.get("core id", str(cpu)))
                if cpu in allowed and cur.get("physical id", "0") == "0" and key not in seen:
                    seen.add(key); cpus.append(cpu)
                cur = {}
    except Exception:
        pass
    return cpus or sorted(allowed)


def _time_oracle(L, O, S, config, seed, runs=20, warm=3, iters_per_run=5):
    """per-run times (after `warm` warm-ups) of one run = iters_per_run Gauss-Newton iterations of the window: (R, [linearize s], [rest s])"""
    I = S.make_inputs(config, seed=seed)
    ob = S.OracleBA(I)
    ob.linearize(); ob.apply(1)
    t_lin, t_rest = [], []
    for k in range(warm + runs):
        a = b = 0.0
        for _ in range(iters_per_run):
            t0 = time.perf_counter()
            L.orc_ba_backup_points(ob.w)
            H = ob.accumulate()
            x, rc = ob.solve(1e-5, *H)
            ob.backsub(x)
            L.orc_ba_step_points(ob.w, None)
            t1 = time.perf_counter()
            ob.linearize(); ob.apply(1)
            t2 = time.perf_counter()
            b += t1 - t0; a += t2 - t1
        if k >= warm:
            t_lin.append(a / iters_per_run); t_rest.append(b / iters_per_run)
    return I.R, t_lin, t_rest


def _cpu_worker(config, seed, mode):
    """Runs in a FRESH interpreter (no torch: its libgomp would have read the OpenMP environment before we could set it).
    mode "single": oracle C port, -O3 -march=native, one thread.  mode "omp": the same sources with -fopenmp — residual loop, pair /
    point accumulation, point Schur and back-substitution spread over the threads (timing build only; the checker build is serial)."""
    import statistics
    from tests import ba_setup as S
    from tests import oracle_lib as O
    target, soname = ("fast", "libcml_oracle_fast.so") if mode == "single" else ("fast_omp", "libcml_oracle_omp.so")
    build = "-O3 -march=native" + ("" if mode == "single" else " -fopenmp")
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), target], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(ROOT, "oracle", soname))
        L.orc_ba_create.restype = C.POINTER(O.OrcBAWindow)
        L.orc_ba_linearize_one.restype = C.c_double
        O._lib = L
    except Exception:
        if mode != "single":
            raise
        L = O.lib(); build = "-O2 (portable checker build)"
    R, t_lin, t_rest = _time_oracle(L, O, S, config, seed)
    tot = [a + b for a, b in zip(t_lin, t_rest)]
    med = statistics.median(tot)
    print(json.dumps({"R": R, "value": R / med, "value_minmax": [R / max(tot), R / min(tot)],
                      "linearize_residuals_per_s": R / statistics.median(t_lin), "linearize_residuals_per_s_minmax": [R / max(t_lin), R / min(t_lin)],
                      "schur_solve_ms": 1e3 * statistics.median(t_rest), "schur_solve_ms_minmax": [1e3 * min(t_rest), 1e3 * max(t_rest)],
                      "build": build, "threads": int(os.environ.get("OMP_NUM_THREADS", "1"))}))


def _run_cpu_worker(config, seed, mode, env_extra):
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "GOMP_CPU_AFFINITY", "OMP_PROC_BIND", "OMP_PLACES", "OMP_WAIT_POLICY"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", config, str(seed), mode], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline worker failed: " + r.stderr[-500:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def cpu_baseline(config, seed):
    """The oracle (plain-C port of the reference CPU path, "kind": "port") timed on this box's host cores, SURVEY §8(d): median of 20
    runs (3 warm-ups) of 5 Gauss-Newton iterations of the same window, (i) single thread — the reference's BA loop is serial
    (BA.cpp:1551-1565) — and (ii) OpenMP over the physical cores of one socket (threads bound one per core, passive waiting, 64 / 32 / 16
    threads tried, best kept).  `value` = the better of (i) and (ii): the single-socket figure the >= 30x target is judged on."""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    one = _run_cpu_worker(config, seed, "single", {"OMP_NUM_THREADS": "1"})
    R = one["R"]
    out = {"value": one["value"], "value_minmax": one.get("value_minmax"), "unit": "point-residuals/s", "cores": 1, "kind": "port",
           "sample": "median of 20 runs (3 warm-ups) of 5 full Gauss-Newton iterations of the same window (R=%d) by the oracle C port" % R,
           "single_thread": one, "host_cpu": model, "host_logical_cpus": os.cpu_count()}
    try:
        cores = _one_socket_cores()
        best = None
        tried = []
        for nt in sorted({len(cores), max(len(cores) // 2, 1), max(len(cores) // 4, 1)}, reverse=True):
            use = cores[:nt]
            r = _run_cpu_worker(config, seed, "omp", {"OMP_NUM_THREADS": str(nt), "GOMP_CPU_AFFINITY": " ".join(str(c) for c in use),
                                                      "OMP_PROC_BIND": "true", "OMP_WAIT_POLICY": "passive"})
            tried.append({"threads": nt, "value": r["value"], "value_minmax": r["value_minmax"], "schur_solve_ms": r["schur_solve_ms"],
                          "linearize_residuals_per_s": r["linearize_residuals_per_s"]})
            if best is None or r["value"] > best["value"]:
                best = r
        out["all_cores_one_socket"] = dict(best, cores_available=len(cores), tried=tried,
                                           sample="same protocol, -fopenmp build: residual loop, pair / point accumulation, point Schur and "
                                                  "back-substitution over the threads (one per physical core of socket 0, bound, passive waiting), the N^3 "
                                                  "sandwiches of stitchDoubleSC and the per-thread accumulator reductions over the threads too; top stitch + "
                                                  "dense 8N solve serial (0.15 ms of the iteration)")
        if best["value"] > out["value"]:
            out["value"] = best["value"]; out["cores"] = best["threads"]; out["value_minmax"] = best.get("value_minmax")
    except Exception as e:
        out["all_cores_one_socket"] = {"value": None, "sample": "failed: %r" % (e,)}
    out["sample"] += "; value = max(single thread, OpenMP on one socket); threads used by that figure = cores"
    return out


def multi_window_bench(device_id, seed, config, steps, half, s_list=(2, 4, 8, 16, 32)):
    """Throughput mode on ONE GPU (north_star: independent keyframe windows / sequence shards): S independent windows of the benchmark
    shape stepped by cmlhip_ba_iteration_batch — one launch per kernel family for all S windows, five launches per round whatever S is.
    value = point-residuals of all S windows per second; parity: after the timed rounds one more batched round whose residual pass is
    replayed on the oracle for the first and the last window (bit for bit, tests/resident_check.py).  The headline stays 1 window / GPU."""
    from libcml_amd import abi, device, host, synth
    smax = max(s_list)
    wins = []
    scenes = {}
    for k in range(smax):
        if k % 8 not in scenes:                                       # eight distinct scenes; windows k and k + 8 hold the same scene in their own contexts
            scenes[k % 8] = synth.make_window(config, seed=seed, shard=100 + k % 8)
        W = scenes[k % 8]
        ctx = device.Ctx(device_id=device_id, max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N,
                         texel_format=abi.TEXEL_F16 if half else abi.TEXEL_F32)
        ba = host.window_to_host_ba(ctx, W, image_id_base=1000 * (k + 1), levels=1)
        ba.set_param("iterations", 1)
        if not ba.run() or not ba.begin_resident():
            raise RuntimeError("multi-window set-up failed: " + ba.last_error())
        _, _, R = ctx.refresh_window_size()
        wins.append((W, ctx, ba, R))
    lam = 1e-5
    out = {"note": "S windows of the benchmark shape per GPU, one cmlhip_ba_iteration_batch per round (5 launches for all S windows)", "runs": []}
    for S in s_list:
        ctxs = [w[1] for w in wins[:S]]
        for _ in range(60):
            device.ba_iteration_batch(ctxs, lam)
        ctxs[0].sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            device.ba_iteration_batch(ctxs, lam)
        ctxs[0].sync()
        dt = time.perf_counter() - t0
        Rs = sum(w[3] for w in wins[:S])
        out["runs"].append({"S": S, "value": Rs * steps / dt, "unit": "point-residuals/s", "ms_per_round": 1e3 * dt / steps, "residuals_per_round": Rs, "rounds": steps})
    # the same windows as G groups on G streams (each group's five launches on the stream of its first context)
    out["streams"] = []
    for S in [s_ for s_ in s_list if s_ >= 4]:
        for G in (2, 4):                                              # G groups of S / G windows, each group's five launches on its own stream
            if S // G < 1 or S % G:
                continue
            groups = [[w[1] for w in wins[g * (S // G):(g + 1) * (S // G)]] for g in range(G)]
            for _ in range(60):
                for g in groups:
                    device.ba_iteration_batch(g, lam)
            for g in groups:
                g[0].sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                for g in groups:
                    device.ba_iteration_batch(g, lam)
            for g in groups:
                g[0].sync()
            dt = time.perf_counter() - t0
            Rs = sum(w[3] for w in wins[:S])
            out["streams"].append({"S": S, "groups": G, "value": Rs * steps / dt, "unit": "point-residuals/s", "ms_per_round": 1e3 * dt / steps, "rounds": steps})
    out["streams_note"] = ("the same S windows as G groups, one cmlhip_ba_iteration_batch per group on the group's own stream: the solve launch of one group — "
                           "S / G workgroups on a 256-CU chip — runs beside the residual / accumulate launches of the others")
    try:
        from tests import resident_check as RC
        ctxs = [w[1] for w in wins]
        pick = [0, smax - 1]
        replays = {k: RC.make_replay(wins[k][1], wins[k][2], wins[k][0]) for k in pick}
        ctxs[0].sync()
        pres = {k: wins[k][1].ba_states() for k in pick}
        device.ba_iteration_batch(ctxs, lam)
        ctxs[0].sync()
        reps = {k: RC.compare_pass(wins[k][1], replays[k], pres[k], with_records=False) for k in pick}
        for r in replays.values():
            r.close()
        out["parity_checked"] = True
        out["parity_ok"] = all(r["ok"] for r in reps.values())
        out["parity"] = {"windows": pick, "S": smax, "mismatches": {str(k): {a: b for a, b in r.items() if a.endswith("_mismatch")} for k, r in reps.items()}}
    except Exception as e:
        out["parity_checked"] = False; out["parity_error"] = repr(e)
    try:                                                              # the opt-in arithmetic mode on the same windows (after the exact figures and their gate)
        for w in wins:
            w[1].ba_set_arithmetic(True)
        rel = []
        for S, G in ((8, 1), (8, 4), (32, 1), (32, 2)):
            if S > smax:
                continue
            groups = [[w[1] for w in wins[g * (S // G):(g + 1) * (S // G)]] for g in range(G)]
            for _ in range(60):
                for g in groups:
                    device.ba_iteration_batch(g, lam)
            for g in groups:
                g[0].sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                for g in groups:
                    device.ba_iteration_batch(g, lam)
            for g in groups:
                g[0].sync()
            dt = time.perf_counter() - t0
            Rs = sum(w[3] for w in wins[:S])
            rel.append({"S": S, "groups": G, "value": Rs * steps / dt, "unit": "point-residuals/s", "ms_per_round": 1e3 * dt / steps})
        out["relaxed_arithmetic"] = {"mode": "CMLHIP_ARITH_RELAXED (opt-in; every figure above is CMLHIP_ARITH_EXACT)", "runs": rel}
    except Exception as e:
        out["relaxed_arithmetic"] = {"error": repr(e)}
    finally:
        for w in wins:
            try:
                w[1].ba_set_arithmetic(False)
            except Exception:
                pass
    best = max(out["runs"] + out.get("streams", []), key=lambda r: r["value"])
    out["S"], out["value"], out["ms_per_round"], out["groups"] = best["S"], best["value"], best["ms_per_round"], best.get("groups", 1)
    for W, ctx, ba, R in wins:
        ba.close(); ctx.close()
    return out


TRACKER_READ_BYTES_PER_POINT = 64      # 4 taps x 12 B + 16 B (u, v, idepth, colour)


def tracker_bench(device_id, seed, want_cpu):
    """The tracker half of north_star on the config-B shape (1241x376, 5 pyramid levels, the 2000 active points of the window splatted
    into the reference keyframe by makeCoarseDepthL0): k_tracker_eval per level (kernel duration from events attached to the dispatch,
    achieved GB/s on 64 B per reference point against the HBM roof), DSOTracker::optimize for one motion hypothesis (host-driven loop and
    device-resident) and for 50 (one launch), and the oracle's orc_tracker_optimize on the host cores beside it."""
    import statistics
    import numpy as np
    from libcml_amd import abi, device, host, synth
    W = synth.make_window("B", seed=seed)
    fx, fy, cx, cy = W.K
    ref, new, L = W.N - 2, W.N - 1, 5
    ctx = device.Ctx(device_id=device_id, max_frames=8)
    ctx.pyramid_build(1, W.gray[ref], L); ctx.pyramid_build(2, W.gray[new], L)
    pts = []
    for i in range(W.P):                                   # host part of makeCoarseDepthL0 (TR.cpp:521-540): the active points seen from the reference
        h = int(W.pts["host"][i]); x, y, idp = float(W.pts["x"][i]), float(W.pts["y"][i]), float(W.pts["idepth"][i])
        Rht = W.R_eval[ref] @ W.R_eval[h].T; tht = W.t_eval[ref] - Rht @ W.t_eval[h]
        q = Rht @ np.array([(x - cx) / fx, (y - cy) / fy, 1.0]) + tht * idp
        pts.append(((q[0] / q[2]) * fx + cx, (q[1] / q[2]) * fy + cy, idp / q[2], 1.0))
    nout = ctx.tracker_make_coarse_depth(1, L, np.array(pts))
    Rt = W.R_true[new] @ W.R_true[ref].T; tt = W.t_true[new] - Rt @ W.t_true[ref]
    prm = abi.default_tracker_params()
    a_r, b_r = W.aff_true[ref]
    ref_exp = [a_r, b_r, float(W.ab_exposure[ref])]; init_exp = [a_r, b_r, float(W.ab_exposure[new])]
    levels = []
    for lvl in range(L):
        d = float(1 << lvl)
        K = np.array([fx / d, fy / d, (cx + 0.5) / d - 0.5, (cy + 0.5) / d - 0.5])
        us = []
        for k in range(25):
            if k >= 5:
                ctx.profile_next_launch()
            r, _ = ctx.tracker_eval(2, lvl, Rt, tt, K, np.array([1.0, 0.0]), 0.0, prm, 1)
            if k >= 5:
                us.append(1e3 * ctx.elapsed_ms())
        t = statistics.median(us)
        gbs = nout[lvl] * TRACKER_READ_BYTES_PER_POINT / (t * 1e-6) / 1e9
        levels.append({"level": lvl, "points": int(nout[lvl]), "warped": int(r.numWarped), "kernel_us": t, "kernel_us_minmax": [min(us), max(us)],
                       "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
    out = {"shape": "config B: %dx%d, %d levels, %d active points -> reference lists %s" % (W.w, W.h, L, W.P, [int(x) for x in nout[:L]]),
           "bytes_per_point": TRACKER_READ_BYTES_PER_POINT, "eval": levels,
           "eval_note": "k_tracker_eval = computeResidual + computeHessian of one level in one launch; kernel_us = median of 20 of the dispatch's own duration "
                        "(events attached to the dispatch)"}
    so3 = synth.so3_exp

    def hyp(i):
        return so3(np.array([0.004 + 0.0002 * i, -0.003, 0.002])) @ Rt, tt + np.array([0.03, -0.02 + 0.001 * i, 0.025])

    trk = host.HostTracker(ctx); trk.set_calibration(*W.K)
    R0, t0 = hyp(0)
    ts = []
    for k in range(23):
        t_ = time.perf_counter(); trk.optimize(2, L, R0, t0, ref_exp, init_exp); ts.append(time.perf_counter() - t_)
    out["optimize_host_driven_ms"] = 1e3 * statistics.median(ts[3:])
    out["optimize_trials"] = int(len(trk.steps()[0]))
    for nh in (1, 50):
        hyps = [hyp(i) for i in range(nh)]
        ts, ks = [], []
        for k in range(23):
            ctx.profile_next_launch()
            t_ = time.perf_counter(); res = ctx.tracker_optimize_batch(2, L, W.K, ref_exp, init_exp, prm, hyps); ts.append(time.perf_counter() - t_)
            ks.append(ctx.elapsed_ms())
        out["optimize_device_resident_%d_hyp" % nh] = {"call_ms": 1e3 * statistics.median(ts[3:]), "kernel_ms": statistics.median(ks[3:]),
                                                      "trials_first": int(res[0].n_steps), "in_kernel_eval_us": float(res[0].eval_us),
                                                      "in_kernel_algebra_us": float(res[0].algebra_us)}
    if want_cpu:
        try:
            from tests import trk_opt_setup as TS
            P = TS.Problem()
            P.W, P.levels, P.prm, P.ref_exp, P.init_exp = W, L, prm, ref_exp, init_exp
            P.uvic = [np.ascontiguousarray(ctx.tracker_get_reference(l), np.float32) for l in range(L)]      # the device's lists and images are the oracle's input here
            P.imgs = [np.ascontiguousarray(ctx.pyramid_get(2, l), np.float32) for l in range(L)]
            P.Rt, P.tt = Rt, tt
            ts = []
            for k in range(23):
                t_ = time.perf_counter(); o = TS.oracle_optimize(P, R0, t0); ts.append(time.perf_counter() - t_)
            out["cpu_baseline"] = {"optimize_ms": 1e3 * statistics.median(ts[3:]), "optimize_ms_minmax": [1e3 * min(ts[3:]), 1e3 * max(ts[3:])], "cores": 1, "kind": "port",
                                   "trials": int(o["out"].n_steps), "sample": "median of 20 orc_tracker_optimize calls (oracle C port, checker build -O2, one thread) on the same problem"}
        except Exception as e:
            out["cpu_baseline"] = {"optimize_ms": None, "sample": "failed: %r" % (e,)}
    trk.close(); ctx.close()
    return out


def _dbg(*a):
    if os.environ.get("CML_BENCH_DEBUG"):
        print("[bench]", *a, file=sys.stderr, flush=True)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(n):
    """`python bench.py --gpus N` started WITHOUT a torchrun environment: launch the N ranks ourselves (one process per GPU, the same
    command line the driver uses for N > 1) and pass their output through.  Rank 0 prints the JSON line; n_gpus in it is the size of
    the process group that actually formed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env, cwd=ROOT)


def _launch_only(args):
    """--launch-only: form the process group, run the timing contract (barrier, sync, max over ranks) around an EMPTY step list and print
    the line skeleton.  Exists so that the N > 1 launch path of this file can be exercised where there is no GPU (gloo, tests/test_shard_cpu.py);
    it touches no device code and reports no throughput."""
    from libcml_amd import shard
    group = shard.Group()
    dt = shard.timed_region(group, lambda: None, lambda: None)
    ranks = group.sum(1.0)
    if group.rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": int(round(ranks)), "requested_gpus": args.gpus, "backend": getattr(group, "backend", None),
                          "steps": 0, "elapsed_s": dt}))
    group.close()


FP64_MATRIX_PEAK_TFLOPS = 78.6           # AMD's published MI355X FP64 matrix figure (the micro-architecture guide gives no fp64 row)


def solve_phases(ctx, N, lam=1e-5):
    """north_star: "MFMA utilisation (for the dense solve) against gfx950 peak".  In-kernel wall-clock stamps of the solve workgroup
    (cmlhip_debug_timestamps, 10 ns ticks) of the LAST of six iterations: load + Jacobi scaling, blocked LDL^T, backward substitution,
    tail (gauge projection + publishing x).  mfma_f64_util = the factorisation's matrix-core flops (trailing updates: 2 x 16^3 per
    16 x 16 x 16 tile product) / its duration / the fp64 matrix peak — tiny by construction (an 8N x 8N system is a 8N-pivot latency
    chain), which is why the honest figure is the microseconds."""
    import numpy as np
    NS = 128 + 5 * 1024 * 2
    out = np.zeros(NS, np.int64)
    ctx.sync()
    ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 1, None))
    for _ in range(6):
        ctx.ba_iteration_async(lam)
    ctx.sync()
    ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 0, out.ctypes.data_as(C.POINTER(C.c_longlong))))

    def seg(a, b):
        return float((out[b] - out[a]) * 0.01) if out[a] > 0 and out[b] > 0 else None
    m = 8 * N
    nb = (m + 15) // 16
    tiles = sum((nb - 1 - k) * (nb - k) // 2 for k in range(nb))          # lower-triangle trailing tiles updated behind block column k
    flops = tiles * 2.0 * 16 ** 3
    us = {"load": seg(48, 49), "factor": seg(49, 50), "backward": seg(51, 52), "tail": seg(52, 53), "total": seg(48, 53)}
    util = (flops / (us["factor"] * 1e-6) / (FP64_MATRIX_PEAK_TFLOPS * 1e12)) if us["factor"] else None
    return {"us": us, "mfma_f64_util": util, "mfma_flops": flops, "peak_tflops": FP64_MATRIX_PEAK_TFLOPS, "unknowns": m,
            "note": "k_ba_solve's own workgroup, in-kernel stamps of one iteration; the launch also carries the back-substitution / frame-step blocks (roofline-irrelevant)"}


def sequence_bench(device_id, seed, want_cpu):
    """north_star's unit of fan-out is a SEQUENCE SHARD: per frame trackWithMotionModel, per keyframe traceNewCoarse -> addNewFrame ->
    activatePoints -> addPoints -> run -> makeCoarseDepthL0 -> tryMarginalize -> marginalizePointsF -> marginalizeFrames (Hybrid.cpp:431-458,
    direct/Mapping.cpp:47-134) with a moving window (2 -> 7 keyframes, then sliding), the marginalisation prior live and image ids recycled.
    libcml_amd/sequence.py drives the host mirror in that order over a seeded 48-frame synthetic sequence of the BASELINE image shape.
    Pass 1 (timed, nobody watching): frames/s and host-clock ms per stage, run() split into upload / kernels / readback.  Pass 2 (want_cpu):
    the same sequence with tests/sequence_check.SequenceChecker attached — every stage replayed from the product's state by the oracle — for
    the parity verdict and the oracle's CPU time on the same stages."""
    import numpy as np
    from libcml_amd import device, sequence
    n_frames = 48
    seq = sequence.make_sequence(n_frames=n_frames, seed=0x5EED + (seed & 0xff))

    def one_pass(observer_factory=None):
        ctx = device.Ctx(device_id=device_id, max_frames=8, max_points=8192, max_residuals=8192 * 8)
        obs = observer_factory(ctx) if observer_factory else None
        pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels, observer=obs)
        t0 = time.perf_counter()
        stats = pipe.run(seq)
        dt = time.perf_counter() - t0
        out = (dict(stats), dt, pipe.timing_summary(), list(pipe.run_split), obs, [pipe.history[-1][0].copy(), pipe.history[-1][1].copy()], pipe.library_summary())
        pipe.close(); ctx.close()
        return out

    one_pass()                                                            # warm-up: allocations, pools, code objects
    stats, dt, stages, split, _o, last, lib = one_pass()
    t_boot = stages.get("bootstrap", {}).get("mean_ms", 0.0) * 1e-3
    R, t = last
    c = -R.T @ t; ct = -seq.R_true[n_frames - 1].T @ seq.t_true[n_frames - 1]
    steady = split[len(split) // 2:]                                      # keyframes with the window at its sliding size
    run_split = {k: float(np.median([s_[k] for s_ in steady])) for k in steady[0]} if steady else {}
    out = {"workload": "%d frames %dx%d, %d pyramid levels, keyframes at %s; window up to %d keyframes, %d marginalised, %d image ids recycled; "
                       "one context, tracker hypotheses batched" % (n_frames, seq.w, seq.h, seq.levels, seq.keyframes, stats["max_window"], stats["marginalized_frames"], stats["ids_recycled"]),
           "frames": n_frames, "keyframes": stats["keyframes"], "tracking_lost": stats["tracking_lost"],
           "frames_per_s": (n_frames - 1) / max(dt - t_boot, 1e-9), "seconds": dt, "bootstrap_s": t_boot,
           "ms_per_stage": {k: v for k, v in stages.items() if k != "bootstrap"},
           "per_frame_ms": sum(stages[k]["median_ms"] for k in ("pyramid_build", "trackWithMotionModel", "traceNewCoarse", "trackAndTrace") if k in stages),
           # library time of a TRACKED FRAME: pyramid (image worker hand-over) + the fused trackWithMotionModel / traceNewCoarse call (one enqueue, one host wait;
           # a trace redone because another hypothesis won is included) + the id's release — per frame of the shard
           "frame_ms": float(sum(lib[k]["total_ms"] for k in ("pyramid_build", "trackWithMotionModel", "traceNewCoarse", "trackAndTrace") if k in lib)) / (n_frames - 1),
           "traces_redone": int(stats.get("traces_redone", 0)),
           "per_keyframe_ms": sum(stages[k]["median_ms"] for k in ("addNewFrame", "activatePoints+addPoints", "run", "makeCoarseDepthL0", "tryMarginalize",
                                                                  "marginalizePointsF", "makeNewTraces", "marginalizeFrames") if k in stages),
           "library_ms_per_stage": {k: v for k, v in lib.items() if k != "bootstrap"},
           "library_seconds": float(sum(v["total_ms"] for k, v in lib.items() if k != "bootstrap")) * 1e-3,
           "library_frames_per_s": (n_frames - 1) / max(float(sum(v["total_ms"] for k, v in lib.items() if k != "bootstrap")) * 1e-3, 1e-9),
           "library_note": "time inside the library's own calls (C++ host mirror + C ABI, device sync behind each) — frames_per_s above is the wall clock of the Python "
                           "driver that stands in for the reference's host code around them (map accessors, motion model, pixel selector: out of scope)",
           "run_us_split_median": run_split,
           "run_us_split_note": "host clock inside DSOBundleAdjustment::run at the sliding window size: commit_window = edits handed over + index positions + pair records, one packed copy; "
                                "enqueue_first_pass = linearizeAll + applyRes enqueued; resident_state = adjoints / states / prior staged (second packed copy); enqueue_iterations = the iterations' "
                                "launches; wait_and_readback = cmlhip_ba_finish_run: the ONE host wait of run() (iterations, re-anchoring of the newest frame on the device, closing pass, one readback); "
                                "bookkeeping = the host lists brought up to date",
           "final_position_error_m": float(np.linalg.norm(c - ct)),
           "note": "stage times are host wall clock with a device sync behind every stage; makeNewTraces / makeCoarseDepthL0 include the Python stand-ins for the "
                   "reference's PixelSelector and map accessors (out of scope)"}
    # the same shard with the mapper on a host thread of its own (two contexts, two streams: Hybrid.cpp:103-106, direct/Mapping.cpp:3-41): frames keep
    # being tracked against the previous keyframe while directMap runs; fixed hand-over schedule (lag 1), so the inline run is the same computation
    try:
        def split_pass(threaded):
            ct = device.Ctx(device_id=device_id, max_frames=8, max_points=8192, max_residuals=8192 * 8)
            cm = device.Ctx(device_id=device_id, max_frames=8, max_points=8192, max_residuals=8192 * 8)
            pipe = sequence.SplitPipeline(ct, cm, seq.K, seq.w, seq.h, seq.levels, threaded=threaded)
            try:
                t0 = time.perf_counter()
                st = pipe.run(seq)
                dt_ = time.perf_counter() - t0
                boot = pipe.mapper.timing_summary().get("bootstrap", {}).get("mean_ms", 0.0) * 1e-3
                mlib = sum(float(np.sum(v)) for k, v in pipe.mapper.lib_times.items() if k != "bootstrap")
                return st, dt_, boot, [r[1:5] for r in pipe.front.results], (pipe.front.lib_s, mlib)
            finally:
                pipe.close(); ct.close(); cm.close()
        split_pass(True)                                                  # warm-up of the two contexts
        st_i, dt_i, boot_i, res_i, lib_i = split_pass(False)
        st_t, dt_t, boot_t, res_t, lib_t = split_pass(True)
        same = len(res_i) == len(res_t) and all(np.array_equal(np.asarray(x, np.float64).view(np.uint64), np.asarray(y, np.float64).view(np.uint64))
                                                for a, b in zip(res_i, res_t) for x, y in zip(a, b))
        out["two_threads"] = {"frames_per_s": (n_frames - 1) / max(dt_t - boot_t, 1e-9), "frames_per_s_same_schedule_one_thread": (n_frames - 1) / max(dt_i - boot_i, 1e-9),
                              "tracker_thread_stall_ms": 1e3 * st_t["tracker_stall_s"], "tracker_thread_library_ms_per_frame": 1e3 * lib_t[0] / (n_frames - 1),
                              "mapper_thread_library_ms_per_frame": 1e3 * lib_t[1] / (n_frames - 1),
                              "library_frames_per_s": (n_frames - 1) / max(lib_t[0], lib_t[1], 1e-9),
                              "library_note": "time inside the library's calls per thread; library_frames_per_s = frames / the busier thread's library time — what two host threads "
                                              "sustain when the host code around the calls keeps up.  The wall-clock figure above is bound by the mapper thread's PYTHON stand-ins "
                                              "(pixel selector, map accessors, a 5.6 MB gradient-image read-back per keyframe for it): tracker_thread_stall_ms is the tracker waiting for them",
                              "tracking_lost": st_t["tracking_lost"], "tracked_poses_bit_identical_to_one_thread": bool(same),
                              "note": "SplitPipeline: tracker context on the calling thread, mapper context (tracer + BA) on its own thread; keyframe j's reference lists and "
                                      "optimised pose are adopted before frame j + 2 is tracked; wall clock of the Python driver, bootstrap excluded"}
    except Exception as e:
        out["two_threads"] = {"error": repr(e)}
    if want_cpu:
        try:
            from tests import sequence_check as SC
            stats2, dt2, _st, _sp, chk, _l, _lib = one_pass(lambda ctx: SC.SequenceChecker(ctx, seq.K, seq.w, seq.h, seq.levels, strict=False))
            rep = chk.report
            out["parity_checked"] = True
            out["parity_ok"] = len(rep["failures"]) == 0
            out["parity"] = {"stages_replayed": rep["stages"], "worst": rep["worst"], "flips": rep["flips"], "failures": rep["failures"][:8],
                             "counts": {k: v for k, v in rep.items() if isinstance(v, int)},
                             "run_yardstick": rep.get("run_yardstick", []), "track_yardstick_used": rep.get("track_yardstick_used", 0),
                             "track_decisions_on_rounding": rep.get("track_decisions_on_rounding", []),
                             "run_yardstick_note": "runs whose distance from the oracle exceeded the fixed bars and were held against the oracle's own response to "
                                                   "rounding-sized noise instead: within the fixed bars of a member of {oracle with inverse depths perturbed by 1e-7 (2 draws) / "
                                                   "1e-6 (4 draws: the rounding of the fp32 accumulations), oracle built with the Release flags}, or no further from the oracle than "
                                                   "four times that ensemble's spread; tracked frames likewise, or separated from the oracle by ONE accept decision taken on a "
                                                   "margin below 1e-4 (tests/sequence_check.py)"}
            out["cpu_baseline"] = {"kind": "port", "cores": 1, "seconds": float(sum(chk.oracle_seconds.values())), "per_stage_s": {k: float(v) for k, v in chk.oracle_seconds.items()},
                                   "sample": "the oracle's replay of every stage of the same sequence from the product's state (tests/sequence_check.py: oracle/*.c through ctypes, "
                                             "checker build -O2, one thread; includes the checker's own set-up of each stage's window)"}
        except Exception as e:
            out["parity_checked"] = False; out["parity_error"] = repr(e)
    return out


def _shard_worker(idx, s_list, device_id, seed, barrier, queue):
    """one sequence shard of the shards-per-GPU sweep, in a process of its own (own context, own stream, own seeded sequence).  Every phase is
    bracketed by the barrier of ALL workers; a worker that is not part of a round only meets the barriers."""
    import hashlib
    import numpy as np
    out = {"idx": idx, "rounds": {}}
    try:
        from libcml_amd import device, sequence
        n_frames = 48
        seq = sequence.make_sequence(n_frames=n_frames, seed=0x5EED + (seed & 0xff), shard=idx)
        ctx = device.Ctx(device_id=device_id, max_frames=8, max_points=8192, max_residuals=8192 * 8)

        def one(share, repeats=1):
            """`repeats` runs of the shard, the fastest kept (the host side is a Python driver on a shared box); the tracked poses of every repeat must agree"""
            ctx.set_device_share(share)
            best = None
            for _ in range(repeats):
                pipe = sequence.DirectPipeline(ctx, seq.K, seq.w, seq.h, seq.levels)
                t0 = time.perf_counter()
                st = pipe.run(seq)
                ctx.sync()
                dt = time.perf_counter() - t0
                boot = pipe.timing_summary().get("bootstrap", {}).get("mean_ms", 0.0) * 1e-3
                lib = float(sum(float(np.sum(v)) for k, v in pipe.lib_times.items() if k != "bootstrap"))
                h = hashlib.sha256(b"".join(np.ascontiguousarray(x, np.float64).tobytes() for Rt in pipe.history for x in Rt)).hexdigest()
                pipe.close()
                r = {"seconds": dt - boot, "library_seconds": lib, "lost": int(st["tracking_lost"]), "poses": h}
                if best is not None and best["poses"] != h:
                    r["poses"] = "differs between repeats"
                    return r
                if best is None or r["library_seconds"] < best["library_seconds"]:
                    best = r
            return best
        one(1)                                                # warm-up: allocations, pools, code objects
        err = None
    except Exception as e:                                    # keep meeting the barriers: nobody may hang on a worker that failed
        err = repr(e)
    try:
      for S in s_list:
        active = idx < S
        # solo reference of every active shard at this share (G depends on it), one after the other
        for k in range(S):
            barrier.wait(timeout=600)
            if active and k == idx and err is None:
                try:
                    out["rounds"].setdefault(S, {})["solo"] = one(S)
                except Exception as e:
                    err = repr(e)
        for _rep in range(3):                                 # three timed passes, every one between barriers of ALL workers (no pass runs beside fewer shards than S); the fastest counts
            barrier.wait(timeout=600)
            if active and err is None:
                try:
                    r = one(S)
                    cur = out["rounds"].setdefault(S, {}).get("together")
                    if cur is not None and cur["poses"] != r["poses"]:
                        r["poses"] = "differs between repeats"
                        out["rounds"][S]["together"] = r
                    elif cur is None or r["library_seconds"] < cur["library_seconds"]:
                        out["rounds"][S]["together"] = r
                except Exception as e:
                    err = repr(e)
        barrier.wait(timeout=600)
    except Exception as e:                                    # a broken barrier (a worker died): report and leave
        err = err or repr(e)
    out["error"] = err
    try:
        ctx.close()
    except Exception:
        pass
    queue.put(out)


def shards_per_gpu_bench(device_id, seed, s_list=(1, 2, 4, 8)):
    """S sequence shards on ONE GPU — the one-device stand-in for BASELINE.json configs[3] (8 independent shards, one per GPU): S processes, each with
    its own context / stream / seeded 48-frame sequence (cmlhip_set_device_share(S): launches whose workgroups wait for one another size themselves for
    1 / S of the device), started together between barriers.  Per S: aggregate frames/s by the wall clock of the slowest shard and by the time inside
    the library's calls; every shard's tracked poses bit-identical to the same shard run alone at the same share."""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    n = max(s_list)
    barrier = mpc.Barrier(n)
    queue = mpc.Queue()
    procs = [mpc.Process(target=_shard_worker, args=(i, tuple(s_list), device_id, seed, barrier, queue), daemon=True) for i in range(n)]
    for p_ in procs:
        p_.start()
    res = []
    try:
        t_end = time.time() + 600
        while len(res) < n and time.time() < t_end:
            try:
                res.append(queue.get(timeout=2))
            except Exception:
                if not any(p_.is_alive() for p_ in procs) and queue.empty():      # every worker gone without a report: do not sit out the time-out
                    break
    finally:
        for p_ in procs:
            p_.join(timeout=30)
            if p_.is_alive():
                p_.terminate()
    if len(res) < n:
        return {"error": "%d of %d shard workers reported" % (len(res), n), "worker_errors": [r.get("error") for r in res if r.get("error")]}
    res.sort(key=lambda r: r["idx"])
    errs = [r["error"] for r in res if r.get("error")]
    out = {"frames_per_shard": 47, "S": {}, "errors": errs,
           "note": "S processes x one 48-frame sequence shard each on one MI355X (own context, stream, seeded sequence; bootstrap excluded); frames_per_s = 47 S / the slowest "
                   "shard's wall clock (Python driver included), library_frames_per_s = 47 S / the largest per-shard time inside the library's calls; "
                   "bit_identical = every shard's tracked poses equal its solo run at the same share; the timed pass runs three times back to back, the fastest counts"}
    for S in s_list:
        rows = [r["rounds"].get(S) or r["rounds"].get(str(S)) for r in res[:S]]
        if any(r is None or "together" not in r or "solo" not in r for r in rows):
            out["S"][str(S)] = {"error": "a shard did not finish"}
            continue
        wall = max(r["together"]["seconds"] for r in rows); lib = max(r["together"]["library_seconds"] for r in rows)
        solo_wall = float(sum(r["solo"]["seconds"] for r in rows)) / S
        out["S"][str(S)] = {"frames_per_s": 47.0 * S / wall, "library_frames_per_s": 47.0 * S / lib, "solo_frames_per_s_mean": 47.0 / solo_wall,
                            "bit_identical": all(r["together"]["poses"] == r["solo"]["poses"] for r in rows),
                            "tracking_lost": int(sum(r["together"]["lost"] for r in rows))}
    return out


class _NoGroup:
    """the secondary configurations run on rank 0 alone: same timing protocol, no process group"""
    rank, world = 0, 1

    def barrier(self):
        pass

    def max(self, v):
        return v

    def sum(self, v):
        return v


def setup_window(config, seed, rank, local_rank):
    """Synthetic window of `config` registered through the host mirror exactly as Hybrid::directMap would (pyramid build, addNewFrame,
    addPoints), one run() to upload it, then the device-resident loop armed.  config "C" = the config-B window + 1000 ORB observations of
    300 points mixed into the pose solution in every iteration (BASELINE.json configs[2]); "E" stores fp16 texels (configs[4])."""
    import numpy as np
    from libcml_amd import abi, device, host, synth
    hybrid = config == "C"
    wcfg = "B" if hybrid else config
    half = config == "E"
    W = synth.make_window(wcfg, seed=seed, shard=rank)                   # one independent sequence shard per rank
    ctx = device.Ctx(device_id=local_rank, max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N,
                     texel_format=abi.TEXEL_F16 if half else abi.TEXEL_F32)
    ba = host.window_to_host_ba(ctx, W, image_id_base=1000 * (rank + 1), levels=1)
    ba.set_param("iterations", 1)
    ind = None
    if hybrid:                                                            # mixed into the pose solution inside the device solve (BA.cpp:1327-1329)
        _, ipts, iobs = synth.indirect_observations(W, n_obs=1000, n_pts=300, seed=11 + rank)
        o = np.zeros(len(iobs), abi.REPROJ_OBS_DTYPE)
        for f in ("frame", "point", "gx", "gy"):
            o[f] = iobs[f]
        ba.set_param("mixedBundleAdjustment", 1)
        ba.set_indirect_points(ipts, o)
        ind = {"pts": ipts, "obs": o, "fx": W.K[0], "fy": W.K[1]}
    if not ba.run():                                                      # uploads the window, leaves adjoints/priors resident
        raise RuntimeError("BA run failed: " + ba.last_error())
    _, _, R = ctx.refresh_window_size()                                   # the window was uploaded by the C++ host mirror
    if not ba.begin_resident():                                           # frame states, adjoints, priors, gauge basis -> device
        raise RuntimeError("begin_resident failed: " + ba.last_error())
    if os.environ.get("BENCH_ARITH_RELAXED"):                              # development: A/B of CMLHIP_ARITH_RELAXED (the bit-exact gate then reports a mismatch)
        ctx.ba_set_arithmetic(True)
    return {"config": config, "wcfg": wcfg, "hybrid": hybrid, "half": half, "W": W, "ctx": ctx, "ba": ba, "R": R, "ind": ind}


def measure(S, steps, warmup, group, sync_extra=None, lam=1e-5):
    """The bench contract on one window: 300 set-up iterations (clocks ramped), `warmup` untimed steps, EXACTLY `steps` timed steps
    between barrier + device sync, MAX over ranks; the residual kernel's own dispatch duration sampled inside that region (HIP events
    attached to the dispatch, every stride-th step); then a region of its own for the Schur-reduce + solve span and eight repeat regions."""
    from libcml_amd import shard
    ctx = S["ctx"]
    for _ in range(300):                                                  # setup, not a step: ~20 ms of work so that the clocks have ramped
        ctx.ba_iteration_async(lam)                                       # before the W warm-up steps, whatever W is
    ctx.sync()
    for _ in range(warmup):
        ctx.ba_iteration_async(lam)
    # the contract region carries NO profile events (an event-carrying dispatch costs the pipeline ~3 us: with events on every 4th step a
    # --steps 20 run read 46.1 us per step against 43.9 at 200 steps).  The residual kernel's dispatch duration comes from the region
    # behind it, which carries events on EVERY residual-kernel dispatch of K further steps of the same loop (`roofline.launch_us`)
    stride = 8 if steps >= 80 else (4 if steps >= 16 else (2 if steps >= 4 else 1))
    ctx.profile_stride(stride)
    ctx.profile_enable(0)

    def run_steps():
        for _ in range(steps):
            ctx.ba_iteration_async(lam)

    def sync():
        ctx.sync()
        if sync_extra is not None:
            sync_extra()

    dt = shard.timed_region(group, sync, run_steps)
    # the Schur-reduce + solve group (K3 begin -> K6 end) is sampled in a region of its own, after the contract region: every
    # event-carrying dispatch costs the pipeline a few microseconds, and the contract region needs the roofline kernel's only
    ctx.profile_select(2)
    ctx.profile_enable((steps + stride - 1) // stride)
    shard.timed_region(group, sync, run_steps)
    _lin0, ss_ms, _empty, _n = ctx.profile_read()
    # every residual-kernel dispatch of K further steps (not the contract region: the events cost the pipeline)
    ctx.profile_stride(1)
    ctx.profile_select(1)
    ctx.profile_enable(steps)
    shard.timed_region(group, sync, run_steps)
    lin_all_ms, _s, _e, n_all = ctx.profile_read()
    lin_ms, n_samples = lin_all_ms, n_all
    ctx.profile_stride(stride)
    ctx.profile_select(3)
    # spread: the contract region above is ONE sample (K steps can be a millisecond); eight more regions of the same K steps, same
    # protocol, reported beside it (not used for `value`)
    ctx.profile_enable(0)
    rep_ms = sorted(1e3 * shard.timed_region(group, sync, run_steps) / steps for _ in range(8))
    st = ctx.ba_states()
    return {"dt": dt, "lin_ms": lin_ms, "ss_ms": ss_ms, "n_samples": n_samples, "rep_ms": rep_ms, "lin_all_ms": lin_all_ms, "n_all": n_all,
            "n_good": int(st["good"].sum()),
            "n_sampled": int((st["state"] != 1).sum())}   # residuals of the timed passes that enter the pixel loop and gather texels (OOB is absorbing, BA.cpp:68-72)


def _quat_to_R(q):
    import numpy as np
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def parity_gate(S, lam=1e-5):
    """"a throughput figure is only valid if the same run passes the fixture comparison".  One more iteration from the state
    the timed region left, its residual pass replayed on the oracle from the device's own state and compared bit for bit over ALL R
    residuals (tests/resident_check.py) — after the timed region, outside it.  Config C (hybrid) in addition: the ORB term the solve
    launch evaluated in that iteration against oracle/orc_base.c's reprojection accumulation + the Eigen-semantics LDLT on the poses the
    frames had at the solve (BA.cpp:2607-2700), bar 1e-8 as tests/test_hybrid_gpu.py."""
    import numpy as np
    ctx, ba, W, R = S["ctx"], S["ba"], S["W"], S["R"]
    try:
        from tests import resident_check as RC
        replay = RC.make_replay(ctx, ba, W)
        pre_w2c = None
        if S["hybrid"]:
            ctx.sync()
            _, pre_w2c = ctx.ba_resident_state()
        rep = RC.check_one_pass(ctx, replay, lam, with_records=True)
        replay.close()
        out = {"parity_checked": True, "parity_ok": bool(rep["ok"]), "parity": rep,
               "parity_note": "one resident iteration after the timed region; its residual pass (%s) replayed by oracle/orc_ba.c from the device's "
                              "pairs / thresholds / inverse depths / prior states: states, energies, JpJdF, centre projections and the re-materialised "
                              "74-float records compared bit for bit over all R residuals" % ("k_ba_lin_rs" if R >= 36 * 1024 else "k_ba_lin_rs4")}
        if S["hybrid"]:
            from tests import oracle_lib as O
            from tests import trk_setup as T
            ind = S["ind"]
            poses = np.zeros((W.N, 12))
            for k in range(W.N):
                poses[k, :9] = _quat_to_R(pre_w2c[k, :4]).ravel(); poses[k, 9:] = pre_w2c[k, 4:]
            _x, x6, _jp = ctx.ba_resident_indirect()
            M6o, b6o, _Jpo, usedo = T.oracle_reproj(poses, ind["pts"], ind["obs"], ind["fx"], ind["fy"])
            Mo = M6o.copy(); Mo[np.diag_indices(len(Mo))] *= (1 + lam)
            xo, rco = O.ldlt_solve(Mo, -b6o)
            err = float(np.abs(x6 - xo).max() / max(np.abs(xo).max(), 1e-300))
            out["parity"]["hybrid_x6_rel_err"] = err; out["parity"]["hybrid_used_obs"] = int(usedo.sum())
            out["parity_ok"] = bool(out["parity_ok"] and rco == 0 and err <= 1e-8)
            out["parity_note"] += "; hybrid term: the 6N indirect solution of that iteration against orc_reproj_accumulate + orc_ldlt_solve on the frames' poses at the solve, bar 1e-8"
        return out
    except Exception as e:
        return {"parity_checked": False, "parity_error": repr(e)}


def _valu_roof(config, R, launch_us):
    """Second roofline entry for the residual kernel: what its own instruction stream costs to ISSUE.  Inputs are the SQ counters of
    the committed rocprofv3 --pmc passes (profiles/round6_valu_roof_<config>.json, tools/valu_roof.py: SQ_WAVES, SQ_INSTS_VALU,
    SQ_ACTIVE_INST_VALU per launch) — instructions per wave and issue cycles per instruction as MEASURED on this kernel — and the wave
    slots the launch occupies; floor = waves per SIMD x instructions per wave x cycles per instruction / clock."""
    path = None
    for rnd in ("round6", "round5", "round4"):
        cand = os.path.join(ROOT, "profiles", "%s_valu_roof_%s.json" % (rnd, config))
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None
    try:
        d = json.load(open(path))
        waves = d["waves_per_launch"]; ipw = d["valu_insts_per_wave"]; cpi = d["cycles_per_valu_inst"]; clk = d["clock_ghz"]; simds = d["simds"]
        wps = waves / simds
        floor_us = max(wps, 1.0) * ipw * cpi / (clk * 1e3)
        return {"bound": "valu_issue", "insts_per_wave": ipw, "cycles_per_inst": cpi, "waves_per_launch": waves, "waves_per_simd": wps,
                "clock_ghz": clk, "floor_us": floor_us, "launch_us": launch_us, "frac": floor_us / launch_us if launch_us > 0 else None,
                "fp64_share": d.get("fp64_share"),
                "source": "profiles/%s (rocprofv3 --pmc passes of this bench command at commit %s; static ISA count beside it)" % (os.path.basename(path), d.get("commit", "?")),
                "note": "frac = the time the kernel's vector instructions need to issue (every SIMD's waves back to back, nothing else waiting) / "
                        "the measured dispatch duration; (waves per SIMD < 1 counts as 1: a lone wave cannot issue faster than its own stream)"}
    except Exception as e:
        return {"bound": "valu_issue", "error": repr(e)}


def roofline_object(S, M, lin_ms_local):
    config, R, half = S["config"], S["R"], S["half"]
    bytes_per_residual = READ_BYTES_PER_RESIDUAL["fp16" if half else "fp32"]
    achieved = R * bytes_per_residual / (lin_ms_local * 1e-3) / 1e9 if lin_ms_local > 0 else 0.0
    # memory-side bytes per launch of the residual kernel come from separate `rocprofv3 --pmc` passes of this same command
    # (tools/profile_bench.py; FETCH_SIZE doubled per the gfx950 note of the micro-architecture guide) and are only
    # quoted for the workload they were collected on
    traffic, traffic_src, rocprof_us = None, None, None
    for rnd in ("round6", "round5", "round4", "round3"):
        pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_linearize_%s.json" % (rnd, config))
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                traffic = pmc.get("traffic_bytes_per_launch")
                rocprof_us = pmc.get("linearize_avg_us")   # kernel-trace average of the same command (dispatches serialised by the profiler)
                traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; taken at commit %s: stale if the residual "
                               "kernel changed since)" % (os.path.basename(pmc_path), pmc.get("commit", "?")))
            except Exception:
                traffic = None
            break
    n_sampled = M["n_sampled"]
    kname = "k_ba_lin_rs (lane per residual, tiled fp16 level 0)" if R >= 36 * 1024 else "k_ba_lin_rs4 (4 lanes per residual)"
    roof = {"bound": "hbm", "kernel": kname + " = linearize + applyRes of the resident loop", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "n_sampled": n_sampled, "frac_sampled": (achieved / HBM_PEAK_GBS) * n_sampled / max(R, 1),
            "frac_sampled_note": "the same fraction billed only for the residuals that gather texels (state != OOB before the pass)",
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": R * bytes_per_residual, "bytes_per_residual": bytes_per_residual,
            "launch_us": 1e3 * lin_ms_local, "launch_samples": M["n_samples"],
            "launch_us_all_steps": 1e3 * M.get("lin_all_ms", 0.0), "launch_samples_all_steps": M.get("n_all", 0),
            "launch_us_note": "mean over EVERY residual-kernel dispatch of the K steps that follow the contract region (same loop, same state; the contract "
                              "region itself carries no events) of hipEventElapsedTime between the start and stop events attached to the dispatch "
                              "itself (hipExtLaunchKernelGGL): the kernel's begin / end timestamps, the quantity rocprofv3 --kernel-trace reports",
            "rocprof_avg_us": rocprof_us}
    if traffic and lin_ms_local > 0:
        # what the memory system actually moved for this launch (PMC, corrected as the guide prescribes) against the same roof: the residual
        # kernel's gathers pull whole 128-byte lines (5.1 per residual from the tiled fp16 image, 7.9 from a row-major one) for 276 / 468
        # algorithmic bytes, so the line traffic, not the algorithmic figure, is what the memory side sees
        roof["traffic_frac"] = traffic / (lin_ms_local * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["traffic_over_algorithmic"] = traffic / max(R * bytes_per_residual, 1)
    valu = _valu_roof(config, R, 1e3 * lin_ms_local)
    if valu is not None:
        roof["also_bound_by"] = [valu]
        roof["binding"] = "valu_issue" if (valu.get("frac") or 0) > roof["frac"] else "hbm"
    return roof


def workload_text(S):
    W = S["W"]
    return (("config C = config B + 1000 ORB reprojection residuals of 300 points mixed into the pose solution in every iteration; " if S["hybrid"] else "") +
            "config %s: %d-KF sliding window, %d active points, R=%d point-residuals, %dx%d level-0 gradient images, %s texels / fp32 arithmetic; "
            "1 step = 1 full Gauss-Newton BA iteration resident on the device (accumulate, Schur, solve + orthogonalize, back-substitution, frame + point step, "
            "pair precompute, linearize + applyRes)" % (S["wcfg"], W.N, W.P, S["R"], W.w, W.h, "fp16" if S["half"] else "fp32"))


def secondary_config(config, seed, local_rank, steps, warmup, relaxed=False):
    """BASELINE.json configs[2] (C) / configs[4] (E) beside the headline, on rank 0 of an N = 1 run: same protocol, own parity gate."""
    S = setup_window(config, seed, 0, local_rank)
    try:
        M = measure(S, steps, warmup, _NoGroup())
        par = parity_gate(S)
        out = {"workload": workload_text(S), "value": S["R"] * steps / M["dt"], "unit": "point-residuals/s", "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * M["dt"] / steps,
               "ms_per_step_repeats": {"min": M["rep_ms"][0], "median": M["rep_ms"][len(M["rep_ms"]) // 2], "max": M["rep_ms"][-1]},
               "schur_solve_ms": M["ss_ms"], "linearize_kernel_us": 1e3 * M["lin_ms"], "good_residuals": M["n_good"], "n_sampled": M["n_sampled"],
               "dtype": "f32", "roofline": roofline_object(S, M, M["lin_ms"])}
        out.update(par)
        if par.get("parity_checked") and not par.get("parity_ok"):
            out["invalid"] = "the residual pass after the timed region does NOT match the oracle: the figures above are void"
        if relaxed and not S["hybrid"]:
            out["relaxed_arithmetic"] = relaxed_arithmetic_leg(S, steps, warmup, out)
        if S["half"]:                                             # config E: how far fp16 texels sit from the reference's fp32 texels (checker side, outside every timed region)
            try:
                from tests import e_texel_check
                out["vs_fp32_texels"] = e_texel_check.device_vs_fp32_oracle(S["W"], local_rank)
            except Exception as e:
                out["vs_fp32_texels"] = {"error": repr(e)}
        return out
    finally:
        S["ba"].close(); S["ctx"].close()


def relaxed_arithmetic_leg(S, steps, warmup, exact):
    """The same window, same protocol, with cmlhip_ba_set_arithmetic(CMLHIP_ARITH_RELAXED) — the opt-in mode of the resident residual
    kernels (include/cmlhip.h) — after the exact figures and their bit-for-bit gate: K1 and the step beside the exact mode's, and the pass after
    the timed region held against the oracle at the stated tolerances (tests/test_relaxed_arithmetic_gpu.py: 1e-4 relative, classification
    identical up to a reported count)."""
    ctx = S["ctx"]
    try:
        from tests import resident_check as RC
        ctx.ba_set_arithmetic(True)
        M = measure(S, steps, warmup, _NoGroup())
        replay = RC.make_replay(ctx, S["ba"], S["W"])
        ctx.sync()
        pre = ctx.ba_states()
        ctx.ba_iteration_async(1e-5)
        ctx.sync()
        rep = RC.compare_pass_tolerant(ctx, replay, pre)
        replay.close()
        flips = max(rep["new_state_flips"], rep["state_flips"], rep["good_flips"])
        ok = (flips <= max(2, S["R"] // 5000) and max(rep["energy_rel"], rep["new_energy_rel"], rep["new_energy_wo_rel"], rep["jpjdf_rel_p999"]) < 1e-4
              and rep["jpjdf_rel"] < 1e-3 and rep["center_abs"] < 1e-3)
        k1 = 1e3 * M["lin_ms"]
        return {"mode": "CMLHIP_ARITH_RELAXED (opt-in; the headline and the figures above are CMLHIP_ARITH_EXACT)",
                "value": S["R"] * steps / M["dt"], "ms_per_step": 1e3 * M["dt"] / steps, "linearize_kernel_us": k1,
                "linearize_kernel_us_exact": exact["linearize_kernel_us"], "kernel_ratio": k1 / exact["linearize_kernel_us"],
                "roofline_frac": roofline_object(S, M, M["lin_ms"])["frac"],
                "tolerance_checked": True, "tolerance_ok": bool(ok), "tolerance": rep,
                "tolerance_note": "the residual pass after the timed region replayed by the oracle from the device's state: energies relative to the oracle's (bar 1e-4), JpJdF rows against their largest entry (99.9 % within 1e-4, every row within 1e-3), "
                                  "centre projections in pixels (bar 1e-3), residuals classified differently (bar R / 5000)"}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        ctx.ba_set_arithmetic(False)

def _commit():
    """HEAD where there is a .git; on the GPU box (snapshot without .git) what libcml_amd/build.py left beside the libraries"""
    try:
        if os.path.isdir(os.path.join(ROOT, ".git")):
            return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
        return open(os.path.join(ROOT, "libcml_amd", "BUILD_COMMIT")).read().strip() or None
    except Exception:
        return None


LINE_LIMIT = 3072                         # the driver keeps ~8 KB of stdout: the last line stays far below that (tests/test_bench_contract_gpu.py)


def _r(v, sig=5):
    """numbers of the compact line at `sig` significant digits"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float("%.*g" % (sig, float(v)))
    except Exception:
        return None


def compact_line(out, detail_path, contract_only=False):
    """The ONE stdout line: the bench contract + roofline + cpu_baseline + one-number summaries.  Everything else (notes, per-stage tables,
    min/max lists, sweeps) is in `detail_path` (the full object `out`)."""
    roof = out.get("roofline") or {}
    also = [{"bound": a.get("bound"), "frac": _r(a.get("frac"))} for a in roof.get("also_bound_by", []) if isinstance(a, dict)]
    cb = out.get("cpu_baseline") or {}
    line = {
        "metric": out["metric"], "value": out["value"], "unit": out["unit"], "n_gpus": out["n_gpus"], "steps": out["steps"], "warmup": out["warmup"],
        "ms_per_step": out["ms_per_step"], "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": None, "dtype": out["dtype"], "data": out["data"],
        "config": {"workload": out["config"]["workload"].split("; 1 step")[0], "shards": out["config"]["shards"], "parallelism": out["config"]["parallelism"]},
        "schur_solve_ms": _r(out.get("schur_solve_ms")),
        "roofline": {"bound": roof.get("bound"), "kernel": str(roof.get("kernel", "")).split(" ")[0], "achieved": roof.get("achieved"), "peak": roof.get("peak"),
                     "unit": roof.get("unit"), "frac": roof.get("frac"), "traffic": _r(roof.get("traffic"), 6),
                     "traffic_over_algorithmic": _r(roof.get("traffic_over_algorithmic")), "launch_us": roof.get("launch_us"),
                     "launch_samples": roof.get("launch_samples"), "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
                     "rocprof_avg_us": _r(roof.get("rocprof_avg_us")), "also_bound_by": also},
        "parity_checked": bool(out.get("parity_checked")), "parity_ok": bool(out.get("parity_ok")),
    }
    if out.get("invalid"):
        line["invalid"] = True
    if cb:
        st = cb.get("single_thread") or {}
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "single_thread_value": _r(st.get("value")), "sample": "median of 20 runs x 5 GN iterations of the same window, oracle C port (-O3 -march=native, OpenMP on one socket)"}
        if cb.get("value") is None:
            line["cpu_baseline"]["sample"] = str(cb.get("sample"))[:120]
        if out.get("gpu_over_cpu_single_socket") is not None:
            line["gpu_over_cpu"] = _r(out["gpu_over_cpu_single_socket"], 4)
    if contract_only:
        line["detail"] = detail_path
        return line
    if isinstance(out.get("configs"), dict):
        line["configs"] = {}
        for k, c in out["configs"].items():
            if "error" in c:
                line["configs"][k] = {"error": str(c["error"])[:80]}
            else:
                line["configs"][k] = {"value": _r(c.get("value")), "ms_per_step": _r(c.get("ms_per_step")), "frac": _r((c.get("roofline") or {}).get("frac")),
                                      "parity_ok": bool(c.get("parity_ok"))}
                v32 = c.get("vs_fp32_texels")
                if isinstance(v32, dict) and "H_A_jacobi_rel" in v32:      # config E: fp16 texels against the reference's fp32 texels (Jacobi-scaled H_A, residuals classified differently)
                    line["configs"][k]["vs_fp32_texels"] = {"H_A_jacobi_rel": _r(v32["H_A_jacobi_rel"], 3), "class_flips": v32.get("class_flips")}
    sq = out.get("sequence")
    if isinstance(sq, dict):
        if "error" in sq:
            line["sequence"] = {"error": str(sq["error"])[:80]}
        else:
            par = sq.get("parity") or {}
            line["sequence"] = {"frames_per_s": _r(sq.get("frames_per_s")), "library_frames_per_s": _r(sq.get("library_frames_per_s")), "frame_ms": _r(sq.get("frame_ms"), 4),
                                "run_ms": _r(sum((sq.get("run_us_split_median") or {}).values()) * 1e-3 or None),
                                "parity_ok": sq.get("parity_ok"), "yardstick_used": len(par.get("run_yardstick", []) or []) + int(par.get("track_yardstick_used", 0) or 0)}
            spg = sq.get("shards_per_gpu") or {}
            if isinstance(spg.get("S"), dict):                # S sequence shards on this one GPU: aggregate frames/s inside the library's calls
                line["sequence"]["shards_per_gpu"] = {k: _r(v.get("library_frames_per_s"), 4) for k, v in spg["S"].items() if isinstance(v, dict) and "library_frames_per_s" in v}
                line["sequence"]["shards_bit_identical"] = all(bool(v.get("bit_identical")) for v in spg["S"].values() if isinstance(v, dict) and "bit_identical" in v)
            tt = sq.get("two_threads") or {}
            if "frames_per_s" in tt:
                line["sequence"]["frames_per_s_two_threads"] = _r(tt["frames_per_s"])
                line["sequence"]["library_frames_per_s_two_threads"] = _r(tt.get("library_frames_per_s"))
                line["sequence"]["two_threads_bit_identical"] = tt.get("tracked_poses_bit_identical_to_one_thread")
    tr = out.get("tracker")
    if isinstance(tr, dict):
        if "error" in tr:
            line["tracker"] = {"error": str(tr["error"])[:80]}
        else:
            line["tracker"] = {"optimize_ms_1": _r((tr.get("optimize_device_resident_1_hyp") or {}).get("kernel_ms")),
                               "optimize_ms_50": _r((tr.get("optimize_device_resident_50_hyp") or {}).get("kernel_ms"))}
    sv = out.get("solve")
    if isinstance(sv, dict) and "error" not in sv:
        line["solve"] = {"us": _r((sv.get("us") or {}).get("total")), "mfma_f64_util": _r(sv.get("mfma_f64_util"))}
    ss = out.get("sequence_shards")
    if isinstance(ss, dict):
        line["sequence_shards"] = {k: _r(v) for k, v in ss.items() if k in ("shards", "frames_per_s_total", "tracking_lost_total")} if "error" not in ss else {"error": str(ss["error"])[:80]}
    line["commit"] = out.get("commit")
    line["detail"] = detail_path
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="B", help="synthetic window (libcml_amd.synth.CONFIGS); B is the benchmark workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs C / E, tracker, sequence and multi-window objects (headline + roofline + parity only)")
    ap.add_argument("--extras", action="store_true", help="also run the sweeps (multi-window S = 2..32, stream groups, relaxed-arithmetic legs); they go to bench_detail.json only")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full (uncompacted) result object is written")
    ap.add_argument("--launch-only", action="store_true", help="exercise the rank launch + process group only (no device work; CPU test)")
    ap.add_argument("--cpu-baseline-worker", nargs=3, metavar=("CONFIG", "SEED", "MODE"), help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_worker:
        return _cpu_worker(args.cpu_baseline_worker[0], int(args.cpu_baseline_worker[1]), args.cpu_baseline_worker[2])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:              # not under torchrun: start the ranks ourselves
        sys.exit(_spawn_ranks(args.gpus))
    if args.launch_only:
        return _launch_only(args)

    # S = 1, 2, 4, 8 sequence shards on this one GPU (eight processes, ~25 s): the one-device stand-in for configs[3].  FIRST, while this process holds no
    # device context: with its own streams alive beside the workers' (eight processes x three streams) the device's hardware queues are oversubscribed and
    # the workers' latencies double (measured: solo 770 -> 500 frames/s with the same workers started from the end of this function)
    spg_early = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_extras and args.config == "B" and not os.environ.get("CML_BENCH_NO_SHARDS_PER_GPU")):
        try:
            from libcml_amd import device as _dev
            if _dev.lib().cmlhip_device_count() > 0:
                spg_early = shards_per_gpu_bench(0, 0xC0FFEE)
        except Exception as e:
            spg_early = {"error": repr(e)}
    from libcml_amd import shard
    rank, local_rank, world = shard.env_world()
    if world != args.gpus and world > 1:
        args.gpus = world
    if os.environ.get("CML_BENCH_SHARE_DEVICE"):                          # tests: every rank on device 0 (the N > 1 path of this file on a one-GPU box, with CML_SHARD_BACKEND=gloo)
        local_rank = 0
    import torch
    dev = None
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    group = shard.Group(device=dev)

    _dbg('group ready')
    seed = 0xC0FFEE
    S = setup_window(args.config, seed, rank, local_rank)
    W, ctx, ba, R, hybrid, half, wcfg = S["W"], S["ctx"], S["ba"], S["R"], S["hybrid"], S["half"], S["wcfg"]
    N, P = W.N, W.P
    _dbg('window resident')

    def sync_extra():
        if dev is not None:
            torch.cuda.synchronize()

    M = measure(S, args.steps, args.warmup, group, sync_extra)
    dt = M["dt"]
    _dbg('timed regions done')
    ranks_joined = int(round(group.sum(1.0)))
    parity = parity_gate(S) if rank == 0 else {"parity_checked": False}
    total_units = group.sum(float(R) * args.steps)
    lin_ms_max = group.max(M["lin_ms"])
    ss_ms_max = group.max(M["ss_ms"])

    # N > 1: the unit north_star fans out is a sequence shard — every rank also runs ONE 48-frame shard (its own seeded sequence) through the
    # host mirror between two barriers; the line carries the aggregate frames/s (weak scaling, no data-path collective)
    seq_shards = None
    if (world > 1 or os.environ.get("CML_BENCH_FORCE_SEQ_SHARDS")) and not args.no_extras:      # (the variable: exercise this path with one rank)
        # (every collective below is reached by every rank whatever happens on one of them: a rank that fails keeps meeting the others)
        ok, err, seq, sctx, st, dts = 1.0, None, None, None, {"tracking_lost": 0}, 0.0
        try:
            from libcml_amd import device, sequence
            seq = sequence.make_sequence(n_frames=48, seed=0x5EED, shard=rank)
            sctx = device.Ctx(device_id=local_rank, max_frames=8, max_points=8192, max_residuals=8192 * 8)
        except Exception as e:
            ok, err = 0.0, repr(e)
        if group.sum(ok) == world:
            for rep in range(2):                                          # first pass warms pools and code objects
                group.barrier()
                t0 = time.perf_counter()
                try:
                    pipe = sequence.DirectPipeline(sctx, seq.K, seq.w, seq.h, seq.levels)
                    st = pipe.run(seq)
                    sctx.sync()
                    pipe.close()
                except Exception as e:
                    ok, err = 0.0, repr(e)
                group.barrier()
                dts = group.max(time.perf_counter() - t0)
        all_ok = group.sum(ok) == world
        lost = int(round(group.sum(float(st["tracking_lost"]))))
        if sctx is not None:
            try:
                sctx.close()
            except Exception:
                pass
        if all_ok:
            seq_shards = {"shards": world, "frames_per_shard": 48, "seconds_max_over_ranks": dts, "frames_per_s_total": 48.0 * world / dts, "tracking_lost_total": lost,
                          "note": "one seeded 48-frame sequence shard per rank through libcml_amd/sequence.py (bootstrap included), barrier on both sides, max over ranks"}
        else:
            seq_shards = {"error": err or "a rank failed"}
    _dbg('reductions done')
    if rank == 0:
        rep_ms = M["rep_ms"]
        out = {
            "metric": _baseline_metric(),
            "value": total_units / dt, "unit": "point-residuals/s",
            "n_gpus": ranks_joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_repeats": {"n": len(rep_ms), "min": rep_ms[0], "median": rep_ms[len(rep_ms) // 2], "max": rep_ms[-1],
                                    "note": "eight further timed regions of the same K steps after the contract region (same barrier / sync protocol)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(S), "shards": world, "parallelism": "1 independent window per GPU, RCCL barrier only"},
            "schur_solve_ms": ss_ms_max, "linearize_kernel_us": 1e3 * lin_ms_max, "good_residuals": M["n_good"], "n_sampled": M["n_sampled"],
            "roofline": roofline_object(S, M, M["lin_ms"]),
        }
        out.update(parity)
        if parity.get("parity_checked") and not parity.get("parity_ok"):
            out["invalid"] = "the residual pass after the timed region does NOT match the oracle: the figures above are void"
        if seq_shards is not None:
            out["sequence_shards"] = seq_shards
        extras = not args.no_extras and world == 1 and args.config == "B"
        if extras:
            try:
                out["solve"] = solve_phases(ctx, N)
            except Exception as e:
                out["solve"] = {"error": repr(e)}
            if args.extras:
                out["relaxed_arithmetic"] = relaxed_arithmetic_leg(S, args.steps, args.warmup, out)      # opt-in mode, after the exact headline and its gate
        ba.close(); ctx.close()                                   # (the objects below build their own contexts)
        if extras:
            out["configs"] = {}
            for cfg in ("C", "E"):                                # BASELINE.json configs[2] and configs[4]: driver-visible, each with its own oracle gate
                try:
                    out["configs"][cfg] = secondary_config(cfg, seed, local_rank, args.steps, args.warmup, relaxed=args.extras)
                except Exception as e:
                    out["configs"][cfg] = {"error": repr(e)}
            if args.extras:
                try:
                    out["multi_window"] = multi_window_bench(local_rank, seed, wcfg, max(args.steps, 50), half)
                except Exception as e:
                    out["multi_window"] = {"error": repr(e)}
            try:
                out["tracker"] = tracker_bench(local_rank, seed, not args.no_cpu_baseline)
            except Exception as e:
                out["tracker"] = {"error": repr(e)}
            try:
                out["sequence"] = sequence_bench(local_rank, seed, not args.no_cpu_baseline)
            except Exception as e:
                out["sequence"] = {"error": repr(e)}
            if spg_early is not None:                         # (measured at the start of main(), before this process had queues of its own on the device)
                if isinstance(out.get("sequence"), dict):
                    out["sequence"]["shards_per_gpu"] = spg_early
                else:
                    out["shards_per_gpu"] = spg_early
        if not args.no_cpu_baseline and world == 1:              # the CPU baseline is reported at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(wcfg, seed)          # (config C: the photometric window of config B; the ORB term is not part of the CPU port's timing)
                if out["cpu_baseline"].get("value"):
                    out["gpu_over_cpu_single_socket"] = out["value"] / out["cpu_baseline"]["value"]     # what north_star's ">= 30x" is judged on
                    out["gpu_over_cpu_single_thread"] = out["value"] / out["cpu_baseline"]["single_thread"]["value"]
                    mm = out["cpu_baseline"].get("value_minmax")
                    if mm:
                        out["gpu_over_cpu_single_socket_minmax"] = [out["value"] / mm[1], out["value"] / mm[0]]
                if hybrid and isinstance(out["cpu_baseline"], dict):
                    out["cpu_baseline"]["sample"] = str(out["cpu_baseline"].get("sample", "")) + " — the config-B window WITHOUT the 1000 ORB residuals of config C"
            except Exception as e:      # the checker must never take the measurement down
                out["cpu_baseline"] = {"value": None, "unit": "point-residuals/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        out["commit"] = _commit()
        try:
            with open(args.detail, "w") as f:
                json.dump(out, f, indent=1)
        except Exception as e:
            print("[bench] could not write %s: %r" % (args.detail, e), file=sys.stderr)
        line = json.dumps(compact_line(out, os.path.relpath(args.detail, ROOT)), separators=(",", ":"))
        if len(line) > LINE_LIMIT:                                # never again a line the driver cannot read: drop the summaries, keep the contract
            small = compact_line(out, os.path.relpath(args.detail, ROOT), contract_only=True)
            line = json.dumps(small, separators=(",", ":"))
    else:
        line = None
        ba.close(); ctx.close()
    group.barrier()
    group.close()
    if line is not None:
        # the JSON line goes out LAST: RCCL prints a version banner through C stdio when the first communicator forms, which would otherwise
        # land behind a line printed earlier (stdout is block-buffered under a pipe)
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
