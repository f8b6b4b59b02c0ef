"""How far does the reference's Release arithmetic sit from the reading the parity tests pin?

The parity tests compare the device path, bit for bit, with oracle/*.c compiled -ffp-contract=off: the written statement order IS the
arithmetic.  The reference ships as `-O3 -march=native` (CMakeLists.txt:39-46), where the compiler may fuse a*b+c into one rounding
wherever it likes — a different, build-dependent, equally legitimate set of bits.  This tool runs the SAME oracle sources in both
builds (make / make contract) over the same seeded windows and reports what moves: per-residual energies, IN/OUTLIER/OOB decisions,
the accumulated system, the LM step, and the state after a few Gauss-Newton iterations.  CPU only; writes a text report.

    python tools/contraction_sensitivity.py [config ...]  > profiles/round2_contraction_sensitivity.txt
"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_variant(soname, target, config, iters, out):
    """child process: one oracle build, `iters` Gauss-Newton iterations, everything of interest dumped to an .npz"""
    from tests import oracle_lib as O
    from tests import ba_setup as S
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), target], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "oracle", soname))
    L.orc_ba_create.restype = C.POINTER(O.OrcBAWindow)
    L.orc_ba_linearize_one.restype = C.c_double
    L.orc_ba_calc_l_energy.restype = C.c_double
    L.orc_ba_calc_m_energy.restype = C.c_double
    O._lib = L
    I = S.make_inputs(config, seed=7)
    ob = S.OracleBA(I)
    rec = {}
    lin = ob.linearize(); ob.apply(1)
    for it in range(iters):
        rec["energy_%d" % it] = ob.view("r_energy", I.R, np.float32).copy() if hasattr(ob.w.contents, "r_energy") else np.zeros(0)
        rec["state_%d" % it] = ob.view("r_state", I.R, np.int32).copy() if hasattr(ob.w.contents, "r_state") else np.zeros(0)
        rec["total_%d" % it] = np.array([lin.energy])
        L.orc_ba_backup_points(ob.w)
        H = ob.accumulate()
        rec["HA_%d" % it] = H[0].copy(); rec["bA_%d" % it] = H[1].copy(); rec["Hsc_%d" % it] = H[4].copy(); rec["bsc_%d" % it] = H[5].copy()
        x, rc = ob.solve(1e-5, *H)
        rec["x_%d" % it] = x.copy()
        step, _ = ob.backsub(x)
        rec["pstep_%d" % it] = step.copy()
        L.orc_ba_step_points(ob.w, None)
        lin = ob.linearize(); ob.apply(1)
    rec["idepth"] = np.array([ob.w.contents.points[i].idepth for i in range(I.P)], np.float64)
    rec["total_final"] = np.array([lin.energy])
    np.savez(out, **rec)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    s = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.where(s > 0, d / s, 0.0)
    return float(np.nanmax(r)) if r.size else 0.0, float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)) if a.size else 0.0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run_variant(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6])
        return
    configs = sys.argv[1:] or ["small", "B"]
    iters = 4
    print("contraction sensitivity of the oracle (same sources; `make` = -ffp-contract=off, `make contract` = -O3 -march=native -ffp-contract=fast)")
    print("gcc:", subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0])
    for cfg in configs:
        outs = []
        for so, tgt in (("libcml_oracle.so", "libcml_oracle.so"), ("libcml_oracle_contract.so", "contract")):
            out = "/tmp/contraction_%s_%s.npz" % (cfg, tgt.replace(".", "_"))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", so, tgt, cfg, str(iters), out], cwd=ROOT)
            outs.append(np.load(out))
        A, B = outs
        print("\n== config %s, %d Gauss-Newton iterations (lambda 1e-5), seed 7" % (cfg, iters))
        for it in range(iters):
            ea, eb = A["energy_%d" % it], B["energy_%d" % it]
            sa, sb = A["state_%d" % it], B["state_%d" % it]
            nbits = int((ea.view(np.uint32) != eb.view(np.uint32)).sum()) if ea.size else -1
            print("  iteration %d: residual energies differing in any bit %d / %d (max rel %.2e); state decisions differing %d; total energy rel %.2e"
                  % (it, nbits, ea.size, rel(ea, eb)[0], int((sa != sb).sum()), rel(A["total_%d" % it], B["total_%d" % it])[0]))
            s0 = max(np.linalg.norm(A["pstep_0"]), 1e-300)
            print("      H_A fro-rel %.2e  b_A %.2e  H_sc %.2e  b_sc %.2e | LM step x: fro-rel %.2e | point steps: |step| / |first step| %.2e, difference / |first step| %.2e"
                  % (rel(A["HA_%d" % it], B["HA_%d" % it])[1], rel(A["bA_%d" % it], B["bA_%d" % it])[1], rel(A["Hsc_%d" % it], B["Hsc_%d" % it])[1],
                     rel(A["bsc_%d" % it], B["bsc_%d" % it])[1], rel(A["x_%d" % it], B["x_%d" % it])[1],
                     np.linalg.norm(A["pstep_%d" % it]) / s0, np.linalg.norm(A["pstep_%d" % it] - B["pstep_%d" % it]) / s0))
        print("  after %d iterations: inverse depths fro-rel %.2e, total energy rel %.2e" % (iters, rel(A["idepth"], B["idepth"])[1], rel(A["total_final"], B["total_final"])[0]))


if __name__ == "__main__":
    main()
