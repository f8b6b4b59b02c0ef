import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, host
from tests import ba_setup as S, sequence_check as SC, oracle_lib as O
I0 = S.make_inputs("medium")
ctx = device.Ctx(max_frames=I0.N + 1, max_points=I0.P, max_residuals=I0.P * (I0.N + 1))
ba = host.window_to_host_ba(ctx, I0.W)
ba.set_param("Minimum iDepth Hessian Marginlaization", 1.0)
mode = sys.argv[1] if len(sys.argv) > 1 else "resident"
ok = ba.run() if mode == "resident" else ba.run_host_loop()
assert ok
ba.flag_frame(1)
assert ba.try_marginalize()
grads0 = ba.synth_inputs["grads0"]
before = ba.export(); pb = ba.prior(); alg = ba.algebra()
assert ba.marginalize_points()
info = dict(before=before, prior_before=pb, algebra=alg, prior_after=ba.prior(), after=ba.export(), grads0=grads0)
chk = SC.SequenceChecker(ctx, I0.W.K, I0.W.w, I0.W.h, 1, strict=False)
chk.on_marginalize_points(info)
print(mode, chk.report["worst"], chk.report["failures"])
H1, b1 = info["prior_after"]
# block structure of the difference
fr, pt, rs = before
I = SC.inputs_from_export(fr, pt, rs, grads0, I0.W.K, I0.W.w, I0.W.h, reset_active=False)
ob = S.OracleBA(I)
ob.view("r_energy", I.R, np.float32)[:] = rs["state_energy"][I.residual_ids].astype(np.float32)
sel_pts = np.flatnonzero(pt["toMarginalize"] != 0)
slot = {int(p): k for k, p in enumerate(I.point_ids)}
sel = np.array([slot[int(p)] for p in sel_pts], np.int32)
ob.view("r_lin", I.R, np.uint8)[:] = 0
ng = ob.relinearize_points(sel)
M, Mb, Msc, Mbsc = ob.marginalize_points(sel)
D = H1 - 0.25 * (M - Msc)
n = len(D); N = (n - 4) // 8
print("sel", len(sel), "ngood", ng, "max|H1|", np.abs(H1).max(), "max|M|", np.abs(M).max(), "max|Msc|", np.abs(Msc).max())
for a in range(N):
    print(" ".join("%8.1e" % np.abs(D[4 + 8 * a:12 + 8 * a, 4 + 8 * b:12 + 8 * b]).max() for b in range(N)))
print("diag rel of block (0,0):", np.abs(np.diag(D)[4:12] / np.diag(H1)[4:12]))
