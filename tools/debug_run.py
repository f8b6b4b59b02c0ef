import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from libcml_amd import device, host
from tests import ba_setup as S, ba_ref_run, oracle_lib as O
cfg = sys.argv[1] if len(sys.argv) > 1 else "tiny"
for iters in (0, 1, 2, 3, 4):
    I = S.make_inputs(cfg)
    ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
    ba = host.window_to_host_ba(ctx, I.W)
    ba.set_param("iterations", iters)
    assert ba.run(), ba.last_error()
    ref = ba_ref_run.oracle_run(I, iterations=iters)
    st, alive, good = ba.residual_states()
    flips = int((good.astype(bool) != ref["good"]).sum())
    dR = max(np.abs(ba.frame(k)["R"] - ref["poses"][k][0]).max() for k in range(I.N))
    dt = max(np.abs(ba.frame(k)["t"] - ref["poses"][k][1]).max() for k in range(I.N))
    idp, pal, ng = ba.points()
    print("iters", iters, "flips", flips, "dR %.2e dt %.2e" % (dR, dt), "didp %.2e" % np.abs(idp / ref["idepth"] - 1).max(),
          "th dev %.3f ref %.3f" % (ba.frame(I.N - 1)["th"], ref["th"]), "E dev", ba.energies()[-3:], "E ref", ref["log"]["energy"][-3:])
    ba.close(); ctx.close()
