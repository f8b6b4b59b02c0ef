"""Differences between the host-driven loop (record-based pair blocks) and the device-resident loop (matrix-core tiles of the
resident residual kernel) after 5 iterations: the two sum the fp32 pair blocks in different orders."""
import sys
import numpy as np
sys.path.insert(0, ".")
from libcml_amd import device, host
from tests import ba_setup as S

for config in sys.argv[1:] or ["small", "medium", "B"]:
    res = []
    for mode in ("host", "resident"):
        I = S.make_inputs(config)
        ctx = device.Ctx(max_frames=I.N, max_points=I.P, max_residuals=I.R)
        ba = host.window_to_host_ba(ctx, I.W)
        ba.set_param("iterations", 5)
        ba.set_param("ThOptIterations", 0.0)
        ok = ba.run_host_loop() if mode == "host" else ba.run_resident()
        assert ok
        idp, alive, ng = ba.points()
        st, ralive, good = ba.residual_states()
        frames = [ba.frame(k) for k in range(I.N)]
        res.append((idp.copy(), good.copy(), frames, ba.energies(8)))
        ba.close(); ctx.close()
    (idp_h, good_h, fr_h, e_h), (idp_r, good_r, fr_r, e_r) = res
    ds = max(np.abs(a["state"] - b["state"]).max() / max(1.0, np.abs(a["state"]).max()) for a, b in zip(fr_h, fr_r))
    dR = max(np.abs(a["R"] - b["R"]).max() for a, b in zip(fr_h, fr_r))
    dt = max(np.abs(a["t"] - b["t"]).max() for a, b in zip(fr_h, fr_r))
    print(config, "good flips", int((good_h != good_r).sum()), "state", ds, "R", dR, "t", dt, "idepth rel", np.abs(idp_h / idp_r - 1).max(),
          "energy rel", np.abs(np.array(e_h) / np.array(e_r) - 1).max())
