"""Host-driven loop (DSOBundleAdjustment::run: one synchronous device call per reference statement) vs run() itself,
which keeps the loop on the device under the default parameters (runResident with the early-exit mirror), per Gauss-Newton iteration, config B."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcml_amd import device, host, synth
W = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "B")
for mode in ("host", "resident"):
    ts = {4: 1e9, 24: 1e9}
    for its in (4, 4, 24, 4, 24, 4, 24):
        ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
        ba = host.window_to_host_ba(ctx, W, levels=1)
        ba.set_param("iterations", its); ba.set_param("ThOptIterations", 0.0)
        t0 = time.perf_counter()
        ok = ba.run_host_loop() if mode == "host" else ba.run()
        ts[its] = min(ts[its], time.perf_counter() - t0)
        assert ok and ba.counts()["iterations"] == its
        ba.close(); ctx.close()
    print("%-8s loop: %.1f us per Gauss-Newton iteration (difference of a 24- and a 4-iteration run)" % (mode, (ts[24] - ts[4]) / 20 * 1e6))
