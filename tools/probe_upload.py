"""Per-keyframe host->device costs of the BA window at the benchmark's size: cmlhip_ba_upload_window, set_pairs, and the first
linearize after it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device, host, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
W = synth.make_window(cfg)
ctx = device.Ctx(max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N)
ba = host.window_to_host_ba(ctx, W, image_id_base=1000, levels=1)
ba.set_param("iterations", 1)
assert ba.run()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); ok = ba.run(); ts.append(time.perf_counter() - t0)
print("DSOBundleAdjustment::run with 1 iteration (window rebuild + upload + 1 resident iteration + readback): %.0f us (min of 8), window N=%d P=%d" % (min(ts) * 1e6, W.N, W.P))
ba.set_param("iterations", 4)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); ok = ba.run(); ts.append(time.perf_counter() - t0)
print("the same with 4 iterations: %.0f us" % (min(ts) * 1e6))
