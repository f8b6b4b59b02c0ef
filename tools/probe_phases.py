import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, host, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
W = synth.make_window(cfg)
ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
ba = host.window_to_host_ba(ctx, W, levels=1); ba.set_param("iterations", 1); assert ba.run()
for _ in range(10): ctx.ba_iteration_async(1e-5)
ctx.sync()
out = np.zeros(128, np.int64)
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 1, None))
for _ in range(3): ctx.ba_iteration_async(1e-5)
ctx.sync()
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 0, out.ctypes.data_as(C.POINTER(C.c_longlong))))
def seg(name, a, b): print("%-34s %7.2f us" % (name, (out[b] - out[a]) * 0.01))
seg("acc pair: load+accumulate loop", 16, 17); seg("acc pair: wave reduce", 17, 18); seg("acc pair: fp64 stitch", 18, 19)
seg("solve: load+scale", 48, 49); seg("solve: factorization", 49, 50); seg("solve: forward subst", 50, 51); seg("solve: backward subst", 51, 52); seg("solve: write x", 52, 53)
seg("solve: total", 48, 53)
