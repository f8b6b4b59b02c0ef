import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import device, host, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
W = synth.make_window(cfg)
ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
ba = host.window_to_host_ba(ctx, W, levels=1); ba.set_param("iterations", 1); assert ba.run(); assert ba.begin_resident()
for _ in range(10): ctx.ba_iteration_async(1e-5)
ctx.sync()
NS = 128 + 5 * 1024 * 2
out = np.zeros(NS, np.int64)
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 1, None))
# several iterations: the stamps kept are those of the LAST one, which runs with the queue ahead of the device (the first launches after
# the synchronising enable call wait for the host between kernels: 'first start' columns 10-20 us apart that no steady-state step shows)
for _ in range(6): ctx.ba_iteration_async(1e-5)
ctx.sync()
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 0, out.ctypes.data_as(C.POINTER(C.c_longlong))))
def seg(name, a, b):
    if out[a] <= 0 or out[b] <= 0:
        print("%-34s     n/a (not stamped on this path)" % name)
    else:
        print("%-34s %7.2f us" % (name, (out[b] - out[a]) * 0.01))
seg("acc pair: load+accumulate loop", 16, 17); seg("acc pair: tile sum + H", 17, 18); seg("acc pair: fp64 stitch", 18, 19)
seg("solve: load+scale (total)", 48, 49); seg("solve:   loads landed (wave 0)", 48, 54); seg("solve: factorization", 49, 50); seg("solve: forward subst", 50, 51); seg("solve: backward subst", 51, 52); seg("solve: write x", 52, 53)
seg("solve: total", 48, 53)
nbk = (4 + 8 * W.N - 4 + 15) // 16
def _bc(k):
    a, b, c_ = out[64 + 2 * k], out[65 + 2 * k], (out[66 + 2 * k] if k + 1 < nbk else out[50])
    if a <= 0 or b <= 0 or c_ <= 0:
        return "n/a"                                              # (the look-ahead factorisation of wide windows does not stamp its block columns)
    return "%.2f|%.2f" % ((b - a) * 0.01, (c_ - b) * 0.01)
print("solve per block column (elimination | trailing update) us:", " ".join(_bc(k) for k in range(nbk)))

names = ["linearize", "acc", "system", "solve", "backsub"]
blk = out[128:].reshape(5, 1024, 2)
# pipeline order inside one iteration: acc, system, solve, backsub, linearize
t0 = min(b[b[:, 0] > 0, 0].min() for b in blk if (b[:, 0] > 0).any())
for k in (1, 2, 3, 4, 0):
    b = blk[k]; b = b[(b[:, 0] > 0) & (b[:, 1] >= b[:, 0])]          # (blocks that only ride in a launch — hybrid-term / merged blocks — leave no end stamp)
    if len(b) == 0: continue
    st = (b[:, 0] - t0) * 0.01; en = (b[:, 1] - t0) * 0.01; d = en - st
    print("%-10s blocks %4d%s first start %7.2f  last start %7.2f  last end %7.2f | block us: min %6.2f med %6.2f max %6.2f" %
          (names[k], len(b), "+" if len(b) >= 1024 else " ", st.min(), st.max(), en.max(), d.min(), np.median(d), d.max()))
    if len(b) >= 1024:
        print("           (the stamp buffer holds the launch's first 1024 workgroups: `last end` is the last STAMPED block, not the end of the launch)")
    if k == 1:
        NN = W.N * W.N
        npt = len(d) - NN                                # the point-row workgroups come first in the grid
        print("   pair blocks: med %.2f max %.2f ; point blocks: med %.2f max %.2f" % (np.median(d[npt:]), d[npt:].max(), np.median(d[:npt]), d[:npt].max()))
b = blk[2]; nb_ = int((b[:, 0] > 0).sum()); d = (b[:nb_, 1] - b[:nb_, 0]) * 0.01
nrow = W.N + 1
print("system: SYRK blocks %d dur med %.2f max %.2f ; row blocks dur med %.2f max %.2f" % (nb_ - nrow, np.median(d[:-nrow]), d[:-nrow].max(), np.median(d[-nrow:]), d[-nrow:].max()))
if out[32] and out[34] and blk[2][33][0] > 0:
    b0 = blk[2][33][0]
    print("SYRK workgroup 33: operands of the last trip landed %.2f us after its first stamp | products done %.2f | barrier %.2f | end %.2f" % (
        (out[32] - b0) * 0.01, (out[33] - b0) * 0.01, (out[34] - b0) * 0.01, (blk[2][33][1] - b0) * 0.01))

if out[96] and out[100]:      # us behind the solve workgroup's last stamp (53)
    print("a point block of the merged launch: x in LDS %.2f us behind the solve workgroup's last stamp | x.adjoint table %.2f | point steps + stores %.2f | block partials %.2f" % (
        (out[97] - out[53]) * 0.01, (out[98] - out[97]) * 0.01, (out[99] - out[98]) * 0.01, (out[100] - out[99]) * 0.01))
if out[104]:
    print("frame-step block: x in LDS %.2f us behind the solve workgroup's last stamp | states stepped, exp %.2f | barrier %.2f | adHTd + pair records %.2f" % (
        (out[104] - out[53]) * 0.01, (out[105] - out[104]) * 0.01, (out[106] - out[105]) * 0.01, (out[107] - out[106]) * 0.01))
