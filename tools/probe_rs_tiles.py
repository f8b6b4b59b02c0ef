"""Per-tile schedule of the lane-per-residual kernel (k_ba_lin_rs) at a large window: begin / loads-issued / end stamps and the
hardware slot of every wave (development stamps behind cmlhip_debug_timestamps, dumped through CMLHIP_RS_TS_FILE).
    python tools/probe_rs_tiles.py [E]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
TS = "/tmp/rs_ts.bin"
os.environ["CMLHIP_RS_TS_FILE"] = TS
from libcml_amd import device, host, synth, abi
cfg = sys.argv[1] if len(sys.argv) > 1 else "E"
W = synth.make_window(cfg, seed=0xC0FFEE)
ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N, texel_format=abi.TEXEL_F16 if cfg == "E" else abi.TEXEL_F32)
ba = host.window_to_host_ba(ctx, W, levels=1); ba.set_param("iterations", 1); assert ba.run(); assert ba.begin_resident()
for _ in range(300): ctx.ba_iteration_async(1e-5)
ctx.sync()
NS = 128 + 5 * 1024 * 2
out = np.zeros(NS, np.int64)
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 1, None))
ctx.ba_iteration_async(1e-5)
ctx.sync()
ctx.ck(ctx.L.cmlhip_debug_timestamps(ctx.h, 0, out.ctypes.data_as(C.POINTER(C.c_longlong))))
t = np.fromfile(TS, np.int64).reshape(-1, 8)
t = t[t[:, 0] > 0]
print("tiles stamped:", len(t), "(needs a build with CML_HIPCC_EXTRA=-DCML_RS_STAMPS)")
t0 = t[:, 0].min()
T = (t[:, :7] - t0) * 0.01
b, e = T[:, 0], T[:, 6]
hw = t[:, 7] & 0xFFFFFFFF; xcc = (t[:, 7] >> 32) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("begin: min %.2f med %.2f p90 %.2f max %.2f us" % (b.min(), np.median(b), np.percentile(b, 90), b.max()))
print("end:   min %.2f med %.2f p90 %.2f max %.2f us" % (e.min(), np.median(e), np.percentile(e, 90), e.max()))
d = e - b
print("wave duration: min %.2f med %.2f p90 %.2f max %.2f us" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
names = ["pair record -> inputs arrived", "projection (+ texel loads issued)", "geometry -> staged", "pixel loop (texel wait + sums)", "classification + state stores", "reduced record + matrix-core tile + partials"]
if cfg != "E":      # small windows: k_ba_lin_rs4 (stamp 6 is taken before the partials)
    names = ["pair record -> inputs arrived", "projection, texel loads issued", "geometry (under the texel round trip)", "texel wait, photometric terms, exchange, pattern sums", "classification + state stores", "staging, JpJdF, matrix-core tile"]
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
slot = key * 4 + simd
u, cnt = np.unique(slot, return_counts=True)
for n in (1, 2, 3):
    sel = np.isin(slot, u[cnt == n])
    if not sel.any(): continue
    print("SIMDs with %d waves (%d SIMDs): wave duration med %.2f us, last end med %.2f us; phases (median us):" % (n, (cnt == n).sum(), np.median(d[sel]), np.median([e[slot == s_].max() for s_ in u[cnt == n]])))
    for i, nm in enumerate(names):
        ph = T[sel, i + 1] - T[sel, i]
        print("    %-46s %6.2f  (p90 %.2f)" % (nm, np.median(ph), np.percentile(ph, 90)))
print("SIMDs used: %d ; waves per SIMD histogram:" % len(u), dict(zip(*[x.tolist() for x in np.unique(cnt, return_counts=True)])))
uc, ccnt = np.unique(key, return_counts=True)
print("CUs used: %d ; waves per CU histogram:" % len(uc), dict(zip(*[x.tolist() for x in np.unique(ccnt, return_counts=True)])))
print("end time of every 256th tile by end order:    ", " ".join("%.1f" % np.sort(e)[i] for i in range(0, len(e), 256)))
