"""How much of the GPU does one sliding window use?  S independent windows (shards) iterate concurrently on one GPU, one
context = one HIP stream each, launched round-robin from one host thread.  Not the headline metric (that is one window)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libcml_amd import device, host, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
steps = 300
for S in (1, 2, 4, 8):
    ctxs, bas, R = [], [], 0
    for s in range(S):
        W = synth.make_window(cfg, shard=s)
        ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
        ba = host.window_to_host_ba(ctx, W, image_id_base=1000 * (s + 1), levels=1)
        ba.set_param("iterations", 1)
        assert ba.run() and ba.begin_resident()
        R = ctx.refresh_window_size()[2]
        ctxs.append(ctx); bas.append(ba)
    for _ in range(100):
        for c in ctxs: c.ba_iteration_async(1e-5)
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs: c.ba_iteration_async(1e-5)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    print("%d concurrent windows: %.1f us per round of %d iterations, aggregate %.3g point-residuals/s" % (S, dt / steps * 1e6, S, S * R * steps / dt))
    for b in bas: b.close()
    for c in ctxs: c.close()
