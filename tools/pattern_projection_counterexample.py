"""Can the 8 pattern pixels of a residual be projected from the CENTRE projection?

The reference projects every pattern pixel on its own (BA.cpp:193-212):   q_k = R * Kinv((x, y) + d_k, 1) + t * idepth,  Ku_k = fx * q_k.x / q_k.z + cx
and the kernels reproduce those bits.  The cheaper form derives q_k from the centre:   q_k' = q_c + R[:, 0] * (d_k.x / fx) + R[:, 1] * (d_k.y / fy)
which is the same number in exact arithmetic but rounds differently: q_k' is a few ulp(double) away from q_k, Ku_k' ~1e-13 away from Ku_k.
The kernel then takes (float)Ku_k (BA.cpp:205, `Vector2f`): the casts differ whenever a float rounding boundary lies between the two doubles
— about once in 1e8..1e9 pixels, i.e. once in a few hundred passes over the config-E window (1.2 M pixel projections per pass).  This script
searches random windows for such a pixel and prints it: a concrete input on which the shortcut changes the sampled position by one float
ulp, hence the interpolation weights, hence the bits of the residual — "bit-exact against the oracle" would not survive it."""
import sys
import numpy as np

rng = np.random.default_rng(12345)
fx = fy = 1400.0; cx, cy = 959.5, 539.5
fxi, fyi = 1.0 / fx, 1.0 / fy
STAR8 = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], np.float64)
N = 2_000_000
tried = 0
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000_000
while tried < budget:
    # one random relative pose per batch (as one (host, target) pair), random integer pixels and inverse depths
    w = rng.normal(0, 0.02, 3); th = np.linalg.norm(w); Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * (Kx @ Kx)
    t = rng.normal(0, 0.15, 3)
    x = rng.integers(8, 1912, N).astype(np.float64); y = rng.integers(8, 1072, N).astype(np.float64)
    rho = rng.uniform(0.05, 0.5, N)
    tid = [t[0] * rho, t[1] * rho, t[2] * rho]
    qx_c, qy_c = (x - cx) * fxi, (y - cy) * fyi
    pc = [(R[i, 0] * qx_c + R[i, 1] * qy_c + R[i, 2] * 1.0) + tid[i] for i in range(3)]       # the expression shape of the kernels (RS_PROJ)
    for k in (0, 1, 2, 3, 5, 6, 7):
        dx, dy = STAR8[k]
        qx, qy = (x + dx - cx) * fxi, (y + dy - cy) * fyi
        p = [(R[i, 0] * qx + R[i, 1] * qy + R[i, 2] * 1.0) + tid[i] for i in range(3)]
        pd = [pc[i] + (R[i, 0] * (dx * fxi) + R[i, 1] * (dy * fyi)) for i in range(3)]         # derived from the centre
        ku = (p[0] / p[2]) * fx + cx; kud = (pd[0] / pd[2]) * fx + cx
        kv = (p[1] / p[2]) * fy + cy
        inside = (ku >= 2) & (kv >= 2) & (ku < 1920 - 2) & (kv < 1080 - 2)          # only pixels the kernel samples (BA.cpp:204)
        bad = np.nonzero((ku.astype(np.float32) != kud.astype(np.float32)) & inside)[0]
        tried += N
        if len(bad):
            i = bad[0]
            print("counter-example after %d pixel projections:" % tried)
            print("  pattern pixel %d offset (%g, %g), corner (%g, %g), idepth %.17g" % (k, dx, dy, x[i], y[i], rho[i]))
            print("  R =", np.array2string(R.ravel(), precision=17), " t =", np.array2string(t, precision=17))
            print("  Ku per-pixel  %.17g -> float %.9g" % (ku[i], np.float32(ku[i])))
            print("  Ku derived    %.17g -> float %.9g" % (kud[i], np.float32(kud[i])))
            print("  doubles differ by %.3g, floats by one ulp (%.3g)" % (abs(ku[i] - kud[i]), abs(float(np.float32(ku[i])) - float(np.float32(kud[i])))))
            sys.exit(0)
print("none found in %d pixel projections" % tried)
