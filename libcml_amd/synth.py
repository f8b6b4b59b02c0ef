"""Seeded synthetic windows for the hot path: procedural textured plane rendered into
N keyframes, active points with known inverse depth, residual lists as BA::addPoints/createResidual
(BA.cpp:336-415) would build them.  numpy only; used by tests/, bench.py and smoke().

The images are renderings of ONE textured plane seen from the N keyframe poses, so photometric
residuals are small at the true state and the optimiser has something real to do once poses / idepths
are perturbed.  Texture = 6 octaves of value noise + 40 random step edges, range [0,255].
"""
import numpy as np

STAR8 = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], np.int32)  # types.h:1381-1393

CONFIGS = {
    # name: (N, P, w, h, levels, fx, fy, cx, cy)
    "A": (2, 200, 640, 480, 2, 525.0, 525.0, 319.5, 239.5),
    "B": (8, 2000, 1241, 376, 4, 718.856, 718.856, 607.19 - 0.5, 185.22 - 0.5),
    "E": (20, 8000, 1920, 1080, 4, 1400.0, 1400.0, 959.5, 539.5),
    "tiny": (3, 64, 160, 120, 2, 140.0, 140.0, 79.5, 59.5),
    "small": (4, 300, 320, 240, 3, 260.0, 260.0, 159.5, 119.5),
    "medium": (5, 600, 480, 360, 4, 390.0, 390.0, 239.5, 179.5),
    "M2": (8, 2800, 640, 480, 3, 525.0, 525.0, 319.5, 239.5),
    "M3": (8, 3600, 640, 480, 3, 525.0, 525.0, 319.5, 239.5),
    "M": (8, 5000, 640, 480, 3, 525.0, 525.0, 319.5, 239.5),          # R = 35 000: between the small-window and the large-window regime of the residual kernel
}


def so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)


class Texture:
    """Continuous procedural texture tex(a, b) on plane coordinates (metres)."""

    def __init__(self, rng, scale=1.0, edge_width=0.0, octave_gain=0.5):
        self.edge_width = edge_width      # metres on the plane over which a step edge ramps (0 = a hard step)
        self.lat = [rng.uniform(0, 1, size=(64, 64)) for _ in range(6)]
        self.freq = [scale * 2 ** o for o in range(6)]
        self.amp = [octave_gain ** o for o in range(6)]      # 0.5: smooth value noise; larger: more of the amplitude in the fine octaves (stronger gradients)
        ang = rng.uniform(0, np.pi, size=40)
        self.en = np.stack([np.cos(ang), np.sin(ang)], 1)
        self.eo = rng.uniform(-12, 12, size=40)
        self.ea = rng.uniform(-0.12, 0.12, size=40)
        self.ebias = 0.5 * float(self.ea.sum())

    def __call__(self, a, b):
        v = np.zeros_like(a)
        for lat, f, amp in zip(self.lat, self.freq, self.amp):
            x = a * f + 1000.0
            y = b * f + 1000.0
            x0 = np.floor(x); y0 = np.floor(y)
            fx = x - x0; fy = y - y0
            sx = fx * fx * (3 - 2 * fx); sy = fy * fy * (3 - 2 * fy)
            i0 = x0.astype(np.int64) % 64; j0 = y0.astype(np.int64) % 64
            i1 = (i0 + 1) % 64; j1 = (j0 + 1) % 64
            v += amp * ((lat[j0, i0] * (1 - sx) + lat[j0, i1] * sx) * (1 - sy)
                        + (lat[j1, i0] * (1 - sx) + lat[j1, i1] * sx) * sy)
        v = v / sum(self.amp)
        for n, o, amp in zip(self.en, self.eo, self.ea):
            if self.edge_width > 0:       # band-limited edge: the renderings of one edge at different distances stay photometrically consistent
                v = v + amp * np.clip((a * n[0] + b * n[1] - o) / self.edge_width + 0.5, 0.0, 1.0)
            else:
                v = v + amp * (a * n[0] + b * n[1] > o)
        # fixed (view-independent) tone mapping: the same world point must get the same value in every keyframe
        v = np.clip(0.5 + 1.6 * (v - 0.5 - self.ebias), 0.0, 1.0)
        return (255.0 * v).astype(np.float32)


def render(tex, K, R, t, w, h, n, d):
    """Image of the plane {X: n.X = d} from the camera Xc = R Xw + t."""
    fx, fy, cx, cy = K
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    r = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
    Rt = R.T
    nR = Rt.T @ n                       # n^T R^T
    s = (d + n @ (Rt @ t)) / (r @ nR)   # depth along the ray (r_z = 1)
    Xw = (s[..., None] * r - t) @ Rt.T
    e1 = np.cross(n, [0, 1, 0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)
    return tex(Xw @ e1, Xw @ e2), s


class Window:
    """A synthetic sliding window: everything a DSOBundleAdjustment::run would see, as flat arrays."""
    pass


# Scene defaults per named configuration.  The BASELINE windows (B, E) model the window a keyframe's run() iterates on: points that
# are still visible in the newest keyframe, inverse depths as accurate as the activation leaves them, and band-limited texture edges —
# so that >= 90 % of the R = P (N-1) residuals are IN after the first pass (the headline must not be billed for
# residuals that gather nothing).  octave_gain 0.7 puts enough of the value noise into the fine octaves that a point's pattern keeps
# gradient when a later keyframe sees it magnified (with 0.5 one residual in ten was an OUTLIER by `wJI2_sum < 2`, BA.cpp:303, at
# the TRUE state).  The small test windows keep hard edges, smooth noise and every candidate point, OOB residuals included.
SCENE_DEFAULTS = {"pose_noise": 1.0, "idepth_noise": 0.03, "state_noise": 2e-3, "eval_noise": 1.0, "edge_px": 0.0, "covisible": False, "octave_gain": 0.5}
SCENES = {
    "B": {"idepth_noise": 0.005, "eval_noise": 0.3, "edge_px": 4.0, "covisible": True, "octave_gain": 0.7},
    "E": {"idepth_noise": 0.005, "eval_noise": 0.3, "edge_px": 4.0, "covisible": True, "octave_gain": 0.7},
}


def make_window(config="B", seed=0xC0FFEE, shard=0, pose_noise=None, idepth_noise=None, state_noise=None,
                eval_noise=None, edge_px=None, covisible=None, octave_gain=None):
    """covisible: keep a candidate point only if it projects inside the NEWEST keyframe (for points hosted there: inside at least one
    other keyframe) — the window the reference's policy leaves behind: points that left the newest frames are marginalised or dropped
    (isOOB / flagPointsForRemoval, BA.cpp:2240-2363), and the closing linearizeAll(true) of every run removes each residual that is
    not IN (BA.cpp:1595-1598, 1624-1638)."""
    N, P, w, h, levels, fx, fy, cx, cy = CONFIGS[config] if isinstance(config, str) else config
    scene = dict(SCENE_DEFAULTS, **(SCENES.get(config, {}) if isinstance(config, str) else {}))
    pose_noise = scene["pose_noise"] if pose_noise is None else pose_noise
    idepth_noise = scene["idepth_noise"] if idepth_noise is None else idepth_noise
    state_noise = scene["state_noise"] if state_noise is None else state_noise
    eval_noise = scene["eval_noise"] if eval_noise is None else eval_noise
    edge_px = scene["edge_px"] if edge_px is None else edge_px
    covisible = scene["covisible"] if covisible is None else covisible
    octave_gain = scene["octave_gain"] if octave_gain is None else octave_gain
    rng = np.random.default_rng(seed + shard)
    W = Window()
    W.config = config; W.N, W.P, W.w, W.h, W.levels = N, P, w, h, levels
    W.K = (fx, fy, cx, cy)
    n = np.array([0.12, -0.08, 1.0]); n /= np.linalg.norm(n)
    d = 9.0
    # highest octave (x32) at ~0.07 cycles/pixel at the plane distance, so the renderings are not aliased
    tex = Texture(rng, scale=0.07 * fx / (d * 32.0), edge_width=edge_px * d / fx, octave_gain=octave_gain)
    # true keyframe poses (world -> cam): forward motion 0.8 m / KF + jitter, SURVEY §8d
    W.R_true, W.t_true, W.aff_true, W.gray, W.depth = [], [], [], [], []
    step = 0.8 if N <= 8 else 0.3
    for k in range(N):
        c = np.array([0, 0, step * k]) + rng.uniform(-0.05, 0.05, 3) * pose_noise
        Rk = so3_exp(np.deg2rad(rng.uniform(-1, 1, 3)) * pose_noise)
        tk = -Rk @ c
        a = rng.uniform(-0.05, 0.05); b = rng.uniform(-5, 5)
        W.R_true.append(Rk); W.t_true.append(tk); W.aff_true.append((a, b))

    def _render(k):
        img, s = render(tex, W.K, W.R_true[k], W.t_true[k], w, h, n, d)
        # the photometric model of the reference: I_k = exp(a_k) * (I_true) + b_k  (exposure time 1)
        a, b = W.aff_true[k]
        return (np.exp(a) * img + b).astype(np.float32), s
    if N * w * h > (1 << 22):      # large windows: the renderings draw no random numbers, so they may run side by side (numpy releases the GIL)
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max_workers=min(8, N)) as ex:
            out = list(ex.map(_render, range(N)))
    else:
        out = [_render(k) for k in range(N)]
    for img, s in out:
        W.gray.append(img); W.depth.append(s)
    # evaluation-point poses = truth perturbed (what tracking would have delivered)
    W.R_eval, W.t_eval, W.aff_eval = [], [], []
    for k in range(N):
        dR = so3_exp(rng.normal(0, 0.0015, 3) * eval_noise) if k else np.eye(3)
        dt = rng.normal(0, 0.004, 3) * eval_noise if k else np.zeros(3)
        W.R_eval.append(dR @ W.R_true[k]); W.t_eval.append(dR @ W.t_true[k] + dt)
        a, b = W.aff_true[k]
        W.aff_eval.append((a + rng.normal(0, 0.01) * eval_noise, b + rng.normal(0, 0.5) * eval_noise))
    # states: older keyframes have drifted from their linearisation point, the newest has not
    W.state = np.zeros((N, 10))
    for k in range(1, N - 1):
        W.state[k, :6] = rng.normal(0, state_noise, 6)
        W.state[k, 6] = rng.normal(0, 2e-4); W.state[k, 7] = rng.normal(0, 2e-4)
    W.state_zero = np.zeros((N, 10))
    for k in range(N):   # setEvalPT_scaled: state_scaled[6:8] = (a, b) -> state = scaled / scale, state_zero = state
        a, b = W.aff_eval[k]
        W.state_zero[k, 6] = a / 10.0; W.state_zero[k, 7] = b / 1000.0
        W.state[k, 6] += W.state_zero[k, 6]; W.state[k, 7] += W.state_zero[k, 7]
    W.ab_exposure = np.ones(N)
    W.keyid = np.arange(N)
    W.frame_energy_th = np.full(N, 8.0 * 8 * 8, np.float32)   # DSOFrame.h:35
    # points: uniform pixels of a round-robin host, keep those seen by >= 1 other keyframe (truth geometry)
    pts = np.zeros(P, dtype=[("x", "f4"), ("y", "f4"), ("idepth", "f8"), ("idepth_true", "f8"), ("host", "i4")])
    k = 0
    tries = 0
    while k < P and tries < 50 * P:
        tries += 1
        hst = k % N
        # stand-in for the reference's PixelSelector (out of scope): best gradient of 12 random candidates
        cx_ = rng.integers(8, w - 8, size=12); cy_ = rng.integers(8, h - 8, size=12)
        g = W.gray[hst]
        mag = np.abs(g[cy_, cx_ + 1] - g[cy_, cx_ - 1]) + np.abs(g[cy_ + 1, cx_] - g[cy_ - 1, cx_])
        best = int(np.argmax(mag))
        x = np.float32(cx_[best]); y = np.float32(cy_[best])
        z = W.depth[hst][int(y), int(x)]
        idt = 1.0 / z
        ray = np.array([(x - cx) / fx, (y - cy) / fy, 1.0])
        Xw = W.R_true[hst].T @ (ray * z - W.t_true[hst])
        seen = 0
        seen_newest = hst == N - 1
        for t_ in range(N):
            if t_ == hst:
                continue
            Xc = W.R_true[t_] @ Xw + W.t_true[t_]
            if Xc[2] <= 0.1:
                continue
            u = fx * Xc[0] / Xc[2] + cx; v = fy * Xc[1] / Xc[2] + cy
            if 4 <= u < w - 4 and 4 <= v < h - 4:
                seen += 1
                seen_newest = seen_newest or t_ == N - 1
        if seen == 0 or (covisible and not seen_newest):
            continue
        pts[k] = (x, y, idt * (1 + rng.normal(0, idepth_noise)), idt, hst)
        k += 1
    assert k == P, "could not place the requested number of points"
    W.pts = pts
    return W


def point_colors_weights(W, grads0):
    """DSOContext::addPoint colours (gray at the integer pixel + shift, DSOContext.h:87-91 / MapObject.h:398-399)
    and BA::addPoints gradient weights sqrt(c / (c + |grad|^2)) (BA.cpp:405-411), c = 2500."""
    P = W.P
    colors = np.zeros((P, 8), np.float32)
    weights = np.zeros((P, 8), np.float32)
    for i in range(P):
        hst = W.pts["host"][i]
        x = int(W.pts["x"][i]); y = int(W.pts["y"][i])
        g = grads0[hst]
        for k, (dx, dy) in enumerate(STAR8):
            colors[i, k] = W.gray[hst][y + dy, x + dx]
            # corners are integer pixels here, so interpolate() reduces to the texel itself
            gx, gy = np.float64(g[y + dy, x + dx, 1]), np.float64(g[y + dy, x + dx, 2])
            weights[i, k] = np.float32(np.sqrt(2500.0 / (2500.0 + (gx * gx + gy * gy))))
    return colors, weights


def residual_list(W, R_eval, t_eval):
    """createResidual (BA.cpp:336-380): one residual per (point, target != host); initial state IN when the
    centre projects inside the image at the evaluation-point poses, else OOB."""
    fx, fy, cx, cy = W.K
    res = []
    for i in range(W.P):
        hst = int(W.pts["host"][i])
        x, y, idp = float(W.pts["x"][i]), float(W.pts["y"][i]), float(W.pts["idepth"][i])
        p = np.array([(x - cx) * (1.0 / fx), (y - cy) * (1.0 / fy), 1.0])
        for t_ in range(W.N):
            if t_ == hst:
                continue
            Rht = R_eval[t_] @ R_eval[hst].T
            tht = t_eval[t_] - Rht @ t_eval[hst]
            q = Rht @ p + tht * idp
            u = fx * q[0] / q[2] + cx; v = fy * q[1] / q[2] + cy
            inside = (0 <= u < W.w) and (0 <= v < W.h)   # Frame::isInside(p, 0, 0), src/cml/map/Frame.h:136-138
            res.append((i, t_, 0 if inside else 1, 0))
    return np.array(res, dtype=[("point", "i4"), ("target", "i4"), ("state", "i4"), ("is_linearized", "i4")])


def indirect_observations(W, n_obs=1000, n_pts=300, seed=3):
    """BASELINE.json configs[2] (hybrid path): n_obs ORB observations of n_pts 3-D points at depth U(2,20) m in front of keyframe 0,
    observed with 0.5 px Gaussian noise, one in fifty a gross outlier.  Returns (poses N x 12 at the evaluation
    point, points n_pts x 3 world XYZ, observations as a structured array {frame, point, gx, gy} in normalised image coordinates)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = W.K
    N = W.N
    poses = np.zeros((N, 12))
    for k in range(N):
        poses[k, :9] = W.R_eval[k].ravel(); poses[k, 9:] = W.t_eval[k]
    pts = np.zeros((n_pts, 3))
    for j in range(n_pts):
        z = rng.uniform(2, 20)
        u = rng.uniform(0, W.w); v = rng.uniform(0, W.h)
        Xc = np.array([(u - cx) / fx * z, (v - cy) / fy * z, z])
        pts[j] = W.R_true[0].T @ (Xc - W.t_true[0])
    obs = np.zeros(n_obs, dtype=[("frame", "i4"), ("point", "i4"), ("gx", "f8"), ("gy", "f8")])
    for k in range(n_obs):
        i = int(rng.integers(0, N)); j = int(rng.integers(0, n_pts))
        Xc = W.R_true[i] @ pts[j] + W.t_true[i]
        noise = rng.normal(0, 0.5, 2) / np.array([fx, fy])
        if k % 50 == 0:
            noise += 0.2      # gross outliers: beyond the Tukey threshold -> zero loss / zero Jacobian
        obs[k] = (i, j, Xc[0] / Xc[2] + noise[0], Xc[1] / Xc[2] + noise[1])
    return poses, pts, obs
