"""Gather roof of the config-E residual kernel: dump the texel-load addresses of one REAL config-E pass and replay
only those loads (tools/microbench8.hip).  Runs on the GPU box:

    python tools/gather_roof.py [out.json]

1. bench.py's config-E set-up (20 keyframes x 8000 points, R = 152 000, tiled fp16 level 0) is stepped to the state the timed region sees;
2. the device's pair records, inverse depths and residual states are read back through the ABI and the sixteen 16-byte loads of every residual
   (8 pattern pixels x 2 bilinear rows, RsRow<true>::load of ba_linearize_rs.hip) are restated in numpy — same fp64 projection, same clamping of
   the lanes that do not sample to texel 0 — in the device's residual order (pair-sorted, tiles of 64 within a pair);
3. microbench8 replays them in the kernel's launch shape and at 8 waves / SIMD and writes the figures; this script adds the product kernel's
   own dispatch time on the same box and its VALU-only floor."""
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OX = np.array([0, -1, 1, -2, 0, 2, -1, 0], np.float64)      # star8, types.h:1381-1393
OY = np.array([-2, -1, -1, 0, 0, 0, 1, 2], np.float64)


def dump(path):
    import bench
    from libcml_amd import synth
    S = bench.setup_window("E", 0xC0FFEE, 0, 0)
    ctx, W = S["ctx"], S["W"]
    for _ in range(300):
        ctx.ba_iteration_async(1e-5)
    ctx.sync()
    # the product kernel's own time on this box, this state (events on every residual-kernel dispatch)
    ctx.profile_stride(1); ctx.profile_select(1); ctx.profile_enable(50)
    for _ in range(50):
        ctx.ba_iteration_async(1e-5)
    ctx.sync()
    lin_ms, _a, _b, n = ctx.profile_read()
    pairs, _th, _b0 = ctx.ba_pairs()
    idepth = ctx.ba_get_idepth()
    st = ctx.ba_states()["state"]
    maps = ctx.ba_index_maps()
    res = synth.residual_list(W, W.R_eval, W.t_eval)
    N = W.N
    fx, fy, cx, cy = [float(v) for v in W.K]
    fxi, fyi = 1.0 / fx, 1.0 / fy
    w, h = W.w, W.h
    tw = (w + 3) // 4
    frame_bytes = tw * ((h + 3) // 4) * 128
    order = maps["by_pair"]                      # caller indices in the device's residual order
    off = maps["by_pair_off"]
    tiles_frame, tile_rows = [], []
    pt = res["point"][order]; tg = res["target"][order]
    host = W.pts["host"][pt]
    pr = pairs[host + tg * N]
    x = W.pts["x"][pt].astype(np.float64); y = W.pts["y"][pt].astype(np.float64)
    idp = idepth[pt]
    run = st[order] != 1
    R_ = pr["R"]; t_ = pr["t"]
    offs = np.zeros((len(order), 16), np.uint32)
    kx = np.zeros((8, len(order))); ky = np.zeros((8, len(order)))
    inb = np.zeros((8, len(order)), bool)
    with np.errstate(all="ignore"):
        for k in range(8):
            qx = (x + OX[k] - cx) * fxi; qy = (y + OY[k] - cy) * fyi
            ppx = (R_[:, 0] * qx + R_[:, 1] * qy + R_[:, 2]) + t_[:, 0] * idp
            ppy = (R_[:, 3] * qx + R_[:, 4] * qy + R_[:, 5]) + t_[:, 1] * idp
            ppz = (R_[:, 6] * qx + R_[:, 7] * qy + R_[:, 8]) + t_[:, 2] * idp
            kx[k] = ppx / ppz * fx + cx; ky[k] = ppy / ppz * fy + cy
            inb[k] = (kx[k] >= 2) & (ky[k] >= 2) & (kx[k] < w - 2) & (ky[k] < h - 2)
    centre = inb[4]
    for k in range(8):
        smp = run & centre & inb[k]
        ix = np.where(smp, kx[k].astype(np.float32).astype(np.int64), 0); iy = np.where(smp, ky[k].astype(np.float32).astype(np.int64), 0)
        for rr in range(2):
            row = iy + rr
            o = (ix & 3) * 6
            byte = ((row >> 2) * tw + (ix >> 2)) * 128 + (row & 3) * 32 + (o & ~3)
            offs[:, 2 * k + rr] = np.where(smp, byte, 0 if rr == 0 else 32).astype(np.uint32)      # a lane that does not sample reads texel (0,0) / (0,1)
    # non-sampling marker for the line census of microbench8: offset 0 (row 1 of a clamped lane reads byte 32 of line 0: counted as a load of line 0, harmless)
    tf, blocks = [], []
    for p in range(N * N):
        a, b = int(off[p]), int(off[p + 1])
        for s in range(a, b, 64):
            e = min(s + 64, b)
            blk = np.zeros((64, 16), np.uint32)
            blk[:e - s] = offs[s:e]
            blocks.append(blk); tf.append(p // N)            # pair index = host + target * N -> target = p // N
    ntiles = len(tf)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiiQ", 0x47415448, ntiles, N, 0, frame_bytes))
        f.write(np.asarray(tf, np.int32).tobytes())
        f.write(np.concatenate(blocks).astype(np.uint32).tobytes())
    info = {"ntiles": ntiles, "R": int(len(order)), "sampling_residuals": int((run & centre).sum()), "product_kernel_us": 1e3 * lin_ms, "product_kernel_samples": int(n)}
    # the kernel with every tap at texel 0 / one cached line per lane is the development switch CMLHIP_RS_DBG (read once per process): run by the caller
    S["ba"].close(); ctx.close()
    return info


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gather_roof_E.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    binp = "/tmp/gather_E.bin"
    info = dump(binp)
    mb = os.path.join(ROOT, "tools", "mb8.bin")
    if not os.path.exists(mb):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "microbench8.hip"), "-o", mb])
    tmp = "/tmp/mb8.json"
    r = subprocess.run([mb, binp, tmp], capture_output=True, text=True)
    print(r.stdout); print(r.stderr, file=sys.stderr)
    d = json.load(open(tmp))
    d.update(info)
    # the product kernel with its texel taps forced to one L1-resident line per lane (CMLHIP_RS_DBG=4): everything but the gather
    env = dict(os.environ, CMLHIP_RS_DBG="4")
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "S = bench.setup_window('E', 0xC0FFEE, 0, 0); c = S['ctx']\n"
            "[c.ba_iteration_async(1e-5) for _ in range(300)]; c.sync()\n"
            "c.profile_stride(1); c.profile_select(1); c.profile_enable(50)\n"
            "[c.ba_iteration_async(1e-5) for _ in range(50)]; c.sync()\n"
            "print('K1CACHED', 1e3 * c.profile_read()[0])\n" % ROOT)
    r2 = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    for l in r2.stdout.splitlines():
        if l.startswith("K1CACHED"):
            d["product_kernel_cached_taps_us"] = float(l.split()[1])
    d["note"] = ("variants: the sixteen 16-byte loads per residual of one real config-E pass, nothing else; shape3 = the product kernel's launch shape (three waves per SIMD), "
                 "shape8 = eight; ALL = sixteen loads in flight, PIPE3 = three pixels in flight (the product's pipeline), PIPE1 = one pixel at a time.  "
                 "product_kernel_us = k_ba_lin_rs on the same box and state; product_kernel_cached_taps_us = the same kernel with every tap on one L1-resident line per lane "
                 "(CMLHIP_RS_DBG=4: its arithmetic, inputs and stores without the gather)")
    try:
        d["commit"] = open(os.path.join(ROOT, "libcml_amd", "BUILD_COMMIT")).read().strip()
    except Exception:
        pass
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
