"""Config E (BASELINE.json configs[4]) stores level-0 texels as halves (CMLHIP_TEXEL_F16); the reference samples fp32 `Vector3f` texels
(image/Array2D.h:265-286, DSOBundleAdjustment.cpp:214-271).  This module states how far the fp16-texel results sit from the results on the
UNROUNDED fp32 images: residual classes, energies, H_A / b_A / H_sc / b_sc (plain and Jacobi-scaled as BA.cpp:1312-1316), the gauge-free pose
update.  Checker side only (oracle + numpy); used by tests/test_config_e_gpu.py and bench.py's detail file."""
import ctypes as C

import numpy as np


def jacobi_scaled(H, b, lam=1e-5, H_extra=None):
    """S H S, S b with S = 1 / sqrt(diag(H_total) + 10) — the scaling the reference solves in (BA.cpp:1312-1316)"""
    Ht = H if H_extra is None else H_extra
    Sv = 1.0 / np.sqrt(np.abs(np.diag(Ht)) + 10.0)
    return Sv[:, None] * H * Sv[None, :], Sv * b


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def final_system(Hs, lam=1e-5):
    HA, bA, HL, bL, Hsc, bsc = Hs
    n = HA.shape[0]
    H = HL + HA
    H[np.diag_indices(n)] *= (1 + lam)
    H = H - Hsc / (1 + lam)
    return H, bL + bA - bsc


def gauge_free_update(I, Hs, lam=1e-5):
    """independent fp64 solve of the Jacobi-scaled final system, frame part, gauge removed (orthogonalize, BA.cpp:1196-1261)"""
    from tests import oracle_lib as O
    H, b = final_system(Hs, lam)
    n = H.shape[0]
    Sv = 1.0 / np.sqrt(np.diag(H) + 10.0)
    x = np.zeros(n)
    x[4:] = Sv[4:] * np.linalg.solve(Sv[4:, None] * H[4:, 4:] * Sv[None, 4:], Sv[4:] * b[4:])
    ns = np.zeros(7 * n)
    O.lib().orc_ba_nullspaces(I.frames, I.N, C.byref(I.scales), O.ptr(ns, C.c_double))
    return O.orthogonalize(x, ns.reshape(7, n), 1e-5)


def compare(I, st16, Hs16, st32, Hs32, lam=1e-5):
    """st*: dicts of per-residual arrays after linearize + applyRes (new_state / state / good / new_energy / energy);  Hs*: the six accumulated
    matrices.  *16 = fp16-texel run (device, or the oracle on rounded images: the two are bit-identical per residual), *32 = oracle on the
    unrounded fp32 images."""
    R = len(st32["state"])
    out = {"R": R}
    out["class_flips"] = int((st16["state"] != st32["state"]).sum())
    out["class_flips_frac"] = out["class_flips"] / max(R, 1)
    out["good_flips"] = int((st16["good"] != st32["good"]).sum())
    both = (st16["state"] == 0) & (st32["state"] == 0)
    e16, e32 = st16["energy"][both].astype(np.float64), st32["energy"][both].astype(np.float64)
    d = np.abs(e16 - e32) / np.maximum(np.abs(e32), 1.0)
    out["energy_rel_median"] = float(np.median(d)); out["energy_rel_p99"] = float(np.percentile(d, 99)); out["energy_rel_max"] = float(d.max())
    out["total_energy_rel"] = float(abs(e16.sum() - e32.sum()) / e32.sum())
    names = ("H_A", "b_A", "H_L", "b_L", "H_sc", "b_sc")
    for k in (0, 4):
        out[names[k] + "_rel"] = rel(Hs16[k], Hs32[k]); out[names[k + 1] + "_rel"] = rel(Hs16[k + 1], Hs32[k + 1])
    # Jacobi-scaled, as the solve sees them (north_star's 1e-3 reference point): scaled by the diagonal of the fp32-texel final system
    H32, b32 = final_system(Hs32, lam)
    H16, b16 = final_system(Hs16, lam)
    for nm, k in (("H_A", 0), ("H_sc", 4)):
        a, _ = jacobi_scaled(Hs16[k][4:, 4:], Hs16[k + 1][4:], H_extra=H32[4:, 4:])
        b, _ = jacobi_scaled(Hs32[k][4:, 4:], Hs32[k + 1][4:], H_extra=H32[4:, 4:])
        out[nm + "_jacobi_rel"] = rel(a, b)
    a, ab = jacobi_scaled(H16[4:, 4:], b16[4:], H_extra=H32[4:, 4:])
    b, bb = jacobi_scaled(H32[4:, 4:], b32[4:], H_extra=H32[4:, 4:])
    out["H_final_jacobi_rel"] = rel(a, b); out["b_final_jacobi_rel"] = rel(ab, bb)
    x16, x32 = gauge_free_update(I, Hs16, lam), gauge_free_update(I, Hs32, lam)
    out["x_gauge_free_rel"] = float(np.abs(x16[4:] - x32[4:]).max() / max(np.abs(x32[4:]).max(), 1e-300))
    out["x_gauge_free_rel_l2"] = float(np.linalg.norm(x16[4:] - x32[4:]) / max(np.linalg.norm(x32[4:]), 1e-300))
    return out


def oracle_side(I):
    """linearize + applyRes + accumulate on the oracle with the images I carries"""
    from tests import ba_setup as S
    ob = S.OracleBA(I)
    ob.linearize(); ob.apply(1)
    return ob.states(), ob.accumulate()


def device_vs_fp32_oracle(W, device_id=0):
    """The whole statement for one config-E window `W` (libcml_amd.synth): a device context with fp16 texels (first linearisation + applyRes +
    accumulation through the C ABI) against the oracle on the unrounded fp32 images of the same window."""
    from libcml_amd import abi
    from tests import ba_setup as S
    from tests import dev_setup as D
    I = S.make_inputs(W=W)
    st32, Hs32 = oracle_side(I)
    fp32 = [I.grads[k][0] for k in range(I.N)]
    for k in range(I.N):
        I.grads[k][0] = fp32[k].astype(np.float16).astype(np.float32)          # what the device stores (pyramid_put rounds the same way)
    ctx = D.make_ctx(I, texel_format=abi.TEXEL_F16, device_id=device_id)
    try:
        ctx.ba_linearize(); ctx.ba_apply(1)
        rep = compare(I, ctx.ba_states(), D.accumulate(ctx, I), st32, Hs32)
    finally:
        ctx.close()
    rep["note"] = ("fp16-texel device run (first linearisation + applyRes + accumulation of this window through the C ABI) against oracle/orc_ba.c on the "
                   "UNROUNDED fp32 images: residuals classified differently, per-residual energies, matrices relative to their largest entry and "
                   "Jacobi-scaled (BA.cpp:1312-1316), gauge-free pose update by an independent fp64 solve; bars in tests/test_config_e_gpu.py")
    return rep
