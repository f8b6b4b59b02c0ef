"""Drive the DEVICE path (through the C ABI) from the same BAInputs the oracle gets."""
import numpy as np

from libcml_amd import abi, device


def make_ctx(I, texel_format=abi.TEXEL_F32, device_id=0):
    ctx = device.Ctx(device_id=device_id, max_frames=max(I.N, 2), max_points=max(I.P, 1), max_residuals=max(I.R, 1),
                     texel_format=texel_format)
    for k in range(I.N):
        ctx.pyramid_put(int(I.frames_dev["image_id"][k]), 0, I.grads[k][0])
    ctx.ba_set_params(I.prm)
    ctx.ba_upload_window(I.frames_dev, I.points, I.residuals)
    ctx.ba_set_pairs(I.pairs)
    return ctx


def accumulate(ctx, I):
    return ctx.ba_accumulate(I.adH, I.adT, I.adHTd, I.cdelta, I.prior, I.dprior, I.cprior)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return d / s
