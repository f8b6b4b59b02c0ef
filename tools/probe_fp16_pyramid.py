"""Device-built fp16 level 0 (cmlhip_pyramid_build on a CMLHIP_TEXEL_F16 context) against the oracle's fp32 pyramid rounded to half."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device, synth
from tests import oracle_lib as O
W = synth.make_window((2, 10, 1920, 1080, 1, 1400.0, 1400.0, 959.5, 539.5), octave_gain=0.7, edge_px=4.0)
ctx = device.Ctx(max_frames=2, texel_format=abi.TEXEL_F16)
ctx.pyramid_build(5, W.gray[0], 1)
d = ctx.pyramid_get(5, 0)
g, gr = O.build_pyramid(W.gray[0], 1)
o = gr[0].astype(np.float16).astype(np.float32)
bad = np.argwhere(d != o)
print("texels differing:", len(bad), "of", d.size)
for b in bad[:10]:
    y, x, c = b
    print(b, "device", d[y, x, c], "oracle-rounded", o[y, x, c], "oracle fp32", gr[0][y, x, c], "gray nb", W.gray[0][y, max(x-1,0):x+2])
