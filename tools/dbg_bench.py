import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def P(*a): print(*a, flush=True)
if "--torch" in sys.argv:
    import torch; P("torch", torch.cuda.is_available()); torch.cuda.set_device(0)
from libcml_amd import device, host, synth
import ctypes as C
W = synth.make_window(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "small"); P("window")
ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N); P("ctx")
ba = host.window_to_host_ba(ctx, W, levels=1); P("ba built")
ba.set_param("iterations", 1)
P("run", ba.run(), ba.last_error())
ctx.ba_iteration_async(1e-5); ctx.sync(); P("iter ok")
L = device.lib()
L.cmlhip_profile_enable.argtypes = [C.c_void_p, C.c_int]
L.cmlhip_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
P("enable", L.cmlhip_profile_enable(ctx.h, 5))
for _ in range(5): ctx.ba_iteration_async(1e-5)
a, b, n = C.c_float(), C.c_float(), C.c_int()
P("read", L.cmlhip_profile_read(ctx.h, C.byref(a), C.byref(b), C.byref(n)), a.value, b.value, n.value)
ba.close(); ctx.close(); P("closed")
