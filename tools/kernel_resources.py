"""Per-kernel register / LDS / scratch usage of the built gfx950 objects (amdhsa metadata notes)."""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objs = sys.argv[1:] or [os.path.join(ROOT, "libcml_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "libcml_amd", "csrc"))) if f.endswith(".o")]
BIN = "/opt/rocm/lib/llvm/bin/"
for o in objs:
    d = tempfile.mkdtemp()
    t = os.path.join(d, os.path.basename(o))
    shutil.copy(o, t)
    subprocess.run([BIN + "llvm-objdump", "--offloading", t], capture_output=True, cwd=d)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    if not co:
        continue
    txt = subprocess.run([BIN + "llvm-readelf", "--notes", os.path.join(d, co[0])], capture_output=True, text=True).stdout
    for blk in txt.split(".agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()[:64]
        print("%-66s vgpr %4s agpr %3s sgpr %4s lds %6s scratch %5s" % (name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    shutil.rmtree(d)
