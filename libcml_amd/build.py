"""Build the gfx950 device layer (libcml_amd/libcmlhip.so) with hipcc.  No CUDA, no Triton, no JIT cache:
the .so is built in-tree so it travels with the repo snapshot to the GPU box."""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcmlhip.so")
HOSTLIB = os.path.join(HERE, "libcmlhost.so")
SOURCES = ["cmlhip_ctx.hip", "ba_linearize.hip", "ba_linearize_rs.hip", "ba_linearize_rs4.hip", "ba_accumulate.hip", "ba_api.hip", "tracker.hip", "tracker_opt.hip", "tracer.hip", "initializer.hip", "pnp.hip", "lba.hip", "reproj.hip"]
HEADERS = ["cmlhip_internal.h", "ba_common.h", "ba_finish.h", "ba_frames.h", "reproj_dev.h", "ba_linearize_rs_body.inc", os.path.join("..", "host", "se3.h"), os.path.join("..", "..", "include", "cmlhip.h")]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable", "-Wno-unused-value"] + os.environ.get("CML_HIPCC_EXTRA", "").split()   # e.g. -DCML_RS_STAMPS (tools/probe_rs_tiles.py)


def _stamp(paths):
    h = hashlib.sha1()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    stamp = obj + ".stamp"
    want = _stamp([os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS])
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        print(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libcmlhip.so (+ the C++ host mirror)."""
    if force:
        for s in SOURCES:
            for ext in (".o", ".o.stamp"):
                p = os.path.join(CSRC, s.replace(".hip", ext))
                if os.path.exists(p):
                    os.remove(p)
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(_compile, SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    build_host(force)
    _write_commit()
    if verbose:
        print("built", OUT)
    return OUT


def _write_commit():
    """the snapshot that travels to the GPU box has no .git: leave the commit the build was made from beside the libraries (bench.py / the profile
    tools stamp their records with it; `+dirty` when the tree had uncommitted changes)"""
    root = os.path.dirname(HERE)
    if not os.path.isdir(os.path.join(root, ".git")):
        return
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, cwd=root).stdout.strip()
        with open(os.path.join(HERE, "BUILD_COMMIT"), "w") as f:
            f.write(head + ("+dirty" if dirty else ""))
    except Exception:
        pass


def build_host(force=False):
    """C++ host mirror of the reference operator interface (libcml_amd/host), linked against libcmlhip.so."""
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, f) for f in sorted(os.listdir(hdir)) if f.endswith(".cpp")] if os.path.isdir(hdir) else []
    if not srcs:
        return None
    deps = srcs + [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".h")] + [OUT]
    if not force and os.path.exists(HOSTLIB) and all(os.path.getmtime(HOSTLIB) >= os.path.getmtime(d) for d in deps):
        return HOSTLIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOSTLIB] + srcs + \
          ["-I", os.path.join(HERE, "..", "include"), "-L", HERE, "-lcmlhip", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host build failed:\n%s\n%s" % (r.stdout, r.stderr))
    return HOSTLIB


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
