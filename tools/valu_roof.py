#!/usr/bin/env python3
"""Run on the GPU box (through gpurun): the inputs of bench.py's second roofline entry for the residual kernel, {"bound": "valu_issue"}.

    python tools/valu_roof.py <config B|C|E>

One `rocprofv3 --pmc <counter>` pass of the bench command per counter (own runs, no traces): SQ_WAVES, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU,
SQ_ACTIVE_INST_ANY, SQ_INSTS_SALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, averaged over the launches of the resident residual kernel.  Derived:
  valu_insts_per_wave  = SQ_INSTS_VALU / SQ_WAVES
  cycles_per_valu_inst = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU        (the SQ_ACTIVE_* counters tick in quad-cycles)
Beside them the static count from the code object (llvm-objdump -d): vector instructions of the kernel by class, fp64 share.
Writes gpurun_out/round6_valu_roof_<config>.json; copy into profiles/."""
import csv, glob, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
KERN = "k_ba_lin_rs"
bench_args = ["--config", cfg, "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-extras"]
env = dict(os.environ, TMPDIR="/tmp")
os.makedirs(OUT, exist_ok=True)
res = {"config": cfg, "bench_args": bench_args, "counters": {}}
for ctr in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
    d = os.path.join(OUT, "prof_valu_%s_%s" % (cfg, ctr.lower()))
    cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=900)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    vals = [float(r["Counter_Value"]) for r in rows if KERN in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr]
    names = sorted({r.get("Kernel_Name", "")[:96] for r in rows if KERN in r.get("Kernel_Name", "")})
    res["counters"][ctr] = {"avg": sum(vals) / len(vals) if vals else None, "launches": len(vals)}
    res["kernel"] = names[0] if names else res.get("kernel")
    shutil.rmtree(d, ignore_errors=True)
c = {k: v["avg"] for k, v in res["counters"].items()}
if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
    res["waves_per_launch"] = c["SQ_WAVES"]
    res["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    act = c.get("SQ_ACTIVE_INST_VALU")
    if act:
        res["cycles_per_valu_inst"] = 4.0 * act / c["SQ_INSTS_VALU"]
        res["cycles_source"] = "4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU"
    elif c.get("SQ_ACTIVE_INST_ANY"):
        res["cycles_per_valu_inst"] = 4.0 * c["SQ_ACTIVE_INST_ANY"] / (c["SQ_INSTS_VALU"] + (c.get("SQ_INSTS_SALU") or 0))
        res["cycles_source"] = "4 x SQ_ACTIVE_INST_ANY / (SQ_INSTS_VALU + SQ_INSTS_SALU)"
res["simds"] = 1024            # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
res["clock_ghz"] = 2.4
# static ISA count of the kernel the launch used
def static_isa(cfg, profiled_kernel):
    """vector / scalar / memory instruction counts of ONE instantiation — the one whose demangled name equals the profiled kernel's (up to its argument list)"""
    BIN = "/opt/rocm/lib/llvm/bin/"
    o = os.path.join(ROOT, "libcml_amd", "csrc", "ba_linearize_rs.o" if cfg == "E" else "ba_linearize_rs4.o")
    d = tempfile.mkdtemp()
    try:
        t = os.path.join(d, os.path.basename(o)); shutil.copy(o, t)
        subprocess.run([BIN + "llvm-objdump", "--offloading", t], capture_output=True, cwd=d)
        co = [f for f in os.listdir(d) if "amdgcn" in f]
        dis = subprocess.run([BIN + "llvm-objdump", "-d", "-C", os.path.join(d, co[0])], capture_output=True, text=True).stdout      # -C: demangled symbols, comparable with the profiler's kernel names
        want = "k_ba_lin_rs" if cfg == "E" else "k_ba_lin_rs4_2d"
        prof = (profiled_kernel or "").replace("void ", "").split("(")[0].replace(" ", "")
        best = None
        for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            m = re.match(r"[0-9a-f]+ <(.+)>:\s*$", blk.splitlines()[0]) if blk else None      # (demangled names contain '>': take everything up to the closing '>:')
            if not m or want not in m.group(1) or "batch" in m.group(1):
                continue
            sym = m.group(1).replace("void ", "").split("(")[0].replace(" ", "")
            if prof and sym != prof:
                continue
            ins = [ln.split("\t")[1].split()[0] for ln in blk.splitlines()[1:] if "\t" in ln and len(ln.split("\t")) > 1 and ln.split("\t")[1].strip()]
            valu = [i for i in ins if i.startswith("v_")]
            f64 = [i for i in valu if "f64" in i]
            cand = {"symbol": m.group(1)[:96], "matches_profiled_kernel": bool(prof), "instructions": len(ins), "valu": len(valu), "valu_f64": len(f64), "salu": len([i for i in ins if i.startswith("s_")]),
                    "mfma": len([i for i in valu if "mfma" in i]), "vmem": len([i for i in ins if i.startswith(("global_", "buffer_", "flat_"))]), "lds": len([i for i in ins if i.startswith("ds_")])}
            if best is None or cand["instructions"] > best["instructions"]:
                best = cand
        return best
    finally:
        shutil.rmtree(d, ignore_errors=True)


try:
    res["static_isa"] = static_isa(cfg, res.get("kernel"))
    if res["static_isa"] and res["static_isa"]["valu"]:
        res["fp64_share"] = res["static_isa"]["valu_f64"] / res["static_isa"]["valu"]
except Exception as e:
    res["static_isa"] = {"error": repr(e)}
try:
    res["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or "worktree"
except Exception:
    res["commit"] = "worktree"
json.dump(res, open(os.path.join(OUT, "round6_valu_roof_%s.json" % cfg), "w"), indent=1)
print(json.dumps(res))
