#!/usr/bin/env python3
"""After `gpurun -- bash tools/refresh_profiles_r6.sh`: copy gpurun_out/refresh6/* into profiles/round6_* — every record stamped with the commit the
measured build was made from (libcml_amd/BUILD_COMMIT as it travelled to the GPU box; must equal HEAD) — and print the figures the documents quote."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out") + "/"
R = G + "refresh6/"
P = os.path.join(ROOT, "profiles") + "/"
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
built = open(R + "BUILD_COMMIT").read().strip() if os.path.exists(R + "BUILD_COMMIT") else "?"
print("HEAD", head, "| measured build", built, "" if built == head else "  <-- NOT the commit at HEAD: rebuild, commit, measure again")
for c in "BCE":
    shutil.copy(G + "refresh6/prof_%s_kernels.md" % c, P + "round6_kernels_%s.md" % c)
    d = json.load(open(G + "refresh6/prof_%s_pmc.json" % c))
    d["commit"] = built
    d["note"] = ("rocprofv3 passes of the bench command at this commit (tools/profile_bench.py): --kernel-trace --stats (linearize_avg_us), --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate runs; "
                 "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (calibrated for streaming 16-B reads: an upper bound for this gather); launch_us of the bench line UNDER the profiler is "
                 "inflated by the profiler itself, the unprofiled bench line (profiles/round6_bench_default.json) agrees with linearize_avg_us")
    json.dump(d, open(P + "round6_pmc_linearize_%s.json" % c, "w"), indent=1)
    print(c, "rocprof K1 %.2f us, fetch raw %.2f MB, write %.2f MB" % (d["linearize_avg_us"], d["read_bytes_per_launch_raw"] / 1e6, d["write_bytes_per_launch_raw"] / 1e6))
for c in "BE":
    src = G + "round6_valu_roof_%s.json" % c
    if os.path.exists(src):
        d = json.load(open(src)); d["commit"] = built
        json.dump(d, open(P + "round6_valu_roof_%s.json" % c, "w"), indent=1)
        print(c, "valu roof: %.0f insts/wave, %.2f cyc/inst, %.0f waves; static symbol %s" % (d.get("valu_insts_per_wave", 0), d.get("cycles_per_valu_inst", 0), d.get("waves_per_launch", 0), (d.get("static_isa") or {}).get("symbol")))
for n in ("phases_B.txt", "phases_E.txt", "kernels_sequence.md", "kernels_batched_S8.md", "kernels_batched_S32.md", "run_cost.txt", "batched_values_S8.txt", "batched_values_S32.txt",
          "rs_stores_E.txt", "tracker_phases.txt", "tracker_opt.txt", "keyframe_calls.txt"):
    if os.path.exists(R + n):
        shutil.copy(R + n, P + "round6_" + n)
for n in ("bench_default", "bench_driver_cmd", "bench_extras"):
    line = open(R + n + ".json").read().strip().splitlines()[-1]
    d = json.loads(line)
    open(P + "round6_" + n + ".json", "w").write(line + "\n")
    if os.path.exists(R + n + "_detail.json"):
        shutil.copy(R + n + "_detail.json", P + "round6_" + n + "_detail.json")
    print(n, "len %d value %.4e ms %.4f K1 %.2f frac %.4f parity %s ss %.4f cpu x%s | %s" % (
        len(line), d["value"], d["ms_per_step"], d["roofline"]["launch_us"], d["roofline"]["frac"], d.get("parity_ok"), d["schur_solve_ms"], d.get("gpu_over_cpu"),
        json.dumps({k: d.get(k) for k in ("configs", "sequence", "tracker", "solve")})))

for n in ("gather_roof_E.json", "shards_per_gpu.json"):
    if os.path.exists(R + n):
        d = json.load(open(R + n)); d["commit"] = built
        json.dump(d, open(P + "round6_" + n, "w"), indent=1)
        print(n, json.dumps(d.get("variants") or d.get("S"))[:600])
v = os.path.join(G, "config_e_vs_fp32_texels.json")
if os.path.exists(v):
    d = json.load(open(v)); d["commit"] = built
    json.dump(d, open(P + "round6_config_e_vs_fp32_texels.json", "w"), indent=1)
