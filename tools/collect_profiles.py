#!/usr/bin/env python3
"""After `gpurun -- bash tools/refresh_profiles.sh`: copy gpurun_out/refresh/* into profiles/round3_* (stamped with the commit) and print the figures the documents quote."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "refresh") + "/"
P = os.path.join(ROOT, "profiles") + "/"
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
for c in "BCE":
    shutil.copy(R + "bench_%s.json" % c, P + "round3_bench_%s.json" % c)
    shutil.copy(R + "prof_%s_kernels.md" % c, P + "round3_kernels_%s.md" % c)
    d = json.load(open(R + "prof_%s_pmc.json" % c))
    d["commit"] = head
    d["note"] = ("rocprofv3 passes of the bench command at this commit (tools/profile_bench.py): --kernel-trace --stats (linearize_avg_us), --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate runs; "
                 "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (calibrated for streaming 16-B reads: an upper bound for this gather); launch_us of the bench line UNDER the profiler is "
                 "inflated by the profiler itself, the unprofiled bench line (profiles/round3_bench_%s.json) agrees with linearize_avg_us" % c)
    json.dump(d, open(P + "round3_pmc_linearize_%s.json" % c, "w"), indent=1)
    print(c, "rocprof K1 %.2f us, fetch raw %.2f MB, write %.2f MB" % (d["linearize_avg_us"], d["read_bytes_per_launch_raw"] / 1e6, d["write_bytes_per_launch_raw"] / 1e6))
shutil.copy(R + "bench_B_20steps.json", P + "round3_bench_B_20steps.json")
for n in ("phases_B.txt", "phases_E.txt", "kernels_tracker.md"):
    shutil.copy(R + n, P + "round3_" + n)
for n in ("bench_B", "bench_C", "bench_E", "bench_B_20steps"):
    d = json.loads(open(R + n + ".json").read().strip().splitlines()[-1])
    rp = d["ms_per_step_repeats"]
    print(n, "value %.4e ms %.4f rep %.4f/%.4f/%.4f K1 %.2f frac %.4f parity %s ss %.4f samples %d IN %s sampled %s" % (
        d["value"], d["ms_per_step"], rp["min"], rp["median"], rp["max"], d["linearize_kernel_us"], d["roofline"]["frac"], d.get("parity_ok"), d["schur_solve_ms"],
        d["roofline"]["launch_samples"], d.get("good_residuals"), d.get("n_sampled")))
    if "multi_window" in d:
        print("   multi_window", [(r["S"], "%.3e" % r["value"], "%.4f" % r["ms_per_round"]) for r in d["multi_window"]["runs"]], d["multi_window"].get("parity_ok"))
    if "cpu_baseline" in d and d["cpu_baseline"].get("value"):
        print("   cpu %.3e (%d threads) single %.3e -> x%.1f / x%.1f" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["single_thread"]["value"],
                                                                      d.get("gpu_over_cpu_single_socket", 0), d.get("gpu_over_cpu_single_thread", 0)))
    if "tracker" in d:
        t = d["tracker"]
        print("   tracker eval", [round(e["kernel_us"], 2) for e in t["eval"]], "host-driven %.3f" % t["optimize_host_driven_ms"], "1 hyp", t["optimize_device_resident_1_hyp"], "50 hyp", t["optimize_device_resident_50_hyp"])
