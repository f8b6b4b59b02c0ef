"""development: bench.py's multi_window leg alone (S list from the command line), e.g. with CMLHIP_RS_TILE=64 to upload the windows in the throughput regime"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
s_list = tuple(int(a) for a in sys.argv[1:]) or (4, 8, 16)
out = bench.multi_window_bench(0, 0xC0FFEE, "B", 200, False, s_list=s_list)
for r in out["runs"]:
    print("S=%2d            %.1f us per round  %.3e residuals/s" % (r["S"], 1e3 * r["ms_per_round"], r["value"]))
for r in out["streams"]:
    print("S=%2d groups=%d   %.1f us per round  %.3e residuals/s" % (r["S"], r["groups"], 1e3 * r["ms_per_round"], r["value"]))
print("parity", out.get("parity_ok"), out.get("parity_error"))
