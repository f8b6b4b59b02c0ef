#!/bin/bash
# Run on the GPU box: two builds of the host mirror libcmlhost.so (ab_tmp/libcmlhost_<name>.so) alternated on one box — the library time of the 48-frame sequence shard per stage
# (bench.py's sequence object), three rounds each.
#   gpurun -- "bash tools/ab_keyframe.sh old new"
cd "${GRAFT_REPO_ROOT:-.}"
cp libcml_amd/libcmlhost.so /tmp/orig.so
for i in 1 2 3; do
  for v in "$@"; do
    cp ab_tmp/libcmlhost_$v.so libcml_amd/libcmlhost.so
    python bench.py --no-cpu-baseline --detail /tmp/ab_$v.json > /dev/null 2>&1
    python - $v <<'PY'
import json, sys
d = json.load(open("/tmp/ab_%s.json" % sys.argv[1])); s = d["sequence"]; L = s["library_ms_per_stage"]
kf = sum(L[k]["mean_ms_per_stage"] for k in ("addNewFrame", "activatePoints+addPoints", "makeCoarseDepthL0", "tryMarginalize", "marginalizePointsF", "marginalizeFrames", "makeNewTraces"))
print("%-4s lib fps %6.0f frame_ms %.4f | keyframe outside run %.3f: " % (sys.argv[1], s["library_frames_per_s"], s["frame_ms"], kf) + " ".join("%s %.3f" % (k[:12], L[k]["mean_ms_per_stage"]) for k in ("addNewFrame", "activatePoints+addPoints", "makeCoarseDepthL0", "tryMarginalize", "marginalizePointsF", "marginalizeFrames", "makeNewTraces", "run")))
PY
  done
done
cp /tmp/orig.so libcml_amd/libcmlhost.so
