#!/bin/bash
# Run on the GPU box: readbacks ending in a completion ticket the host spins on (default) against the stream-synchronise path (CMLHIP_NO_POLL=1),
# same box, alternating: the bench's sequence object and the per-call table of a keyframe.
set -u
OUT=gpurun_out/ab_poll
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do
  for mode in poll nopoll; do
    if [ $mode = nopoll ]; then export CMLHIP_NO_POLL=1; else unset CMLHIP_NO_POLL; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sequence']; print('$mode', 'ms_per_step', round(d['ms_per_step'],5), 'frame_ms', s['frame_ms'], 'lib fps', s['library_frames_per_s'], 'run_ms', s['run_ms'], 'fps', s['frames_per_s'], 'schur_solve', d.get('schur_solve_ms'))" | tee -a $OUT/bench.txt
  done
done
for mode in poll nopoll; do
  if [ $mode = nopoll ]; then export CMLHIP_NO_POLL=1; else unset CMLHIP_NO_POLL; fi
  echo "== $mode" | tee -a $OUT/keyframe_calls.txt
  python tools/probe_keyframe_calls.py 2>&1 | tee -a $OUT/keyframe_calls.txt
done
