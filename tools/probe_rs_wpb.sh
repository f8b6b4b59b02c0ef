#!/bin/bash
# K1 at config E: waves per workgroup of the lane-per-residual kernel (CMLHIP_RS_WPB 4 = shipped | 1) x phase shift
for wpb in 4 1; do for st in 8 6 4 0; do
  echo "wpb=$wpb stagger=$st: $(CMLHIP_RS_WPB=$wpb CMLHIP_RS_STAGGER=$st bash tools/probe_rs_dbg.sh 0)"
done; done
