#!/bin/bash
cp libcml_amd/libcmlhip.so /tmp/orig.so
cp ab_tmp/libcmlhip_stamps.so libcml_amd/libcmlhip.so
for dbg in 0 48; do echo "== CMLHIP_RS_DBG=$dbg"; CMLHIP_RS_DBG=$dbg timeout 300 python tools/probe_rs_tiles.py E 2>&1 | tail -32; done
cp /tmp/orig.so libcml_amd/libcmlhip.so
