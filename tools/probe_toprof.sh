#!/bin/bash
# Run on the GPU box: tools/probe_tracker_algebra.py on the -DTO_PROFILE variant (bash tools/build_variant.sh toprof tracker_opt.hip -DTO_PROFILE)
cd "${GRAFT_REPO_ROOT:-.}"
cp libcml_amd/libcmlhip.so /tmp/orig.so; cp ab_tmp/libcmlhip_toprof.so libcml_amd/libcmlhip.so
for i in 1 2 3; do python tools/probe_tracker_algebra.py; done
cp /tmp/orig.so libcml_amd/libcmlhip.so
