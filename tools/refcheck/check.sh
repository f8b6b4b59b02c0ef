#!/bin/sh
# Compile-only check of the reference-side adapter (tools/refcheck/adapter_check.cpp) against the reference's own headers.
# Build container only: needs /root/reference; writes nothing into the repository (cml/config.h is produced from the reference's
# config.h.in in a temporary directory, by the substitutions its cmake performs with every optional dependency switched off).
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
[ -d "$REF/src/cml" ] || { echo "reference tree absent: nothing to check"; exit 0; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/cml"
sed -e 's/#cmakedefine01 \([A-Z_]*\)/#define \1 0/' \
    -e 's/#cmakedefine \(CML_[A-Z_]*MAP_IMPLEMENTATION\) .*/#define \1 \1_PHMAP/' \
    "$REF/src/cml/config.h.in" > "$TMP/cml/config.h"
g++ -std=c++17 -fsyntax-only -Wall -Wno-unused -Wno-deprecated-declarations \
    -I"$TMP" -I"$REF/src" -I"$REF/thirdparty/eigen" -I"$REF/thirdparty/Sophus" -I"$REF/thirdparty/spdlog/include" -I"$REF/thirdparty" \
    "$HERE/adapter_check.cpp"
echo "adapter_check.cpp: the boundary records of include/cmlhip.h fill from the reference's types (syntax + type check passed)"
