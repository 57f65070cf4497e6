#!/bin/bash
# Builds a variant of libwpdsp.so for A/B measurements with `bench.py --lib`:
#   tools/exp/build_variant.sh NAME [GIT_REF|.] [extra hipcc flags for the kernel files...]
# GIT_REF: take dsp_amd/csrc and include/ from that commit ("." = the working tree).  The library lands in
# variants/NAME/libwpdsp.so (ignored by git, sent to the GPU box with the tree).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; REF=${2:-.}; shift; shift || true
W=$(mktemp -d /tmp/wpvar.XXXXXX)
mkdir -p "$W/dsp_amd" "$ROOT/variants/$NAME"
if [ "$REF" = "." ]; then
  cp -r "$ROOT/dsp_amd/csrc" "$W/dsp_amd/csrc"; cp -r "$ROOT/include" "$W/include"
else
  (cd "$ROOT" && git archive "$REF" dsp_amd/csrc include) | tar -x -C "$W"
fi
rm -rf "$W/dsp_amd/csrc/build"; mkdir -p "$W/dsp_amd/lib"
make -C "$W/dsp_amd/csrc" -j4 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $*" >/dev/null
cp "$W/dsp_amd/lib/libwpdsp.so" "$ROOT/variants/$NAME/libwpdsp.so"
rm -rf "$W"
echo "variants/$NAME/libwpdsp.so"
