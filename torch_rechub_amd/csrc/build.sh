#!/bin/bash
# Builds librechub_hip.so (gfx950 only) in-tree next to the sources.  Usage: build.sh [-j N]
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
INC=../../include
FLAGS="--offload-arch=gfx950 -mcode-object-version=5 -munsafe-fp-atomics -O3 -std=c++17 -fPIC -I$INC -I. -Wall -Wno-unused-function"
mkdir -p _build
pids=()
for src in api.cpp embed.hip cross.hip optim.hip data.hip match.hip fm.hip seqpool.hip mlp.hip din.hip dinmlp.hip crossmix.hip moe.hip linear.hip gemm.hip shard.hip augru.hip; do
  [ -f "$src" ] || continue
  obj="_build/${src%.*}.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ common.h -nt "$obj" ] || [ wgrad_body.h -nt "$obj" ] || [ $INC/rechub_hip.h -nt "$obj" ]; then
    ( $HIPCC $FLAGS -x hip -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC _build/*.o -o librechub_hip.so
echo "built $(pwd)/librechub_hip.so"
