#!/bin/bash
# Builds a variant of libtrag.so with extra compiler flags for the two prefilter files:
#   scripts/build_variant.sh <name> [-DQS_FLUSH_MIN=8 ...]   ->  trueno-rag_amd/lib_var/<name>/libtrag.so
# (run a bench against it with TRAG_LIB=<that path>). The other objects come from trueno-rag_amd/lib.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N="$1"; shift
OUT=$ROOT/trueno-rag_amd/lib_var/$N
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -I$ROOT/include -I$ROOT/trueno-rag_amd/csrc"
cd $ROOT/trueno-rag_amd/csrc
/opt/rocm/bin/hipcc $FLAGS "$@" -c trag_qstat.hip -o $OUT/trag_qstat.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c trag_bf16.hip -o $OUT/trag_bf16.o &
wait
L=$ROOT/trueno-rag_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libtrag.so $L/trag_api.o $L/trag_kernels.o $L/trag_stream.o $L/trag_batch.o $OUT/trag_bf16.o $OUT/trag_qstat.o $L/trag_mirror.o $L/trag_bm25.o $L/trag_shard.o $L/trag_fuse_device.o $L/trag_fuse.o $L/trag_embed.o $L/trag_error.o -ldl
echo built $OUT/libtrag.so
