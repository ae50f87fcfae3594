#!/bin/bash
# Builds a variant of libtrag.so with extra compiler flags for some of its files:
#   scripts/build_variant.sh <name> [-f "<file.hip> ..."] [-DQS_FLUSH_MIN=8 ...]  ->  trueno-rag_amd/lib_var/<name>/libtrag.so
# (run a bench against it with TRAG_LIB=<that path>). Default files: the two prefilter files; the other
# objects come from trueno-rag_amd/lib.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N="$1"; shift
FILES="trag_qstat.hip trag_bf16.hip"
if [ "$1" = "-f" ]; then FILES="$2"; shift 2; fi
OUT=$ROOT/trueno-rag_amd/lib_var/$N
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -I$ROOT/include -I$ROOT/trueno-rag_amd/csrc"
cd $ROOT/trueno-rag_amd/csrc
for F in $FILES; do rm -f $OUT/${F%.hip}.o; /opt/rocm/bin/hipcc $FLAGS "$@" -c $F -o $OUT/${F%.hip}.o & done
wait
for F in $FILES; do [ -f $OUT/${F%.hip}.o ] || { echo "compile of $F failed: no variant built"; rm -f $OUT/libtrag.so; exit 1; }; done
L=$ROOT/trueno-rag_amd/lib
OBJS=""
for O in trag_api trag_kernels trag_stream trag_batch trag_bf16 trag_qstat trag_mirror trag_bm25 trag_shard trag_fuse_device trag_fuse trag_embed trag_error; do
  if [ -f $OUT/$O.o ]; then OBJS="$OBJS $OUT/$O.o"; else OBJS="$OBJS $L/$O.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libtrag.so $OBJS -ldl
echo built $OUT/libtrag.so
