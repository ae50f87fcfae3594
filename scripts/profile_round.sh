#!/bin/bash
# GPU-box helper: kernel statistics and counter passes for every benchmark configuration, from the
# library as built. `bash scripts/profile_round.sh <tag>` leaves gpurun_out/<tag>_<case>_kernel_stats.csv,
# gpurun_out/<tag>_<case>_pmc_<pass>.csv and gpurun_out/<tag>_<case>.json (the bench line of the
# kernel-stats run). One program per profiler run, each under its own time limit; stops at the first failure.
set -e
TAG="$1"
FETCH="FETCH_SIZE GRBM_GUI_ACTIVE"
GEMM="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run() {   # name, counter passes ("fetch", "gemm", "fetch gemm"), bench args...
  local NAME="$1" PASSES="$2"; shift; shift
  bash scripts/prof_case.sh "${TAG}_$NAME" "$@" > gpurun_out/${TAG}_${NAME}_top.txt
  mv gpurun_out/prof_${TAG}_${NAME}_kernel_stats.csv gpurun_out/${TAG}_${NAME}_kernel_stats.csv
  mv gpurun_out/prof_${TAG}_${NAME}.json gpurun_out/${TAG}_${NAME}.json
  rm -f gpurun_out/prof_${TAG}_${NAME}.err
  for P in $PASSES; do
    if [ "$P" = fetch ]; then C="$FETCH"; else C="$GEMM"; fi
    bash scripts/pmc_case.sh "${TAG}_${NAME}_$P" "$C" "$@" > /dev/null
    mv gpurun_out/pmc_${TAG}_${NAME}_$P.csv gpurun_out/${TAG}_${NAME}_pmc_$P.csv
    rm -f gpurun_out/pmc_${TAG}_${NAME}_$P.err
  done
  echo "done $NAME"
}
run default_10Mx384 fetch --steps 30
run c2_1Mx384 fetch --rows 1000000 --steps 100
run c3_1Mx768_b256 "fetch gemm" --rows 1000000 --dim 768 --batch 256 --steps 20
run c3_1Mx768_b256_mirror "fetch gemm" --rows 1000000 --dim 768 --batch 256 --steps 20 --mirror
run c4shard_1.25Mx384_b1 fetch --rows 1250000 --steps 100
# the same single-query step with the exchange a sharded run adds (RCCL at world size 1: pack, all-gather, merge)
export TRAG_BENCH_FORCE_EXCHANGE=1
run c4shard_1.25Mx384_b1_exchange "" --rows 1250000 --steps 100
unset TRAG_BENCH_FORCE_EXCHANGE
run c4shard_1.25Mx384_b1000 "fetch gemm" --rows 1250000 --batch 1000 --steps 20
run c4shard_1.25Mx384_b1000_mirror "fetch gemm" --rows 1250000 --batch 1000 --steps 20 --mirror
run c4shard_1.25Mx384_b256 gemm --rows 1250000 --batch 256 --steps 20
run c5_2Mx4096_hybrid fetch --rows 2000000 --dim 4096 --hybrid --steps 20
