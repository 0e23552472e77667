#!/bin/bash
# Run ON the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args for the PMC passes...]
# The kernel-trace pass runs the default bench workload (TRACE_ARGS, default --steps 10 --warmup 2 --no-cpu).
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by hand afterwards.
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 2 --warmup 1 --no-cpu --no-extra}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py ${TRACE_ARGS:---steps 10 --warmup 2 --no-cpu --no-extra} > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
