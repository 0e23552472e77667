#!/bin/bash
# Run ON the GPU box (via gpurun): kernel-trace stats + separate PMC passes (never combined with a trace domain) for any
# python workload of this repo.   Usage: tools/profile_cmd.sh <tag> <script.py> [args...]
# The program itself follows `--` (python3 <script>): no env / bash -c hop between rocprofv3 and the GPU program.
# Output: gpurun_out/prof_<tag>/summary.txt (copy what matters to profiles/ by hand).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== kernel trace: $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
tail -2 $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 "$@" > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
