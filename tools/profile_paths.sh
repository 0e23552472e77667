#!/bin/bash
# Run ON the GPU box (via gpurun): for every named workload of tools/prof_paths.py one kernel-trace pass and separate PMC
# passes (never combined with a trace domain; the program itself follows `--`).
# Usage: tools/profile_paths.sh <tag> <name> [<name> ...]      Output: gpurun_out/prof_<tag>/<name>/summary.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $ROOT
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/prof_$TAG/$NAME
  mkdir -p $OUT
  echo "== $NAME: kernel trace"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/prof_paths.py $NAME 3 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
  grep '^{' $OUT/trace.log | tail -1
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 tools/prof_paths.py $NAME 1 > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
  done
  python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  grep '^{' $OUT/trace.log | tail -1 >> $OUT/summary.txt
  find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
  rm -rf $OUT/trace $OUT/pmc_*/  # raw csv: large; the summary keeps the per-kernel means
done
echo done
