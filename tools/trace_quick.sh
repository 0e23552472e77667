#!/bin/bash
# Run ON the GPU box: kernel trace of the default bench workload, steady-state (median) duration per kernel.
# Usage: tools/trace_quick.sh [bench args...]
set -u
ARGS=${@:---steps 6 --warmup 2 --no-cpu}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_quick
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py $ARGS > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, statistics
f = glob.glob("$OUT/*kernel_trace.csv")[0]
d = {}
for r in csv.DictReader(open(f)):
    d.setdefault(r["Kernel_Name"][:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 100: print("%-72s n=%3d median %9.1f us  min %9.1f us" % (k, len(v), statistics.median(v), min(v)))
PY
