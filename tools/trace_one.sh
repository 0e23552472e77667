#!/bin/bash
# Run ON the GPU box: kernel trace only (no PMC passes) of one tools/prof_paths.py workload; per-kernel median / min
# durations and the gaps between consecutive kernels of the last call.   Usage: tools/trace_one.sh <name> [reps]
set -u
NAME=$1; REPS=${2:-10}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/prof_paths.py $NAME $REPS > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
tail -1 $OUT/log.txt
python3 - <<PY
import csv, glob, statistics
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64]) for r in csv.DictReader(open(f))))
d = {}
for s, e, k in rows: d.setdefault(k, []).append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 50: print("%-66s n=%3d median %9.1f us  min %9.1f us" % (k, len(v), statistics.median(v), min(v)))
# the last call: from the last launch of the first kernel name of a call backwards is fragile; print the last 14 launches
print("last launches (start offset us, duration us, gap to previous end us):")
tail = rows[-14:]
t0 = tail[0][0]
prev = None
for s, e, k in tail:
    print("  %9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, k))
    prev = e
PY
