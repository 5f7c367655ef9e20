#!/bin/bash
# GPU box: per-kernel times of the animated-frames job (a k-d rebuild on the device per frame), rocprofv3 --kernel-trace --stats.
# Usage: tools/prof_animation.sh TAG
set -o pipefail
TAG=${1:-anim}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench_animation.py --frames 40 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee $OUT/summary.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-34s %8s %12s %10s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for r in rows[:24]:
    n = r["Name"]; m = re.search(r"k_\w+", n); n = m.group(0) if m else n[-34:]
    print("%-34s %8s %12.3f %10.2f %6s" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $OUT/trace.log | cut -c1-200
