#!/bin/bash
# GPU box: kernel trace of device k-d rebuilds of the bench mesh; prints, for the last rebuild, the time the GPU was
# busy, the time from its first kernel to its last, and the number of kernels. Usage: tools/kd_build_trace.sh
set -o pipefail
OUT=gpurun_out/kdtrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/kd_build_bench.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# builds are separated by traces: split at k_kd_root_box
starts = [i for i, r in enumerate(rows) if "k_kd_root_box" in r["Kernel_Name"]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(rows)) if "k_kd_leaf_emit" in rows[i]["Kernel_Name"])
b = rows[i0:i1 + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in b)
span = int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])
print("last rebuild: %d kernels, GPU busy %.1f us, first start to last end %.1f us" % (len(b), busy / 1e3, span / 1e3))
agg = {}
for r in b:
    k = r["Kernel_Name"].split("(")[0][-40:]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("  %-42s x%3d %8.1f us" % (k, n, t / 1e3))
PY
tail -3 $OUT/log.txt
