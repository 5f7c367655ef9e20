"""Condenses rocprofv3 output directories (kernel-trace stats + PMC passes) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-60s calls %6s  total %12.3f us  avg %10.3f us  %6s%%" % (row["Name"][:60], row["Calls"], float(row["TotalDurationNs"]) / 1e3,
                                                                         float(row["AverageNs"]) / 1e3, row["Percentage"]))
print()
print("== per-dispatch durations of k_trace_level (us), in dispatch order ==")
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        rows = [r for r in csv.DictReader(fh) if "k_trace_level" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
    if rows:
        r = rows[0]
        print("VGPR", r.get("VGPR_Count"), "SGPR", r.get("SGPR_Count"), "LDS", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"),
              "wg", r.get("Workgroup_Size"), "grid", r.get("Grid_Size"))
print()
print("== PMC: sums per kernel over all dispatches ==")
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in find("pmc_*/**/*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-36s %18.0f  (%d dispatches)" % (c, acc[k][c], cnt[k][c]))
k = next((x for x in acc if "k_trace_level" in x), None)
if k:
    a = acc[k]
    def g(n): return a.get(n, 0.0)
    print()
    print("== derived for k_trace_level ==")
    if g("SQ_ACTIVE_INST_VALU"):
        print("VALU lane utilisation  = THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64) = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)))
    if g("SQ_WAVE_CYCLES"):
        print("wave-cycle split: WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
        print("VALU busy share of wave cycles = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
    if g("TCC_HIT_sum") + g("TCC_MISS_sum"):
        print("L2 hit rate = %.4f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    if g("FETCH_SIZE") or g("WRITE_SIZE"):
        n = max(cnt[k].get("FETCH_SIZE", 1), 1)
        print("HBM per dispatch: FETCH_SIZE %.1f KB (x2 gfx950 correction for wide reads: %.1f KB), WRITE_SIZE %.1f KB" % (g("FETCH_SIZE") / n, 2 * g("FETCH_SIZE") / n, g("WRITE_SIZE") / cnt[k].get("WRITE_SIZE", 1)))
