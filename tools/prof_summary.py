"""Condenses the rocprofv3 output of tools/prof.sh (kernel-trace stats + PMC passes) into a text summary on
stdout and a machine-readable digest (<dir>/digest.json) that bench.py reads for the `traffic` figure.

HBM bytes follow the gfx950 recipe: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 128-byte read
requests at 64 bytes, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (an upper bound for narrow gathers).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("rt::", "").replace("void ", "").split("(")[0]


digest = {}
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-60s calls %6s  total %12.3f us  avg %10.3f us  %6s%%" % (row["Name"][:60], row["Calls"], float(row["TotalDurationNs"]) / 1e3,
                                                                         float(row["AverageNs"]) / 1e3, row["Percentage"]))
            digest[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                          "total_us": float(row["TotalDurationNs"]) / 1e3}
print()
print("== per-dispatch durations (us) in dispatch order, last frame; registers / LDS / scratch per kernel ==")
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = defaultdict(list)
    for r in rows:
        by[short(r["Kernel_Name"])].append(r)
    frames = max(1, len(by.get("k_resolve", [])))   # k_resolve runs once per frame
    for k, rs in by.items():
        if not k.startswith("k_"):
            continue
        per_frame = max(1, len(rs) // frames)
        last = rs[-per_frame:]
        r = rs[0]
        print("%-20s %s" % (k, [round((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3, 1) for x in last]))
        print("%-20s VGPR %s AGPR %s SGPR %s LDS %s scratch %s wg %s grid %s" % ("", r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                                                                          r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size")))
        if k in digest:
            digest[k].update(vgpr=r.get("VGPR_Count"), sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"))
print()
print("== PMC: sums per kernel over all dispatches of the run ==")
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in find("pmc_*/**/*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    if not k.startswith("k_"):
        continue
    a = acc[k]

    def g(n):
        return a.get(n, 0.0)

    print(k)
    for c in sorted(a):
        print("   %-36s %18.0f  (%d dispatches)" % (c, a[c], cnt[k][c]))
    d = digest.setdefault(k, {})
    if g("SQ_ACTIVE_INST_VALU"):
        d["valu_lanes_per_inst"] = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")
        print("   -> active lanes per VALU instruction = %.1f of 64" % d["valu_lanes_per_inst"])
    if g("SQ_WAVE_CYCLES"):
        d["wave_cycles_wait"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        d["wave_cycles_issue_stall"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        d["wave_cycles_active"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
        print("   -> wave cycles: waiting %.3f  issue-stalled %.3f  issuing %.3f" % (d["wave_cycles_wait"], d["wave_cycles_issue_stall"], d["wave_cycles_active"]))
    if g("GRBM_GUI_ACTIVE") and g("SQ_INSTS_VALU"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a CU (four 16-lane SIMDs) issues at most one wave64 vector
        # instruction per cycle, so 256 per cycle for the chip
        cycles = g("GRBM_GUI_ACTIVE") / 8.0
        d["valu_issue_frac"] = g("SQ_INSTS_VALU") / (cycles * 256.0)
        d["salu_per_cycle_cu"] = g("SQ_INSTS_SALU") / (cycles * 256.0)
        # per launch, so that a live duration can be priced against the same instruction count
        d["valu_insts_per_launch"] = g("SQ_INSTS_VALU") / max(cnt[k]["SQ_INSTS_VALU"], 1)
        d["gui_cycles_per_launch"] = cycles / max(cnt[k]["GRBM_GUI_ACTIVE"], 1)
        if d.get("avg_us"):
            d["clock_mhz_profiled"] = d["gui_cycles_per_launch"] / d["avg_us"]
        print("   -> vector instructions issued per cycle and CU = %.3f of 1.0 (scalar: %.3f)" % (d["valu_issue_frac"], d["salu_per_cycle_cu"]))
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        d["insts_per_wave"] = {n: g("SQ_INSTS_" + n) / g("SQ_WAVES") for n in ("VALU", "SALU", "VMEM_RD", "LDS", "SMEM")}
    if g("TCC_HIT_sum") + g("TCC_MISS_sum"):
        d["l2_hit"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        d["tcc_req_per_launch"] = g("TCC_REQ_sum") / max(cnt[k]["TCC_REQ_sum"], 1)
        print("   -> L2 hit rate = %.4f, %.1f M requests per launch" % (d["l2_hit"], d["tcc_req_per_launch"] / 1e6))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_hit"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
        print("   -> vector L1 hit rate (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES) = %.4f" % d["l1_hit"])
    if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
        nf, nw = max(cnt[k].get("FETCH_SIZE", 1), 1), max(cnt[k].get("WRITE_SIZE", 1), 1)
        d["fetch_kib_per_launch"] = g("FETCH_SIZE") / nf
        d["write_kib_per_launch"] = g("WRITE_SIZE") / nw
        d["hbm_bytes_per_launch"] = (2.0 * d["fetch_kib_per_launch"] + d["write_kib_per_launch"]) * 1024.0
        print("   -> HBM per launch: FETCH_SIZE %.1f KiB (x2 = %.1f KiB), WRITE_SIZE %.1f KiB, corrected total %.2f MB"
              % (d["fetch_kib_per_launch"], 2 * d["fetch_kib_per_launch"], d["write_kib_per_launch"], d["hbm_bytes_per_launch"] / 1e6))
# which binary and workload the counts belong to (bench.py --identity, written by tools/prof.sh)
ident = os.path.join(out, "identity.json")
if os.path.exists(ident):
    with open(ident) as fh:
        digest["__identity"] = json.load(fh)
with open(os.path.join(out, "digest.json"), "w") as fh:
    json.dump(digest, fh, indent=1, sort_keys=True)
