"""Per-kernel digest of the PMC passes written by tools/prof.sh."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rt::", "").replace("void ", "").split("(")[0][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if a.get("SQ_ACTIVE_INST_VALU", 0) > 0 and a.get("SQ_WAVE_CYCLES", 0) > 1e8:
        print("%-22s lanes/VALU %4.1f VALU %.3g SALU %.3g VMEM %.3g LDS %.3g SMEM %.3g | wait %.2f issue-stall %.2f active %.2f | waves %d L2hit %.3f TCPacc %.3g TCCreq %.3g busy %.3g ldsconf %.3g"
              % (k, a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"], a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_VMEM_RD"],
                 a["SQ_INSTS_LDS"], a["SQ_INSTS_SMEM"], a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
                 a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAVES"],
                 a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"] + 1), a["TCP_TOTAL_CACHE_ACCESSES_sum"], a["TCC_REQ_sum"], a["SQ_BUSY_CYCLES"], a["SQ_LDS_BANK_CONFLICT"]))
