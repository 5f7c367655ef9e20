#!/bin/bash
# GPU box: counters of the vector-memory path (texture addresser, vector L1) and of the issue ports for the bench
# workload, one frame at a time; sums for k_kd_walk. Usage: tools/prof_mem.sh TAG [bench args...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--steps 3 --warmup 1 --no-cpu --in-flight 1 $*"
for pass in "TA_TA_BUSY_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS" \
            "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$n -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$n.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$n.log; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_kd_walk" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-44s %18.0f  (%d rows) per dispatch %14.0f" % (k, tot[k], n[k], tot[k] / max(n[k], 1)))
PY
