#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes for the bench workload.
# Usage: tools/prof.sh TAG [bench args...]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--steps 3 --warmup 1 --no-cpu --in-flight 1 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$n -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$n.log 2>&1 || { echo "pmc pass failed: $pass"; tail -3 $OUT/pmc_$n.log; }
done
python3 bench.py $ARGS --identity > $OUT/identity.json 2>/dev/null
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
