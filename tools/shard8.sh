#!/bin/bash
# GPU box: per-GPU time of an N-way tile split, emulated on one GPU (shard 0 of N, no exchange)
set -o pipefail
for n in ${SHARDS:-1 2 4 8}; do
  timeout -k 10 200 python bench.py --no-cpu --steps 30 --emulate-shards $n $EXTRA > gpurun_out/shard_$n.json 2> gpurun_out/shard_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/shard_$n.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/shard_$n.json"))
print("shards %d: %7.3f ms/frame | one at a time: trace %.3f stages %s | steps %s" % ($n, d["ms_per_step"], d["trace_only_ms"], d["stage_ms"], d["step_kernel_ms"]))
PY
done
