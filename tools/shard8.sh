#!/bin/bash
# GPU box: per-GPU time of an N-way tile split, rehearsed on one GPU (shard 0 of N without the collectives): the device
# work of rank 0 (own tiles, present of them, scatter of the gathered canvas) and of any other rank.
set -o pipefail
for n in ${SHARDS:-1 2 4 8}; do
  for role in root worker; do
    [ $n = 1 ] && [ $role = worker ] && continue
    timeout -k 10 200 python3 bench.py --no-cpu --steps 30 --emulate-shards $n --emulate-role $role $EXTRA > gpurun_out/shard_${n}_$role.json 2> gpurun_out/shard_${n}_$role.err || { echo "$n $role FAILED"; tail -3 gpurun_out/shard_${n}_$role.err; exit 1; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/shard_${n}_$role.json"))
print("shards %d %-6s: %7.3f ms/frame (%d in flight) | one at a time: trace %.3f stages %s | steps %s" % ($n, "$role", d["ms_per_step"], d["frames_in_flight"], d["trace_only_ms"], d["stage_ms"], d["step_kernel_ms"]))
PY
  done
done
