#!/bin/bash
# GPU box: per-GPU time of an N-GPU job (shard 0 of N, no exchange) for several part counts.
# Usage: tools/shard_sweep.sh "1 8" "1 2 4"
for n in $1; do for p in $2; do
  timeout -k 10 120 python3 -u bench.py --steps 10 --warmup 3 --no-cpu --emulate-shards $n --parts $p 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('shards %d parts %d: %7.3f ms/step trace %.3f  stage_ms %s' % ($n, $p, d['ms_per_step'], d['trace_only_ms'], d['stage_ms']))
for r in d['step_stage_ms']: print('      ', r)" || exit 1
done; done
