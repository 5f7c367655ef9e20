#!/bin/bash
# GPU box: frame time for combinations of emulated shard count, frames in flight and walk occupancy.
# Usage: tools/flight_sweep.sh "2 4 8" "1 2 3" "12 16"
for n in $1; do for f in $2; do for w in $3; do
  timeout -k 10 200 python3 -u bench.py --steps 20 --warmup 3 --no-cpu --emulate-shards $n --in-flight $f --walk-waves $w 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('shards %d in-flight %d waves %2d: %8.1f Mrays/s %7.3f ms/frame  retraced %d' % ($n, $f, $w, d['value'], d['ms_per_step'], d['frames_retraced']))" || exit 1
done; done; done
