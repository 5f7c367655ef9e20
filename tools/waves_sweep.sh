#!/bin/bash
# GPU box: frame time for several walk occupancies. Usage: tools/waves_sweep.sh "8 12 16 24" [bench args]
W=$1; shift
for w in $W; do
  timeout -k 10 120 python3 -u bench.py --steps 10 --warmup 3 --no-cpu --walk-waves $w "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('waves/CU %2d: %8.1f Mrays/s %7.3f ms  stage_ms %s' % ($w, d['value'], d['ms_per_step'], d['stage_ms']))" || exit 1
done
