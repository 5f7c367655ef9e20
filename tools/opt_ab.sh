#!/bin/bash
# GPU box: bench.py with a list of option sets, one line each. Usage: tools/opt_ab.sh "" "--opt sort_rays=1" ...
set -o pipefail
for o in "$@"; do
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 20 $o 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s %8.1f Mrays/s %7.3f ms/frame | one at a time: trace %.3f stages %s' % ('$o', d['value'], d['ms_per_step'], d['trace_only_ms'], d['stage_ms']))
print('      walk per step', [r[1] for r in d['step_stage_ms']], ' gen per step', [r[0] for r in d['step_stage_ms']])" || exit 1
done
