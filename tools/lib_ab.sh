#!/bin/bash
# GPU box: bench.py for the default library and the named tuning builds (ray-tracer_amd/csrc/build/librt_hip_NAME.so). Usage: tools/lib_ab.sh "NAME ..." [bench args]
set -o pipefail
names=$1; shift
for v in default $names default; do
  if [ $v = default ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_$v.so; fi
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 20 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-10s %8.1f Mrays/s %7.3f ms/frame | one at a time: trace %.3f stages %s records %s' % ('$v', d['value'], d['ms_per_step'], d['trace_only_ms'], d['stage_ms'], d['tri_records']))" || exit 1
done
