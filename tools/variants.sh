#!/bin/bash
# Runs bench.py once per tuning build under ray-tracer_amd/csrc/build (GPU box). Usage: tools/variants.sh [bench args]
for lib in ray-tracer_amd/csrc/build/librt_hip_*.so; do
  n=$(basename $lib .so | sed 's/librt_hip_//')
  RT_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 -u bench.py --steps 5 --warmup 2 --no-cpu "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s %8.1f Mrays/s %7.2f ms stages %s step1 %s kdrays %d %d' % ('$n', d['value'], d['ms_per_step'], d['stage_ms'], d['step_stage_ms'][1], d['kd_closest_rays'], d['kd_shadow_rays']))" || exit 1
done
