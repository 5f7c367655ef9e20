#!/bin/bash
# GPU box: parity tests of the default build, then bench.py for the default library and the named tuning builds
# (ray-tracer_amd/csrc/build/librt_hip_NAME.so), at full load and at 1/8 load. Usage: tools/variants_ab.sh "NAME ..."
set -o pipefail
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/vab_tests.log 2>&1 || { tail -30 gpurun_out/vab_tests.log; exit 1; }
tail -1 gpurun_out/vab_tests.log
for n in 1 8; do for v in default $1; do
  if [ $v = default ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_$v.so; fi
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 20 --emulate-shards $n 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('shards %d %-10s %8.1f Mrays/s %7.3f ms/frame | one at a time: trace %.3f stages %s' % ($n, '$v', d['value'], d['ms_per_step'], d['trace_only_ms'], d['stage_ms']))" || exit 1
done; done
