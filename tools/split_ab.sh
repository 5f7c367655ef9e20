#!/bin/bash
# GPU box: parity tests, then frame time of a whole frame and of a 1/8 shard, one frame at a time and pipelined, default library against build/librt_hip_base.so
set -o pipefail
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_prune.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/vab_tests.log 2>&1 || { tail -30 gpurun_out/vab_tests.log; exit 1; }
tail -1 gpurun_out/vab_tests.log
for n in 1 8; do for fl in 1 0; do for v in default base; do
  if [ $v = default ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_$v.so; fi
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 20 --emulate-shards $n --in-flight $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('shards %d in-flight %d %-8s %8.1f Mrays/s %7.3f ms/frame | one at a time: trace %.3f stages %s' % ($n, d['frames_in_flight'], '$v', d['value'], d['ms_per_step'], d['trace_only_ms'], d['stage_ms']))" || exit 1
done; done; done
