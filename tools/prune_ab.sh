#!/bin/bash
# GPU box: parity tests, then the bench with and without pruning, then the visit counters of a stats build.
set -o pipefail
tools/variants_ab.sh "noprune" || exit 1
for v in stats statsnp; do
  RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_$v.so timeout -k 10 200 python3 -u bench.py --no-cpu --steps 1 --warmup 1 --in-flight 1 > gpurun_out/prune_$v.json 2> gpurun_out/prune_$v.err || { tail -5 gpurun_out/prune_$v.err; exit 1; }
  echo "== $v"; grep "walk stats" gpurun_out/prune_$v.err | tail -5
done
