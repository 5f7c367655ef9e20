#!/bin/bash
# Runs on the GPU box: bench.py (no CPU leg) for the default library, every build/librt_hip_*.so variant and a few
# walk-wave counts; one summary line each.
set -o pipefail
run() {  # name, lib ("" = default), extra bench args
  if [ -n "$2" ]; then export RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_$2.so; else unset RT_HIP_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu --steps 15 $3 > gpurun_out/sweep_$1.json 2> gpurun_out/sweep_$1.err || { echo "$1 FAILED"; tail -3 gpurun_out/sweep_$1.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$1.json"))
print("%-14s %7.1f Mrays/s %7.3f ms | one at a time: stages %s trace %.3f" % ("$1", d["value"], d["ms_per_step"], d["stage_ms"], d["trace_only_ms"]))
PY
}
run default "" "" || exit 1
for v in $VARIANTS; do run $v $v "" || exit 1; done
for w in $WAVES; do run waves$w "" "--walk-waves $w" || exit 1; done
for f in $FLIGHT; do run flight$f "" "--in-flight $f" || exit 1; done
