set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for v in new base; do
  if [ $v = base ]; then export RT_HIP_LIB=$PWD/ray-tracer_amd/csrc/build/librt_hip_base.so; else unset RT_HIP_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu --steps 20 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v", round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms; stage_ms", d["stage_ms"], "trace_only", round(d["trace_only_ms"],3), "records", d["tri_records"])
print("   step_stage_ms", d["step_stage_ms"])
PY
done
