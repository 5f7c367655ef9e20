#!/bin/bash
# GPU box: the headline bench, the shard rehearsal, configs 2 and 5 and the two-rank run in one go (logs under gpurun_out/final_*).
set -o pipefail
O=gpurun_out
timeout -k 10 400 python3 -u bench.py > $O/final_bench.json 2> $O/final_bench.err || { echo bench failed; tail -5 $O/final_bench.err; exit 1; }
echo "== bench.py (default) =="; python3 - <<PY
import json
d=json.load(open("$O/final_bench.json"))
print(d["value"], "Mrays/s", d["ms_per_step"], "ms; roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","l2_frac","hbm_frac","traffic","valu_flop_frac")})
print("cpu", d["cpu_baseline"]); print("pcie", d["pcie_inclusive"]); print("stage_ms", d["stage_ms"], "trace_only", d["trace_only_ms"])
PY
echo "== shards =="; bash tools/shard8.sh || exit 1
echo "== config 2 =="; timeout -k 10 200 python3 -u bench.py --steps 50 --warmup 3 --no-cpu --model bun_zipper_res4.ply --teapot 0 --width 1920 --height 1080 --depth 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config 2:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame', d['mshadow_rays_per_s'], 'Mshadow/s')" || exit 1
echo "== config 5 (device builder / static / host builder / ss2) =="
timeout -k 10 280 python3 -u bench_animation.py --frames 120 2>&1 | tail -1 | cut -c1-330 || exit 1
timeout -k 10 280 python3 -u bench_animation.py --frames 120 --static 2>&1 | tail -1 | cut -c1-330 || exit 1
timeout -k 10 280 python3 -u bench_animation.py --frames 60 --kd-builder host 2>&1 | tail -1 | cut -c1-330 || exit 1
timeout -k 10 280 python3 -u bench_animation.py --frames 60 --ss 2 2>&1 | tail -1 | cut -c1-330 || exit 1
echo "== 2 ranks on one GPU, gloo, one command =="
timeout -k 10 280 python3 -u bench.py --gpus 2 --backend gloo --device 0 --no-cpu --steps 10 2>/dev/null | tail -1 | cut -c1-300
