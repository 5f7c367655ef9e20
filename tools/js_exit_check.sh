#!/bin/bash
# Runs the 4K JavaScript render several times; stops at the first non-zero exit (GPU box).
for i in 1 2 3 4 5 6; do
  timeout -k 10 120 node ray-tracer_amd/js/examples/render.js --width 3840 --height 2160 --depth 5 --model models/bun_zipper.ply --mesh-material 0.7,0.6,0.7,16,0.3,0.5,0.9 --frames 5 > gpurun_out/js_run_$i.json
  rc=$?
  echo "run $i rc=$rc $(cut -c1-90 gpurun_out/js_run_$i.json)"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
