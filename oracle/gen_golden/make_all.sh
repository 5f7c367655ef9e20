#!/bin/bash
# Regenerates every fixture under tests/golden/ from the reference (this container only).
set -e
cd "$(dirname "$0")/../.."
G=tests/golden
H=oracle/gen_golden/ref_capture.js
K8=8-kd_tree.js
node $H --script $K8 --model models/bun_zipper_res4.ply --out $G/k8_res4_640x480 --full-arrays
node $H --script $K8 --model models/bun_zipper_res3.ply --out $G/k8_res3_640x480 --full-arrays
node $H --script $K8 --model models/bun_zipper_res2.ply --out $G/k8_res2_640x480
node $H --script $K8 --model models/bun_zipper.ply      --out $G/k8_full_640x480
# benchmark configuration 2: res4, depth 1, 1920x1080, light animated to t = 7.5 s
node $H --script $K8 --model models/bun_zipper_res4.ply --width 1920 --height 1080 --time 7.5 --out $G/c2_res4_1920x1080
# benchmark configuration 3 host side: full bunny, depth 5, 3840x2160 (camera rays + uniforms; the k-d arrays equal k8_full)
node $H --script $K8 --model models/bun_zipper_res4.ply --width 3840 --height 2160 --depth 5 --out $G/c3_rays_3840x2160
# project.js / 7-tone_reproduction.js: depth 4, 2x2 supersampling, triangles off; tone stage on synthetic frames
node $H --script project.js --out $G/project_640x480
node $H --script 7-tone_reproduction.js --pixels zero --out $G/tone7_zero_640x480
node $H --script 7-tone_reproduction.js --pixels const:10,200,30 --set TONE_REPRODUCTION_METHOD=1 --out $G/tone7_const_640x480
# benchmark configuration 1: checkpoint 6
node $H --script 6-transmission.js --ticks 2 --out $G/c1_ch6_640x480
# benchmark configuration 3 as benchmarked: full bunny + teapot.obj (the reference's own loadOBJ and Mesh class; the harness
# rebases the second mesh's indices), reflective + transmissive mesh material, depth 5, 3840x2160
node $H --script $K8 --model models/bun_zipper.ply --width 3840 --height 2160 --depth 5 \
     --mesh-material 0.7,0.6,0.7,16,0.3,0.5,0.9 --extra-obj models/teapot.obj --extra-scale 60 --extra-translate 50,75,10 \
     --out $G/c3_bunny_teapot
# the same two meshes at the smallest bunny resolution, full arrays kept (teapot ingest can be diffed element by element)
node $H --script $K8 --model models/bun_zipper_res4.ply --depth 3 \
     --mesh-material 0.7,0.6,0.7,16,0.3,0.5,0.9 --extra-obj models/teapot.obj --extra-scale 60 --extra-translate 50,75,10 \
     --out $G/k8_res4_teapot_640x480 --full-arrays
# N3: the mesh picked with a click, grabbed and dragged with the reference's own UI.mouseDown / mouseMove / mouseUp
# (js/8-kd_tree.js:2718-2792); the arrays and the tree are those of the frame after the drag
node $H --script $K8 --model models/bun_zipper_res4.ply --depth 3 --mesh-material 0.7,0.6,0.7,16,0.3,0.5,0.9 \
     --drag 330,40,335,120,330,200 --out $G/k8_res4_drag_640x480
# N3: camera controls. Mouse events delivered to the listeners the reference's THREE.TrackballControls registered (a pan
# with the right button, a rotation with the left one, a wheel zoom), a tick after each; the camera is recorded per tick
node $H --script $K8 --model models/bun_zipper_res4.ply \
     --trackball "down:2:20:20,move:24:22,tick,up:24:22,tick,tick,down:0:20:20,move:22:21,tick,move:23:23,tick,up:23:23,tick,tick,wheel:-100,tick,tick" \
     --out $G/k8_res4_trackball_640x480
