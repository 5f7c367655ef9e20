#!/bin/bash
# Host builder (kd_build.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer: builds the
# trees of two fixture meshes with 1, 4 and 16 threads, compares the export with the reference's arrays, and feeds the
# degenerate inputs (one face, no face). CPU only. Usage: tools/sanitize/run.sh
set -e
cd "$(dirname "$0")/../.."
OUT=$(mktemp -d)
SRC="tools/sanitize/kd_driver.cpp ray-tracer_amd/csrc/kd_build.cpp"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Iray-tracer_amd/csrc $SRC -o $OUT/asan -lpthread
g++ -O1 -g -std=c++17 -fsanitize=thread -Iray-tracer_amd/csrc $SRC -o $OUT/tsan -lpthread
for bin in asan tsan; do
  for g in k8_res4_640x480 k8_res3_640x480; do
    G=tests/golden/$g
    echo "== $bin $g"; $OUT/$bin $G/verts_f64.bin $G/faces_u32.bin $G/kd_nodes_sparse_u32.bin
  done
done
rm -rf $OUT
