#!/bin/bash
# Host builder (kd_build.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer: builds the
# trees of two fixture meshes with 1, 4 and 16 threads, compares the export with the reference's arrays, and feeds the
# degenerate inputs (one face, no face); then the library's host code behind the C ABI under AddressSanitizer on a
# host-only context. CPU only (GPU sanitizers are not used). Usage: tools/sanitize/run.sh
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
# the library's own host code (rt_hip.hip, the C ABI) with AddressSanitizer, driven on a host-only context
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
(cd ray-tracer_amd/csrc && $HIPCC -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Xarch_host -fsanitize=address \
    -Xarch_host -fno-omit-frame-pointer -shared -o $OUT/librt_hip_asan.so rt_hip.hip kd_build_gpu.hip kd_build.cpp -lpthread)
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -Iinclude tools/sanitize/abi_driver.cpp -o $OUT/abi -L$OUT -lrt_hip_asan \
    -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib
G=tests/golden/k8_res4_640x480
echo "== C ABI, host-only context, ASan"; ASAN_OPTIONS=detect_leaks=1 $OUT/abi $G/verts_f64.bin $G/faces_u32.bin $G/face_mat_u32.bin $G/kd_nodes_sparse_u32.bin
rm -rf $OUT
