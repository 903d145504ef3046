#!/bin/bash
# The library's HOST code under AddressSanitizer on the device: every .hip / .cpp of quilt_amd/csrc compiled with
# -fsanitize=address for the host side only (-fno-gpu-sanitize: gfx950 is not an xnack+ target), linked into
# quilt_amd/csrc/libquilt_amd_asan.so, and the GPU suite run against it with the sanitizer's runtime preloaded into python.
#   build here (no GPU needed):   bash scripts/asan_host.sh build
#   run on the GPU box:           gpurun --timeout 2400 -- 'bash scripts/asan_host.sh run'     (reports: gpurun_out/asan.<pid>)
# Round 6's run: 209 tests of 14 files, one finding -- qa_gibbs_batch read runif_shard (45 doubles) although no pass would draw from
# it, past the end of a test's one-element array (csrc/gibbs.hip, fixed) -- and a clean report after the fix.  Not covered: the shim
# tests (HIP does not initialise under the sanitizer when the library comes in through the shim's DT_NEEDED on this image), the
# multi-process test.  Remove the *_asan.so afterwards: they are not the product.
set -e
cd "$(dirname "$0")/.."
CSRC=quilt_amd/csrc
case "$1" in
build)
    mkdir -p /tmp/asanlib/obj
    for f in $CSRC/*.hip; do
        /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fsanitize=address -fsanitize-recover=address \
            -fno-gpu-sanitize -shared-libsan -Wno-unused-function -c $f -o /tmp/asanlib/obj/$(basename ${f%.hip}).o &
    done
    for f in $CSRC/*.cpp; do
        /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fsanitize=address -fsanitize-recover=address \
            -fno-gpu-sanitize -shared-libsan -c $f -o /tmp/asanlib/obj/$(basename ${f%.cpp}).host.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -shared-libsan /tmp/asanlib/obj/*.o -lz -o $CSRC/libquilt_amd_asan.so
    ls -la $CSRC/libquilt_amd_asan.so ;;
run)
    A=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
    mkdir -p gpurun_out
    export LD_PRELOAD=$A ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$PWD/gpurun_out/asan
    export QA_DEV=1 QUILT_AMD_LIB=$PWD/$CSRC/libquilt_amd_asan.so
    python -m pytest tests/test_native_driver_gpu.py tests/test_gibbs_gpu.py tests/test_fullpass_gpu.py tests/test_configs_gpu.py \
        tests/test_mode_matrix_gpu.py tests/test_pipeline_gpu.py tests/test_rare_common_gpu.py tests/test_select_gpu.py tests/test_mspbwt_gpu.py \
        tests/test_sum_order_gpu.py tests/test_headline_gpu.py tests/test_golden_gpu.py tests/test_rtwin_gpu.py tests/test_panel_build_gpu.py \
        -m gpu -q -p no:cacheprovider > gpurun_out/asan_suite.log 2>&1 || true
    tail -3 gpurun_out/asan_suite.log
    grep -h SUMMARY gpurun_out/asan.* 2>/dev/null | sort | uniq -c || echo "no sanitizer report" ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
