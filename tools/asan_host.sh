#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libsspgpu.so (device code untouched: GPU ASan is not available on this pool), then the CPU
# test suite through it: plan / segment / table-building / error-path code of the C-ABI under the sanitizers.  Run on the CPU box:
#   tools/asan_host.sh            -> tools/scratch/asan/libsspgpu_asan.so, pytest -m "not gpu" with it
set -e
cd "$(dirname "$0")/.."
OUT=tools/scratch/asan
mkdir -p $OUT
SRC=$(python - <<'PY'
from speech_signal_processing_amd import build
print(" ".join(build.SOURCES))
PY
)
pids=()
for s in $SRC; do
  extra=""
  [ "$s" = "gmm.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -Wno-pass-failed $extra \
      -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
      -c speech_signal_processing_amd/csrc/$s -o $OUT/${s%.hip}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libsspgpu_asan.so $OUT/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
echo "built $OUT/libsspgpu_asan.so; running the CPU suite under $RT"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD=$RT SSP_LIB_PATH=$PWD/$OUT/libsspgpu_asan.so python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -15
