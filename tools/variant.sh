#!/bin/bash
# diagnostic: build a variant of libsspgpu.so with extra -D flags for the fused MFCC kernels only (benchmark instances only)
#   tools/variant.sh <name> [-DFLAG ...]   ->  tools/scratch/variants/<name>.so   (run with SSP_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/scratch/variants
O=speech_signal_processing_amd/csrc/_obj
for src in mfcc_fast mfcc_stream; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -DSSP_FAST_MINIMAL "$@" \
      -c speech_signal_processing_amd/csrc/$src.hip -o tools/scratch/variants/$name.$src.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/scratch/variants/$name.so tools/scratch/variants/$name.mfcc_fast.o tools/scratch/variants/$name.mfcc_stream.o \
    $(ls $O/*.o | grep -v "mfcc_fast.o\|mfcc_stream.o")
rm tools/scratch/variants/$name.*.o
echo built tools/scratch/variants/$name.so
