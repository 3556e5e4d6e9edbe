#!/bin/bash
# diagnostic: a variant of libsspgpu.so with ONE source rebuilt with extra flags
#   tools/variant1.sh <name> <source-stem> [flags ...]   ->  tools/scratch/variants/<name>.so   (run with SSP_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
mkdir -p tools/scratch/variants
O=speech_signal_processing_amd/csrc/_obj
# (the per-source flags of build.py apply to variants too: a gmm.hip variant without them is a different kernel)
SF=$(python3 -c "import sys; sys.path.insert(0, '.'); from speech_signal_processing_amd.build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$src.hip', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -Wno-pass-failed $SF "$@" \
    -c speech_signal_processing_amd/csrc/$src.hip -o tools/scratch/variants/$name.$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/scratch/variants/$name.so tools/scratch/variants/$name.$src.o \
    $(ls $O/*.o | grep -v "/$src.o")
rm tools/scratch/variants/$name.$src.o
echo built tools/scratch/variants/$name.so
