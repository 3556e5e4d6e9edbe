#!/bin/bash
# diagnostic: compile ONE csrc source to gfx950 assembly and list every kernel's register / spill / scratch figures
#   tools/isa_regs.sh <source.hip> [out.s] [-DFLAG ...]
cd "$(dirname "$0")/.."
src=$1; out=${2:-tools/scratch/${1%.hip}.s}; shift; shift
mkdir -p tools/scratch
SF=$(python3 -c "import sys; sys.path.insert(0, '.'); from speech_signal_processing_amd.build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$src', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -Wno-pass-failed $SF "$@" -S --cuda-device-only \
    -o $out speech_signal_processing_amd/csrc/$src 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|name:|private_segment_fixed_size)" $out | paste - - - - - | \
    sed 's/_ZN3ssp[0-9]*//; s/EEvNS_.*E\t/\t/; s/private_segment_fixed_size/scratch/; s/ \+/ /g'
