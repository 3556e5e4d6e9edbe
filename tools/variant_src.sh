#!/bin/bash
# diagnostic: a variant of libsspgpu.so whose named sources come from ANOTHER COMMIT, compiled against today's headers and linked with
# today's other objects (same host code, same ABI, same bench.py: only the kernels of those files differ)
#   tools/variant_src.sh <name> <commit> <source-stem>[,<source-stem>...] [flags ...]   ->  tools/scratch/variants/<name>.so
# (works as long as the headers stayed source-compatible between the commits: true for rounds 3..5)
set -e
cd "$(dirname "$0")/.."
name=$1; commit=$2; stems=$3; shift; shift; shift
V=tools/scratch/variants; W=$V/src_$name
mkdir -p $W
O=speech_signal_processing_amd/csrc/_obj
cp speech_signal_processing_amd/csrc/*.hpp $W/; mkdir -p tools/scratch/include; cp include/ssp.h tools/scratch/include/   # (common.hpp includes ../../include/ssp.h)
excl=""
objs=""
for s in ${stems//,/ }; do
  git show $commit:speech_signal_processing_amd/csrc/$s.hip > $W/$s.hip
  SF=$(python3 -c "import sys; sys.path.insert(0, '.'); from speech_signal_processing_amd.build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$s.hip', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -Wno-pass-failed -Iinclude $SF "$@" -c $W/$s.hip -o $W/$s.o &
  excl="$excl\|/$s.o"
  objs="$objs $W/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $objs $(ls $O/*.o | grep -v "/NONE.o$excl")
rm -rf $W
echo built $V/$name.so
