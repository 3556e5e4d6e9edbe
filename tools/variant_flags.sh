#!/bin/bash
# diagnostic: a variant of libsspgpu.so whose named sources are recompiled with extra flags (today's tree, the build's own flags + yours);
# every other object is today's.    tools/variant_flags.sh <name> <source-stem>[,<stem>...] [-DFLAG ...]  ->  tools/scratch/variants/<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; stems=$2; shift; shift
V=tools/scratch/variants; mkdir -p $V
O=speech_signal_processing_amd/csrc/_obj
excl=""; objs=""
for s in ${stems//,/ }; do
  SF=$(python3 -c "import sys; sys.path.insert(0, '.'); from speech_signal_processing_amd.build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$s.hip', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -munsafe-fp-atomics -Wno-pass-failed $SF "$@" \
      -c speech_signal_processing_amd/csrc/$s.hip -o $V/$name.$s.o &
  excl="$excl\|/$s.o"; objs="$objs $V/$name.$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $objs $(ls $O/*.o | grep -v "/NONE.o$excl")
rm -f $objs
echo built $V/$name.so
