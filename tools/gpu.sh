#!/bin/bash
# build in-tree, then run a command on the GPU box:  tools/gpu.sh [timeout_s] '<command>'
set -e
cd /root/repo
python -m speech_signal_processing_amd.build
T=${1:-900}; shift || true
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
