#!/bin/bash
# Rebuild libsvcmi.so for gfx950 if any kernel source changed, then hand the command to gpurun (the built .so travels with the
# snapshot; the GPU box does not rebuild).  Usage: scripts/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python whisper-vits-svc_amd/build.py > /dev/null
exec /usr/local/graft/bin/gpurun "$@"
