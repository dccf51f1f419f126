#!/bin/bash
# Builds (here or on the GPU box) and runs the VALU issue-cost table: tools/ubench/run.sh
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_ab
python3 tools/ubench/gen_valu_issue.py build_ab/valu_issue.hip
[ -x build_ab/valu_issue ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build_ab/valu_issue build_ab/valu_issue.hip
timeout -k 5 60 build_ab/valu_issue
