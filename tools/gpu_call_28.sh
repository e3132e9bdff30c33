#!/bin/bash
# Round 3, GPU call 28: role timeline of the FIRST steps of a launch (fill of the pipeline) at config B; whole GPU suite on the current build
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c28
mkdir -p $O
timeout -k 10 200 python tools/mega_stamps.py 4096 6 > $O/stamps_fill.txt 2>&1; echo "stamps exit $?"
grep -v amdgpu.ids $O/stamps_fill.txt | grep -E "\[[012]\]" | head -150
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
