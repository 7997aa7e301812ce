#!/bin/bash
# FaceReducer with the maximal-independent-set selection: GPU == host run, contract tests, smoke, and the cleaner probe again
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_qem_gpu.py tests/test_mesh_gpu.py -m gpu -q -x 2>&1 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/r04_run12.sh 2>&1 | tail -25
