#!/bin/bash
# Route 1 of INTEGRATION.md on the MI355X: the reference's stage script and the one module it imports are put
# into oracle/_ref/reference_src (git-ignored, travels with the gpurun snapshot) for ONE gpurun call and removed again; the log
# goes to profiles/.  Run from the build container (the only place /root/reference exists).
set -e
cd /root/repo
REF=/root/reference
DST=oracle/_ref/reference_src
mkdir -p $DST/src/2d_to_3d_models $DST/src/utils
cp $REF/src/2d_to_3d_models/run.py $DST/src/2d_to_3d_models/run.py
cp $REF/src/utils/global_utils.py $DST/src/utils/global_utils.py
trap 'rm -rf /root/repo/oracle/_ref/reference_src' EXIT
/usr/local/graft/bin/gpurun --timeout ${1:-600} -- 'mkdir -p gpurun_out; timeout 500 python -m pytest tests/test_reference_script_gpu.py -m gpu -q -rs 2>&1 | tail -30 > gpurun_out/r05_reference_script_gpu.txt; cat gpurun_out/r05_reference_script_gpu.txt'
cp gpurun_out/r05_reference_script_gpu.txt profiles/r05_reference_script_gpu.txt
