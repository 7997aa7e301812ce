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
/usr/local/graft/bin/gpurun --timeout ${1:-600} -- 'mkdir -p gpurun_out; timeout 500 python -m pytest tests/test_reference_script_gpu.py -m gpu -v -s -rs 2>&1 | grep -v amdgpu.ids | tail -70 > gpurun_out/r05_reference_script_gpu.txt; cat gpurun_out/r05_reference_script_gpu.txt'
{ echo "# round 5: the reference's own stage script (src/2d_to_3d_models/run.py, unmodified) on one MI355X through libr3g.so -- INTEGRATION.md route 1,"
  echo "# both of its routes (sequential :194-213, multiprocessing pool :176-193); tools/r05_reference_on_gpu.sh, commit $(git rev-parse --short HEAD), mini-sized synthetic snapshot"
  echo "# (the pool's workers are spawned processes: the parent, whose maps the last line of that route reports, never loads the library)"
  cat gpurun_out/r05_reference_script_gpu.txt; } > profiles/r05_reference_script_gpu.txt
