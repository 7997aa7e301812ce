#!/bin/bash
# round 4: where the texture stage's time goes at upstream's sizes (kernel trace of tests/tex_stage_time.py)
set -x
R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_tex_trace -o t -- python $R/tests/tex_stage_time.py > $R/gpurun_out/r04_texture_stage_time.json 2> $R/gpurun_out/r04_tex_trace.log
cd $R
tail -c 1500 gpurun_out/r04_texture_stage_time.json
DB=$(ls gpurun_out/r04_tex_trace/*/*_results.db gpurun_out/r04_tex_trace/*_results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB "x" > gpurun_out/r04_tex_kernel_stats.md
head -45 gpurun_out/r04_tex_kernel_stats.md
rm -rf gpurun_out/r04_tex_trace
