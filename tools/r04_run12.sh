#!/bin/bash
# the cleaners' wall time against their kernel time on the bench's object (is the stage's 0.45 s of cleaners GPU time or latency?)
set -x
R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_clean_trace -o c -- python $R/tools/r04_cleaner_probe.py 3 > $R/gpurun_out/r04_cleaner_probe.json 2> $R/gpurun_out/r04_cleaner_probe.err
cd $R
tail -c 600 gpurun_out/r04_cleaner_probe.json; tail -3 gpurun_out/r04_cleaner_probe.err
DB=$(ls gpurun_out/r04_clean_trace/*/*_results.db gpurun_out/r04_clean_trace/*_results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB "x" > gpurun_out/r04_clean_kernel_stats.md
grep -v "gemm\|attn\|layernorm\|ln_dot" gpurun_out/r04_clean_kernel_stats.md | head -40
rm -rf gpurun_out/r04_clean_trace
