#!/bin/bash
# Round 5: ONE script for every GPU measurement of the round (no more per-run scripts).
#   tools/r05_gpu.sh <step> [args]      -- run ON the GPU box (through gpurun), from the repository root; writes gpurun_out/r05_*
# steps:
#   micro      MFMA shape microbenchmark, GEMM GELU A/B (bench_gemm), attention generations (bench_attn)
#   newtests   the GPU tests added this round (RCCL one-rank group, reference script on the GPU, packed GELU, phased attention, fp16 guard)
#   ab <opts_a> <opts_b> [steps] [warmup]   bench.py A/B/A/B with R3G_OPTIONS=<opts_a> / <opts_b> ("-" = no options)
#   splitk     split-K of the texture models' convolutions: tests, kernel A/B, texture step A/B
#   texprof    kernel trace of the texture step
#   mc / tex / attn   marching-cubes timeline, texture step time, attention generations
#   suite      the whole -m gpu suite
#   bench      the driver's invocation of bench.py
#   evidence <commit> [objects per launch]  kernel trace + PMC passes (tools/profile_evidence.sh), tables -> gpurun_out/r05_*.md
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
case "$1" in
micro)
    # (tools/ubench binaries: `make -C tools/ubench` in the build container; they travel with the snapshot)
    timeout 120 tools/ubench/mfma_shape 1.0 > $O/r05_mfma_shape.jsonl 2>&1
    timeout 400 python tools/bench_gemm.py --variants 12 --gelu-pk 0,1 --shapes gelu --screen 2 --rounds 5 --iters 10 > $O/r05_gemm_gelu_ab.jsonl 2> $O/r05_gemm_gelu_ab.err
    timeout 300 python tools/bench_attn.py --gens 2,6,8,9 --shapes 0,1,2,3 --rounds 4 --iters 5 > $O/r05_attn_gens.jsonl 2> $O/r05_attn_gens.err
    tail -n 20 $O/r05_mfma_shape.jsonl $O/r05_gemm_gelu_ab.jsonl $O/r05_attn_gens.jsonl | cut -c1-400
    ;;
attn)
    timeout 300 python tools/bench_attn.py --gens "${2:-2,6,8,9}" --shapes 0,1,2,3 --rounds 4 --iters 5 > $O/r05_attn_gens.jsonl 2> $O/r05_attn_gens.err
    python - <<PY
import json
for l in open("$O/r05_attn_gens.jsonl"):
    d = json.loads(l)
    print(d["Lq"], d["Lk"], "gen", d["gen"], round(d["us_med"], 1), "us", round(d["tflops_med"]), "TF/s") if d["op"] == "attn" else print(d)
PY
    tail -3 $O/r05_attn_gens.err
    ;;
mc)
    # per-launch timeline of one marching-cubes call on the object-like field and on a noise field
    R=$(pwd)
    for f in blob noise; do
        timeout 120 python tools/bench_mc.py --field $f --iters 20 > $O/r05_mc_bench_$f.json 2>/dev/null
        (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$O/r05_mc_trace_$f -o t -- python $R/tools/bench_mc.py --field $f --iters 5 > $R/$O/r05_mc_trace_$f.log 2>&1)
        DB=$(ls $O/r05_mc_trace_$f/*/*_results.db $O/r05_mc_trace_$f/*_results.db 2>/dev/null | head -1)
        python tools/mc_timeline.py $DB "marching cubes on the 257^3 '$f' field: one call" > $O/r05_mc_timeline_$f.md
        cat $O/r05_mc_bench_$f.json | cut -c1-500; cat $O/r05_mc_timeline_$f.md
        rm -rf $O/r05_mc_trace_$f
    done
    ;;
tex)
    timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_aekl_gpu.py tests/test_unet2p5d_gpu.py -m gpu -q -x 2>&1 | tail -4
    timeout 600 python tests/tex_stage_time.py > $O/r05_texture_stage_time.json 2> $O/r05_texture_stage_time.err; tail -3 $O/r05_texture_stage_time.json | cut -c1-600
    ;;
newtests)
    timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nccl_one_rank_gpu.py tests/test_reference_script_gpu.py \
        "tests/test_model_gpu.py::test_fp16_stream_overflow_runs_the_group_again_on_the_fp32_stream" \
        "tests/test_model_gpu.py::test_pipeline_takes_a_list_of_images" -m gpu -q -x -rs 2>&1 | tail -25 > $O/r05_newtests.txt
    cat $O/r05_newtests.txt
    ;;
ab)
    A="$2"; B="$3"; K="${4:-8}"; W="${5:-4}"
    for i in 1 2; do
        for side in a b; do
            if [ $side = a ]; then OPT="$A"; else OPT="$B"; fi
            [ "$OPT" = "-" ] && OPT=""
            R3G_OPTIONS="$OPT" timeout 300 python bench.py --gpus 1 --steps $K --warmup $W --no-cpu-baseline > $O/r05_ab_${side}${i}.json 2> $O/r05_ab_${side}${i}.err
            python - <<PY
import json
d=json.loads(open("$O/r05_ab_${side}${i}.json").read().strip().splitlines()[-1])
f = d.get("roofline", {}).get("families_ms_per_object", {})
print("AB ${side}${i} opts='$OPT'", round(d["value"], 4), "obj/s", round(d["ms_per_step"], 1), "ms", {k: v for k, v in f.items() if v >= 1.0})
PY
        done
    done
    ;;
splitk)
    # split-K of the texture models' convolutions: kernel test, the texture models' tests, A/B of the conv shapes and of the texture step
    timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "split_k_of_the_convolutions" 2>&1 | tail -5
    timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_aekl_gpu.py tests/test_unet2p5d_gpu.py -m gpu -q -x 2>&1 | tail -4
    timeout 300 python tools/bench_gemm.py --splitk --shapes tex6,tex1 --screen 0 --rounds 3 --iters 10 > $O/r05_gemm_splitk.jsonl 2> $O/r05_gemm_splitk.err; tail -2 $O/r05_gemm_splitk.err
    for v in 0 1; do
        R3G_OPTIONS="gemm_splitk128=$v" timeout 600 python tests/tex_stage_time.py > $O/r05_texture_stage_time_splitk$v.json 2> $O/r05_texture_stage_time_splitk$v.err
        grep -o '"times_ms": {[^}]*}' $O/r05_texture_stage_time_splitk$v.json
    done
    ;;
texprof)
    # where the texture step's time goes at upstream's sizes: kernel trace of tests/tex_stage_time.py, by kernel and by (kernel, grid)
    R=$(pwd)
    (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/r05_tex_trace -o t -- python $R/tests/tex_stage_time.py > $R/$O/r05_texture_stage_time_traced.json 2> $R/$O/r05_tex_trace.log)
    DB=$(ls $O/r05_tex_trace/*/*_results.db $O/r05_tex_trace/*_results.db 2>/dev/null | head -1)
    python tools/rocprof_summary.py $DB "texture step (tests/tex_stage_time.py: warm-up + whole + delight only + six views only)" > $O/r05_tex_kernel_stats.md
    python tools/rocprof_summary.py $DB "texture step, by (kernel, grid)" --by-grid > $O/r05_tex_kernel_stats_by_grid.md
    head -40 $O/r05_tex_kernel_stats.md; head -60 $O/r05_tex_kernel_stats_by_grid.md
    rm -rf $O/r05_tex_trace
    ;;
suite)
    timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > $O/r05_gpu_tests.txt
    tail -15 $O/r05_gpu_tests.txt
    ;;
bench)
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench.json 2> $O/r05_bench.err
    cut -c1-600 $O/r05_bench.json; tail -2 $O/r05_bench.err
    ;;
evidence)
    bash tools/profile_evidence.sh "$2" "${3:-4}" r05
    ;;
*)
    echo "unknown step $1"; exit 2;;
esac
