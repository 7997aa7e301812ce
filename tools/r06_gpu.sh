#!/bin/bash
# Round 6: ONE script for every GPU measurement of the round (as tools/r05_gpu.sh was for round 5).
#   tools/r06_gpu.sh <step> [args]      -- run ON the GPU box (through gpurun), from the repository root; writes gpurun_out/r06_*
# steps:
#   newtests            the GPU tests added / changed this round
#   attn [gens]         attention generations x variants (tools/bench_attn.py), e.g. 2v0,2v1,2v3 and 6v0,6v1,6v3
#   abn <steps> <warmup> <opts_1> <opts_2> ...   bench.py once per option set, the whole list twice ("-" = no options)
#   ablib <other libr3g.so> [steps] [warmup]     two BUILDS of the library against each other, A/B/A/B (R3G_LIBRARY; the other build: a
#                       git worktree of the commit to compare with, built there, its .so copied next to this tree's)
#   texprof             kernel trace of the texture step (tests/tex_stage_time.py) by kernel and by (kernel, grid)
#   mc                  marching-cubes timeline on the blob and noise fields
#   suite               the whole -m gpu suite
#   bench               the driver's invocation of bench.py
#   cfg4                configs[3]-shaped bench line (513^3, fp8 geo decoder) + the texture step's time
#   evidence <commit> [objects per launch]   kernel trace + PMC passes (tools/profile_evidence.sh), tables -> gpurun_out/r06_*
#   attnpmc             SQ_INSTS_VALU / SQ_INSTS_MFMA of the attention kernels, variant 0 and 1
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
case "$1" in
newtests)
    timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -8 > $O/r06_newtests.txt
    timeout 900 python -m pytest "tests/test_model_gpu.py::test_linear1_as_one_persistent_launch_and_sliced_epilogues_are_bit_identical" \
        "tests/test_model_gpu.py::test_objects_sharing_a_launch_get_bit_identical_results" \
        "tests/test_model_gpu.py::test_full_width_dit_block_pair" \
        "tests/test_unet2p5d_gpu.py::test_guidance_pair_in_one_launch_set_equals_the_two_calls" -m gpu -q -x -rs 2>&1 | tail -15 >> $O/r06_newtests.txt
    cat $O/r06_newtests.txt
    ;;
attn)
    timeout 400 python tools/bench_attn.py --gens "${2:-2v0,2v1}" --shapes 0,2,3 --rounds 4 --iters 5 > $O/r06_attn_gen2.jsonl 2> $O/r06_attn_gen2.err
    timeout 400 python tools/bench_attn.py --gens "${3:-6v0,6v1}" --shapes 1 --rounds 4 --iters 5 > $O/r06_attn_gen6.jsonl 2> $O/r06_attn_gen6.err
    python - <<PY
import json
for f in ("$O/r06_attn_gen2.jsonl", "$O/r06_attn_gen6.jsonl"):
    for l in open(f):
        d = json.loads(l)
        if d["op"] == "attn": print(d["Lq"], d["Lk"], "gen", d["gen"], round(d["us_med"], 1), "us", round(d["tflops_med"]), "TF/s")
        else: print(d)
PY
    tail -3 $O/r06_attn_gen2.err $O/r06_attn_gen6.err
    ;;
abn)
    K="$2"; W="$3"; shift 3
    for i in 1 2; do
        n=0
        for OPT in "$@"; do
            n=$((n + 1))
            [ "$OPT" = "-" ] && OPT=""
            R3G_OPTIONS="$OPT" timeout 300 python bench.py --gpus 1 --steps $K --warmup $W --no-cpu-baseline > $O/r06_ab_${n}_${i}.json 2> $O/r06_ab_${n}_${i}.err
            python - <<PY
import json
try:
    d=json.loads(open("$O/r06_ab_${n}_${i}.json").read().strip().splitlines()[-1])
    f = d.get("roofline", {}).get("families_ms_per_object", {})
    print("AB set $n run $i opts='$OPT'", round(d["value"], 4), "obj/s", round(d["ms_per_step"], 1), "ms", {k: v for k, v in f.items() if v >= 1.0})
except Exception as e:
    print("AB set $n run $i opts='$OPT' FAILED", e)
PY
            tail -2 $O/r06_ab_${n}_${i}.err | cut -c1-300
        done
    done
    ;;
ablib)
    # two BUILDS of the library against each other on one box: tools/r06_gpu.sh ablib <other .so> [steps] [warmup]  (A = the other, B = this tree's)
    OTHER="$2"; K="${3:-8}"; W="${4:-4}"
    for i in 1 2; do
        for side in a b; do
            if [ $side = a ]; then export R3G_LIBRARY="$(pwd)/$OTHER"; else unset R3G_LIBRARY; fi
            timeout 300 python bench.py --gpus 1 --steps $K --warmup $W --no-cpu-baseline > $O/r06_ablib_${side}${i}.json 2> $O/r06_ablib_${side}${i}.err
            python - <<PY
import json
try:
    d=json.loads(open("$O/r06_ablib_${side}${i}.json").read().strip().splitlines()[-1])
    f = d.get("roofline", {}).get("families_ms_per_object", {})
    print("ABLIB ${side}${i} lib='${R3G_LIBRARY:-this tree}'", round(d["value"], 4), "obj/s", round(d["ms_per_step"], 1), "ms", {k: v for k, v in f.items() if v >= 1.0}, d.get("mc_parity"))
except Exception as e:
    print("ABLIB ${side}${i} FAILED", e)
PY
            tail -2 $O/r06_ablib_${side}${i}.err | cut -c1-300
        done
    done
    unset R3G_LIBRARY
    ;;
mc)
    R=$(pwd)
    for f in blob noise; do
        timeout 120 python tools/bench_mc.py --field $f --iters 20 > $O/r06_mc_bench_$f.json 2>/dev/null
        (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$O/r06_mc_trace_$f -o t -- python $R/tools/bench_mc.py --field $f --iters 5 > $R/$O/r06_mc_trace_$f.log 2>&1)
        DB=$(ls $O/r06_mc_trace_$f/*/*_results.db $O/r06_mc_trace_$f/*_results.db 2>/dev/null | head -1)
        python tools/mc_timeline.py $DB "marching cubes on the 257^3 '$f' field: one call" > $O/r06_mc_timeline_$f.md
        cat $O/r06_mc_bench_$f.json | cut -c1-600; cat $O/r06_mc_timeline_$f.md
        rm -rf $O/r06_mc_trace_$f
    done
    ;;
suite)
    timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > $O/r06_gpu_tests.txt
    tail -15 $O/r06_gpu_tests.txt
    ;;
bench)
    timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
    cut -c1-800 $O/r06_bench.json; tail -2 $O/r06_bench.err
    ;;
cfg4)
    timeout 900 python bench.py --gpus 1 --octree-resolution 512 --fp8-geo --steps 4 --warmup 4 --no-cpu-baseline > $O/r06_bench_cfg4.json 2> $O/r06_bench_cfg4.err
    cut -c1-1200 $O/r06_bench_cfg4.json; tail -3 $O/r06_bench_cfg4.err
    ;;
texprof)
    # where the texture step's time goes at upstream's sizes: kernel trace of tests/tex_stage_time.py, by kernel and by (kernel, grid)
    R=$(pwd)
    (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/r06_tex_trace -o t -- python $R/tests/tex_stage_time.py > $R/$O/r06_texture_stage_time_traced.json 2> $R/$O/r06_tex_trace.log)
    DB=$(ls $O/r06_tex_trace/*/*_results.db $O/r06_tex_trace/*_results.db 2>/dev/null | head -1)
    python tools/rocprof_summary.py $DB "texture step (tests/tex_stage_time.py: warm-up + whole + delight + six views + the two event-timed passes)" > $O/r06_tex_kernel_stats.md
    python tools/rocprof_summary.py $DB "texture step, by (kernel, grid)" --by-grid > $O/r06_tex_kernel_stats_by_grid.md
    head -30 $O/r06_tex_kernel_stats.md | cut -c1-200
    rm -rf $O/r06_tex_trace
    ;;
evidence)
    bash tools/profile_evidence.sh "$2" "${3:-4}" r06
    ;;
attnpmc)
    # VALU and MFMA instruction counts of the attention kernels: one counter pass per variant
    R=$(pwd)
    cd /tmp && export TMPDIR=/tmp
    for v in 0 1; do
        R3G_OPTIONS="attn_variant=$v" timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-include-regex "attn" --output-format csv -d $R/$O/r06_attnpmc_v$v -o p -- python $R/tools/bench_attn.py --gens 2,6 --shapes 0,1 --rounds 1 --iters 1 > $R/$O/r06_attnpmc_v$v.log 2>&1
    done
    cd $R
    python tools/attn_insts_table.py $O/r06_attnpmc_v0 $O/r06_attnpmc_v1 > $O/r06_attn_insts.md 2>&1
    cat $O/r06_attn_insts.md
    rm -rf $O/r06_attnpmc_v0 $O/r06_attnpmc_v1
    ;;
*)
    echo "unknown step $1"; exit 2;;
esac
