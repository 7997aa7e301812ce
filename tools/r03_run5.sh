# round 3, fifth GPU call: the persistent form for the read-modify-write epilogues, in situ A/B on one box
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q 2>&1 | tail -3
for cfg in "gemm_persistent_resid=0" "gemm_persistent_resid=1" "gemm_persistent_resid=2" "gemm_persistent_resid=3" "gemm_persistent_resid=0" "gemm_persistent_resid=1"; do
  R3G_OPTIONS=$cfg timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_ab4_${cfg}_$(date +%s).json 2>> gpurun_out/r03_ab4.err
  python - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/r03_ab4_${cfg}_*.json"))[-1]
d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
print("${cfg}", round(d["value"],4), round(d["ms_per_step"],1), "gemm", round(r["achieved"]), r["families_ms_per_object"]["gemm"], "mc", d.get("mc_parity"))
PY
done
