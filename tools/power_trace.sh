#!/bin/bash
# Samples rocm-smi (power, sclk) twice a second while `bench.py --steps 3` runs; prints the busy-phase statistics.
cd "$(dirname "$0")/.."
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > gpurun_out/power_trace.jsonl &
SAMPLER=$!
timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/power_bench.json 2>/dev/null
kill $SAMPLER
python - <<'PY'
import json, re
pw, ck = [], []
for l in open("gpurun_out/power_trace.jsonl"):
    try:
        d = json.loads(l)
    except Exception:
        continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "ower" in k and "W" in k:
            try: pw.append(float(v))
            except Exception: pass
        if k.startswith("sclk"):
            m = re.search(r"(\d+)Mhz", str(v))
            if m: ck.append(int(m.group(1)))
pw_busy = [p for p in pw if p > 0.6 * max(pw)] if pw else []
print("samples", len(pw), "max W", max(pw) if pw else None, "busy mean W", sum(pw_busy) / max(1, len(pw_busy)),
      "sclk busy MHz", sorted(ck)[len(ck) // 2] if ck else None, "sclk max", max(ck) if ck else None)
print(open("gpurun_out/power_bench.json").read()[:160])
PY
head -c 600 gpurun_out/power_trace.jsonl
