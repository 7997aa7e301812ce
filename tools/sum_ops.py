import sys, json
from collections import OrderedDict
rows = [json.loads(l) for l in open(sys.argv[1]) if '"gemm"' in l or "hipBLASLt" in l]
d = OrderedDict()
for r in rows:
    d.setdefault((r["M"], r["N"], r["K"]), []).append(r)
for k, v in d.items():
    print(k, " ".join("%s:%d" % (str(r.get("waves", "lt")) + ("p" if r.get("persist") else ""), r["tflops"]) for r in v))
