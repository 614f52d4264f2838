#!/bin/bash
# quick A/B of encode scheduling variants on one GPU box (prints value / ms per step / top kernels)
for cfg in "$@"; do
  python bench.py --steps 3 --warmup 1 --skip-cpu --skip-decode $cfg > gpurun_out/sweep_tmp.log 2>&1
  python - "$cfg" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/sweep_tmp.log") if x.startswith("{")]
if not l: print(sys.argv[1], "FAILED", open("gpurun_out/sweep_tmp.log").read()[-400:]); sys.exit()
d=json.loads(l[-1]); k=d["roofline"]["kernels_ms"]
print("%-44s %8.1f MB/s %7.2f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {a:round(b,1) for a,b in list(k.items())[:6]}))
PY
done
