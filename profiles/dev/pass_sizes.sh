#!/bin/bash
# bench.py at several pass sizes (samples of every pixel in flight at once).  usage: profiles/dev/pass_sizes.sh 64 128 256
for s in "$@"; do
  timeout -k 10 300 python3 bench.py --spp-per-step $s --steps 3 --warmup 1 --cpu-baseline-spp 0 2>/dev/null | tail -1 > /tmp/pass.json || exit 1
  python3 - $s <<'PY'
import json, sys
j = json.load(open("/tmp/pass.json")); r = j["roofline"]; n = r["launches"]
print("spp per pass", sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms per step", {k: round(v / n, 2) for k, v in j["kernel_ms"].items() if v})
PY
done
