#!/bin/bash
# On the GPU box: bench.py once per library in variants/ (selected with PTAMD_LIB), one line each.
# BENCH_ARGS: extra arguments for bench.py (e.g. "--traversal wide --spp-per-step 32").
for f in variants/*.so; do
  out=$(PTAMD_LIB=$PWD/$f timeout -k 10 120 python bench.py --steps 4 --warmup 1 --cpu-baseline-spp 0 $BENCH_ARGS) || { echo "$f FAILED"; exit 1; }
  echo "$f $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms"]; n=d["roofline"]["launches"]; print(d["value"], "Mrays/s  per launch ms: extend", round(k["extend"]/n,3), "connect", round(k["connect"]/n,3), "shade", round(k["shade"]/n,3))')"
done
