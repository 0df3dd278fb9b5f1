#!/bin/bash
# Here, after profiles/dev/final_measurements.sh <tag> has run on the GPU box: copies what the tables are made from
# out of gpurun_out/ into profiles/ (gpurun_out/ keeps the traces of earlier runs too: the newest file is the one meant).  usage: profiles/dev/publish_measurements.sh <tag>
T=${1:-r03}
for m in wide ordered reference; do
  d=gpurun_out/${T}_$m
  cp $d/pmc_summary.json profiles/${T}_${m}_pmc_summary.json
  cp $d/pmc_summary.txt profiles/${T}_${m}_pmc_summary.txt
  cp $(ls -t $d/trace/*/*kernel_stats.csv | head -1) profiles/${T}_${m}_bench_kernel_stats.csv
  grep "^{" gpurun_out/${T}_bench_$m.json | tail -1 > profiles/${T}_bench_$m.json
done
for s in knot cornell soup; do for m in wide ordered reference; do
  d=gpurun_out/${T}_${s}_$m; [ -f $d/pmc_summary.json ] || continue
  cp $d/pmc_summary.json profiles/${s}_${T}_${m}_pmc_summary.json
  cp $d/pmc_summary.txt profiles/${s}_${T}_${m}_pmc_summary.txt
  cp $(ls -t $d/trace/*/*kernel_stats.csv | head -1) profiles/${s}_${T}_${m}_kernel_stats.csv
  grep -h '"scene"' $d/trace.log | tail -1 > profiles/${s}_${T}_${m}_render.json
done; done
grep "^{" gpurun_out/${T}_other_configs.jsonl > profiles/${T}_other_configs.jsonl
cp gpurun_out/${T}_valu_rate.txt profiles/
for f in cli cli_soup; do [ -f gpurun_out/${T}_$f.log ] && grep -v amdgpu.ids gpurun_out/${T}_$f.log > profiles/${T}_$f.log; done
python3 profiles/config_table.py $T
