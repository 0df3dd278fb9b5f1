#!/bin/bash
# On the GPU box: the 20 M triangle soup (wide traversal) once per library in variants/, closest-hit time per frame.
for f in variants/*.so; do
  PTAMD_LIB=$PWD/$f timeout -k 10 300 python3 profiles/other_configs.py --traversal wide 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$f', d['config'][:12], d['mrays_per_s'], 'Mrays/s, closest-hit ms', d['extend_ms'], 'redone', d['rays_redone'])" || exit 1
done
