for f in variants/*.so; do
  echo "== $f"
  PTAMD_LIB=$PWD/$f timeout -k 10 300 python3 profiles/other_configs.py --traversal wide 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config'][:24], 'ms', d['render_ms'], 'extend', d['extend_ms'], 'redone', d['rays_redone'])
"
done
