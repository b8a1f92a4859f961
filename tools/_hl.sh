#!/bin/bash
# headline value under env settings: hl.sh "ENV=V ENV2=V" ...
for cfg in "$@"; do
  v=$(env $cfg python bench.py --steps 300 --warmup 30 --min-time 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d.get('single_stream_frames_per_s') or 0))")
  echo "$cfg -> $v"
done
