#!/bin/bash
# headline value (and the single-stream figure) of bench.py under environment settings, one line per setting: headline_sweep.sh "ENV=V ENV2=V" ... (HIP box only)
for cfg in "$@"; do
  v=$(env $cfg python bench.py --steps 300 --warmup 30 --min-time 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d.get('single_stream_frames_per_s') or 0))")
  echo "$cfg -> $v"
done
