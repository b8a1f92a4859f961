B="python bench.py --no-cpu-baseline --no-extra-configs --min-time 2"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['ms_per_step'])"; }
$B | f base3
$B --streams 1 | f base1
