python tools/layer_sweep.py 1 fp32 "X=1" model.1.pw model.2.pw model.3.pw model.4.pw model.5.pw model.6.pw model.8.pw cpm.trunk.0.pw
python tools/layer_sweep.py 32 fp32 "X=1" model.1.pw model.2.pw model.3.pw model.4.pw model.5.pw model.6.pw model.8.pw cpm.trunk.0.pw
B="python bench.py --no-cpu-baseline --no-extra-configs --min-time 2"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['ms_per_step'])"; }
$B | f base3
$B --streams 1 | f base1
$B --batch 32 --streams 2 | f fp32_b32
