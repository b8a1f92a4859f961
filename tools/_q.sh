python tools/layer_sweep.py 32 bf16 "X=1;LWP_GEMMH_1X1=0" cpm.align refinement_stages.0.trunk.0.initial refinement_stages.0.trunk.1.initial
B="python bench.py --no-cpu-baseline --no-extra-configs --min-time 2"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), d['ms_per_step'])"; }
$B --batch 32 --dtype bf16 --streams 2 | f bf16_b32_s2
