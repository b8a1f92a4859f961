python tools/layer_sweep.py 32 bf16 "X=1;LWP_DWPWH_PERSIST=0" model.1.pw model.2.pw model.3.pw model.4.pw model.5.pw model.6.pw model.7.pw model.8.pw cpm.trunk.0.pw
