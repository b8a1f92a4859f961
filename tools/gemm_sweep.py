#!/usr/bin/env python
"""A/B tile configurations of the implicit-GEMM conv on the real layer shapes (one process per config,
because the override is read from the environment once).  usage: gemm_sweep.py [batch]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import lwpose_amd
from lwpose_amd import workload
batch = int(sys.argv[1])
net, _ = workload.build_net(1, 1, 0, "fp32", 368, 656, calibrate=False)
eng = net.engine
names = {l["name"]: l["index"] for l in eng.layers()}
out = {}
for nm in sys.argv[2:]:
    out[nm] = eng.time_layer(names[nm], batch, 368, 656, 60) * 1e3
print("RESULT " + json.dumps(out))
''' % ROOT

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C3 = ["initial_stage.trunk.0", "refinement_stages.0.trunk.0.trunk.1"]
PW = ["model.7.pw", "model.5.pw", "model.3.pw", "model.1.pw", "cpm.align", "initial_stage.heatmaps.0", "initial_stage.heatmaps.1",
      "refinement_stages.0.trunk.1.initial", "refinement_stages.0.pafs.1"]
cfgs = ["heur", "32,64,1", "32,64,2", "32,64,4", "64,64,1", "64,64,2", "64,64,4", "64,128,1", "64,128,2", "128,128,1", "32,32,4", "32,32,8"]
table = {}
for cfg in cfgs:
    env = dict(os.environ)
    if cfg != "heur":
        env["LWP_GEMM_C3"] = cfg
        env["LWP_GEMM_PW"] = cfg
    r = subprocess.run([sys.executable, "-c", CHILD, str(batch)] + C3 + PW, capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    table[cfg] = json.loads(line[0][7:]) if line else {"error": (r.stderr or r.stdout)[-300:]}
print("%-40s" % "layer (us per launch)" + "".join("%10s" % c for c in cfgs))
for nm in C3 + PW:
    print("%-40s" % nm + "".join("%10.1f" % table[c].get(nm, float("nan")) for c in cfgs))
for c in cfgs:
    if "error" in table[c]:
        print(c, table[c]["error"])
