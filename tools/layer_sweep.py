"""time selected layers under a list of env settings: layer_sweep.py batch dtype 'ENV=V|ENV2=V;...' layer...
(assignments of one setting are separated by '|', settings by ';')"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import lwpose_amd
from lwpose_amd import workload
net, _ = workload.build_net(1, 1, 0, sys.argv[2], 368, 656, calibrate=False)
eng = net.engine
names = {l["name"]: l["index"] for l in eng.layers()}
out = {nm: eng.time_layer(names[nm], int(sys.argv[1]), 368, 656, 30) * 1e3 for nm in sys.argv[3:]}
print("RESULT " + json.dumps(out))
''' % ROOT
batch, dtype, cfgs, layers = sys.argv[1], sys.argv[2], sys.argv[3].split(";"), sys.argv[4:]
tab = {}
for cfg in cfgs:
    env = dict(os.environ)
    for kv in cfg.split("|"):
        if "=" in kv:
            k, v = kv.split("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, batch, dtype] + layers, capture_output=True, text=True, env=env, timeout=400)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    tab[cfg] = json.loads(line[0][7:]) if line else {}
    if not line: print(cfg, (r.stderr or r.stdout)[-300:])
print("%-28s" % "layer (us)" + "".join("%22s" % c[-21:] for c in cfgs))
for nm in layers:
    print("%-28s" % nm + "".join("%22.1f" % tab[c].get(nm, float("nan")) for c in cfgs))
