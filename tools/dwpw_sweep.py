import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import lwpose_amd
from lwpose_amd import workload
net, _ = workload.build_net(1, 1, 0, "fp32", 368, 656, calibrate=False)
eng = net.engine
names = {l["name"]: l["index"] for l in eng.layers()}
out = {nm: eng.time_layer(names[nm], int(sys.argv[1]), 368, 656, 60) * 1e3 for nm in sys.argv[2:]}
print("RESULT " + json.dumps(out))
''' % ROOT
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = ["model.7.pw", "model.6.pw", "model.5.pw", "model.4.pw", "model.3.pw", "model.2.pw", "model.1.pw", "cpm.trunk.0.pw"]
cfgs = [("full", "", ""), ("nophase1", "1", ""), ("noB", "2", ""), ("noMFMA", "4", ""), ("noP1noB", "3", ""), ("onlyP1", "6", ""), ("none", "7", "")]
tab = {}
for name, bm, nw in cfgs:
    env = dict(os.environ)
    if bm: env["LWP_DWPW_DEBUG"] = bm
    r = subprocess.run([sys.executable, "-c", CHILD, str(batch)] + L, capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    tab[name] = json.loads(line[0][7:]) if line else {}
    if not line: print(name, (r.stderr or r.stdout)[-300:])
print("%-20s" % "layer (us)" + "".join("%9s" % c[0] for c in cfgs))
for nm in L:
    print("%-20s" % nm + "".join("%9.1f" % tab[c[0]].get(nm, float("nan")) for c in cfgs))
