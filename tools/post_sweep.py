"""per-kernel time of the post-processing chain (HIP events) under a list of env settings:
    post_sweep.py batch dtype 'ENV=V|ENV2=V;...'"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, lwpose_amd
from lwpose_amd import synth, workload
net, _ = workload.build_net(1, 1, 0, sys.argv[2], 368, 656)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(int(sys.argv[1]), 368, 656))).cuda()
eng = net.engine
for _ in range(3):
    eng.infer_poses_async(x); eng.fetch_poses()
rows = eng.profile_launches(x, 20)
post = [(n, ms * 1e3) for n, kc, ms in rows if kc == 4]
print("RESULT " + json.dumps(post))
''' % ROOT
batch, dtype, cfgs = sys.argv[1], sys.argv[2], sys.argv[3].split(";")
tab = {}
for cfg in cfgs:
    env = dict(os.environ)
    for kv in cfg.split("|"):
        if "=" in kv:
            k, v = kv.split("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, batch, dtype], capture_output=True, text=True, env=env, timeout=400)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    tab[cfg] = dict(json.loads(line[0][7:])) if line else {}
    if not line: print(cfg, (r.stderr or r.stdout)[-300:])
names = ["find_peaks", "nms", "score_pairs", "match", "assemble"]
print("%-14s" % "kernel (us)" + "".join("%24s" % c[-23:] for c in cfgs))
for nm in names:
    print("%-14s" % nm + "".join("%24.1f" % tab[c].get(nm, float("nan")) for c in cfgs))
