#!/usr/bin/env python
"""Per-launch table of one pass (HIP events around each launch): time, achieved TFLOP/s or GB/s.
    python tools/profile_layers.py [--batch 1] [--nref 1] [--dtype fp32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import lwpose_amd  # noqa: F401,E402
from lwpose_amd import synth, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--nref", type=int, default=1)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--height", type=int, default=368)
ap.add_argument("--width", type=int, default=656)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
net, _ = workload.build_net(a.nref, 1, 0, a.dtype, a.height, a.width)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(a.batch, a.height, a.width))).cuda()
eng = net.engine
for _ in range(3):
    eng.infer_poses_async(x); eng.fetch_poses()
rows = eng.profile_launches(x, a.reps)
layers = eng.layers()
h, w = a.height, a.width
eb = 4 if a.dtype == "fp32" else 2
tot = 0.0
print("%-40s %-4s %12s %9s %10s" % ("launch", "cls", "shape", "us", "rate"))
for i, (name, kc, ms) in enumerate(rows):
    tot += ms
    if i < len(layers):
        l = layers[i]
        if l["stride"] == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        m = a.batch * h * w
        if l["kind"] == 3:
            fl = 2.0 * m * l["macs_per_pixel"]
            rate = "%7.1f TF" % (fl / (ms * 1e-3) / 1e12)
        elif l["kind"] == 2:
            fl = 2.0 * m * l["macs_per_pixel"]
            rate = "%7.1f TF" % (fl / (ms * 1e-3) / 1e12)
        elif l["kind"] == 1:
            by = (m * l["stride"] ** 2 * l["cin"] + m * l["cout"] + 9 * l["cin"]) * eb
            rate = "%7.0f GB/s" % (by / (ms * 1e-3) / 1e9)
        else:
            rate = ""
        shape = "%dx%d %d>%d k%d" % (h, w, l["cin"], l["cout"], l["ksize"])
    else:
        rate, shape = "", ""
    print("%-40s %-4d %12s %9.1f %10s" % (name, kc, shape, ms * 1e3, rate))
print("sum of launches: %.1f us" % (tot * 1e3))
