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
by_name = {l["name"]: i for i, l in enumerate(layers)}
dims, hh, ww = [], a.height, a.width            # output map of every layer
for l in layers:
    if l["stride"] == 2:
        hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    dims.append((hh, ww))
for name, kc, ms in rows:
    tot += ms
    if name in by_name:
        li = by_name[name]
        l = layers[li]
        h, w = dims[li]
        m = a.batch * h * w
        macs, cout, label = l["macs_per_pixel"], l["cout"], name
        if name.endswith(".heads.0") and ms > 0 and li + 1 < len(layers) and not any(r[0] == layers[li + 1]["name"] for r in rows):
            macs += layers[li + 1]["macs_per_pixel"]          # fused head pair: one launch covers both 1x1 convs
            cout, label = layers[li + 1]["cout"], name[:-2] + "{0,1}"
        elif l["ksize"] == 3 and li + 1 < len(layers) and layers[li + 1]["ksize"] == 1 and not any(r[0] == layers[li + 1]["name"] for r in rows):
            macs += layers[li + 1]["macs_per_pixel"]          # the next block's `initial` 1x1 rides in this 3x3's epilogue
            label = name + " +1x1"
        if l["kind"] in (2, 3):
            rate = "%7.1f TF" % (2.0 * m * macs / (ms * 1e-3) / 1e12)
        elif l["kind"] == 1:
            by = (m * l["stride"] ** 2 * l["cin"] + m * l["cout"] + 9 * l["cin"]) * eb
            rate = "%7.0f GB/s" % (by / (ms * 1e-3) / 1e9)
        else:
            rate = ""
        shape = "%dx%d %d>%d k%d" % (h, w, l["cin"], cout, l["ksize"])
        name = label
    else:
        rate, shape = "", ""
    print("%-40s %-4d %12s %9.1f %10s" % (name, kc, shape, ms * 1e3, rate))
print("sum of launches: %.1f us" % (tot * 1e3))
