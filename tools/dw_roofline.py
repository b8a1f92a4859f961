"""Stand-alone depthwise kernels (LWP_FUSE_DWPW=0) and the stem at batch N: per-layer time and achieved algorithmic GB/s.
    python tools/dw_roofline.py [batch]"""
import os, sys
os.environ["LWP_FUSE_DWPW"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lwpose_amd  # noqa
from lwpose_amd import workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net, _ = workload.build_net(1, 1, 0, "fp32", 368, 656, calibrate=False)
eng = net.engine
h, w = 368, 656
tot_b = tot_t = 0.0
for l in eng.layers():
    hi, wi = h, w
    if l["stride"] == 2:
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if l["kind"] not in (0, 1):
        continue
    us = eng.time_layer(l["index"], B, 368, 656, 20) * 1e3
    if l["kind"] == 0:
        byt = B * (hi * wi * 3 + h * w * 32) * 4
    else:
        byt = B * (hi * wi * l["cin"] + h * w * l["cout"]) * 4 + 9 * l["cin"] * 4
        tot_b += byt; tot_t += us
    print("%-18s %4dx%-4d C=%-4d s%d d%d %8.1f us %7.0f GB/s  %.0f%% of 8 TB/s" % (l["name"], h, w, l["cin"], l["stride"], l["dilation"], us, byt / us / 1e3, byt / us / 1e3 / 80))
print("depthwise total: %.1f MB/frame, %.1f us, %.0f GB/s" % (tot_b / B / 1e6, tot_t, tot_b / tot_t / 1e3))
