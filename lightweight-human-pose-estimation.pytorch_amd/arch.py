"""Parameter table of the Lightweight-OpenPose network (host-side mirror).

The table lists every ``state_dict`` entry of the reference
``PoseEstimationWithMobileNet`` (reference: models/with_mobilenet.py:89-112,
modules/conv.py:4-32) in registration order, with its shape and role.  It is
what ``load_state`` / ``state_dict`` iterate and what the synthetic-weight
generator fills.  The C-ABI library carries the same table in C++
(csrc/net_graph.cpp, ``lwp_param_spec``); tests check that both agree with the
key list captured from the reference (tests/golden/state_dict_keys_*.json).
"""
from collections import namedtuple

Param = namedtuple("Param", "key shape role fan_in")
# role: conv_w | conv_b | bn_w | bn_b | bn_mean | bn_var | bn_nbt


def _conv(prefix, cin, cout, k, groups=1, bias=True):
    fan_in = (cin // groups) * k * k
    out = [Param(prefix + ".weight", (cout, cin // groups, k, k), "conv_w", fan_in)]
    if bias:
        out.append(Param(prefix + ".bias", (cout,), "conv_b", fan_in))
    return out


def _bn(prefix, c):
    return [Param(prefix + ".weight", (c,), "bn_w", 0),
            Param(prefix + ".bias", (c,), "bn_b", 0),
            Param(prefix + ".running_mean", (c,), "bn_mean", 0),
            Param(prefix + ".running_var", (c,), "bn_var", 0),
            Param(prefix + ".num_batches_tracked", (), "bn_nbt", 0)]


BACKBONE = [  # (cin, cout, stride, dilation) of the 11 conv_dw blocks, with_mobilenet.py:94-104
    (32, 64, 1, 1), (64, 128, 2, 1), (128, 128, 1, 1), (128, 256, 2, 1),
    (256, 256, 1, 1), (256, 512, 1, 1), (512, 512, 1, 2), (512, 512, 1, 1),
    (512, 512, 1, 1), (512, 512, 1, 1), (512, 512, 1, 1)]


def param_table(num_refinement_stages=1, num_channels=128, num_heatmaps=19, num_pafs=38):
    C, NH, NP = num_channels, num_heatmaps, num_pafs
    t = []
    # backbone: stem conv + 11 depthwise-separable blocks
    t += _conv("model.0.0", 3, 32, 3, bias=False) + _bn("model.0.1", 32)
    for i, (cin, cout, _s, _d) in enumerate(BACKBONE, start=1):
        t += _conv("model.%d.0" % i, cin, cin, 3, groups=cin, bias=False) + _bn("model.%d.1" % i, cin)
        t += _conv("model.%d.3" % i, cin, cout, 1, bias=False) + _bn("model.%d.4" % i, cout)
    # cpm
    t += _conv("cpm.align.0", 512, C, 1)
    for j in range(3):
        t += _conv("cpm.trunk.%d.0" % j, C, C, 3, groups=C, bias=False)
        t += _conv("cpm.trunk.%d.2" % j, C, C, 1, bias=False)
    t += _conv("cpm.conv.0", C, C, 3)
    # initial stage
    for j in range(3):
        t += _conv("initial_stage.trunk.%d.0" % j, C, C, 3)
    t += _conv("initial_stage.heatmaps.0.0", C, 512, 1) + _conv("initial_stage.heatmaps.1.0", 512, NH, 1)
    t += _conv("initial_stage.pafs.0.0", C, 512, 1) + _conv("initial_stage.pafs.1.0", 512, NP, 1)
    # refinement stages
    for k in range(num_refinement_stages):
        p = "refinement_stages.%d" % k
        for b in range(5):
            cin = C + NH + NP if b == 0 else C
            q = "%s.trunk.%d" % (p, b)
            t += _conv(q + ".initial.0", cin, C, 1)
            t += _conv(q + ".trunk.0.0", C, C, 3) + _bn(q + ".trunk.0.1", C)
            t += _conv(q + ".trunk.1.0", C, C, 3) + _bn(q + ".trunk.1.1", C)
        t += _conv(p + ".heatmaps.0.0", C, C, 1) + _conv(p + ".heatmaps.1.0", C, NH, 1)
        t += _conv(p + ".pafs.0.0", C, C, 1) + _conv(p + ".pafs.1.0", C, NP, 1)
    return t
