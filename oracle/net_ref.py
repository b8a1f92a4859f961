"""ORACLE — test infrastructure, NOT product code.

CPU restatement of the reference network forward, written as straight-line functional
calls on a ``state_dict`` (no nn.Module), fp32, stock ``torch.nn.functional`` CPU kernels.
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import it.

Follows (reference file:line):
  modules/conv.py:4-10    conv      = Conv2d(+BatchNorm2d eval)(+ReLU)
  modules/conv.py:13-22   conv_dw   = dw3x3+BN+ReLU, pw1x1+BN+ReLU
  modules/conv.py:25-32   conv_dw_no_bn = dw3x3+ELU, pw1x1+ELU
  models/with_mobilenet.py:7-21    Cpm
  models/with_mobilenet.py:24-45   InitialStage
  models/with_mobilenet.py:48-60   RefinementStageBlock
  models/with_mobilenet.py:63-86   RefinementStage
  models/with_mobilenet.py:89-123  PoseEstimationWithMobileNet

Pinning: tests/test_oracle_golden.py checks this file against outputs captured from the
reference itself (oracle/make_golden.py, fixtures tests/golden/net_*.npz).
"""
import torch
import torch.nn.functional as F

_BACKBONE = [(1, 1), (2, 1), (1, 1), (2, 1), (1, 1), (1, 1), (1, 2), (1, 1), (1, 1), (1, 1), (1, 1)]  # (stride, dilation)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.1, 1e-5)


def _conv(x, sd, p, stride=1, pad=0, dil=1, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, pad, dil, groups)


def _c3(x, sd, p, dil=1, bn=None):
    y = _conv(x, sd, p, 1, dil, dil)
    if bn is not None:
        y = _bn(y, sd, bn)
    return F.relu(y)


def _heads(x, sd, p):
    h = _conv(F.relu(_conv(x, sd, p + ".heatmaps.0.0")), sd, p + ".heatmaps.1.0")
    q = _conv(F.relu(_conv(x, sd, p + ".pafs.0.0")), sd, p + ".pafs.1.0")
    return [h, q]


def forward(sd, x, num_refinement_stages=1, taps=None):
    """x: (N,3,H,W) f32 -> [heat0, paf0, heat1, paf1, ...] like the reference forward.
    ``taps``: optional dict filled with intermediate activations (NCHW) for per-layer parity tests."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    with torch.no_grad():
        x = F.relu(_bn(_conv(x, sd, "model.0.0", 2, 1), sd, "model.0.1"))
        tap("model.0", x)
        for i, (s, d) in enumerate(_BACKBONE, start=1):
            c = x.shape[1]
            x = F.relu(_bn(_conv(x, sd, "model.%d.0" % i, s, d, d, c), sd, "model.%d.1" % i))
            tap("model.%d.dw" % i, x)
            x = F.relu(_bn(_conv(x, sd, "model.%d.3" % i), sd, "model.%d.4" % i))
            tap("model.%d" % i, x)
        # Cpm
        a = F.relu(_conv(x, sd, "cpm.align.0"))
        tap("cpm.align", a)
        t = a
        for j in range(3):
            t = F.elu(_conv(t, sd, "cpm.trunk.%d.0" % j, 1, 1, 1, t.shape[1]))
            tap("cpm.trunk.%d.dw" % j, t)
            t = F.elu(_conv(t, sd, "cpm.trunk.%d.2" % j))
            tap("cpm.trunk.%d" % j, t)
        tap("cpm.sum", a + t)          # x + trunk(x) (with_mobilenet.py:20): the operand of cpm.conv
        feat = _c3(a + t, sd, "cpm.conv.0")
        tap("cpm", feat)
        # initial stage
        t = feat
        for j in range(3):
            t = _c3(t, sd, "initial_stage.trunk.%d.0" % j)
            tap("initial_stage.trunk.%d" % j, t)
        outs = _heads(t, sd, "initial_stage")
        # refinement stages
        for k in range(num_refinement_stages):
            p = "refinement_stages.%d" % k
            t = torch.cat([feat, outs[-2], outs[-1]], 1)
            for b in range(5):
                q = "%s.trunk.%d" % (p, b)
                ini = F.relu(_conv(t, sd, q + ".initial.0"))
                u = _c3(ini, sd, q + ".trunk.0.0", 1, q + ".trunk.0.1")
                u = _c3(u, sd, q + ".trunk.1.0", 2, q + ".trunk.1.1")
                t = ini + u
                tap(q, t)
            outs.extend(_heads(t, sd, p))
    return outs
