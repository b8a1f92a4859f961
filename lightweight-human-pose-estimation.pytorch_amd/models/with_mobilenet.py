"""Drop-in for the reference's ``models.with_mobilenet.PoseEstimationWithMobileNet``
(reference: models/with_mobilenet.py:89-123), backed by the HIP engine.

Same constructor, ``state_dict`` keys/shapes, ``load_state_dict``, ``eval``/``cuda`` and call
convention: ``net(x)`` with x (N,3,H,W) float32 returns the list [heat0, paf0, heat1, paf1, ...]
of (N,{19,38},H/8,W/8) tensors on x's device.  There is no CPU execution path: a CPU tensor is
staged to the GPU by the library and the results are copied back.
"""
from collections import OrderedDict

import numpy as np

from .. import _lib, synth
from ..runtime import Engine


class PoseEstimationWithMobileNet(object):
    def __init__(self, num_refinement_stages=1, num_channels=128, num_heatmaps=19, num_pafs=38, dtype="fp32"):
        import torch
        self.num_refinement_stages = num_refinement_stages
        self.num_channels, self.num_heatmaps, self.num_pafs = num_channels, num_heatmaps, num_pafs
        self.dtype = {"fp32": _lib.F32, "bf16": _lib.BF16}[dtype]
        # parameters start from a deterministic random init (the reference starts from torch's default init)
        self._state = synth.make_state_dict(num_refinement_stages, seed=0, num_channels=num_channels,
                                            num_heatmaps=num_heatmaps, num_pafs=num_pafs)
        self._engine = None
        self._dirty = True
        self._device_id = None
        self.training = False
        self._torch = torch

    # ---- nn.Module-like surface used by the reference's callers (demo.py:82-84,156-158; val.py:114,174-178)
    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._state.items())

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._state if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._state]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict: missing %s unexpected %s" % (missing, unexpected))
        for k in self._state:
            if k in state_dict:
                v = state_dict[k]
                v = v.detach().cpu() if hasattr(v, "detach") else self._torch.as_tensor(np.asarray(v))
                if tuple(v.shape) != tuple(self._state[k].shape):
                    raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(v.shape), tuple(self._state[k].shape)))
                self._state[k] = v.to(self._state[k].dtype).clone()
        self._dirty = True

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("lwpose_amd is an inference path; training is out of scope")
        return self

    def cuda(self, device=None):
        idx = 0 if device is None else (device if isinstance(device, int) else self._torch.device(device).index or 0)
        self._ensure(idx)
        return self

    def to(self, device):
        d = self._torch.device(device)
        if d.type != "cuda":
            raise RuntimeError("lwpose_amd has no CPU execution path")
        return self.cuda(d.index or 0)

    def cpu(self):
        raise RuntimeError("lwpose_amd has no CPU execution path")

    @property
    def engine(self):
        self._ensure(self._device_id if self._device_id is not None else 0)
        return self._engine

    def _ensure(self, device_id):
        if self._engine is None or self._device_id != device_id:
            self._engine = Engine(device_id, self.num_refinement_stages, self.num_channels, self.num_heatmaps,
                                  self.num_pafs, self.dtype)
            self._device_id = device_id
            self._dirty = True
        if self._dirty:
            self._engine.load_state_dict(self._state)
            self._dirty = False

    def forward(self, x):
        dev = x.device.index if getattr(x, "is_cuda", False) else (self._device_id if self._device_id is not None else 0)
        self._ensure(dev)
        return self._engine.forward(x)

    __call__ = forward
