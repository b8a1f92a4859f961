"""1-Euro smoothing of tracked key-point coordinates (public names as in the reference's
modules/one_euro_filter.py:4-43; state is kept as plain floats in one object).

A first-order low-pass whose cut-off grows with the (low-passed) speed of the signal:
    alpha(cutoff) = 1 / (1 + (1 / (2*pi*cutoff)) / (1 / freq))
    speed_hat     = lowpass(speed, alpha(dcutoff))
    x_hat         = lowpass(x, alpha(mincutoff + beta * |speed_hat|))
"""
import math

__all__ = ["get_alpha", "OneEuroFilter"]


def get_alpha(rate=30, cutoff=1):
    time_constant = 1 / (2 * math.pi * cutoff)
    period = 1 / rate
    return 1 / (1 + time_constant / period)


def _blend(new, old, alpha):
    """One low-pass step; the very first sample passes through unchanged."""
    return new if old is None else alpha * new + (1 - alpha) * old


class OneEuroFilter:
    __slots__ = ("freq", "mincutoff", "beta", "dcutoff", "_raw", "_smooth", "_speed")

    def __init__(self, freq=15, mincutoff=1, beta=0.05, dcutoff=1):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self._raw = None       # previous raw sample
        self._smooth = None    # previous filtered sample
        self._speed = None     # previous filtered speed

    def __call__(self, x):
        speed = 0 if self._raw is None else (x - self._raw) * self.freq
        self._speed = _blend(speed, self._speed, get_alpha(self.freq, self.dcutoff))
        self._smooth = _blend(x, self._smooth, get_alpha(self.freq, self.mincutoff + self.beta * abs(self._speed)))
        self._raw = x
        return self._smooth
