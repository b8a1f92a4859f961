"""1-Euro smoothing of tracked key-point coordinates, with the reference's public surface
(modules/one_euro_filter.py:4-43): ``get_alpha``, ``LowPassFilter`` (``x_previous``) and ``OneEuroFilter``
(``freq, mincutoff, beta, dcutoff, filter_x, filter_dx, x_previous, dx``).

A first-order low-pass whose cut-off grows with the (low-passed) speed of the signal:
    alpha(cutoff) = 1 / (1 + (1 / (2*pi*cutoff)) / (1 / freq))
    speed_hat     = lowpass(speed, alpha(dcutoff))
    x_hat         = lowpass(x, alpha(mincutoff + beta * |speed_hat|))
The output sequence is pinned to the reference's by tests/golden/one_euro.json.
"""
import math

__all__ = ["get_alpha", "LowPassFilter", "OneEuroFilter"]


def get_alpha(rate=30, cutoff=1):
    time_constant = 1 / (2 * math.pi * cutoff)
    period = 1 / rate
    return 1 / (1 + time_constant / period)


class LowPassFilter:
    """y[n] = alpha * x[n] + (1 - alpha) * y[n-1]; the very first sample passes through unchanged."""
    __slots__ = ("x_previous",)

    def __init__(self):
        self.x_previous = None          # last OUTPUT (the reference keeps the filtered value under this name)

    def __call__(self, x, alpha=0.5):
        self.x_previous = x if self.x_previous is None else alpha * x + (1 - alpha) * self.x_previous
        return self.x_previous


class OneEuroFilter:
    __slots__ = ("freq", "mincutoff", "beta", "dcutoff", "filter_x", "filter_dx", "x_previous", "dx")

    def __init__(self, freq=15, mincutoff=1, beta=0.05, dcutoff=1):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.filter_x, self.filter_dx = LowPassFilter(), LowPassFilter()
        self.x_previous = None          # last RAW sample
        self.dx = None                  # last raw speed

    def __call__(self, x):
        self.dx = 0 if self.dx is None else (x - self.x_previous) * self.freq
        speed = self.filter_dx(self.dx, get_alpha(self.freq, self.dcutoff))
        smoothed = self.filter_x(x, get_alpha(self.freq, self.mincutoff + self.beta * abs(speed)))
        self.x_previous = x
        return smoothed
