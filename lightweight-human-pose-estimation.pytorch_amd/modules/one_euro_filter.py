"""1-Euro filter used by pose tracking (reference: modules/one_euro_filter.py:4-43)."""
import math


def get_alpha(rate=30, cutoff=1):
    return 1 / (1 + (1 / (2 * math.pi * cutoff)) / (1 / rate))


class LowPassFilter:
    def __init__(self):
        self.x_previous = None

    def __call__(self, x, alpha=0.5):
        if self.x_previous is not None:
            x = alpha * x + (1 - alpha) * self.x_previous
        self.x_previous = x
        return x


class OneEuroFilter:
    def __init__(self, freq=15, mincutoff=1, beta=0.05, dcutoff=1):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.filter_x, self.filter_dx = LowPassFilter(), LowPassFilter()
        self.x_previous = None
        self.dx = None

    def __call__(self, x):
        self.dx = 0 if self.dx is None else (x - self.x_previous) * self.freq
        dx_smoothed = self.filter_dx(self.dx, get_alpha(self.freq, self.dcutoff))
        cutoff = self.mincutoff + self.beta * abs(dx_smoothed)
        x_filtered = self.filter_x(x, get_alpha(self.freq, cutoff))
        self.x_previous = x
        return x_filtered
