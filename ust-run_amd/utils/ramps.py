"""Ramp-up schedules (host scalars).  Same surface as the reference's utils/ramps.py:19-41."""
import math


def sigmoid_rampup(current, rampup_length):
    """Exponential rampup exp(-5 (1 - t/T)^2) with t clipped to [0, T]; 1.0 when T == 0 (ramps.py:19-26)."""
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length))
    phase = 1.0 - t / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def linear_rampup(current, rampup_length):
    """Linear rampup (ramps.py:29-35)."""
    assert current >= 0 and rampup_length >= 0
    if current >= rampup_length:
        return 1.0
    return current / rampup_length


def cosine_rampdown(current, rampdown_length):
    """Cosine rampdown (ramps.py:38-41)."""
    assert 0 <= current <= rampdown_length
    return float(.5 * (math.cos(math.pi * current / rampdown_length) + 1))
