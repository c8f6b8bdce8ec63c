"""Config arithmetic shared by the model mirrors (semantics of models/mn/utils.py:8-26,
models/dymn/utils.py:5-23 and helpers/utils.py:1-32 of the reference)."""
import math

import numpy as np

_MN_WIDTHS = {"mn01": 0.1, "mn02": 0.2, "mn04": 0.4, "mn05": 0.5, "mn06": 0.6, "mn08": 0.8, "mn10": 1.0,
              "mn12": 1.2, "mn14": 1.4, "mn16": 1.6, "mn20": 2.0, "mn30": 3.0, "mn40": 4.0}
_DYMN_WIDTHS = {"dymn04": 0.4, "dymn10": 1.0, "dymn20": 2.0}


def NAME_TO_WIDTH(name):
    """'mn10_as' -> 1.0, 'dymn20_as(2)' -> 2.0; unknown names map to 1.0 like the reference."""
    if name.startswith("dymn"):
        return _DYMN_WIDTHS.get(name[:6], 1.0)
    return _MN_WIDTHS.get(name[:4], 1.0)


def make_divisible(v, divisor, min_value=None):
    """Round to a multiple of ``divisor`` (>= min_value) without losing more than 10 %."""
    floor = divisor if min_value is None else min_value
    rounded = max(floor, int(v + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < 0.9 * v else rounded


def cnn_out_size(in_size, padding, dilation, kernel, stride):
    return math.floor((in_size + 2 * padding - dilation * (kernel - 1) - 1) / stride + 1)


def exp_rampup(rampup_length):
    """helpers/utils.py:35-46: exp(-5 (1 - e/L)^2) for e < L (e clipped to >= 0.5), then 1."""
    def f(epoch):
        if epoch >= rampup_length:
            return 1.0
        phase = 1.0 - float(np.clip(epoch, 0.5, rampup_length)) / rampup_length
        return float(np.exp(-5.0 * phase * phase))
    return f


def linear_rampdown(rampdown_length, start=0, last_value=0):
    """helpers/utils.py:49-58."""
    def f(epoch):
        if epoch <= start:
            return 1.0
        if epoch - start >= rampdown_length:
            return last_value
        return last_value + (1.0 - last_value) * (rampdown_length - epoch + start) / rampdown_length
    return f


def exp_warmup_linear_down(warmup, rampdown_length, start_rampdown, last_value):
    """helpers/utils.py:61-66: the per-epoch LambdaLR factor of ex_audioset.py:93-96."""
    up, down = exp_rampup(warmup), linear_rampdown(rampdown_length, start_rampdown, last_value)
    return lambda epoch: up(epoch) * down(epoch)
