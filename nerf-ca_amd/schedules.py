"""Host-side schedules of the training loop (tiny, scalar work; no device arithmetic)."""
import numpy as np
import torch


def freq_mask(L: int, current_iter: int, max_iter: int, window_start: int):
    """FreeNeRF band mask (model/CPPN.py:144-159).  Returns (f32[L] mask, windowed_alpha).

    Bands below the moving pointer are 1, the band under it carries the fractional part, the rest
    are 1e-8 (the clip's lower bound -- not zero); after ``max_iter`` everything is 1.
    """
    if current_iter >= max_iter:
        return torch.ones(L).float(), L + 1
    pointer = (L * current_iter) / max_iter + window_start
    whole = int(pointer)
    m = np.zeros(L)
    m[:whole + 1] = 1.0
    m[whole:whole + 1] = pointer - whole
    return torch.clip(torch.from_numpy(m), 1e-8, 1 - 1e-8).float(), pointer


def nerfies_window(L: int, alpha) -> torch.Tensor:
    """Cosine-eased window of Nerfies (model/CPPN.py:137-142)."""
    t = torch.clip(alpha - torch.arange(0, L), 0.0, 1.0)
    return 0.5 * (1 + torch.cos(torch.pi * t + torch.pi))


def linear_param_decay(curr_iter, start_weight, end_weight, steps, delay_steps=0):
    """train/model_helpers.py:264-269."""
    if curr_iter < delay_steps:
        return 0
    a = min((curr_iter - delay_steps) / steps, 1.0)
    return (1.0 - a) * start_weight + a * end_weight


def exp_param_decay(curr_iter, start_weight, end_weight, steps, delay_steps=0):
    """train/model_helpers.py:271-282."""
    if curr_iter < delay_steps:
        return 0
    if start_weight == end_weight:
        return start_weight
    if curr_iter >= steps:
        return end_weight
    return start_weight * (end_weight / start_weight) ** (curr_iter / (steps - 1))
