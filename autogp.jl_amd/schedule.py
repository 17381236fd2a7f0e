"""Data-annealing schedules (src/Schedule.jl:24-84) — only used to generate the n-sequences of
the benchmark configs; the SMC driver itself is out of scope."""
from __future__ import annotations


def linear_schedule(n: int, percent: float):
    """src/Schedule.jl:24-39."""
    assert 0 < n and 0 < percent < 1
    step = int(round(percent * n))
    checkpoints = list(range(step, n + 1, step))
    remaining = n - checkpoints[-1]
    assert 0 <= remaining < step
    if remaining == 0:
        return checkpoints
    if remaining < step / 2:
        checkpoints[-1] = n
        return checkpoints
    return checkpoints + [n]
