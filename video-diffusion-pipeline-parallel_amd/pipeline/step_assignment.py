"""Which diffusion steps does each pipeline rank own?

Torch-free on purpose (like the reference module) so the arithmetic is testable without a
process group.

Mirrors ``/root/reference/src/pipeline/step_assignment.py``:
  * ``StepRange``      – ref ``:12-32`` (frozen ``[start, end)``; ``count``; iterable; ValueError
    on negative bounds or ``end < start``)
  * ``assign_steps``   – ref ``:35-69`` (uniform contiguous split; ValueError on
    ``total_steps <= 0``, ``world_size <= 0``, rank outside ``[0, world_size)``, or
    ``total_steps % world_size != 0``)

Extension (NOT in the reference, separately named so the strict API above is unchanged):
  * ``assign_steps_balanced`` – contiguous split that tolerates a remainder; the first
    ``total_steps % world_size`` ranks take one extra step.  Needed for the 25-step
    schedule on 2/4/8 GPUs that BASELINE.json asks for (25 -> [4,3,3,3,3,3,3,3]).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator


@dataclass(frozen=True)
class StepRange:
    """Half-open interval ``[start, end)`` of schedule positions owned by one rank."""

    start: int
    end: int

    def __post_init__(self) -> None:
        if min(self.start, self.end) < 0:
            raise ValueError("Step indices must be non-negative.")
        if self.start > self.end:
            raise ValueError("Step range end must be >= start.")

    @property
    def count(self) -> int:
        return self.end - self.start

    def __iter__(self) -> Iterator[int]:
        return iter(range(self.start, self.end))


def _validate(total_steps: int, world_size: int, rank: int) -> None:
    if total_steps <= 0:
        raise ValueError("total_steps must be positive.")
    if world_size <= 0:
        raise ValueError("world_size must be positive.")
    if rank < 0 or rank >= world_size:
        raise ValueError("rank must satisfy 0 <= rank < world_size.")


def assign_steps(total_steps: int, world_size: int, rank: int) -> StepRange:
    """Uniform contiguous split: rank ``r`` owns ``[r*k, (r+1)*k)`` with ``k = T // N``.

    Same contract as the reference (``step_assignment.py:53-69``): a schedule that does not
    divide evenly is rejected rather than silently rebalanced.
    """

    _validate(total_steps, world_size, rank)
    per_rank, remainder = divmod(total_steps, world_size)
    if remainder:
        raise ValueError(
            "total_steps must be divisible by world_size for uniform step assignment."
        )
    first = rank * per_rank
    return StepRange(start=first, end=first + per_rank)


def assign_steps_balanced(total_steps: int, world_size: int, rank: int) -> StepRange:
    """Contiguous split allowing a remainder (extension; see module docstring).

    Ranks ``0 .. (T % N) - 1`` own ``T // N + 1`` steps, the rest ``T // N``.  Requires
    ``total_steps >= world_size`` so no stage is empty.
    """

    _validate(total_steps, world_size, rank)
    if total_steps < world_size:
        raise ValueError("total_steps must be >= world_size so every rank owns a step.")
    per_rank, remainder = divmod(total_steps, world_size)
    first = rank * per_rank + min(rank, remainder)
    return StepRange(start=first, end=first + per_rank + (1 if rank < remainder else 0))


def stage_sizes(total_steps: int, world_size: int, *, balanced: bool = False) -> list[int]:
    """Step count of every stage, e.g. ``stage_sizes(25, 8, balanced=True) == [4,3,3,3,3,3,3,3]``."""

    fn = assign_steps_balanced if balanced else assign_steps
    return [fn(total_steps, world_size, r).count for r in range(world_size)]
