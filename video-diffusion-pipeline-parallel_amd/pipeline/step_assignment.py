"""Which diffusion steps does each pipeline rank own?

Torch-free on purpose (like the reference module) so the arithmetic is testable without a
process group.

Mirrors ``/root/reference/src/pipeline/step_assignment.py``:
  * ``StepRange``      – ref ``:12-32`` (frozen ``[start, end)``; ``count``; iterable; ValueError
    on negative bounds or ``end < start``)
  * ``assign_steps``   – ref ``:35-69`` (uniform contiguous split; ValueError on
    ``total_steps <= 0``, ``world_size <= 0``, rank outside ``[0, world_size)``, or
    ``total_steps % world_size != 0``)

Extensions (NOT in the reference, separately named so the strict API above is unchanged):
  * ``assign_steps_balanced`` – contiguous split that tolerates a remainder; the first
    ``total_steps % world_size`` ranks take one extra step.  Needed for the 25-step
    schedule on 2/4/8 GPUs that BASELINE.json asks for (25 -> [4,3,3,3,3,3,3,3]).
  * ``assign_steps_rotating`` – the same split, but the stages that take the extra step rotate with the
    sample index, which removes the permanent bottleneck stage.
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


def assign_steps_rotating(total_steps: int, world_size: int, rank: int, sample_idx: int) -> StepRange:
    """Balanced contiguous split whose "+1 step" stages rotate with the sample index (extension).

    With ``R = total_steps % world_size`` the stages ``(sample_idx + k) % world_size`` for ``k < R`` own
    ``total_steps // world_size + 1`` steps of THAT sample, the others one fewer.  Every sample still runs steps
    ``0 .. total_steps-1`` in order, but over ``world_size`` consecutive samples each stage does the same amount of
    work, so the steady-state ceiling of e.g. 25 steps on 8 stages is 8x instead of 25/4 = 6.25x.
    """

    _validate(total_steps, world_size, rank)
    if total_steps < world_size:
        raise ValueError("total_steps must be >= world_size so every rank owns a step.")
    if sample_idx < 0:
        raise ValueError("sample_idx must be non-negative.")
    per_rank, remainder = divmod(total_steps, world_size)

    def extra(r: int) -> int:
        return 1 if (r - sample_idx) % world_size < remainder else 0

    first = rank * per_rank + sum(extra(r) for r in range(rank))
    return StepRange(start=first, end=first + per_rank + extra(rank))


def stage_sizes(total_steps: int, world_size: int, *, balanced: bool = False) -> list[int]:
    """Step count of every stage, e.g. ``stage_sizes(25, 8, balanced=True) == [4,3,3,3,3,3,3,3]``."""

    fn = assign_steps_balanced if balanced else assign_steps
    return [fn(total_steps, world_size, r).count for r in range(world_size)]


# ---- ring schedule (PipelineConfig.ring): pure index arithmetic, shared by the executor and its tests -------------
def ring_sample(rank: int, batch: int, slot: int, world_size: int) -> int:
    """Sample that ``rank`` works on in ``slot`` of ``batch`` (it started on rank ``(rank - slot) mod N``)."""

    return batch * world_size + ((rank - slot) % world_size)


def ring_rank(sample_idx: int, stage: int, world_size: int) -> int:
    """Rank that runs ``stage`` of ``sample_idx``: home rank ``i mod N`` for stage 0, then one rank further per stage."""

    return (sample_idx + stage) % world_size


def ring_finish_rank(sample_idx: int, world_size: int) -> int:
    return ring_rank(sample_idx, world_size - 1, world_size)
