"""Step-assignment known-answer tests.  Same KAT set the reference pins
(/root/reference/tests/test_step_assignment.py: 28 steps over 1/2/4/7 ranks, 29/7 rejected, StepRange
count/iteration/validation) expressed as tables, plus the balanced-split extension."""

import pytest

from vdpp_amd.pipeline.step_assignment import (StepRange, assign_steps, assign_steps_balanced,
                                               assign_steps_rotating, stage_sizes)

KAT_28 = {
    1: [(0, 28)],
    2: [(0, 14), (14, 28)],
    4: [(0, 7), (7, 14), (14, 21), (21, 28)],
    7: [(4 * r, 4 * r + 4) for r in range(7)],
}


@pytest.mark.parametrize("world", sorted(KAT_28))
def test_uniform_split_known_answers(world):
    got = [assign_steps(28, world, r) for r in range(world)]
    assert [(s.start, s.end) for s in got] == KAT_28[world]
    assert all(s.count == 28 // world for s in got)


@pytest.mark.parametrize("total,world", [(28, 7), (8, 2), (30, 5), (105, 7)])
def test_partition_is_exact(total, world):
    seen = []
    for r in range(world):
        seen.extend(assign_steps(total, world, r))
    assert seen == list(range(total))


@pytest.mark.parametrize("kwargs", [
    dict(total_steps=0, world_size=1, rank=0), dict(total_steps=-1, world_size=1, rank=0),
    dict(total_steps=28, world_size=0, rank=0), dict(total_steps=28, world_size=4, rank=4),
    dict(total_steps=28, world_size=4, rank=-1), dict(total_steps=29, world_size=7, rank=0),
    dict(total_steps=25, world_size=8, rank=0),
])
def test_invalid_arguments_raise(kwargs):
    with pytest.raises(ValueError):
        assign_steps(**kwargs)


def test_step_range_behaviour():
    sr = StepRange(start=5, end=10)
    assert sr.count == 5 and list(sr) == [5, 6, 7, 8, 9]
    assert StepRange(0, 7).count == 7 and StepRange(3, 3).count == 0
    for bad in ((-1, 5), (10, 5), (0, -2)):
        with pytest.raises(ValueError):
            StepRange(*bad)
    with pytest.raises(Exception):
        sr.start = 1  # frozen


def test_balanced_split_extension():
    assert stage_sizes(25, 8, balanced=True) == [4, 3, 3, 3, 3, 3, 3, 3]
    assert stage_sizes(25, 4, balanced=True) == [7, 6, 6, 6]
    assert stage_sizes(25, 2, balanced=True) == [13, 12]
    assert stage_sizes(30, 8, balanced=True) == [4, 4, 4, 4, 4, 4, 3, 3]
    assert stage_sizes(28, 7, balanced=True) == stage_sizes(28, 7) == [4] * 7
    for total, world in [(25, 8), (30, 8), (7, 7), (9, 2)]:
        seen = []
        for r in range(world):
            seen.extend(assign_steps_balanced(total, world, r))
        assert seen == list(range(total))
    with pytest.raises(ValueError):
        assign_steps_balanced(3, 4, 0)
    with pytest.raises(ValueError):
        assign_steps_balanced(25, 8, 8)


def test_rotating_split_extension():
    for total, world in [(25, 8), (25, 4), (25, 2), (30, 8), (28, 7), (9, 4)]:
        work = [0] * world
        for sample in range(world):
            seen = []
            for r in range(world):
                rng = assign_steps_rotating(total, world, r, sample)
                seen.extend(rng)
                work[r] += rng.count
                assert rng.count in (total // world, total // world + 1)
            assert seen == list(range(total))                      # every sample runs all steps, in order
        assert len(set(work)) == 1 and work[0] == total            # equal work per stage over `world` samples
    # sample 0 equals the plain balanced split; the heavy stage moves with the sample index
    assert [assign_steps_rotating(25, 8, r, 0).count for r in range(8)] == stage_sizes(25, 8, balanced=True)
    assert [assign_steps_rotating(25, 8, r, 3).count for r in range(8)] == [3, 3, 3, 4, 3, 3, 3, 3]
    assert [assign_steps_rotating(30, 8, r, 7).count for r in range(8)] == [4, 4, 4, 4, 4, 3, 3, 4]
    with pytest.raises(ValueError):
        assign_steps_rotating(25, 8, 0, -1)
    with pytest.raises(ValueError):
        assign_steps_rotating(3, 4, 0, 0)


# ---- ring schedule index arithmetic (PipelineConfig.ring) ----------------------------------------------------------
def test_ring_schedule_properties():
    """For every world size / sample count: each sample visits ranks home, home+1, ... once per stage, in stage order;
    in every slot the ranks work on DIFFERENT samples (all at the same stage index); the sender/receiver formulas of a
    slot boundary agree; and the finishing rank is the one the collection step expects."""
    from vdpp_amd.pipeline.step_assignment import ring_finish_rank, ring_rank, ring_sample
    for n in (1, 2, 3, 4, 8):
        for num in (1, n - 1 or 1, n, n + 1, 2 * n, 3 * n + 2):
            nbatch = -(-num // n)
            seen = {}                                     # sample -> list of (slot, rank)
            for b in range(nbatch):
                for s in range(n):
                    here = [ring_sample(r, b, s, n) for r in range(n)]
                    assert sorted(here) == list(range(b * n, b * n + n))          # a permutation of the batch
                    for r, i in enumerate(here):
                        if i < num:
                            seen.setdefault(i, []).append((s, r))
                            assert ring_rank(i, s, n) == r
                        # what rank r sends after slot s is what rank r+1 expects in slot s+1
                        if s < n - 1:
                            assert ring_sample((r + 1) % n, b, s + 1, n) == i
            assert sorted(seen) == list(range(num))
            for i, visits in seen.items():
                assert [s for s, _ in visits] == list(range(n))                    # stages 0..N-1 in order
                assert [r for _, r in visits] == [(i + s) % n for s in range(n)]   # one rank further per stage
                assert visits[-1][1] == ring_finish_rank(i, n)
