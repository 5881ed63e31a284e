"""configs[4] over N ranks (mgnns_amd/stress.py::plan_shards): channels first, the read-out batch beyond three ranks; every
(channel, sample) pair is computed exactly once.  No GPU."""
import pytest

from mgnns_amd import stress


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 8, 16])
def test_every_channel_sample_pair_exactly_once(world):
    plan = stress.plan_shards(world)
    assert len(plan) == world
    cover = {}
    for rank, shards in enumerate(plan):
        assert shards, "rank %d idle" % rank
        for c, b0, b1 in shards:
            assert 0 <= c < stress.N_CHANNELS and 0 <= b0 < b1 <= stress.BATCH
            for b in range(b0, b1):
                assert (c, b) not in cover
                cover[(c, b)] = rank
    assert len(cover) == stress.N_CHANNELS * stress.BATCH
    if world <= stress.N_CHANNELS:                        # whole channels only
        assert all(b0 == 0 and b1 == stress.BATCH for sh in plan for _, b0, b1 in sh)
    else:                                                 # one channel per rank, batch split inside a channel's group
        assert all(len(sh) == 1 for sh in plan)
        sizes = [b1 - b0 for sh in plan for _, b0, b1 in sh]
        assert max(sizes) - min(sizes) <= stress.BATCH // (world // stress.N_CHANNELS) - stress.BATCH // (world // stress.N_CHANNELS + 1) + 1


def test_bad_world():
    with pytest.raises(ValueError):
        stress.plan_shards(0)
