"""The persistent kernels' work queue (kernels_tiled.hpp, WaveQueue) as a host model under random interleavings.

Device code cannot run here, so the PROTOCOL is restated and checked: 8 segments of entries, the first entries of a segment
fixed per wave (no atomics at kernel start), the rest handed out by one ticket counter per segment whose high bits carry the
set of segments known to be dry; the wave whose ticket equals a segment's size announces it to the other counters; a wave pops
only segments outside the set it has seen.  Properties, whatever the interleaving of the waves' atomic steps: every entry is
served exactly once, every wave terminates, and a wave never issues more than one failed pop per segment (the
queue's liveness must not rest on one schedule)."""
import random

import pytest

K_QUEUES, COUNT_BITS = 8, 24


class Device:
    def __init__(self, seg, grid):
        self.seg, self.grid = seg, grid                    # seg: K_QUEUES + 1 boundaries
        self.q = [0] * K_QUEUES                            # counter = tickets | dry set << COUNT_BITS
        self.atomics = 0

    def n_static(self, c):                                 # waves of XCD c = fixed first entries of segment c
        return ((self.grid + K_QUEUES - 1 - c) // K_QUEUES) * 4

    def avail(self, c):
        return self.seg[c + 1] - self.seg[c] - self.n_static(c)


def wave_program(dev, block, wave, served, failed):
    """generator: yields once per atomic step (the scheduler interleaves the waves between steps)"""
    n = dev.seg[K_QUEUES]
    own = block % K_QUEUES
    dry = 0
    e = dev.seg[own] + (block // K_QUEUES) * 4 + wave      # first(): the fixed entry, if the segment is that long
    if e < dev.seg[own + 1]:
        served.append(e)
    while True:
        # pop(): first segment outside the dry set, starting at the XCD's own
        opn = ~dry & ((1 << K_QUEUES) - 1)
        rot = ((opn >> own) | (opn << (K_QUEUES - own))) & ((1 << K_QUEUES) - 1)
        if not rot:
            return
        cur = (own + (rot & -rot).bit_length() - 1) % K_QUEUES
        yield                                              # -- the atomic add lands here
        old = dev.q[cur]
        dev.q[cur] = old + 1
        dev.atomics += 1
        ticket, seen = old & ((1 << COUNT_BITS) - 1), old >> COUNT_BITS
        dry |= seen & ((1 << K_QUEUES) - 1)
        av = dev.avail(cur)
        if ticket < av:
            served.append(dev.seg[cur] + dev.n_static(cur) + ticket)
            continue
        failed[cur] = failed.get(cur, 0) + 1
        dry |= 1 << cur
        if ticket == max(av, 0):                           # the first pop to fail here: tell the other counters
            yield                                          # -- one vector atomic OR
            for c in range(K_QUEUES):
                if c != cur:
                    dev.q[c] |= 1 << (COUNT_BITS + cur)
        assert n >= 0


@pytest.mark.parametrize("n_entries,grid", [(0, 1), (1, 1), (5, 3), (31, 8), (64, 16), (1000, 24), (9200, 64), (333, 100)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_every_entry_served_once_under_any_interleaving(n_entries, grid, seed):
    rng = random.Random(1000 * seed + n_entries + grid)
    cuts = sorted(rng.randint(0, n_entries) for _ in range(K_QUEUES - 1))      # unequal segments (equal COST, not count)
    seg = [0] + cuts + [n_entries]
    dev = Device(seg, grid)
    served, waves, fails = [], [], []
    for b in range(grid):
        for w in range(4):
            f = {}
            fails.append(f)
            waves.append(wave_program(dev, b, w, served, f))
    live = list(range(len(waves)))
    steps = 0
    while live:
        i = rng.choice(live) if seed else live[0]          # seed 0: one wave runs to completion before the next starts
        try:
            next(waves[i])
        except StopIteration:
            live.remove(i)
        steps += 1
        assert steps < 50 * (n_entries + 64 * grid) + 1000, "the queue does not terminate"
    assert sorted(served) == list(range(n_entries)), "an entry was skipped or served twice"
    assert all(v == 1 for f in fails for v in f.values()), "a wave popped a segment again after it had failed there"
    assert all(len(f) <= K_QUEUES for f in fails)
