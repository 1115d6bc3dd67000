"""Host side of rank-consistent speculative culling (litegs_amd/dp.py: LockstepSpeculation) against a scripted device: whatever steps fail --
culled forwards, record-block overflows, failures inside a replay, several in a row -- every step's update is applied exactly once and in
order, a forced replay is never asked for twice, and nothing is left pending after flush().  (The two-rank agreement itself is
tests/test_dp_gloo.py::test_speculative_culling_stays_rank_consistent_world2; the real kernels: tests/test_gpu_dp.py.)"""
import random

import pytest
from hypothesis import given, settings, strategies as st

from litegs_amd import dp


class ScriptedDevice:
    """what csrc/dp.hip does with the poison word, with failures drawn from a seeded script: an execution of step `no` that is not forced
    fails with probability p_fail (its culled forward) or p_over (its record block); a forced one (unculled, exact capacity) cannot"""

    def __init__(self, seed, p_fail, p_over):
        self.rng = random.Random(seed)
        self.p_fail, self.p_over = p_fail, p_over
        self.poison = 0
        self.status_words = {}
        self.applied = []
        self.executions = []

    def execute(self, no, force):
        self.executions.append((no, force))
        flags = 0
        if not force:
            if self.rng.random() < self.p_fail:
                self.poison = 1                      # the culled forward's bound check raises the sticky word ...
            if self.rng.random() < self.p_over:
                flags |= dp.Speculation.OVERFLOW
        if self.poison:
            flags |= 1                               # ... which travels in the record header from then on
        if flags:
            self.poison = 1
        self.status_words[no] = flags
        if not self.poison:
            self.applied.append(no)

    def status(self, no):
        return self.status_words[no]


class Done:
    def synchronize(self):
        pass


@settings(max_examples=200, deadline=None)
@given(seed=st.integers(0, 10_000), depth=st.integers(1, 4), steps=st.integers(1, 40),
       p_fail=st.sampled_from([0.0, 0.05, 0.3, 0.7]), p_over=st.sampled_from([0.0, 0.1, 0.5]))
def test_every_step_is_applied_exactly_once_and_in_order(seed, depth, steps, p_fail, p_over):
    dev = ScriptedDevice(seed, p_fail, p_over)
    forced = []

    def on_failed(rec, flags):
        assert flags != 0
        forced.append(rec[0])
        dev.poison = 0                               # clear_poison + after_failed_step

    ls = dp.LockstepSpeculation(dev, depth, lambda rec, force: dev.execute(rec[0], force), on_failed, Done, lambda: None)
    for no in range(1, steps + 1):
        rec = (no,)
        ls.before_step(rec)
        dev.execute(no, False)
        ls.after_step(rec)
        assert len(ls.events) <= depth               # the host never runs further ahead than `depth` unverified steps
    ls.flush()
    assert dev.applied == list(range(1, steps + 1))
    assert dev.poison == 0 and not ls.events and not ls.ring
    assert len(forced) == len(set(forced))           # a forced execution cannot fail: no step is forced twice ...
    for no in forced:                                # ... and a forced execution is what applied it
        assert (no, True) in dev.executions
    if p_fail == 0.0 and p_over == 0.0:
        assert ls.replays == 0 and dev.executions == [(no, False) for no in range(1, steps + 1)]


def test_a_forced_replay_that_fails_is_an_error():
    class Broken(ScriptedDevice):
        def execute(self, no, force):
            self.status_words[no] = 1                # fails whatever the host does

    dev = Broken(0, 0, 0)
    ls = dp.LockstepSpeculation(dev, 1, lambda rec, force: dev.execute(rec[0], force), lambda rec, flags: None, Done, lambda: None)
    ls.before_step((1,))
    dev.execute(1, False)
    ls.after_step((1,))
    with pytest.raises(RuntimeError, match="failed again"):
        ls.flush()


def test_status_words_refuse_a_step_that_never_reported():
    import numpy as np
    sp = dp.Speculation(None, None, words=np.zeros((2 * dp.Speculation.RING,), dtype=np.int32))
    sp.words[sp.status_addr(5)] = 5
    sp.words[sp.status_addr(5) + 1] = 0b10
    assert sp.status(5) == 0b10
    with pytest.raises(RuntimeError, match="ring overrun"):
        sp.status(5 + dp.Speculation.RING)           # same slot of the ring, another step
