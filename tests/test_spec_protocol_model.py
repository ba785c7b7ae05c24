"""A CPU model of the two-stream speculative tail's PROTOCOL (include/bjx_nuts.h "Speculative tail";
csrc/bjx_nuts_spec.hip k_nuts_spec_integrate / k_nuts_spec_book): one row, the integrator ("A") and the bookkeeper ("B") as
the state machines the kernels implement, driven by an adversarial random scheduler.  It checks what the GPU parity tests
cannot enumerate -- every interleaving the memory protocol allows:

* B consumes exactly the leaves (epoch t, leaf 0 .. L_t - 1) of every transition, in order, and nothing else;
* a ring slot is never overwritten before B has consumed (or skipped) it, whatever the lag, for every `lead` < ring;
* the leaves A speculates past the end of a transition are skipped by the acknowledged count, never read as data;
* the run terminates, and the waste per transition end is bounded by lead + the visibility latencies.

The arithmetic is not modelled (a leaf is its identity); visibility is: a record is ANNOUNCED by A's next tick only, B's
epoch is seen by A at its next tick only, and A's acknowledgement / B's consumed count are seen with arbitrary delay."""
import random

import pytest


class Model:
    def __init__(self, lengths, l_max, ring, lead, rng):
        self.lengths, self.l_max, self.ring, self.lead, self.rng = lengths, l_max, ring, lead, rng
        # shared memory
        self.slots = [None] * ring          # ring: (epoch, leaf) identity of the pushed record
        self.avail = 0                      # records announced by A
        self.ack = (0, 0)                   # (epoch, first record number of that epoch)
        self.b_ep, self.b_cnt = 0, 0        # B -> A: epoch (-1 = chain finished), consumed count
        # A's private words
        self.a_ep, self.a_leaf, self.a_cnt, self.a_state = 0, 0, 0, "run"
        self.a_seen_cnt_b = 0               # a possibly stale view of b_cnt (only ever too small: safe)
        # B's private words
        self.my_ep, self.cnt, self.need_ack, self.b_leaf = 0, 0, False, 0
        self.n_avail_cached = 0
        self.consumed, self.stale, self.pushed, self.finished = [], 0, 0, False
        self.unconsumed = set()             # record numbers pushed and not yet consumed / skipped

    # ---- stream A: one integrate launch (after the callable)
    def tick_a(self):
        if self.a_state == "done":
            return
        self.avail = self.a_cnt             # everything pushed by EARLIER launches is complete: announce it
        if self.rng.random() < 0.7:         # a relaxed load may or may not see B's latest consumed count
            self.a_seen_cnt_b = self.b_cnt
        if self.b_ep == -1:
            self.a_state = "done"
            return
        if self.b_ep != self.a_ep:          # restart: acknowledge with the first record number of the new epoch
            self.a_ep, self.a_leaf, self.a_state = self.b_ep, 0, "run"
            self.ack = (self.a_ep, self.a_cnt)
            return
        if self.a_state != "run":
            return
        if self.a_cnt - self.a_seen_cnt_b >= self.lead:
            return                          # far enough ahead: wait (pushes nothing)
        slot = self.a_cnt % self.ring
        assert (self.a_cnt - self.ring) not in self.unconsumed, "ring slot overwritten before it was consumed"
        self.slots[slot] = (self.a_ep, self.a_leaf)
        self.unconsumed.add(self.a_cnt)
        self.a_cnt += 1
        self.pushed += 1
        self.a_leaf += 1
        if self.a_leaf >= self.l_max:
            self.a_state = "wait"           # tree exhausted: nothing left to speculate in this transition

    # ---- stream B: one iteration of the bookkeeper's loop
    def step_b(self):
        if self.finished:
            return
        if self.need_ack:
            ep, first = self.ack
            if ep != self.my_ep:
                return                      # not acknowledged yet
            assert first >= self.cnt
            for n in range(self.cnt, first):
                self.unconsumed.discard(n)
            self.stale += first - self.cnt
            self.cnt = first
            self.b_cnt = self.cnt
            self.need_ack = False
        if self.n_avail_cached - self.cnt <= 0:
            self.n_avail_cached = self.avail
            if self.n_avail_cached - self.cnt <= 0:
                return                      # nothing announced
        tag = self.slots[self.cnt % self.ring]
        assert tag == (self.my_ep, self.b_leaf), (tag, self.my_ep, self.b_leaf)   # the next leaf of THIS transition
        self.consumed.append(tag)
        self.unconsumed.discard(self.cnt)
        self.cnt += 1
        self.b_cnt = self.cnt
        self.b_leaf += 1
        if self.b_leaf >= self.lengths[self.my_ep]:            # the transition ends at this leaf
            if self.my_ep + 1 >= len(self.lengths):
                self.finished = True
                self.b_ep = -1
            else:
                self.my_ep += 1
                self.b_leaf = 0
                self.b_ep = self.my_ep     # (published behind a fence: key, step size, first pending position)
                self.need_ack = True


@pytest.mark.parametrize("ring,lead", [(8, 1), (8, 4), (8, 7), (64, 4), (64, 63)])
@pytest.mark.parametrize("p_a", [0.2, 0.5, 0.8])
def test_protocol_under_random_schedules(ring, lead, p_a):
    for seed in range(40):
        rng = random.Random(1000 * seed + ring + lead)
        l_max = rng.choice([1, 3, 7, 31])
        lengths = [rng.randint(1, l_max) for _ in range(rng.randint(1, 12))]
        m = Model(lengths, l_max, ring, lead, rng)
        for _ in range(200000):
            if m.finished and m.a_state == "done":
                break
            if rng.random() < p_a:
                m.tick_a()
            else:
                m.step_b()
        assert m.finished and m.a_state == "done", "the protocol did not terminate"
        assert m.consumed == [(t, i) for t, n in enumerate(lengths) for i in range(n)]
        assert m.pushed == len(m.consumed) + m.stale + len(m.unconsumed)
        # waste: per transition end at most `lead` records in flight past the end (plus what the exhausted-tree stop allows)
        assert m.stale <= (len(lengths) - 1) * lead


def test_lead_of_a_full_ring_is_refused_by_construction():
    """lead == ring would let A overwrite the slot B is reading: the kernels clamp lead to ring - 1."""
    rng = random.Random(7)
    m = Model([5] * 6, 7, 4, 4, rng)
    with pytest.raises(AssertionError):
        for _ in range(20000):
            m.tick_a() if rng.random() < 0.9 else m.step_b()
        raise AssertionError("no overwrite was provoked")  # (an overwrite shows up as a tag or slot assertion above)
