"""CPU model of the push transport's message protocol (sph_project_amd/csrc/sph_halo.hpp; SURVEY 8e messages): does any slot of a
rank's inbox get overwritten before its last reader is through?

Why a model: round 5 fixed a race empirically (one step-message header per inbox side -> one per message parity; 4 of 4 full-suite runs
failed before, 0 of 8 after) without being able to name the interleaving.  This file enumerates interleavings instead of arguing.

What is modelled, per ordered pair (writer w -> reader r), exactly the regions of sph_halo_impl.hpp's inbox layout:
  rec_pay[2]   step-message payload, region = message number & 1       (k_halo_classify / the presending force pass store it)
  rec_hdr[H]   step-message header {seq, count, status, stride}        (H = 1: the pre-round-5 protocol; H = 2: rec[m & 1])
  fld_pay[2]   field-message payload, region = field number & 1        (k_halo_pack2 / the field-sending density pass)
  fld_seq      number of the last complete field message: ONE word, no header (both sides know the sizes from their own SlabDyn)
and per rank, its own stream-ordered state that the consumers read: SlabDyn[2] (bank = step message & 1), halo_counts[2].

Execution model: every rank runs ONE in-order stream of kernels; a kernel's workgroups start in any order and at any time after the
previous kernel of that stream has completed (all its workgroups), and run at any relative speed -- eight ranks time-sliced on one GPU
next to other GPU contexts give no better guarantee, and neither do eight GPUs.  The waiting kernels (k_halo_unpack2, k_halo_unpack2f)
are modelled with W workgroups each: workgroup 0 first ANNOUNCES this rank's own message (header + number / fld_seq), then every
workgroup on its own polls the number in its own inbox (accepting seq >= expected, halo_poll), reads the header, consumes the payload.

A violation is a workgroup consuming a slot whose content is not the message it is consuming (another message's header or payload,
or a payload still being written).  The search is exhaustive (every interleaving, memoised on the global state) for small
configurations and randomised (seeded) for larger ones.
"""
import random

import pytest


class Violation(Exception):
    pass


def wcsph_program(steps, presend=True, fused_fields=True, fields_per_step=1, prepare_fields=0):
    """Kernel sequence of one rank: sph_prepare (a step message and NO field message: WCSPH's prepare() sorts, nothing else) followed by
    `steps` asynchronous steps.  Step message numbers m = 1 .., field numbers f = 1 .. (both in lockstep on every rank).
    presend: the force pass of step k stores the payload of step k + 1's message (HaloSend) -- else k_halo_classify does, at the head of
    step k + 1.  fused_fields: the density pass stores the field payload (HaloFieldSend) -- same position in the stream as k_halo_pack2,
    so the model has one 'write_fld' either way.  fields_per_step > 1: the ghost refreshes of an iterative solver."""
    prog, m, f = [], 1, 0
    prog += [("write_rec", m), ("unpack2", m)]
    for _ in range(prepare_fields):     # (DFSPH's prepare exchanges the ghost densities)
        f += 1
        prog += [("write_fld", f), ("unpack2f", f)]
    present = False
    for k in range(steps):
        m += 1
        if not present:
            prog.append(("write_rec", m))            # k_halo_classify
        prog.append(("unpack2", m))                  # announce, wait, append, settle SlabDyn[m & 1]
        for _ in range(fields_per_step):
            f += 1
            prog += [("write_fld", f), ("unpack2f", f)]
        present = presend and k + 1 < steps
        if present:
            prog.append(("write_rec", m + 1))        # the force pass classifies and streams the next message
    return prog


class Model:
    def __init__(self, nranks, program, workgroups=2, headers=2, strict_accept=False):
        self.R, self.prog, self.W, self.H, self.strict = nranks, program, workgroups, headers, strict_accept
        self.index, self.init_slots, self._acts = {}, [], {}
        for w in range(self.R):
            for r in (w - 1, w + 1):
                if 0 <= r < self.R:
                    for i in range(2):
                        self._slot((w, r, "rec_pay", i), 0)
                        self._slot((w, r, "fld_pay", i), 0)
                    for i in range(self.H):
                        self._slot((w, r, "rec_hdr", i), (0, 0))      # (seq, message the count belongs to)
                    self._slot((w, r, "fld_seq", 0), 0)

    def _slot(self, key, value):
        self.index[key] = len(self.init_slots)
        self.init_slots.append(value)

    # state = (ranks, slots); ranks[r] = (kernel index, tuple of per-workgroup pcs); slots: tuple, position = self.index[(w, r, kind, idx)]
    def initial(self):
        ranks = tuple((0, (0,) * self._nthreads(0)) for _ in range(self.R))
        return ranks, tuple(self.init_slots)

    def _nthreads(self, kidx):
        if kidx >= len(self.prog):
            return 0
        return self.W if self.prog[kidx][0].startswith("unpack") else 1

    def neighbours(self, r):
        return [q for q in (r - 1, r + 1) if 0 <= q < self.R]

    def thread_steps(self, r, kidx, wg):
        """The atomic actions of workgroup `wg` of kernel `kidx` on rank r, in program order."""
        got = self._acts.get((r, kidx, wg))
        if got is None:
            got = self._acts[(r, kidx, wg)] = [(op, self.index[key], num, key) for op, key, num in self._thread_steps(r, kidx, wg)]
        return got

    def _thread_steps(self, r, kidx, wg):
        kind, num = self.prog[kidx]
        nb = self.neighbours(r)
        acts = []
        if kind == "write_rec":
            for q in nb:
                acts += [("begin", (r, q, "rec_pay", num & 1), num), ("end", (r, q, "rec_pay", num & 1), num)]
        elif kind == "write_fld":
            for q in nb:
                acts += [("begin", (r, q, "fld_pay", num & 1), num), ("end", (r, q, "fld_pay", num & 1), num)]
        elif kind == "unpack2":
            hi = num & (self.H - 1)
            if wg == 0:
                for q in nb:
                    acts.append(("hdr", (r, q, "rec_hdr", hi), num))
            for q in nb:      # (the kernel's lanes 0 and 1 poll both sides at once; per side the order is poll -> header -> payload)
                acts += [("poll_hdr", (q, r, "rec_hdr", hi), num), ("read_hdr", (q, r, "rec_hdr", hi), num),
                         ("read", (q, r, "rec_pay", num & 1), num)]
        elif kind == "unpack2f":
            if wg == 0:
                for q in nb:
                    acts.append(("fseq", (r, q, "fld_seq", 0), num))
            for q in nb:
                acts += [("poll_f", (q, r, "fld_seq", 0), num), ("read", (q, r, "fld_pay", num & 1), num)]
        return acts

    def enabled(self, state):
        ranks, slots = state
        out = []
        for r, (kidx, pcs) in enumerate(ranks):
            if kidx >= len(self.prog):
                continue
            for wg, pc in enumerate(pcs):
                acts = self.thread_steps(r, kidx, wg)
                if pc >= len(acts):
                    continue
                a = acts[pc]
                if a[0] == "poll_hdr":
                    seq = slots[a[1]][0]
                    if not (seq == a[2] if self.strict else seq >= a[2]):
                        continue
                elif a[0] == "poll_f":
                    seq = slots[a[1]]
                    if not (seq == a[2] if self.strict else seq >= a[2]):
                        continue
                out.append((r, wg))
        return out

    def step(self, state, who):
        ranks, slots_t = state
        slots = list(slots_t)
        r, wg = who
        kidx, pcs = ranks[r]
        op, key, num, name = self.thread_steps(r, kidx, wg)[pcs[wg]]
        if op == "begin":
            slots[key] = -num                       # being written
        elif op == "end":
            slots[key] = num
        elif op == "hdr":
            slots[key] = (num, num)                 # count / status / stride, fence, then the number: one release
        elif op == "fseq":
            slots[key] = num
        elif op == "read_hdr":
            seq, owner = slots[key]
            if owner != num:
                raise Violation("rank %d workgroup %d consuming step message %d took the header of message %d (slot %s)" % (r, wg, num, owner, name[2:]))
        elif op == "read":
            if slots[key] != num:
                raise Violation("rank %d workgroup %d consuming message %d found %s in %s" % (
                    r, wg, num, ("message %d being written" % -slots[key]) if slots[key] < 0 else "message %d" % slots[key], name[2:]))
        pcs = list(pcs)
        pcs[wg] += 1
        # kernel complete -> the stream's next kernel may start
        if all(pcs[w] >= len(self.thread_steps(r, kidx, w)) for w in range(len(pcs))):
            kidx += 1
            pcs = [0] * self._nthreads(kidx)
        ranks = ranks[:r] + ((kidx, tuple(pcs)),) + ranks[r + 1:]
        return ranks, tuple(slots)

    def finished(self, state):
        return all(k >= len(self.prog) for k, _ in state[0])

    def exhaustive(self, limit=3_000_000):
        """Every interleaving (depth-first, memoised).  Returns the number of distinct states; raises Violation with the trace."""
        init = self.initial()
        seen = {init}
        stack = [init]
        parent = {init: None}
        deadlocks = 0
        while stack:
            st = stack.pop()
            en = self.enabled(st)
            if not en:
                if not self.finished(st):
                    deadlocks += 1
                continue
            for who in en:
                try:
                    nx = self.step(st, who)
                except Violation as v:
                    trace, cur = [who], st
                    while parent[cur] is not None:
                        cur, w = parent[cur]
                        trace.append(w)
                    raise Violation("%s after %d actions; schedule (rank, workgroup): %s" % (v, len(trace), trace[::-1][-40:])) from None
                if nx not in seen:
                    seen.add(nx)
                    parent[nx] = (st, who)
                    stack.append(nx)
                    if len(seen) > limit:
                        raise RuntimeError("state space larger than %d" % limit)
        return len(seen), deadlocks

    def random_walks(self, walks, seed, laggard_bias=0.0):
        """Seeded random schedules.  laggard_bias: probability of NOT scheduling a workgroup other than 0 of a waiting kernel when something
        else can run (late workgroups: the CU slots are taken, the process is time-sliced off the GPU)."""
        rng = random.Random(seed)
        for _ in range(walks):
            st = self.initial()
            while True:
                en = self.enabled(st)
                if not en:
                    assert self.finished(st), "deadlock"
                    break
                if laggard_bias > 0.0:
                    early = [w for w in en if w[1] == 0]
                    if early and rng.random() < laggard_bias:
                        en = early
                st = self.step(st, rng.choice(en))


# ------------------------------------------------------------------------------------------------------------ tests
def test_double_header_protocol_is_safe_exhaustively():
    """H = 2 (the protocol in the tree): no interleaving of 2 ranks x 3 or 4 workgroups over prepare + 2 steps, or of 3 ranks x 2
    workgroups over prepare + 3 steps (presend + fused field send on) consumes a wrong or half-written slot, and none deadlocks.
    SPH_MODEL_FULL=1 adds 3 ranks x 3 workgroups and 4 ranks x 2 workgroups (3.8 M / 3.6 M states, ~4 minutes each; both clean when
    this was written)."""
    import os
    sizes = ((2, 3, 2), (3, 2, 3), (2, 4, 2)) + (((3, 3, 1), (4, 2, 1)) if os.environ.get("SPH_MODEL_FULL") else ())
    for nranks, wgs, steps in sizes:
        m = Model(nranks, wcsph_program(steps), workgroups=wgs, headers=2)
        states, deadlocks = m.exhaustive(limit=6_000_000)
        print("double header, %d ranks x %d workgroups, prepare + %d steps: %d states, no violation" % (nranks, wgs, steps, states))
        assert deadlocks == 0 and states > 1000


def test_single_header_race_is_found_and_named():
    """H = 1 (pre-round-5, SPH_TEST_SINGLE_HEADER in the test-hook build): the model FINDS the race, and the trace names it.  Between
    sph_prepare's step message and the first step's there is NO field message (WCSPH's prepare only sorts), so nothing holds a writer
    back once it has the reader's announce -- and a consumer announces its own message BEFORE it polls (so that two ranks waiting for each
    other cannot deadlock).  Reader R: announce(1) ... stalls (its queue is time-sliced off the GPU: 8 processes + the suite's other
    contexts on one device; or simply a workgroup that gets its CU slot late).  Writer W: sees R's announce, finishes k_halo_unpack2(1),
    prepare returns, the first step classifies and k_halo_unpack2(2) announces message 2 INTO THE SAME HEADER.  R resumes, polls: seq
    2 >= 1, accepted; it takes message 2's record count for message 1's payload -- more records than message 1 holds: uninitialised inbox
    memory appended (non-finite positions, id 0); fewer: arrivals lost.  Both symptoms of profiles/r05_halo_header_race_ab.txt."""
    for wgs in (1, 2):      # no late workgroup needed: the stall between a workgroup's own announce and its poll is enough
        m = Model(2, wcsph_program(1), workgroups=wgs, headers=1)
        with pytest.raises(Violation, match="consuming step message 1 took the header of message 2"):
            m.exhaustive()
    # it is the prepare -> first step seam only: once every step message is followed by a field message (the steady state inside an
    # advance()), even the single header survives every interleaving -- the writer's message m + 1 waits for the reader's field
    # message, which is enqueued behind the reader's whole k_halo_unpack2(m).  (Which is why 30 of 30 stand-alone runs passed: the seam
    # is crossed once per test, and the stall has to last as long as the neighbour's host needs to get from prepare() into the first step.)
    seam_free = wcsph_program(3)[2:]          # drop prepare's exchange: every step message is followed by a field message
    seam_free = [(k, n - 1 if k in ("write_rec", "unpack2") else n) for k, n in seam_free]
    states, deadlocks = Model(2, seam_free, workgroups=3, headers=1).exhaustive()
    assert deadlocks == 0
    states, deadlocks = Model(3, seam_free[:9], workgroups=2, headers=1).exhaustive()
    assert deadlocks == 0
    # strict acceptance (seq == expected) would not have saved the single header: a consumer that stalls between its poll and its header
    # read still takes the next message's count (and one that stalls before the poll hangs instead)
    with pytest.raises(Violation, match="took the header of message 2"):
        Model(2, wcsph_program(1), workgroups=2, headers=1, strict_accept=True).exhaustive()
    # two step messages back to back are the general form of the seam (a re-sort after particles were appended between steps,
    # dfsph_step_begin's sort_dirty path): safe with two headers
    back_to_back = [("write_rec", 1), ("unpack2", 1), ("write_rec", 2), ("unpack2", 2), ("write_rec", 3), ("unpack2", 3), ("write_rec", 4), ("unpack2", 4)]
    states, deadlocks = Model(3, back_to_back, workgroups=2, headers=2).exhaustive()
    assert deadlocks == 0


def test_headerless_field_messages_are_safe():
    """The field messages carry no header -- one fld_seq word per inbox side, sizes from the reader's own SlabDyn -- and `seq >= expected`
    lets a late workgroup accept a LATER number.  Safe because a later number implies nothing about the payload region being read: field
    message f + 2 (same region) is only written after the writer has passed k_halo_unpack2f(f + 1), i.e. after the reader announced f + 1,
    which is enqueued behind the reader's whole k_halo_unpack2f(f).  Checked with several field messages per step (iterative solvers'
    ghost refreshes), with and without presend, and with field messages in prepare (DFSPH)."""
    for kw in (dict(fields_per_step=3), dict(fields_per_step=2, presend=False), dict(fields_per_step=1, prepare_fields=1)):
        states, deadlocks = Model(2, wcsph_program(2, **kw), workgroups=3, headers=2).exhaustive()
        assert deadlocks == 0
        states, deadlocks = Model(3, wcsph_program(1, **kw), workgroups=2, headers=2).exhaustive()
        assert deadlocks == 0


def test_random_schedules_of_larger_configurations():
    """4 ranks x 4 workgroups x (prepare + 5 steps), 600 seeded random schedules, half of them biased towards late workgroups: the
    double-header protocol holds; the single header fails within the same budget."""
    prog = wcsph_program(5)
    Model(4, prog, workgroups=4, headers=2).random_walks(300, seed=1)
    Model(4, prog, workgroups=4, headers=2).random_walks(300, seed=2, laggard_bias=0.9)
    with pytest.raises(Violation):
        Model(4, prog, workgroups=4, headers=1).random_walks(600, seed=3, laggard_bias=0.9)
