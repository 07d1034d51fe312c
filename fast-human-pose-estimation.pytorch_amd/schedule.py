"""Multi-lane schedule of a plan phase: which ops must wait for which ops of OTHER lanes.

A phase (a range of plan ops replayed as one unit) is issued on several HIP streams ("lanes"): ops of one lane run in
list order on that lane's stream; an op additionally waits (event) for the ops of other lanes it conflicts with.
Conflicts are found on PHYSICAL memory intervals (arena, first element, last element+1) after memory planning, so
tensors that share storage -- planner reuse, in-place gradient accumulation, deliberate aliases -- are ordered exactly
as in the sequential list: any execution that respects lane order + the returned waits is equivalent to running the
list front to back.  Ops with unknown access sets (memset, loss, Adam, table-driven launches) are barriers.

Pure Python + numpy (no torch, no GPU): tests/test_graph_cpu.py checks the soundness of the result by brute force.
"""
import numpy as np


class PhaseSchedule:
    """ops: list of (lane, accesses) in issue order; accesses = (reads, writes) lists of graph.Buf, or None (barrier).

    After construction: .lanes[i], .waits[i] (indices, relative to the phase, of ops in other lanes that op i must
    wait for; transitively reduced against lane order) and .clock[i] (vector clock, used by the tests)."""

    def __init__(self, ops, n_lanes):
        self.n = len(ops)
        self.n_lanes = n_lanes
        self.lanes = [l for l, _ in ops]
        self.waits = [[] for _ in ops]
        self.clock = np.full((self.n, n_lanes), -1, dtype=np.int64)
        arena_ids = {}
        cap = 64
        rec = np.zeros((cap, 5), dtype=np.int64)       # arena, start, end, op, is_write
        nrec = 0
        lane_last = [-1] * n_lanes
        last_barrier = -1
        for i, (lane, acc) in enumerate(ops):
            need = [-1] * n_lanes
            if acc is None:
                for l in range(n_lanes):
                    need[l] = lane_last[l]
            else:
                rd, wr = acc
                for bufs, is_write in ((rd, 0), (wr, 1)):
                    for b in bufs:
                        a = arena_ids.setdefault(b.arena, len(arena_ids))
                        s, e = b.off, b.off + b.numel
                        if nrec:
                            r = rec[:nrec]
                            hit = (r[:, 0] == a) & (r[:, 1] < e) & (r[:, 2] > s)
                            if not is_write:
                                hit &= r[:, 4] == 1
                            for j in np.unique(r[hit, 3]):
                                j = int(j)
                                need[self.lanes[j]] = max(need[self.lanes[j]], j)
                        if nrec == cap:
                            cap *= 2
                            rec = np.concatenate([rec, np.zeros_like(rec)])
                        rec[nrec] = (a, s, e, i, is_write)
                        nrec += 1
                if last_barrier >= 0:
                    need[self.lanes[last_barrier]] = max(need[self.lanes[last_barrier]], last_barrier)
            vc = self.clock[lane_last[lane]].copy() if lane_last[lane] >= 0 else np.full(n_lanes, -1, dtype=np.int64)
            # strongest requirements first, so that what they imply transitively prunes the rest
            for l in sorted(range(n_lanes), key=lambda l: -need[l]):
                if l == lane or need[l] < 0 or need[l] <= vc[l]:
                    continue
                self.waits[i].append(need[l])
                vc = np.maximum(vc, self.clock[need[l]])
            vc[lane] = i
            self.clock[i] = vc
            lane_last[lane] = i
            if acc is None:
                last_barrier = i
                # everything before a barrier is ordered before everything after it: old records are dead
                nrec = 0

    def happens_before(self, j, i):
        """True if op j is guaranteed to complete before op i starts under lane order + waits."""
        return j < i and self.clock[i][self.lanes[j]] >= j
