#!/usr/bin/env python
"""How much matrix-pipe time could a better schedule of the two co-resident head workgroups recover?  (NOTES.md 4.2)

k_head_phase<false> keeps two 4-wave workgroups on a CU.  Each alternates MFMA segments (80 K matrix-pipe ticks per 128-sample round)
with scalar pieces (gathers, skinny layers, write-backs, march, composite: ~47 K ticks alone); the pipe idles whenever BOTH are in a
scalar piece.  Round 1 measured 192 K ticks per round pair = 83 % pipe occupancy with the two workgroups free-running.  Two redesigns were
on the table for round 2 (task 4b): one 8-wave workgroup whose two halves are phase-locked, either in lock step (every
barrier is the one hardware s_barrier, so each phase pairs one piece of each half) or free-running but started at a chosen offset.

This model answers what either could gain before a 1 200-line kernel is rewritten for it.  Pieces and durations are the measured
alone-on-a-CU values of profiles/round1/r1o_head_timeline.txt; two halves in MFMA pieces share the pipe (each advances at half rate).

    python tools/overlap_model.py

Result (ticks per round pair; lower is better; 164 K = pipe never idle):
  lock step, natural barriers          best offset 207 K  (WORSE than today: a 1.5 K write-back paired with a 16 K MFMA piece idles its half)
  free running, random offset          ~184 K (model) vs 192 K measured (co-running also slows the scalar pieces themselves)
  free running, best fixed offset      ~175 K: the upper bound of phase locking = 5 % below the model's random-offset figure
so the redesign is worth at most ~5-9 % of the kernel, before the cost of 8-wave barriers, and was not built.
"""
import random

# (type, ticks in thousands): one round of one workgroup, in program order (barrier-delimited pieces of k_head_phase)
PIECES = [("S", 1.5), ("S", 5.0), ("S", 2.0), ("S", 1.0), ("S", 9.0),        # refill, march, scan, dense map, 3-D gather
          ("M", 8.2), ("S", 1.5), ("M", 16.4), ("S", 1.5), ("S", 12.0),       # amb L1 + sig L1a, store, amb L2, store, amb L3 + 2-D gather
          ("M", 4.1), ("S", 1.5), ("M", 16.4), ("S", 1.5), ("M", 17.4), ("S", 1.5),   # sig L1b, store, sig L2, store, sig L3 (+ sigma row), store
          ("M", 18.4), ("S", 1.5), ("S", 5.0), ("S", 2.0)]                    # col L1, store, col L2 rows, composite
BARRIER = 0.3
MFMA_PER_ROUND = sum(d for t, d in PIECES if t == "M")


def lock_step(offset):
    n, tot = len(PIECES), 0.0
    for i in range(n):
        a, b = PIECES[i], PIECES[(i - offset) % n]
        tot += (a[1] + b[1] if a[0] == b[0] == "M" else max(a[1], b[1])) + BARRIER
    return tot


def free_running(offset_ticks, rounds=200, jitter=0.0, seed=0):
    rnd = random.Random(seed)

    def dur(i):
        d = PIECES[i][1]
        return d * (1 + jitter * (rnd.random() * 2 - 1)) + BARRIER
    st = [{"i": 0, "rem": dur(0), "done": 0, "idle": 0.0}, {"i": 0, "rem": dur(0) + offset_ticks, "done": 0, "idle": offset_ticks}]
    t = t0 = 0.0
    d0 = None
    while min(s["done"] for s in st) < rounds:
        types = ["S" if s["idle"] > 0 else PIECES[s["i"]][0] for s in st]
        rate = 0.5 if types == ["M", "M"] else 1.0
        dt = min(s["rem"] / rate for s in st)
        t += dt
        for s in st:
            s["rem"] -= dt * rate
            s["idle"] = max(0.0, s["idle"] - dt)
            if s["rem"] <= 1e-9:
                s["i"] = (s["i"] + 1) % len(PIECES)
                s["done"] += s["i"] == 0
                s["rem"] = dur(s["i"])
        if d0 is None and min(s["done"] for s in st) >= 20:
            t0, d0 = t, sum(s["done"] for s in st)
    return (t - t0) / (sum(s["done"] for s in st) - d0) * 2


if __name__ == "__main__":
    print(f"MFMA ticks per round pair: {2 * MFMA_PER_ROUND:.0f} K (pipe never idle); one workgroup alone: {sum(d for _, d in PIECES):.0f} K per round")
    best = min((lock_step(o), o) for o in range(len(PIECES)))
    print(f"lock step on the natural barriers: best offset {best[1]} pieces -> {best[0]:.0f} K per round pair ({2 * MFMA_PER_ROUND / best[0]:.0%} pipe)")
    rows = [(free_running(o), o) for o in range(0, 130, 5)]
    print(f"free running, fixed offset, no jitter: best {min(rows)[0]:.0f} K at {min(rows)[1]} K ticks, worst {max(rows)[0]:.0f} K")
    rnd = [free_running(o, jitter=0.3, seed=o) for o in range(0, 130, 10)]
    print(f"free running, 30 % duration jitter (offset forgotten within a few rounds): mean {sum(rnd) / len(rnd):.0f} K "
          f"({2 * MFMA_PER_ROUND / (sum(rnd) / len(rnd)):.0%} pipe); measured on the GPU: 192 K (83 %)")
