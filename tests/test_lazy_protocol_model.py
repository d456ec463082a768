"""Executable model of the blocked-lazy optimizer's ROW PROTOCOL in the step-ahead form: which launch touches
which table rows, and which launches may be in flight at the same time.  No kernel runs here -- the model restates what
csrc/optim.hip's parts do to a row's last-step word (`last`) and checks the two claims the design rests on:

  safety     a deferred window sweep S_t (launched behind the end-of-step launch TR(t), allowed to run under the next `depth`
             replays) never works on a row that an end-of-step launch running meanwhile also works on;
  exactness  every row receives every step exactly once, in order, the gradient of step t at step t, and a row is at state
             t - 1 when the gather of step t reads it -- i.e. what torch.optim.Adam's dense step does (reference:
             trainers/ctr_trainer.py:59-61,99), executed lazily.

The negative controls run the same model with one batch of lookahead less than the sweep's lifetime needs, and with the
lookahead switched off under the relaxed join: the safety check must then FAIL (the rows in question are the ~B*F/K per step the
lookahead exists for)."""
import numpy as np
import pytest


def simulate(rows, K, B, steps, depth, look, seed):
    """One table of `rows` rows, window divisor K, `steps` training steps of B lookups.  depth = replays a sweep may overlap,
    look = batches beyond the next one whose lookups in the coming sweep's window are refreshed early.
    Returns (conflicts, violations): rows worked on by a sweep and by an overlapping end-of-step launch; exactness failures."""
    rng = np.random.default_rng(seed)
    w = -(-rows // K)
    batches = [rng.integers(0, rows, B) for _ in range(steps + look + 3)]
    last = np.zeros(rows, dtype=np.int64)          # step every row is at
    applied = [[] for _ in range(rows)]            # (step, with_gradient) per row, in order
    grad_pending = np.zeros(rows, dtype=bool)      # gradient row not yet consumed
    violations = []

    def bring(r, t, who):
        # replay the zero-gradient steps last[r] + 1 .. t - 1, then step t with the gradient row if it is non-zero
        for s in range(last[r] + 1, t):
            applied[r].append((s, False))
        applied[r].append((t, bool(grad_pending[r])))
        grad_pending[r] = False
        last[r] = t

    def window(t):
        lo = ((t - 1) % K) * w
        return lo, min(rows, lo + w)

    # the ordinary head of the first step: refresh batch(1) to state 0 = nothing to do
    live = []          # sweeps in flight: (t, row set it may still work on, replays it may still overlap)
    conflicts = 0
    for t in range(1, steps + 1):
        # gather of step t reads batch(t): rows must be at state t - 1
        bt = np.unique(batches[t])
        if not (last[bt] == t - 1).all():
            violations.append(("stale row read by the gather", t))
        grad_pending[bt] = True                    # backward of step t leaves gradient rows
        # TR(t): touched rows of batch t + refresh of batch t + 1 + lookahead, every part claims by last < t
        lo, hi = window(t)
        tr_rows = set(bt.tolist()) | set(np.unique(batches[t + 1]).tolist())
        for j in range(2, 2 + look):
            nb = np.unique(batches[t + j])
            tr_rows |= set(nb[(nb >= lo) & (nb < hi)].tolist())
        # sweeps still in flight must not share a row with this launch
        for st, srows, left in live:
            conflicts += len(srows & tr_rows)
        for r in tr_rows:
            if last[r] < t:
                bring(r, t, "tr")
        if grad_pending.any():
            violations.append(("gradient row left behind", t))
        # sweeps age: the one that has overlapped `depth` replays is joined before the NEXT end-of-step launch
        live = [(st, srows, left - 1) for st, srows, left in live if left - 1 > 0]
        # S_t: launched behind TR(t); the rows it will work on = window rows behind step t NOW (it reads `last` when it gets there,
        # any time during its life -- rows stamped later by an overlapping launch are exactly the conflicts counted above)
        srows = {r for r in range(lo, hi) if last[r] < t}
        for r in srows:
            bring(r, t, "sweep")
        if depth > 0:  # (depth 0 = strict join: the sweep is joined before the next end-of-step launch)
            live.append((t, srows, depth))
    # flush: everything to the last step
    for r in range(rows):
        if last[r] < steps:
            for s in range(last[r] + 1, steps + 1):
                applied[r].append((s, False))
            last[r] = steps
    for r in range(rows):
        if [s for s, _ in applied[r]] != list(range(1, steps + 1)):
            violations.append(("row did not receive every step once, in order", r))
            break
    want = np.zeros((steps + 1, rows), dtype=bool)
    for t in range(1, steps + 1):
        want[t, np.unique(batches[t])] = True
    for r in range(rows):
        got = {s for s, g in applied[r] if g}
        if got != set(np.nonzero(want[:, r])[0].tolist()):
            violations.append(("gradient applied at the wrong step", r))
            break
    return conflicts, violations


@pytest.mark.parametrize("rows,K,B", [(4096, 64, 256), (1000, 8, 300), (50000, 128, 2048)])
def test_step_ahead_protocol_is_safe_and_exact(rows, K, B):
    """Default form: a sweep may run under the next LOOK_DEPTH replays (joined before the one after), lookahead over LOOK_DEPTH
    batches -- the constant the product uses (optim.LOOK_DEPTH: the depth of its sweep-event ring, the `look_depth` argument of
    rh_adam_lazy_step_ahead*), not a copy of it."""
    import torch_rechub_amd.optim as optim
    d = optim.LOOK_DEPTH
    assert K >= d + 2  # the product's own guard for this form (TableAdam._merge_ahead_ok)
    conflicts, violations = simulate(rows, K, B, steps=3 * K + 5, depth=d, look=d, seed=1)
    assert conflicts == 0 and violations == []
    # one batch of lookahead less than the product's constant is a race (the negative control below, at the product's depth)
    conflicts, _ = simulate(rows, K, B, steps=3 * K + 5, depth=d, look=d - 1, seed=1)
    assert conflicts > 0


def test_relaxed_join_protocol_is_safe_and_exact():
    """Eager-head form: a sweep may run under ONE further head (joined before the one after), lookahead over one batch."""
    conflicts, violations = simulate(4096, 64, 256, steps=200, depth=1, look=1, seed=2)
    assert conflicts == 0 and violations == []


@pytest.mark.parametrize("depth,look", [(2, 1), (1, 0)])
def test_one_batch_of_lookahead_less_is_a_race(depth, look):
    """Negative control: with less lookahead than the sweep's lifetime needs, sweeps and end-of-step launches DO meet on rows --
    the B * F / K rows per step the lookahead exists for.  (Exactness of the bookkeeping itself is unaffected in this
    sequential model; on the device the two launches would be writing the same row concurrently.)"""
    conflicts, violations = simulate(4096, 64, 256, steps=200, depth=depth, look=look, seed=3)
    assert conflicts > 50 and violations == []


def test_strict_join_needs_no_lookahead():
    conflicts, violations = simulate(4096, 64, 256, steps=200, depth=0, look=0, seed=4)
    assert conflicts == 0 and violations == []


@pytest.mark.parametrize("K,safe", [(2, False), (3, True), (4, True)])
def test_step_ahead_needs_windows_of_three_consecutive_steps_to_be_disjoint(K, safe):
    """With LOOK_DEPTH = 2 the sweeps of steps t - 2 and t - 1 may be in flight while the end-of-step launch of step t claims
    rows of window(t): at K = 2, window(t) IS window(t - 2) and the two meet (round-4 advisor finding).  optim.TableAdam
    therefore takes the step-ahead form only for lazy_k >= LOOK_DEPTH + 2 (one window of margin) and the relaxed / strict
    head otherwise (the relaxed join, one sweep in flight, is safe from K = 2 on)."""
    conflicts, violations = simulate(1200, K, 100, steps=60, depth=2, look=2, seed=5)
    assert violations == [] and (conflicts == 0) == safe
    assert simulate(1200, K, 100, steps=60, depth=1, look=1, seed=5) == (0, [])


def test_step_ahead_guard_matches_the_model():
    import torch_rechub_amd.optim as optim
    src = open(optim.__file__).read()
    assert "self.lazy_k >= LOOK_DEPTH + 2" in src and optim.LOOK_DEPTH == 2


def simulate_foreign(rows, K, B, ranks, steps, join_before_touched, seed):
    """The data-parallel step with REPLICATED tables (strict deferred form): the head of step t refreshes the rows of the LOCAL
    batch, the sweep of step t - 1 is launched behind it and runs beside the step, and the end-of-step touched pass works on
    the rows of EVERY rank's batch (gathered index matrix).  Returns the number of rows the sweep and the touched pass both
    work on while both are in flight."""
    rng = np.random.default_rng(seed)
    w = -(-rows // K)
    last = np.zeros(rows, dtype=np.int64)
    conflicts = 0
    for t in range(1, steps + 1):
        local = np.unique(rng.integers(0, rows, B))
        foreign = np.unique(rng.integers(0, rows, B * (ranks - 1)))
        last[local] = np.maximum(last[local], t - 1)          # head: pre-gather refresh of the local batch (stamps the rows)
        lo = ((t - 2) % K) * w if t > 1 else 0
        srows = {r for r in range(lo, min(rows, lo + w)) if last[r] < t - 1} if t > 1 else set()
        touched = set(local.tolist()) | set(foreign.tolist())
        if not join_before_touched:
            conflicts += len(srows & touched)                  # the sweep may still be on these rows when they are stepped
        for r in srows:
            last[r] = t - 1
        for r in touched:
            last[r] = t
    return conflicts


def test_foreign_rows_race_with_the_deferred_sweep_unless_it_is_joined_first():
    """Round 5: optim.TableAdam._join_before_foreign_rows.  With one rank (or row-sharded tables) every row the touched pass
    steps was stamped by the head, so the deferred sweep never meets it; with the rows of other ranks' batches it does."""
    assert simulate_foreign(4096, 16, 128, ranks=1, steps=80, join_before_touched=False, seed=7) == 0
    assert simulate_foreign(4096, 16, 128, ranks=4, steps=80, join_before_touched=False, seed=7) > 50
    assert simulate_foreign(4096, 16, 128, ranks=4, steps=80, join_before_touched=True, seed=7) == 0
