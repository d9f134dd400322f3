"""Pins of the checker's integer layer (edo_hmm / edo_callcnvs) that do not depend on anyone's reading of it.

`src/hmm.cpp` includes <Rinternals.h>; R is not in the image and no stand-in header is written, so the file cannot be
executed here.  What this file holds instead is a SECOND restatement of its semantics, written independently of
oracle/ed_oracle.c and by a different method -- exhaustive enumeration instead of dynamic programming:

  * forward pass + trace-back (src/hmm.cpp:42-100): the score the reference accumulates for a state sequence is a
    left-to-right sum in a fixed association order, `(proba + previous) + log(trans)` (:79); because floating-point
    addition is monotone, its running maximum over predecessors (:81-84) equals the maximum of that accumulated score
    over ALL prefixes ending in the state.  So the forward table can be obtained by brute force over prefixes, the
    back-pointer as "the first k, in ascending order, whose best prefix extended by this step attains that maximum"
    (strict '>' keeps the first), and the path by following the pointers from the forced final state 0 (:96).
    Integer emissions and uniform transitions make exact ties the rule rather than the exception.
  * run-length summary (:104-126): stated as rules on the path (maximal stretches of non-zero states, runs of equal
    states inside a stretch) and checked on EVERY path of up to 9 observations.

Plus the reference-held facts for the bundled chr1 data with the exact (eta, phi) the survey's probe of the
reference's compiled C used.
"""
import itertools
import json
import math
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NINF = float("-inf")


def _log_trans(T, pos, L, i, j, k):
    """log of the transition k -> j across gap i, operation for operation as src/hmm.cpp:62-79 (libm via math)."""
    dist = float(pos[i]) - float(pos[i - 1])
    d = math.exp(-dist / L)
    if k == 0:
        tr = T[0][j]
    else:
        tr = d * T[k][j] + (1.0 - d) * T[0][j]
    return math.log(tr) if tr > 0 else NINF


def _enumerated_viterbi(T, ll, pos, L):
    """Path of src/hmm.cpp:42-100 by enumeration of prefixes (no recurrence on the forward table)."""
    n = len(ll)
    lt = [[[_log_trans(T, pos, L, i, j, k) if i > 0 else 0.0 for k in range(3)] for j in range(3)] for i in range(n)]

    def accumulate(seq):            # seq[0] == 0: the chain starts from (0, -inf, -inf)  (:48-52)
        sc = 0.0
        for i in range(1, len(seq)):
            sc = (ll[i][seq[i]] + sc) + lt[i][seq[i]][seq[i - 1]]
        return sc

    best = [[NINF] * 3 for _ in range(n)]   # best[i][j]: max accumulated score over prefixes 0 .. i ending in j
    best[0][0] = 0.0
    for i in range(1, n):
        for mid in itertools.product(range(3), repeat=i - 1):
            for j in range(3):
                sc = accumulate((0,) + mid + (j,))
                if sc > best[i][j]:
                    best[i][j] = sc
    n_ties = 0
    frm = [[0] * 3 for _ in range(n)]
    for i in range(1, n):
        for j in range(3):
            cands = [(ll[i][j] + best[i - 1][k]) + lt[i][j][k] for k in range(3)]
            m = max(cands)
            assert m == best[i][j] or (m != m)
            winners = [k for k in range(3) if cands[k] == m and m > NINF]
            n_ties += len(winners) > 1
            frm[i][j] = winners[0] if winners else 0     # -1 in the reference (then UB); 0 here and in the checker
            if ll[i][j] == NINF:
                frm[i][j] = 0                            # :87
    path = [0] * n
    for i in range(n - 1, 0, -1):
        path[i - 1] = frm[i][path[i]]
    return path, n_ties


def _calls_by_rules(path):
    """src/hmm.cpp:104-126 as rules on a path whose first and last observations are in state 0 (all the forward pass and
    the forced end, :96, can produce), 1-based start.p / end.p like the reference's table: every maximal run of equal
    non-zero states is a call, ended where the state changes; its `start` is the beginning of the maximal STRETCH of
    non-zero states it lies in (a direct deletion <-> duplication switch does not reset it); `nexons` is the run's length."""
    assert path[0] == 0 and path[-1] == 0
    calls = []
    x = 1
    while x < len(path):
        if path[x] == 0:
            x += 1
            continue
        a = x                                   # a stretch of non-zero states begins
        while path[x] != 0:
            y = x
            while path[y + 1] == path[x]:
                y += 1
            calls.append([a + 1, y + 1, path[x], y - x + 1])
            x = y + 1
    return calls


def test_viterbi_tie_breaking_by_enumeration(oracle):
    rng = np.random.default_rng(20250928)
    third = [[1 / 3] * 3] * 3
    t = 1e-2
    callcnvs_T = [[1 - t, t / 2, t / 2], [.5, .5, 0.], [.5, 0., .5]]
    ties_seen = 0
    for case in range(400):
        n = int(rng.integers(3, 10))
        T = third if case % 2 == 0 else callcnvs_T
        ll = rng.integers(-3, 1, size=(n, 3)).astype(float)      # small integers: exact ties
        if case % 5 == 0:
            ll[rng.integers(0, n), rng.integers(0, 3)] = NINF    # the :87 rule
        if case % 7 == 0:
            ll[:, 0] = 0.0; ll[:, 1:] = ll[:, 1:2]               # deletion and duplication equally likely everywhere
        pos = np.cumsum(rng.integers(1, 4000, n)).astype(np.int32)
        L = float(rng.choice([1.0, 3000.0, 50000.0]))
        want, nt = _enumerated_viterbi(T, ll.tolist(), pos.tolist(), L)
        ties_seen += nt
        got, calls = oracle.hmm(np.array(T), ll, pos, L)
        assert got.tolist() == want, (case, ll.tolist(), pos.tolist(), L)
        assert calls.tolist() == [[float(v) for v in c] for c in _calls_by_rules(want)], (case, want)
    assert ties_seen > 400          # the cases really are tie-ridden


def test_segmenter_on_every_short_path(oracle):
    """All 3^(n-2) paths of n <= 9 observations that start and end in state 0 (the forward pass starts from (0,-inf,-inf),
    the trace-back forces the end, :96), forced through the checker by one-hot emissions."""
    T = np.full((3, 3), 1 / 3)
    for n in range(2, 10):
        pos = np.arange(1, n + 1, dtype=np.int32) * 10
        for tail in itertools.product(range(3), repeat=n - 1):
            if tail[-1] != 0:
                continue          # the trace-back forces the last observation into state 0 (:96)
            path = (0,) + tail
            ll = np.full((n, 3), -1000.0)
            ll[np.arange(n), path] = 0.0
            got, calls = oracle.hmm(T, ll, pos, 1.0)
            assert tuple(got.tolist()) == path
            assert calls.tolist() == [[float(v) for v in c] for c in _calls_by_rules(list(path))], path


def test_segmenter_rules_match_the_documented_toy():
    # reference R/tools.R:74-85 + SURVEY 8c G3: path 0 0 0 2 2 2 1 1 1 0 -> calls [4,6,2,3], [4,9,1,3]
    assert _calls_by_rules([0, 0, 0, 2, 2, 2, 1, 1, 1, 0]) == [[4, 6, 2, 3], [4, 9, 1, 3]]
    assert _calls_by_rules([0, 1, 1, 1, 0]) == [[2, 4, 1, 3]]
    assert _calls_by_rules([0, 1, 2, 1, 0, 0, 2, 0]) == [[2, 2, 1, 1], [2, 3, 2, 1], [2, 4, 1, 1], [7, 7, 2, 1]]


def test_config1_with_the_survey_probes_exact_parameters(oracle):
    """SURVEY.md 8c (G4), recorded from the reference's own compiled C on Exome1 vs Exome2+3+4 of the bundled data
    with (eta, phi) = (-1.36727, 0.0049568): chr1 path state counts 26 320 / 121 / 108 over the 26 549 padded
    observations and a 25-row call table.  The checker is driven with exactly those parameters (not its own fit);
    the call table it gives is stored beside the counts as a regression vector (tests/golden/make_golden.py)."""
    d = np.load(os.path.join(G, "exomecount_chr1.npz"))
    fx = json.load(open(os.path.join(G, "config1_survey_probe.json")))
    assert fx["reference_facts"] == {"eta": -1.36727, "phi": 0.0049568, "padded_observations": 26549,
                                     "state_counts_padded": [26320, 121, 108], "n_calls": 25}
    counts = d["counts"]
    test = counts[:, 0].astype(np.int32)
    ref = counts[:, 1:].sum(axis=1).astype(np.int32)
    p = 1.0 / (1.0 + math.exp(-fx["reference_facts"]["eta"]))
    L, nerr = oracle.get_loglike_matrix(fx["reference_facts"]["phi"], p, test + ref, test, 1.0, oracle.LIBM)
    assert nerr == 0
    path, calls = oracle.callcnvs(L, np.array([0, test.size], np.int32), d["start"], d["end"])
    cnt = np.bincount(path, minlength=3)
    assert [int(cnt[0]) + 2, int(cnt[1]), int(cnt[2])] == fx["reference_facts"]["state_counts_padded"]   # + the two dummy rows
    assert len(calls) == fx["reference_facts"]["n_calls"]
    assert calls.astype(np.int64).tolist() == fx["calls_regression"]
    # every call is a run of its own type in the path, closed on both sides as the rules say
    for s, e, typ, nex in calls.astype(np.int64):
        assert np.all(path[s - 1:e] != 0) and path[e - 1] == typ and (e == path.size or path[e] != typ)
        assert nex == int(np.sum(path[e - nex:e] == typ)) and (e - nex == 0 or path[e - nex - 1] != typ)
