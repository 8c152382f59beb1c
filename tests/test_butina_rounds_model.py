"""The parallel rounds of fused Butina (nvmolkit_amd/csrc/butina.hip: par_mark / par_decide / par_subtract sweeps over a degree
bucket, par_emit, par_final) restated step by step in Python and held to the sequential greedy loop of the C oracle
(orc_butina_from_pairs) on thousands of small graphs — the claim the device code rests on, checked without a GPU: the sweeps select
the lexicographically first maximal independent set of a bucket's candidates under "live closed neighbourhoods intersect",
which is what one-round-at-a-time processing selects, cluster order included.  (The GPU tests then hold the kernels themselves
to the oracle: tests/test_clustering_gpu.py.)"""

import numpy as np
import pytest

import oracle

DEAD = -1
NO_OWNER = 1 << 30


def parallel_rounds(n, counts, pairs):
    """Clusters (centroid first, members ascending; greedy list then the ascending singleton tail) by the device algorithm."""
    counts = [int(c) for c in counts]
    nbr = [[] for _ in range(n)]
    for i, j in pairs:
        nbr[i].append(j)
        nbr[j].append(i)
    member_of = [-1] * n
    greedy = []
    while True:
        alive = [c for c in counts if c > 0]
        degree = max(alive) if alive else 0
        if degree < 2:
            break
        cand = [r for r in range(n - 1, -1, -1) if counts[r] == degree]  # c_1 > c_2 > ...: the sequential loop takes the LAST row
        status = [0] * len(cand)
        undecided = list(range(len(cand)))
        sweeps = 0
        while undecided:
            sweeps += 1
            assert sweeps <= len(cand) + 2, "every sweep decides at least the first undecided candidate"
            owner = {}
            # mark: every undecided candidate that still has the bucket's degree stamps its live closed neighbourhood
            nxt = []
            for k in undecided:
                if status[k] != 0:
                    continue
                c = cand[k]
                if counts[c] != degree:
                    status[k] = -1
                    continue
                nxt.append(k)
                owner[c] = min(owner.get(c, NO_OWNER), k)
                for v in nbr[c]:
                    if counts[v] > 0:
                        owner[v] = min(owner.get(v, NO_OWNER), k)
            undecided = nxt
            # decide: own stamp on the whole live closed neighbourhood (no degree is read: members may die in the same step)
            selected = [k for k in undecided
                        if owner[cand[k]] == k and all(owner.get(v, NO_OWNER) in (NO_OWNER, k) for v in nbr[cand[k]])]
            for k in selected:
                c = cand[k]
                status[k] = 1
                counts[c] = DEAD
                for v in nbr[c]:
                    if owner.get(v, NO_OWNER) == k:
                        counts[v] = DEAD
                        member_of[v] = c
            # subtract: the members of this sweep's clusters leave their surviving neighbours
            for k in selected:
                c = cand[k]
                for v in nbr[c]:
                    if member_of[v] != c:
                        continue
                    for j in nbr[v]:
                        if counts[j] > 0:
                            counts[j] -= 1
        for k in range(len(cand)):  # emit in candidate order: every cluster of the bucket has exactly `degree` rows
            if status[k] == 1:
                c = cand[k]
                members = sorted(v for v in nbr[c] if member_of[v] == c)
                assert len(members) + 1 == degree
                greedy.append((c, *members))
    # the sequential loop's last act (par_final_kernel)
    if greedy:
        last = [j for m in greedy[-1] for j in nbr[m] if counts[j] == 1]
    else:
        last = [i for i in range(n) if counts[i] == 1]
    if last:
        top = max(last)
        greedy.append((top,))
        counts[top] = DEAD
    tail = sorted(i for i in range(n) if counts[i] >= 0)  # degree 1 (harvested) and degree 0 rows
    return greedy + [(i,) for i in tail]


def reference_rounds(n, counts, pairs):
    clusters, _, _ = oracle.butina_from_pairs(n, np.asarray(counts, dtype=np.int32), np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
    return [tuple(int(x) for x in c) for c in clusters]


def graph(n, pairs, no_self=()):
    deg = np.ones(n, dtype=np.int64)
    for i, j in pairs:
        deg[i] += 1
        deg[j] += 1
    for i in no_self:  # an all-zero fingerprint is not its own neighbour (and has none)
        deg[i] = 0
    return deg


@pytest.mark.parametrize("seed", range(40))
def test_random_graphs(seed):
    rng = np.random.default_rng(seed)
    for _ in range(60):
        n = int(rng.integers(1, 48))
        p = float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.4, 0.8]))
        iu = np.triu_indices(n, 1)
        keep = rng.random(len(iu[0])) < p
        pairs = [(int(a), int(b)) for a, b in zip(iu[0][keep], iu[1][keep])]
        isolated = [i for i in range(n) if all(i not in e for e in pairs)]
        no_self = [i for i in isolated if rng.random() < 0.3]
        counts = graph(n, pairs, no_self)
        assert parallel_rounds(n, counts, pairs) == reference_rounds(n, counts, pairs)


def test_structured_graphs():
    cases = []
    for n in (2, 3, 7, 20, 33):
        cases.append((n, [(i, i + 1) for i in range(n - 1)]))                      # a path: one long dependency chain per bucket
        cases.append((n, [(i, (i + 1) % n) for i in range(n)] if n > 2 else [(0, 1)]))  # a ring
        cases.append((n, [(0, i) for i in range(1, n)]))                           # a star
        cases.append((n, [(i, j) for i in range(n) for j in range(i + 1, n)]))     # a clique
    # equal cliques chained by single edges: every candidate of a bucket conflicts with its neighbours in the chain
    for size, copies in ((3, 6), (4, 5), (5, 4)):
        pairs, n = [], size * copies
        for b in range(copies):
            pairs += [(b * size + i, b * size + j) for i in range(size) for j in range(i + 1, size)]
            if b:
                pairs.append((b * size - 1, b * size))
        cases.append((n, pairs))
    # a ladder and a grid: many rows of one degree with overlapping neighbourhoods
    w = 6
    cases.append((2 * w, [(i, i + 1) for i in range(w - 1)] + [(w + i, w + i + 1) for i in range(w - 1)] + [(i, w + i) for i in range(w)]))
    g = 5
    cases.append((g * g, [(r * g + c, r * g + c + 1) for r in range(g) for c in range(g - 1)] + [(r * g + c, (r + 1) * g + c) for r in range(g - 1) for c in range(g)]))
    cases.append((5, []))                                                          # no edges at all
    for n, pairs in cases:
        pairs = sorted(set((min(a, b), max(a, b)) for a, b in pairs if a != b))
        counts = graph(n, pairs)
        assert parallel_rounds(n, counts, pairs) == reference_rounds(n, counts, pairs), (n, pairs)
