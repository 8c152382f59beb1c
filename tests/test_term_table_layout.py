"""The restatement the table builder is held to (tests/table_model.py: the diagonal ordering of the O(N^2) pair groups and the
merge of the MMFF van der Waals and electrostatic tables) checked against first principles — pure index work on the CPU.
tests/test_table_build.py then compares the library's builder with it row by row."""

import numpy as np
import torch

from tests.table_model import diagonal_pair_order, merge_mmff_nonbonded


def _random_pair_group(rng, n_systems, n_atoms, n_par, density=0.6):
    starts, idx, par = [0], [], []
    for _ in range(n_systems):
        pairs = [(i, j) for i in range(n_atoms) for j in range(i + 1, n_atoms) if rng.random() < density]
        rng.shuffle(pairs)
        pairs = [(j, i) if rng.random() < 0.3 else (i, j) for i, j in pairs]      # either orientation
        idx += pairs
        par += rng.standard_normal((len(pairs), n_par)).tolist()
        starts.append(len(idx))
    return (torch.tensor(starts, dtype=torch.int32), torch.tensor(idx, dtype=torch.int32).reshape(-1, 2),
            torch.tensor(par, dtype=torch.float64).reshape(-1, n_par))


def test_diagonal_order_is_a_permutation_inside_every_system():
    rng = np.random.default_rng(0)
    starts, idx, par = _random_pair_group(rng, 5, 12, 3)
    oidx, opar = diagonal_pair_order(starts, idx, par)
    for s in range(5):
        lo, hi = int(starts[s]), int(starts[s + 1])
        before = sorted((tuple(i), tuple(p)) for i, p in zip(idx[lo:hi].tolist(), par[lo:hi].tolist()))
        after = sorted((tuple(i), tuple(p)) for i, p in zip(oidx[lo:hi].tolist(), opar[lo:hi].tolist()))
        assert before == after                                          # same terms, parameters still attached
        key = [(abs(a - b), min(a, b)) for a, b in oidx[lo:hi].tolist()]
        assert key == sorted(key)                                       # walked diagonal by diagonal
    # along a diagonal consecutive terms touch distinct atoms on both sides: no two neighbours share their first atom
    d1 = [t for t in oidx[: int(starts[1])].tolist() if abs(t[0] - t[1]) == 1]
    assert all(min(a) != min(b) for a, b in zip(d1, d1[1:]))


def test_merge_puts_the_electrostatic_parameters_on_the_matching_van_der_waals_row():
    rng = np.random.default_rng(1)
    s5, i5, p5 = _random_pair_group(rng, 4, 10, 2, density=0.8)
    keep = rng.random(len(i5)) < 0.6                                     # electrostatics on a subset of the pairs, other order
    sel = np.flatnonzero(keep)
    counts = [int(((sel >= int(s5[s])) & (sel < int(s5[s + 1]))).sum()) for s in range(4)]
    perm = np.concatenate([rng.permutation(sel[(sel >= int(s5[s])) & (sel < int(s5[s + 1]))]) for s in range(4)]).astype(np.int64)
    i6 = i5[perm].flip(1)                                                # opposite orientation
    p6 = torch.tensor(rng.standard_normal((len(perm), 3)))
    s6 = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32)
    starts, idx, par = merge_mmff_nonbonded((s5, i5, p5), (s6, i6, p6))
    assert torch.equal(starts, s5) and par.shape == (len(i5), 5)
    want = {}
    for s in range(4):
        for k in range(int(s5[s]), int(s5[s + 1])):
            want[(s, *sorted(i5[k].tolist()))] = p5[k].tolist() + [0.0, 0.0, 0.0]
        for k in range(int(s6[s]), int(s6[s + 1])):
            want[(s, *sorted(i6[k].tolist()))][2:] = p6[k].tolist()
    for s in range(4):
        for k in range(int(starts[s]), int(starts[s + 1])):
            assert par[k].tolist() == want[(s, *sorted(idx[k].tolist()))]


def test_merge_refuses_lists_that_do_not_match():
    s5 = torch.tensor([0, 2], dtype=torch.int32)
    i5 = torch.tensor([[0, 3], [1, 2]], dtype=torch.int32)
    p5 = torch.ones((2, 2), dtype=torch.float64)
    ele = lambda pairs: (torch.tensor([0, len(pairs)], dtype=torch.int32), torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2),  # noqa: E731
                         torch.ones((len(pairs), 3), dtype=torch.float64))
    assert merge_mmff_nonbonded((s5, i5, p5), ele([[0, 3]])) is not None
    assert merge_mmff_nonbonded((s5, i5, p5), ele([])) is not None                         # no charges at all
    assert merge_mmff_nonbonded((s5, i5, p5), ele([[0, 2]])) is None                       # a pair without van der Waals term
    assert merge_mmff_nonbonded((s5, i5, p5), ele([[0, 3], [3, 0]])) is None               # listed twice
    assert merge_mmff_nonbonded((s5, torch.tensor([[0, 3], [3, 0]], dtype=torch.int32), p5), ele([[0, 3]])) is None
    assert merge_mmff_nonbonded((torch.tensor([0, 0], dtype=torch.int32), i5[:0], p5[:0]), ele([])) is None   # nothing to merge
