"""The library's host-side table builder (nvmk_etkdg_molset_build / nvmk_ff_tables_build, csrc/table_build.cpp) in HOST mode,
row by row against the restatement of tests/table_model.py — the tables ``FlatMoleculeSet`` and ``MoleculeTermTables`` hand to
the kernels must be exactly what rounds 1-4 produced with torch.  No GPU: ``device="cpu"`` makes the builder write to host
memory through the same fill code; the GPU twin (pinned ring, chunked uploads) is tests/test_table_build_gpu.py."""

import numpy as np
import pytest

from nvmolkit_amd import _build, _native, synthetic
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet
from nvmolkit_amd.forcefield import DG, ETK, MMFF, UFF, MoleculeTermTables, stack_molecule_tables
from tests import table_model as tm


@pytest.fixture(scope="module", autouse=True)
def _built():
    _build.build()
    _native.lib()
    _native.pyglue()


@pytest.fixture(scope="module")
def library():
    return synthetic.druglike_library(40, seed=11, mean_atoms=30, processes=1)


def _check_molset(got, want):
    assert np.array_equal(got["n_atoms"], want["n_atoms"]) and np.array_equal(got["num_impropers"], want["num_impropers"])
    for g in range(3):
        tm.assert_groups_equal(got["dg"][g], want["dg"][g], f"dg group {g}")
    assert (got["etk"] is None) == (want["etk"] is None)
    if want["etk"] is not None:
        for g in range(6):
            tm.assert_groups_equal(got["etk"][g], want["etk"][g], f"etk group {g}")
        assert np.array_equal(got["d12"], np.diff(want["etk"][2][0])) and np.array_equal(got["d13"], np.diff(want["etk"][3][0]))
    assert (got["checks"] is None) == (want["checks"] is None)
    if want["checks"] is not None:
        for a, b in zip(got["checks"], want["checks"]):
            assert np.array_equal(a, b)


def test_molecule_set_equals_the_restatement(library):
    mols = [FlatMolecule(**m["embed"]) for m in library]
    _check_molset(tm.read_molset(FlatMoleculeSet(mols, device="cpu")), tm.expected_molset(mols))
    # one thread, many threads: the same tables
    _check_molset(tm.read_molset(FlatMoleculeSet(mols, device="cpu", preprocessing_threads=1)), tm.expected_molset(mols))
    _check_molset(tm.read_molset(FlatMoleculeSet(mols, device="cpu", preprocessing_threads=7)), tm.expected_molset(mols))


def test_molecule_set_takes_int32_arrays_lists_and_strided_views(library):
    """The glue references C-contiguous int32 / int64 / float64 arrays where they lie and converts anything else."""
    mols = []
    for k, m in enumerate(library[:9]):
        e = m["embed"]
        if k % 3 == 0:      # int32 indices
            conv = lambda g: (np.ascontiguousarray(g[0], dtype=np.int32), g[1])  # noqa: E731
        elif k % 3 == 1:    # plain lists
            conv = lambda g: (np.asarray(g[0]).tolist(), np.asarray(g[1]).tolist())  # noqa: E731
        else:               # strided views and float32 parameters that need a copy
            conv = lambda g: (np.asarray(np.repeat(np.asarray(g[0]), 2, axis=0)[::2]), np.asarray(g[1], dtype=np.float64)[::1])  # noqa: E731
        mols.append(FlatMolecule(e["n_atoms"], [conv(g) for g in e["dg"]], [conv(g) for g in e["etk"]], e["checks"], e["num_impropers"]))
    want = tm.expected_molset([FlatMolecule(**m["embed"]) for m in library[:9]])
    _check_molset(tm.read_molset(FlatMoleculeSet(mols, device="cpu")), want)


def test_stereo_checks_as_arrays_and_as_tuples_give_the_same_tables(library):
    """FlatMolecule.checks is a list of (kind, idx, par) tuples or an embedMolecules.StereoChecks (three arrays the glue references
    where they lie): both read the same — length, iteration, indexing, pickling — and build the same tables; a set may mix them."""
    import pickle

    from nvmolkit_amd.embedMolecules import StereoChecks

    with_arrays = [FlatMolecule(**m["embed"]) for m in library[:12]]
    assert all(isinstance(m.checks, StereoChecks) for m in with_arrays)
    with_tuples = [FlatMolecule(m.n_atoms, m.dg, m.etk, list(m.checks), m.num_impropers) for m in with_arrays]
    assert all(isinstance(c, tuple) and len(c) == 3 for m in with_tuples for c in m.checks)
    one = with_arrays[0].checks
    assert len(one) == len(with_tuples[0].checks) and one[1] == with_tuples[0].checks[1] and list(one) == with_tuples[0].checks
    again = pickle.loads(pickle.dumps(one))
    assert list(again) == list(one) and StereoChecks(list(one)).idx.tolist() == one.idx.tolist()
    want = tm.expected_molset(with_tuples)
    _check_molset(tm.read_molset(FlatMoleculeSet(with_arrays, device="cpu")), want)
    _check_molset(tm.read_molset(FlatMoleculeSet(with_tuples, device="cpu")), want)
    mixed = [a if k % 2 else t for k, (a, t) in enumerate(zip(with_arrays, with_tuples))]
    _check_molset(tm.read_molset(FlatMoleculeSet(mixed, device="cpu")), want)
    with pytest.raises(ValueError):
        StereoChecks([(0, (0, 1, 2, 3, 4, 5), ())])


def test_molecule_set_without_etk_without_checks_and_empty():
    rng = np.random.default_rng(3)
    e, _, _ = synthetic.synthetic_embed_molecule(rng, 9, with_etk=False)
    mols = [FlatMolecule(e["n_atoms"], e["dg"]), FlatMolecule(e["n_atoms"], e["dg"])]
    got = tm.read_molset(FlatMoleculeSet(mols, device="cpu"))
    assert got["etk"] is None and got["checks"] is None
    _check_molset(got, tm.expected_molset(mols))
    # one molecule with ETK groups, one without: the set has none (every molecule must bring them)
    e2, _, _ = synthetic.synthetic_embed_molecule(rng, 7, with_etk=True)
    mixed = FlatMoleculeSet([FlatMolecule(**e2), FlatMolecule(e["n_atoms"], e["dg"])], device="cpu")
    assert not mixed.has_etk
    empty = FlatMoleculeSet([], device="cpu")
    assert empty.c.n_mols == 0 and not empty.has_etk


def test_molecule_set_refuses_bad_input(library):
    e = library[0]["embed"]
    bad_idx = [(np.array(e["dg"][0][0]), e["dg"][0][1]), e["dg"][1], e["dg"][2]]
    bad_idx[0][0][3, 1] = e["n_atoms"]            # an atom index outside the molecule
    with pytest.raises(ValueError, match="atom index outside the molecule"):
        FlatMoleculeSet([FlatMolecule(e["n_atoms"], bad_idx)], device="cpu")
    with pytest.raises(ValueError, match="parameter values"):
        FlatMoleculeSet([FlatMolecule(e["n_atoms"], [(e["dg"][0][0], e["dg"][0][1][:-1]), e["dg"][1], e["dg"][2]])], device="cpu")
    with pytest.raises(ValueError, match="needs 3 term groups"):
        FlatMoleculeSet([FlatMolecule(e["n_atoms"], e["dg"][:2])], device="cpu")
    with pytest.raises(ValueError, match="whole rows"):
        FlatMoleculeSet([FlatMolecule(e["n_atoms"], [(np.arange(5), np.zeros((0, 3))), e["dg"][1], e["dg"][2]])], device="cpu")
    with pytest.raises(ValueError, match="at most 5 indices"):
        FlatMoleculeSet([FlatMolecule(e["n_atoms"], e["dg"], None, [(0, (1, 2, 3, 4, 5, 6), ())])], device="cpu")


@pytest.mark.parametrize("kind", [DG, ETK, MMFF, UFF])
def test_term_tables_equal_the_restatement(kind, library):
    if kind == MMFF:
        tables = [m["mmff"] for m in library]
    elif kind == UFF:
        rng = np.random.default_rng(5)
        tables = [synthetic.random_ff_system(UFF, int(n), rng)[1] for n in rng.integers(5, 40, 12)]
    else:
        tables = [m["embed"]["dg" if kind == DG else "etk"] for m in library]
    got, got_merged = tm.read_tables(MoleculeTermTables(kind, tables, device="cpu"))
    want, want_merged = tm.expected_term_tables(kind, tables)
    assert len(got) == len(want)
    for g, (a, b) in enumerate(zip(got, want)):
        tm.assert_groups_equal(a, b, f"kind {kind} group {g}")
    tm.assert_groups_equal(got_merged, want_merged, "merged non-bonded group")
    assert (want_merged is not None) == (kind == MMFF)
    # the stacked form (a batch without system_mol) goes through the same builder
    got2, merged2 = tm.read_tables(MoleculeTermTables.from_stacked(kind, stack_molecule_tables(kind, tables), device="cpu"))
    for g, (a, b) in enumerate(zip(got2, want)):
        tm.assert_groups_equal(a, b, f"stacked kind {kind} group {g}")
    tm.assert_groups_equal(merged2, want_merged, "stacked merged group")


def test_switches_keep_the_callers_order_and_the_separate_tables(library, monkeypatch):
    tables = [m["mmff"] for m in library[:10]]
    monkeypatch.setenv("NVMK_PAIR_ORDER", "input")
    got, merged = tm.read_tables(MoleculeTermTables(MMFF, tables, device="cpu"))
    want, want_merged = tm.expected_term_tables(MMFF, tables, pair_order=False)
    for g, (a, b) in enumerate(zip(got, want)):
        tm.assert_groups_equal(a, b, f"group {g}")
    tm.assert_groups_equal(merged, want_merged, "merged group in (min, max) order")
    monkeypatch.setenv("NVMK_MMFF_MERGE", "0")
    assert tm.read_tables(MoleculeTermTables(MMFF, tables, device="cpu"))[1] is None


def test_merged_group_only_when_every_molecule_allows_it(library):
    good = [m["mmff"] for m in library[:6]]
    assert tm.read_tables(MoleculeTermTables(MMFF, good, device="cpu"))[1] is not None
    t = list(good[3])
    vdw_idx, vdw_par = np.array(t[5][0]), np.array(t[5][1])
    for case in ("ele without vdw", "vdw twice", "ele twice"):
        bad = list(t)
        if case == "ele without vdw":
            bad[5] = (vdw_idx[1:], vdw_par[1:])
            bad[6] = (np.concatenate([t[6][0], vdw_idx[:1]]), np.concatenate([t[6][1], np.ones((1, 3))]))
        elif case == "vdw twice":
            bad[5] = (np.concatenate([vdw_idx, vdw_idx[:1, ::-1]]), np.concatenate([vdw_par, vdw_par[:1]]))
        else:
            bad[6] = (np.concatenate([t[6][0], t[6][0][:1]]), np.concatenate([t[6][1], t[6][1][:1]]))
        tables = good[:3] + [bad] + good[4:]
        groups, merged = tm.read_tables(MoleculeTermTables(MMFF, tables, device="cpu"))
        want, want_merged = tm.expected_term_tables(MMFF, tables)
        assert merged is None and want_merged is None, case
        for g, (a, b) in enumerate(zip(groups, want)):
            tm.assert_groups_equal(a, b, f"{case}: group {g}")       # the separate tables are complete all the same


def test_constraint_groups_travel_behind_the_force_field_groups(library):
    rng = np.random.default_rng(8)
    tables = [m["mmff"] for m in library[:5]]
    stacked = stack_molecule_tables(MMFF, tables)
    counts = rng.integers(0, 4, 5)
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    stacked.append((starts, rng.integers(0, 5, (int(starts[-1]), 2)), rng.standard_normal((int(starts[-1]), 3))))   # distance constraints
    stacked.append((np.zeros(6, dtype=np.int32), np.zeros((0, 1), dtype=np.int32), np.zeros((0, 5))))              # no position constraints
    t = MoleculeTermTables.from_stacked(MMFF, stacked, device="cpu")
    got, merged = tm.read_tables(t)
    want, want_merged = tm.expected_from_stacked(MMFF, stacked)
    assert len(got) == 9
    for g, (a, b) in enumerate(zip(got, want)):
        tm.assert_groups_equal(a, b, f"group {g}")
    tm.assert_groups_equal(merged, want_merged)
    with pytest.raises(ValueError, match="needs 3 term groups"):
        MoleculeTermTables.from_stacked(DG, stack_molecule_tables(DG, [m["embed"]["dg"] for m in library[:2]]) + [stacked[7]], device="cpu")
    with pytest.raises(ValueError, match="inconsistent"):
        MoleculeTermTables.from_stacked(MMFF, stacked[:5] + [(stacked[5][0], stacked[5][1][:-1], stacked[5][2])] + stacked[6:7], device="cpu")


def test_builder_through_the_c_abi_alone():
    """The entry points as a C caller uses them: descriptors filled by hand (ctypes), int32 indices, host mode."""
    import ctypes

    lib = _native.lib()
    idx = np.array([[0, 2], [0, 1], [1, 2]], dtype=np.int32)
    par = np.arange(9, dtype=np.float64).reshape(3, 3)
    w = np.array([[0], [1], [2]], dtype=np.int32)
    mol = _native.FlatMoleculeDesc()
    mol.n_atoms, mol.num_impropers = 3, 2
    mol.dg[0].n_terms, mol.dg[0].idx_bytes, mol.dg[0].idx, mol.dg[0].par = 3, 4, idx.ctypes.data, par.ctypes.data
    mol.dg[2].n_terms, mol.dg[2].idx_bytes, mol.dg[2].idx = 3, 4, w.ctypes.data
    handle = ctypes.c_void_p()
    _native.check(lib.nvmk_etkdg_molset_build(ctypes.addressof(mol), 1, 0, _native.BUILD_HOST, None, ctypes.byref(handle)))
    view = _native.EtkdgMolset()
    _native.check(lib.nvmk_etkdg_molset_view(handle, ctypes.byref(view)))
    starts, oidx, opar = tm.read_group(view.dg[0], 1, 2, 3, False)
    assert starts.tolist() == [0, 3] and oidx.tolist() == [[0, 1], [1, 2], [0, 2]] and opar.tolist() == [[3, 4, 5], [6, 7, 8], [0, 1, 2]]
    assert tm.read_group(view.dg[1], 1, 4, 2, False)[0].tolist() == [0, 0] and not view.dg[1].idx
    assert not view.check_starts and not view.h_etk_d12_counts
    assert np.frombuffer(ctypes.string_at(view.num_impropers, 4), dtype=np.int32)[0] == 2
    _native.check(lib.nvmk_etkdg_molset_free(handle))
    # argument errors
    assert lib.nvmk_etkdg_molset_build(None, 1, 0, _native.BUILD_HOST, None, ctypes.byref(handle)) == _native.ERR_INVALID_ARGUMENT
    assert lib.nvmk_etkdg_molset_build(ctypes.addressof(mol), 1, 0, _native.BUILD_HOST, None, None) == _native.ERR_INVALID_ARGUMENT
    mol.dg[0].idx_bytes = 2
    assert lib.nvmk_etkdg_molset_build(ctypes.addressof(mol), 1, 0, _native.BUILD_HOST, None, ctypes.byref(handle)) == _native.ERR_INVALID_ARGUMENT
    assert "int32 or int64" in _native.last_error()
    assert lib.nvmk_etkdg_molset_view(None, ctypes.byref(view)) == _native.ERR_INVALID_ARGUMENT
    terms = (_native.HostTerms * 3)()
    assert lib.nvmk_ff_tables_build(_native.FF_QUARTIC, ctypes.addressof(terms), 1, 0, 0, _native.BUILD_HOST, None, ctypes.byref(handle)) \
        == _native.ERR_INVALID_ARGUMENT
    assert lib.nvmk_ff_tables_build(DG, ctypes.addressof(terms), 1, 4, 0, _native.BUILD_HOST, None, ctypes.byref(handle)) == _native.ERR_INVALID_ARGUMENT
    assert lib.nvmk_etkdg_molset_free(None) == 0 and lib.nvmk_ff_tables_free(None) == 0


def test_default_builder_threads_are_this_process_share_of_the_host(monkeypatch):
    """One process per GPU: LOCAL_WORLD_SIZE processes assemble tables on one host at once (bench.py --gpus N, inside every rank's
    clock), so the default thread count divides the cores this process may run on among them; an explicit count is taken as is."""
    import os

    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    cores = len(os.sched_getaffinity(0))
    assert _native.build_threads(-1) == max(1, min(64, cores)) and _native.build_threads(0) == _native.build_threads(-1)
    assert _native.build_threads(7) == 7
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert _native.build_threads(-1) == max(1, min(64, cores // 8)) and _native.build_threads(7) == 7
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "nonsense")
    assert _native.build_threads(-1) == max(1, min(64, cores))
