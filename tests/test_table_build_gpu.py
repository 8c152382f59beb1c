"""The table builder's DEVICE path: host threads -> ring of pinned staging slots -> chunked hipMemcpyAsync -> one device block
(csrc/table_build.cpp).  The tables read back from the GPU must equal the restatement of tests/table_model.py row by row — with
the default 32 MB slots (one chunk here) and with slots so small that the ring is reused many times; on a side stream; from two
host threads at once (two rings); and a minimisation on builder-made tables gives the bits of one on per-call tables."""

import threading

import numpy as np
import pytest
import torch

from nvmolkit_amd import _native, mmffOptimization, synthetic
from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat
from nvmolkit_amd.forcefield import MMFF, MoleculeTermTables
from nvmolkit_amd.types import CoordinateOutput
from tests import table_model as tm
from tests.test_table_build import _check_molset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def library():
    return synthetic.druglike_library(120, seed=13, mean_atoms=34, processes=4)


@pytest.mark.parametrize("slot_kb", [None, 64, 700])
def test_device_tables_equal_the_restatement(library, slot_kb):
    mols = [FlatMolecule(**m["embed"]) for m in library]
    tables = [m["mmff"] for m in library]
    with _native.options(NVMK_BUILD_SLOT_KB=slot_kb):
        molset = FlatMoleculeSet(mols)
        resident = MoleculeTermTables(MMFF, tables)
    _check_molset(tm.read_molset(molset), tm.expected_molset(mols))
    got, merged = tm.read_tables(resident)
    want, want_merged = tm.expected_term_tables(MMFF, tables)
    for g, (a, b) in enumerate(zip(got, want)):
        tm.assert_groups_equal(a, b, f"group {g}")
    tm.assert_groups_equal(merged, want_merged, "merged group")


def test_builds_on_a_side_stream_and_from_two_threads(library):
    mols = [FlatMolecule(**m["embed"]) for m in library]
    want = tm.expected_molset(mols)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), _native.options(NVMK_BUILD_SLOT_KB=128):
        molset = FlatMoleculeSet(mols)
    side.synchronize()
    _check_molset(tm.read_molset(molset), want)
    results, errors = [None, None], []

    def work(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                results[k] = FlatMoleculeSet(mols[k::2], preprocessing_threads=3)
                torch.cuda.current_stream().synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    with _native.options(NVMK_BUILD_SLOT_KB=96):
        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    for k in range(2):
        _check_molset(tm.read_molset(results[k]), tm.expected_molset(mols[k::2]))


def test_embedding_does_not_depend_on_how_the_tables_were_staged(library):
    mols = [FlatMolecule(**m["embed"]) for m in library[:40]]
    runs = []
    for slot_kb in (None, 64):
        with _native.options(NVMK_BUILD_SLOT_KB=slot_kb):
            res = embed_flat(FlatMoleculeSet(mols), confs_per_molecule=2, max_iterations=10, seed=4)
        runs.append((res.coords.cpu().numpy(), res.conf_counts.copy()))
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][0], runs[1][0])


def test_minimisation_on_resident_tables_while_they_are_still_uploading(library):
    """The builder returns with its last uploads in flight on the current stream; work enqueued behind them sees whole tables."""
    lib = library[:60]
    tables = [m["mmff"] for m in lib]
    dev = embed_flat(FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib]), confs_per_molecule=2, max_iterations=10, seed=2,
                     output=CoordinateOutput.DEVICE)
    want = mmffOptimization.optimize_device(tables, dev, max_iters=30)
    for _ in range(3):
        with _native.options(NVMK_BUILD_SLOT_KB=64):
            got = mmffOptimization.optimize_device(mmffOptimization.resident_tables(tables), dev, max_iters=30)
        assert torch.equal(got.values.torch(), want.values.torch()) and torch.equal(got.energies.torch(), want.energies.torch())


def test_asynchronous_build_fills_under_the_embedding_and_survives_an_early_free(library):
    """NVMK_BUILD_ASYNC (FlatMoleculeSet's default on a GPU): the constructor returns once the tables are planned, the library's
    threads fill and upload the rows in molecule order, nvmk_etkdg_embed waits for a batch's molecules before the batch runs
    (reference structure: src/etkdg.cpp:175-191,211-240).  Same tables and the same conformers as the blocking build — with
    tiny staging slots so that many chunks are still being filled when the first batch starts, and small batches so that later
    batches meet later chunks; a consumer on ANOTHER stream waits through .wait(stream); and a set that is dropped while its
    fill is still running is freed safely (the free joins the fill before the handle's arrays go)."""
    mols = [FlatMolecule(**m["embed"]) for m in library]
    want = tm.expected_molset(mols)
    with _native.options(NVMK_BUILD_SLOT_KB=16):
        blocking = FlatMoleculeSet(mols, asynchronous=False)
        ref = embed_flat(blocking, confs_per_molecule=2, max_iterations=10, seed=6, batch_size=40)
        for _ in range(3):
            molset = FlatMoleculeSet(mols)
            res = embed_flat(molset, confs_per_molecule=2, max_iterations=10, seed=6, batch_size=40)  # first batch: molecules 0 .. 19
            assert np.array_equal(res.conf_counts, ref.conf_counts) and torch.equal(res.coords, ref.coords)
            _check_molset(tm.read_molset(molset), want)
        side = torch.cuda.Stream()
        molset = FlatMoleculeSet(mols)
        molset.wait(side)
        with torch.cuda.stream(side):
            res = embed_flat(molset, confs_per_molecule=2, max_iterations=10, seed=6, batch_size=40, stream=side)
        side.synchronize()
        assert torch.equal(res.coords, ref.coords)
        for _ in range(20):  # dropped at once: the finalizer runs while the fill is in flight
            FlatMoleculeSet(mols)
    torch.cuda.synchronize()
