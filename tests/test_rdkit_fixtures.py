"""Consumers of the RDKit-made fixtures of tests/golden/rdkit/ (written by tests/golden/make_rdkit_fixtures.py on a host that has
RDKit — none has run it yet, so today every fixture test here SKIPS with that message).  They need NO RDKit themselves: once the
files are committed, the CPU oracle (``-m "not gpu"``) and the kernels (``-m gpu``) are held to RDKit's own numbers on every host,
with the reference's tolerances (tests/test_mmff.cu:51-56: energy 5e-5, gradient 1e-4).

So that the checkers are known to work the day the files arrive, each one also runs on a SELF-MADE fixture in the same format
whose "RDKit" numbers come from the independent numpy restatement oracle/ff.py (force fields) or from oracle/smiles.py's
exhaustive search (self matches): same file layout, same code path, different source of truth."""

import json
from pathlib import Path

import numpy as np
import pytest

from nvmolkit_amd import synthetic
from nvmolkit_amd.forcefield import GROUP_LAYOUT, MMFF, UFF
from oracle import ff as off
from oracle import ffc

FIXTURES = Path(__file__).parent / "golden" / "rdkit"
MMFF_TERMS = ["Bond", "Angle", "StretchBend", "Oop", "Torsion", "VdW", "Ele"]
FUNCTION_E_TOL, GRAD_TOL = 5.0e-5, 1.0e-4     # tests/test_mmff.cu:53-55


def need(name: str) -> Path:
    path = FIXTURES / name
    if not path.exists():
        pytest.skip(f"{path.relative_to(FIXTURES.parents[2])} is not there yet: run tests/golden/make_rdkit_fixtures.py on a host with RDKit")
    return path


def groups_of(z, key: str, kind: int):
    return [(z[f"{key}_g{g}_idx"].reshape(-1, n_idx), z[f"{key}_g{g}_par"].reshape(-1, n_par)) for g, (n_idx, n_par) in enumerate(GROUP_LAYOUT[kind])]


def only(groups, k):
    return [g if (k is None or i == k) else (g[0][:0], g[1][:0]) for i, g in enumerate(groups)]


def stacked(groups):
    return [(np.array([0, len(idx)], dtype=np.int32), idx, par) for idx, par in groups]


# ---- the checkers: fixture file + an evaluator (CPU oracle or GPU kernels) -----------------------------------------------------
def cpu_eval(kind, groups, pos):
    b = ffc.Batch(kind, np.array([0, len(pos)], dtype=np.int32), stacked(groups))
    return float(b.energy(pos.reshape(-1))[0]), b.gradient(pos.reshape(-1))


def gpu_eval(kind, groups, pos):
    import torch

    from nvmolkit_amd.forcefield import FlatForcefieldBatch

    b = FlatForcefieldBatch(kind, np.array([0, len(pos)], dtype=np.int32), stacked(groups))
    p = torch.from_numpy(pos.reshape(-1).copy()).cuda()
    return float(b.compute_energy(p)[0]), b.compute_gradient(p).cpu().numpy()


def check_force_field_fixture(path: Path, kind: int, evaluate, terms) -> int:
    z = np.load(path)
    n = int(z["n_molecules"])
    for m in range(n):
        key, pos = f"m{m}", z[f"m{m}_pos"]
        groups = groups_of(z, key, kind)
        for k, term in terms:
            e, g = evaluate(kind, only(groups, k), pos)
            want_e, want_g = float(z[f"{key}_energy_{term}"]), z[f"{key}_grad_{term}"]
            assert abs(e - want_e) <= FUNCTION_E_TOL * max(1.0, abs(want_e) * 1e-3), (path.name, m, term, e, want_e)
            assert np.max(np.abs(g - want_g)) <= GRAD_TOL * max(1.0, np.max(np.abs(want_g)) * 1e-3), (path.name, m, term)
    return n


MMFF_CASES = list(enumerate(MMFF_TERMS)) + [(None, "All")]


def check_self_matches_fixture(path: Path) -> int:
    from nvmolkit_amd.fingerprints import SmilesSet

    rows = json.loads(path.read_text())
    s = SmilesSet([r["smiles"] for r in rows], perceive_aromaticity=True)
    for i, r in enumerate(rows):
        assert s.status[i] == 0 and int(s.n_atoms[i]) == r["atoms"], r["smiles"]
        plain = s.self_matches(i, symmetrize_terminal_groups=False)
        assert sorted(map(tuple, plain.tolist())) == sorted(map(tuple, r["plain"])), r["smiles"]
        sym = s.self_matches(i, symmetrize_terminal_groups=True)
        assert len(sym) == r["matches_terminal_groups_symmetric"], (r["smiles"], len(sym), r["matches_terminal_groups_symmetric"])
        if "symmetric" in r:
            assert sorted(map(tuple, sym.tolist())) == sorted(map(tuple, r["symmetric"])), r["smiles"]
    return len(rows)


def check_morgan_fixture(path: Path, fingerprints) -> int:
    """``fingerprints(smiles_set, ids, radius, bits)`` -> (len(ids), bits // 32) uint32."""
    from nvmolkit_amd.fingerprints import SmilesSet

    z = np.load(path)
    smiles = [line.split()[0] for line in (Path(__file__).parent / "golden" / "chembl_1k.smi").read_text().splitlines() if line.strip()]
    s = SmilesSet(smiles, perceive_aromaticity=True)
    assert np.array_equal(s.status == 0, z["parsed"] != 0), "the library ingests exactly the molecules RDKit parses"
    ids = np.flatnonzero(s.status == 0)
    for key in [k for k in z.files if k.startswith("r")]:
        radius, bits = int(key[1:key.index("_")]), int(key[key.index("b") + 1:])
        got = fingerprints(s, ids, radius, bits)
        bad = np.flatnonzero((got != z[key][ids]).any(axis=1))
        assert len(bad) == 0, (key, [smiles[ids[i]] for i in bad[:5]])
    return len(ids)


def oracle_fingerprints(s, ids, radius, bits):
    import oracle

    size = np.maximum(s.n_atoms, s.n_bonds)
    out = np.zeros((len(ids), bits // 32), dtype=np.uint32)
    lo = 0
    for stride in (32, 64, 128, 256, 512, 1024):
        sel = np.flatnonzero((size[ids] >= lo) & (size[ids] < stride))
        lo = stride
        if len(sel):
            out[sel] = oracle.morgan_fingerprints(*s.morgan_inputs(ids[sel], stride), stride, radius, bits)
    return out


# ---- RDKit-made fixtures (skipped until they exist) ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mmff_MMFF94_dative_first5.npz", "mmff_MMFF94_dative_every4th.npz", "mmff_MMFF94_hypervalent_every4th.npz",
                                  "mmff_larger_molecules.npz"])
def test_oracle_mmff_terms_equal_rdkit(name):
    assert check_force_field_fixture(need(name), MMFF, cpu_eval, MMFF_CASES) > 0


@pytest.mark.parametrize("name", ["uff_MMFF94_dative_first5.npz", "uff_MMFF94_dative_every4th.npz", "uff_larger_molecules.npz"])
def test_oracle_uff_equals_rdkit(name):
    assert check_force_field_fixture(need(name), UFF, cpu_eval, [(None, "All")]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mmff_MMFF94_dative_first5.npz", "mmff_MMFF94_dative_every4th.npz", "mmff_MMFF94_hypervalent_every4th.npz",
                                  "mmff_larger_molecules.npz"])
def test_kernel_mmff_terms_equal_rdkit(name):
    assert check_force_field_fixture(need(name), MMFF, gpu_eval, MMFF_CASES) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["uff_MMFF94_dative_first5.npz", "uff_MMFF94_dative_every4th.npz", "uff_larger_molecules.npz"])
def test_kernel_uff_equals_rdkit(name):
    assert check_force_field_fixture(need(name), UFF, gpu_eval, [(None, "All")]) > 0


def test_oracle_minimiser_reaches_rdkits_minimised_energies():
    """RDKit's own BFGS from the file's coordinates, 200 iterations: the oracle's minimiser (a restatement of it) on the fixture's
    tables ends at the same energy (1e-3, tests/test_mmff.cu:55) wherever RDKit converged."""
    mins = json.loads(need("mmff_minimised.json").read_text())
    for name, rows in mins.items():
        z = np.load(need(f"mmff_{Path(name).stem}.npz"))
        by_record = {int(z[f"m{m}_record"]): m for m in range(int(z["n_molecules"]))}
        for r in rows:
            if r["not_converged_200"] or r["record"] not in by_record:
                continue
            m = by_record[r["record"]]
            b = ffc.Batch(MMFF, np.array([0, r["atoms"]], dtype=np.int32), stacked(groups_of(z, f"m{m}", MMFF)))
            _, e, st, _ = b.minimize(z[f"m{m}_pos"].reshape(-1), max_iters=200)
            assert st[0] == 0 and abs(e[0] - r["energy_200"]) <= 1.0e-3 * max(1.0, abs(r["energy_200"])), (name, r["record"], e[0], r["energy_200"])


def test_distance_terms_carry_rdkits_bounds():
    """tests/test_flattened_builder.cu:188-189: the distance-violation terms hold the squares of RDKit's smoothed bounds."""
    for name in ("etkdg_MMFF94_dative_first5.npz", "etkdg_MMFF94_dative_every4th.npz"):
        z = np.load(need(name))
        for m in range(int(z["n_molecules"])):
            bounds, idx, par = z[f"m{m}_bounds"], z[f"m{m}_dg_g0_idx"].reshape(-1, 2), z[f"m{m}_dg_g0_par"].reshape(-1, 3)
            n = bounds.shape[0]
            assert len(idx) == n * (n - 1) // 2
            lo, hi = np.minimum(idx[:, 0], idx[:, 1]), np.maximum(idx[:, 0], idx[:, 1])
            np.testing.assert_allclose(par[:, 0], bounds[hi, lo] ** 2, rtol=1e-12)     # lower bounds below the diagonal
            np.testing.assert_allclose(par[:, 1], bounds[lo, hi] ** 2, rtol=1e-12)


@pytest.mark.gpu
def test_embedding_on_rdkits_tables_stays_within_rdkits_bounds_and_matches_its_yield():
    import torch  # noqa: F401

    from nvmolkit_amd.embedMolecules import FlatMolecule, FlatMoleculeSet, embed_flat

    z = np.load(need("etkdg_MMFF94_dative_first5.npz"))
    accept = json.loads(need("etkdg_acceptance.json").read_text())["ETKDGv3"]
    mols = []
    for m in range(int(z["n_molecules"])):
        key = f"m{m}"
        dg = [(z[f"{key}_dg_g{g}_idx"], z[f"{key}_dg_g{g}_par"]) for g in range(3)]
        etk = [(z[f"{key}_etk_g{g}_idx"], z[f"{key}_etk_g{g}_par"]) for g in range(6)] if f"{key}_etk_g0_idx" in z.files else None
        checks = [(int(k), tuple(int(x) for x in i), tuple(float(x) for x in p))
                  for k, i, p in zip(z[f"{key}_check_kind"], z[f"{key}_check_idx"], z[f"{key}_check_par"])]
        mols.append(FlatMolecule(int(z[f"{key}_n_atoms"]), dg, etk, checks, int(z[f"{key}_num_impropers"])))
    res = embed_flat(FlatMoleculeSet(mols), confs_per_molecule=5, max_iterations=10, seed=42)
    for m, mol in enumerate(mols):
        assert res.conf_counts[m] >= accept[m]["conformers"] - 1, (m, res.conf_counts[m], accept[m]["conformers"])
        bounds = z[f"m{m}_bounds"]
        for xyz in res.conformers(m).cpu().numpy():
            d = np.linalg.norm(xyz[:, None, :] - xyz[None, :, :], axis=2)
            iu = np.triu_indices(len(xyz), 1)
            over = np.maximum(d[iu] - bounds[iu], 0.0) / bounds[iu]
            under = np.maximum(bounds.T[iu] - d[iu], 0.0) / np.maximum(bounds.T[iu], 1e-9)
            assert np.percentile(np.maximum(over, under), 99) < 0.15


def test_oracle_fingerprints_equal_rdkit():
    assert check_morgan_fixture(need("morgan_chembl_1k.npz"), oracle_fingerprints) > 900


@pytest.mark.gpu
def test_kernel_fingerprints_equal_rdkit():
    from nvmolkit_amd.fingerprints import MorganFingerprintGenerator

    def kernel(s, ids, radius, bits):
        return MorganFingerprintGenerator(radius=radius, fpSize=bits).GetFingerprintsFromSmiles(s, on_error="zero").torch().cpu().numpy().view(np.uint32)[ids]

    assert check_morgan_fixture(need("morgan_chembl_1k.npz"), kernel) > 900


def test_self_matches_equal_rdkit():
    assert check_self_matches_fixture(need("self_matches.json")) > 20


def test_recorded_rdkit_rates_are_well_formed():
    conv = json.loads(need("mmff_convergence.json").read_text())
    assert 0.0 <= conv["mmff_converged_fraction_at_200_iterations"] <= 1.0 and conv["conformers"] > 1000
    base = json.loads(need("baseline_b.json").read_text())
    assert base["etkdg10_mmff200_threads_all"]["molecules_per_second"] > 0.0 and base["bulk_tanimoto_pairs_per_second_one_thread"] > 0.0


# ---- the same checkers on self-made fixtures: they run today, everywhere ---------------------------------------------------
def _self_made_force_field_fixture(tmp_path, kind, terms):
    rng = np.random.default_rng(31)
    store = {"n_molecules": np.int32(3)}
    for m, n_atoms in enumerate((9, 17, 30)):
        pos, groups = synthetic.random_ff_system(kind, n_atoms, rng)
        key = f"m{m}"
        store[f"{key}_pos"] = pos[:, :3].copy()
        for g, (idx, par) in enumerate(groups):
            store[f"{key}_g{g}_idx"], store[f"{key}_g{g}_par"] = np.asarray(idx, dtype=np.int32), np.asarray(par, dtype=np.float64)
        for k, term in terms:
            sub = only([(np.asarray(i), np.asarray(p)) for i, p in groups], k)
            flat = pos[:, :3].reshape(-1)
            store[f"{key}_energy_{term}"] = np.float64(off.system_energy(kind, flat.reshape(-1, 3), sub))
            store[f"{key}_grad_{term}"] = off.system_gradient(kind, flat.reshape(-1, 3), sub).reshape(-1)
    path = tmp_path / "self_made.npz"
    np.savez_compressed(path, **store)
    return path


@pytest.mark.parametrize("kind,terms", [(MMFF, MMFF_CASES), (UFF, [(None, "All")])])
def test_the_force_field_checker_on_a_self_made_fixture(tmp_path, kind, terms):
    """Format and code path of the RDKit fixtures with oracle/ff.py (independent numpy restatement, central differences) as the
    source of truth: the C oracle passes the reference's tolerances against it."""
    path = _self_made_force_field_fixture(tmp_path, kind, terms)
    assert check_force_field_fixture(path, kind, cpu_eval, terms) == 3


def test_the_self_match_checker_on_a_self_made_fixture(tmp_path):
    from oracle import smiles as osm

    rows = []
    for smi in ("c1ccccc1", "CC(C)(C)C", "CC(=O)[O-]", "OC(=O)CC(=O)O", "NC(=N)N", "CCO"):
        atoms, bonds = osm.molecule(smi)
        plain, sym = osm.self_matches(atoms, bonds, False), osm.self_matches(atoms, bonds, True)
        rows.append({"smiles": smi, "atoms": len(atoms), "matches": len(plain), "matches_terminal_groups_symmetric": len(sym),
                     "plain": [list(m) for m in plain], "symmetric": [list(m) for m in sym]})
    path = tmp_path / "self_matches.json"
    path.write_text(json.dumps(rows))
    assert check_self_matches_fixture(path) == 6


def test_the_generator_script_is_importable_and_says_what_it_needs():
    import subprocess
    import sys

    script = Path(__file__).parent / "golden" / "make_rdkit_fixtures.py"
    run = subprocess.run([sys.executable, str(script), "--out", "/tmp/nvmk_rdkit_fixture_probe"], capture_output=True, text=True, timeout=300)
    try:
        import rdkit  # noqa: F401
    except ImportError:
        assert run.returncode == 2 and "RDKit is not installed" in run.stderr
    else:
        assert run.returncode in (0, 1), run.stderr[-2000:]
