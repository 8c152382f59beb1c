"""The RDKit -> flattened-ETKDG adapter (nvmolkit_amd/_rdkit_embed.py; reference: src/embedder_utils.cpp:102-210,617-712,
rdkit_extensions/dist_geom_flattened_builder.cpp:56-540) on duck-typed molecules with a fake ``rdkit`` module: chemistry
lists, term tables and check lists, then (GPU) ``EmbedMolecules`` end to end in both output modes."""

import numpy as np
import pytest

from nvmolkit_amd import _native, _rdkit_embed, synthetic
from tests import fake_rdkit as fr


def methyl_ethene_like():
    """C0(=C1)(H2)(H3) ... a hand-made molecule: C0=C1 double bond, C1-C2 single, C2 with 4 neighbours in two rings."""
    #  atoms: 0 C(sp2) 1 C(sp2) 2 H 3 H 4 C(sp3, ring junction) 5 H 6..9 ring atoms
    z = [6, 6, 1, 1, 6, 1, 6, 6, 6, 6]
    bonds = [fr.FakeBond(0, 1, "DOUBLE", "STEREOE", (2, 4)), (0, 2), (0, 3), (1, 4), (1, 5), (4, 6), (4, 7), (4, 8), (6, 7), (8, 9), (9, 6)]
    rings = [(4, 6, 7), (4, 8, 9, 6)]
    m = fr.FakeMol(z, bonds, rings)
    m.atoms[0].hyb = m.atoms[1].hyb = "SP2"
    return m


def test_topology_chiral_sets_and_double_bonds():
    m = methyl_ethene_like()
    nbrs, bonds, btype, angles = _rdkit_embed.topology(m)
    assert nbrs[4] == [1, 6, 7, 8] and btype[(1, 0)] == "DOUBLE"
    assert (2, 0, 3, 0) in angles and len([a for a in angles if a[1] == 4]) == 6
    chiral, tetra = _rdkit_embed.chiral_sets(m, nbrs)
    # atom 4: degree-4 carbon in two rings but one of them is a 3-ring -> skipped (findChiralSets :191-196)
    assert chiral == [] and tetra == []
    m.ring_info = fr.FakeRingInfo([{4, 6, 7, 9}, {4, 8, 9, 6}])
    chiral, tetra = _rdkit_embed.chiral_sets(m, nbrs)
    assert tetra == [(4, 1, 6, 7, 8, 0.0, 0.0, 1)]            # two rings smaller than 5 -> fused-small-rings flag
    m.atoms[4].tag = "CHI_TETRAHEDRAL_CW"
    chiral, tetra = _rdkit_embed.chiral_sets(m, nbrs)
    assert chiral == [(4, 1, 6, 7, 8, -100.0, -5.0, 0)] and tetra == []
    ends, stereo = _rdkit_embed.double_bonds(m, nbrs, btype)
    assert set(ends) == {(2, 0, 1), (3, 0, 1), (4, 1, 0), (5, 1, 0)}
    assert stereo == [((2, 0, 1, 4), 1)]
    imps = _rdkit_embed.improper_atoms(m, nbrs, btype)
    assert [i[1] for i in imps] == [0, 1] and imps[0][4:] == (6, False)


def test_inversion_coefficients():
    assert _rdkit_embed.inversion_coefficients(6, False) == (2.0, 1.0, -1.0, 0.0)
    assert _rdkit_embed.inversion_coefficients(6, True)[0] == pytest.approx(50.0 / 3.0)
    k, c0, c1, c2 = _rdkit_embed.inversion_coefficients(15, False)
    w = np.deg2rad(84.4339)
    assert c2 == 1.0 and c1 == pytest.approx(-4 * np.cos(w)) and c0 == pytest.approx(-(c1 * np.cos(w) + np.cos(2 * w)))
    assert k == pytest.approx(22.0 / (c0 + c1 + c2) / 3.0)


def test_flattened_tables_from_a_druglike_molecule():
    rng = np.random.default_rng(4)
    src = synthetic.druglike_molecule(rng, 30)
    mol = fr.FakeMol.from_druglike(src)
    with fr.install():
        flat = _rdkit_embed.flatten_etkdg_from_rdkit(mol, fr.FakeEmbedParameters())
    n = 30
    assert flat["n_atoms"] == n
    pairs, par = flat["dg"][0]
    assert len(pairs) == n * (n - 1) // 2 and np.all(pairs[:, 0] > pairs[:, 1])
    # squared bounds of the bounds matrix, pair by pair
    spairs, lb, ub = src["bounds"]
    want = {(int(j), int(i)): (l * l, u * u) for (i, j), l, u in zip(spairs, lb, ub)}
    for (i, j), (l2, u2, w) in zip(pairs, par):
        assert want[(int(i), int(j))] == pytest.approx((l2, u2)) and w == 1.0
    etk = flat["etk"]
    assert len(etk) == 6 and [g[1].shape[1] for g in etk] == [12, 4, 4, 4, 2, 4]
    assert len(etk[0][0]) == len(src["embed"]["etk"][0][0])                      # one term per experimental torsion
    assert len(etk[2][0]) == len(src["bonds"]) and np.all(etk[2][1][:, :2] == [-0.01, 0.01]) and np.all(etk[2][1][:, 3] == 0)
    n_planar = flat["num_impropers"]
    assert len(etk[1][0]) == 3 * n_planar and np.all(etk[1][1][:, 3] == 2.0 * 10.0)    # C sp2: k = 6 / 3 x scaling 10
    pinned = etk[3][1][:, 3] == 1.0                                              # 1-3 terms across an improper centre
    centres = {int(t[1]) for t in etk[1][0]}
    nbrs, _, _, angles = _rdkit_embed.topology(mol)
    assert int(pinned.sum()) == sum(1 for a in angles if a[1] in centres)
    # every atom pair is restrained exactly once by a 1-2, 1-3, torsion-end or long-range term
    seen = set()
    for g in (2, 3, 5):
        for i, j in etk[g][0]:
            key = (min(int(i), int(j)), max(int(i), int(j)))
            assert key not in seen
            seen.add(key)
    tor_ends = {(min(int(t[0]), int(t[3])), max(int(t[0]), int(t[3]))) for t in etk[0][0]}
    assert len(seen | tor_ends) == n * (n - 1) // 2
    kinds = [c[0] for c in flat["checks"]]
    assert kinds.count(_native.CHECK_DOUBLE_BOND_GEOMETRY) == 0                   # the fake molecule has single bonds only


def test_bad_parameters_raise():
    mol = fr.FakeMol([6, 6], [(0, 1)], bounds=np.array([[0, 1.6], [1.4, 0]]))
    with fr.install():
        with pytest.raises(ValueError, match="ETversion"):
            _rdkit_embed.flatten_etkdg_from_rdkit(mol, fr.FakeEmbedParameters(ETversion=3))
        with pytest.raises(ValueError, match="no atoms"):
            _rdkit_embed.flatten_etkdg_from_rdkit(fr.FakeMol([], []), fr.FakeEmbedParameters())
        mol.bounds = None
        with pytest.raises(ValueError, match="triangle bounds smooth"):
            _rdkit_embed.flatten_etkdg_from_rdkit(mol, fr.FakeEmbedParameters())


def test_embed_molecules_argument_checks():
    from nvmolkit_amd.embedMolecules import EmbedMolecules
    from nvmolkit_amd.types import CoordinateOutput

    p = fr.FakeEmbedParameters()
    assert EmbedMolecules([], p) is None
    with pytest.raises(ValueError, match="at least one molecule"):
        EmbedMolecules([], p, output=CoordinateOutput.DEVICE)
    with pytest.raises(ValueError, match="index 1 is None"):
        EmbedMolecules([fr.FakeMol([6], []), None], p)
    with pytest.raises(ValueError, match="useRandomCoords"):
        EmbedMolecules([fr.FakeMol([6], [])], fr.FakeEmbedParameters(useRandomCoords=False))
    with pytest.raises(ValueError, match="confsPerMolecule"):
        EmbedMolecules([fr.FakeMol([6], [])], p, confsPerMolecule=0)
    with pytest.raises(ValueError, match="pruning"):
        EmbedMolecules([fr.FakeMol([6], [])], fr.FakeEmbedParameters(pruneRmsThresh=0.5), output=CoordinateOutput.DEVICE)


@pytest.mark.gpu
def test_embed_molecules_end_to_end_on_duck_typed_molecules():
    """EmbedMolecules(mols, params, ...) in both output modes (nvmolkit/embedMolecules.py:55-158) — no NotImplementedError."""
    import torch

    from nvmolkit_amd.embedMolecules import EmbedMolecules
    from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

    rng = np.random.default_rng(12)
    src = [synthetic.druglike_molecule(rng, n) for n in (14, 22, 31, 18)]
    mols = [fr.FakeMol.from_druglike(m) for m in src]
    params = fr.FakeEmbedParameters(randomSeed=7)
    with fr.install():
        assert EmbedMolecules(mols, params, confsPerMolecule=3, maxIterations=10,
                              hardwareOptions=HardwareOptions(batchSize=5, gpuIds=[0])) is None
        for mol, m in zip(mols, src):
            assert mol.GetNumConformers() == 3
            pairs, lb, ub = m["bounds"]
            for conf in mol.GetConformers():
                d = np.linalg.norm(conf.xyz[pairs[:, 0]] - conf.xyz[pairs[:, 1]], axis=1)
                assert np.max(np.maximum(np.maximum(lb - d, d - ub), 0.0) / ub) < 0.2
        first = [c.xyz.copy() for c in mols[0].GetConformers()]
        dev = EmbedMolecules(mols, params, confsPerMolecule=3, maxIterations=10, output=CoordinateOutput.DEVICE, targetGpu=0)
        assert isinstance(dev, Device3DResult) and dev.num_conformers == 12 and dev.n_mols == 4
        assert dev.mol_indices.torch().tolist() == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3
        assert [c.xyz.tolist() for c in mols[0].GetConformers()] == [f.tolist() for f in first]   # DEVICE leaves them alone
        per = dev.per_molecule()
        assert per[2][0].shape == (31, 3) and bool(torch.isfinite(dev.values.torch()).all())
        with pytest.raises(ValueError, match="targetGpu"):
            EmbedMolecules(mols, params, output=CoordinateOutput.DEVICE, hardwareOptions=HardwareOptions(gpuIds=[0]), targetGpu=5)
        pruned = fr.FakeEmbedParameters(randomSeed=7, pruneRmsThresh=50.0)         # absurd threshold: one survivor each
        EmbedMolecules(mols, pruned, confsPerMolecule=3, maxIterations=10)
        assert [m.GetNumConformers() for m in mols] == [1, 1, 1, 1]


def test_morgan_invariants_adapter_matches_the_scalar_restatement():
    """fingerprints.morgan_invariants_from_rdkit (vectorised: M1, src/morgan_fingerprint_common.cpp:43-124) on duck-typed
    molecules == tests/util.flatten_molecules (atom-by-atom hashing through the C oracle's hash_range)."""
    from nvmolkit_amd.fingerprints import morgan_invariants_from_rdkit
    from tests import util

    graphs = util.random_molecule_batch(40, 64, seed=3, min_atoms=1) + [([], []), ([(6, 4, 0, False)], [])]
    mols = []
    for atoms, bonds in graphs:
        m = fr.FakeMol([a[0] for a in atoms], [fr.FakeBond(i, j, int(t)) for i, j, t in bonds],
                       rings=[{i for i, a in enumerate(atoms) if a[3]}] if any(a[3] for a in atoms) else ())
        for fa, (z, nh, q, ring) in zip(m.atoms, atoms):
            fa.n_h, fa.charge = nh, q
        mols.append(m)
    with fr.install():
        got = morgan_invariants_from_rdkit(mols, 64)
    want = util.flatten_molecules(graphs, 64)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and np.array_equal(g, w)
    with fr.install(), pytest.raises(ValueError, match="bucket"):
        morgan_invariants_from_rdkit(mols[:3], 4)


@pytest.mark.gpu
def test_get_fingerprints_from_duck_typed_molecules_buckets_and_async_staging():
    """MorganFingerprintGenerator.GetFingerprints (M1 + M4: invariants, buckets 32 / 64 / 128 / 256, staging and launches
    queued on one stream without host synchronisation) == the oracle on the same graphs, rows in input order."""
    import torch

    import oracle
    from nvmolkit_amd.fingerprints import MorganFingerprintGenerator
    from tests import util

    rng = np.random.default_rng(5)
    graphs = []
    for stride in (32, 64, 128, 256, 32, 64):
        graphs += util.random_molecule_batch(7, stride, seed=int(rng.integers(1 << 30)), min_atoms=max(1, stride // 2 - 4))
    order = rng.permutation(len(graphs))
    graphs = [graphs[i] for i in order]
    mols = []
    for atoms, bonds in graphs:
        m = fr.FakeMol([a[0] for a in atoms], [fr.FakeBond(i, j, int(t)) for i, j, t in bonds],
                       rings=[{i for i, a in enumerate(atoms) if a[3]}] if any(a[3] for a in atoms) else ())
        for fa, (z, nh, q, ring) in zip(m.atoms, atoms):
            fa.n_h, fa.charge = nh, q
        mols.append(m)
    side = torch.cuda.Stream()
    with fr.install():
        res = MorganFingerprintGenerator(2, 2048).GetFingerprints(mols, stream=side)
    side.synchronize()
    got = res.torch().cpu().numpy().view(np.uint32)
    for i, g in enumerate(graphs):
        size = max(len(g[0]), len(g[1]))
        stride = next(b for b in (32, 64, 128, 256) if size < b)
        flat = util.flatten_molecules([g], stride)
        assert np.array_equal(got[i], oracle.morgan_fingerprints(*flat, stride, 2, 2048)[0]), i
    with fr.install(), pytest.raises(ValueError):
        MorganFingerprintGenerator(2, 2048).GetFingerprints([mols[0], None])
