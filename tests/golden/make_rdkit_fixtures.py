#!/usr/bin/env python3
"""The "RDKit day" kit: ONE run of this script on a host where ``import rdkit`` works turns every "parity unpinned vs RDKit" row
of DESIGN.md section 2 into committed data that any later host — with or without RDKit, with or without a GPU — is tested
against (tests/test_rdkit_fixtures.py consumes whatever it finds under tests/golden/rdkit/ and skips what is not there).

It has NEVER run in the images this project is built and tested in (RDKit is in neither); it is written against RDKit's public
Python API only, every section stands alone (a failing section is reported and the others still write their files), and nothing
it writes is read by the product.

    python tests/golden/make_rdkit_fixtures.py [--out tests/golden/rdkit] [--only mmff,uff,etkdg,morgan,self_matches,convergence,baseline]

What it writes (all inputs are files already under tests/golden/, i.e. the reference's own test and benchmark data):

  versions.json              RDKit / numpy versions, host, date, the sections that ran
  mmff_<file>.npz            per molecule of MMFF94_dative_first5.sdf, MMFF94_dative_every4th.sdf, larger_molecules.sdf: coordinates,
                             the 7 flattened term groups made by the PRODUCT's flattener (mmffOptimization.flatten_mmff_from_rdkit),
                             and RDKit's own energy + gradient of every term kind alone (MMFFMolProperties.SetMMFF*Term) and of
                             the whole field — the reference's per-term checks (tests/test_mmff.cu:795-1016, tolerances :51-56)
  mmff_minimised.json        RDKit's minimised MMFF energies from the perturbed starts the reference pins
                             (tests/test_mmff.cu:1521-1608) and after MMFFOptimizeMoleculeConfs(maxIters=200)
  uff_<file>.npz             the same for UFF (whole field only: RDKit exposes no per-term switch for UFF)
  etkdg_<file>.npz           per molecule: RDKit's smoothed bounds matrix (rdDistGeom.GetMoleculeBoundsMatrix), and the DG / ETK term
                             tables, stereo checks and improper count of the product's flattener (_rdkit_embed.flatten_etkdg_from_rdkit)
                             — tests/test_flattened_builder.cu:188-189 checks the first distance term's bounds
  etkdg_acceptance.json      how many conformers RDKit's own EmbedMultipleConfs makes per molecule for the seven ETKDG variants
                             (nvmolkit/tests/test_embed_molecules.py:115-330) with the seeds used
  morgan_chembl_1k.npz       RDKit's Morgan fingerprints (radius 0 / 2 / 3; 512 / 2048 / 1024 bits) of chembl_1k.smi in the packed
                             layout (bit j = bit j % 32 of word j // 32) — tests/test_morgan_fingerprint.cpp:115-164
  self_matches.json          SubstructMatch(mol, mol, uniquify=False, maxMatches=1000) counts on the hydrogen-free molecule for a
                             set of symmetric molecules and conjugated terminal groups, plain and with the terminal groups made
                             symmetric the way MolAlign::details::symmetrizeTerminalAtoms does (emulated with query atoms / bonds:
                             that function has no Python binding) — rdkit_extensions/conformer_pruning.cpp:24-60
  mmff_convergence.json      fraction of ETKDG conformers (first 300 molecules of chembl_10k.smi, 10 each) whose
                             MMFFOptimizeMoleculeConfs(maxIters=200) reports convergence — the benchmark's maxIters
                             (benchmarks/ff_optimize_bench.py); DESIGN 4.4 reports 7.5 % / 22 % with generic parameters
  baseline_b.json            BASELINE.md's baseline B on THIS host: RDKit ETKDG(10) + MMFF94(200) molecules/s on chembl_10k.smi
                             (numThreads = all and 1), BulkTanimotoSimilarity pairs/s on its 2048-bit Morgan fingerprints
"""

from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time
import traceback
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent
ROOT = GOLDEN.parents[1]
sys.path.insert(0, str(ROOT))

MMFF_TERMS = ["Bond", "Angle", "StretchBend", "Oop", "Torsion", "VdW", "Ele"]  # = term group g of the MMFF layout
SD_FILES = ["MMFF94_dative_first5.sdf", "MMFF94_dative_every4th.sdf", "MMFF94_hypervalent_every4th.sdf", "larger_molecules.sdf"]
SYMMETRIC = ["c1ccccc1", "Cc1ccccc1", "C1CCCCC1", "CC(C)(C)C", "Cc1ccc(C)cc1", "c1ccc(cc1)-c1ccccc1", "CCO", "CC", "C1CC1",
             "FC(F)(F)c1ccccc1", "C1CCC2CCCCC2C1", "C[N+](C)(C)C", "O=C=O", "N#N",
             # conjugated terminal groups (symmetrizeConjugatedTerminalGroupsForPruning)
             "CC(=O)[O-]", "CC(=O)O", "C[N+](=O)[O-]", "NC(=N)c1ccccc1", "CC(=O)OC", "CC(=O)N", "OC(=O)CC(=O)O", "NC(=N)N",
             "NC(=O)CC(=N)O", "NC(=O)CC(N)O", "CS(=O)(=O)N", "CS(=O)(=O)[O-]", "OP(=O)(O)O", "[O-]c1ccc(cc1)C(=O)[O-]",
             "NC(=[NH2+])c1ccccc1", "O=C(O)c1ccc(O)cc1", "CC(=O)Oc1ccccc1C(=O)O"]


def sd_molecules(name):
    from rdkit import Chem

    return [m for m in Chem.SDMolSupplier(str(GOLDEN / name), removeHs=False, sanitize=True) if m is not None]


def smiles_of(name, limit=None):
    rows = [line.split()[0] for line in (GOLDEN / name).read_text().splitlines() if line.strip() and not line.startswith("#")]
    return rows[:limit] if limit else rows


def put_groups(store: dict, prefix: str, groups) -> None:
    for g, (idx, par) in enumerate(groups):
        store[f"{prefix}_g{g}_idx"] = np.asarray(idx, dtype=np.int32)
        store[f"{prefix}_g{g}_par"] = np.asarray(par, dtype=np.float64)


def section_mmff(out: Path) -> dict:
    from rdkit.Chem import rdForceFieldHelpers as ffh

    from nvmolkit_amd import mmffOptimization

    info = {}
    for name in SD_FILES:
        store, kept = {}, 0
        for m, mol in enumerate(sd_molecules(name)):
            if not ffh.MMFFHasAllMoleculeParams(mol):
                continue
            props = ffh.MMFFGetMoleculeProperties(mol)
            key = f"m{kept}"
            xyz = np.asarray(mol.GetConformer().GetPositions(), dtype=np.float64)
            store[f"{key}_record"] = np.int32(m)
            store[f"{key}_pos"] = xyz
            put_groups(store, key, mmffOptimization.flatten_mmff_from_rdkit(mol, props))
            for term in MMFF_TERMS + [None]:
                p = ffh.MMFFGetMoleculeProperties(mol)
                if term is not None:
                    for t in MMFF_TERMS:
                        getattr(p, f"SetMMFF{t}Term")(t == term)
                ff = ffh.MMFFGetMoleculeForceField(mol, p)
                tag = term or "All"
                store[f"{key}_energy_{tag}"] = np.float64(ff.CalcEnergy())
                store[f"{key}_grad_{tag}"] = np.asarray(ff.CalcGrad(), dtype=np.float64)
            kept += 1
        store["n_molecules"] = np.int32(kept)
        np.savez_compressed(out / f"mmff_{Path(name).stem}.npz", **store)
        info[name] = kept
    # minimised energies: RDKit's own minimiser from the reference's perturbed starts and from the file's coordinates
    mins = {}
    for name in ("MMFF94_dative_first5.sdf", "larger_molecules.sdf"):
        rows = []
        for m, mol in enumerate(sd_molecules(name)):
            if not ffh.MMFFHasAllMoleculeParams(mol):
                continue
            from rdkit import Chem

            work = Chem.Mol(mol)
            res = ffh.MMFFOptimizeMoleculeConfs(work, numThreads=1, maxIters=200)
            long_run = Chem.Mol(mol)
            res_long = ffh.MMFFOptimizeMoleculeConfs(long_run, numThreads=1, maxIters=10000)
            rows.append({"record": m, "atoms": mol.GetNumAtoms(), "not_converged_200": int(res[0][0]), "energy_200": float(res[0][1]),
                         "not_converged_10000": int(res_long[0][0]), "energy_10000": float(res_long[0][1]),
                         "pos_200": np.asarray(work.GetConformer().GetPositions()).tolist()})
        mins[name] = rows
    (out / "mmff_minimised.json").write_text(json.dumps(mins))
    return info


def section_uff(out: Path) -> dict:
    from rdkit.Chem import rdForceFieldHelpers as ffh

    from nvmolkit_amd import uffOptimization

    info = {}
    for name in SD_FILES:
        store, kept = {}, 0
        for m, mol in enumerate(sd_molecules(name)):
            if not ffh.UFFHasAllMoleculeParams(mol):
                continue
            try:
                groups = uffOptimization.flatten_uff_from_rdkit(mol)
            except Exception as exc:  # noqa: BLE001 - the flattener refuses a few centre types on purpose
                info.setdefault("refused", []).append([name, m, str(exc)[:120]])
                continue
            key = f"m{kept}"
            ff = ffh.UFFGetMoleculeForceField(mol)
            store[f"{key}_record"] = np.int32(m)
            store[f"{key}_pos"] = np.asarray(mol.GetConformer().GetPositions(), dtype=np.float64)
            put_groups(store, key, groups)
            store[f"{key}_energy_All"] = np.float64(ff.CalcEnergy())
            store[f"{key}_grad_All"] = np.asarray(ff.CalcGrad(), dtype=np.float64)
            kept += 1
        store["n_molecules"] = np.int32(kept)
        np.savez_compressed(out / f"uff_{Path(name).stem}.npz", **store)
        info[name] = kept
    return info


def etkdg_variants():
    from rdkit.Chem import rdDistGeom

    return {"ETKDGv3": rdDistGeom.ETKDGv3(), "ETKDGv2": rdDistGeom.ETKDGv2(), "ETKDG": rdDistGeom.ETKDG(), "KDG": rdDistGeom.KDG(),
            "ETDG": rdDistGeom.ETDG(), "srETKDGv3": rdDistGeom.srETKDGv3(), "DG": rdDistGeom.EmbedParameters()}


def section_etkdg(out: Path) -> dict:
    from rdkit import Chem
    from rdkit.Chem import rdDistGeom

    from nvmolkit_amd import _rdkit_embed

    info = {}
    for name in ("MMFF94_dative_first5.sdf", "MMFF94_dative_every4th.sdf"):
        store, kept = {}, 0
        for m, mol in enumerate(sd_molecules(name)):
            params = rdDistGeom.ETKDGv3()
            params.useRandomCoords = True
            try:
                flat = _rdkit_embed.flatten_etkdg_from_rdkit(mol, params)
            except Exception as exc:  # noqa: BLE001
                info.setdefault("refused", []).append([name, m, str(exc)[:120]])
                continue
            key = f"m{kept}"
            store[f"{key}_record"] = np.int32(m)
            store[f"{key}_bounds"] = np.asarray(rdDistGeom.GetMoleculeBoundsMatrix(mol), dtype=np.float64)
            store[f"{key}_n_atoms"] = np.int32(flat["n_atoms"])
            store[f"{key}_num_impropers"] = np.int32(flat["num_impropers"])
            put_groups(store, f"{key}_dg", flat["dg"])
            if flat["etk"] is not None:
                put_groups(store, f"{key}_etk", flat["etk"])
            checks = flat["checks"]
            store[f"{key}_check_kind"] = np.array([c[0] for c in checks], dtype=np.int32)
            store[f"{key}_check_idx"] = np.array([list(c[1]) + [0] * (5 - len(c[1])) for c in checks], dtype=np.int32).reshape(-1, 5)
            store[f"{key}_check_par"] = np.array([list(c[2]) + [0.0] * (2 - len(c[2])) for c in checks], dtype=np.float64).reshape(-1, 2)
            kept += 1
        store["n_molecules"] = np.int32(kept)
        np.savez_compressed(out / f"etkdg_{Path(name).stem}.npz", **store)
        info[name] = kept
    accept = {}
    mols = sd_molecules("MMFF94_dative_first5.sdf")
    for vname, params in etkdg_variants().items():
        params.useRandomCoords = True
        params.randomSeed = 42
        rows = []
        for mol in mols:
            work = Chem.Mol(mol)
            work.RemoveAllConformers()
            ids = list(rdDistGeom.EmbedMultipleConfs(work, 5, params))
            rows.append({"atoms": work.GetNumAtoms(), "conformers": len(ids),
                         "coords": [np.asarray(work.GetConformer(i).GetPositions()).round(4).tolist() for i in ids]})
        accept[vname] = rows
    (out / "etkdg_acceptance.json").write_text(json.dumps(accept))
    return info


def section_morgan(out: Path) -> dict:
    from rdkit import Chem
    from rdkit.Chem import rdFingerprintGenerator

    smiles = smiles_of("chembl_1k.smi")
    store = {"parsed": np.zeros(len(smiles), dtype=np.uint8)}
    for radius, bits in ((2, 2048), (3, 1024), (0, 512)):
        gen = rdFingerprintGenerator.GetMorganGenerator(radius=radius, fpSize=bits)
        fps = np.zeros((len(smiles), bits // 32), dtype=np.uint32)
        for i, smi in enumerate(smiles):
            mol = Chem.MolFromSmiles(smi)
            if mol is None:
                continue
            store["parsed"][i] = 1
            for bit in gen.GetFingerprint(mol).GetOnBits():
                fps[i, bit // 32] |= np.uint32(1) << np.uint32(bit % 32)
        store[f"r{radius}_b{bits}"] = fps
    np.savez_compressed(out / "morgan_chembl_1k.npz", **store)
    return {"molecules": len(smiles), "parsed": int(store["parsed"].sum())}


def symmetrized_probe(mol):
    """Python emulation of MolAlign::details::symmetrizeTerminalAtoms (no Python binding): terminal N / O of X-[*]=X / X=[*]-X
    become element-only query atoms and their bonds single-or-double query bonds, in a copy that serves as the PROBE."""
    from rdkit import Chem

    pat = Chem.MolFromSmarts("[O,N;D1;$([O,N;D1]-[*]=[O,N;D1]),$([O,N;D1]=[*]-[O,N;D1])]~[*]")
    probe = Chem.RWMol(mol)
    for term, centre in mol.GetSubstructMatches(pat):
        bond = probe.GetBondBetweenAtoms(term, centre)
        probe.ReplaceBond(bond.GetIdx(), Chem.BondFromSmarts("-,="))
        probe.ReplaceAtom(term, Chem.AtomFromSmarts(f"[#{mol.GetAtomWithIdx(term).GetAtomicNum()}]"))
    return probe


def section_self_matches(out: Path) -> dict:
    from rdkit import Chem

    rows = []
    for smi in SYMMETRIC:
        mol = Chem.RemoveHs(Chem.MolFromSmiles(smi), sanitize=False)
        plain = mol.GetSubstructMatches(mol, uniquify=False, maxMatches=1000)
        sym = mol.GetSubstructMatches(symmetrized_probe(mol), uniquify=False, maxMatches=1000)
        rows.append({"smiles": smi, "atoms": mol.GetNumAtoms(), "matches": len(plain), "matches_terminal_groups_symmetric": len(sym),
                     "plain": [list(map(int, m)) for m in plain], "symmetric": [list(map(int, m)) for m in sym]})
    (out / "self_matches.json").write_text(json.dumps(rows))
    return {"molecules": len(rows)}


def embed_and_optimise(smiles, confs, threads, max_iters=200):
    from rdkit import Chem
    from rdkit.Chem import rdDistGeom
    from rdkit.Chem import rdForceFieldHelpers as ffh

    params = rdDistGeom.ETKDGv3()
    params.useRandomCoords = True
    params.randomSeed = 42
    params.numThreads = threads
    t_embed = t_mmff = 0.0
    n_mols = n_confs = converged = 0
    for smi in smiles:
        mol = Chem.MolFromSmiles(smi)
        if mol is None:
            continue
        mol = Chem.AddHs(mol)
        t0 = time.perf_counter()
        ids = rdDistGeom.EmbedMultipleConfs(mol, confs, params)
        t_embed += time.perf_counter() - t0
        if len(ids) == 0 or not ffh.MMFFHasAllMoleculeParams(mol):
            continue
        t0 = time.perf_counter()
        res = ffh.MMFFOptimizeMoleculeConfs(mol, numThreads=threads, maxIters=max_iters)
        t_mmff += time.perf_counter() - t0
        n_mols += 1
        n_confs += len(res)
        converged += sum(1 for nc, _ in res if nc == 0)
    return {"molecules": n_mols, "conformers": n_confs, "mmff_converged": converged, "etkdg_seconds": t_embed, "mmff_seconds": t_mmff,
            "threads": threads}


def section_convergence(out: Path) -> dict:
    r = embed_and_optimise(smiles_of("chembl_10k.smi", 300), 10, 0)
    r["mmff_converged_fraction_at_200_iterations"] = r["mmff_converged"] / max(r["conformers"], 1)
    (out / "mmff_convergence.json").write_text(json.dumps(r))
    return r


def section_baseline(out: Path) -> dict:
    from rdkit import Chem, DataStructs
    from rdkit.Chem import rdFingerprintGenerator

    smiles = smiles_of("chembl_10k.smi")
    res = {"host": platform.node(), "cpu_count": os.cpu_count()}
    for threads, n in ((0, 2000), (1, 100)):
        r = embed_and_optimise(smiles[:n], 10, threads)
        r["molecules_per_second"] = r["molecules"] / max(r["etkdg_seconds"] + r["mmff_seconds"], 1e-9)
        res[f"etkdg10_mmff200_threads_{'all' if threads == 0 else threads}"] = r
    gen = rdFingerprintGenerator.GetMorganGenerator(radius=2, fpSize=2048)
    t0 = time.perf_counter()
    fps = [gen.GetFingerprint(m) for m in (Chem.MolFromSmiles(s) for s in smiles) if m is not None]
    res["morgan_fingerprints_per_second_one_thread"] = len(fps) / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    rows = 500
    for i in range(rows):
        DataStructs.BulkTanimotoSimilarity(fps[i], fps)
    res["bulk_tanimoto_pairs_per_second_one_thread"] = rows * len(fps) / (time.perf_counter() - t0)
    (out / "baseline_b.json").write_text(json.dumps(res))
    return res


SECTIONS = {"mmff": section_mmff, "uff": section_uff, "etkdg": section_etkdg, "morgan": section_morgan, "self_matches": section_self_matches,
            "convergence": section_convergence, "baseline": section_baseline}


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=str(GOLDEN / "rdkit"))
    ap.add_argument("--only", default=",".join(SECTIONS))
    args = ap.parse_args()
    try:
        import rdkit
    except ImportError:
        print("RDKit is not installed on this host: nothing written (this script is for the first host that has it).", file=sys.stderr)
        return 2
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    done, failed = {}, {}
    for name in [s for s in args.only.split(",") if s]:
        if name not in SECTIONS:
            print(f"unknown section {name!r}; known: {', '.join(SECTIONS)}", file=sys.stderr)
            return 2
        t0 = time.perf_counter()
        try:
            done[name] = {"seconds": None, "result": SECTIONS[name](out)}
            done[name]["seconds"] = time.perf_counter() - t0
            print(f"[{name}] ok in {done[name]['seconds']:.1f} s")
        except Exception:  # noqa: BLE001 - every section stands alone
            failed[name] = traceback.format_exc()
            print(f"[{name}] FAILED\n{failed[name]}", file=sys.stderr)
    (out / "versions.json").write_text(json.dumps({"rdkit": rdkit.__version__, "numpy": np.__version__, "python": platform.python_version(),
                                                   "host": platform.node(), "date": time.strftime("%Y-%m-%d"), "sections": done,
                                                   "failed": failed}, indent=1, default=str))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
