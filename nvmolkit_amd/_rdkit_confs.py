"""RDKit conformers <-> flattened conformer batches for the MMFF / UFF optimisers.

The Python counterpart of the reference's ``flattenConformers`` / ``writeBackResults``
(src/minimizer/bfgs_common.cpp:42-105) and of the batch loop of ``MMFFMinimizeMoleculesConfs`` /
``UFFMinimizeMoleculesConfs`` (src/minimizer/bfgs_mmff.cpp:139-328, bfgs_uff.cpp:36-255).  Needs RDKit molecules, so
it is exercised only where RDKit is installed; the force-field work it drives (``FlatForcefieldBatch.minimize`` with
per-molecule tables shared through ``system_mol``) is what the GPU tests cover.
"""

from __future__ import annotations

import numpy as np
import torch

from nvmolkit_amd.forcefield import FlatForcefieldBatch, stack_molecule_tables
from nvmolkit_amd.types import CoordinateOutput, Device3DResult, HardwareOptions

def run_per_gpu(worker, n_slots: int) -> None:
    """Run ``worker(slot)`` for every slot, slots > 0 on their own host threads; the first exception is re-raised."""
    if n_slots <= 1:
        worker(0)
        return
    import threading

    errors: list = []

    def guarded(slot: int) -> None:
        try:
            worker(slot)
        except BaseException as exc:  # noqa: BLE001  (carried to the caller's thread)
            errors.append(exc)

    threads = [threading.Thread(target=guarded, args=(slot,), name=f"nvmk-gpu-{slot}") for slot in range(1, n_slots)]
    for t in threads:
        t.start()
    guarded(0)
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


DEFAULT_BATCH = 4096  # conformers per launch when HardwareOptions.batchSize is -1 (the reference uses 500; see embedMolecules.AUTO_BATCH_SIZE)


def optimize_rdkit_conformers(kind: int, molecules, flatten, max_iters: int, grad_tol: float,
                              hardware_options: HardwareOptions | None, output: CoordinateOutput, target_gpu: int):
    """Minimise every conformer of every molecule.  ``flatten(mol_index, conf_id)`` returns the molecule's term groups
    and is called once per molecule, on its first conformer (bfgs_mmff.cpp:159,195-201)."""
    batch_size = hardware_options.batchSize if hardware_options and hardware_options.batchSize > 0 else DEFAULT_BATCH
    gpu_ids = list(hardware_options.gpuIds) if hardware_options and hardware_options.gpuIds else [torch.cuda.current_device()]
    device_out = output == CoordinateOutput.DEVICE
    if device_out:
        if target_gpu is None or target_gpu < 0:
            target_gpu = gpu_ids[0]
        if target_gpu not in gpu_ids:
            raise ValueError(f"targetGpu {target_gpu} is not in the configured set of execution GPUs; pass it via "
                             "hardwareOptions.gpuIds first.")
    systems = [(mi, ci, conf.GetId()) for mi, m in enumerate(molecules) for ci, conf in enumerate(m.GetConformers())]
    chunks = [systems[lo:lo + batch_size] for lo in range(0, len(systems), batch_size)]
    # term tables: once per molecule, on its first conformer, on the host thread (RDKit objects are not shared between threads)
    tables = {}
    for mi, _, cid in systems:
        if mi not in tables:
            tables[mi] = flatten(mi, cid)
    staged = []
    for chunk in chunks:
        local, atom_starts, pos, system_mol = {}, [0], [], []
        for mi, _, cid in chunk:
            m = molecules[mi]
            system_mol.append(local.setdefault(mi, len(local)))
            atom_starts.append(atom_starts[-1] + m.GetNumAtoms())
            pos.append(np.asarray(m.GetConformer(cid).GetPositions(), dtype=np.float64).reshape(-1))
        staged.append((chunk, np.array(atom_starts, dtype=np.int32), np.concatenate(pos) if pos else np.zeros(0),
                       np.array(system_mol, dtype=np.int32), stack_molecule_tables(kind, [tables[mi] for mi in local])))

    # Batches are dealt round-robin to the configured GPUs and every GPU works through its share on a host thread of its
    # own with its own stream (the reference: one OpenMP thread per (GPU, batch slot), src/minimizer/bfgs_mmff.cpp:139-164):
    # a blocking minimise on one GPU no longer holds up the others.
    done = [None] * len(staged)

    def gpu_worker(slot: int) -> None:
        device = torch.device("cuda", gpu_ids[slot])
        with torch.cuda.device(device):
            stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(stream):
                for b in range(slot, len(staged), len(gpu_ids)):
                    chunk, a_s, pos, sys_mol, groups = staged[b]
                    positions = torch.from_numpy(pos).to(device)
                    batch = FlatForcefieldBatch(kind, a_s, groups, device=device, system_mol=sys_mol)
                    energies, statuses, _ = batch.minimize(positions, max_iters=max_iters, grad_tol=grad_tol, scale_grads=True,
                                                           stream=stream)
                    done[b] = (chunk, a_s, positions, energies, statuses)
            stream.synchronize()

    run_per_gpu(gpu_worker, len(gpu_ids))
    results = [[] for _ in molecules]
    kept = [d for d in done if d is not None]
    if not device_out:
        for chunk, atom_starts, positions, energies, _ in kept:
            out, e = positions.cpu().numpy(), energies.cpu().numpy()
            for s, (mi, _, cid) in enumerate(chunk):
                conf = molecules[mi].GetConformer(cid)
                xyz = out[atom_starts[s] * 3:atom_starts[s + 1] * 3].reshape(-1, 3)
                if hasattr(conf, "SetPositions"):  # RDKit >= 2022.09
                    conf.SetPositions(np.ascontiguousarray(xyz))
                else:
                    from rdkit.Geometry import Point3D

                    for a, (x, y, z) in enumerate(xyz):
                        conf.SetAtomPosition(a, Point3D(float(x), float(y), float(z)))
                results[mi].append(float(e[s]))
    if not device_out:
        return results
    # consolidate on the target GPU in input order (detail::finalizeOnTarget, src/conformer/device_coord_collector.cpp)
    tgt = torch.device("cuda", target_gpu)
    values = torch.cat([k[2].to(tgt) for k in kept]).view(-1, 3) if kept else torch.zeros((0, 3), dtype=torch.float64, device=tgt)
    sizes = np.concatenate([np.diff(k[1]) for k in kept]) if kept else np.zeros(0, dtype=np.int64)
    atom_starts = np.zeros(len(sizes) + 1, dtype=np.int32)
    atom_starts[1:] = np.cumsum(sizes)
    mk = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(tgt)  # noqa: E731
    return Device3DResult(values, mk(atom_starts, np.int32), mk([s[0] for s in systems], np.int32),
                          mk([s[1] for s in systems], np.int32), target_gpu, len(molecules),
                          energies=torch.cat([k[3].to(tgt) for k in kept]) if kept else torch.zeros(0, dtype=torch.float64, device=tgt),
                          converged=torch.cat([(k[4] == 0).to(torch.int8).to(tgt) for k in kept]) if kept else torch.zeros(0, dtype=torch.int8, device=tgt))
