"""The multi-GPU entries of the C ABI (device set, peer copy, the all-gather of the reference block) and the
ETKDG stage-timing table, called through the library as a C or C++ host would (ctypes, no torch.distributed) — on the one-GPU
box as one rank, which exercises everything but the wire."""

import ctypes

import numpy as np
import pytest
import torch

from nvmolkit_amd import _native, synthetic
from nvmolkit_amd.embedMolecules import STAGE_NAMES, FlatMolecule, FlatMoleculeSet, embed_flat, format_stage_timings, stage_timings

pytestmark = pytest.mark.gpu


def test_device_set_and_peer_copy():
    lib = _native.lib()
    n = ctypes.c_int(0)
    _native.check(lib.nvmk_device_count(ctypes.byref(n)), "count")
    ids = np.arange(n.value, dtype=np.int32)
    _native.check(lib.nvmk_set_devices(ids.ctypes.data, len(ids)), "nvmk_set_devices")
    _native.check(lib.nvmk_set_devices(ids.ctypes.data, len(ids)), "nvmk_set_devices twice (idempotent)")
    got, k = np.zeros(16, dtype=np.int32), ctypes.c_int(0)
    _native.check(lib.nvmk_get_devices(got.ctypes.data, 16, ctypes.byref(k)), "nvmk_get_devices")
    assert got[:k.value].tolist() == ids.tolist()
    bad = np.array([n.value + 3], dtype=np.int32)
    assert lib.nvmk_set_devices(bad.ctypes.data, 1) == _native.ERR_INVALID_ARGUMENT and b"does not exist" in lib.nvmk_last_error()
    twice = np.array([0, 0], dtype=np.int32)
    assert lib.nvmk_set_devices(twice.ctypes.data, 2) == _native.ERR_INVALID_ARGUMENT
    _native.check(lib.nvmk_set_devices(None, 0), "nvmk_set_devices(all)")
    # same-device "peer" copy: ordered behind the SOURCE stream's work although it runs on the destination stream — the source
    # stream is still busy producing the buffer when the copy is asked for (300 dependent updates of 32 MB; nothing waits for them)
    src_stream, dst_stream = torch.cuda.Stream(), torch.cuda.Stream()
    src = torch.zeros(1 << 22, dtype=torch.float64, device="cuda")
    dst = torch.full_like(src, -1.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(src_stream):
        for _ in range(300):
            src.add_(1.0)
    _native.check(lib.nvmk_copy_peer_async(dst.data_ptr(), 0, dst_stream.cuda_stream, src.data_ptr(), 0, src_stream.cuda_stream, src.numel() * 8),
                  "nvmk_copy_peer_async")
    dst_stream.synchronize()
    assert float(dst.min()) == 300.0 and float(dst.max()) == 300.0
    # one stream on both sides: plain stream order, no event
    _native.check(lib.nvmk_copy_peer_async(dst.data_ptr(), 0, src_stream.cuda_stream, src.data_ptr(), 0, src_stream.cuda_stream, 4096), "same stream")
    src_stream.synchronize()


def test_all_gather_of_fingerprint_rows_as_one_rank():
    lib = _native.lib()
    uid = ctypes.create_string_buffer(128)
    _native.check(lib.nvmk_comm_unique_id(uid), "nvmk_comm_unique_id")
    comm = ctypes.c_void_p()
    _native.check(lib.nvmk_comm_init_rank(ctypes.byref(comm), 1, uid, 0), "nvmk_comm_init_rank")
    rows, words = 1000, 64
    send = torch.randint(-2**31, 2**31 - 1, (rows, words), dtype=torch.int32, device="cuda")
    recv = torch.zeros_like(send)
    stream = torch.cuda.current_stream()
    _native.check(lib.nvmk_allgather_rows(comm, send.data_ptr(), rows, words, recv.data_ptr(), stream.cuda_stream), "nvmk_allgather_rows")
    stream.synchronize()
    assert torch.equal(recv, send)
    assert lib.nvmk_allgather_rows(None, send.data_ptr(), rows, words, recv.data_ptr(), stream.cuda_stream) == _native.ERR_INVALID_ARGUMENT
    assert lib.nvmk_allgather_rows(comm, send.data_ptr(), -1, words, recv.data_ptr(), stream.cuda_stream) == _native.ERR_INVALID_ARGUMENT
    _native.check(lib.nvmk_comm_destroy(comm), "nvmk_comm_destroy")


def test_stage_timing_table_of_an_etkdg_call():
    lib_mols = synthetic.druglike_library(48, seed=2, mean_atoms=30, processes=1)
    molset = FlatMoleculeSet([FlatMolecule(**m["embed"]) for m in lib_mols])
    with _native.options(NVMK_ETKDG_TIMING="1"):
        res = embed_flat(molset, confs_per_molecule=3, max_iterations=10, seed=4, batch_size=64)
        rows = stage_timings()
    assert [r["stage"] for r in rows[:len(STAGE_NAMES)]] == [
        "Coordinate Generation", "First Minimization", "Tetrahedral Checks", "First Chirality Check", "Fourth Dimension Minimization",
        "ETK 3D Minimization", "Double bond geometry check", "Final Chirality Check", "Chirality Distance Matrix Check",
        "Final Chiral Center in Volume Check", "Double bond stereo check"]
    batches = rows[0]["calls"]
    assert batches >= 3 and all(r["calls"] == batches for r in rows[:len(STAGE_NAMES) + 1]) and rows[-1]["calls"] == 1
    assert all(0.0 <= r["min_ms"] <= r["max_ms"] <= r["total_ms"] + 1e-9 for r in rows)
    # the two minimisation stages dominate; the whole call covers the sum of its parts
    by = {r["stage"]: r["total_ms"] for r in rows}
    assert by["First Minimization"] > by["Tetrahedral Checks"] and by["ETK 3D Minimization"] > by["Double bond stereo check"]
    assert rows[-1]["total_ms"] >= 0.9 * sum(r["total_ms"] for r in rows[:-1])
    assert "First Minimization" in format_stage_timings(rows) and int(res.conf_counts.sum()) > 100
