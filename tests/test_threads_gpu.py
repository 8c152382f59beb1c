"""The C ABI promises that every entry point may be called concurrently from different host threads on different
streams (include/nvmolkit_amd.h; the reference runs one OpenMP thread per (GPU, batch slot)).  Four
threads hammer the similarity, neighbour-count, Morgan and BFGS entry points at once, each on its own stream, and every
result must equal the single-threaded one; the thread-local error slot must not leak between threads."""

import threading

import numpy as np
import pytest
import torch

import oracle
from nvmolkit_amd import _native
from nvmolkit_amd.clustering import fused_butina, update_neighbor_counts
from nvmolkit_amd.forcefield import MMFF, FlatForcefieldBatch
from nvmolkit_amd.similarity import crossTanimotoSimilarity
from tests import util

pytestmark = pytest.mark.gpu


def test_concurrent_calls_from_host_threads(native_lib):
    n_threads, rounds = 4, 6
    fps = [util.clustered_fingerprints(700 + 50 * t, 64, 20, seed=100 + t) for t in range(n_threads)]
    want_sim = [oracle.cross_similarity(f, f) for f in fps]
    want_cnt = [oracle.neighbor_counts(f, f, 0.6) for f in fps]
    want_but = [oracle.butina_fused(f, 0.4) for f in fps]
    rng = np.random.default_rng(3)
    ff_sys = [[util.random_ff_system(MMFF, n, rng) for n in (6, 9, 12)] for _ in range(n_threads)]
    errors, barrier = [], threading.Barrier(n_threads)

    def work(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            x = torch.from_numpy(fps[t].view(np.int32)).cuda()
            a_s, flat, groups = util.build_ff_batch_arrays(MMFF, ff_sys[t])
            batch = FlatForcefieldBatch(MMFF, a_s, groups)
            pos0 = torch.from_numpy(flat).cuda()
            e_ref = batch.compute_energy(pos0).cpu().numpy()
            barrier.wait()
            for _ in range(rounds):
                with torch.cuda.stream(stream):
                    sim = crossTanimotoSimilarity(x, stream=stream).torch()
                    counts = torch.zeros(len(x), dtype=torch.int32, device="cuda")
                    update_neighbor_counts(x, x, counts, 0.6)            # runs on the thread's current stream
                    e = batch.compute_energy(pos0, stream=stream)
                stream.synchronize()
                assert np.array_equal(sim.cpu().numpy(), want_sim[t])
                assert np.array_equal(counts.cpu().numpy(), want_cnt[t])
                np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=1e-12)
                got = fused_butina(x, 0.4, return_centroids=True, stream=stream)
                assert got[0] == want_but[t][0] and got[2] == want_but[t][2]
            # the error slot is thread-local: an error raised here must carry THIS thread's message
            rc = native_lib.nvmk_cross_tanimoto_f64(x.data_ptr(), len(x), x.data_ptr(), len(x), 2048 + t + 1, None, 0, None)
            assert rc != 0 and str(2048 + t + 1) in native_lib.nvmk_last_error().decode()
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_concurrent_calls_with_team_systems_take_turns():
    """The workgroups of a BFGS team wait for each other inside the launch (csrc/bfgs_device.inc `Team`): two calls whose team
    kernels each got only part of their workgroups resident would wait for CUs the other holds.  Calls with team systems
    therefore take turns per device inside the library: three host threads minimise three large systems each, at once, with
    teams as wide as an XCD — every call completes (no barrier gives up) with the bits of the same call made alone."""
    from nvmolkit_amd.forcefield import DG

    n_threads = 3
    rng = np.random.default_rng(5)
    systems = [util.random_ff_system(DG, n, rng) for n in (210, 260, 300)]
    a_s, flat, groups = util.build_ff_batch_arrays(DG, systems)
    alone = torch.from_numpy(flat).cuda()
    with _native.options(NVMK_BFGS_TEAM_WIDTH="32", NVMK_BFGS_TEAM_TIMEOUT_MS="20000"):
        FlatForcefieldBatch(DG, a_s, groups).minimize(alone, max_iters=6, grad_tol=1e-14, w0=0.7, w1=0.3)
        errors, results, barrier = [], [None] * n_threads, threading.Barrier(n_threads)

        def work(t):
            try:
                torch.cuda.set_device(0)
                stream = torch.cuda.Stream()
                with torch.cuda.stream(stream):
                    batch = FlatForcefieldBatch(DG, a_s, groups)
                    pos = torch.from_numpy(flat).cuda()
                    barrier.wait()
                    for _ in range(3):
                        p = pos.clone()
                        batch.minimize(p, max_iters=6, grad_tol=1e-14, w0=0.7, w1=0.3, stream=stream)
                    stream.synchronize()
                    results[t] = p.cpu().numpy()
            except Exception as exc:  # noqa: BLE001
                errors.append((t, repr(exc)))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    assert not errors, errors
    for r in results:
        assert np.array_equal(r, alone.cpu().numpy())
