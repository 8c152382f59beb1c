"""ETKDG scheduler: the reference's exact dispatch sequences (tests/test_etkdg_result_tracker.cpp:45-190 — the
expected id lists are data held by those tests).  Host logic only: runs without a GPU."""

import ctypes

import numpy as np
import pytest

from nvmolkit_amd import _native


class Sched:
    def __init__(self, lib, n, confs, iters):
        self.lib = lib
        self.h = lib.nvmk_scheduler_create(n, confs, iters)

    def dispatch(self, batch):
        out = np.zeros(max(batch, 1), dtype=np.int32)
        n = ctypes.c_int(0)
        _native.check(self.lib.nvmk_scheduler_dispatch(self.h, batch, out.ctypes.data, ctypes.byref(n)))
        return out[:n.value].tolist()

    def record(self, ids, res):
        ids = np.asarray(ids, dtype=np.int32)
        res = np.asarray(res, dtype=np.int16)
        _native.check(self.lib.nvmk_scheduler_record(self.h, ids.ctypes.data, res.ctypes.data, len(ids)))

    def __del__(self):
        if self.h:
            self.lib.nvmk_scheduler_destroy(self.h)


@pytest.mark.parametrize("args", [(-1, 3, 2), (5, -1, 2), (5, 3, -1), (0, 3, 2), (5, 0, 2), (5, 3, 0)])
def test_constructor_rejects_non_positive(native_lib, args):
    assert not native_lib.nvmk_scheduler_create(*args)
    assert b"greater than 0" in native_lib.nvmk_last_error()


def test_basic_dispatch_oversubscribe(native_lib):
    s = Sched(native_lib, 5, 3, 2)
    for _ in range(2):
        assert s.dispatch(5) == [0, 0, 0, 1, 1]
        assert s.dispatch(5) == [1, 2, 2, 2, 3]
        assert s.dispatch(5) == [3, 3, 4, 4, 4]
    assert s.dispatch(5) == []


def test_full_complete_out_of_order_records(native_lib):
    s = Sched(native_lib, 5, 3, 2)
    a, b, c = s.dispatch(5), s.dispatch(5), s.dispatch(5)
    for ids in (c, a, b):
        s.record(ids, [1] * 5)
    assert s.dispatch(5) == []


def test_partial_complete(native_lib):
    s = Sched(native_lib, 5, 3, 2)
    a = s.dispatch(5)
    b = s.dispatch(5)
    s.record(a, [1] * 5)
    c = s.dispatch(5)
    assert c == [3, 3, 4, 4, 4]
    assert s.dispatch(5) == [1, 1, 1, 2, 2]
    s.record(b, [1] * 5)
    s.record(c, [1] * 5)
    assert s.dispatch(5) == []


def test_some_failures(native_lib):
    s = Sched(native_lib, 5, 3, 2)
    a, b, c = s.dispatch(5), s.dispatch(5), s.dispatch(5)
    s.record(a, [3, 2, 1, 4, 0])
    s.record(b, [-1, -1, 0, 1, 2])
    s.record(c, [3, 2, 1, 4, 0])
    assert s.dispatch(5) == [1, 1, 1, 2, 2]
    assert s.dispatch(5) == [2]
    assert s.dispatch(5) == []


def test_large_batch_and_edges(native_lib):
    assert Sched(native_lib, 2, 2, 4).dispatch(100) == [0, 0, 1, 1] * 4
    s = Sched(native_lib, 5, 3, 2)
    assert s.dispatch(0) == []
    assert len(s.dispatch(1)) == 1
    with pytest.raises(ValueError):
        s.record([0, 5], [1, 1])
    with pytest.raises(ValueError):
        s.record([0, -1], [1, 1])
