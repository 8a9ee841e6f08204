"""evofst_run_pairs_noalloc (the form of evofst_run that allocates nothing: one-kernel pairs pass on the library's static
device scratch array) against evofst_run, through the C ABI, on two calls in a row, the second with a smaller input.

Bars (as in test_gpu_pairs.py): window membership, end sites, included totals and the integer dxy numerators exact; the
double sums bit-identical to evofst_run's one-kernel form (same kernel, caller-owned buffers) and within 1e-12 of a
segment's terms of its default three-kernel form."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import random_codes, round_robin_pops

pytestmark = pytest.mark.gpu

H2D, D2H = 1, 2


class Hip:
    """The few HIP runtime calls the test needs for buffers of its own (the caller's inputs and outputs)."""

    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    def free_bytes(self):
        """Free device memory as the runtime reports it: any hipMalloc / hipFree between two readings shows."""
        fr, tot = C.c_size_t(), C.c_size_t()
        assert self.L.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
        return fr.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(16, nbytes)) == 0
        self.bufs.append(p)
        return p

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        assert self.L.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, H2D) == 0
        return p

    def down(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.L.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, D2H) == 0
        return out

    def free_all(self):
        for p in self.bufs:
            self.L.hipFree(p)
        self.bufs = []


ROW = np.dtype([("num", "<f8"), ("den", "<f8"), ("k", "<f8"), ("end", "<i4"), ("pad", "<i4")])


def _one_call(engine, hip, rng, S, N, P, size, step, miss):
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    fl = (rng.random(S) < 0.03).astype(np.uint8)
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(size, step)
    tiles = engine.pack_codes(codes)
    nq = engine.n_pairs
    # the existing entry point, in both of its forms
    engine.upload(tiles, S, site_flags=fl)
    engine.run()
    ref = dict(seg=engine.segments(), tot=engine.included_totals().copy(), carry=engine.boundary().copy())
    engine.set_option("pairs_pass", 3)               # the lock-step one-kernel pass: what the entry point below runs
    try:
        engine.run()
        one = dict(seg=engine.segments(), tot=engine.included_totals().copy(), carry=engine.boundary().copy())
    finally:
        engine.set_option("pairs_pass", 0)
    # the entry point that allocates nothing, on buffers of the caller
    g = int(np.gcd(size, step))
    cap = S // g + 2
    d_tiles, d_fl = hip.up(tiles), hip.up(fl)
    d_rows, d_first, d_tot, d_carry = hip.alloc(cap * nq * 32), hip.alloc(nq * 8), hip.alloc(nq * 8), hip.alloc(nq * 32)
    rc = engine.L.evofst_run_pairs_noalloc(engine.h, S, d_tiles, d_fl, d_rows, cap, d_first, d_tot, d_carry)
    assert rc == 0, engine.L.evofst_last_error(engine.h)
    engine.sync()
    # the property the entry point exists for: once more on the same buffers (code objects are loaded now), and the
    # runtime's free memory is the same before and after -- nothing was allocated, nothing freed
    free_before = hip.free_bytes()
    rc = engine.L.evofst_run_pairs_noalloc(engine.h, S, d_tiles, d_fl, d_rows, cap, d_first, d_tot, d_carry)
    assert rc == 0, engine.L.evofst_last_error(engine.h)
    engine.sync()
    assert hip.free_bytes() == free_before
    rows = hip.down(d_rows, (cap, nq), ROW)
    first = hip.down(d_first, nq, np.int64); tot = hip.down(d_tot, nq, np.int64); carry = hip.down(d_carry, (nq, 4), np.float64)
    hip.free_all()
    for other, exact in ((one, True), (ref, False)):
        assert np.array_equal(tot, other["tot"])
        assert np.array_equal(first, other["seg"]["first_segment"])
        nseg = other["seg"]["n_segments"]
        assert nseg.max() > 3
        for q in range(nq):
            n = int(nseg[q])
            assert np.array_equal(rows["end"][:n, q], other["seg"]["end_site"][:n, q])
            assert np.array_equal(rows["k"][:n, q], other["seg"]["sum_dxy_k"][:n, q])
            for name, key in (("num", "sum_num"), ("den", "sum_den")):
                a, b = rows[name][:n, q], other["seg"][key][:n, q]
                assert (np.isnan(a) == np.isnan(b)).all()
                if exact:
                    assert np.array_equal(a, b, equal_nan=True)
                else:
                    # every |term| of either sum is at most the site's denominator term
                    fin = ~np.isnan(b) & ~np.isnan(other["seg"]["sum_den"][:n, q])
                    scale = np.abs(other["seg"]["sum_den"][:n, q][fin])
                    assert np.all(np.abs(a[fin] - b[fin]) <= 2e-12 * scale + 1e-300)
        assert np.array_equal(carry[:, 2:], other["carry"][:, 2:])
        if exact:
            assert np.array_equal(carry, other["carry"], equal_nan=True)
    return S


def test_no_alloc_entry_point_twice(engine):
    hip = Hip()
    rng = np.random.default_rng(77)
    assert engine.L.evofst_noalloc_scratch_bytes() == 16 << 20
    _one_call(engine, hip, rng, 30000, 40, 4, 100, 100, 0.05)
    _one_call(engine, hip, rng, 2500, 60, 5, 20, 10, 0.0)       # a smaller input on the same scratch, other shapes


def test_no_alloc_entry_point_more_than_64_pairs(engine):
    """78 pairs: the workgroup-per-chunk shape of the one-kernel pass (two groups of lanes share a chunk's region)."""
    hip = Hip()
    rng = np.random.default_rng(78)
    _one_call(engine, hip, rng, 20000, 52, 13, 50, 25, 0.02)


def test_no_alloc_entry_point_refuses_what_does_not_fit(engine):
    """Beyond the static array's size the call fails with a clear error before any launch."""
    hip = Hip()
    N, P, S = 120, 20, 2_000_000          # 190 pairs x 31 250 chunks
    engine.set_layout(round_robin_pops(N, P), P)
    engine.set_pairs()
    engine.set_windows(100, 100)
    d = hip.alloc(64)
    rc = engine.L.evofst_run_pairs_noalloc(engine.h, S, d, None, d, S // 100 + 2, d, d, d)
    hip.free_all()
    assert rc == -5
    assert b"scratch" in engine.L.evofst_last_error(engine.h)
