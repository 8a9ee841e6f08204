#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload: evofst_upload of cfg2's tiles from pageable and from
pinned host memory + one pass, against the HBM-resident pass.  python3 tools/pcie_rate.py   (needs a GPU)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evo_amd import EvoFst

S, N, P = 1_000_000, 200, 10
e = EvoFst(0)
e.set_layout((np.arange(N) % P).astype(np.int32), P)
e.set_pairs(); e.set_windows(100, 100); e.reserve(S)
e.synth(S, seed=20261110); e.sync()
tiles = e.download_tiles()
nbytes = tiles.nbytes


def timed(fn, reps=20):
    fn(); e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    e.sync()
    return (time.perf_counter() - t0) / reps


def resident():
    e.run()


def from_pageable():
    e.upload(tiles, S); e.run()


e.L.evofst_alloc_pinned.restype = C.c_void_p
pin = e.L.evofst_alloc_pinned(e.h, nbytes)
ptiles = np.ctypeslib.as_array((C.c_uint64 * (nbytes // 8)).from_address(pin))
ptiles[:] = tiles


def from_pinned():
    e.upload(ptiles, S); e.run()


evals = S * e.n_pairs
for name, fn in (("HBM-resident", resident), ("upload from pageable memory + pass", from_pageable), ("upload from pinned memory + pass", from_pinned)):
    dt = timed(fn)
    print("%-36s %.3f ms  %.3g site*pair/s  (%.1f GB/s of tiles)" % (name, dt * 1e3, evals / dt, nbytes / dt / 1e9))
