"""Run by tests/test_gpu_shards.py::test_device_side_exchange_equals_host_forms in a process of its own: torch has to
initialise the GPU before libevofst.so does (as in bench.py), which a pytest session that already holds an EvoFst
context cannot offer.  Exit status 0 = all comparisons held."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import EvoFst, shard                      # noqa: E402
from tests.helpers import random_codes, round_robin_pops   # noqa: E402


def main():
    engine = EvoFst(0)
    rng = np.random.default_rng(31)
    S, N, P, size, step = 5000, 40, 5, 60, 20
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.15)
    cuts = [0, 1700, 3333, S]
    n_shards = 3
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(size, step)
    nq = engine.n_pairs
    stream = torch.cuda.Stream()
    engine.set_stream(stream.cuda_stream)

    def load(r):
        engine.upload(engine.pack_codes(codes[cuts[r]:cuts[r + 1]]), cuts[r + 1] - cuts[r])
        engine.run_counts()

    # host forms
    totals = []
    for r in range(n_shards):
        load(r); totals.append(engine.included_totals().copy())
    host = []
    for r in range(n_shards):
        load(r)
        engine.set_base_offsets(shard.base_offsets(totals, r))
        engine.run_stats()
        host.append((engine.segments(), engine.boundary().copy()))
    # device forms: nothing between run_counts and copy_boundary waits on the host
    t_all = torch.zeros(n_shards * nq, dtype=torch.int64, device="cuda")
    stream.wait_stream(torch.cuda.current_stream())      # the fill above ran on torch's current stream
    with torch.cuda.stream(stream):
        for r in range(n_shards):
            load(r)
            engine.copy_included_totals(t_all[r * nq:(r + 1) * nq].data_ptr())
    stream.synchronize()
    assert np.array_equal(t_all.cpu().numpy().reshape(n_shards, nq), np.array(totals))
    t_carry = torch.zeros(nq * 4, dtype=torch.float64, device="cuda")
    stream.wait_stream(torch.cuda.current_stream())
    for r in range(n_shards):
        with torch.cuda.stream(stream):
            load(r)
            base = t_all.view(n_shards, nq)[:r].sum(0) if r > 0 else torch.zeros(nq, dtype=torch.int64, device="cuda")
            engine.set_base_offsets_device(base.data_ptr())
            engine.run_stats()
            engine.copy_boundary(t_carry.data_ptr())
        stream.synchronize()
        seg, carry = host[r]
        assert np.array_equal(t_carry.cpu().numpy().reshape(nq, 4), carry, equal_nan=True), r
        got = engine.segments()
        assert np.array_equal(got["first_segment"], seg["first_segment"]) and np.array_equal(got["n_segments"], seg["n_segments"])
        for k in ("sum_num", "sum_den", "sum_dxy_k", "end_site"):
            for q in range(nq):
                n = int(seg["n_segments"][q])
                assert np.array_equal(np.asarray(got[k])[:n, q], np.asarray(seg[k])[:n, q], equal_nan=True), (r, k, q)
    # the same offsets formed by the library from the gathered buffer itself: rows padded to 5 * nq values (totals
    # first, as bench.py's message is laid out), shard r sums rows 0 .. r-1
    t_msg = torch.full((n_shards, 5 * nq), 7_777, dtype=torch.int64, device="cuda")
    t_msg[:, :nq] = t_all.view(n_shards, nq)
    stream.wait_stream(torch.cuda.current_stream())
    for r in range(n_shards):
        with torch.cuda.stream(stream):
            load(r)
            engine.set_base_offsets_gathered(t_msg.data_ptr(), 5 * nq, r)
            engine.run_stats()
            engine.copy_boundary(t_carry.data_ptr())
        stream.synchronize()
        seg, carry = host[r]
        assert np.array_equal(t_carry.cpu().numpy().reshape(nq, 4), carry, equal_nan=True), r
        got = engine.segments()
        assert np.array_equal(got["first_segment"], seg["first_segment"]) and np.array_equal(got["n_segments"], seg["n_segments"])
        for q in range(nq):
            n = int(seg["n_segments"][q])
            assert np.array_equal(np.asarray(got["sum_num"])[:n, q], np.asarray(seg["sum_num"])[:n, q], equal_nan=True), (r, q)
    engine.close()
    print("device-side exchange: ok")


if __name__ == "__main__":
    main()
