"""Multi-rank path on CPU: world_size-2 (and 3) gloo process groups run the two window-boundary
exchanges of evo_amd.shard on per-shard outputs and must reproduce the single-shard window table."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from evo_amd import shard
from oracle import oracle as O
from tests.helpers import random_codes, round_robin_pops, n_full_of, emulate_shard_outputs, CpuGlobalShard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    rng = np.random.default_rng(42)
    S, N, P = 3000, 40, 4
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.1)
    alt, nm = O.counts_from_codes(codes, sp, P)
    nf = n_full_of(sp, P)
    pi, pj = np.triu_indices(P, 1)
    num, den, dxy, inc = O.pair_site_stats(alt, nm, nf, pi, pj)
    kk = np.rint(dxy * (nf[pi] * nf[pj])[None, :])          # integer dxy numerators
    return S, nf, pi, pj, num, den, dxy, kk, inc


def _worker(rank, world, port, size, step, cuts, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, nf, pi, pj, num, den, dxy, kk, inc = _problem()
        lo, hi = cuts[rank], cuts[rank + 1]
        g = math.gcd(size, step); per = size // g
        ag = shard.torch_all_gather()
        # exchange 1
        _, _, tot = emulate_shard_outputs(num[lo:hi], den[lo:hi], kk[lo:hi], inc[lo:hi], np.zeros(len(pi), np.int64), g)
        base = shard.base_offsets(ag(tot), rank)
        seg, carry, _ = emulate_shard_outputs(num[lo:hi], den[lo:hi], kk[lo:hi], inc[lo:hi], base, g)
        # exchange 2
        seg = shard.add_carry(seg, shard.carry_in(ag(carry), ag(seg["n_segments"]), rank))
        tail, cnt = shard.tail_segments(seg, per)
        halo, hcnt = shard.halo_segments(ag(tail), ag(cnt), rank, per)
        rows = []
        for q in range(len(pi)):
            w = shard.shard_windows(seg, q, size, step, nf[pi[q]], nf[pj[q]], halo, hcnt, site_offset=lo)
            rows.append((w["first_var"], w["end_site"], w["sum_num"], w["sum_den"], w["sum_dxy"]))
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        if rank == 0:
            ok = True
            for q in range(len(pi)):
                ow = O.snp_windows(num, den, dxy, inc, q, size, step)
                fv = np.concatenate([gathered[r][q][0] for r in range(world)])
                es = np.concatenate([gathered[r][q][1] for r in range(world)])
                sn = np.concatenate([gathered[r][q][2] for r in range(world)])
                sd = np.concatenate([gathered[r][q][3] for r in range(world)])
                sx = np.concatenate([gathered[r][q][4] for r in range(world)])
                ok &= np.array_equal(fv, ow["first_var"]) and np.array_equal(es, ow["end_site"])
                ok &= bool(np.allclose(sn, ow["sum_num"], rtol=0, atol=1e-12 * size, equal_nan=True))
                ok &= bool(np.allclose(sd, ow["sum_den"], rtol=0, atol=1e-12 * size, equal_nan=True))
                ok &= bool(np.allclose(sx, ow["sum_dxy"], rtol=0, atol=1e-12 * size, equal_nan=True))
            ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,step,cuts", [(2, 100, 100, [0, 1500, 3000]), (2, 60, 20, [0, 1111, 3000]),
                                                  (3, 90, 30, [0, 40, 90, 3000]), (2, 500, 500, [0, 100, 3000])])
def test_window_exchange_matches_single_shard(world, size, step, cuts):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, size, step, cuts, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _global_problem():
    rng = np.random.default_rng(7)
    S, N, P = 2500, 48, 6
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.25)
    fl = (rng.random(S) < 0.05).astype(np.uint8)
    alt, nm = O.counts_from_codes(codes, sp, P)
    use = np.arange(P, dtype=np.int32)
    sizes = np.array([(sp == k).sum() for k in range(P)], np.int64)
    kept, num, den = O.global_site_stats(alt, nm, use, sizes, fl)
    return kept, num, den, fl


def _worker_global(rank, world, port, size, step, cuts, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kept, num, den, fl = _global_problem()
        lo, hi = cuts[rank], cuts[rank + 1]
        eng = CpuGlobalShard(kept[lo:hi], num[lo:hi], den[lo:hi], fl[lo:hi])
        w = shard.run_sharded_global(eng, shard.torch_all_gather(), rank, size, step)
        gathered = [None] * world
        dist.all_gather_object(gathered, w)
        if rank == 0:
            ow = O.global_windows(kept, num, den, size, step, fl)
            cat = lambda k: np.concatenate([g[k] for g in gathered])
            ok = np.array_equal(cat("end_site"), ow["end_site"]) and np.array_equal(cat("start_site"), ow["start_site"])
            with np.errstate(all="ignore"):
                mn, md = cat("mean_num"), cat("mean_den")
                f = mn / md
                f[(f < 0) | (md == 0)] = 0
            ok = ok and np.array_equal(f, ow["fst"], equal_nan=True)
            ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,step,cuts", [(2, 20, 10, [0, 1200, 2500]), (3, 50, 25, [0, 30, 1000, 2500]),
                                                  (2, 200, 200, [0, 150, 2500])])
def test_against_all_exchange_matches_single_shard(world, size, step, cuts):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, size, step, cuts, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _message_worker(rank, world, port, n_pass, ret):
    """bench.py's multi-rank pairs pass with the kernels replaced by known numbers: per pass every rank learns the
    totals of all ranks (base offsets = sum of the earlier rows) and, one gather later, the carries of that pass."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nq = 6
        msg = shard.PairsPassMessage(nq, world, rank, "cpu")
        tot_of = lambda r, k: torch.arange(nq, dtype=torch.int64) * 7 + 100 * r + k             # noqa: E731
        car_of = lambda r, k: (torch.arange(nq * 4, dtype=torch.float64) + r + 0.25 * k).view(nq, 4)  # noqa: E731
        ok = True
        for k in range(n_pass):
            msg.send[:nq] = tot_of(rank, k)                   # evofst_copy_included_totals
            msg.gather(dist)
            base = msg.gathered_totals()[:rank].sum(0)        # what evofst_set_base_offsets_gathered forms
            ok &= bool(torch.equal(base, sum((tot_of(r, k) for r in range(rank)), torch.zeros(nq, dtype=torch.int64))))
            if k > 0:
                for r in range(world):
                    ok &= bool(torch.equal(msg.gathered_carries()[r], car_of(r, k - 1)))
            msg.send[nq:] = car_of(rank, k).view(-1).view(torch.int64)   # evofst_copy_boundary (bit patterns)
        msg.gather(dist)                                      # the flush after the last pass
        for r in range(world):
            ok &= bool(torch.equal(msg.gathered_carries()[r], car_of(r, n_pass - 1)))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_message_per_pass_protocol(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_message_worker, args=(world, _free_port(), 4, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)) and len(ret) == world
