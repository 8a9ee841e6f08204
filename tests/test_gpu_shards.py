"""GPU: the same input cut into 1 / 2 / 4 site shards (processed one after the other on one GPU, the
"fake cluster" of SURVEY section 4) gives the same window table as the oracle on the whole input."""
import math

import numpy as np
import pytest

from evo_amd import shard
from oracle import oracle as O
from tests.helpers import random_codes, round_robin_pops, n_full_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_shards,size,step", [(1, 100, 100), (2, 100, 100), (4, 100, 100), (3, 60, 20), (4, 1000, 1000)])
def test_shards_agree_with_oracle(engine, n_shards, size, step):
    rng = np.random.default_rng(17)
    S, N, P = 6000, 40, 4
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.1)
    nf = n_full_of(sp, P)
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(size, step)
    pi, pj = engine.all_pairs()
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(64, S - 64), n_shards - 1, replace=False)) + [S]
    g = math.gcd(size, step); per = size // g
    # phase 1: K1 on every shard, exchange 1
    totals = []
    for r in range(n_shards):
        engine.upload(engine.pack_codes(codes[cuts[r]:cuts[r + 1]]), cuts[r + 1] - cuts[r])
        engine.run_counts()
        totals.append(engine.included_totals().copy())
    # phase 2: K2 on every shard with its base offsets
    segs, carries = [], []
    for r in range(n_shards):
        engine.upload(engine.pack_codes(codes[cuts[r]:cuts[r + 1]]), cuts[r + 1] - cuts[r])
        engine.run_counts()
        engine.set_base_offsets(shard.base_offsets(totals, r))
        engine.run_stats()
        segs.append(engine.segments()); carries.append(engine.boundary())
    nsegs = [s["n_segments"] for s in segs]
    # exchange 2
    segs = [shard.add_carry(segs[r], shard.carry_in(carries, nsegs, r)) for r in range(n_shards)]
    tails = [shard.tail_segments(s, per) for s in segs]
    oalt, onm = O.counts_from_codes(codes, sp, P)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj)
    for q in range(len(pi)):
        parts = []
        for r in range(n_shards):
            halo, hcnt = shard.halo_segments([t[0] for t in tails], [t[1] for t in tails], r, per)
            parts.append(shard.shard_windows(segs[r], q, size, step, nf[pi[q]], nf[pj[q]], halo, hcnt, site_offset=cuts[r]))
        ow = O.snp_windows(onum, oden, odxy, oinc, q, size, step)
        cat = lambda k: np.concatenate([p[k] for p in parts])
        assert np.array_equal(cat("first_var"), ow["first_var"])
        assert np.array_equal(cat("end_site"), ow["end_site"])
        for k in ("sum_num", "sum_den", "sum_dxy"):
            assert np.allclose(cat(k), ow[k], rtol=0, atol=1e-12 * size, equal_nan=True), (q, k)


@pytest.mark.parametrize("n_shards,size,step,miss", [(2, 20, 10, 0.05), (4, 20, 10, 0.3), (3, 50, 50, 0.1), (5, 200, 40, 0.6),
                                                     (6, 600, 200, 0.2)])
def test_global_shards_agree_with_oracle(engine, n_shards, size, step, miss):
    """Fst-against-all over site shards: the deques cross shard edges through the halo of the last SIZE kept
    sites (evo_amd.shard.run_sharded_global); rows, sites and window Fst equal the oracle's on the whole input."""
    from tests.helpers import assert_same_bits
    rng = np.random.default_rng(23 + n_shards)
    S, N, P = 5000, 48, 6
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    fl = (rng.random(S) < 0.05).astype(np.uint8)
    use = np.arange(P, dtype=np.int32)
    sizes = np.array([(sp == k).sum() for k in range(P)], np.int64)
    engine.set_layout(sp, P)
    engine.set_against_all(use, sizes)
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(70, S - 70), n_shards - 1, replace=False)) + [S]

    # the shards run one after the other on one GPU; each all_gather call is answered from a script recorded
    # in a first pass (what the other ranks would send), the way the fake cluster of SURVEY section 4 works
    def load(r):
        engine.upload(engine.pack_codes(codes[cuts[r]:cuts[r + 1]]), cuts[r + 1] - cuts[r], site_flags=fl[cuts[r]:cuts[r + 1]])
    sent = []
    for r in range(n_shards):
        load(r)
        engine.set_windows(size, step)
        engine.run_counts()
        s, nu, de = shard.global_tail(engine, size)
        pad = lambda a, shape: np.concatenate([a, np.zeros((size - a.shape[0],) + shape, a.dtype)])
        sent.append([np.asarray(engine.included_totals(), np.int64), np.asarray([engine.n_sites], np.int64),
                     pad(s, ()), pad(nu, (P,)), pad(de, (P,)), np.asarray([len(s)], np.int64)])
    rows = []
    for r in range(n_shards):
        load(r)
        state = {"i": 0}

        def all_gather(a, state=state):
            i = state["i"]; state["i"] += 1
            assert np.array_equal(np.asarray(a), sent[r][i])          # this rank sends what the script recorded
            return [sent[q][i] for q in range(n_shards)]
        rows.append(shard.run_sharded_global(engine, all_gather, r, size, step))
    oalt, onm = O.counts_from_codes(codes, sp, P)
    okept, onum, oden = O.global_site_stats(oalt, onm, use, sizes, fl)
    ow = O.global_windows(okept, onum, oden, size, step, fl)
    cat = lambda k: np.concatenate([w[k] for w in rows])
    assert np.array_equal(cat("end_site"), ow["end_site"])
    assert np.array_equal(cat("start_site"), ow["start_site"])
    with np.errstate(all="ignore"):
        mn, md = cat("mean_num"), cat("mean_den")
        gf = mn / md
        gf[(gf < 0) | (md == 0)] = 0
    assert_same_bits(gf, ow["fst"], "window Fst across shards")


def test_device_side_exchange_equals_host_forms():
    """The entry points bench.py's multi-rank step uses -- evofst_copy_included_totals, evofst_set_base_offsets_device,
    evofst_copy_boundary, all on a caller's stream with device buffers -- against the host forms (included_totals,
    set_base_offsets, boundary, segments) on three shards with non-zero base offsets.  The "all-gather" is a plain
    device tensor that collects what each shard copied out.  Runs in a process of its own
    (tests/_device_exchange_check.py): torch must initialise the GPU before the library does, as in bench.py."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_device_exchange_check.py")], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "device-side exchange: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
