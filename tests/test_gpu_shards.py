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
