"""GPU parity: outgroup-to-ingroup dxy in caller-cut windows (evo_distanceToOutgroups.cpp:62-65, 274-294) through the
C ABI vs the CPU oracle.  Counters identical and -- because a lane adds up its (outgroup, ingroup) pair in site order,
as the reference's vector does -- the sums bit-identical.  The oracle's functions here are a restatement (the per-site
formula is the Fst denominator's, which is pinned on the reference header): parity unpinned beyond that."""
import numpy as np
import pytest

from tests.helpers import assert_same_bits, random_codes, round_robin_pops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cuts(rng, S, n):
    c = np.sort(rng.integers(0, S + 1, n))
    c[1] = c[0]                                  # an empty window
    return np.concatenate([[0], c, [S]]).astype(np.int64)


def _same(a, b):
    for k in ("nsnp", "miss", "used", "missing"):
        assert np.array_equal(a[k], b[k]), k
    assert_same_bits(a["sum"], b["sum"], "dxy sums")


@pytest.mark.parametrize("S,N,P,miss,flags", [(5000, 40, 7, 0.2, True), (3000, 12, 5, 0.6, False), (2500, 600, 4, 0.05, True)])
def test_dist_windows(engine, S, N, P, miss, flags):
    rng = np.random.default_rng(11 * S + P)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    engine.set_layout(sp, P)
    fl = (rng.random(S) < 0.07).astype(np.uint8) if flags else None
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    engine.run_counts()
    oalt, onm = O.counts_from_codes(codes, sp, P)
    out_idx, in_idx = [0, P - 1, 2], [1, 2, 3, 0]            # an outgroup may also be an ingroup
    wo = _cuts(rng, S, 23)
    got = engine.dist_windows(wo, out_idx, in_idx)
    want = O.dist_windows(oalt, onm, wo, out_idx, in_idx, fl)
    _same(got, want)
    assert want["nsnp"].sum() > S and want["sum"].max() > 1
    if miss >= 0.5:
        assert want["miss"].sum() > 0 and want["missing"].sum() > 0
    # windows that do not start at 0 / do not reach the end, and no ingroup at all
    sub = wo[3:12]
    _same(engine.dist_windows(sub, out_idx, in_idx), O.dist_windows(oalt, onm, sub, out_idx, in_idx, fl))
    e = engine.dist_windows(wo, out_idx, [])
    assert np.array_equal(e["used"], want["used"]) and np.array_equal(e["missing"], want["missing"])


def test_dist_windows_carry_across_shards(engine):
    """A window that straddles two uploads: the second call continues the first call's last window, and the sums
    stay bit-identical to one pass over all sites.  70 ingroups: more than one lane pass."""
    S, N, P = 4000, 150, 75
    rng = np.random.default_rng(77)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.3)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    out_idx, in_idx = [3, 74], list(range(4, 74))
    wo = np.array([0, 700, 1500, 2600, 3100, S], np.int64)     # window [1500, 2600) straddles the cut at 2000
    want = O.dist_windows(oalt, onm, wo, out_idx, in_idx)
    engine.set_layout(sp, P)
    cut = 2000
    engine.upload(engine.pack_codes(codes[:cut]), cut)
    engine.run_counts()
    a = engine.dist_windows(np.array([0, 700, 1500, cut]), out_idx, in_idx)
    engine.upload(engine.pack_codes(codes[cut:]), S - cut)
    engine.run_counts()
    b = engine.dist_windows(np.array([0, 2600 - cut, 3100 - cut, S - cut]), out_idx, in_idx, carry={k: v[-1] for k, v in a.items()})
    got = {k: np.concatenate([a[k][:-1], b[k]]) for k in a}
    _same(got, want)


def test_dist_windows_rejects_bad_arguments(engine):
    S, N, P = 256, 8, 4
    sp = round_robin_pops(N, P)
    codes = random_codes(np.random.default_rng(1), S, sp, P, miss=0.1)
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    engine.run_counts()
    with pytest.raises(Exception):
        engine.dist_windows([0, 300], [0], [1])               # beyond the shard
    with pytest.raises(Exception):
        engine.dist_windows([0, 100, 50], [0], [1])           # decreasing
    with pytest.raises(Exception):
        engine.dist_windows([0, 100], [P], [1])               # no such population
