"""Multi-GPU sharding of the hot path: contiguous site shards, one per rank.

Sites are independent; only SNP windows couple neighbouring shards.  Two small exchanges per pass,
both all-gathers of per-pair vectors (RCCL over xGMI when the tensors live on the GPU, gloo in the
CPU tests):

  1. after K1:  included-site totals of every shard  -> each rank's base offsets (global window phase),
  2. after K2:  the carry behind the last complete segment of every shard (+ its last SIZE/g - 1
                complete segments when windows overlap) -> each rank completes the windows that
                straddle its left edge.

All functions work on numpy arrays (what EvoFst returns) plus an `all_gather(array) -> [array per
rank]` callable, so the same code runs under torch.distributed and in single-process tests.
"""
import math

import numpy as np


def torch_all_gather(group=None, device=None):
    """all_gather callable on top of torch.distributed (nccl = RCCL on ROCm, or gloo)."""
    import torch
    import torch.distributed as dist

    def ag(a):
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a)
        if device is not None:
            t = t.to(device)
        out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, t, group=group)
        return [o.cpu().numpy() for o in out]
    return ag


class PairsPassMessage:
    """The one message a rank sends per pass in pairs mode when the pass stays on the device (bench.py):
    row = [included-site totals of this pass (n_pairs x int64) | carries of the pass before (n_pairs x 4 float64, bit
    patterns)].  The totals are needed on the device before K2 (evofst_set_base_offsets_gathered sums the rows of the
    earlier ranks); the carries are host bookkeeping after the pass, so they ride with the next pass's totals and one
    more gather after the last pass delivers the rest.  Tensors are torch tensors on `device`."""

    def __init__(self, n_pairs, world, rank, device):
        import torch
        self.nq, self.world, self.rank = n_pairs, world, rank
        self.row = 5 * n_pairs                                   # int64 values per rank
        self.send = torch.zeros(self.row, dtype=torch.int64, device=device)
        self.recv = torch.zeros(world * self.row, dtype=torch.int64, device=device)

    # where the library writes / reads (device pointers for evofst_copy_included_totals, evofst_copy_boundary,
    # evofst_set_base_offsets_gathered)
    def totals_ptr(self):
        return self.send.data_ptr()

    def carry_ptr(self):
        return self.send[self.nq:].data_ptr()

    def recv_ptr(self):
        return self.recv.data_ptr()

    def gather(self, dist):
        dist.all_gather_into_tensor(self.recv, self.send)

    def gathered_totals(self):
        """[world][n_pairs] int64: every rank's included-site totals of the pass just gathered"""
        return self.recv.view(self.world, self.row)[:, :self.nq]

    def gathered_carries(self):
        """[world][n_pairs][4] float64: every rank's carries of the pass before the one just gathered"""
        import torch
        return self.recv.view(self.world, self.row)[:, self.nq:].contiguous().view(torch.float64).view(self.world, self.nq, 4)


def base_offsets(all_totals, rank):
    """Exchange 1: included sites (per pair) that precede shard `rank`."""
    tot = np.asarray(all_totals[0]) * 0
    for r in range(rank):
        tot = tot + np.asarray(all_totals[r])
    return tot.astype(np.int64)


def carry_in(all_carry, all_nseg, rank):
    """Exchange 2a: what precedes shard `rank` inside its first segment, per pair: the carries of the
    previous shards back to (and including) the last one that completed a segment.
    all_carry[r]: [n_pairs, 4] {num, den, k, n_sites}; all_nseg[r]: [n_pairs] complete segments."""
    n_pairs = np.asarray(all_carry[0]).shape[0]
    out = np.zeros((n_pairs, 4))
    for q in range(n_pairs):
        r = rank - 1
        chain = []
        while r >= 0:
            chain.append(np.asarray(all_carry[r])[q])
            if all_nseg[r][q] > 0:
                break
            r -= 1
        acc = np.zeros(4)
        for c in reversed(chain):          # site order: oldest shard first
            acc = acc + c
        out[q] = acc
    return out


def tail_segments(seg, per):
    """The last per-1 complete segments of a shard, per pair, newest last: [n_pairs, per-1, 3] and the
    number that are valid.  `seg` is EvoFst.segments() with carry_in already added to row 0."""
    Q = seg["n_segments"].shape[0]
    k = max(per - 1, 0)
    out = np.zeros((Q, k, 3)); cnt = np.zeros(Q, np.int64)
    for q in range(Q):
        n = int(seg["n_segments"][q]); m = min(n, k)
        cnt[q] = m
        if m:
            out[q, k - m:, 0] = seg["sum_num"][n - m:n, q]
            out[q, k - m:, 1] = seg["sum_den"][n - m:n, q]
            out[q, k - m:, 2] = seg["sum_dxy_k"][n - m:n, q]
    return out, cnt


def halo_segments(all_tail, all_cnt, rank, per):
    """Exchange 2b: the per-1 complete segments that precede shard `rank` globally, newest last."""
    Q = np.asarray(all_tail[0]).shape[0]
    k = max(per - 1, 0)
    out = np.zeros((Q, k, 3)); cnt = np.zeros(Q, np.int64)
    for q in range(Q):
        rows = []
        r = rank - 1
        while r >= 0 and len(rows) < k:
            m = int(all_cnt[r][q])
            t = np.asarray(all_tail[r])[q]
            for i in range(m):                 # newest first
                if len(rows) < k:
                    rows.append(t[k - 1 - i])
            r -= 1
        cnt[q] = len(rows)
        for i, row in enumerate(rows):
            out[q, k - 1 - i] = row
    return out, cnt


def add_carry(seg, cin):
    """Adds carry_in to the first complete segment of each pair (in place on copies)."""
    seg = dict(seg)
    for name in ("sum_num", "sum_den", "sum_dxy_k"):
        seg[name] = seg[name].copy()
    for q in range(seg["n_segments"].shape[0]):
        if seg["n_segments"][q] > 0:
            seg["sum_num"][0, q] = cin[q, 0] + seg["sum_num"][0, q]
            seg["sum_den"][0, q] = cin[q, 1] + seg["sum_den"][0, q]
            seg["sum_dxy_k"][0, q] = cin[q, 2] + seg["sum_dxy_k"][0, q]
    return seg


def shard_windows(seg, pair, size, step, n_full_i, n_full_j, halo=None, halo_cnt=None, site_offset=0):
    """Windows (process_vcf_fst.cpp:638-650) that END in this shard, from its carried segments and the
    halo segments of the previous shards.  Same output dict as evo_amd.compose_snp_windows; end_site
    is made global with site_offset."""
    g = math.gcd(size, step)
    per = size // g
    first = int(seg["first_segment"][pair]); rows = int(seg["n_segments"][pair])
    loc = np.stack([seg["sum_num"][:rows, pair], seg["sum_den"][:rows, pair], seg["sum_dxy_k"][:rows, pair]], axis=1)
    nh = int(halo_cnt[pair]) if halo is not None else 0
    pre = halo[pair][per - 1 - nh:] if nh else np.zeros((0, 3))
    allseg = np.concatenate([pre, loc], axis=0)
    es = seg["end_site"][:rows, pair]
    out = dict(first_var=[], end_site=[], sum_num=[], sum_den=[], sum_dxy=[], fst=[], mean_dxy=[])
    denom = float(n_full_i) * float(n_full_j)
    for r in range(rows):
        e = (first + r + 1) * g
        if e % step != 0 or e < size:
            continue
        hi = nh + r
        lo = hi - per + 1
        if lo < 0:
            continue
        w = np.zeros(3)
        for t in range(lo, hi + 1):
            w = w + allseg[t]
        with np.errstate(all="ignore"):
            f = float(np.float64(w[0] / size) / np.float64(w[1] / size))
        if f < 0:
            f = 0.0
        out["first_var"].append(e - size + 1); out["end_site"].append(int(es[r]) + site_offset)
        out["sum_num"].append(w[0]); out["sum_den"].append(w[1]); out["sum_dxy"].append(w[2] / denom)
        out["fst"].append(f); out["mean_dxy"].append(w[2] / denom / size)
    return {k: np.asarray(v) for k, v in out.items()}


def run_sharded_pairs(engine, all_gather, rank, size, step):
    """One distributed pass in pairs mode on data already loaded in `engine`.
    Returns (segments with carry, halo, halo_cnt, base) for shard_windows()."""
    engine.set_windows(size, step)
    engine.run_counts()
    base = base_offsets(all_gather(engine.included_totals()), rank)          # exchange 1
    engine.set_base_offsets(base)
    engine.run_stats()
    seg = engine.segments()
    carry = engine.boundary()
    per = size // math.gcd(size, step)
    all_carry = all_gather(carry)                                           # exchange 2a
    all_nseg = all_gather(seg["n_segments"].astype(np.int64))
    seg = add_carry(seg, carry_in(all_carry, all_nseg, rank))
    tail, cnt = tail_segments(seg, per)
    all_tail = all_gather(tail) if per > 1 else [tail]                       # exchange 2b
    all_cnt = all_gather(cnt) if per > 1 else [cnt]
    halo, hcnt = halo_segments(all_tail, all_cnt, rank, per) if per > 1 else (None, None)
    return seg, halo, hcnt, base


# ---------------------------------------------------------------------------------------------
# Fst-against-all (evo_FstAgainstAll.cpp): the deques of the last SIZE kept sites cross shard edges
# ---------------------------------------------------------------------------------------------

def global_tail(engine, size, block=8192):
    """The last (up to) `size` kept sites of the shard loaded in `engine` (after run_counts), oldest first:
    (local site index [n], num [n, U], den [n, U])."""
    hi = engine.n_sites
    sites, nums, dens = [], [], []
    got = 0
    while hi > 0 and got < size:
        lo = max(0, hi - block)
        num, den, kept = engine.global_site_stats(lo, hi - lo)           # [U, n]
        idx = np.nonzero(kept)[0]
        if len(idx) > size - got:
            idx = idx[len(idx) - (size - got):]
        sites.insert(0, idx + lo); nums.insert(0, num[:, idx].T); dens.insert(0, den[:, idx].T)
        got += len(idx)
        hi = lo
    U = engine.n_use
    if not sites:
        return np.zeros(0, np.int64), np.zeros((0, U)), np.zeros((0, U))
    return np.concatenate(sites).astype(np.int64), np.concatenate(nums), np.concatenate(dens)


def global_halo(all_tails, all_first_site, rank, size):
    """What the deques hold when shard `rank` starts: the last `size` kept sites of the shards before it, oldest
    first.  all_tails[r] = (sites, num, den, n) as gathered (padded to `size` rows, n valid, newest last at the
    end of the valid part).  Returns (global site index [m], num [m, U], den [m, U])."""
    sites, nums, dens = [], [], []
    got = 0
    r = rank - 1
    while r >= 0 and got < size:
        s, nu, de, n = all_tails[r]
        n = int(np.asarray(n).reshape(-1)[0])
        take = min(n, size - got)
        if take:
            sites.insert(0, np.asarray(s)[n - take:n] + int(all_first_site[r]))
            nums.insert(0, np.asarray(nu)[n - take:n]); dens.insert(0, np.asarray(de)[n - take:n])
            got += take
        r -= 1
    if not sites:
        U = np.asarray(all_tails[0][1]).shape[1]
        return np.zeros(0, np.int64), np.zeros((0, U)), np.zeros((0, U))
    return np.concatenate(sites), np.concatenate(nums), np.concatenate(dens)


def compute_sharded_global(engine, all_gather, rank, size, step):
    """The device work and the two exchanges of one distributed against-all pass ({used, kept} totals; the last
    SIZE kept sites of every shard); leaves the window rows on the device.  Returns (global sites of the halo,
    first global site of every shard) for run_sharded_global's bookkeeping."""
    engine.set_windows(size, step)
    engine.run_counts()
    tot = engine.included_totals()                                            # {used, kept}
    all_tot = all_gather(np.asarray(tot, np.int64))
    all_n = all_gather(np.asarray([engine.n_sites], np.int64))
    first_site = np.concatenate([[0], np.cumsum([int(x[0]) for x in all_n])])[:-1]
    engine.set_base_offsets(base_offsets(all_tot, rank))                      # exchange 1
    s, nu, de = global_tail(engine, size)
    U = engine.n_use
    pad = lambda a, shape: np.concatenate([a, np.zeros((size - a.shape[0],) + shape, a.dtype)])
    all_s = all_gather(pad(s, ())); all_nu = all_gather(pad(nu, (U,))); all_de = all_gather(pad(de, (U,)))
    all_cnt = all_gather(np.asarray([len(s)], np.int64))                      # exchange 2
    tails = [(all_s[r], all_nu[r], all_de[r], all_cnt[r]) for r in range(len(all_cnt))]
    hs, hn, hd = global_halo(tails, first_site, rank, size)
    engine.set_global_halo(-2 - np.arange(len(hs)), hn, hd)        # -1 stays the reference's empty deque slot
    engine.run_stats()
    return hs, first_site


def run_sharded_global(engine, all_gather, rank, size, step):
    """One distributed pass in against-all mode on the shard loaded in `engine`.  Returns the window rows that
    END in this shard with global site indices: dict(start_site, end_site, mean_num, mean_den)."""
    hs, first_site = compute_sharded_global(engine, all_gather, rank, size, step)
    w = engine.global_windows()
    start = w["start_site"].astype(np.int64)
    out_start = start + int(first_site[rank])
    out_start[start == -1] = -1                                   # the reference's zero-initialised deque
    halo_rows = start < -1
    if halo_rows.any():
        out_start[halo_rows] = hs[(-2 - start[halo_rows]).astype(np.int64)]
    return dict(start_site=out_start, end_site=w["end_site"].astype(np.int64) + int(first_site[rank]),
                mean_num=w["mean_num"], mean_den=w["mean_den"])
