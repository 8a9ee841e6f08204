"""Shared generators for the parity tests (seeded, numpy only)."""
import numpy as np


def random_codes(rng, n_sites, sample_pop, n_pops, miss=0.05, edge=True):
    """Allele codes [S, 2N] (0 ref, 1 alt, 2 missing) with per-pop frequencies and the edge cases the
    reference's arithmetic is sensitive to: fixed sites, all-missing pops, pops with one allele left."""
    sample_pop = np.asarray(sample_pop)
    N = len(sample_pop)
    f = rng.beta(0.5, 0.5, size=(n_sites, 1))
    pk = np.clip(f + rng.normal(0, 0.15, size=(n_sites, n_pops)), 0, 1)
    hp = np.where(sample_pop >= 0, sample_pop, 0)
    p_h = np.repeat(pk[:, hp], 2, axis=1)                         # [S, 2N]
    codes = (rng.random((n_sites, 2 * N)) < p_h).astype(np.uint8)
    # missing: mostly whole genotypes, sometimes half calls
    mg = rng.random((n_sites, N)) < miss
    codes[np.repeat(mg, 2, axis=1)] = 2
    half = rng.random((n_sites, 2 * N)) < miss * 0.2
    codes[half] = 2
    if edge and n_sites >= 16:
        hap_pop = np.repeat(sample_pop, 2)
        codes[0, :] = 0                                            # monomorphic ref
        codes[1, :] = 1                                            # fixed alt everywhere
        codes[2, hap_pop == 0] = 2                                 # pop 0 all missing
        idx = np.where(hap_pop == (1 % n_pops))[0]
        codes[3, idx] = 2; codes[3, idx[0]] = 1                    # pop 1 has exactly one allele left (nm == 1)
        codes[4, :] = 2                                            # everything missing
        codes[5, hap_pop == 0] = 1; codes[5, hap_pop != 0] = 0     # fixed difference
        codes[6, idx] = 2; codes[6, idx[:2]] = [0, 1]              # nm == 2
    return codes


class options:
    """with options(engine, chunk_tiles=1, pairs_pass=1): ... -- evofst_set_option for the block, back to 0 (the library's
    own choice) afterwards; the GPU tests share one context."""

    def __init__(self, engine, **kw):
        self.engine, self.kw = engine, kw

    def __enter__(self):
        for k, v in self.kw.items():
            self.engine.set_option(k, v)
        return self.engine

    def __exit__(self, *exc):
        for k in self.kw:
            self.engine.set_option(k, 0)
        return False


# how evofst_run takes pairs mode (evofst_set_option "pairs_pass"): tag suffix of the parity log -> value
PAIRS_PASSES = (("", 1), ("_fused", 3), ("_stream", 2))


def round_robin_pops(n_samples, n_pops):
    return (np.arange(n_samples) % n_pops).astype(np.int32)


def n_full_of(sample_pop, n_pops):
    sp = np.asarray(sample_pop)
    return np.array([2 * int((sp == k).sum()) for k in range(n_pops)], np.int32)


def assert_same_bits(a, b, what=""):
    """Doubles equal bit for bit (NaN == NaN, +0 == +0 but not -0)."""
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ia = a.view(np.uint64); ib = b.view(np.uint64)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = (ia != ib) & ~both_nan
    if bad.any():
        k = np.argwhere(bad)[0]
        raise AssertionError("%s: %d of %d doubles differ, first at %s: %r vs %r" %
                             (what, int(bad.sum()), a.size, tuple(k), a[tuple(k)], b[tuple(k)]))


def write_vcf(path, rng, n_sites, n_samples, scaffolds=("scaffold_0", "chr2"), miss=0.05, crlf=False, weird=True, vary_aa=False):
    """Small seeded VCF with the cases the reference's parsers treat specially: half calls, '*' ALT
    in either position, indels, multi-allelics, long REF, several scaffolds, extra FORMAT fields."""
    names = ["S%03d" % i for i in range(n_samples)]
    eol = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        f.write("##fileformat=VCFv4.2" + eol)
        f.write("##source=evo_amd tests" + eol)
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + eol)
        per = max(1, n_sites // len(scaffolds))
        for si in range(n_sites):
            scaf = scaffolds[min(si // per, len(scaffolds) - 1)]
            pos = (si % per) * 137 + int(rng.integers(1, 120))
            ref, alt = "A", "G"
            kind = rng.random() if weird else 1.0
            if kind < 0.03: alt = "G,*"
            elif kind < 0.05: alt = "*,G"
            elif kind < 0.07: alt = "GT"
            elif kind < 0.09: alt = "G,T"
            elif kind < 0.11: ref = "AT"
            elif kind < 0.12: alt = "*"
            f0 = rng.beta(0.5, 0.5)
            gts = []
            for i in range(n_samples):
                p = min(1.0, max(0.0, f0 + 0.25 * ((i * 7919) % 5 - 2) / 2.0 * rng.random()))
                a = ["1" if rng.random() < p else "0" for _ in range(2)]
                if alt in ("G,*", "*,G", "G,T") and rng.random() < 0.15:
                    a[int(rng.integers(0, 2))] = "2"
                r = rng.random()
                if r < miss: a = [".", "."]
                elif r < miss * 1.3: a[int(rng.integers(0, 2))] = "."
                sep = "|" if rng.random() < 0.3 else "/"
                gts.append(a[0] + sep + a[1] + ":%d:%d" % (rng.integers(0, 60), rng.integers(0, 99)))
            aa = "A"
            if vary_aa:                                     # ancestral allele = REF, ALT or neither, by a draw of its own
                aa = (ref, alt, "N", "T")[(si * 2654435761 >> 7) % 4]
            f.write("\t".join([scaf, str(pos), ".", ref, alt, "999", ".", "DP=10;AA=" + aa, "GT:DP:GQ"] + gts) + eol)
    return names


def emulate_shard_outputs(num, den, kk, inc, base, g):
    """What EvoFst.segments() / boundary() / included_totals() return for one shard, computed on the CPU
    from per-site oracle values [S, Q] (CPU stand-in for the engine in the gloo tests)."""
    S, Q = num.shape
    cap = S // g + 2
    seg = dict(first_segment=np.zeros(Q, np.int64), n_segments=np.zeros(Q, np.int64),
               sum_num=np.zeros((cap, Q)), sum_den=np.zeros((cap, Q)), sum_dxy_k=np.zeros((cap, Q)),
               end_site=np.zeros((cap, Q), np.int32))
    carry = np.zeros((Q, 4)); totals = np.zeros(Q, np.int64)
    for q in range(Q):
        rank = int(base[q]); first = rank // g
        seg["first_segment"][q] = first
        a = np.zeros(3); cnt = 0; row = 0
        for s in range(S):
            if not inc[s, q]:
                continue
            rank += 1; cnt += 1
            a = a + np.array([num[s, q], den[s, q], kk[s, q]])
            if rank % g == 0:
                seg["sum_num"][row, q], seg["sum_den"][row, q], seg["sum_dxy_k"][row, q] = a
                seg["end_site"][row, q] = s
                row += 1; a = np.zeros(3); cnt = 0
        seg["n_segments"][q] = row
        carry[q] = [a[0], a[1], a[2], cnt]
        totals[q] = rank - int(base[q])
    return seg, carry, totals


class CpuGlobalShard:
    """CPU stand-in for an EvoFst context in against-all mode on one shard (gloo tests): per-site values come
    from the oracle, the deque walk is restated here (evo_FstAgainstAll.cpp:198-203, 237-240, 276-280)."""

    def __init__(self, kept, num, den, flags):
        self.kept, self.num, self.den = kept, num, den
        self.flags = flags if flags is not None else np.zeros(len(kept), np.uint8)
        self.n_sites, self.n_use = num.shape
        self.base = np.zeros(2, np.int64)
        self.halo = (np.zeros(0, np.int32), np.zeros((0, self.n_use)), np.zeros((0, self.n_use)))

    def set_windows(self, size, step):
        self.size, self.step = size, step

    def run_counts(self):
        pass

    def included_totals(self):
        return np.array([int((self.flags == 0).sum()), int(self.kept.sum())], np.int64)

    def set_base_offsets(self, base):
        self.base = np.asarray(base, np.int64)

    def global_site_stats(self, first, n):
        sl = slice(first, first + n)
        return self.num[sl].T.copy(), self.den[sl].T.copy(), self.kept[sl].copy()

    def set_global_halo(self, tag, num, den):
        self.halo = (np.asarray(tag), np.asarray(num), np.asarray(den))

    def run_stats(self):
        size, step, U = self.size, self.step, self.n_use
        dq = [(-1, np.zeros(U), np.zeros(U))] * size
        tag, hn, hd = self.halo
        for e in range(len(tag)):
            dq = dq[1:] + [(int(tag[e]), hn[e], hd[e])]
        used = int(self.base[0])
        rows = dict(start_site=[], end_site=[], mean_num=[], mean_den=[])
        for s in range(self.n_sites):
            if self.flags[s]:
                continue
            used += 1
            if not self.kept[s]:
                continue
            dq = dq[1:] + [(s, self.num[s], self.den[s])]
            if (used > size or size == step) and used % step == 0:
                an = np.zeros(U); ad = np.zeros(U)
                for _, a, b in dq:
                    an = an + a; ad = ad + b
                rows["start_site"].append(dq[0][0]); rows["end_site"].append(s)
                rows["mean_num"].append(an / size); rows["mean_den"].append(ad / size)
        self.rows = {k: np.asarray(v) for k, v in rows.items()}
        if not rows["end_site"]:
            self.rows["mean_num"] = np.zeros((0, U)); self.rows["mean_den"] = np.zeros((0, U))
            self.rows["start_site"] = np.zeros(0, np.int32); self.rows["end_site"] = np.zeros(0, np.int32)

    def global_windows(self):
        return self.rows
