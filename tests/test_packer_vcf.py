"""Host packer in bulk: evofst_pack_vcf_lines parses and packs a block of VCF lines on several
threads; the tiles and site flags must equal what the per-line path (evofst_codes_from_vcf_line +
evofst_pack_codes, itself checked against the oracle's string-level rules in test_abi.py) produces.
No GPU needed: a host-only context."""
import numpy as np
import pytest

from evo_amd import EvoFst, codes_from_vcf_line
from evo_amd.api import MODE_FST, MODE_GLOBAL
from tests.helpers import write_vcf, round_robin_pops


def _lines(tmp_path, seed, S, N, crlf=False):
    rng = np.random.default_rng(seed)
    path = tmp_path / "t.vcf"
    write_vcf(str(path), rng, S, N, crlf=crlf)
    out = []
    for ln in open(path, "rb").read().split(b"\n"):
        if ln and not ln.startswith(b"#"):
            out.append(ln.rstrip(b"\r") if crlf else ln)
    return out


@pytest.mark.parametrize("mode", [MODE_FST, MODE_GLOBAL])
@pytest.mark.parametrize("S,N,P,threads", [(300, 24, 3, 1), (1000, 50, 5, 4), (130, 70, 2, 8)])
def test_bulk_packer_equals_per_line_path(tmp_path, mode, S, N, P, threads):
    lines = _lines(tmp_path, S + N, S, N)
    e = EvoFst(-1)                                    # host-only context
    rng = np.random.default_rng(1)
    # layout: a subset of the VCF's samples in shuffled order, one sample used twice (the fst sets allow it)
    vcf_idx = rng.permutation(N)[: N - 3].astype(np.int32)
    vcf_idx[-1] = vcf_idx[0]
    sp = round_robin_pops(len(vcf_idx), P)
    e.set_layout(sp, P)
    tiles, flags = e.pack_vcf_lines(lines, mode, N, vcf_idx, n_threads=threads)
    codes = np.empty((S, 2 * len(vcf_idx)), np.uint8)
    rflags = np.empty(S, np.uint8)
    for s, ln in enumerate(lines):
        cd, fl, _, _ = codes_from_vcf_line(ln.decode(), mode, N)
        c2 = cd.reshape(N, 2)[vcf_idx]
        codes[s] = c2.reshape(-1)
        rflags[s] = fl
    ref = e.pack_codes(codes)
    assert np.array_equal(flags, rflags)
    assert np.array_equal(tiles, np.ascontiguousarray(ref).view(np.uint8).reshape(-1))
    e.close()


def test_bulk_packer_rejects_malformed_lines(tmp_path):
    lines = _lines(tmp_path, 3, 80, 10)
    e = EvoFst(-1)
    e.set_layout(round_robin_pops(10, 2), 2)
    bad = list(lines)
    bad[41] = b"\t".join(bad[41].split(b"\t")[:15])          # too few genotype columns
    with pytest.raises(Exception):
        e.pack_vcf_lines(bad, MODE_FST, 10, None, n_threads=3)
    bad = list(lines)
    f = bad[7].split(b"\t"); f[12] = b"1"; bad[7] = b"\t".join(f)   # haploid genotype field
    if not (f[3] != b"A" or b"," in f[4] or len(f[4]) > 1):           # only lines whose genotypes are read
        with pytest.raises(Exception):
            e.pack_vcf_lines(bad, MODE_FST, 10, None, n_threads=2)
    e.close()


def test_packer_survives_malformed_lines():
    """Truncated and corrupted lines: every entry point that reads VCF text answers EVOFST_OK or an error code, never
    reads past the terminator (the tests run this a few thousand times on seeded mutations of valid lines)."""
    import ctypes as C
    from evo_amd import api
    rng = np.random.default_rng(99)
    n = 11
    good = ["chr1\t%d\t.\tA\t%s\t9\t.\tDP=1\tGT:DP\t" % (100 + i, alt) + "\t".join(
        "%s%s%s:%d" % (rng.choice(list("01.2")), rng.choice(list("/|")), rng.choice(list("01.2")), rng.integers(0, 99)) for _ in range(n))
        for i, alt in enumerate(["G", "G,*", "*,G", "G,T,*", "GT", "*", ""])]
    e = api.EvoFst(-1)
    e.set_layout(np.arange(n) % 3, 3)
    L = api.load_library()
    seen = set()
    for trial in range(3000):
        b = bytearray(good[trial % len(good)].encode())
        kind = trial % 4
        if kind == 0:
            b = b[:int(rng.integers(0, len(b) + 1))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(1, 256))
        elif kind == 2:
            k = int(rng.integers(0, len(b)))
            del b[k:k + int(rng.integers(1, 8))]
        line = bytes(b).replace(b"\x00", b"x")
        for mode in (0, 1, 2):
            tiles = np.zeros(e.tiles_bytes(1), np.uint8); flags = np.zeros(1, np.uint8)
            arr = (C.c_char_p * 1)(line)
            rc = L.evofst_pack_vcf_lines(e.h, mode, 1, arr, n, None, 1, tiles.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(api._u8p))
            seen.add(rc)
        al = np.zeros(2 * n, np.uint8)
        seen.add(L.evofst_alleles_from_vcf_line(line, n, al.ctypes.data_as(api._u8p)))
        d = C.c_int64(0)
        arr = (C.c_char_p * 1)(line)
        seen.add(L.evofst_stats_phase_draws(1, arr, n, 1, C.byref(d)))
    assert seen <= {0, -1} and seen == {0, -1}, seen          # EVOFST_OK and EVOFST_ERR_ARG both occur
