"""ctypes binding of include/evofst.h (evo_amd/libevofst.so).

This is plumbing: every statistic is computed by the HIP kernels behind the C ABI.  There is no
CPU fallback -- loading fails loudly when the shared library is missing and `EvoFst()` raises
when no GPU is available.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVOFST_LIBRARY") or os.path.join(_HERE, "libevofst.so")   # EVOFST_LIBRARY: a developer build to test against

EVOFST_OK = 0
ERR_ARG, ERR_HIP, ERR_STATE, ERR_NOMEM, ERR_UNSUPPORTED = -1, -2, -3, -4, -5
REF, ALT, MISSING = 0, 1, 2
SITE_SKIP = 1
MODE_FST, MODE_GLOBAL = 0, 1
MODE_GLOBAL_INDELS = 3      # MODE_GLOBAL with -i of evo PBS / DistOutgroups: only ALT == "*" is skipped

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p


class Geometry(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("n_pops", C.c_int32), ("n_haplotypes", C.c_int32),
                ("words_per_site", C.c_int32), ("tile_sites", C.c_int32), ("count_bytes", C.c_int32),
                ("tile_bytes", C.c_int64)]


# every symbol include/evofst.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "evofst_create": (_vp, [C.c_int]),
    "evofst_destroy": (None, [_vp]),
    "evofst_last_error": (C.c_char_p, [_vp]),
    "evofst_version": (C.c_char_p, []),
    "evofst_set_stream": (C.c_int, [_vp, _vp]),
    "evofst_set_layout": (C.c_int, [_vp, C.c_int, C.c_int, _i32p]),
    "evofst_get_geometry": (C.c_int, [_vp, C.POINTER(Geometry)]),
    "evofst_get_pop_chroms": (C.c_int, [_vp, _i32p]),
    "evofst_tiles_bytes": (C.c_size_t, [_vp, C.c_int64]),
    "evofst_pack_codes": (C.c_int, [_vp, C.c_int64, C.c_int64, _u8p, _vp]),
    "evofst_unpack_codes": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _u8p]),
    "evofst_codes_from_vcf_line": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _u8p, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "evofst_pack_vcf_lines": (C.c_int, [_vp, C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.c_int, _i32p, C.c_int, _vp, _u8p]),
    "evofst_reserve": (C.c_int, [_vp, C.c_int64]),
    "evofst_alloc_pinned": (_vp, [_vp, C.c_size_t]),
    "evofst_free_pinned": (None, [_vp, _vp]),
    "evofst_upload": (C.c_int, [_vp, C.c_int64, _vp, _u8p, _i32p]),
    "evofst_attach_device": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "evofst_synth": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_uint64, C.c_double]),
    "evofst_run_pairs_noalloc": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "evofst_noalloc_scratch_bytes": (C.c_int64, []),
    "evofst_download_tiles": (C.c_int, [_vp, _vp]),
    "evofst_download_tile_range": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "evofst_set_pairs": (C.c_int, [_vp, _i32p, _i32p, C.c_int]),
    "evofst_set_against_all": (C.c_int, [_vp, _i32p, C.c_int, _i64p]),
    "evofst_set_base_offsets_device": (C.c_int, [_vp, _vp]),
    "evofst_set_base_offsets_gathered": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "evofst_copy_included_totals": (C.c_int, [_vp, _vp]),
    "evofst_copy_boundary": (C.c_int, [_vp, _vp]),
    "evofst_set_global_halo": (C.c_int, [_vp, C.c_int, _i32p, _dp, _dp]),
    "evofst_set_gram": (C.c_int, [_vp]),
    "evofst_set_site_windows": (C.c_int, [_vp, _i64p, C.c_int64]),
    "evofst_run_gram": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "evofst_get_gram": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "evofst_set_windows": (C.c_int, [_vp, C.c_int, C.c_int]),
    "evofst_set_base_offsets": (C.c_int, [_vp, _i64p]),
    "evofst_run_counts": (C.c_int, [_vp]),
    "evofst_run_stats": (C.c_int, [_vp]),
    "evofst_run": (C.c_int, [_vp]),
    "evofst_sync": (C.c_int, [_vp]),
    "evofst_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "evofst_last_pairs_pass": (C.c_int, [_vp]),
    "evofst_device_count": (C.c_int, []),
    "evofst_get_positions": (C.c_int, [_vp, C.c_int64, C.c_int64, _i32p]),
    "evofst_last_global_walk": (C.c_int, [_vp]),
    "evofst_last_gram_streams": (C.c_int, [_vp]),
    "evofst_get_pair_range_sums": (C.c_int, [_vp, C.c_int, C.c_int64, _i64p, _i64p, _dp, _dp]),
    "evofst_set_profiling": (C.c_int, [_vp, C.c_int]),
    "evofst_get_kernel_ms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "evofst_get_counts": (C.c_int, [_vp, C.c_int64, C.c_int64, _i32p, _i32p]),
    "evofst_get_included_totals": (C.c_int, [_vp, _i64p]),
    "evofst_get_site_stats": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp, _dp, _dp, _u8p]),
    "evofst_get_pbs_site_stats": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int, _i32p, _dp, _u8p]),
    "evofst_get_abs_site_stats": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int, _i32p, _dp, _u8p]),
    "evofst_alleles_from_vcf_line": (C.c_int, [C.c_char_p, C.c_int, _u8p]),
    "evofst_stats_phase_draws": (C.c_int, [C.c_int64, C.POINTER(C.c_char_p), C.c_int, C.c_int, _i64p]),
    "evofst_set_multiallelic": (C.c_int, [_vp, C.c_int64, _i64p, _u8p]),
    "evofst_get_indiv_diffs": (C.c_int, [_vp, C.c_int, _i64p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "evofst_get_indiv_hets": (C.c_int, [_vp, _i32p, _i32p]),
    "evofst_get_dist_windows": (C.c_int, [_vp, C.c_int, _i64p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _dp, _i32p, _i32p,
                                          _i32p, _i32p]),
    "evofst_get_pop_site_div": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp, _dp, _dp]),
    "evofst_max_segments": (C.c_int64, [_vp]),
    "evofst_get_segments": (C.c_int, [_vp, _i64p, _i64p, _dp, _dp, _dp, _i32p]),
    "evofst_get_boundary": (C.c_int, [_vp, _dp]),
    "evofst_get_totals": (C.c_int, [_vp, _dp]),
    "evofst_get_global_site_stats": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp, _dp, _u8p]),
    "evofst_max_global_rows": (C.c_int64, [_vp]),
    "evofst_get_global_windows": (C.c_int, [_vp, _i64p, _i32p, _i32p, _dp, _dp]),
}

_lib = None


def load_library():
    """Loads libevofst.so (built by __graft_entry__.build() / evo_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("evo_amd: %s is missing -- build it with `make -C evo_amd/csrc` "
                               "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(os.environ.get("EVOFST_LIB", LIB_PATH))   # tools/ point this at the -DEVOFST_EXPERIMENTS build
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export the symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class EvoFstError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("evofst error %d: %s" % (code, msg))
        self.code = code


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def codes_from_vcf_line(line, mode, n_samples):
    """One VCF data line -> (codes[2N], site_flags, chrom, pos) with the rules of `mode`."""
    L = load_library()
    buf = C.create_string_buffer(line.encode() if isinstance(line, str) else line)
    codes = np.empty(2 * n_samples, np.uint8)
    fl = C.c_uint32(0)
    chrom = C.c_char_p(); pos = C.c_char_p()
    rc = L.evofst_codes_from_vcf_line(buf, mode, n_samples, _ptr(codes, _u8p), C.byref(fl), C.byref(chrom),
                                      C.byref(pos))
    if rc != 0:
        raise EvoFstError(rc, "malformed VCF line")
    return codes, fl.value, chrom.value.decode(), pos.value.decode()


def alleles_from_vcf_line(line, n_samples):
    """Allele numbers of one (multi-allelic) VCF data line as `evo stats` reads them -> uint8 [2N], 255 = missing."""
    L = load_library()
    out = np.empty(2 * n_samples, np.uint8)
    rc = L.evofst_alleles_from_vcf_line(line.encode() if isinstance(line, str) else line, n_samples, _ptr(out, _u8p))
    if rc != 0:
        raise EvoFstError(rc, "malformed VCF line")
    return out


class EvoFst:
    """One context = one GPU.  Mirrors the C ABI one to one; results come back as numpy arrays."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = self.L.evofst_create(device)
        if not self.h:
            raise EvoFstError(ERR_HIP, self.L.evofst_last_error(None).decode())
        self.n_pairs = 0
        self.n_use = 0
        self.n_pops = 0
        self.n_samples = 0
        self.n_sites = 0

    def close(self):
        if self.h:
            self.L.evofst_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EvoFstError(rc, self.L.evofst_last_error(self.h).decode())

    # ---- layout / packing
    def set_layout(self, sample_pop, n_pops):
        sp = np.ascontiguousarray(sample_pop, np.int32)
        self._chk(self.L.evofst_set_layout(self.h, len(sp), n_pops, _ptr(sp, _i32p)))
        self.n_samples, self.n_pops = len(sp), n_pops
        self.n_pairs = self.n_use = 0

    def geometry(self):
        g = Geometry()
        self._chk(self.L.evofst_get_geometry(self.h, C.byref(g)))
        return g

    def pop_chroms(self):
        out = np.empty(self.n_pops, np.int32)
        self._chk(self.L.evofst_get_pop_chroms(self.h, _ptr(out, _i32p)))
        return out

    def tiles_bytes(self, n_sites):
        return self.L.evofst_tiles_bytes(self.h, n_sites)

    def pack_codes(self, codes):
        codes = np.ascontiguousarray(codes, np.uint8)
        S = codes.shape[0]
        assert codes.shape[1] == 2 * self.n_samples
        tiles = np.zeros(self.tiles_bytes(S) // 8, np.uint64)
        self._chk(self.L.evofst_pack_codes(self.h, 0, S, _ptr(codes, _u8p), tiles.ctypes.data_as(_vp)))
        return tiles

    def pack_vcf_lines(self, lines, mode, n_vcf_samples, layout_to_vcf=None, n_threads=1):
        """Parse + pack VCF data lines (bytes objects) into tiles on n_threads host threads -> (tiles, site_flags)."""
        n = len(lines)
        arr = (C.c_char_p * n)(*lines)
        tiles = np.zeros(self.tiles_bytes(n), np.uint8)
        flags = np.zeros(n, np.uint8)
        l2v = None if layout_to_vcf is None else np.ascontiguousarray(layout_to_vcf, np.int32)
        self._chk(self.L.evofst_pack_vcf_lines(self.h, mode, n, arr, n_vcf_samples, _ptr(l2v, _i32p) if l2v is not None else None,
                                               n_threads, tiles.ctypes.data_as(_vp), _ptr(flags, _u8p)))
        return tiles, flags

    def unpack_codes(self, tiles, n_sites):
        codes = np.empty((n_sites, 2 * self.n_samples), np.uint8)
        self._chk(self.L.evofst_unpack_codes(self.h, 0, n_sites, tiles.ctypes.data_as(_vp), _ptr(codes, _u8p)))
        return codes

    # ---- data
    def reserve(self, max_sites):
        self._chk(self.L.evofst_reserve(self.h, max_sites))

    def upload(self, tiles, n_sites, site_flags=None, pos=None):
        self._tiles_keep = np.ascontiguousarray(tiles)
        self._fl_keep = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
        self._pos_keep = np.ascontiguousarray(pos, np.int32) if pos is not None else None
        self._chk(self.L.evofst_upload(self.h, n_sites, self._tiles_keep.ctypes.data_as(_vp),
                                       _ptr(self._fl_keep, _u8p), _ptr(self._pos_keep, _i32p)))
        self.sync()
        self.n_sites = n_sites

    def attach_device(self, n_sites, d_tiles, d_flags=None, d_pos=None):
        self._chk(self.L.evofst_attach_device(self.h, n_sites, d_tiles, d_flags, d_pos))
        self.n_sites = n_sites

    def synth(self, n_sites, seed=20261110, missing_rate=0.0, first_site_global=0):
        self._chk(self.L.evofst_synth(self.h, n_sites, first_site_global, seed, missing_rate))
        self.n_sites = n_sites

    def download_tiles(self, first_site=0, n_sites=None):
        """Tiles of the shard (or of the 64-site-aligned range [first_site, first_site + n_sites)) as uint64 words."""
        if first_site == 0 and n_sites is None:
            tiles = np.empty(self.tiles_bytes(self.n_sites) // 8, np.uint64)
            self._chk(self.L.evofst_download_tiles(self.h, tiles.ctypes.data_as(_vp)))
            return tiles
        assert first_site % 64 == 0
        n_sites = self.n_sites - first_site if n_sites is None else n_sites
        nt = (n_sites + 63) // 64
        tiles = np.empty(self.tiles_bytes(nt * 64) // 8, np.uint64)
        self._chk(self.L.evofst_download_tile_range(self.h, first_site // 64, nt, tiles.ctypes.data_as(_vp)))
        return tiles

    # ---- modes
    def set_pairs(self, pair_i=None, pair_j=None):
        if pair_i is None:
            self._chk(self.L.evofst_set_pairs(self.h, None, None, 0))
            self.n_pairs = self.n_pops * (self.n_pops - 1) // 2
        else:
            pi = np.ascontiguousarray(pair_i, np.int32); pj = np.ascontiguousarray(pair_j, np.int32)
            self._chk(self.L.evofst_set_pairs(self.h, _ptr(pi, _i32p), _ptr(pj, _i32p), len(pi)))
            self.n_pairs = len(pi)
        self.n_use = 0

    def all_pairs(self):
        P = self.n_pops
        i, j = np.triu_indices(P, 1)
        return i.astype(np.int32), j.astype(np.int32)

    def set_against_all(self, use_pops, set_sizes):
        up = np.ascontiguousarray(use_pops, np.int32); ss = np.ascontiguousarray(set_sizes, np.int64)
        self._chk(self.L.evofst_set_against_all(self.h, _ptr(up, _i32p), len(up), _ptr(ss, _i64p)))
        self.n_use = len(up)
        self.n_pairs = 0

    # device-side exchange (pointers to device memory of the caller, e.g. torch tensors; asynchronous)
    def set_stream(self, hip_stream):
        self._chk(self.L.evofst_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_base_offsets_device(self, d_ptr):
        self._chk(self.L.evofst_set_base_offsets_device(self.h, C.c_void_p(d_ptr)))

    def set_base_offsets_gathered(self, d_gathered, row_stride, n_rows):
        """Base offsets = sum of the first n_rows rows (row_stride int64 values each, totals first) of an all-gather
        buffer in device memory."""
        self._chk(self.L.evofst_set_base_offsets_gathered(self.h, d_gathered, row_stride, n_rows))

    def copy_included_totals(self, d_ptr):
        self._chk(self.L.evofst_copy_included_totals(self.h, C.c_void_p(d_ptr)))

    def copy_boundary(self, d_ptr):
        self._chk(self.L.evofst_copy_boundary(self.h, C.c_void_p(d_ptr)))

    def set_global_halo(self, tag, num, den):
        """Against-all mode, multi-GPU: the kept sites (<= SIZE, oldest first) that precede this shard."""
        tag = np.ascontiguousarray(tag, np.int32)
        num = np.ascontiguousarray(num, np.float64); den = np.ascontiguousarray(den, np.float64)
        self._chk(self.L.evofst_set_global_halo(self.h, len(tag), _ptr(tag, _i32p), _ptr(num, _dp), _ptr(den, _dp)))

    def set_gram(self):
        """All pairs i<j through the Gram path (windows = site ranges, see set_site_windows)."""
        self._chk(self.L.evofst_set_gram(self.h))
        P = self.geometry().n_pops
        self.n_pairs = P * (P - 1) // 2
        self.n_use = 0

    def set_site_windows(self, win_start):
        ws = np.ascontiguousarray(win_start, np.int64)
        self._chk(self.L.evofst_set_site_windows(self.h, _ptr(ws, _i64p), len(ws) - 1))
        self.n_site_windows = len(ws) - 1

    def run_gram(self, first_window=0, n_windows=None):
        if n_windows is None:
            n_windows = self.n_site_windows - first_window
        self._chk(self.L.evofst_run_gram(self.h, first_window, n_windows))
        self._gram_n = n_windows

    def gram(self):
        """dict of [n_windows][n_pairs] arrays of the last run_gram: sum_num, sum_den, sum_dxy_k, n_included."""
        shape = (self._gram_n, self.n_pairs)
        out = {k: np.empty(shape, np.float64) for k in ("sum_num", "sum_den", "sum_dxy_k", "n_included")}
        self._chk(self.L.evofst_get_gram(self.h, _ptr(out["sum_num"], _dp), _ptr(out["sum_den"], _dp),
                                         _ptr(out["sum_dxy_k"], _dp), _ptr(out["n_included"], _dp)))
        return out

    def set_windows(self, size, step):
        self._chk(self.L.evofst_set_windows(self.h, size, step))
        self.size, self.step = size, step

    def set_base_offsets(self, offsets):
        o = np.ascontiguousarray(offsets, np.int64)
        self._chk(self.L.evofst_set_base_offsets(self.h, _ptr(o, _i64p)))

    # ---- compute
    def run_counts(self):
        self._chk(self.L.evofst_run_counts(self.h))

    def run_stats(self):
        self._chk(self.L.evofst_run_stats(self.h))

    def run(self):
        self._chk(self.L.evofst_run(self.h))

    def sync(self):
        self._chk(self.L.evofst_sync(self.h))

    def set_option(self, name, value):
        """Tuning / test switches of this context (include/evofst.h: evofst_set_option)."""
        self._chk(self.L.evofst_set_option(self.h, name.encode(), int(value)))

    def positions(self, first=0, n=None):
        """Positions of the shard's sites (evofst_get_positions)."""
        n = self.n_sites - first if n is None else n
        pos = np.empty(n, np.int32)
        self._chk(self.L.evofst_get_positions(self.h, first, n, _ptr(pos, _i32p)))
        return pos

    def last_global_walk(self):
        """1 = global_window_kernel, 2 = global_tile_kernel (the last against-all run's; 0: none)."""
        return int(self.L.evofst_last_global_walk(self.h))

    def last_gram_streams(self):
        """FP64 contractions the last run_gram spent on the denominators: 1 (centred frequencies) or 2 (0: none yet)."""
        return int(self.L.evofst_last_gram_streams(self.h))

    def last_pairs_pass(self):
        """1 = separate kernels, 2 = stream pass, 3 = lock-step one-kernel pass (the last run's; 0: none)."""
        return int(self.L.evofst_last_pairs_pass(self.h))

    def set_profiling(self, on=True):
        self._chk(self.L.evofst_set_profiling(self.h, 1 if on else 0))

    def kernel_ms(self, which):
        ms = C.c_float(0)
        self._chk(self.L.evofst_get_kernel_ms(self.h, which, C.byref(ms)))
        return ms.value

    # ---- results
    def counts(self, first=0, n=None):
        n = self.n_sites - first if n is None else n
        alt = np.empty((n, self.n_pops), np.int32); nm = np.empty((n, self.n_pops), np.int32)
        self._chk(self.L.evofst_get_counts(self.h, first, n, _ptr(alt, _i32p), _ptr(nm, _i32p)))
        return alt, nm

    def included_totals(self):
        out = np.empty(self.n_pairs if self.n_pairs else 2, np.int64)
        self._chk(self.L.evofst_get_included_totals(self.h, _ptr(out, _i64p)))
        return out

    def site_stats(self, first=0, n=None):
        """-> num, den, dxy [n_pairs, n], include [n_pairs, n]"""
        n = self.n_sites - first if n is None else n
        Q = self.n_pairs
        num = np.empty((Q, n)); den = np.empty((Q, n)); dxy = np.empty((Q, n)); inc = np.empty((Q, n), np.uint8)
        self._chk(self.L.evofst_get_site_stats(self.h, first, n, _ptr(num, _dp), _ptr(den, _dp), _ptr(dxy, _dp),
                                               _ptr(inc, _u8p)))
        return num, den, dxy, inc

    def pbs_site_stats(self, trios, first=0, n=None):
        """PBS per site: (pbs [T, 3, n], used [T, n]) for trios [T, 3] of population indices."""
        n = self.n_sites - first if n is None else n
        tr = np.ascontiguousarray(trios, np.int32).reshape(-1, 3)
        pbs = np.empty((len(tr), 3, n)); used = np.empty((len(tr), n), np.uint8)
        self._chk(self.L.evofst_get_pbs_site_stats(self.h, first, n, len(tr), _ptr(tr, _i32p), _ptr(pbs, _dp), _ptr(used, _u8p)))
        return pbs, used

    def abs_site_stats(self, quartets, first=0, n=None):
        """ABS per site: (abs [T, 2, n] = {ABS, ABSalt}, used [T, n]) for quartets [T, 4] of population indices W, X, Y, Z."""
        n = self.n_sites - first if n is None else n
        qt = np.ascontiguousarray(quartets, np.int32).reshape(-1, 4)
        out = np.empty((len(qt), 2, n)); used = np.empty((len(qt), n), np.uint8)
        self._chk(self.L.evofst_get_abs_site_stats(self.h, first, n, len(qt), _ptr(qt, _i32p), _ptr(out, _dp), _ptr(used, _u8p)))
        return out, used

    def dist_windows(self, win_off, out_idx, in_idx, carry=None):
        """Outgroup-to-ingroup dxy per window [win_off[w], win_off[w+1]) -> dict(sum, nsnp, miss [W, O, I], used, missing
        [W, O]); carry = the same dict for one window (shapes [O, I] / [O]) whose running values window 0 continues."""
        wo = np.ascontiguousarray(win_off, np.int64)
        oi = np.ascontiguousarray(out_idx, np.int32); ii = np.ascontiguousarray(in_idx, np.int32)
        W, O, I = len(wo) - 1, len(oi), len(ii)
        r = dict(sum=np.zeros((W, O, I)), nsnp=np.zeros((W, O, I), np.int32), miss=np.zeros((W, O, I), np.int32),
                 used=np.zeros((W, O), np.int32), missing=np.zeros((W, O), np.int32))
        if carry is not None:
            for k in r:
                r[k][0] = carry[k]
        self._chk(self.L.evofst_get_dist_windows(self.h, W, _ptr(wo, _i64p), O, _ptr(oi, _i32p), I, _ptr(ii, _i32p),
                                                 0 if carry is None else 1, _ptr(r["sum"], _dp), _ptr(r["nsnp"], _i32p),
                                                 _ptr(r["miss"], _i32p), _ptr(r["used"], _i32p), _ptr(r["missing"], _i32p)))
        return r

    def set_multiallelic(self, site_index, alleles):
        """Multi-allelic sites of the uploaded shard: ascending site indices, alleles uint8 [n, 2 * n_samples] (255 = missing)."""
        si = np.ascontiguousarray(site_index, np.int64)
        al = np.ascontiguousarray(alleles, np.uint8).reshape(len(si), -1)
        self._chk(self.L.evofst_set_multiallelic(self.h, len(si), _ptr(si, _i64p), _ptr(al, _u8p)))

    def indiv_diffs(self, block_off):
        """Five counts per block and pair j <= i of layout individuals -> dict(hh, hd, oh, ms, d4) int32 [B, n(n+1)/2]."""
        bo = np.ascontiguousarray(block_off, np.int64)
        B = len(bo) - 1
        n = self.geometry().n_haplotypes // 2
        r = {k: np.zeros((B, n * (n + 1) // 2), np.int32) for k in ("hh", "hd", "oh", "ms", "d4")}
        self._chk(self.L.evofst_get_indiv_diffs(self.h, B, _ptr(bo, _i64p), *[_ptr(r[k], _i32p) for k in ("hh", "hd", "oh", "ms", "d4")]))
        return r

    def indiv_hets(self):
        """Per layout individual: heterozygous sites among the sites without a missing genotype, and those of them where
        ALT was seen more than once -> (het, shared) int32 [n]."""
        n = self.geometry().n_haplotypes // 2
        het = np.zeros(n, np.int32); shared = np.zeros(n, np.int32)
        self._chk(self.L.evofst_get_indiv_hets(self.h, _ptr(het, _i32p), _ptr(shared, _i32p)))
        return het, shared

    def pop_site_div(self, first=0, n=None):
        n = self.n_sites - first if n is None else n
        P = self.n_pops
        het = np.empty((P, n)); hn = np.empty((P, n)); pi = np.empty((P, n))
        self._chk(self.L.evofst_get_pop_site_div(self.h, first, n, _ptr(het, _dp), _ptr(hn, _dp), _ptr(pi, _dp)))
        return het, hn, pi

    def pair_range_sums(self, pair, lo, hi, init=None):
        """Sums of one pair's per-site values over site ranges [lo, hi), added in site order from `init` (include/evofst.h:
        evofst_get_pair_range_sums).  Returns [n_ranges][12]."""
        lo = np.ascontiguousarray(lo, np.int64); hi = np.ascontiguousarray(hi, np.int64)
        out = np.empty((len(lo), 12))
        if init is not None:
            init = np.ascontiguousarray(init, np.float64)
            assert init.shape == out.shape
        self._chk(self.L.evofst_get_pair_range_sums(self.h, int(pair), len(lo), _ptr(lo, _i64p), _ptr(hi, _i64p), _ptr(init, _dp), _ptr(out, _dp)))
        return out

    def segments(self):
        cap = self.L.evofst_max_segments(self.h)
        Q = self.n_pairs
        first = np.empty(Q, np.int64); rows = np.empty(Q, np.int64)
        sn = np.empty((cap, Q)); sd = np.empty((cap, Q)); sk = np.empty((cap, Q)); es = np.empty((cap, Q), np.int32)
        self._chk(self.L.evofst_get_segments(self.h, _ptr(first, _i64p), _ptr(rows, _i64p), _ptr(sn, _dp),
                                             _ptr(sd, _dp), _ptr(sk, _dp), _ptr(es, _i32p)))
        return dict(first_segment=first, n_segments=rows, sum_num=sn, sum_den=sd, sum_dxy_k=sk, end_site=es)

    def boundary(self):
        out = np.empty((self.n_pairs, 4))
        self._chk(self.L.evofst_get_boundary(self.h, _ptr(out, _dp)))
        return out

    def totals(self):
        out = np.empty((self.n_pairs, 4))
        self._chk(self.L.evofst_get_totals(self.h, _ptr(out, _dp)))
        return out

    def global_site_stats(self, first=0, n=None):
        n = self.n_sites - first if n is None else n
        U = self.n_use
        num = np.empty((U, n)); den = np.empty((U, n)); kept = np.empty(n, np.uint8)
        self._chk(self.L.evofst_get_global_site_stats(self.h, first, n, _ptr(num, _dp), _ptr(den, _dp),
                                                      _ptr(kept, _u8p)))
        return num, den, kept

    def global_windows(self):
        cap = self.L.evofst_max_global_rows(self.h)
        U = self.n_use
        n = C.c_int64(0)
        ss = np.empty(cap, np.int32); es = np.empty(cap, np.int32)
        mn = np.empty((cap, U)); md = np.empty((cap, U))
        self._chk(self.L.evofst_get_global_windows(self.h, C.byref(n), _ptr(ss, _i32p), _ptr(es, _i32p),
                                                   _ptr(mn, _dp), _ptr(md, _dp)))
        r = n.value
        return dict(start_site=ss[:r], end_site=es[:r], mean_num=mn[:r], mean_den=md[:r])


# ---------------------------------------------------------------------------
# host-side composition of windows from device segment sums
# ---------------------------------------------------------------------------

def compose_snp_windows(seg, pair, size, step, n_full_i, n_full_j, carry_in=None):
    """Windows of process_vcf_fst.cpp:638-650 for one pair from the device's segment sums.

    A window ends at included-site rank e (a multiple of `step`, e >= size) and is the sum of the
    size/g segments that precede it.  Returns first_var, end_site, sum_num, sum_den, sum_dxy, fst,
    mean_dxy like the oracle's snp_windows().  carry_in = (num, den, k) to add to the first
    segment of a shard (multi-GPU).
    """
    g = math.gcd(size, step)
    first = int(seg["first_segment"][pair]); rows = int(seg["n_segments"][pair])
    sn = seg["sum_num"][:rows, pair].copy(); sd = seg["sum_den"][:rows, pair].copy()
    sk = seg["sum_dxy_k"][:rows, pair].copy(); es = seg["end_site"][:rows, pair]
    if carry_in is not None and rows > 0:
        sn[0] = carry_in[0] + sn[0]; sd[0] = carry_in[1] + sd[0]; sk[0] = carry_in[2] + sk[0]
    per = size // g
    out = dict(first_var=[], end_site=[], sum_num=[], sum_den=[], sum_dxy=[], fst=[], mean_dxy=[])
    denom = float(n_full_i) * float(n_full_j)
    for r in range(rows):
        m = first + r                 # global segment index; segment ends at rank (m+1)*g
        e = (m + 1) * g
        if e % step != 0 or e < size:
            continue
        lo = r - per + 1
        if lo < 0:
            continue                  # window starts in a previous shard
        wn = 0.0; wd = 0.0; wk = 0.0
        for t in range(lo, r + 1):
            wn += sn[t]; wd += sd[t]; wk += sk[t]
        with np.errstate(all="ignore"):
            f = float(np.float64(wn / size) / np.float64(wd / size))   # calculateFst, fst.cpp:354-360
        if f < 0:
            f = 0.0
        out["first_var"].append(e - size + 1); out["end_site"].append(int(es[r]))
        out["sum_num"].append(wn); out["sum_den"].append(wd); out["sum_dxy"].append(wk / denom)
        out["fst"].append(f); out["mean_dxy"].append(wk / denom / size)
    return {k: np.asarray(v) for k, v in out.items()}
