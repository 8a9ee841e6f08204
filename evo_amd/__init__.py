"""evo_amd -- MI355X-native engine for evo's pairwise Fst / dxy hot path.

The product is evo_amd/libevofst.so (HIP kernels + C ABI, include/evofst.h) and the `evo`
CLI front-end (evo_amd/cli); this package is the thin ctypes binding used by tests and bench.
"""
from .api import EvoFst, EvoFstError, load_library, compose_snp_windows, codes_from_vcf_line  # noqa: F401
from .api import REF, ALT, MISSING, SITE_SKIP, MODE_FST, MODE_GLOBAL, LIB_PATH, SIGNATURES  # noqa: F401
