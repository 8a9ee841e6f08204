import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from evo_amd import EvoFst, compose_snp_windows
from oracle import oracle as O
import test_gpu_pairs as T
from helpers import n_full_of
S, N, P, miss, size, step = 3000, 30, 3, 0.0, 7, 7
e = EvoFst(0)
rng = np.random.default_rng(S + size)
sp, codes, fl = T._setup(e, rng, S, N, P, miss, flags=True)
e.set_pairs()
pi, pj = e.all_pairs(); nf = n_full_of(sp, P)
oalt, onm = O.counts_from_codes(codes, sp, P)
onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, fl)
e.set_windows(size, step); e.run()
seg = e.segments()
q = 0
ow = O.snp_windows(onum, oden, odxy, oinc, q, size, step)
gw = compose_snp_windows(seg, q, size, step, nf[pi[q]], nf[pj[q]])
np.set_printoptions(linewidth=200, precision=6)
print("nm min", onm.min(), "nan in onum", np.isnan(onum).sum())
print("gpu", gw["sum_num"][:16]); print("ora", ow["sum_num"][:16])
print("gpu end", gw["end_site"][:16]); print("ora end", ow["end_site"][:16])
print("gpu den", gw["sum_den"][:16]); print("ora den", ow["sum_den"][:16])
print("gpu k", gw["sum_dxy"][:16]); print("ora k", ow["sum_dxy"][:16])
