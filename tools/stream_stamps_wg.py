#!/usr/bin/env python3
"""Per-workgroup view of one stream-pass launch's stamps (see stream_stamps.py): is a slow chunk a slow workgroup?
Prints, for the time a chunk is in flight (first producer starts -> last producer done) and for P0's own count time,
the spread over workgroups of the per-workgroup mean, the persistence from one iteration to the next, and the means
by XCD (workgroup index mod 8)."""
import sys

import numpy as np

IT = 48
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(4096, IT, 16)
wg = np.where(a[:, 0, 0] != 0)[0]
a = a[wg].astype(np.int64)
n_it = int((a[:, :, 0] != 0).all(axis=0).sum())
lo, hi = 2, max(3, n_it - 2)
for name, (x, y) in (("in flight (P0 start -> last done)", (1, 11)), ("P0 count", (1, 2))):
    d = (a[:, lo:hi, y] - a[:, lo:hi, x]) / 100.0          # [wg][it] us
    m = d.mean(axis=1)
    print("%s: all %.1f us; per-workgroup means min %.1f  median %.1f  max %.1f; per-iteration max over workgroups %.1f (mean)" %
          (name, d.mean(), m.min(), np.median(m), m.max(), d.max(axis=0).mean()))
    r = np.corrcoef(d[:, :-1].ravel(), d[:, 1:].ravel())[0, 1]
    print("   correlation of a workgroup's value with its next iteration's: %.2f;  std within a workgroup %.1f, of the workgroup means %.1f" %
          (r, d.std(axis=1).mean(), m.std()))
    print("   by XCD (workgroup mod 8):", " ".join("%.1f" % m[(wg % 8) == k].mean() for k in range(8)))
    order = np.argsort(-m)[:8]
    print("   slowest workgroups:", " ".join("%d:%.1f" % (wg[i], m[i]) for i in order))
