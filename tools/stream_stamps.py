#!/usr/bin/env python3
"""Reads the stamp file of one stream-pass launch (EVOFST_STREAM_STAMPS, -DEVOFST_EXPERIMENTS build) and prints, per
iteration, the mean duration of each phase in microseconds (stamps are the 100 MHz real-time counter)."""
import sys

import numpy as np

IT = 48
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(4096, IT, 16)
wg = np.where(a[:, 0, 0] != 0)[0]
a = a[wg].astype(np.int64)
t0 = a[:, 0, 0].min()
print("workgroups", len(wg), "launch span %.1f us" % ((a[:, :, :10].max() - t0) / 100.0))
names = ["P0 waits for buffer", "P0 count", "first done", "last done", "publish", "C: waits for chunk", "C: flags seen", "C: first sums", "C: sums->walked", "C: begin->done"]
pairs = [(0, 1), (1, 2), (1, 12), (1, 11), (11, 13), (3, 4), (4, 5), (5, 6), (6, 8), (3, 9)]
for it in range(IT):
    live = a[:, it, 0] != 0
    if not live.any():
        break
    row = ["it %2d n %4d start %7.1f" % (it, live.sum(), (np.median(a[live, it, 0]) - t0) / 100.0)]
    for (x, y), nm in zip(pairs, names):
        ok = live & (a[:, it, x] != 0) & (a[:, it, y] != 0)
        if ok.any():
            d = (a[ok, it, y] - a[ok, it, x]) / 100.0
            row.append("%s %.1f/%.1f" % (nm.split(":")[-1].strip()[:18], d.mean(), d.max()))
    pw = a[live, it, 14:16].astype(np.uint64).view(np.uint16).reshape(-1, 8) / 100.0
    if pw.any():
        row.append("producers " + " ".join("%.1f" % x for x in pw.mean(axis=0)))
    pr = a[live, it, 10]
    if (pr != 0).any():
        row.append("flag polls %.1f/%d row retries %.2f/%d" % ((pr >> 32).mean(), (pr >> 32).max(), (pr & 0xffffffff).mean(), (pr & 0xffffffff).max()))
    print(" | ".join(row))
