#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file): one line per kernel."""
import re
import sys

txt = (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read()
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    print("%-90s sgpr %-4s vgpr %-4s agpr %-3s spill s/v %s/%s scratch %-5s occ %s lds %s" % (
        r["name"][:90], r.get("TotalSGPRs", "?"), r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs Spill", "?"),
        r.get("VGPRs Spill", "?"), r.get("ScratchSize", "?"), r.get("Occupancy", "?"), r.get("LDS Size", "?")))
