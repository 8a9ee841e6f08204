"""Host-only commands of the `evo` CLI (no GPU is touched): they run wherever the binary of __graft_entry__.build() is."""
import os
import subprocess

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EVO = os.path.join(ROOT, "evo_amd", "cli", "evo")


def test_stats_accessible_genome_windows(tmp_path):
    """evo stats --accessibleGenomeBED=BED [--accessibleGenBedWindow=N] LENGTHS (process_vcf_stats.cpp:146-164): one row
    per window of every scaffold in the lengths file with the accessible bases of [i, i + N], counted by the reference's
    four overlap rules (the oracle's restatement, pinned on the reference's own header in tests/test_oracle.py)."""
    assert os.path.exists(EVO), "evo_amd/cli/evo is not built: run __graft_entry__.build()"
    rng = np.random.default_rng(7)
    bed, lengths = tmp_path / "acc.bed", tmp_path / "genome.txt"
    sizes = {"chr1": 23000, "scaffold_7": 5100, "tiny": 40, "absent": 2500}
    with open(bed, "w") as f:
        for sc in ("chr1", "scaffold_7", "tiny"):
            pos = 0
            while pos < sizes[sc]:
                a = pos + int(rng.integers(0, 400)); b = a + int(rng.integers(1, 900))
                f.write("%s\t%d\t%d\n" % (sc, a, b))
                pos = b
    with open(lengths, "w") as f:
        for sc, n in sizes.items():
            f.write("%s\t%d\n" % (sc, n))
    for window in (None, 700):
        args = ["--accessibleGenomeBED=" + str(bed)] + (["--accessibleGenBedWindow=%d" % window] if window else [])
        r = subprocess.run([EVO, "stats"] + args + [str(lengths)], capture_output=True, text=True, timeout=60, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
        w = window or 10000
        want = []
        for sc, n in sizes.items():
            for i in range(0, n, w):
                want.append("%s\t%d\t%d\t%d" % (sc, i, i + w, O.bed_accessible_bp(str(bed), sc, i, i + w)))
        assert r.stdout.splitlines() == want
        assert any(not row.endswith("\t0") for row in want)
    assert sorted(os.listdir(tmp_path)) == ["acc.bed", "genome.txt"]          # nothing is written but stdout
    r = subprocess.run([EVO, "stats", str(lengths)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Which statistics" in r.stderr               # :398
