"""GPU, end to end: the `evo` CLI front-end (evo_amd/cli/evo -> libevofst.so -> HIP kernels) writes the
same bytes as the oracle's restatement of `evo fst` / `evo FstGlobal` / `evo alleleFreq` / `evo PBS` on small seeded VCFs.

The expected files come from the CPU oracle (oracle/evo_oracle.c text drivers); beyond the behaviours
recorded in tests/golden/survey_known_answers.json the text level is "parity unpinned" (the reference
binary cannot be built in this image)."""
import filecmp
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import write_vcf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EVO = os.path.join(ROOT, "evo_amd", "cli", "evo")


def _same_files(a_dir, b_dir):
    names = sorted(os.listdir(b_dir))
    assert names, "oracle wrote nothing"
    for n in names:
        pa, pb = os.path.join(a_dir, n), os.path.join(b_dir, n)
        assert os.path.exists(pa), "CLI did not write %s" % n
        if not filecmp.cmp(pa, pb, shallow=False):
            la, lb = open(pa).read().splitlines(), open(pb).read().splitlines()
            for i, (x, y) in enumerate(zip(la, lb)):
                assert x == y, "%s line %d:\n cli    %r\n oracle %r" % (n, i + 1, x, y)
            assert len(la) == len(lb), "%s: %d vs %d lines" % (n, len(la), len(lb))


def _fst_case(tmp_path, rng, args, size, step, n_sites=900, n_samples=24, scaffolds=("scaffold_0", "chrB"),
              crlf=False, gz=False, overlap=False, samples_file=False, phys=10000, reg_above=0.0, bed_rows=None, annot_text=None, anc=False):
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, n_sites, n_samples, scaffolds=scaffolds, crlf=crlf, miss=0.08, vary_aa=anc)
    if gz:
        with open(vcf, "rb") as f, gzip.open(vcf + ".gz", "wb") as g:
            shutil.copyfileobj(f, g)
        vcf += ".gz"
    half = n_samples // 2
    s1, s2 = names[:half], names[half:]
    anc_sets = None
    if anc:                                # the last samples of each half form the two ancestral sets (one also sits in set 2)
        anc_sets = str(tmp_path / "anc.txt")
        open(anc_sets, "w").write(",".join(s1[-3:]) + "\n" + ",".join(s2[-2:] + ["NOT_THERE"]) + "\n")
        s1, s2 = s1[:-3], s2[:-1]
    if overlap:
        s2 = s2 + names[:2]           # samples that sit in both sets
        s1 = s1 + ["NOT_THERE"]       # and a name the VCF does not have
    sets = str(tmp_path / "sets.txt")
    open(sets, "w").write(",".join(s1) + "\n" + ",".join(s2) + "\n")
    sf = None
    if samples_file:
        sf = str(tmp_path / "samples.txt")
        open(sf, "w").write("".join(n + "\n" for n in reversed(names)))
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    cmd = [EVO, "fst", "--vcf=" + vcf, "--sets=" + sets, "-n", "run"] + args
    if sf:
        cmd += ["-s", sf]
    bed = None
    if bed_rows is not None:
        bed = str(tmp_path / "accessible.bed")
        open(bed, "w").write("".join("%s\t%d\t%d\n" % row for row in bed_rows))
        cmd += ["--accessibleGenomeBED=" + bed]
    if anc_sets:
        cmd += ["--ancSets=" + anc_sets]
    annot = None
    if annot_text is not None:
        annot = str(tmp_path / "genes.gffExtract")
        open(annot, "w").write(annot_text)
        cmd += ["--annot=" + annot]
    r = subprocess.run(cmd, cwd=str(cli_dir), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    O.fst_run(vcf, sets, str(orc_dir), run_name="run", size=size, step=step, phys=phys, reg_above=reg_above, samples=sf,
              bed=bed, annot=annot, anc_sets=anc_sets)
    _same_files(str(cli_dir), str(orc_dir))
    return r


def test_cli_fst_per_site(tmp_path):
    _fst_case(tmp_path, np.random.default_rng(1), ["-w", "1,1"], 1, 1)


def test_cli_fst_equal_windows_multi_scaffold(tmp_path):
    r = _fst_case(tmp_path, np.random.default_rng(2), ["-w", "50,50", "--physicalWindowSize=5000"], 50, 50, phys=5000)
    assert "Fst:" in r.stderr


def test_cli_fst_first_window_swallowed_on_other_scaffold(tmp_path):
    _fst_case(tmp_path, np.random.default_rng(3), ["-w", "40,40"], 40, 40, scaffolds=("chr7", "chr8"))


def test_cli_fst_overlapping_windows(tmp_path):
    _fst_case(tmp_path, np.random.default_rng(4), ["--window=40,20"], 40, 20)


def test_cli_fst_no_windows_gz_crlf(tmp_path):
    _fst_case(tmp_path, np.random.default_rng(5), [], 0, 0, gz=True, crlf=True)


def test_cli_fst_sets_overlap_samples_file_regions_above(tmp_path):
    _fst_case(tmp_path, np.random.default_rng(6), ["-w", "30,30", "--regions-above=0.05"], 30, 30, overlap=True,
              samples_file=True, reg_above=0.05, scaffolds=("scaffold_0",))


def test_cli_fst_accessible_genome_bed(tmp_path):
    """--accessibleGenomeBED (evo_Fst.cpp:498-505 through getAccessibleBPinRegion): intervals that start and end
    inside windows, one that spans several windows, a stretch of windows with no accessible base, touching
    intervals, and a scaffold the BED does not name."""
    rows = [("scaffold_0", 0, 1200), ("scaffold_0", 1200, 1900), ("scaffold_0", 4100, 4990), ("scaffold_0", 4995, 5005),
            ("scaffold_0", 9000, 23017), ("scaffold_0", 41000, 41001), ("scaffold_0", 44999, 45000),
            ("scaffold_0", 45000, 45001), ("scaffold_0", 52000, 90000), ("other", 0, 100000)]
    _fst_case(tmp_path, np.random.default_rng(41), ["-w", "50,50", "--physicalWindowSize=5000"], 50, 50, phys=5000,
              bed_rows=rows)


def test_cli_fst_accessible_genome_bed_no_snp_windows(tmp_path):
    """Fixed windows that hold no SNP take their accessible length from the BED too (evo_Fst.cpp:486-497)."""
    rows = [("chr7", 100, 2500), ("chr7", 3000, 3100), ("chr7", 7990, 8010), ("chr8", 500, 61000)]
    _fst_case(tmp_path, np.random.default_rng(42), ["-w", "40,40", "--physicalWindowSize=1000"], 40, 40, phys=1000,
              scaffolds=("chr7", "chr8"), n_sites=300, bed_rows=rows)


def test_cli_fst_annot_snp_categories(tmp_path):
    """--annot (process_vcf_fst.cpp:558-562): runSNPcategory_fst.txt with the category of every included SNP.  The
    annotation has transcripts on both strands, a partial transcript, overlapping transcripts, positions of three to five
    digits (the reference compares them as strings) and a last scaffold, which the reference's loader never stores."""
    rows = []
    for k in range(12):                                   # scaffold_0: two-exon transcripts every 5 kb, alternating strands
        a = 900 + 5000 * k
        ex = [(a, a + 400), (a + 1500, a + 2100)]
        if k % 2:
            ex.reverse()
        name = "gene.%d.1.1.1" % k if k % 3 else "g%d" % k
        if k == 7:
            name = "5partial-" + name
        rows += ["scaffold_0\t%d\t%d\t%s\t%s" % (s, e, "-" if k % 2 else "+", name) for s, e in ex]
    rows += ["scaffold_0\t20000\t30000\t+\tlong.1", "scaffold_0\t95\t99\t+\ttiny"]
    rows += ["chrB\t100\t60000\t+\twhole", "chrB\t200\t300\t+\twhole", "chrZ\t1\t10\t+\tlast"]
    r = _fst_case(tmp_path, np.random.default_rng(43), ["-w", "20,20"], 20, 20, annot_text="\n".join(rows) + "\n")
    cats = [l.split("\t")[0] for l in (tmp_path / "cli" / "runSNPcategory_fst.txt").read_text().splitlines()[1:]]
    assert {"exon", "intron", "other non-coding"} <= set(cats) and len(cats) > 500


def test_cli_fst_anc_sets(tmp_path):
    """--ancSets (process_vcf_fst.cpp:563-588): runancestralSNPs_fst.txt; ancestral allele = REF, = ALT, and neither
    ("-888"), on lines with '*' alleles and missing calls too."""
    _fst_case(tmp_path, np.random.default_rng(44), ["-w", "25,25"], 25, 25, anc=True)
    rows = [l.split("\t") for l in (tmp_path / "cli" / "runancestralSNPs_fst.txt").read_text().splitlines()[1:]]
    assert len(rows) > 500 and {r[2] for r in rows} >= {"-888", "0"} and {r[4] for r in rows} == {"0", "1"}


def test_cli_fst_sc3749(tmp_path):
    vcf = os.path.join(ROOT, "tests", "golden", "sc3749.vcf")
    names = open(vcf).readline().rstrip("\n").split("\t")[9:]
    sets = tmp_path / "sets.txt"
    sets.write_text(",".join(names[:25]) + "\n" + ",".join(names[25:]) + "\n")
    r = subprocess.run([EVO, "fst", "--vcf=" + vcf, "--sets=" + str(sets), "-w", "1,1", "-n", "t"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = (tmp_path / "t_w_1_fst.txt").read_text().splitlines()
    assert rows[1] == "1\tscaffold_3749\t904\t0.0962099\t0.224"       # the reference's own output
    assert "Fst: 0.0962099" in r.stderr


def _global_case(tmp_path, rng, args, size, step, fixed=10000, n_sites=1200, n_samples=36, crlf=False, pops=None, annot_text=None):
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, n_sites, n_samples, scaffolds=("chr1", "chr2"), crlf=crlf, miss=0.1)
    pops = pops or ["popA", "popB", "Outgroup", "popC"]
    pf = str(tmp_path / "pops.txt")
    with open(pf, "w") as f:
        for i, n in enumerate(names):
            if i % 11 == 10:
                continue                                          # sample that is in no population
            f.write("%s\t%s\n" % (n, pops[i % len(pops)]))
        f.write("GHOST\t%s\n" % pops[0])                          # ID the VCF does not have
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    annot = None
    if annot_text is not None:
        annot = str(tmp_path / "genes.gffExtract")
        open(annot, "w").write(annot_text)
        args = args + ["--annot=" + annot]
    r = subprocess.run([EVO, "FstGlobal", "-n", "g"] + args + [vcf, pf], cwd=str(cli_dir), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    O.fstglobal_run(vcf, pf, str(orc_dir), run_name="g", size=size, step=step, fixed=fixed, annot=annot)
    _same_files(str(cli_dir), str(orc_dir))
    return cli_dir


def test_cli_fstglobal_default(tmp_path):
    _global_case(tmp_path, np.random.default_rng(7), [], 20, 10)


def test_cli_fstglobal_equal_windows_crlf(tmp_path):
    _global_case(tmp_path, np.random.default_rng(8), ["-w", "25,25", "-f", "3000"], 25, 25, fixed=3000, crlf=True)


def test_cli_fstglobal_annot_gene_rows(tmp_path):
    """FstGlobal --annot (evo_FstAgainstAll.cpp:215-219, 267-297): one row per gene with its exon / intron / promoter SNP
    counts and every population's Fst over them; genes on both strands, genes without a kept SNP in some category, a
    partial transcript, and the first SNP of a gene going into the row of the gene before it."""
    rows = []
    for k in range(14):
        a = 4000 + 5500 * k
        ex = [(a, a + 300), (a + 900, a + 1300), (a + 2000, a + 2400)]
        if k % 2:
            ex.reverse()
        name = ("5partial-" if k == 5 else "") + "tr%d" % k
        rows += ["chr1\t%d\t%d\t%s\t%s" % (s, e, "-" if k % 2 else "+", name) for s, e in ex]
    rows += ["chr2\t1000\t30000\t+\tbig.1", "chr2\t40000\t40100\t-\tsmall", "chr2\t45000\t45100\t-\tsmall",
             "chrZ\t1\t10\t+\tlast"]
    d = _global_case(tmp_path, np.random.default_rng(71), ["-w", "20,10"], 20, 10, annot_text="\n".join(rows) + "\n")
    genes = (d / "g_FstGlobalGenes_g_20_10.txt").read_text().splitlines()
    assert len(genes) > 10 and genes[0].startswith("gene\tnumSNPsExons")


def test_cli_fstglobal_outgroup_sorts_first(tmp_path):
    """'Outgroup' and 'xxx' sort before the used populations: the setSizes index mismatch of
    evo_FstAgainstAll.cpp:253 becomes visible."""
    _global_case(tmp_path, np.random.default_rng(9), ["-w", "20,10"], 20, 10, pops=["zeta", "Outgroup", "yak", "xxx", "wolf"])


def test_cli_fst_several_device_batches(tmp_path):
    """70 k sites: the front-end hands the device three batches of 32768 lines; windows and the bp-window
    table run across the batch edges."""
    _fst_case(tmp_path, np.random.default_rng(21), ["-w", "50,50", "--physicalWindowSize=5000"], 50, 50, n_sites=70000,
              n_samples=8, scaffolds=("scaffold_0", "chrB", "chrC"), phys=5000)


def test_cli_fstglobal_several_device_batches(tmp_path):
    _global_case(tmp_path, np.random.default_rng(22), [], 20, 10, n_sites=70000, n_samples=8, pops=["popA", "popB", "popC"])


def _all_pairs_case(tmp_path, rng, size, step, n_sites, n_samples, pops, scaffolds, miss=0.08):
    """evo fst --pops --pairs=all (one pass, window sums from the device's segment table) against P (P - 1) / 2 two-set runs
    of the oracle driver: each pair's file equals the _fst.txt of `evo fst --sets` on the two populations."""
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, n_sites, n_samples, scaffolds=scaffolds, miss=miss)
    member = {}
    pf = str(tmp_path / "pops.txt")
    with open(pf, "w") as f:
        for i, n in enumerate(names):
            if i % 9 == 8:
                continue                                          # a sample in no population
            member.setdefault(pops[i % len(pops)], []).append(n)
            f.write("%s\t%s\n" % (n, pops[i % len(pops)]))
    cli_dir = tmp_path / "cli"
    cli_dir.mkdir()
    r = subprocess.run([EVO, "fst", "--vcf=" + vcf, "--pops=" + pf, "--pairs=all", "-n", "run", "-w", "%d,%d" % (size, step)],
                       cwd=str(cli_dir), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    sp = sorted(pops)
    n_rows = 0
    for a in range(len(sp)):
        for b in range(a + 1, len(sp)):
            od = tmp_path / ("orc_%d_%d" % (a, b))
            od.mkdir()
            sets = str(od / "sets.txt")
            open(sets, "w").write(",".join(member[sp[a]]) + "\n" + ",".join(member[sp[b]]) + "\n")
            O.fst_run(vcf, sets, str(od), run_name="run", size=size, step=step)
            want = open(str(od / ("run_w_%d_fst.txt" % size))).read().splitlines()
            got = open(str(cli_dir / ("run_%s_%s_w_%d_fst.txt" % (sp[a], sp[b], size)))).read().splitlines()
            for i, (x, y) in enumerate(zip(got, want)):
                if x == y:
                    continue
                # The sums are the device's (another order of additions than the reference's left-to-right one, equal to
                # 1e-12 of a window's terms): a window whose numerator cancels to zero prints the rounding noise of its own
                # order -- 0 against 6.4e-18, say.  Every other field, and every Fst that is not such noise, is the same text.
                fx, fy = x.split("\t"), y.split("\t")
                assert len(fx) == len(fy), (x, y)
                diff = [k for k in range(len(fx)) if fx[k] != fy[k]]
                assert all(abs(float(fx[k])) < 1e-12 and abs(float(fy[k])) < 1e-12 for k in diff), \
                    "pair %s/%s line %d:\n cli    %r\n oracle %r" % (sp[a], sp[b], i + 1, x, y)
            assert len(got) == len(want)
            n_rows += len(want) - 1
    assert n_rows > 20


def test_cli_fst_all_pairs_equal_windows(tmp_path):
    _all_pairs_case(tmp_path, np.random.default_rng(31), 40, 40, 4000, 27, ["zeta", "alpha", "mid", "beta"], ("scaffold_0", "chrB", "chrC"))


def test_cli_fst_all_pairs_first_window_swallowed_and_overlap(tmp_path):
    _all_pairs_case(tmp_path, np.random.default_rng(32), 30, 30, 2500, 18, ["p1", "p2", "p3"], ("chr1", "chr2"))
    (tmp_path / "second").mkdir()
    _all_pairs_case(tmp_path / "second", np.random.default_rng(33), 30, 10, 2500, 18, ["p1", "p2", "p3"], ("chr1", "chr2"))


def test_cli_fst_all_pairs_several_device_batches(tmp_path):
    _all_pairs_case(tmp_path, np.random.default_rng(34), 60, 20, 70000, 10, ["a", "b", "c"], ("scaffold_0", "chrB"))


def test_cli_fst_all_pairs_devices(tmp_path):
    """evo fst --pairs=all --devices N: N consecutive device batches are N site shards on contexts of their own (host
    threads; on a one-GPU box they share the GPU), the included-site totals and the carries go from shard to shard as in
    evo_amd/shard.py.  The files are the same bytes as with one device, whatever N: 3 shards in one round (the last one
    short), 2 shards in two rounds (the second round with one shard), 5 shards of which two stay empty."""
    rng = np.random.default_rng(36)
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, 70000, 12, scaffolds=("scaffold_0", "chrB"), miss=0.05)
    pf = str(tmp_path / "pops.txt")
    with open(pf, "w") as f:
        for i, n in enumerate(names):
            f.write("%s\t%s\n" % (n, ["a", "b", "c"][i % 3]))
    outs = {}
    for nd in (1, 2, 3, 5):
        d = tmp_path / ("dev%d" % nd)
        d.mkdir()
        r = subprocess.run([EVO, "fst", "--vcf=" + vcf, "--pops=" + pf, "--pairs=all", "-n", "run", "-w", "60,20", "--devices=%d" % nd],
                           cwd=str(d), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[nd] = {n: open(str(d / n), "rb").read() for n in sorted(os.listdir(str(d)))}
    assert len(outs[1]) == 3 and all(len(v.splitlines()) > 100 for v in outs[1].values())
    for nd in (2, 3, 5):
        assert outs[nd] == outs[1], "files differ with --devices=%d" % nd
    r = subprocess.run([EVO, "fst", "--vcf=" + vcf, "--sets=" + pf, "--devices=2"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0                                   # the option belongs to the all-pairs mode


def test_cli_fst_all_pairs_clean_chunks_across_batches(tmp_path):
    """No missing calls: most chunks are clean, K1 leaves them one byte per count and clean_walk_kernel takes them, with
    windows that run across the device batches (the ALT = "*" / indel lines of the file still mark their own chunks)."""
    _all_pairs_case(tmp_path, np.random.default_rng(35), 100, 50, 70000, 12, ["a", "b", "c", "d"], ("scaffold_0", "chrB"), miss=0.0)


def test_cli_allele_freq(tmp_path):
    """evo alleleFreq: per-population allele frequencies straight from K1's counts, -1 where a
    population has no called allele; the file is written next to the populations file, as the reference does."""
    rng = np.random.default_rng(31)
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, 40000, 10, scaffolds=("chr1", "chr2"), crlf=True, miss=0.3)
    pf = str(cli_dir / "pops.v2.txt")
    with open(pf, "w") as f:
        for i, n in enumerate(names):
            if i != 4:
                f.write("%s\t%s\n" % (n, ["popB", "Outgroup", "popA"][i % 3]))
    r = subprocess.run([EVO, "alleleFreq", "-n", "af", vcf, pf], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    O.af_run(vcf, pf, str(orc_dir), run_name="af")
    got = cli_dir / "pops.v2_af_AF.txt"
    assert got.exists()
    assert filecmp.cmp(str(got), str(orc_dir / "pops.v2_af_AF.txt"), shallow=False)
    assert sum(1 for _ in open(got)) > 30000


def test_cli_allele_freq_genotype_probabilities_fall_back(tmp_path):
    """--use-genotype-probabilities on a VCF whose FORMAT is GT alone (evo_AlleleFeq.cpp:141-150,
    process_vcf_utils.cpp:1122-1124): the reference warns, drops the option, and the first used line still prints the
    likelihood-based frequencies nothing has set (-1 for every population); `-g` itself is no short option there (:28).
    A VCF with other FORMAT tags would take the per-sample likelihood path (not carried: the binary says so)."""
    rng = np.random.default_rng(32)
    full = str(tmp_path / "full.vcf")
    names = write_vcf(full, rng, 500, 8)
    vcf = str(tmp_path / "gt.vcf")
    with open(vcf, "w") as f:
        for ln in open(full):
            if ln.startswith("#"):
                f.write(ln); continue
            c = ln.rstrip("\n").split("\t")
            f.write("\t".join(c[:8] + ["GT"] + [g.split(":")[0] for g in c[9:]]) + "\n")
    pf = str(tmp_path / "pops.txt")
    open(pf, "w").write("".join("%s\t%s\n" % (n, "AB"[i % 2]) for i, n in enumerate(names)))
    orc_dir = tmp_path / "orc"; orc_dir.mkdir()
    r = subprocess.run([EVO, "alleleFreq", "--use-genotype-probabilities", vcf, pf], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stderr.count("Reverting to using called genotypes") == 1, r.stderr[-2000:]
    O.af_run(vcf, pf, str(orc_dir))
    want = (orc_dir / "pops_out_AF.txt").read_text().splitlines()
    want[1] = "\t".join(want[1].split("\t")[:4] + ["-1", "-1"])
    assert (tmp_path / "pops_out_AF.txt").read_text().splitlines() == want
    r = subprocess.run([EVO, "alleleFreq", "--use-genotype-probabilities", full, pf], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "not carried" in r.stderr
    r = subprocess.run([EVO, "alleleFreq", "-g", vcf, pf], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def _pbs_case(tmp_path, rng, args, size, step, fixed=10000, n_sites=3000, n_samples=25, crlf=False, miss=0.25, quartets=False, annot_text=None, allow_indels=False, af=False):
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, n_sites, n_samples, scaffolds=("chr1", "chr2"), crlf=crlf, miss=miss)
    pf, tf = str(tmp_path / "pops.txt"), str(tmp_path / "trios.txt")
    pops = ["popC", "popA", "popD", "popB"]
    with open(pf, "w") as f:
        f.write("%s\tsolo\n" % names[0])                             # one sample: half calls leave a single allele
        f.write("%s\tduo\n%s\tduo\n" % (names[1], names[2]))
        for i, n in enumerate(names[3:]):
            if i % 9 != 8:
                f.write("%s\t%s\n" % (n, pops[i % 4]))
        f.write("GHOST\tpopA\n")
    if quartets:
        open(tf, "w").write("popA\tpopB\tpopC\tpopD\nsolo\tpopA\tpopB\tduo\npopA\tsolo\tduo\tpopC\npopD\tduo\tsolo\tpopB\n"
                            "popC\tpopB\tpopA\tsolo\nduo\tsolo\tduo\tpopA\npopD\tpopC\tpopB\tpopA\n")
    else:
        open(tf, "w").write("popA\tpopB\tpopC\nsolo\tpopA\tpopB\npopA\tsolo\tduo\npopD\tduo\tsolo\npopC\tpopB\tpopA\nduo\tsolo\tduo\n")
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    annot = None
    if annot_text is not None:
        annot = str(tmp_path / "genes.gffExtract")
        open(annot, "w").write(annot_text)
        args = args + ["--annot=" + annot]
    r = subprocess.run([EVO, "ABS" if quartets else "PBS", "-n", "p"] + args + [vcf, pf, tf], cwd=str(cli_dir), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    if quartets:
        O.abs_run(vcf, pf, tf, str(orc_dir), run_name="p", size=size, step=step)
    else:
        O.pbs_run(vcf, pf, tf, str(orc_dir), run_name="p", size=size, step=step, fixed=fixed, annot=annot, allow_indels=allow_indels, af=af)
    if af:                                                           # written next to the populations file (evo_PBS.cpp:265)
        os.rename(str(tmp_path / "pops_AF.txt"), str(cli_dir / "pops_AF.txt"))
    _same_files(str(cli_dir), str(orc_dir))
    return cli_dir


def _gene_annotation(scaffold_a, scaffold_b):
    rows = []
    for k in range(14):
        a = 4000 + 5500 * k
        ex = [(a, a + 300), (a + 900, a + 1300), (a + 2000, a + 2400)]
        if k % 2:
            ex.reverse()
        name = ("5partial-" if k == 5 else "") + "tr%d" % k
        rows += ["%s\t%d\t%d\t%s\t%s" % (scaffold_a, s, e, "-" if k % 2 else "+", name) for s, e in ex]
    rows += ["%s\t1000\t30000\t+\tbig.1" % scaffold_b, "%s\t40000\t40100\t-\tsmall" % scaffold_b,
             "%s\t45000\t45100\t-\tsmall" % scaffold_b, "chrZ\t1\t10\t+\tlast"]
    return "\n".join(rows) + "\n"


def test_cli_pbs_annot_gene_rows(tmp_path):
    """PBS --annot (evo_PBS.cpp:282-286, 333-361): one row per gene and trio; ABS --annot loads the file and writes
    nothing more (its gene rows are commented out in the reference)."""
    d = _pbs_case(tmp_path, np.random.default_rng(45), [], 20, 10, annot_text=_gene_annotation("chr1", "chr2"))
    genes = (d / "popA_popB_popC_PBSGenes_p_20_10.txt").read_text().splitlines()
    assert len(genes) > 10 and len(os.listdir(d)) == 18
    (tmp_path / "abs").mkdir()
    _pbs_case(tmp_path / "abs", np.random.default_rng(46), ["-w", "9,4"], 9, 4, quartets=True, annot_text=_gene_annotation("chr1", "chr2"))


def test_cli_pbs_allow_indels(tmp_path):
    """PBS -i (evo_PBS.cpp:248-256): indels and multi-allelic lines are used too, counted by their '0' and '1' characters;
    more sites are used than without it."""
    d = _pbs_case(tmp_path, np.random.default_rng(47), ["-i"], 20, 10, allow_indels=True)
    with_i = len((d / "popA_popB_popC_PBS_p_20_10.txt").read_text().splitlines())
    (tmp_path / "plain").mkdir()
    d0 = _pbs_case(tmp_path / "plain", np.random.default_rng(47), [], 20, 10)
    assert with_i > len((d0 / "popA_popB_popC_PBS_p_20_10.txt").read_text().splitlines())


def test_cli_pbs_af_leaves_the_empty_file(tmp_path):
    """PBS --af=ARG (evo_PBS.cpp:48, 264-271, 393): the reference reopens POPULATIONS_AF.txt at every used line on a stream it
    never flushes or closes, so a run leaves the file empty; the other files are those of a run without it."""
    d = _pbs_case(tmp_path, np.random.default_rng(48), ["--af=1"], 20, 10, af=True)
    assert (d / "pops_AF.txt").stat().st_size == 0 and len(os.listdir(d)) == 13
    r = subprocess.run([EVO, "ABS", "--af=1", "a", "b", "c"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0                                        # ABS has no such option (evo_ABS.cpp:39-45)


def test_cli_pbs_default(tmp_path):
    """evo PBS: per-site PBS of six trios from the device, windows replayed on the host.  The solo
    population makes NaN rows ("nan" and "-nan", as the x86 build of the reference signs them)."""
    d = _pbs_case(tmp_path, np.random.default_rng(41), [], 20, 10)
    assert len(os.listdir(d)) == 12
    text = (d / "solo_popA_popB_PBS_p_20_10.txt").read_text() + (d / "popD_duo_solo_PBS_p_20_10.txt").read_text()
    assert "\tnan" in text and "\t-nan" in text
    assert sum(1 for _ in open(d / "popA_popB_popC_PBS_p_20_10.txt")) > 100


def test_cli_pbs_windows_fixed_crlf(tmp_path):
    _pbs_case(tmp_path, np.random.default_rng(42), ["-w", "7,3", "-f", "2500"], 7, 3, fixed=2500, crlf=True, miss=0.1)


def test_cli_pbs_several_device_batches(tmp_path):
    _pbs_case(tmp_path, np.random.default_rng(43), ["--window=50,25", "--fixedW", "20000"], 50, 25, fixed=20000, n_sites=70000,
              n_samples=12, miss=0.15)


def test_cli_abs_default(tmp_path):
    """evo ABS: per-site ABS / ABSalt of seven quartets from the device, windows replayed on the host;
    the solo population makes NaN rows of either sign, and quartets where only T_WY or T_XZ is NaN stay finite."""
    d = _pbs_case(tmp_path, np.random.default_rng(51), [], 20, 10, quartets=True)
    assert len(os.listdir(d)) == 7
    text = "".join((d / n).read_text() for n in os.listdir(d))
    assert "\tnan" in text and "\t-nan" in text
    assert sum(1 for _ in open(d / "popA_popB_popC_popD_ABS_p_20_10.txt")) > 100


def test_cli_abs_windows_crlf_batches(tmp_path):
    _pbs_case(tmp_path, np.random.default_rng(52), ["-w", "9,4"], 9, 4, crlf=True, miss=0.12, n_sites=40000, n_samples=14, quartets=True)


def _dist_case(tmp_path, rng, args, fixed=10000, n_sites=3000, n_samples=25, crlf=False, miss=0.3, bed_rows=None, allow_indels=False):
    vcf = str(tmp_path / "in.vcf")
    names = write_vcf(vcf, rng, n_sites, n_samples, scaffolds=("chr1", "chr2"), crlf=crlf, miss=miss)
    pf, of, inf = str(tmp_path / "pops.txt"), str(tmp_path / "outgroups.txt"), str(tmp_path / "ingroups.txt")
    pops = ["popC", "popA", "popD", "popB"]
    with open(pf, "w") as f:
        f.write("%s\tsolo\n" % names[0])                             # one sample: often entirely missing
        f.write("%s\tout2\n%s\tout2\n" % (names[1], names[2]))
        for i, n in enumerate(names[3:]):
            if i % 9 != 8:
                f.write("%s\t%s\n" % (n, pops[i % 4]))
    open(of, "w").write("solo\nout2\npopA\n")
    open(inf, "w").write("popA\npopB\npopC\nsolo\npopD\n")
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    bed = None
    if bed_rows is not None:
        bed = str(tmp_path / "accessible.bed")
        open(bed, "w").write("".join("%s\t%d\t%d\n" % row for row in bed_rows))
        args = args + ["--accessibleGenomeBED=" + bed]
    r = subprocess.run([EVO, "DistOutgroups", "-n", "d"] + args + [vcf, pf, of, inf], cwd=str(cli_dir), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    O.distout_run(vcf, pf, of, inf, str(orc_dir), run_name="d", fixed=fixed, bed=bed, allow_indels=allow_indels)
    _same_files(str(cli_dir), str(orc_dir))
    return cli_dir


def test_cli_distoutgroups_default(tmp_path):
    """evo DistOutgroups: windowed dxy of 3 outgroups x 5 ingroups; the device adds up in site order, so
    the rows match the oracle's to the last printed digit; windows with too much missing data print nan."""
    d = _dist_case(tmp_path, np.random.default_rng(61), [])
    assert len(os.listdir(d)) == 3
    rows = (d / "solo_DIST_d_FW10000.txt").read_text().splitlines()
    assert len(rows) > 30 and all(r.split("\t")[5] == "10000" for r in rows[1:])


def test_cli_distoutgroups_allow_indels(tmp_path):
    """DistOutgroups -i (evo_distanceToOutgroups.cpp:194-202): only ALT == "*" is left out."""
    _dist_case(tmp_path, np.random.default_rng(64), ["-i", "-f", "5000"], fixed=5000, allow_indels=True)


def test_cli_distoutgroups_small_windows_crlf_batches(tmp_path):
    d = _dist_case(tmp_path, np.random.default_rng(62), ["-f", "2000"], fixed=2000, n_sites=70000, n_samples=12, crlf=True, miss=0.4)
    assert "\tnan" in (d / "out2_DIST_d_FW2000.txt").read_text()       # windows where an ingroup is mostly missing


def test_cli_distoutgroups_accessible_genome_bed(tmp_path):
    """--accessibleGenomeBED (evo_distanceToOutgroups.cpp:223-230): the divisor of a window comes from the BED; the row that
    closes a scaffold asks the NEXT scaffold's features, as the reference does; windows without an accessible base
    divide by 0 (inf / nan rows)."""
    rows = [("chr1", 0, 3000), ("chr1", 3500, 9000), ("chr1", 12000, 12001), ("chr1", 30000, 150000), ("chr1", 160000, 200000),
            ("chr2", 100, 1100), ("chr2", 4000, 100000)]
    d = _dist_case(tmp_path, np.random.default_rng(63), ["-f", "4000"], fixed=4000, bed_rows=rows)
    sizes = {r.split("\t")[5] for r in (d / "out2_DIST_d_FW4000.txt").read_text().splitlines()[1:]}
    assert len(sizes) > 3 and "4000" in sizes and "0" in sizes


def _stats_case(tmp_path, rng, args, block=0, accessible=-1, n_sites=3000, n_samples=19, crlf=False, miss=0.1, diffs=True, hets=False):
    vdir = tmp_path / "data"
    vdir.mkdir()
    vcf = str(vdir / "cohort.v1.vcf")
    write_vcf(vcf, rng, n_sites, n_samples, crlf=crlf, miss=miss)
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    r = subprocess.run([EVO, "stats"] + (["--diff-matrix"] if diffs else []) + (["--hets-per-individual"] if hets else []) + args + [vcf],
                       cwd=str(cli_dir), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in os.listdir(vdir):                                        # the bootstrap replicates and the het counts land next to the VCF
        if "boot." in f or "ets.txt" in f:
            shutil.move(str(vdir / f), str(cli_dir / f))
    if hets or not diffs:
        O.stats_run(vcf, str(orc_dir), diffs=diffs, hets=hets, block=block, accessible=accessible)
    else:
        O.stats_diffs_run(vcf, str(orc_dir), block=block, accessible=accessible)
    _same_files(str(cli_dir), str(orc_dir))
    return cli_dir


def test_cli_stats_diff_matrix(tmp_path):
    """evo stats --diff-matrix: the four matrix files from the device's five counts per pair; multi-allelic
    lines, '*' alleles, half calls, indels (this sub-program filters nothing) are all in the input."""
    d = _stats_case(tmp_path, np.random.default_rng(71), [])
    assert sorted(os.listdir(d)) == ["cohort.v1.diff_matrix.txt", "cohort.v1.diff_me_matrix.txt",
                                     "cohort.v1.hets_over_homs_matrix.txt", "cohort.v1.parwiseMissingness.txt"]
    rows = (d / "cohort.v1.diff_me_matrix.txt").read_text().splitlines()
    assert len(rows) == 3 + 1 + 19 and rows[-1].split("\t")[0] not in ("0", "")


def test_cli_stats_hets_per_individual(tmp_path):
    """evo stats --hets-per-individual (process_vcf_stats.cpp:239-242, 336-337; het_analysis, print_het_counts): heterozygous
    calls per individual at the sites without a missing genotype, alone and next to --diff-matrix, over several device
    batches; an individual without such a site prints the proportion 0 / 0 as the reference's build does ("-nan")."""
    d = _stats_case(tmp_path, np.random.default_rng(74), [], n_sites=70000, n_samples=11, miss=0.004, diffs=False, hets=True)
    assert sorted(os.listdir(d)) == ["cohort.v1.hets.txt", "cohort.v1.sharedHets.txt"]
    counts = [int(x) for x in (d / "cohort.v1.hets.txt").read_text().splitlines()[3].split("\t")]
    assert len(counts) == 11 and min(counts) > 1000
    (tmp_path / "both").mkdir()
    d = _stats_case(tmp_path / "both", np.random.default_rng(75), [], n_sites=900, n_samples=3, miss=0.3, hets=True)
    assert len(os.listdir(d)) == 6
    (tmp_path / "none").mkdir()
    d = _stats_case(tmp_path / "none", np.random.default_rng(76), [], n_sites=40, n_samples=30, miss=0.3, diffs=False, hets=True)
    assert "-nan" in (d / "cohort.v1.sharedHets.txt").read_text()


def test_cli_stats_names_file_and_population_options(tmp_path):
    """--ind-file replaces the header names (process_vcf_stats.cpp:195-197); --pop-file / --private-variants write the
    reference's summary of zeros (its counting is commented out, :243-246, print_routines.cpp:59-69) and
    --doubleton-distribution writes nothing (:247-249, 332-333)."""
    rng = np.random.default_rng(77)
    vcf = str(tmp_path / "c.vcf")
    names = write_vcf(vcf, rng, 600, 7, miss=0.05)
    ind = tmp_path / "names.txt"
    ind.write_text("".join("ind_%d\n" % i for i in range(7)))
    pops = tmp_path / "pops.txt"
    # the populations are looked for among the names of --ind-file, as in the reference (:201-216)
    pops.write_text("".join("ind_%d\t%s\n" % (i, ["west", "Outgroup", "east", "xxx"][i % 4]) for i in range(7)))
    cli_dir, orc_dir = tmp_path / "cli", tmp_path / "orc"
    cli_dir.mkdir(); orc_dir.mkdir()
    r = subprocess.run([EVO, "stats", "--diff-matrix", "--hets-per-individual", "--ind-file=" + str(ind), "--doubleton-distribution",
                        "--private-variants", "--pop-file=pops.txt", vcf], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    O.stats_run(vcf, str(orc_dir), diffs=True, hets=True)
    for f in os.listdir(orc_dir):                                     # the oracle driver has no --ind-file: put the names in
        want = (orc_dir / f).read_text().replace("\t".join(names), "\t".join("ind_%d" % i for i in range(7)))
        assert (tmp_path / f).read_text() == want, f
    assert (tmp_path / "c_pops.privateFixedVars.txt").read_text() == (
        "# Counts of private fixed variants:\n# Input file:%s.vcf\n# Groups defined in:pops.txt\neast\twest\n0\t0\n" % str(tmp_path / "c"))
    assert not any("doubleton" in f for f in os.listdir(tmp_path))
    r = subprocess.run([EVO, "stats", "--private-variants", vcf], cwd=str(cli_dir), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "You need to supply a file defining populations" in r.stderr            # :393-396
    (tmp_path / "short.txt").write_text("a\nb\n")
    r = subprocess.run([EVO, "stats", "--diff-matrix", "--ind-file=" + str(tmp_path / "short.txt"), vcf], cwd=str(cli_dir), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "--ind-file" in r.stderr


def test_cli_stats_bootstrap_across_batches(tmp_path):
    """Blocks of 700 variants over 70 k lines (three device batches; blocks straddle the batch edges), 100 replicates
    drawn with the C library's unseeded rand() as the reference does, and --numAccessibleBP."""
    d = _stats_case(tmp_path, np.random.default_rng(72), ["--block-bootstrap=700", "--numAccessibleBP=12345"], block=700, accessible=12345,
                    n_sites=70000, n_samples=9, crlf=True, miss=0.05)
    assert len(os.listdir(d)) == 104


def test_cli_stats_bootstrap_without_a_full_block(tmp_path):
    d = _stats_case(tmp_path, np.random.default_rng(73), ["--block-bootstrap=5000"], block=5000, n_sites=800, n_samples=5)
    assert "nan" in (d / "cohort.v1boot.0.txt").read_text()


def test_cli_rejects_unknown_command():
    r = subprocess.run([EVO, "getCodingSeq"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unrecognized command" in r.stderr
