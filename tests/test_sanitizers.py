"""The library's VCF text parsers under AddressSanitizer + UndefinedBehaviorSanitizer (host code; the pool has no
sanitizer for device code).  evo_amd/csrc/build/san_driver is evofst_api.cpp compiled with -fsanitize=address,undefined
plus tools/san_driver.cpp (`make -C evo_amd/csrc san`; __graft_entry__.build() makes it); it feeds every corpus line to
evofst_pack_vcf_lines (three modes, one line and a threaded block), evofst_codes_from_vcf_line,
evofst_alleles_from_vcf_line and evofst_stats_phase_draws on exactly-sized heap buffers.  The corpus is the seeded
mutation set of test_packer_vcf.py::test_packer_survives_malformed_lines plus hand-made truncations."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "evo_amd", "csrc", "build", "san_driver")


def _corpus(n):
    rng = np.random.default_rng(99)
    good = ["chr1\t%d\t.\tA\t%s\t9\t.\tDP=1\tGT:DP\t" % (100 + i, alt) + "\t".join(
        "%s%s%s:%d" % (rng.choice(list("01.2")), rng.choice(list("/|")), rng.choice(list("01.2")), rng.integers(0, 99)) for _ in range(n))
        for i, alt in enumerate(["G", "G,*", "*,G", "G,T,*", "GT", "*", ""])]
    out = [g.encode() for g in good]
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
        out.append(bytes(b).replace(b"\x00", b"x"))
    g = good[0].encode()
    # every prefix of a good line; a line of tabs; fields cut to one and two characters; a very long field
    out += [g[:k] for k in range(len(g))]
    out += [b"\t" * 40, b"", b"\t".join([b"c", b"1", b".", b"A", b"G", b".", b".", b".", b"GT"] + [b"0"] * n),
            b"\t".join([b"c", b"1", b".", b"A", b"G", b".", b".", b".", b"GT"] + [b"0/"] * n),
            b"\t".join([b"c", b"1", b".", b"A" * 70000, b"G", b".", b".", b".", b"GT"] + [b"0/1"] * n),
            b"\t".join([b"c", b"1", b".", b"A", b"G", b".", b".", b".", b"GT"] + [b"0/1"] * (n - 1)) + b"\t"]
    return out


def test_vcf_parsers_under_asan_ubsan(tmp_path):
    if not os.path.exists(DRIVER):
        rc = subprocess.call(["make", "-C", os.path.join(ROOT, "evo_amd", "csrc"), "san"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if rc != 0 or not os.path.exists(DRIVER):
            pytest.skip("the sanitizer build needs hipcc (make -C evo_amd/csrc san)")
    n = 11
    path = tmp_path / "corpus.bin"
    with open(path, "wb") as f:
        for line in _corpus(n):
            f.write(struct.pack("<I", len(line)) + line)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([DRIVER, str(path), str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "refused" in r.stdout
