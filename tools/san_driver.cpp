// san_driver.cpp -- the host code that reads untrusted text, under AddressSanitizer + UndefinedBehaviorSanitizer.
//
// Built by `make -C evo_amd/csrc san` from the library's own sources (evofst_api.cpp compiled with the host sanitizers;
// the GPU code objects are untouched: no sanitizer runs on the device) and run by tests/test_sanitizers.py on a corpus
// of truncated and corrupted VCF data lines.  No GPU: every call here goes through a host-only context
// (evofst_create(-1)).  Exit code 0 = every line was answered with EVOFST_OK or EVOFST_ERR_ARG and the sanitizers
// found nothing (they abort the process otherwise: -fno-sanitize-recover).
//
// Corpus file: records of {uint32 length, bytes} (lines may hold any byte but NUL).
#include "../include/evofst.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: san_driver corpus.bin n_samples\n"); return 2; }
    const int n = atoi(argv[2]);
    FILE *fp = fopen(argv[1], "rb");
    if (!fp || n <= 0) { fprintf(stderr, "san_driver: cannot read %s\n", argv[1]); return 2; }
    std::vector<std::string> lines;
    for (;;) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, fp) != 1) break;
        std::string s(len, '\0');
        if (len && fread(&s[0], 1, len, fp) != len) { fprintf(stderr, "san_driver: truncated corpus\n"); return 2; }
        lines.push_back(s);
    }
    fclose(fp);
    evofst_ctx *c = evofst_create(-1);
    if (!c) { fprintf(stderr, "san_driver: no host-only context\n"); return 2; }
    std::vector<int32_t> sp(n);
    for (int i = 0; i < n; ++i) sp[i] = i % 3;
    if (evofst_set_layout(c, n, 3, sp.data()) != EVOFST_OK) { fprintf(stderr, "san_driver: set_layout: %s\n", evofst_last_error(c)); return 2; }
    const size_t tb = (size_t)evofst_tiles_bytes(c, 64);
    long ok = 0, bad = 0, other = 0;
    auto tally = [&](int rc) { if (rc == EVOFST_OK) ++ok; else if (rc == EVOFST_ERR_ARG) ++bad; else ++other; };
    // exactly-sized heap buffers, so that one byte too far is a report
    for (const std::string &s : lines) {
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<unsigned char> tiles(tb);
            std::vector<uint8_t> flags(1);
            char *copy = (char *)malloc(s.size() + 1);
            memcpy(copy, s.c_str(), s.size() + 1);
            const char *arr[1] = {copy};
            tally(evofst_pack_vcf_lines(c, mode, 1, arr, n, nullptr, 1, tiles.data(), flags.data()));
            std::vector<uint8_t> codes(2 * (size_t)n);
            uint32_t fl = 0;
            const char *chrom = nullptr, *pos = nullptr;
            tally(evofst_codes_from_vcf_line(copy, mode, n, codes.data(), &fl, &chrom, &pos));   // modifies the copy
            free(copy);
        }
        {
            char *copy = (char *)malloc(s.size() + 1);
            memcpy(copy, s.c_str(), s.size() + 1);
            std::vector<uint8_t> al(2 * (size_t)n);
            tally(evofst_alleles_from_vcf_line(copy, n, al.data()));
            const char *arr[1] = {copy};
            int64_t d = 0;
            tally(evofst_stats_phase_draws(1, arr, n, 1, &d));
            free(copy);
        }
    }
    // a block of lines on several threads, good and bad mixed: the bulk path's error handling
    {
        std::vector<const char *> arr;
        for (const std::string &s : lines) arr.push_back(s.c_str());
        const int64_t nl = (int64_t)arr.size();
        std::vector<unsigned char> tiles((size_t)evofst_tiles_bytes(c, nl));
        std::vector<uint8_t> flags((size_t)nl);
        for (int mode = 0; mode < 3; ++mode) tally(evofst_pack_vcf_lines(c, mode, nl, arr.data(), n, nullptr, 4, tiles.data(), flags.data()));
    }
    evofst_destroy(c);
    printf("san_driver: %zu lines, %ld calls ok, %ld refused, %ld other\n", lines.size(), ok, bad, other);
    return other == 0 && ok > 0 && bad > 0 ? 0 : 1;
}
