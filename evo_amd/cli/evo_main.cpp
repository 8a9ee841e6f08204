// evo_main.cpp -- CLI front-end that keeps `evo fst` and `evo FstGlobal` (options, input files,
// output file names and formats) and runs the statistics on the GPU through include/evofst.h.
//
//   reference entry points replaced:  fstMain        process_vcf_fst.cpp:936   (VCF mode only)
//                                     FstGlobalMain  evo_FstAgainstAll.cpp:62
//                                     AFmain         evo_AlleleFeq.cpp:46
//                                     PBSmain        evo_PBS.cpp:126
//                                     ABSmain        evo_ABS.cpp:109
//                                     DistOutMain    evo_distanceToOutgroups.cpp:68
//                                     statsMain      process_vcf_stats.cpp:98     (--diff-matrix only)
//
// Division of labour: the host parses VCF text and packs genotypes (stage K0), the device computes
// per-site counts and statistics (K1 + per-site kernels), and the host replays the reference's two
// window state machines over those per-site values, so that the text outputs -- including the
// reference's load-bearing quirks -- are byte-identical.  Output goes
// through iostreams with default formatting, exactly like the reference.
// The optional host-side consumers of the per-site values are carried too: --annot (fst, FstGlobal), --ancSets,
// --accessibleGenomeBED (fst, DistOutgroups), --regions-above, PBS --annot / --af / -i.  Not carried over: the --ms* /
// --eigen modes of `evo fst` (text post-processing of other programs' output, out of scope).
#include <getopt.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <functional>
#include <numeric>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <condition_variable>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/evofst.h"

using std::string;

namespace {

const int kChunkSites = 1 << 15;  // sites per device batch

[[noreturn]] void die(const string &msg) {
    std::cerr << "Error: " << msg << std::endl;
    exit(EXIT_FAILURE);
}
void check(evofst_ctx *c, int rc, const char *what) {
    if (rc != EVOFST_OK) die(string(what) + ": " + evofst_last_error(c));
}

// split() of the reference (process_vcf_utils.cpp:104-130): getline on a stringstream
std::vector<string> split(const string &s, char delim) {
    std::vector<string> elems;
    std::stringstream ss(s);
    string item;
    while (std::getline(ss, item, delim)) elems.push_back(item);
    return elems;
}
template <typename T> string numToString(T i) { std::stringstream out; out << i; return out.str(); }

// createReader (process_vcf_utils.cpp:872-886): gz by suffix; zlib reads plain files transparently
// A reader thread takes the decompressed text in 4 MiB blocks, cuts it at '\n' with memchr and hands the lines over in
// blocks of 2048, so reading overlaps the packing and the device work of the batch before.
struct LineReader {
    gzFile f = nullptr;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cvData, cvRoom;
    std::deque<std::vector<string>> queue;
    bool done = false, stop = false;
    std::vector<string> cur;
    size_t curPos = 0;
    static constexpr size_t kBlockLines = 2048, kMaxQueued = 16;

    explicit LineReader(const string &path) {
        f = gzopen(path.c_str(), "rb");
        if (!f) die("could not open " + path);
        gzbuffer(f, 1 << 20);
        worker = std::thread([this] { produce(); });
    }
    ~LineReader() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cvRoom.notify_all();
        if (worker.joinable()) worker.join();
        if (f) gzclose(f);
    }
    // false when the consumer has gone away
    bool push(std::vector<string> &block) {
        std::unique_lock<std::mutex> lk(mu);
        cvRoom.wait(lk, [this] { return stop || queue.size() < kMaxQueued; });
        if (stop) return false;
        queue.emplace_back(std::move(block));
        block.clear();
        block.reserve(kBlockLines);
        lk.unlock();
        cvData.notify_one();
        return true;
    }
    void produce() {
        std::vector<char> buf(4u << 20);
        std::vector<string> block;
        block.reserve(kBlockLines);
        string partial;
        bool havePartial = false;        // text behind the last '\n' seen so far (std::getline returns it as the last line)
        for (;;) {
            const int n = gzread(f, buf.data(), (unsigned)buf.size());
            if (n <= 0) break;
            const char *p = buf.data(), *end = p + n;
            while (p < end) {
                const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
                if (!nl) { partial.append(p, (size_t)(end - p)); havePartial = true; break; }
                if (havePartial) { partial.append(p, (size_t)(nl - p)); block.emplace_back(std::move(partial)); partial.clear(); havePartial = false; }
                else block.emplace_back(p, (size_t)(nl - p));
                p = nl + 1;
                if (block.size() == kBlockLines && !push(block)) return;
            }
        }
        if (havePartial) block.emplace_back(std::move(partial));
        if (!block.empty() && !push(block)) return;
        { std::lock_guard<std::mutex> lk(mu); done = true; }
        cvData.notify_all();
    }
    // std::getline's contract: false only when nothing is left; a last line without '\n' is returned
    bool getline(string &line) {
        if (curPos == cur.size()) {
            std::unique_lock<std::mutex> lk(mu);
            cvData.wait(lk, [this] { return done || !queue.empty(); });
            if (queue.empty()) { line.clear(); return false; }
            cur = std::move(queue.front());
            queue.pop_front();
            curPos = 0;
            lk.unlock();
            cvRoom.notify_one();
        }
        line = std::move(cur[curPos++]);
        return true;
    }
};

// locateSet (process_vcf_utils.cpp:923-936)
std::vector<size_t> locateSet(const std::vector<string> &names, const std::vector<string> &set) {
    std::vector<size_t> locs;
    for (const string &s : set) {
        auto it = std::find(names.begin(), names.end(), s);
        if (it == names.end()) std::cerr << "Did not find the sample: \"" << s << "\"" << std::endl;
        else locs.push_back((size_t)std::distance(names.begin(), it));
    }
    return locs;
}

double vector_average(const std::vector<double> &v) {   // process_vcf_utils.h:369-376
    double sum = 0;
    for (size_t i = 0; i < v.size(); i++) sum += v[i];
    return (double)sum / (double)v.size();
}
double calculateFst(double numAvg, double denAvg) {      // process_vcf_fst.cpp:354-360
    double Fst = numAvg / denAvg;
    if (Fst < 0) Fst = 0;
    return Fst;
}

// An invalid operation (0/0, inf*0) yields the default NaN: sign bit SET on x86 (prints "-nan", what the
// reference writes) but CLEAR on the GPU (prints "nan").  The replay below continues the reference's
// x86 arithmetic, so device NaNs are given the x86 default's sign before they enter it.
void x86_nan_sign(std::vector<double> &v) {
    for (double &x : v)
        if (x != x) x = -__builtin_fabs(x);
}

// EVO_TIMING=1: where the wall time of a run goes (stderr, at exit).  Reading = everything outside the batches.
struct Timing {
    bool on = getenv("EVO_TIMING") != nullptr;
    double pack = 0, device = 0, batches = 0;
    std::mutex mu;                      // (several device threads report: evo fst --devices)
    void add(double &field, double dt) { std::lock_guard<std::mutex> lock(mu); field += dt; }
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    static double since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
    ~Timing() {
        if (!on) return;
        const double total = since(t0);
        std::cerr << "[timing] total " << total << " s: reading + start-up " << total - batches << ", packing " << pack << ", upload + counts " << device
                  << ", per-site results + window replay " << batches - pack - device << std::endl;
    }
} g_timing;

// One batch of parsed sites waiting for the device.
struct Batch {
    std::vector<string> lines;    // the VCF data lines as read
    std::vector<uint8_t> flags;   // filled by the packer
    std::vector<string> chrom, pos;
    size_t n = 0;
    void clear() { lines.clear(); flags.clear(); chrom.clear(); pos.clear(); n = 0; }
    // keeps the line and its first two columns; the genotype columns are parsed in bulk by Device::load
    void add(string &line, long long lineNo) {
        const size_t t1 = line.find('\t');
        const size_t t2 = t1 == string::npos ? string::npos : line.find('\t', t1 + 1);
        if (t2 == string::npos) die("malformed VCF line " + numToString(lineNo));
        chrom.emplace_back(line, 0, t1); pos.emplace_back(line, t1 + 1, t2 - t1 - 1);
        lines.emplace_back(std::move(line));
        ++n;
    }
};

struct Device {
    evofst_ctx *ctx = nullptr;
    // two pinned staging buffers (hipHostMalloc behind evofst_alloc_pinned), used in turn: a batch is packed straight into
    // page-locked memory and its hipMemcpyAsync runs at the full PCIe rate (54 against 29 GB/s from pageable memory,
    // tools/pcie_rate.py) while the other buffer is free for the next batch
    uint8_t *pinned[2] = {nullptr, nullptr};
    size_t pinnedBytes[2] = {0, 0};
    int turn = 0;
    void open(int device = 0) {
        ctx = evofst_create(device);
        if (!ctx) die(string("no usable GPU: ") + evofst_last_error(nullptr));
    }
    void close() {
        if (!ctx) return;
        for (uint8_t *p : pinned) if (p) evofst_free_pinned(ctx, p);
        evofst_destroy(ctx);
        ctx = nullptr;
    }
    uint8_t *stage(size_t bytes) {
        turn ^= 1;
        if (pinnedBytes[turn] < bytes) {
            if (pinned[turn]) evofst_free_pinned(ctx, pinned[turn]);
            pinned[turn] = static_cast<uint8_t *>(evofst_alloc_pinned(ctx, bytes));
            if (!pinned[turn]) die(string("pinned staging buffer: ") + evofst_last_error(ctx));
            pinnedBytes[turn] = bytes;
        }
        return pinned[turn];
    }
    // layoutToVcf[k] = genotype column of layout sample k (empty: identity)
    // allowIndels (-i of PBS and DistOutgroups, evo_PBS.cpp:248-256): only ALT == "*" is left out (EVOFST_MODE_GLOBAL_INDELS)
    void load(Batch &b, int mode, int nVcfSamples, const std::vector<int32_t> &layoutToVcf, long long firstLineNo, bool allowIndels = false) {
        if (allowIndels && mode == EVOFST_MODE_GLOBAL) mode = EVOFST_MODE_GLOBAL_INDELS;
        uint8_t *tiles = stage(std::max(evofst_tiles_bytes(ctx, (int64_t)b.n), evofst_tiles_bytes(ctx, (int64_t)kChunkSites)));
        b.flags.assign(b.n, 0);
        std::vector<const char *> ptr(b.n);
        for (size_t i = 0; i < b.n; ++i) ptr[i] = b.lines[i].c_str();
        const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        auto t = std::chrono::steady_clock::now();
        if (evofst_pack_vcf_lines(ctx, mode, (int64_t)b.n, ptr.data(), nVcfSamples, layoutToVcf.empty() ? nullptr : layoutToVcf.data(),
                                  threads, tiles, b.flags.data()) != EVOFST_OK)
            die("malformed VCF line among lines " + numToString(firstLineNo) + ".." + numToString(firstLineNo + (long long)b.n - 1) +
                " (diploid GT fields with single-digit alleles are required)");
        g_timing.add(g_timing.pack, Timing::since(t));
        t = std::chrono::steady_clock::now();
        check(ctx, evofst_upload(ctx, (int64_t)b.n, tiles, b.flags.data(), nullptr), "upload");
        check(ctx, evofst_run_counts(ctx), "run_counts");
        check(ctx, evofst_sync(ctx), "sync");
        g_timing.add(g_timing.device, Timing::since(t));
    }
};

// ============================================================================================ fst

// --accessibleGenomeBED (process_vcf_fst.cpp:414-419, 593-595, 613-616; AccessibleGenome / BedCoordinateFeatures,
// process_vcf_annotation_tools.h:340-530).  Host bookkeeping next to the device path: per scaffold the feature starts and
// ends in file order (a scaffold that comes back later in the file replaces its earlier entry, as the reference's map
// assignment does), and the bases of a 1-based region the features cover, counted with the reference's four overlap rules
// (a feature that begins before the region adds one base more than its overlap).
struct AccessibleBed {
    std::map<string, std::pair<std::vector<int>, std::vector<int>>> features;
    void load(const string &path) {
        std::ifstream in(path.c_str());
        if (!in) die("could not open " + path);
        string line, previous;
        std::vector<int> starts, ends;
        while (getline(in, line)) {
            std::vector<string> f = split(line, '\t');
            if (f.size() < 3) continue;
            if (f[0] != previous && !previous.empty()) { features[previous] = std::make_pair(starts, ends); starts.clear(); ends.clear(); }
            starts.push_back(atoi(f[1].c_str())); ends.push_back(atoi(f[2].c_str()));
            previous = f[0];
        }
        features[previous] = std::make_pair(starts, ends);
    }
    int basesIn(const string &scaffold, int start, int end) const {
        auto it = features.find(scaffold);
        if (it == features.end()) return 0;
        const std::vector<int> &s = it->second.first, &e = it->second.second;
        size_t i = (size_t)(std::lower_bound(e.begin(), e.end(), start) - e.begin());   // first feature that ends at or behind `start`
        int bases = 0;
        for (; i < s.size() && s[i] <= end; ++i) {
            const bool before = s[i] < start, beyond = e[i] > end;
            if (before && !beyond) bases += e[i] - start + 1;
            else if (!before && !beyond) bases += e[i] - s[i];
            else if (!before && beyond) bases += end - s[i];
            else bases += end - start + 1;
        }
        return bases;
    }
};

// --annot (process_vcf_fst.cpp:394-402, 558-562; Annotation, process_vcf_annotation_tools.h:177-327 and
// process_vcf_annotation_tools.cpp:84-109): the category of an included SNP -- "exon", "intron" or "other non-coding".
// Per scaffold the kept transcripts in file order with the four strings the lookup compares.  The reference's behaviour
// is kept where it is surprising: positions compare as strings; only the first line of a transcript counts as its
// exon; a partial transcript ("X-name" in column 5) is dropped together with the transcript before it; the last
// transcript of a scaffold is stored when another scaffold follows, so the file's last scaffold never is; a scaffold
// that comes back replaces its entry; two transcripts of one name share the later one's exon line.
struct GeneAnnotation {
    struct Transcript { string name, start, end, exonStart, exonEnd, strand; std::vector<std::pair<int, int>> exons; };
    std::map<string, std::vector<Transcript>> scaffolds;
    static string transcriptName(const string &geneColumn, bool &partial) {
        std::vector<string> parts = split(geneColumn, '-');
        if (parts.size() == 2) { partial = true; return parts[1]; }
        return parts.size() == 1 ? geneColumn : string();
    }
    void store(const string &scaffold, std::vector<std::vector<string>> &transcripts) {
        std::vector<Transcript> kept;
        std::map<string, size_t> lastOfName;
        for (size_t i = 0; i < transcripts.size(); ++i) {
            std::vector<string> first = split(transcripts[i].front(), '\t'), last = split(transcripts[i].back(), '\t');
            if (first.size() < 5 || last.size() < 3) die("annotation lines need five tab-separated columns");
            Transcript t;
            t.name = first[4];
            if (first[3] == "+") { t.start = first[1]; t.end = last[2]; } else { t.end = first[2]; t.start = last[1]; }
            t.exonStart = first[1]; t.exonEnd = first[2]; t.strand = first[3];
            for (const string &l : transcripts[i]) {
                std::vector<string> f = split(l, '\t');
                if (f.size() < 3) die("annotation lines need five tab-separated columns");
                t.exons.emplace_back(atoi(f[1].c_str()), atoi(f[2].c_str()));
            }
            lastOfName[t.name] = i;
            kept.push_back(t);
        }
        for (size_t i = 0; i < kept.size(); ++i) {
            const size_t l = lastOfName[kept[i].name];
            if (l != i) { kept[i].exonStart = kept[l].exonStart; kept[i].exonEnd = kept[l].exonEnd; kept[i].exons = kept[l].exons; }
        }
        scaffolds[scaffold] = kept;
        transcripts.clear();
    }
    void load(const string &path) {
        std::ifstream in(path.c_str());
        if (!in) die("could not open " + path);
        string line, scaffold, name;
        std::vector<std::vector<string>> transcripts;
        std::vector<string> current;
        bool partial = false, first = true;
        while (getline(in, line)) {
            std::vector<string> f = split(line, '\t');
            if (f.size() < 5) die("annotation lines need five tab-separated columns");
            bool linePartial = false;
            const string lineName = transcriptName(f[4], linePartial);
            if (first) { current.push_back(line); scaffold = f[0]; name = lineName; partial = linePartial; first = false; continue; }
            if (linePartial) partial = true;
            if (f[0] == scaffold && lineName == name) { current.push_back(line); continue; }
            if (!partial) transcripts.push_back(current);
            if (f[0] != scaffold) { store(scaffold, transcripts); scaffold = f[0]; }
            current.assign(1, line); name = lineName; partial = linePartial;
        }
        if (first) die("empty annotation file " + path);
    }
    const char *category(const string &scaffold, const string &locus) const {
        const char *c = "other non-coding";
        auto it = scaffolds.find(scaffold);
        if (it == scaffolds.end()) return c;
        for (const Transcript &t : it->second)
            if (locus >= t.start && locus <= t.end) {
                c = "intron";
                if (locus >= t.exonStart && locus <= t.exonEnd) return "exon";
            }
        return c;
    }
    // getSNPgeneDetails (process_vcf_annotation_tools.cpp:39-80), the lookup of FstGlobal --annot: integers here, every
    // line of the transcript, promoters 3 kb upstream by strand; the first transcript that holds the locus decides
    const char *geneDetails(const string &scaffold, int locus, const string *&gene) const {
        static const string none;
        gene = &none;
        auto it = scaffolds.find(scaffold);
        if (it == scaffolds.end()) return "nonCoding";
        for (const Transcript &t : it->second) {
            const int geneStart = atoi(t.start.c_str()), geneEnd = atoi(t.end.c_str());
            if (t.strand == "+") { if (locus >= geneStart - 3000 && locus < geneStart) { gene = &t.name; return "promoter"; } }
            else if (t.strand == "-") { if (locus > geneEnd && locus <= geneEnd + 3000) { gene = &t.name; return "promoter"; } }
            if (locus >= geneStart && locus <= geneEnd) {
                gene = &t.name;
                for (const auto &ex : t.exons) if (locus >= ex.first && locus <= ex.second) return "exon";
                return "intron";
            }
        }
        return "nonCoding";
    }
};

namespace fstopt {
string vcfFile, sampleSets, sampleNameFile, runName = "", popsFile, pairsArg, accessibleBed, annotFile, ancSets;
int windowSize = 0, windowStep = 0, physicalWindowSize = 10000, devices = 1;
double regAbove = 0;
}

const char *FST_USAGE =
    "Usage: evo fst [OPTIONS]\n"
    "Calculate Fst statistic from a vcf (GPU build: VCF mode only)\n"
    "       -h, --help                              display this help and exit\n"
    "       -n, --run-name                          run-name will be included in the output file name(s)\n"
    "       --vcf=VCF_FILE.vcf                      (required)\n"
    "       --sets=SAMPLE_SETS.txt                  (required) two lines, one sample set per line\n"
    "       -s SAMPLES.txt, --samples=SAMPLES.txt   (optional) sample identifiers to be used for the VCF file\n"
    "       -w SIZE,STEP --window=SIZE,STEP         (optional) sliding window of SIZE SNPs moving by STEP\n"
    "       --regions-above=minFst                  (optional, requires -w)\n"
    "       --pops=POPULATIONS.txt --pairs=all      (GPU build) instead of --sets: SAMPLE<TAB>POPULATION per line; every pair of\n"
    "                                               populations in one pass over the VCF, one RUN_<POP1>_<POP2>_w_<SIZE>_fst.txt per\n"
    "                                               pair = the _fst.txt file of `evo fst --sets` on those two populations (-w SIZE,STEP\n"
    "                                               with SIZE > 1 required; no heterozygosity / fixed-window files in this mode)\n"
    "       --devices=N                             (GPU build, all-pairs mode) N device batches in flight, one per GPU: consecutive\n"
    "                                               batches of the VCF are contiguous site shards, counted and walked at the same\n"
    "                                               time; the files are the same bytes as with one device (N > GPUs: the shards\n"
    "                                               share the GPUs in turn)\n"
    "       --physicalWindowSize=SIZE               (optional; default 10000bp)\n"
    "       --accessibleGenomeBED=BEDfile.bed       (optional) regions of the genome where SNPs could be called: the fixed-window\n"
    "                                               file reports and divides by the accessible bases of each window\n"
    "       --ancSets=ANCESTRAL_SAMPLE_SETS.txt     (optional) two sets of samples that form outgroup populations to the two populations:\n"
    "                                               writes RUN_NAMEancestralSNPs_fst.txt (needs AA=allele as the last INFO entry)\n"
    "       --annot=ANNOTATION.gffExtract           (optional) gene annotation (scaffold, start, end, strand, transcript per line):\n"
    "                                               writes RUN_NAMESNPcategory_fst.txt with the category, Fst and dxy of every SNP\n\n";

enum { OPT_VCF = 2, OPT_SETS, OPT_REG_ABOVE, OPT_PHYS, OPT_UNSUPPORTED, OPT_POPS, OPT_PAIRS, OPT_ACC_BED, OPT_ANNOT, OPT_ANC_SETS, OPT_AF, OPT_DEVICES };

void parseFstOptions(int argc, char **argv) {
    static const struct option longopts[] = {
        {"vcf", required_argument, NULL, OPT_VCF}, {"sets", required_argument, NULL, OPT_SETS},
        {"window", required_argument, NULL, 'w'}, {"regions-above", required_argument, NULL, OPT_REG_ABOVE},
        {"physicalWindowSize", required_argument, NULL, OPT_PHYS}, {"samples", required_argument, NULL, 's'},
        {"run-name", required_argument, NULL, 'n'}, {"help", no_argument, NULL, 'h'},
        {"pops", required_argument, NULL, OPT_POPS}, {"pairs", required_argument, NULL, OPT_PAIRS},
        {"devices", required_argument, NULL, OPT_DEVICES},
        {"ancSets", required_argument, NULL, OPT_ANC_SETS}, {"annot", required_argument, NULL, OPT_ANNOT},
        {"ms", required_argument, NULL, OPT_UNSUPPORTED}, {"eigen", required_argument, NULL, OPT_UNSUPPORTED},
        {"accessibleGenomeBED", required_argument, NULL, OPT_ACC_BED}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    std::vector<string> ws;
    for (int c; (c = getopt_long(argc, argv, "hn:s:w:", longopts, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case OPT_VCF: arg >> fstopt::vcfFile; break;
            case OPT_SETS: arg >> fstopt::sampleSets; break;
            case OPT_POPS: arg >> fstopt::popsFile; break;
            case OPT_PAIRS: arg >> fstopt::pairsArg; break;
            case OPT_DEVICES: arg >> fstopt::devices; break;
            case 'w':
                ws = split(arg.str(), ',');
                if (ws.size() < 2) die("-w needs SIZE,STEP");
                fstopt::windowSize = atoi(ws[0].c_str()); fstopt::windowStep = atoi(ws[1].c_str());
                break;
            case OPT_REG_ABOVE: arg >> fstopt::regAbove; break;
            case OPT_PHYS: arg >> fstopt::physicalWindowSize; break;
            case OPT_ACC_BED: arg >> fstopt::accessibleBed; break;
            case OPT_ANNOT: arg >> fstopt::annotFile; break;
            case OPT_ANC_SETS: arg >> fstopt::ancSets; break;
            case 's': arg >> fstopt::sampleNameFile; break;
            case 'n': arg >> fstopt::runName; break;
            case OPT_UNSUPPORTED: die("this option belongs to a mode the GPU build does not carry (see evo_main.cpp header)");
            case '?': dieFlag = true; break;
            case 'h': std::cout << FST_USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind > 0) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (!fstopt::pairsArg.empty() || !fstopt::popsFile.empty()) {
        if (fstopt::pairsArg != "all" || fstopt::popsFile.empty() || fstopt::vcfFile.empty() || !fstopt::sampleSets.empty()) {
            std::cerr << "The all-pairs mode needs --vcf, --pops=POPULATIONS.txt and --pairs=all (and no --sets)\n";
            dieFlag = true;
        } else if (fstopt::windowSize <= 1) {
            std::cerr << "The all-pairs mode needs -w SIZE,STEP with SIZE > 1\n";
            dieFlag = true;
        }
        if (fstopt::devices < 1 || fstopt::devices > 64) { std::cerr << "--devices takes 1 .. 64\n"; dieFlag = true; }
    } else if (fstopt::devices != 1) {
        std::cerr << "--devices belongs to the all-pairs mode (--pops, --pairs=all)\n";
        dieFlag = true;
    } else if (fstopt::vcfFile.empty() || fstopt::sampleSets.empty()) {
        std::cerr << "To calculate Fst statistics from a vcf, you need to use both the --vcf and the --sets options\n";
        dieFlag = true;
    }
    if (fstopt::regAbove > 0 && fstopt::windowSize == 0) {
        std::cerr << "The window size (-w option) needs to be set for --regions-above option to work (you can set -w 1,1 for per variant Fst)\n";
        dieFlag = true;
    }
    if (fstopt::windowSize > 0 && fstopt::windowStep <= 0) { std::cerr << "window STEP must be positive\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << FST_USAGE; exit(EXIT_FAILURE); }
}

// evo fst --pops=POPULATIONS.txt --pairs=all (GPU build): the SNP windows of getFstFromVCF (process_vcf_fst.cpp:638-692) for
// every pair of populations from ONE pass over the VCF.  The window sums are the device's segment sums (K1 + scan + pair walk,
// evofst_get_segments); the host only replays, per pair, the coordinate bookkeeping of the reference's emitter -- the
// "scaffold_0\t0" start string that swallows the first window elsewhere (:452, 682-686), the start / end strings -- and
// carries the partial segment from one device batch to the next.
int fstAllPairsMain() {
    using namespace fstopt;
    std::cerr << "Calculating Fst using variants from: " << vcfFile << std::endl;
    std::cerr << "Between every pair of the populations defined in: " << popsFile << std::endl;
    std::ifstream pf(popsFile.c_str());
    if (!pf.is_open()) die("could not open " + popsFile);
    std::map<string, std::vector<string>> popToIDs;
    std::map<string, string> idToPop;
    string line;
    while (getline(pf, line)) {
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());
        if (line.empty()) continue;
        std::vector<string> f = split(line, '\t');
        if (f.size() < 2) die("POPULATIONS.txt needs two tab-separated columns per line");
        popToIDs[f[1]].push_back(f[0]); idToPop[f[0]] = f[1];
    }
    std::vector<string> pops;
    std::map<string, int> popIndex;
    for (auto &kv : popToIDs) { popIndex[kv.first] = (int)pops.size(); pops.push_back(kv.first); }
    const int P = (int)pops.size();
    if (P < 2) die("the all-pairs mode needs at least two populations");
    std::vector<std::pair<int, int>> pairs;
    for (int i = 0; i < P; ++i) for (int j = i + 1; j < P; ++j) pairs.emplace_back(i, j);
    const size_t Q = pairs.size();
    const int W = windowSize, g = std::gcd(windowSize, windowStep), per = W / g;

    struct PairState {
        string path, out;                               // output file and the rows waiting to be appended to it
        string windowStartEnd = "scaffold_0\t0", windowMiddleVariant = "first\tWindow";   // fst.cpp:450-452
        int windowStart = 0, windowEnd = 0;
        int64_t base = 0;                               // included sites of the batches before this one
        double carry[3] = {0, 0, 0};                    // sums behind the last complete segment
        std::deque<std::array<double, 3>> recent;       // the last per - 1 segments {num, den, k} (overlapping windows)
        double nn = 1;                                  // n1 * n2
    };
    std::vector<PairState> st(Q);
    LineReader vcf(vcfFile);
    // --devices N: N consecutive batches of the VCF are N contiguous site shards, one per context (one per GPU while there
    // are GPUs; more shards than GPUs share them in turn).  What the shards exchange is what evo_amd/shard.py exchanges
    // between ranks, here between host threads of one process (the messages are a few KB: no RCCL):
    //   1. every shard counts (K1) and reports its included sites per pair  -> base offset of shard i = sites of all
    //      shards before it;
    //   2. every shard walks its windows from that offset and reports the sums behind its last complete segment, which the
    //      next shard's first segment continues.
    // The rows are written shard by shard in file order, so the files are the same bytes whatever N is.
    const int nDev = devices;
    struct Shard {
        Device dev;
        Batch batch;
        long long firstLine = 0;
        std::vector<int64_t> tot, firstSeg, nSeg, base;
        std::vector<double> sNum, sDen, sK, carry;
        std::vector<int32_t> sEnd;
    };
    std::vector<Shard> shard((size_t)nDev);
    for (Shard &sh : shard) { sh.tot.resize(Q); sh.firstSeg.resize(Q); sh.nSeg.resize(Q); sh.base.resize(Q); sh.carry.resize(Q * 4); }
    int nSamples = 0, cur = 0;
    long long totalVariantNumber = 0;
    bool haveHeader = false;
    std::ostringstream row;

    // the rows of one shard (all pairs), continuing the pairs' window state
    auto emit = [&](Shard &sh) {
        const Batch &batch = sh.batch;
        for (size_t q = 0; q < Q; ++q) {
            PairState &ps = st[q];
            row.str(""); row.clear();
            for (int64_t r = 0; r < sh.nSeg[q]; ++r) {
                std::array<double, 3> seg = {sh.sNum[(size_t)r * Q + q], sh.sDen[(size_t)r * Q + q], sh.sK[(size_t)r * Q + q]};
                if (r == 0) { seg[0] = ps.carry[0] + seg[0]; seg[1] = ps.carry[1] + seg[1]; seg[2] = ps.carry[2] + seg[2]; }
                const int64_t counted = (sh.firstSeg[q] + r + 1) * g;       // countedVariantNumber at the segment's last site
                ps.recent.push_back(seg);
                if ((int)ps.recent.size() > per) ps.recent.pop_front();
                if (counted % windowStep != 0 || counted < W) continue;      // fst.cpp:638
                double wn = 0, wd = 0, wk = 0;
                for (auto &x : ps.recent) { wn += x[0]; wd += x[1]; wk += x[2]; }
                if (wn != wn) wn = -__builtin_fabs(wn);                      // the sign of x86's default NaN ("-nan")
                if (wd != wd) wd = -__builtin_fabs(wd);
                const double windowFst = calculateFst(wn / (double)W, wd / (double)W);
                const double sumDxy = wk / ps.nn, windowDxy = sumDxy / (double)W;
                const int32_t e = sh.sEnd[(size_t)r * Q + q];
                const string &scaffold = batch.chrom[e], &loc = batch.pos[e];
                if (windowSize == windowStep) {                              // :640-650, 682-688
                    const size_t tab0 = ps.windowStartEnd.find('\t');
                    const bool same = ps.windowStartEnd.compare(0, tab0, scaffold) == 0 && tab0 == scaffold.size();
                    if (same) {
                        ps.windowStartEnd = ps.windowStartEnd + "\t" + loc;
                        ps.windowEnd = atoi(loc.c_str());
                        const double dxyInc = sumDxy / (double)(ps.windowEnd - ps.windowStart);
                        row << counted - W + 1 << "\t" << ps.windowStartEnd << "\t" << windowFst << "\t" << windowDxy << "\t" << dxyInc << "\t" << (size_t)W << std::endl;
                        ps.windowStartEnd = scaffold + "\t" + loc;
                        ps.windowStart = atoi(loc.c_str());
                    } else {
                        ps.windowStartEnd = scaffold + "\t0";
                        ps.windowStart = 0;
                    }
                } else {                                                     // :662, 690
                    row << counted - W + 1 << "\t" << ps.windowMiddleVariant << "\t" << windowFst << "\t" << windowDxy << "\t" << (size_t)W << std::endl;
                    ps.windowMiddleVariant = scaffold + "\t" + loc;
                }
            }
            if (sh.nSeg[q] > 0) { ps.carry[0] = sh.carry[q * 4]; ps.carry[1] = sh.carry[q * 4 + 1]; ps.carry[2] = sh.carry[q * 4 + 2]; }
            else { ps.carry[0] += sh.carry[q * 4]; ps.carry[1] += sh.carry[q * 4 + 1]; ps.carry[2] += sh.carry[q * 4 + 2]; }
            ps.base += sh.tot[q];
            const string text = row.str();
            if (!text.empty()) { std::ofstream f(ps.path.c_str(), std::ios::app); f << text; }
        }
    };
    // every shard that holds sites does `work` at the same time (one host thread per shard; a single shard runs inline)
    auto onShards = [&](int nUsed, const std::function<void(Shard &)> &work) {
        if (nUsed == 1) { work(shard[0]); return; }
        std::vector<std::thread> th;
        for (int i = 0; i < nUsed; ++i) th.emplace_back([&, i]() { work(shard[(size_t)i]); });
        for (std::thread &t : th) t.join();
    };
    auto flush = [&]() {
        int nUsed = 0;
        while (nUsed < nDev && shard[(size_t)nUsed].batch.n > 0) ++nUsed;
        if (nUsed == 0) return;
        onShards(nUsed, [&](Shard &sh) {                                     // exchange 1: included sites per pair
            sh.dev.load(sh.batch, EVOFST_MODE_FST, nSamples, std::vector<int32_t>(), sh.firstLine);
            check(sh.dev.ctx, evofst_get_included_totals(sh.dev.ctx, sh.tot.data()), "included_totals");
        });
        for (size_t q = 0; q < Q; ++q) {
            int64_t before = st[q].base;
            for (int i = 0; i < nUsed; ++i) { shard[(size_t)i].base[q] = before; before += shard[(size_t)i].tot[q]; }
        }
        onShards(nUsed, [&](Shard &sh) {                                     // exchange 2: the walk from the shard's offset
            check(sh.dev.ctx, evofst_set_base_offsets(sh.dev.ctx, sh.base.data()), "set_base_offsets");
            check(sh.dev.ctx, evofst_run_stats(sh.dev.ctx), "run_stats");
            const size_t cap = (size_t)evofst_max_segments(sh.dev.ctx);
            sh.sNum.resize(cap * Q); sh.sDen.resize(cap * Q); sh.sK.resize(cap * Q); sh.sEnd.resize(cap * Q);
            check(sh.dev.ctx, evofst_get_segments(sh.dev.ctx, sh.firstSeg.data(), sh.nSeg.data(), sh.sNum.data(), sh.sDen.data(), sh.sK.data(), sh.sEnd.data()), "segments");
            check(sh.dev.ctx, evofst_get_boundary(sh.dev.ctx, sh.carry.data()), "boundary");
        });
        for (int i = 0; i < nUsed; ++i) { emit(shard[(size_t)i]); shard[(size_t)i].batch.clear(); }
        cur = 0;
    };

    while (vcf.getline(line)) {
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames;
            nSamples = (int)fields.size() - 9;
            if (nSamples <= 0) die("VCF header has no sample columns");
            if (sampleNameFile.empty()) sampleNames.assign(fields.begin() + 9, fields.end());
            else { std::ifstream sf(sampleNameFile.c_str()); string l; while (getline(sf, l)) sampleNames.push_back(l); }
            if ((int)sampleNames.size() < nSamples) die("the -s file has fewer names than the VCF has sample columns");
            std::vector<int32_t> sp(nSamples, -1);
            std::vector<int> size(P, 0);
            for (int i = 0; i < nSamples; ++i) {
                auto it = idToPop.find(sampleNames[i]);
                if (it != idToPop.end()) { sp[i] = popIndex[it->second]; size[sp[i]]++; }
            }
            for (int k = 0; k < P; ++k) if (size[k] == 0) die("population \"" + pops[k] + "\" has no sample in the VCF");
            const int nGpu = std::max(1, evofst_device_count());
            for (int i = 0; i < nDev; ++i) {
                Device &dev = shard[(size_t)i].dev;
                dev.open(i % nGpu);
                check(dev.ctx, evofst_set_layout(dev.ctx, nSamples, P, sp.data()), "set_layout");
                check(dev.ctx, evofst_set_pairs(dev.ctx, nullptr, nullptr, 0), "set_pairs");
                check(dev.ctx, evofst_set_windows(dev.ctx, windowSize, windowStep), "set_windows");
                check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            }
            for (size_t q = 0; q < Q; ++q) {
                PairState &ps = st[q];
                ps.nn = (double)(2 * size[pairs[q].first]) * (double)(2 * size[pairs[q].second]);
                ps.path = runName + "_" + pops[pairs[q].first] + "_" + pops[pairs[q].second] + "_w_" + numToString(windowSize) + "_fst.txt";
                std::ofstream f(ps.path.c_str());                            // header, fst.cpp:510-518
                f << "var_num" << "\t" << "scaffold" << "\t" << "Start" << "\t" << "End" << "\t" << "Fst" << "\t" << "Dxy_onlyVariants" << "\t" << "Dxy_AllSites" << "\t" << "windowSize" << std::endl;
            }
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        Shard &sh = shard[(size_t)cur];
        if (sh.batch.n == 0) sh.firstLine = totalVariantNumber;
        sh.batch.add(line, totalVariantNumber);
        if (sh.batch.n == (size_t)kChunkSites && ++cur == nDev) flush();
        if (totalVariantNumber % 100000 == 0) std::cerr << totalVariantNumber << " variants processed..." << std::endl;
    }
    flush();
    for (Shard &sh : shard) sh.dev.close();
    return 0;
}

int fstMain(int argc, char **argv) {
    using namespace fstopt;
    parseFstOptions(argc, argv);
    if (!pairsArg.empty()) return fstAllPairsMain();
    std::cerr << "Calculating Fst using variants from: " << vcfFile << std::endl;
    std::cerr << "Between the two 'populations' defined in: " << sampleSets << std::endl;
    LineReader vcf(vcfFile);
    std::ifstream setsFile(sampleSets.c_str());
    if (!setsFile.is_open()) die("could not open " + sampleSets);

    std::ofstream regionsAboveFstFile; bool inRegAbove = false;
    if (regAbove > 0) regionsAboveFstFile.open((runName + "_w_" + numToString(windowSize) + "_fst_above" + numToString(regAbove) + ".txt").c_str());
    std::ofstream pFst((runName + "_w_" + numToString(windowSize) + "_fst.txt").c_str());
    AccessibleBed bed;
    const bool haveBed = !accessibleBed.empty();
    if (haveBed) { std::cerr << "Loading the accessible genome annotation" << std::endl; bed.load(accessibleBed); std::cerr << "Done" << std::endl; }
    GeneAnnotation annotation;                                       // fst.cpp:394-402
    std::ofstream snpCategoryFstFile;
    const bool haveAnnot = !annotFile.empty();
    if (haveAnnot) {
        annotation.load(annotFile);
        snpCategoryFstFile.open((runName + "SNPcategory_fst.txt").c_str());
        snpCategoryFstFile << "SNPcategory" << "\t" << "thisSNPFst" << "\t" << "thisSNPDxy" << "\t" << "scaffold" << "\t" << "position" << std::endl;
    }
    // --ancSets, fst.cpp:403-413: per included SNP the derived-allele frequency difference of the two sets and whether the two
    // ancestral sets segregate.  Host bookkeeping on the line's text, as in the reference (getFourSetVariantCounts,
    // process_vcf_utils.cpp:577-627, counts the characters '1' whatever the ALT column says, so the device's counts of
    // the Fst rules are not the same thing on lines with a '*' allele).
    std::ofstream ancSetsOutFile;
    std::vector<string> ancSet1, ancSet2;
    std::vector<size_t> ancSet1Loci, ancSet2Loci;
    const bool haveAnc = !ancSets.empty();
    if (haveAnc) {
        std::ifstream ancSetsFile(ancSets.c_str());
        if (!ancSetsFile.is_open()) die("could not open " + ancSets);
        ancSetsOutFile.open((runName + "ancestralSNPs_fst.txt").c_str());
        ancSetsOutFile << "scaffold" << "\t" << "position" << "\t" << "AncAllelePopulation" << "\t" << "Fst" << "\t" << "ancSet1_segregating" << "\t" << "ancSet2_segregating" << std::endl;
        string a1, a2;
        getline(ancSetsFile, a1); getline(ancSetsFile, a2);
        ancSet1 = split(a1, ','); ancSet2 = split(a2, ',');
        std::sort(ancSet1.begin(), ancSet1.end()); std::sort(ancSet2.begin(), ancSet2.end());
    }
    auto altCharacters = [](const std::vector<string> &fields, const std::vector<size_t> &loci) {
        int alt = 0;
        for (size_t i = 9; i < fields.size(); ++i) {
            if (std::find(loci.begin(), loci.end(), i - 9) == loci.end()) continue;
            if (fields[i].size() > 0 && fields[i][0] == '1') alt++;
            if (fields[i].size() > 2 && fields[i][2] == '1') alt++;
        }
        return alt;
    };
    std::ofstream fwFile((runName + "_w_" + numToString(physicalWindowSize) + "_fixedWindowStats.txt").c_str());
    std::ofstream pHet((runName + "_w_" + numToString(windowSize) + "_heterozygosity.txt").c_str());
    fwFile << "scaffold" << "\t" << "Start" << "\t" << "End" << "\t" << "Fst" << "\t" << "Dxy" << "\t" << "Set1_pi" << "\t" << "Set2_pi" << "\t" << "Accessible_bp" << "\t" << "Set1_VariantDensity" << "\t" << "Set2_VariantDensity" << std::endl;

    string set1String, set2String;
    getline(setsFile, set1String); getline(setsFile, set2String);
    std::vector<string> set1 = split(set1String, ','), set2 = split(set2String, ',');
    std::sort(set1.begin(), set1.end()); std::sort(set2.begin(), set2.end());

    // What the host keeps is coordinates and counters (process_vcf_fst.cpp:448-472): which included site a window starts and
    // ends at, scaffold and position strings, the accessible bases.  Every sum -- SNP windows (:638-692), physical windows
    // (:589-629), the genome-wide averages (:701-709) -- is formed on the device over site ranges this bookkeeping cuts
    // (evofst_get_pair_range_sums: the sites of a range added one after the other, as vector_average does), and a window a
    // batch boundary cuts in two goes on from the sums of its first part.
    int totalVariantNumber = 0, countedVariantNumber = 0;
    string windowMiddleVariant = "first\tWindow", windowStartEnd = "scaffold_0\t0";
    int windowStart = 0, windowEnd = 0, fixedWindowStart = 0;
    std::vector<size_t> set1Loci, set2Loci;
    typedef std::array<double, 12> Sums;   // {num, den, dxy, het1, het2, hetNei1, hetNei2, pi1, pi2, sites, sites pi1 != 0, sites pi2 != 0}
    enum RequestKind { SNP_WINDOW, FIXED_WINDOW, OPEN_SNP_WINDOW, OPEN_FIXED_WINDOW, WHOLE_BATCH };
    struct Request {
        RequestKind kind;
        int64_t lo = 0, hi = 0;            // sites of this batch
        bool carried = false; Sums init{}; // goes on from an earlier batch's part
        long long endCount = 0;            // SNP windows: countedVariantNumber at the last site
        bool equalStep = false, sameScaffold = false;
        string where, scaffold, regionStart, after;   // coordinates of the row; rows without sums that follow it
        int region = 0, acc = 0, fwStart = 0;
    };
    std::vector<Request> req;
    std::map<long long, Sums> openSnp;     // SNP windows begun in an earlier batch: end count -> sums so far
    Sums fwCarry{}; bool fwCarried = false;   // the physical window begun in an earlier batch
    Sums total{};                          // all included sites so far

    Device dev;
    Batch batch;
    int nSamples = 0, nVirtual = 0;
    std::vector<size_t> virt;          // virtual sample -> VCF column (set 1 entries, then set 2 entries)
    std::vector<int32_t> virt32;
    std::vector<double> num, den, dxy;   // per-site values: only for the outputs that are per site (-w 1,1, --annot, --ancSets)
    std::vector<uint8_t> inc;
    std::vector<int64_t> incSites, rlo, rhi;
    std::vector<double> rinit, rout;
    bool haveHeader = false;
    const bool perSite = windowSize == 1 || haveAnc || haveAnnot;
    const size_t W = (size_t)std::max(windowSize, 0);

    auto negNaN = [](Sums &a) { for (int k = 0; k < 9; ++k) if (a[k] != a[k]) a[k] = -__builtin_fabs(a[k]); };   // the sign of x86's default NaN ("-nan")

    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        const auto tFlush = std::chrono::steady_clock::now();
        struct Acc { std::chrono::steady_clock::time_point t; ~Acc() { g_timing.batches += Timing::since(t); } } acc{tFlush};
        dev.load(batch, EVOFST_MODE_FST, nSamples, virt32, totalVariantNumber - (long long)n + 1);
        inc.resize(n);
        if (perSite) { num.resize(n); den.resize(n); dxy.resize(n); }
        check(dev.ctx, evofst_get_site_stats(dev.ctx, 0, (int64_t)n, perSite ? num.data() : nullptr, perSite ? den.data() : nullptr,
                                             perSite ? dxy.data() : nullptr, inc.data()), "site_stats");
        if (perSite) { x86_nan_sign(num); x86_nan_sign(den); x86_nan_sign(dxy); }
        incSites.clear();
        for (size_t s = 0; s < n; ++s) if (inc[s]) incSites.push_back((int64_t)s);
        const long long counted0 = countedVariantNumber;
        // first site of a window that ends at count `endCount`, or -1 when it began in an earlier batch
        auto firstSiteOf = [&](long long endCount) -> int64_t {
            const long long first = endCount - (long long)W + 1;
            return first > counted0 ? incSites[(size_t)(first - counted0 - 1)] : -1;
        };
        req.clear();
        int64_t fwLo = 0;
        for (size_t s = 0; s < n; ++s) {
            if (!inc[s]) continue;                                  // filter, fst.cpp:543-544
            const string &scaffold = batch.chrom[s];
            const string &loc = batch.pos[s];
            countedVariantNumber++;
            if (haveAnc) {                                          // fst.cpp:563-588
                double thisSNPFst = num[s] / den[s];
                if (thisSNPFst < 0) thisSNPFst = 0;
                const std::vector<string> fields = split(batch.lines[s], '\t');
                const std::vector<string> info = split(fields[7], ';');
                const std::vector<string> lastInfo = info.empty() ? info : split(info.back(), '=');
                if (lastInfo.size() < 2) die("--ancSets needs AA=<allele> as the last INFO entry (line of " + scaffold + ":" + loc + ")");
                const string &AA = lastInfo[1];
                const int a1 = altCharacters(fields, set1Loci), a2 = altCharacters(fields, set2Loci);
                const int a3 = altCharacters(fields, ancSet1Loci), a4 = altCharacters(fields, ancSet2Loci);
                const double m1 = (double)(set1Loci.size() * 2), m2 = (double)(set2Loci.size() * 2);
                const double m3 = (double)(ancSet1Loci.size() * 2), m4 = (double)(ancSet2Loci.size() * 2);
                double f3, f4;
                if (AA == fields[3]) {
                    f3 = a3 / m3; f4 = a4 / m4;
                    ancSetsOutFile << scaffold << "\t" << fields[1] << "\t" << a1 / m1 - a2 / m2 << "\t" << thisSNPFst << "\t";
                } else if (AA == fields[4]) {
                    f3 = (double)((int)m3 - a3) / m3; f4 = (double)((int)m4 - a4) / m4;
                    ancSetsOutFile << scaffold << "\t" << fields[1] << "\t" << (double)((int)m1 - a1) / m1 - (double)((int)m2 - a2) / m2 << "\t" << thisSNPFst << "\t";
                } else {
                    f3 = a3 / m3; f4 = a4 / m4;
                    ancSetsOutFile << scaffold << "\t" << fields[1] << "\t" << "-888" << "\t" << thisSNPFst << "\t";
                }
                ancSetsOutFile << (((f3 > 0) & (f3 < 1)) ? "1" : "0") << "\t" << (((f4 > 0) & (f4 < 1)) ? "1" : "0") << std::endl;
            }
            if (haveAnnot)                                          // fst.cpp:558-562
                snpCategoryFstFile << annotation.category(scaffold, loc) << "\t" << num[s] / den[s] << "\t" << dxy[s] << "\t" << scaffold << "\t" << loc << std::endl;
            // physical windows, fst.cpp:589-629: this site already belongs to the window it closes; a window is the included
            // sites since the last one closed (or since the scaffold changed, which drops this site too)
            const size_t tab0 = windowStartEnd.find('\t');
            const size_t len0 = tab0 == string::npos ? windowStartEnd.size() : tab0;
            if (len0 == scaffold.size() && windowStartEnd.compare(0, len0, scaffold) == 0) {
                const int here = atoi(loc.c_str());
                if (here > (fixedWindowStart + physicalWindowSize)) {
                    Request r; r.kind = FIXED_WINDOW; r.lo = fwLo; r.hi = (int64_t)s + 1; r.carried = fwCarried; r.init = fwCarry;
                    r.scaffold = scaffold; r.fwStart = fixedWindowStart;
                    r.acc = haveBed ? bed.basesIn(scaffold, fixedWindowStart, fixedWindowStart + physicalWindowSize) : physicalWindowSize;
                    std::ostringstream empty;                    // the windows without a site between this one and the next
                    int without = 0;
                    while (here > (fixedWindowStart + physicalWindowSize)) {
                        if (without > 0) {
                            const int acc0 = haveBed ? bed.basesIn(scaffold, fixedWindowStart, fixedWindowStart + physicalWindowSize) : physicalWindowSize;
                            empty << scaffold << "\t" << fixedWindowStart << "\t" << fixedWindowStart + physicalWindowSize << "\t" << "NA" << "\t" << 0 << "\t" << 0 << "\t" << 0 << "\t" << acc0 << "\t" << 0 << "\t" << 0 << std::endl;
                        }
                        fixedWindowStart = fixedWindowStart + physicalWindowSize;
                        without++;
                    }
                    r.after = empty.str();
                    req.push_back(std::move(r));
                    fwLo = (int64_t)s + 1; fwCarried = false;
                }
            } else {
                fixedWindowStart = 0;
                fwLo = (int64_t)s + 1; fwCarried = false;
            }
            if (windowSize == 1) {                                   // fst.cpp:633-636
                double Fst = num[s] / den[s];
                if (Fst < 0) Fst = 0;
                pFst << countedVariantNumber << "\t" << scaffold + "\t" + loc << "\t" << Fst << "\t" << dxy[s] << std::endl;
            } else if ((windowSize > 0) && (countedVariantNumber % windowStep == 0) && countedVariantNumber >= windowSize) {
                Request r; r.kind = SNP_WINDOW; r.endCount = countedVariantNumber; r.hi = (int64_t)s + 1;
                const int64_t first = firstSiteOf(r.endCount);
                if (first >= 0) r.lo = first;
                else { r.lo = 0; r.carried = true; r.init = openSnp.at(r.endCount); }
                r.equalStep = windowSize == windowStep;
                if (r.equalStep) {                                   // :640-650, 676-688: the window's coordinates
                    const size_t t0 = windowStartEnd.find('\t');
                    r.sameScaffold = windowStartEnd.compare(0, t0, scaffold) == 0 && t0 == scaffold.size();
                    if (r.sameScaffold) {
                        r.scaffold = scaffold; r.regionStart = split(windowStartEnd, '\t')[1];
                        windowStartEnd = windowStartEnd + "\t" + loc;
                        windowEnd = atoi(loc.c_str());
                        r.where = windowStartEnd; r.region = windowEnd - windowStart;
                        windowStartEnd = scaffold + "\t" + loc;
                        windowStart = atoi(loc.c_str());
                    } else {
                        windowStartEnd = scaffold + "\t0";
                        windowStart = 0;
                    }
                } else {                                             // :662, 690-691
                    r.where = windowMiddleVariant;
                    windowMiddleVariant = scaffold + "\t" + loc;
                }
                req.push_back(std::move(r));
            }
        }
        // the windows that go on in the next batch, the physical window that is open, and the genome-wide sums
        if (windowSize > 1) {
            for (long long e = (countedVariantNumber / windowStep + 1) * (long long)windowStep; e - (long long)W + 1 <= countedVariantNumber; e += windowStep) {
                if (e < (long long)W) continue;
                Request r; r.kind = OPEN_SNP_WINDOW; r.endCount = e; r.hi = (int64_t)n;
                const int64_t first = firstSiteOf(e);
                if (first >= 0) r.lo = first;
                else { r.lo = 0; r.carried = true; r.init = openSnp.at(e); }
                req.push_back(std::move(r));
            }
        }
        { Request r; r.kind = OPEN_FIXED_WINDOW; r.lo = fwLo; r.hi = (int64_t)n; r.carried = fwCarried; r.init = fwCarry; req.push_back(std::move(r)); }
        { Request r; r.kind = WHOLE_BATCH; r.lo = 0; r.hi = (int64_t)n; r.carried = true; r.init = total; req.push_back(std::move(r)); }
        const size_t nr = req.size();
        rlo.resize(nr); rhi.resize(nr); rinit.assign(nr * 12, 0.0); rout.resize(nr * 12);
        for (size_t i = 0; i < nr; ++i) {
            rlo[i] = req[i].lo; rhi[i] = req[i].hi;
            if (req[i].carried) std::copy(req[i].init.begin(), req[i].init.end(), rinit.begin() + (std::ptrdiff_t)(i * 12));
        }
        check(dev.ctx, evofst_get_pair_range_sums(dev.ctx, 0, (int64_t)nr, rlo.data(), rhi.data(), rinit.data(), rout.data()), "pair_range_sums");
        std::map<long long, Sums> stillOpen;
        for (size_t i = 0; i < nr; ++i) {
            const Request &r = req[i];
            Sums a;
            std::copy(rout.begin() + (std::ptrdiff_t)(i * 12), rout.begin() + (std::ptrdiff_t)(i * 12 + 12), a.begin());
            if (r.kind == OPEN_SNP_WINDOW) { stillOpen[r.endCount] = a; continue; }
            if (r.kind == OPEN_FIXED_WINDOW) { fwCarry = a; fwCarried = true; continue; }
            if (r.kind == WHOLE_BATCH) { total = a; continue; }
            negNaN(a);
            if (r.kind == FIXED_WINDOW) {                            // fst.cpp:598-612
                const double fDxy = a[2] / (double)r.acc;
                const double fFst = calculateFst(a[0] / a[9], a[1] / a[9]);
                const double fPi1 = a[7] / (double)r.acc, fPi2 = a[8] / (double)r.acc;
                const int nv1 = (int)a[10], nv2 = (int)a[11];
                fwFile << r.scaffold << "\t" << r.fwStart << "\t" << r.fwStart + physicalWindowSize << "\t" << fFst << "\t" << fDxy << "\t" << fPi1 << "\t" << fPi2 << "\t" << r.acc << "\t" << (double)nv1 / r.acc << "\t" << (double)nv2 / r.acc << std::endl;
                fwFile << r.after;
                continue;
            }
            // SNP window, fst.cpp:638-692
            const double windowFst = calculateFst(a[0] / (double)W, a[1] / (double)W);
            const double windowDxy = a[2] / (double)W;
            const double wS1 = a[3] / (double)W, wS2 = a[4] / (double)W, wN1 = a[5] / (double)W, wN2 = a[6] / (double)W;
            if (r.equalStep) {
                if (!r.sameScaffold) continue;
                const double dxyInc = a[2] / (double)r.region;
                pFst << r.endCount - windowSize + 1 << "\t" << r.where << "\t" << windowFst << "\t" << windowDxy << "\t" << dxyInc << "\t" << W << std::endl;
                if (regAbove > 0) {
                    if (windowFst >= regAbove && !inRegAbove) { inRegAbove = true; regionsAboveFstFile << r.scaffold << "\t" << r.regionStart << "\t"; }
                    else if (windowFst < regAbove && inRegAbove) { inRegAbove = false; regionsAboveFstFile << r.regionStart << std::endl; }
                }
                const double wPi1 = a[7] / (double)r.region, wPi2 = a[8] / (double)r.region;
                pHet << r.where << "\t" << wS1 << "\t" << wS2 << "\t" << wN1 << "\t" << wN2 << "\t" << wPi1 << "\t" << wPi2 << std::endl;
            } else {
                pFst << r.endCount - windowSize + 1 << "\t" << r.where << "\t" << windowFst << "\t" << windowDxy << "\t" << W << std::endl;
                pHet << r.where << "\t" << wS1 << "\t" << wS2 << "\t" << wN1 << "\t" << wN2 << std::endl;
            }
        }
        openSnp.swap(stillOpen);
        batch.clear();
    };

    string line;
    while (vcf.getline(line)) {
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames;
            nSamples = (int)fields.size() - 9;
            if (nSamples <= 0) die("VCF header has no sample columns");
            if (sampleNameFile.empty()) sampleNames.assign(fields.begin() + 9, fields.end());
            else { std::ifstream sf(sampleNameFile.c_str()); string l; while (getline(sf, l)) sampleNames.push_back(l); }
            set1Loci = locateSet(sampleNames, set1); set2Loci = locateSet(sampleNames, set2);
            if (set1Loci.empty() || set2Loci.empty()) die("a sample set has no samples in the VCF");
            if (haveAnc) { ancSet1Loci = locateSet(sampleNames, ancSet1); ancSet2Loci = locateSet(sampleNames, ancSet2); }   // fst.cpp:497-505
            for (size_t i : set1Loci) virt.push_back(i);
            for (size_t i : set2Loci) virt.push_back(i);
            for (size_t i : virt) if ((int)i >= nSamples) die("sample index beyond the VCF columns (check the -s file)");
            nVirtual = (int)virt.size();
            virt32.assign(virt.begin(), virt.end());
            std::vector<int32_t> sp(nVirtual, 1);
            for (size_t i = 0; i < set1Loci.size(); ++i) sp[i] = 0;
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, nVirtual, 2, sp.data()), "set_layout");
            int32_t pi0 = 0, pj0 = 1;
            check(dev.ctx, evofst_set_pairs(dev.ctx, &pi0, &pj0, 1), "set_pairs");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            if (windowSize > 0) {                                    // fst.cpp:507-520
                if (windowSize == windowStep) {
                    if (windowSize == 1) {
                        pFst << "var_num" << "\t" << "scaffold" << "\t" << "Position" << "\t" << "Fst" << "\t" << "Dxy_thisVariant" << std::endl;
                    } else {
                        pHet << "scaffold" << "\t" << "Start" << "\t" << "End" << "\t" << "Set1_heterozygosity" << "\t" << "Set2_heterozygosity" << "\t" << "Set1_heterozygosity_Nei" << "\t" << "Set2_heterozygosity_Nei" << "\t" << "Set1_nucleotideDiversity_pi" << "\t" << "Set2_nucleotideDiversity_pi" << std::endl;
                        pFst << "var_num" << "\t" << "scaffold" << "\t" << "Start" << "\t" << "End" << "\t" << "Fst" << "\t" << "Dxy_onlyVariants" << "\t" << "Dxy_AllSites" << "\t" << "windowSize" << std::endl;
                    }
                    if (regAbove > 0) regionsAboveFstFile << "scaffold" << "\t" << "Start" << "\t" << "End" << std::endl;
                } else {
                    pHet << "Middle_SNP_position" << "\t" << "Set1_heterozygosity" << "\t" << "Set2_heterozygosity" << "\t" << "Set1_heterozygosity_Nei" << "\t" << "Set2_heterozygosity_Nei" << "\t" << "Set1_nucleotideDiversity_pi" << "\t" << "Set2_nucleotideDiversity_pi" << std::endl;
                    pFst << "var_num" << "\t" << "scaffold" << "\t" << "Start" << "\t" << "End" << "\t" << "Fst" << "\t" << "Dxy_onlyVariants" << "\t" << "Dxy_AllSites" << "\t" << "windowSize" << std::endl;
                }
            }
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
        if (totalVariantNumber % 100000 == 0) std::cerr << totalVariantNumber << " variants processed..." << std::endl;
    }
    flush();
    negNaN(total);
    const double nCounted = (double)countedVariantNumber;                                          // fst.cpp:701-709
    double Fst = calculateFst(total[0] / nCounted, total[1] / nCounted);
    double oS1 = total[3] / nCounted, oS2 = total[4] / nCounted, oN1 = total[5] / nCounted, oN2 = total[6] / nCounted;
    std::cerr << "Fst: " << Fst << std::endl;
    std::cerr << "Heterozygosities: " << "\tS1:" << oS1 << "\tS2:" << oS2 << "\tNei1:" << oN1 << "\tNei2" << oN2 << std::endl;
    pHet << "#Heterozygosities: " << "\tS1:" << oS1 << "\tS2:" << oS2 << "\tNei1:" << oN1 << "\tNei2" << oN2 << std::endl;
    dev.close();
    return 0;
}

// ====================================================================================== FstGlobal

namespace gopt {
string vcfFile, setsFile, runName = "", annotFile;
int fixedWindowSize = 10000, windowSize = 20, windowStep = 10;
}

const char *FSTGLOBAL_USAGE =
    "Usage: evo FstGlobal [OPTIONS] INPUT_FILE.vcf POPULATIONS.txt\n"
    "Calculate the Fst statistic for each population from the POPULATIONS.txt file, when compared against the rest of the dataset\n"
    "       -h, --help                              display this help and exit\n"
    "       -f, --fixedW sizeKb                     fixed window size (default: 10kb)\n"
    "       -w SIZE,STEP --window=SIZE,STEP         sliding window: SIZE SNPs, moving by STEP (default: 20,10)\n"
    "       -n, --run-name=RUN_NAME                 run-name will be included in the output file name\n"
    "       --annot=ANNOTATION.gffExtract           (optional) gene annotation: writes RUN_FstGlobalGenes_RUN_SIZE_STEP.txt, one row per gene\n\n";

void parseFstGlobalOptions(int argc, char **argv) {
    static const struct option longopts[] = {{"fixedW", required_argument, NULL, 'f'}, {"window", required_argument, NULL, 'w'},
                                             {"annot", required_argument, NULL, OPT_ANNOT}, {"help", no_argument, NULL, 'h'},
                                             {"run-name", required_argument, NULL, 'n'}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    std::vector<string> ws;
    for (int c; (c = getopt_long(argc, argv, "hw:n:f:", longopts, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case '?': dieFlag = true; break;
            case 'f': arg >> gopt::fixedWindowSize; break;
            case 'w':
                ws = split(arg.str(), ',');
                if (ws.size() < 2) die("-w needs SIZE,STEP");
                gopt::windowSize = atoi(ws[0].c_str()); gopt::windowStep = atoi(ws[1].c_str());
                break;
            case 'n': arg >> gopt::runName; break;
            case OPT_ANNOT: arg >> gopt::annotFile; break;
            case 'h': std::cout << FSTGLOBAL_USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 2) { std::cerr << "missing arguments\n"; dieFlag = true; }
    else if (argc - optind > 2) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (gopt::windowSize <= 0 || gopt::windowStep <= 0) { std::cerr << "window SIZE and STEP must be positive\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << FSTGLOBAL_USAGE; exit(EXIT_FAILURE); }
    gopt::vcfFile = argv[optind++]; gopt::setsFile = argv[optind++];
}

int FstGlobalMain(int argc, char **argv) {
    using namespace gopt;
    parseFstGlobalOptions(argc, argv);
    LineReader vcf(vcfFile);
    std::ifstream setsF(setsFile.c_str());
    if (!setsF.is_open()) die("could not open " + setsFile);
    std::map<string, std::vector<string>> popToIDsMap;
    std::map<string, string> IDsToPopMap;
    string line;
    while (getline(setsF, line)) {                                   // FstAgainstAll.cpp:83-89
        std::vector<string> ID_Pop = split(line, '\t');
        if (ID_Pop.size() < 2) die("POPULATIONS.txt needs two tab-separated columns per line");
        popToIDsMap[ID_Pop[1]].push_back(ID_Pop[0]);
        IDsToPopMap[ID_Pop[0]] = ID_Pop[1];
    }
    std::vector<string> populations, populationsToUse;
    std::map<string, int> popIndex;
    for (auto &kv : popToIDsMap) {
        popIndex[kv.first] = (int)populations.size();
        populations.push_back(kv.first);
        if (kv.first != "Outgroup" && kv.first != "xxx") populationsToUse.push_back(kv.first);
    }
    std::cerr << "There are " << populations.size() << " populations " << std::endl;
    if (populationsToUse.empty()) die("no population to use");

    std::ofstream outFile((runName + "_FstGlobal_" + numToString(windowSize) + "_" + numToString(windowStep) + ".txt").c_str());
    outFile << "chr\tpos1\tpos2\tFstGlobal"; for (size_t i = 0; i < populationsToUse.size(); i++) outFile << "\t" << populationsToUse[i];
    outFile << std::endl;
    std::ofstream outFW((runName + "_FstGlobal_FW" + numToString(fixedWindowSize) + ".txt").c_str());
    outFW << "chr\tpos1\tpos2\tnSNPs"; for (size_t i = 0; i < populationsToUse.size(); i++) outFile << "\t" << populationsToUse[i];   // sic: FstAgainstAll.cpp:104
    outFW << std::endl;

    const size_t nUse = populationsToUse.size();
    // --annot (FstAgainstAll.cpp:66-72, 107-116, 215-219, 267-297): per gene the kept SNPs in its exons, introns and promoter
    // and each population's Fst over them.  A row is written when a SNP of another gene arrives, after that SNP's own
    // values have joined the sums -- the reference's order, kept.
    GeneAnnotation annotation;
    const bool haveAnnot = !annotFile.empty();
    std::ofstream outFileGenes;
    std::vector<std::vector<double>> geneNum(nUse * 3), geneDen(nUse * 3);   // [population][exon | intron | promoter]
    string currentGene, previousGene;
    if (haveAnnot) {
        annotation.load(annotFile);
        outFileGenes.open((runName + "_FstGlobalGenes_" + runName + "_" + numToString(windowSize) + "_" + numToString(windowStep) + ".txt").c_str());
        outFileGenes << "gene\t" << "numSNPsExons\tnumSNPsIntrons\tnumSNPs3kbPromoter";
        for (size_t i = 0; i < nUse; i++) outFileGenes << "\t" << populationsToUse[i] << "_exons" << "\t" << populationsToUse[i] << "_wIntrons" << "\t" << populationsToUse[i] << "_promoter";
        outFileGenes << std::endl;
    }
    std::vector<double> numAll, denAll;
    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    int totalVariantNumber = 0;
    int64_t usedBefore = 0, keptBefore = 0;          // used / kept sites of the batches before this one (:189, :204)
    // the last SIZE kept sites seen so far, oldest first: what numDeque / denomDeque / coordDeque hold (:119-122, 237-240);
    // they go to the device as the next batch's halo
    struct HaloSite { string coord; std::vector<double> num, den; };
    std::deque<HaloSite> halo;

    Device dev;
    Batch batch;
    int nSamples = 0;
    std::vector<double> num, den, haloNum, haloDen, rowNum, rowDen;
    std::vector<int32_t> haloTag, rowStart, rowEnd;
    std::vector<uint8_t> kept;
    bool haveHeader = false;

    // The sliding windows are the device's: evofst_get_global_windows sums each deque oldest -> newest exactly as
    // vector_average does (bit-identical rows); the host keeps the coordinates, the fixed-window table (which the reference
    // fills with zeros, :222-235) and hands the deque contents from batch to batch.
    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        dev.load(batch, EVOFST_MODE_GLOBAL, nSamples, std::vector<int32_t>(), totalVariantNumber - (long long)n + 1);
        int64_t tot[2] = {0, 0}, base[2] = {usedBefore, keptBefore};
        check(dev.ctx, evofst_get_included_totals(dev.ctx, tot), "included_totals");
        check(dev.ctx, evofst_set_base_offsets(dev.ctx, base), "set_base_offsets");
        const int hn = (int)halo.size();
        haloTag.resize(hn); haloNum.resize((size_t)hn * nUse); haloDen.resize((size_t)hn * nUse);
        for (int e = 0; e < hn; ++e) {
            haloTag[e] = -2 - e;
            std::copy(halo[e].num.begin(), halo[e].num.end(), haloNum.begin() + (size_t)e * nUse);
            std::copy(halo[e].den.begin(), halo[e].den.end(), haloDen.begin() + (size_t)e * nUse);
        }
        check(dev.ctx, evofst_set_global_halo(dev.ctx, hn, haloTag.data(), haloNum.data(), haloDen.data()), "set_global_halo");
        check(dev.ctx, evofst_run_stats(dev.ctx), "run_stats");
        const int64_t cap = evofst_max_global_rows(dev.ctx);
        rowStart.resize(cap); rowEnd.resize(cap); rowNum.resize((size_t)cap * nUse); rowDen.resize((size_t)cap * nUse);
        int64_t nRows = 0;
        check(dev.ctx, evofst_get_global_windows(dev.ctx, &nRows, rowStart.data(), rowEnd.data(), rowNum.data(), rowDen.data()), "global_windows");
        rowNum.resize((size_t)nRows * nUse); rowDen.resize((size_t)nRows * nUse);
        x86_nan_sign(rowNum); x86_nan_sign(rowDen);
        for (int64_t r = 0; r < nRows; ++r) {                        // :276-282, 298-300
            const int32_t st = rowStart[r], en = rowEnd[r];
            const string &first = st >= 0 ? batch.pos[st] : (st == -1 ? string("0") : halo[(size_t)(-2 - st)].coord);
            outFile << batch.chrom[en] << "\t" << first << "\t" << batch.pos[en] << "\t" << "N";
            for (size_t i = 0; i != nUse; i++) {
                const double an = rowNum[(size_t)r * nUse + i], ad = rowDen[(size_t)r * nUse + i];
                double Fst = an / ad; if (Fst < 0 || ad == 0) Fst = 0;
                outFile << "\t" << Fst;
            }
            outFile << std::endl;
        }
        // kept sites in order: the fixed-window table, and the deque contents behind this batch
        kept.resize(n);
        if (haveAnnot) {
            numAll.resize(nUse * n); denAll.resize(nUse * n);
            check(dev.ctx, evofst_get_global_site_stats(dev.ctx, 0, (int64_t)n, numAll.data(), denAll.data(), kept.data()), "global_site_stats");
            x86_nan_sign(numAll); x86_nan_sign(denAll);
        } else
        check(dev.ctx, evofst_get_global_site_stats(dev.ctx, 0, (int64_t)n, nullptr, nullptr, kept.data()), "global_site_stats");
        size_t tailFrom = n; int tailKept = 0;
        for (size_t s = n; s-- > 0 && tailKept < windowSize;)
            if (kept[s] && !(batch.flags[s] & EVOFST_SITE_SKIP)) { tailFrom = s; ++tailKept; }
        for (size_t s = 0; s < n; ++s) {
            if ((batch.flags[s] & EVOFST_SITE_SKIP) || !kept[s]) continue;
            const string &chr = batch.chrom[s];
            double coordDouble; { std::stringstream ss(batch.pos[s]); ss >> coordDouble; }
            if (haveAnnot) {                                         // :215-219, 267-275, 283-297
                const string *gene;
                const string category = annotation.geneDetails(chr, atoi(batch.pos[s].c_str()), gene);
                if (!gene->empty()) {
                    currentGene = *gene;
                    if (previousGene.empty()) previousGene = currentGene;
                    const int cat = category == "exon" ? 0 : category == "intron" ? 1 : category == "promoter" ? 2 : -1;
                    if (cat >= 0)
                        for (size_t i = 0; i < nUse; ++i) { geneNum[i * 3 + cat].push_back(numAll[i * n + s]); geneDen[i * 3 + cat].push_back(denAll[i * n + s]); }
                }
                if (!previousGene.empty() && currentGene != previousGene) {
                    const int nExonSNPs = (int)geneNum[0].size(), nIntronSNPs = (int)geneNum[1].size(), nPromoterSNPs = (int)geneNum[2].size();
                    const int counts[3] = {nExonSNPs, nIntronSNPs, nPromoterSNPs};
                    outFileGenes << previousGene << "\t" << nExonSNPs << "\t" << nIntronSNPs << "\t" << nPromoterSNPs;
                    for (size_t i = 0; i < nUse; i++)
                        for (int c3 = 0; c3 < 3; ++c3) {
                            double f = counts[c3] > 0 ? vector_average(geneNum[i * 3 + c3]) / vector_average(geneDen[i * 3 + c3]) : 0;
                            if (f < 0) f = 0;
                            outFileGenes << "\t" << f;
                            geneNum[i * 3 + c3].clear(); geneDen[i * 3 + c3].clear();
                        }
                    outFileGenes << std::endl;
                    previousGene = currentGene;
                }
            }
            if (coordDouble > currentWindowEnd || coordDouble < currentWindowStart) {   // :222-235
                outFW << chr << "\t" << currentWindowStart << "\t" << currentWindowEnd << "\t" << 0;
                for (size_t i = 0; i != nUse; i++) { double FstFW = 0; outFW << "\t" << FstFW; }
                outFW << std::endl;
                if (coordDouble > currentWindowEnd) { currentWindowStart = currentWindowStart + fixedWindowSize; currentWindowEnd = currentWindowEnd + fixedWindowSize; }
                else if (coordDouble < currentWindowStart) { currentWindowStart = 0; currentWindowEnd = 0 + fixedWindowSize; }
            }
        }
        if (tailKept > 0) {
            const size_t m = n - tailFrom;
            num.resize(nUse * m); den.resize(nUse * m);
            check(dev.ctx, evofst_get_global_site_stats(dev.ctx, (int64_t)tailFrom, (int64_t)m, num.data(), den.data(), nullptr), "global_site_stats");
            for (size_t s = tailFrom; s < n; ++s) {
                if ((batch.flags[s] & EVOFST_SITE_SKIP) || !kept[s]) continue;
                HaloSite h; h.coord = batch.pos[s]; h.num.resize(nUse); h.den.resize(nUse);
                for (size_t i = 0; i < nUse; ++i) { h.num[i] = num[i * m + (s - tailFrom)]; h.den[i] = den[i * m + (s - tailFrom)]; }
                halo.push_back(std::move(h));
                if ((int)halo.size() > windowSize) halo.pop_front();
            }
        }
        usedBefore += tot[0]; keptBefore += tot[1];
        batch.clear();
    };

    while (vcf.getline(line)) {
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());   // :146
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames(fields.begin() + std::min<size_t>(9, fields.size()), fields.end());
            nSamples = (int)sampleNames.size();
            if (nSamples <= 0) die("VCF header has no sample columns");
            std::vector<int32_t> sp(nSamples, -1);
            for (int i = 0; i < nSamples; ++i) {                     // :153-155
                auto it = IDsToPopMap.find(sampleNames[i]);
                if (it != IDsToPopMap.end()) sp[i] = popIndex[it->second];
            }
            std::vector<size_t> allSizes;                            // setSizes, utils.h:77 (ALL pops, sorted order)
            for (auto &kv : popToIDsMap) {
                std::vector<size_t> spPos = locateSet(sampleNames, kv.second);
                if (spPos.empty()) { std::cerr << "Did not find any samples in the VCF for \"" << kv.first << "\"" << std::endl; exit(EXIT_FAILURE); }
                allSizes.push_back(spPos.size());
            }
            std::vector<int32_t> use; std::vector<int64_t> setSizes;
            for (size_t u = 0; u < nUse; ++u) { use.push_back(popIndex[populationsToUse[u]]); setSizes.push_back((int64_t)allSizes[u]); }  // c->setSizes[i], :253
            // a population whose samples were all re-assigned by a later line of the file keeps an entry
            // in the reference's maps but never gets an allele; give it no layout slot problems:
            for (size_t k = 0; k < populations.size(); ++k)
                if (std::find(sp.begin(), sp.end(), (int32_t)k) == sp.end()) die("population \"" + populations[k] + "\" has no sample column of its own");
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, nSamples, (int)populations.size(), sp.data()), "set_layout");
            check(dev.ctx, evofst_set_against_all(dev.ctx, use.data(), (int)nUse, setSizes.data()), "set_against_all");
            check(dev.ctx, evofst_set_windows(dev.ctx, windowSize, windowStep), "set_windows");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        if (totalVariantNumber % 10000 == 0) std::cerr << "Processed " << totalVariantNumber << " variants" << std::endl;
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
    }
    flush();
    dev.close();
    return 0;
}

// ===================================================================================== alleleFreq

namespace afopt { string vcfFile, setsFile, runName = "out"; }

const char *AF_USAGE =
    "Usage: evo alleleFreq [OPTIONS] INPUT_FILE.vcf POPULATIONS.txt\n"
    "Calculate the Allele Frequencies per population/species from a VCF \n"
    "       -h, --help                              display this help and exit\n"
    "       -n, --run-name                          run-name will be included in the output file name\n\n";

string stripExtension(const string &filename) {                      // process_vcf_utils.cpp:164-171
    size_t suffixPos = filename.find_last_of('.');
    return suffixPos == string::npos ? filename : filename.substr(0, suffixPos);
}

// evo alleleFreq (AFmain, evo_AlleleFeq.cpp:46-166); --use-genotype-probabilities only on VCFs whose FORMAT is GT alone
// (where the reference warns and falls back to the called genotypes): the per-population
// counts come from K1, the frequency is alt / called alleles, or -1 for a population without a called allele.
int AFmain(int argc, char **argv) {
    using namespace afopt;
    static const struct option longopts[] = {{"help", no_argument, NULL, 'h'}, {"run-name", required_argument, NULL, 'n'},
                                             {"use-genotype-probabilities", no_argument, NULL, 'g'}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    bool useGenotypeProbabilities = false;
    for (int c; (c = getopt_long(argc, argv, "hn:", longopts, NULL)) != -1;) {   // :28: no 'g' among the short options
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case '?': dieFlag = true; break;
            case 'n': arg >> runName; break;
            case 'g': useGenotypeProbabilities = true; break;
            case 'h': std::cout << AF_USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 2) { std::cerr << "missing arguments\n"; dieFlag = true; }
    else if (argc - optind > 2) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << AF_USAGE; exit(EXIT_FAILURE); }
    vcfFile = argv[optind++]; setsFile = argv[optind++];

    LineReader vcf(vcfFile);
    std::ifstream setsF(setsFile.c_str());
    if (!setsF.is_open()) die("could not open " + setsFile);
    std::map<string, std::vector<string>> popToIDsMap;
    std::map<string, string> IDsToPopMap;
    string line;
    while (getline(setsF, line)) {                                   // AlleleFeq.cpp:62-68
        std::vector<string> ID_Pop = split(line, '\t');
        if (ID_Pop.size() < 2) die("POPULATIONS.txt needs two tab-separated columns per line");
        popToIDsMap[ID_Pop[1]].push_back(ID_Pop[0]);
        IDsToPopMap[ID_Pop[0]] = ID_Pop[1];
    }
    std::vector<string> populations;
    std::map<string, int> popIndex;
    for (auto &kv : popToIDsMap) { popIndex[kv.first] = (int)populations.size(); populations.push_back(kv.first); }
    std::cerr << "There are " << populations.size() << " populations " << std::endl;
    std::ofstream out((stripExtension(setsFile) + "_" + runName + "_AF" + ".txt").c_str());   // :82
    if (!out.is_open()) die("could not open the output file");

    const size_t P = populations.size();
    Device dev;
    Batch batch;
    std::vector<string> refs, alts;
    int nSamples = 0, totalVariantNumber = 0;
    std::vector<int32_t> altc, nmc;
    bool haveHeader = false;
    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        dev.load(batch, EVOFST_MODE_GLOBAL, nSamples, std::vector<int32_t>(), totalVariantNumber - (long long)n + 1);
        altc.resize(n * P); nmc.resize(n * P);
        check(dev.ctx, evofst_get_counts(dev.ctx, 0, (int64_t)n, altc.data(), nmc.data()), "get_counts");
        for (size_t s = 0; s < n; ++s) {
            if (batch.flags[s] & EVOFST_SITE_SKIP) continue;         // :124-127
            out << batch.chrom[s] << "\t" << batch.pos[s] << "\t" << refs[s] << "\t" << alts[s];
            if (useGenotypeProbabilities) {
                // :141-150 with a FORMAT column of GT alone (checkForGenotypeLikelihoodsOrProbabilities, utils.cpp:1122-1124):
                // the warning, the option is dropped for the rest of the run, and this one line still prints the
                // likelihood-based frequencies, which nothing has set: -1 for every population (utils.h:76).  Lines that
                // do carry GP / GL / PL take a per-sample floating-point path that is another input format than genotype
                // bit planes: not carried.
                const string &l = batch.lines[s];
                size_t a = 0;
                for (int f = 0; f < 8 && a != string::npos; ++f) { a = l.find('\t', a); if (a != string::npos) ++a; }
                const size_t b = a == string::npos ? string::npos : l.find('\t', a);
                if (b == string::npos) die("malformed VCF line");
                if (l.find(':', a) < b) die("--use-genotype-probabilities on a VCF with more FORMAT tags than GT is not carried by the GPU build");
                std::cerr << "WARNING: Could not fing genotype likelihoods/probabilities (GP, PL, or GL fields) for variant at " << batch.chrom[s] << " " << batch.pos[s] << std::endl;
                std::cerr << "WARNING: Did you really mean to use the -g option? Reverting to using called genotypes." << std::endl;
                useGenotypeProbabilities = false;
                for (size_t k = 0; k < P; ++k) out << "\t" << -1.0;
                out << "\n";
                continue;
            }
            for (size_t k = 0; k < P; ++k) {                          // utils.cpp:56-58, setAAFs init -1.0 (utils.h:75)
                double aaf = -1.0;
                if (nmc[s * P + k] > 0) aaf = (double)altc[s * P + k] / nmc[s * P + k];
                out << "\t" << aaf;
            }
            out << "\n";
        }
        batch.clear(); refs.clear(); alts.clear();
    };
    while (vcf.getline(line)) {
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());   // :85
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames(fields.begin() + std::min<size_t>(9, fields.size()), fields.end());
            nSamples = (int)sampleNames.size();
            if (nSamples <= 0) die("VCF header has no sample columns");
            std::vector<int32_t> sp(nSamples, -1);
            for (int i = 0; i < nSamples; ++i) {
                auto it = IDsToPopMap.find(sampleNames[i]);
                if (it != IDsToPopMap.end()) sp[i] = popIndex[it->second];
            }
            for (auto &kv : popToIDsMap)                              // :100-109
                if (locateSet(sampleNames, kv.second).empty()) { std::cerr << "Did not find any samples in the VCF for \"" << kv.first << "\"" << std::endl; exit(EXIT_FAILURE); }
            for (size_t k = 0; k < P; ++k)
                if (std::find(sp.begin(), sp.end(), (int32_t)k) == sp.end()) die("population \"" + populations[k] + "\" has no sample column of its own");
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, nSamples, (int)P, sp.data()), "set_layout");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        if (totalVariantNumber == 1) {                                // :114-120
            out << "chr" << "\t" << "coord" << "\t" << "ref" << "\t" << "alt";
            for (auto &kv : popToIDsMap) out << "\t" << kv.first;
            out << "\n";
        }
        if (totalVariantNumber % 10000 == 0) std::cerr << "Processed " << totalVariantNumber << " variants" << std::endl;
        {   // REF and ALT as written (columns 4 and 5)
            size_t p0 = 0;
            for (int f = 0; f < 3; ++f) { p0 = line.find('\t', p0); if (p0 == string::npos) die("malformed VCF line " + numToString(totalVariantNumber)); ++p0; }
            const size_t p1 = line.find('\t', p0);
            const size_t p2 = p1 == string::npos ? string::npos : line.find('\t', p1 + 1);
            if (p2 == string::npos) die("malformed VCF line " + numToString(totalVariantNumber));
            refs.emplace_back(line, p0, p1 - p0); alts.emplace_back(line, p1 + 1, p2 - p1 - 1);
        }
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
    }
    flush();
    out.flush();
    dev.close();
    return 0;
}

// ====================================================================================== PBS, ABS

namespace bopt {
string vcfFile, setsFile, tuplesFile, runName = "", annotFile;
int fixedWindowSize = 10000, windowSize = 20, windowStep = 10;
bool allowIndels = false, af = false;
}

const char *PBS_USAGE =
    "Usage: evo PBS [OPTIONS] INPUT_FILE.vcf POPULATIONS.txt PBS_trios.txt\n"
    "Population branch statistic for each trio of populations named in PBS_trios.txt (three tab-separated names per line);\n"
    "writes POP1_POP2_POP3_PBS_RUN_SIZE_STEP.txt and POP1_POP2_POP3_PBS_RUN_FWsize.txt per trio\n"
    "       -h, --help                              display this help and exit\n"
    "       -f, --fixedW size                       fixed (physical) window size (default: 10000)\n"
    "       -w SIZE,STEP --window=SIZE,STEP         sliding window: SIZE SNPs, moving by STEP (default: 20,10)\n"
    "       -n, --run-name=RUN_NAME                 run-name will be included in the output file names\n"
    "       --annot=ANNOTATION.gffExtract           (optional) gene annotation: writes P1_P2_P3_PBSGenes_RUN_SIZE_STEP.txt, one row per gene\n"
    "       --af=ARG                                (optional) as in the reference: leaves POPULATIONS_AF.txt, empty (ARG is not read)\n"
    "       -i, --allow-indels-and-multiallelics    (optional) also use indels, and multiallelics with the second allele taken as REF\n\n";

const char *ABS_USAGE =
    "Usage: evo ABS [OPTIONS] INPUT_FILE.vcf POPULATIONS.txt ABS_quartets.txt\n"
    "Ancestral branch statistic for each quartet POP_W POP_X POP_Y POP_Z named in ABS_quartets.txt (four tab-separated\n"
    "names per line); writes POPW_POPX_POPY_POPZ_ABS_RUN_SIZE_STEP.txt per quartet\n"
    "       -h, --help                              display this help and exit\n"
    "       -w SIZE,STEP --window=SIZE,STEP         sliding window: SIZE SNPs, moving by STEP (default: 20,10)\n"
    "       -n, --run-name=RUN_NAME                 run-name will be included in the output file names\n\n";

// Sign of a NaN value as the reference's x86 build writes it ("nan" / "-nan").  A population with a single called
// allele makes its variance term 0/0 and every branch length T = -log(1 - Fst) that uses it NaN; which sign reaches the
// output depends on WHICH branch lengths are NaN (the compiler folds the negations of -log() into the sums, and an x86
// addition keeps its first NaN operand).  The tables are what g++ -O2 makes of evo_PBS.cpp:69-98 and evo_ABS.cpp:58-106
// on this image; tests/test_oracle.py pins the oracle, sign included, against those builds, and the CLI tests compare
// this program's text with the oracle's.
//   PBS  index: T12 | T13 << 1 | T23 << 2;                                   bit x set = value x is "-nan"
//   ABS  index: TWX | TWY << 1 | TWZ << 2 | TXY << 3 | TXZ << 4 | TYZ << 5;  bit 0 = ABS, bit 1 = ABSalt
const uint8_t kPbsNanNegative[8] = {0, 4, 3, 0, 7, 4, 3, 0};
const uint8_t kAbsNanNegative[64] = {0, 0, 0, 0, 0, 0, 0, 0, 3, 2, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 2, 3, 2, 0, 0, 0, 0,
                                     3, 0, 3, 0, 1, 0, 0, 0, 3, 2, 3, 0, 1, 0, 1, 0, 3, 0, 3, 0, 1, 0, 1, 0, 0, 2, 3, 2, 1, 0, 1, 0};

// evo PBS (PBSmain, evo_PBS.cpp:126-369) and evo ABS (ABSmain, evo_ABS.cpp:109-293): the reference's two mains are one loop over a different per-site statistic.  Per-site values of
// every trio / quartet come from the device; the sliding windows (and, for PBS, the fixed windows) are replayed on the
// host in file order.  Line citations are PBSmain's unless marked.
int branchMain(const int arity, int argc, char **argv) {
    using namespace bopt;
    const bool pbsMode = arity == 3;
    const int nVal = pbsMode ? 3 : 2;
    const char *usage = pbsMode ? PBS_USAGE : ABS_USAGE;
    static const struct option longopts[] = {{"fixedW", required_argument, NULL, 'f'}, {"window", required_argument, NULL, 'w'},
                                             {"annot", required_argument, NULL, OPT_ANNOT}, {"af", required_argument, NULL, OPT_AF},
                                             {"allow-indels-and-multiallelics", no_argument, NULL, 'i'},
                                             {"help", no_argument, NULL, 'h'}, {"run-name", required_argument, NULL, 'n'}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    std::vector<string> ws;
    for (int c; (c = getopt_long(argc, argv, pbsMode ? "hw:n:f:i" : "hw:n:", pbsMode ? longopts : longopts + 1, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case '?': dieFlag = true; break;
            case 'f': arg >> fixedWindowSize; break;
            case 'w':
                ws = split(arg.str(), ',');
                if (ws.size() < 2) die("-w needs SIZE,STEP");
                windowSize = atoi(ws[0].c_str()); windowStep = atoi(ws[1].c_str());
                break;
            case 'n': arg >> runName; break;
            case 'i': if (!pbsMode) { dieFlag = true; break; } allowIndels = true; break;
            case OPT_ANNOT: arg >> annotFile; break;
            case OPT_AF: if (!pbsMode) { dieFlag = true; break; } af = true; break;                           // :48, 393: the argument is not read
            case 'h': std::cout << usage; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 3) { std::cerr << "missing arguments\n"; dieFlag = true; }
    else if (argc - optind > 3) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (windowSize <= 0 || windowStep <= 0) { std::cerr << "window SIZE and STEP must be positive\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << usage; exit(EXIT_FAILURE); }
    vcfFile = argv[optind++]; setsFile = argv[optind++]; tuplesFile = argv[optind++];

    LineReader vcf(vcfFile);
    std::ifstream setsF(setsFile.c_str()), tuplesF(tuplesFile.c_str());
    if (!setsF.is_open()) die("could not open " + setsFile);
    if (!tuplesF.is_open()) die("could not open " + tuplesFile);
    std::map<string, std::vector<string>> popToIDsMap;
    std::map<string, string> IDsToPopMap;
    string line;
    while (getline(setsF, line)) {                                   // :148-154
        std::vector<string> ID_Pop = split(line, '\t');
        if (ID_Pop.size() < 2) die("POPULATIONS.txt needs two tab-separated columns per line");
        popToIDsMap[ID_Pop[1]].push_back(ID_Pop[0]);
        IDsToPopMap[ID_Pop[0]] = ID_Pop[1];
    }
    std::vector<string> populations;
    std::map<string, int> popIndex;
    for (auto &kv : popToIDsMap) { popIndex[kv.first] = (int)populations.size(); populations.push_back(kv.first); }
    std::cerr << "There are " << populations.size() << " populations " << std::endl;
    const size_t P = populations.size();

    // --annot (:130-136): PBS writes a row per gene and trio (:176-181, 282-286, 333-361); ABS loads the file and does
    // nothing with it (its gene rows are commented out in the reference, ABS.cpp:267-285)
    GeneAnnotation annotation;
    const bool haveAnnot = !annotFile.empty();
    if (haveAnnot) annotation.load(annotFile);
    std::vector<std::ofstream> outFilesGenes;
    string currentGene, previousGene;
    std::vector<std::ofstream> outFiles, outFilesFixedWindow;        // :163-183, ABS.cpp:146-156
    std::vector<int32_t> tuples;
    while (getline(tuplesF, line)) {
        std::vector<string> names = split(line, '\t');
        if ((int)names.size() != arity)
            die(string(pbsMode ? "PBS_trios.txt needs three" : "ABS_quartets.txt needs four") + " tab-separated population names per line");
        string stem;
        for (const string &name : names) {
            auto it = popIndex.find(name);
            if (it == popIndex.end()) die("population \"" + name + "\" of " + tuplesFile + " is not in " + setsFile);
            tuples.push_back(it->second);
            stem += name + "_";
        }
        stem += string(pbsMode ? "PBS_" : "ABS_") + runName;
        outFiles.emplace_back((stem + "_" + numToString(windowSize) + "_" + numToString(windowStep) + ".txt").c_str());
        if (pbsMode) {
            outFilesFixedWindow.emplace_back((stem + "_FW" + numToString(fixedWindowSize) + ".txt").c_str());
            outFiles.back() << "chr\twStart\twEnd\t" << names[0] << "\t" << names[1] << "\t" << names[2] << std::endl;
            outFilesFixedWindow.back() << "chr\twStart\twEnd\t" << names[0] << "\t" << names[1] << "\t" << names[2] << "\t"
                                       << "nFwSNPs1" << "\t" << "nFwSNPs2" << "\t" << "nFwSNPs3" << std::endl;
            if (haveAnnot) {
                outFilesGenes.emplace_back((names[0] + "_" + names[1] + "_" + names[2] + "_PBSGenes_" + runName + "_" + numToString(windowSize) + "_" + numToString(windowStep) + ".txt").c_str());
                outFilesGenes.back() << "gene\t" << "numSNPsExons\tnumSNPsIntrons\tnumSNPs3kbPromoter\t" << names[0] << "_exons\t" << names[1] << "_exons\t" << names[2] << "_exons\t"
                                     << names[0] << "_wIntrons\t" << names[1] << "_wIntrons\t" << names[2] << "_wIntrons\t" << names[0] << "_promoter\t" << names[1] << "_promoter\t" << names[2] << "_promoter" << std::endl;
            }
        } else {
            outFiles.back() << "chr\tpos1\tpos2\tABS\tABSalt" << std::endl;
        }
    }
    const size_t T = outFiles.size();
    if (T == 0) die("no line in " + tuplesFile);

    std::deque<double> initDeq(windowSize, 0.0);                     // :186-192: nVal value deques + one of coordinates
    std::vector<std::vector<std::deque<double>>> results(T, std::vector<std::deque<double>>(nVal + 1, initDeq));
    std::vector<std::vector<std::vector<double>>> fixedWindowResults(T, std::vector<std::vector<double>>(3));
    std::vector<std::vector<std::vector<double>>> geneResults(T, std::vector<std::vector<double>>(9));   // exon x3, intron x3, promoter x3
    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    std::vector<int> usedVars(T, 0);
    int totalVariantNumber = 0;

    Device dev;
    Batch batch;
    int nSamples = 0;
    std::vector<double> val;
    std::vector<uint8_t> used;
    std::vector<int32_t> altc, nmc;
    bool haveHeader = false, afMade = false;

    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        dev.load(batch, EVOFST_MODE_GLOBAL, nSamples, std::vector<int32_t>(), totalVariantNumber - (long long)n + 1, allowIndels);
        val.resize(T * nVal * n); used.resize(T * n);
        if (pbsMode) check(dev.ctx, evofst_get_pbs_site_stats(dev.ctx, 0, (int64_t)n, (int)T, tuples.data(), val.data(), used.data()), "pbs_site_stats");
        else check(dev.ctx, evofst_get_abs_site_stats(dev.ctx, 0, (int64_t)n, (int)T, tuples.data(), val.data(), used.data()), "abs_site_stats");
        bool anyNan = false;
        for (double x : val) if (x != x) { anyNan = true; break; }
        if (anyNan) {                                                // give NaNs the sign the reference prints
            altc.resize(n * P); nmc.resize(n * P);
            check(dev.ctx, evofst_get_counts(dev.ctx, 0, (int64_t)n, altc.data(), nmc.data()), "get_counts");
            // is T = -log(1 - Fst) of populations a, b NaN, when na1 / nb1 say whether the two variance terms the
            // reference subtracts for this pair come from a single-allele population
            auto tNan = [&](size_t s, int a, int b, bool na1, bool nb1) {
                const int ca = altc[s * P + a], na = nmc[s * P + a], cb = altc[s * P + b], nb = nmc[s * P + b];
                if ((ca == 0 && cb == 0) || (ca == na && cb == nb)) return 0;   // Fst forced to 0 (:82-84, ABS.cpp:77-82)
                return (na1 || nb1) ? 1 : 0;
            };
            for (size_t t = 0; t < T; ++t)
                for (size_t s = 0; s < n; ++s) {
                    double *v = &val[t * nVal * n + s];
                    bool any = false;
                    for (int x = 0; x < nVal; ++x) any |= v[x * n] != v[x * n];
                    if (!any) continue;
                    const int32_t *q = &tuples[arity * t];
                    bool one[4];
                    for (int x = 0; x < arity; ++x) one[x] = nmc[s * P + q[x]] == 1;
                    int neg;
                    if (pbsMode) {
                        neg = kPbsNanNegative[tNan(s, q[0], q[1], one[0], one[1]) | tNan(s, q[0], q[2], one[0], one[2]) << 1 |
                                              tNan(s, q[1], q[2], one[1], one[2]) << 2];
                    } else {
                        // ABS.cpp:66: W's pairs subtract X's variance term and X's pairs subtract W's
                        const bool fW = one[1], fX = one[0], fY = one[2], fZ = one[3];
                        neg = kAbsNanNegative[tNan(s, q[0], q[1], fW, fX) | tNan(s, q[0], q[2], fW, fY) << 1 | tNan(s, q[0], q[3], fW, fZ) << 2 |
                                              tNan(s, q[1], q[2], fX, fY) << 3 | tNan(s, q[1], q[3], fX, fZ) << 4 | tNan(s, q[2], q[3], fY, fZ) << 5];
                    }
                    for (int x = 0; x < nVal; ++x)
                        if (v[x * n] != v[x * n]) v[x * n] = (neg >> x & 1) ? -__builtin_fabs(v[x * n]) : __builtin_fabs(v[x * n]);
                }
        }
        for (size_t s = 0; s < n; ++s) {
            if (batch.flags[s] & EVOFST_SITE_SKIP) continue;         // :248-256, ABS.cpp:217-221
            const string &chr = batch.chrom[s];
            const string &coord = batch.pos[s];
            double coordDouble; { std::stringstream ss(coord); ss >> coordDouble; }
            // :264-271 --af: at every used line the reference opens stripExtension(setsFile) + "_AF.txt" anew, which
            // empties it, on a stream that is never closed, flushed or deleted: the line it writes stays in a buffer
            // that is lost at exit.  What a run leaves is the empty file; made once here, at the first used line.
            if (pbsMode && af && !afMade) { std::ofstream((stripExtension(setsFile) + "_AF" + ".txt").c_str()); afMade = true; }
            int geneCategory = -1;                                   // :282-286
            if (pbsMode && haveAnnot) {
                const string *gene;
                const string category = annotation.geneDetails(chr, atoi(coord.c_str()), gene);
                if (!gene->empty()) {
                    currentGene = *gene;
                    if (previousGene.empty()) previousGene = currentGene;
                    geneCategory = category == "exon" ? 0 : category == "intron" ? 1 : category == "promoter" ? 2 : -1;
                }
            }
            if (pbsMode && (coordDouble > currentWindowEnd || coordDouble < currentWindowStart)) {   // :289-303 (PBS only)
                for (size_t i = 0; i != T; i++) {
                    std::vector<std::vector<double>> &fw = fixedWindowResults[i];
                    const int nFwSNPs1 = (int)fw[0].size(), nFwSNPs2 = (int)fw[1].size(), nFwSNPs3 = (int)fw[2].size();
                    double PBSfw1 = 0; if (nFwSNPs1 > 0) PBSfw1 = vector_average(fw[0]);
                    double PBSfw2 = 0; if (nFwSNPs2 > 0) PBSfw2 = vector_average(fw[1]);
                    double PBSfw3 = 0; if (nFwSNPs3 > 0) PBSfw3 = vector_average(fw[2]);
                    outFilesFixedWindow[i] << chr << "\t" << currentWindowStart << "\t" << currentWindowEnd << "\t" << PBSfw1 << "\t" << PBSfw2
                                           << "\t" << PBSfw3 << "\t" << nFwSNPs1 << "\t" << nFwSNPs2 << "\t" << nFwSNPs3 << std::endl;
                    fw[0].clear(); fw[1].clear(); fw[2].clear();
                }
                if (coordDouble > currentWindowEnd) { currentWindowStart = currentWindowStart + fixedWindowSize; currentWindowEnd = currentWindowEnd + fixedWindowSize; }
                else if (coordDouble < currentWindowStart) { currentWindowStart = 0; currentWindowEnd = 0 + fixedWindowSize; }
            }
            for (size_t i = 0; i != T; i++) {                        // :309-347, ABS.cpp:244-278
                if (!used[i * n + s]) continue;
                usedVars[i]++;
                std::vector<std::deque<double>> &r = results[i];
                for (int x = 0; x < nVal; ++x) {
                    const double v = val[(i * nVal + x) * n + s];
                    r[x].push_back(v); r[x].pop_front();
                    if (pbsMode) fixedWindowResults[i][x].push_back(v);
                    if (geneCategory >= 0) geneResults[i][3 * geneCategory + x].push_back(v);   // :333-341
                }
                r[nVal].push_back(coordDouble); r[nVal].pop_front();
                if (usedVars[i] > windowSize && (usedVars[i] % windowStep == 0)) {
                    double av[3];
                    for (int x = 0; x < nVal; ++x) {
                        double sum = 0;
                        for (size_t k = 0; k < r[x].size(); k++) sum += r[x][k];
                        av[x] = (double)sum / (double)r[x].size();
                    }
                    if (pbsMode) outFiles[i] << chr << "\t" << (int)r[3][0] << "\t" << coord << "\t" << av[0] << "\t" << av[1] << "\t" << av[2] << std::endl;
                    else outFiles[i] << chr << "\t" << r[2][0] << "\t" << coord << "\t" << av[0] << "\t" << av[1] << std::endl;   // ABS.cpp:275: a double
                }
            }
            if (pbsMode && haveAnnot && !previousGene.empty() && currentGene != previousGene) {   // :348-361
                for (size_t i = 0; i != T; i++) {
                    std::vector<std::vector<double>> &g = geneResults[i];
                    const int counts[3] = {(int)g[0].size(), (int)g[3].size(), (int)g[6].size()};
                    outFilesGenes[i] << previousGene << "\t" << counts[0] << "\t" << counts[1] << "\t" << counts[2];
                    for (int y = 0; y < 9; ++y) { const double a = counts[y / 3] > 0 ? vector_average(g[y]) : 0; outFilesGenes[i] << "\t" << a; }
                    outFilesGenes[i] << std::endl;
                    for (int y = 0; y < 9; ++y) g[y].clear();
                }
                previousGene = currentGene;
            }
        }
        batch.clear();
    };

    while (vcf.getline(line)) {
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());   // :210
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames(fields.begin() + std::min<size_t>(9, fields.size()), fields.end());
            nSamples = (int)sampleNames.size();
            if (nSamples <= 0) die("VCF header has no sample columns");
            std::vector<int32_t> sp(nSamples, -1);
            for (int i = 0; i < nSamples; ++i) {                     // :217-219
                auto it = IDsToPopMap.find(sampleNames[i]);
                if (it != IDsToPopMap.end()) sp[i] = popIndex[it->second];
            }
            for (auto &kv : popToIDsMap)                              // :222-232
                if (locateSet(sampleNames, kv.second).empty()) { std::cerr << "Did not find any samples in the VCF for \"" << kv.first << "\"" << std::endl; exit(EXIT_FAILURE); }
            for (size_t k = 0; k < P; ++k)
                if (std::find(sp.begin(), sp.end(), (int32_t)k) == sp.end()) die("population \"" + populations[k] + "\" has no sample column of its own");
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, nSamples, (int)P, sp.data()), "set_layout");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        if (totalVariantNumber % 10000 == 0) std::cerr << "Processed " << totalVariantNumber << " variants" << std::endl;
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
    }
    flush();
    dev.close();
    return 0;
}

// ================================================================================== DistOutgroups

namespace dopt {
string vcfFile, setsFile, outgroupFile, ingroupFile, runName = "", accessibleBed = "";
int fixedWindowSize = 10000;
bool allowIndels = false;
}

const char *DISTOUT_USAGE =
    "Usage: evo DistOutgroups [OPTIONS] INPUT_FILE.vcf POPULATIONS.txt OUTGROUPS.txt INGROUPS.txt\n"
    "Mean genetic distance (dxy) between each outgroup population (one name per line in OUTGROUPS.txt) and each ingroup\n"
    "population (one per line in INGROUPS.txt) in fixed windows; writes OUTGROUP_DIST_RUN_FWsize.txt per outgroup\n"
    "       -h, --help                              display this help and exit\n"
    "       -f, --fixedW size                       fixed (physical) window size (default: 10000)\n"
    "       -n, --run-name=RUN_NAME                 run-name will be included in the output file names\n"
    "       --accessibleGenomeBED=BEDfile.bed           (optional) a bed file specifying the regions of the genome where we could call SNPs\n"
    "       -i, --allow-indels-and-multiallelics    (optional) also use indels, and multiallelics with the second allele taken as REF\n\n";

// evo DistOutgroups (DistOutMain, evo_distanceToOutgroups.cpp:68-306).  The host
// replays the fixed-window state machine on the coordinates alone to find where rows are emitted, the device adds up
// the per-site dxy of every (outgroup, ingroup) pair between two emissions in site order, the host turns the sums
// into rows.  A window left open at the end of a device batch is continued by the next batch (carry_first).
int DistOutMain(int argc, char **argv) {
    using namespace dopt;
    static const struct option longopts[] = {{"fixedW", required_argument, NULL, 'f'}, {"help", no_argument, NULL, 'h'},
                                             {"run-name", required_argument, NULL, 'n'},
                                             {"allow-indels-and-multiallelics", no_argument, NULL, 'i'},
                                             {"accessibleGenomeBED", required_argument, NULL, OPT_ACC_BED}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    for (int c; (c = getopt_long(argc, argv, "hn:f:i", longopts, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case '?': dieFlag = true; break;
            case 'f': arg >> fixedWindowSize; break;
            case 'n': arg >> runName; break;
            case 'i': allowIndels = true; break;
            case OPT_ACC_BED: arg >> accessibleBed; break;
            case 'h': std::cout << DISTOUT_USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 4) { std::cerr << "missing arguments\n"; dieFlag = true; }
    else if (argc - optind > 4) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << DISTOUT_USAGE; exit(EXIT_FAILURE); }
    vcfFile = argv[optind++]; setsFile = argv[optind++]; outgroupFile = argv[optind++]; ingroupFile = argv[optind++];

    LineReader vcf(vcfFile);
    std::ifstream setsF(setsFile.c_str()), outgroupF(outgroupFile.c_str()), ingroupF(ingroupFile.c_str());
    if (!setsF.is_open()) die("could not open " + setsFile);
    if (!outgroupF.is_open()) die("could not open " + outgroupFile);
    if (!ingroupF.is_open()) die("could not open " + ingroupFile);
    std::map<string, std::vector<string>> popToIDsMap;
    std::map<string, string> IDsToPopMap;
    string line;
    while (getline(setsF, line)) {                                   // :88-94
        std::vector<string> ID_Pop = split(line, '\t');
        if (ID_Pop.size() < 2) die("POPULATIONS.txt needs two tab-separated columns per line");
        popToIDsMap[ID_Pop[1]].push_back(ID_Pop[0]);
        IDsToPopMap[ID_Pop[0]] = ID_Pop[1];
    }
    std::vector<string> populations, ingroups, outgroups;
    std::map<string, int> popIndex;
    for (auto &kv : popToIDsMap) { popIndex[kv.first] = (int)populations.size(); populations.push_back(kv.first); }
    std::cerr << "There are " << populations.size() << " populations " << std::endl;
    const size_t P = populations.size();
    auto indexOf = [&](const string &name, const string &file) {
        auto it = popIndex.find(name);
        if (it == popIndex.end()) die("population \"" + name + "\" of " + file + " is not in " + setsFile);
        return (int32_t)it->second;
    };
    std::vector<int32_t> inIdx, outIdx;
    while (getline(ingroupF, line)) { ingroups.push_back(line); inIdx.push_back(indexOf(line, ingroupFile)); }   // :104-106
    std::vector<std::ofstream> outFilesFixedWindow;                  // :109-117
    auto printRow = [](std::ofstream &f, const auto &v) {            // print_vector, process_vcf_utils.h:338-347
        for (size_t k = 0; k < v.size(); k++) {
            if (k == v.size() - 1) f << v[k] << std::endl;
            else f << v[k] << '\t';
        }
    };
    while (getline(outgroupF, line)) {
        outIdx.push_back(indexOf(line, outgroupFile));
        outFilesFixedWindow.emplace_back((line + "_DIST_" + runName + "_FW" + numToString(fixedWindowSize) + ".txt").c_str());
        outFilesFixedWindow.back() << "chr\twStart\twEnd\tSNPs_used\tSNPs_missing\tAccessibleSizeBP" << "\t";
        printRow(outFilesFixedWindow.back(), ingroups);
        outgroups.push_back(line);
    }
    const size_t nOut = outgroups.size(), nIn = ingroups.size();
    if (nOut == 0) die("no outgroup in " + outgroupFile);
    AccessibleBed bed;                                               // :121-127
    const bool haveBed = !accessibleBed.empty();
    if (haveBed) { std::cerr << "Loading the accessible genome annotation" << std::endl; bed.load(accessibleBed); std::cerr << "Done" << std::endl; }

    int currentWindowStart = 0, currentWindowEnd = currentWindowStart + fixedWindowSize;
    int totalVariantNumber = 0;
    Device dev;
    Batch batch;
    int nSamples = 0;
    bool haveHeader = false;
    struct Emission { int64_t site; string chr; int wStart, wEnd, accessible; };
    std::vector<Emission> rows;
    std::vector<int64_t> winOff;
    std::vector<double> sum, carrySum(nOut * nIn, 0.0), averaged(nIn);
    std::vector<int32_t> nsnp, miss, usedV, missingV, carryNsnp(nOut * nIn, 0), carryMiss(nOut * nIn, 0), carryUsed(nOut, 0), carryMissing(nOut, 0);

    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        dev.load(batch, EVOFST_MODE_GLOBAL, nSamples, std::vector<int32_t>(), totalVariantNumber - (long long)n + 1, allowIndels);
        rows.clear();
        for (size_t s = 0; s < n; ++s) {                             // the state machine of :221-269 on coordinates alone
            if (batch.flags[s] & EVOFST_SITE_SKIP) continue;         // :194-202
            double coordDouble; { std::stringstream ss(batch.pos[s]); ss >> coordDouble; }
            if (coordDouble > currentWindowEnd || coordDouble < currentWindowStart) {
                int accessible = fixedWindowSize;                    // :223-230: asked of the scaffold of the line that closes the window
                if (haveBed) accessible = coordDouble > currentWindowEnd ? bed.basesIn(batch.chrom[s], currentWindowStart, currentWindowStart + fixedWindowSize)
                                                                         : bed.basesIn(batch.chrom[s], 0, fixedWindowSize);
                rows.push_back(Emission{(int64_t)s, batch.chrom[s], currentWindowStart, currentWindowEnd, accessible});
                if (coordDouble > currentWindowEnd) { currentWindowStart = currentWindowStart + fixedWindowSize; currentWindowEnd = currentWindowEnd + fixedWindowSize; }
                else if (coordDouble < currentWindowStart) { currentWindowStart = 0; currentWindowEnd = 0 + fixedWindowSize; }
            }
        }
        const size_t W = rows.size() + 1;                            // the last one stays open
        winOff.assign(1, 0);
        for (const Emission &e : rows) winOff.push_back(e.site);
        winOff.push_back((int64_t)n);
        sum.assign(W * nOut * nIn, 0.0); nsnp.assign(W * nOut * nIn, 0); miss.assign(W * nOut * nIn, 0);
        usedV.assign(W * nOut, 0); missingV.assign(W * nOut, 0);
        std::copy(carrySum.begin(), carrySum.end(), sum.begin()); std::copy(carryNsnp.begin(), carryNsnp.end(), nsnp.begin());
        std::copy(carryMiss.begin(), carryMiss.end(), miss.begin());
        std::copy(carryUsed.begin(), carryUsed.end(), usedV.begin()); std::copy(carryMissing.begin(), carryMissing.end(), missingV.begin());
        check(dev.ctx, evofst_get_dist_windows(dev.ctx, (int)W, winOff.data(), (int)nOut, outIdx.data(), (int)nIn, inIdx.data(), 1,
                                               sum.data(), nsnp.data(), miss.data(), usedV.data(), missingV.data()), "get_dist_windows");
        for (size_t w = 0; w + 1 < W; ++w) {
            const int accessibleInThisWindow = rows[w].accessible;
            for (size_t i = 0; i != nOut; i++) {                     // :232-254
                for (size_t j = 0; j != nIn; j++) {
                    const size_t o = (w * nOut + i) * nIn + j;
                    const int nSNPs = nsnp[o], md = miss[o];
                    if (nSNPs > 0) {
                        double missingProportion = (double)md / (nSNPs + md);
                        if (missingProportion > 0.5) averaged[j] = NAN;
                        else {
                            averaged[j] = (double)sum[o] / (double)accessibleInThisWindow;     // vector_average_withRegion
                            double nonMissingProportion = (double)nSNPs / (nSNPs + md);
                            double scalingFactor = 1 + (missingProportion / nonMissingProportion);
                            averaged[j] = averaged[j] * scalingFactor;
                        }
                    } else averaged[j] = NAN;
                }
                outFilesFixedWindow[i] << rows[w].chr << "\t" << rows[w].wStart << "\t" << rows[w].wEnd << "\t" << usedV[w * nOut + i] << "\t"
                                       << missingV[w * nOut + i] << "\t" << accessibleInThisWindow << "\t";
                printRow(outFilesFixedWindow[i], averaged);
            }
        }
        const size_t last = W - 1;
        std::copy(sum.begin() + last * nOut * nIn, sum.end(), carrySum.begin());
        std::copy(nsnp.begin() + last * nOut * nIn, nsnp.end(), carryNsnp.begin());
        std::copy(miss.begin() + last * nOut * nIn, miss.end(), carryMiss.begin());
        std::copy(usedV.begin() + last * nOut, usedV.end(), carryUsed.begin());
        std::copy(missingV.begin() + last * nOut, missingV.end(), carryMissing.begin());
        batch.clear();
    };

    while (vcf.getline(line)) {
        line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());   // :156
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            std::vector<string> sampleNames(fields.begin() + std::min<size_t>(9, fields.size()), fields.end());
            nSamples = (int)sampleNames.size();
            if (nSamples <= 0) die("VCF header has no sample columns");
            std::vector<int32_t> sp(nSamples, -1);
            for (int i = 0; i < nSamples; ++i) {                     // :163-165
                auto it = IDsToPopMap.find(sampleNames[i]);
                if (it != IDsToPopMap.end()) sp[i] = popIndex[it->second];
            }
            for (auto &kv : popToIDsMap)                              // :168-178
                if (locateSet(sampleNames, kv.second).empty()) { std::cerr << "Did not find any samples in the VCF for \"" << kv.first << "\"" << std::endl; exit(EXIT_FAILURE); }
            for (size_t k = 0; k < P; ++k)
                if (std::find(sp.begin(), sp.end(), (int32_t)k) == sp.end()) die("population \"" + populations[k] + "\" has no sample column of its own");
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, nSamples, (int)P, sp.data()), "set_layout");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        if (totalVariantNumber % 10000 == 0) std::cerr << "Processed " << totalVariantNumber << " variants" << std::endl;
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
    }
    flush();                                                         // the window open at the end is never written (:298)
    dev.close();
    return 0;
}

// ========================================================================================== stats

namespace sopt {
string vcfFile, accessibleGenBedFile, populationsFile, sampleNameFile;
int bootstrapBlockSize = 1000, numAccessibleBP = -1, accessibleGenBedWindow = 10000;
bool bDiffs = false, bDoBootstrap = false, countHets = false, bDoubleton = false, countPrivateVars = false;
const int n_bootstrap_replicates = 100;
}

const char *STATS_USAGE =
    "Usage: evo stats [OPTIONS] INPUT_FILE.vcf\n"
    "Pairwise difference statistics between the individuals of a VCF (GPU build: --diff-matrix, --hets-per-individual)\n"
    "       -h, --help                              display this help and exit\n"
    "       --diff-matrix                           writes FILE.diff_matrix.txt, FILE.diff_me_matrix.txt,\n"
    "                                               FILE.hets_over_homs_matrix.txt and FILE.parwiseMissingness.txt\n"
    "       --block-bootstrap=BLOCKSIZE             also 100 bootstrap replicates of the diff_me matrix, resampling blocks of\n"
    "                                               BLOCKSIZE variants with replacement (written next to the VCF)\n"
    "       --numAccessibleBP=NUM                   divide the diff_me matrix by the number of accessible base pairs\n"
    "       --hets-per-individual                   writes FILE.hets.txt and FILE.sharedHets.txt (next to the VCF): heterozygous calls\n"
    "                                               per individual at the sites without a missing genotype\n"
    "       --ind-file=NAMES                        sample names for the output headers, one per line, instead of the VCF's\n"
    "       --pop-file=POPULATIONS                  sample<TAB>population lines; with --private-variants writes the reference's\n"
    "                                               summary (FILE_POPULATIONS.privateFixedVars.txt; all 0: the counting is commented\n"
    "                                               out there); --doubleton-distribution is accepted and writes nothing, as there\n"
    "       --accessibleGenomeBED=BEDfile.bed       instead of a VCF the input is a file of scaffold names and lengths; prints the\n"
    "                                               accessible bases of every window to stdout and does nothing else\n"
    "       --accessibleGenBedWindow=NUMbp          the size of those windows (default 10000)\n\n";

enum { OPT_DIFF_MATRIX = 64, OPT_BLOCK_BOOTSTRAP, OPT_NUM_ACCESSIBLE, OPT_ACC_GEN_BED, OPT_ACC_GEN_BED_WINDOW, OPT_HETS, OPT_INDIV, OPT_POP,
       OPT_DOUBLETON, OPT_PRIVATE_VARS };

string stripPath(const string &filename) {                           // process_vcf_utils.cpp:153-160
    size_t slashPos = filename.find_last_of('/');
    return slashPos == string::npos ? filename : filename.substr(slashPos + 1);
}

// evo stats --diff-matrix (statsMain, process_vcf_stats.cpp:98-355: the bDiffs and bDoBootstrap parts).  The device
// returns, per bootstrap block and pair of individuals, the five counts the reference's per-site if-chains
// (process_vcf_stats_functions.cpp:180-281) reduce to; every matrix the reference prints follows from them.  All
// weights of the diff_me matrix are multiples of 1/4, so it is exact; Richard's matrix adds up 2/3 site by site in the
// reference and is a product here (equal to ~1e-13 relative, printed with 6 digits).
int statsMain(int argc, char **argv) {
    using namespace sopt;
    static const struct option longopts[] = {
        {"help", no_argument, NULL, 'h'}, {"diff-matrix", no_argument, NULL, OPT_DIFF_MATRIX},
        {"block-bootstrap", required_argument, NULL, OPT_BLOCK_BOOTSTRAP}, {"numAccessibleBP", required_argument, NULL, OPT_NUM_ACCESSIBLE},
        {"ind-file", required_argument, NULL, OPT_INDIV}, {"pop-file", required_argument, NULL, OPT_POP},
        {"hets-per-individual", no_argument, NULL, OPT_HETS}, {"doubleton-distribution", no_argument, NULL, OPT_DOUBLETON},
        {"diff-matrix-h1", no_argument, NULL, OPT_UNSUPPORTED}, {"diff-matrix-allH", no_argument, NULL, OPT_UNSUPPORTED},
        {"private-variants", no_argument, NULL, OPT_PRIVATE_VARS}, {"accessibleGenomeBED", required_argument, NULL, OPT_ACC_GEN_BED},
        {"accessibleGenBedWindow", required_argument, NULL, OPT_ACC_GEN_BED_WINDOW}, {NULL, 0, NULL, 0}};
    bool dieFlag = false;
    for (int c; (c = getopt_long(argc, argv, "h", longopts, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
            case '?': dieFlag = true; break;
            case OPT_DIFF_MATRIX: bDiffs = true; break;
            case OPT_BLOCK_BOOTSTRAP: arg >> bootstrapBlockSize; bDoBootstrap = true; break;
            case OPT_NUM_ACCESSIBLE: arg >> numAccessibleBP; break;
            case OPT_HETS: countHets = true; break;
            case OPT_POP: arg >> populationsFile; break;
            case OPT_INDIV: arg >> sampleNameFile; break;
            case OPT_DOUBLETON: bDoubleton = true; break;
            case OPT_PRIVATE_VARS: countPrivateVars = true; break;
            case OPT_ACC_GEN_BED: arg >> accessibleGenBedFile; break;
            case OPT_ACC_GEN_BED_WINDOW: arg >> accessibleGenBedWindow; break;
            case OPT_UNSUPPORTED: die("`evo stats --diff-matrix-h1` and `--diff-matrix-allH` are not carried by the GPU build");
            case 'h': std::cout << STATS_USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 1) { std::cerr << "missing arguments\n"; dieFlag = true; }
    else if (argc - optind > 1) { std::cerr << "too many arguments\n"; dieFlag = true; }
    if (bDoBootstrap && !bDiffs) { std::cerr << "--block-bootstrap needs --diff-matrix\n"; dieFlag = true; }
    if (countPrivateVars && populationsFile.empty()) {               // :393-396
        std::cerr << "You need to supply a file defining populations to count private fixed variants\n"; dieFlag = true;
    }
    if (!bDiffs && !bDoubleton && !countHets && !countPrivateVars && accessibleGenBedFile.empty()) { std::cerr << "Which statistics should the program calculate?\n"; dieFlag = true; }   // :398
    if (accessibleGenBedWindow <= 0) { std::cerr << "--accessibleGenBedWindow must be positive\n"; dieFlag = true; }
    if (bDoBootstrap && bootstrapBlockSize <= 0) { std::cerr << "the bootstrap BLOCKSIZE must be positive\n"; dieFlag = true; }
    if (dieFlag) { std::cout << "\n" << STATS_USAGE; exit(EXIT_FAILURE); }
    vcfFile = argv[optind++];

    // --accessibleGenomeBED (:146-164): the input is a file of scaffold names and lengths, not a VCF; one row per window
    // of --accessibleGenBedWindow bases with the accessible bases in [i, i + window] goes to stdout, and nothing else is
    // done (host only: this is where --numAccessibleBP of the difference matrices comes from)
    if (!accessibleGenBedFile.empty()) {
        std::cerr << "Loading the accessible genome annotation" << std::endl;
        AccessibleBed ag;
        ag.load(accessibleGenBedFile);
        std::cerr << "Done" << std::endl;
        LineReader lengths(vcfFile);
        string l;
        while (lengths.getline(l)) {
            std::vector<string> fields = split(l, '\t');
            if (fields.size() < 2) die("the scaffold lengths file needs two tab-separated columns per line");
            const string &sc = fields[0];
            const int len = atoi(fields[1].c_str());
            for (int i = 0; i < len; i = i + accessibleGenBedWindow)
                std::cout << sc << "\t" << i << "\t" << i + accessibleGenBedWindow << "\t" << ag.basesIn(sc, i, i + accessibleGenBedWindow) << std::endl;
        }
        return 0;
    }

    const string fileRoot = stripExtension(vcfFile);                 // :103
    std::cerr << "Calculating statistics from: " << vcfFile << std::endl;
    // --pop-file (:116-135): the populations are read and, at the header, looked for among the samples; the two
    // analyses that would use them (--doubleton-distribution, --private-variants) are commented out in the reference
    // (:243-249, :332-333), so the first writes nothing and the second a summary of zeros (:340-341)
    std::map<string, std::vector<string>> popToIDsMap;
    std::vector<string> populationsToUse;
    if (!populationsFile.empty()) {
        std::ifstream setsF(populationsFile.c_str());
        if (!setsF.is_open()) die("could not open " + populationsFile);
        string l;
        while (getline(setsF, l)) {
            std::vector<string> ID_Pop = split(l, '\t');
            if (ID_Pop.size() < 2) die("the --pop-file needs two tab-separated columns per line");
            popToIDsMap[ID_Pop[1]].push_back(ID_Pop[0]);
        }
        for (auto &kv : popToIDsMap) if (kv.first != "Outgroup" && kv.first != "xxx") populationsToUse.push_back(kv.first);
        std::cerr << "There are " << popToIDsMap.size() << " populations " << std::endl;
    }
    if (countHets) {                                                 // :167-170
        std::cerr << "Het count per individual are going to be output to: " << fileRoot + ".hets.txt" << std::endl;
        std::cerr << "Counts of hets (per individual) shared with others are going to be output to: " << fileRoot + ".sharedHets.txt" << std::endl;
    }
    if (bDoubleton) std::cerr << "The distribution of doubletons is going to be output to: " << fileRoot + ".doubletons.txt" << std::endl;   // :171-174
    if (countPrivateVars)                                            // :175-179
        std::cerr << "The numbers of private variants fixed in groups defined in " << populationsFile << " are going to be output to: "
                  << fileRoot + "_" + stripExtension(populationsFile) + ".privateFixedVars.txt" << std::endl;
    std::vector<int> hetCounts, hetsSharedWithOthers;                // :138, :224
    std::vector<int32_t> bhet, bshared;
    LineReader vcf(vcfFile);
    std::vector<string> sampleNames;
    int numSamples = 0, totalVariantNumber = 0;
    long long phaseDraws = 0;                                        // rand() draws the reference has made by the end of the file
    size_t T = 0;                                                    // pairs j <= i
    std::vector<int64_t> hh, hd, oh, ms, d4;                         // totals per pair
    std::vector<int32_t> openMe4, openMs;                            // the bootstrap block being filled (:220-221)
    std::vector<std::vector<int32_t>> blockMe4, blockMs;             // finished blocks (:266-267), diff_me in quarters
    Device dev;
    Batch batch;
    bool haveHeader = false;
    std::vector<int32_t> bhh, bhd, boh, bms, bd4;
    std::vector<int64_t> multiSite, blockOff;
    std::vector<uint8_t> multiAlleles;

    // diff_me of one pair in quarters: off the diagonal hh / 2 + hd + oh / 2 + d4 / 4, on it hh + d4 (:218-226, :269-274)
    auto me4 = [](bool diag, int64_t a_hh, int64_t a_hd, int64_t a_oh, int64_t a_d4) {
        return diag ? 4 * (a_hh + a_d4) : 2 * a_hh + 4 * a_hd + 2 * a_oh + a_d4;
    };
    std::vector<uint8_t> isDiag;

    auto flush = [&]() {
        if (batch.n == 0) return;
        const size_t n = batch.n;
        const long long g0 = totalVariantNumber - (long long)n;      // variants before this batch
        uint8_t *stTiles = dev.stage(std::max(evofst_tiles_bytes(dev.ctx, (int64_t)n), evofst_tiles_bytes(dev.ctx, (int64_t)kChunkSites)));
        batch.flags.assign(n, 0);
        std::vector<const char *> ptr(n);
        for (size_t i = 0; i < n; ++i) ptr[i] = batch.lines[i].c_str();
        const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (evofst_pack_vcf_lines(dev.ctx, EVOFST_MODE_STATS, (int64_t)n, ptr.data(), numSamples, nullptr, threads, stTiles,
                                  batch.flags.data()) != EVOFST_OK)
            die("malformed VCF line among lines " + numToString(g0 + 1) + ".." + numToString(g0 + (long long)n) +
                " (diploid GT fields with single-digit alleles are required)");
        check(dev.ctx, evofst_upload(dev.ctx, (int64_t)n, stTiles, batch.flags.data(), nullptr), "upload");
        if (bDoBootstrap) {
            int64_t draws = 0;
            if (evofst_stats_phase_draws((int64_t)n, ptr.data(), numSamples, threads, &draws) != EVOFST_OK) die("malformed VCF line");
            phaseDraws += draws;
        }
        if (countHets) {                                             // :239-242 het_analysis at the sites without a missing genotype
            bhet.resize(numSamples); bshared.resize(numSamples);
            check(dev.ctx, evofst_get_indiv_hets(dev.ctx, bhet.data(), bshared.data()), "get_indiv_hets");
            for (int i = 0; i < numSamples; ++i) { hetCounts[i] += bhet[i]; hetsSharedWithOthers[i] += bshared[i]; }
        }
        if (!bDiffs) { batch.clear(); return; }
        multiSite.clear(); multiAlleles.clear();
        for (size_t s = 0; s < n; ++s)
            if (batch.flags[s] & EVOFST_SITE_MULTI) {
                multiSite.push_back((int64_t)s);
                multiAlleles.resize(multiSite.size() * 2 * (size_t)numSamples);
                if (evofst_alleles_from_vcf_line(ptr[s], numSamples, &multiAlleles[(multiSite.size() - 1) * 2 * (size_t)numSamples]) != EVOFST_OK)
                    die("malformed VCF line " + numToString(g0 + (long long)s + 1));
            }
        check(dev.ctx, evofst_set_multiallelic(dev.ctx, (int64_t)multiSite.size(), multiSite.data(), multiAlleles.data()), "set_multiallelic");
        // blocks end after every variant whose number is a multiple of the block size (:265)
        blockOff.assign(1, 0);
        if (bDoBootstrap)
            for (long long k = g0 / bootstrapBlockSize + 1; k * bootstrapBlockSize < g0 + (long long)n; ++k)
                blockOff.push_back(k * bootstrapBlockSize - g0);
        else                                                         // only totals wanted: cut the batch so that the chip fills
            for (int64_t k = 4096; k < (int64_t)n; k += 4096) blockOff.push_back(k);
        blockOff.push_back((int64_t)n);
        const bool lastCloses = bDoBootstrap && (g0 + (long long)n) % bootstrapBlockSize == 0;
        const size_t nBlocks = blockOff.size() - 1;
        const size_t perCall = std::max<size_t>(1, std::min<size_t>(65535, ((size_t)64 << 20) / (T * 20)));
        for (size_t b0 = 0; b0 < nBlocks; b0 += perCall) {
            const size_t nb = std::min(perCall, nBlocks - b0);
            bhh.resize(nb * T); bhd.resize(nb * T); boh.resize(nb * T); bms.resize(nb * T); bd4.resize(nb * T);
            check(dev.ctx, evofst_get_indiv_diffs(dev.ctx, (int)nb, &blockOff[b0], bhh.data(), bhd.data(), boh.data(), bms.data(), bd4.data()),
                  "get_indiv_diffs");
            for (size_t b = 0; b < nb; ++b) {
                const int32_t *a = &bhh[b * T], *d = &bhd[b * T], *o = &boh[b * T], *m = &bms[b * T], *q = &bd4[b * T];
                for (size_t k = 0; k < T; ++k) { hh[k] += a[k]; hd[k] += d[k]; oh[k] += o[k]; ms[k] += m[k]; d4[k] += q[k]; }
                if (!bDoBootstrap) continue;
                for (size_t k = 0; k < T; ++k) { openMe4[k] += (int32_t)me4(isDiag[k], a[k], d[k], o[k], q[k]); openMs[k] += m[k]; }
                const bool closes = (b0 + b + 1 < nBlocks) || lastCloses;
                if (closes) {
                    blockMe4.push_back(openMe4); blockMs.push_back(openMs);
                    std::fill(openMe4.begin(), openMe4.end(), 0); std::fill(openMs.begin(), openMs.end(), 0);
                }
            }
        }
        batch.clear();
    };

    string line;
    while (vcf.getline(line)) {                                      // :188 (no \r handling in this sub-program)
        if (line[0] == '#' && line[1] == '#') continue;
        if (line[0] == '#' && line[1] == 'C') {
            std::vector<string> fields = split(line, '\t');
            if (sampleNameFile.empty()) {
                for (size_t i = 9; i < fields.size(); i++) sampleNames.push_back(fields[i]);   // :191-194
            } else {                                                 // :195-197 readSampleNamesFromTextFile: a name per line
                std::ifstream namesF(sampleNameFile.c_str());
                if (!namesF.is_open()) die("could not open " + sampleNameFile);
                string l;
                while (getline(namesF, l)) sampleNames.push_back(l);
                // the reference sizes its matrices by this list and indexes the genotype columns with it; a list of
                // another length reads past them there
                if (sampleNames.size() + 9 != fields.size()) die("--ind-file must name every sample column of the VCF (" + numToString((int)fields.size() - 9) + ")");
            }
            for (auto &kv : popToIDsMap)                              // :206-216
                if (locateSet(sampleNames, kv.second).empty()) { std::cerr << "Did not find any samples in the VCF for \"" << kv.first << "\"" << std::endl; exit(EXIT_FAILURE); }
            numSamples = (int)sampleNames.size();
            if (numSamples <= 0) die("VCF header has no sample columns");
            std::cerr << "Number of chromosomes: " << numSamples * 2 << std::endl;
            T = (size_t)numSamples * ((size_t)numSamples + 1) / 2;
            hh.assign(T, 0); hd.assign(T, 0); oh.assign(T, 0); ms.assign(T, 0); d4.assign(T, 0);
            isDiag.assign(T, 0);
            for (size_t i = 0; i < (size_t)numSamples; ++i) isDiag[i * (i + 1) / 2 + i] = 1;
            if (bDoBootstrap) { openMe4.assign(T, 0); openMs.assign(T, 0); }
            hetCounts.assign(numSamples, 0); hetsSharedWithOthers.assign(numSamples, 0);
            std::vector<int32_t> sp(numSamples, 0);                  // one population: the layout keeps the VCF order
            dev.open();
            check(dev.ctx, evofst_set_layout(dev.ctx, numSamples, 1, sp.data()), "set_layout");
            check(dev.ctx, evofst_reserve(dev.ctx, kChunkSites), "reserve");
            haveHeader = true;
            continue;
        }
        if (!haveHeader) die("variant line before the #CHROM header");
        totalVariantNumber++;
        if (totalVariantNumber % 100000 == 0) std::cerr << "Processed " << totalVariantNumber << " variants" << std::endl;
        batch.add(line, totalVariantNumber);
        if (batch.n == (size_t)kChunkSites) flush();
    }
    flush();
    if (!haveHeader) die("no #CHROM header in " + vcfFile);
    const size_t N = (size_t)numSamples;
    auto tri = [](size_t i, size_t j) { return i * (i + 1) / 2 + j; };

    if (bDoBootstrap) {                                              // :296-325
        const int nBlocks = (int)blockMe4.size();
        // the reference never seeds rand(), and has drawn from it once per unphased heterozygous genotype while
        // reading (randomisePhase, process_vcf_utils.cpp:268-276): start the block picks at the same place of the stream
        srand(1);
        for (long long k = 0; k < phaseDraws; ++k) (void)rand();
        std::vector<double> rep(N * N), repMiss(N * N);
        for (int r = 0; r < n_bootstrap_replicates; r++) {
            std::ofstream bootOut((fileRoot + "boot." + numToString(r) + ".txt").c_str());
            std::fill(rep.begin(), rep.end(), 0.0); std::fill(repMiss.begin(), repMiss.end(), 0.0);
            for (int b = 0; b < nBlocks; b++) {
                const int block = rand() % nBlocks;
                for (size_t i = 0; i < N; i++)
                    for (size_t j = 0; j <= i; j++) {
                        rep[i * N + j] = rep[i * N + j] + blockMe4[block][tri(i, j)] / 4.0;
                        repMiss[i * N + j] = repMiss[i * N + j] + blockMs[block][tri(i, j)];
                    }
            }
            const int totalSitesBootstrap = nBlocks * bootstrapBlockSize;
            for (size_t k = 0; k < N * N; k++) {
                const double proportionUsed = 1 - (repMiss[k] / totalSitesBootstrap);
                rep[k] = rep[k] / proportionUsed;
            }
            for (size_t i = 0; i < N; i++) bootOut << sampleNames[i] << (i == N - 1 ? "" : "\t");
            bootOut << std::endl;
            for (size_t i = 0; i < N; i++) {
                for (size_t j = 0; j < N; j++) { bootOut << rep[i * N + j]; if (j != N - 1) bootOut << "\t"; }
                bootOut << std::endl;
            }
        }
    }

    if (countHets) {                                                 // :336-337 print_het_counts (process_vcf_print_routines.cpp:29-56)
        std::ofstream hetsOut((fileRoot + ".hets.txt").c_str()), sharedOut((fileRoot + ".sharedHets.txt").c_str());
        hetsOut << "# Het counts" << std::endl;
        hetsOut << "# Input file:" << fileRoot << ".vcf" << std::endl;
        sharedOut << "# Shared het counts (line1) and proportions (line 2)" << std::endl;
        sharedOut << "# Input file:" << fileRoot << ".vcf" << std::endl;
        std::vector<double> sharedHetProportions;
        for (size_t i = 0; i < N; i++) sharedHetProportions.push_back((double)hetsSharedWithOthers[i] / hetCounts[i]);   // 0 / 0: "-nan"
        auto row = [&](std::ofstream &f, const auto &v) { for (size_t i = 0; i < N; i++) { f << v[i]; if (i == N - 1) f << std::endl; else f << "\t"; } };
        row(hetsOut, sampleNames); row(hetsOut, hetCounts);
        row(sharedOut, sampleNames); row(sharedOut, hetsSharedWithOthers); row(sharedOut, sharedHetProportions);
    }
    if (countPrivateVars) {                                          // :340-341 print_privateFixedVarsSummary (print_routines.cpp:59-69)
        std::ofstream pv((fileRoot + "_" + stripExtension(populationsFile) + ".privateFixedVars.txt").c_str());
        pv << "# Counts of private fixed variants:" << std::endl;
        pv << "# Input file:" << fileRoot << ".vcf" << std::endl;
        pv << "# Groups defined in:" << populationsFile << std::endl;
        const size_t np = populationsToUse.size();                   // the counts are never touched after :227: all 0
        for (size_t i = 0; i < np; i++) { pv << populationsToUse[i]; if (i == np - 1) pv << std::endl; else pv << "\t"; }
        for (size_t i = 0; i < np; i++) { pv << 0; if (i == np - 1) pv << std::endl; else pv << "\t"; }
    }
    if (!bDiffs) { dev.close(); return 0; }

    // print_pairwise_diff_stats (process_vcf_print_routines.cpp:74-120) after finalize_diffs_Hets_vs_Homs_proportions
    const string fileNoPath = stripPath(fileRoot);                   // :328
    std::ofstream diffOut((fileNoPath + ".diff_matrix.txt").c_str()), diffMeOut((fileNoPath + ".diff_me_matrix.txt").c_str()),
        hetHomOut((fileNoPath + ".hets_over_homs_matrix.txt").c_str()), pairMisOut((fileNoPath + ".parwiseMissingness.txt").c_str());
    diffOut << "# Input file:" << fileNoPath << ".vcf" << std::endl;
    diffOut << "# Total number of segragating variant sites in this sample:" << totalVariantNumber << std::endl;
    diffOut << "# Richard's scoring scheme" << std::endl;
    diffMeOut << "# Input file:" << fileNoPath << ".vcf" << std::endl;
    diffMeOut << "# Total number of segragating variant sites in this sample: " << totalVariantNumber << std::endl;
    diffMeOut << "# Homozygous difference = 1, one homozygous, another heterozygous = 0.5" << std::endl;
    hetHomOut << "# Input file:" << fileNoPath << ".vcf" << std::endl;
    hetHomOut << "# number of sites both individuals hets/number of sites individuals have a homozygous difference; i.e. num(1/0::1/0)/num(1/1::0/0)" << std::endl;
    hetHomOut << "# For a free mixing population, we expect this number ~2; for fully separated species ~0" << std::endl;
    pairMisOut << "# Input file:" << fileNoPath << ".vcf" << std::endl;
    pairMisOut << "# Total number of segragating variant sites in this sample: " << totalVariantNumber << std::endl;
    pairMisOut << "# Pairwise missingness:" << std::endl;
    for (std::ofstream *f : {&diffOut, &diffMeOut, &hetHomOut, &pairMisOut}) {
        for (size_t i = 0; i < N; i++) *f << sampleNames[i] << (i == N - 1 ? "" : "\t");
        *f << std::endl;
    }
    diffMeOut.precision(10);
    for (size_t i = 0; i < N; i++)
        for (size_t j = 0; j < N; j++) {
            double richard = 0, me = 0, hetHom = 0;
            long long mis = 0;
            if (j <= i) {
                const size_t k = tri(i, j);
                if (j < i) {
                    richard = (2.0 / 3.0) * (double)hh[k] + (2.0 / 3.0) * (double)hd[k] + 0.5 * (double)oh[k];
                    hetHom = (double)hh[k] / (double)hd[k];          // utils.cpp:1151
                } else richard = (double)hh[k];
                me = (double)me4(j == i, hh[k], hd[k], oh[k], d4[k]) / 4.0;
                if (numAccessibleBP > -1) me = (double)me / numAccessibleBP;                 // :286-293
                mis = ms[k];
            } else if (numAccessibleBP > -1) me = 0.0 / numAccessibleBP;
            const char *sep = j == N - 1 ? "\n" : "\t";
            diffOut << richard; diffMeOut << me; hetHomOut << hetHom; pairMisOut << mis;
            if (j == N - 1) { diffOut << std::endl; diffMeOut << std::endl; hetHomOut << std::endl; pairMisOut << std::endl; }
            else { diffOut << sep; diffMeOut << sep; hetHomOut << sep; pairMisOut << sep; }
        }
    dev.close();
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc <= 1) {
        std::cout << "evo (MI355X build): commands  fst  FstGlobal  alleleFreq  PBS  ABS  DistOutgroups  stats\n";
        return 0;
    }
    string command(argv[1]);
    if (command == "fst") return fstMain(argc - 1, argv + 1);
    if (command == "FstGlobal") return FstGlobalMain(argc - 1, argv + 1);
    if (command == "alleleFreq") return AFmain(argc - 1, argv + 1);
    if (command == "PBS") return branchMain(3, argc - 1, argv + 1);
    if (command == "ABS") return branchMain(4, argc - 1, argv + 1);
    if (command == "DistOutgroups") return DistOutMain(argc - 1, argv + 1);
    if (command == "stats") return statsMain(argc - 1, argv + 1);
    if (command == "help" || command == "--help") {
        std::cout << "evo (MI355X build): commands  fst  FstGlobal  alleleFreq  PBS  ABS  DistOutgroups  stats\n" << FST_USAGE << FSTGLOBAL_USAGE << AF_USAGE << PBS_USAGE << ABS_USAGE << DISTOUT_USAGE << STATS_USAGE;
        return 0;
    }
    std::cerr << "Unrecognized command: " << command << " (this build carries only the Fst / dxy path: fst, FstGlobal, alleleFreq, PBS, ABS, DistOutgroups, stats --diff-matrix)\n";
    return 1;
}
