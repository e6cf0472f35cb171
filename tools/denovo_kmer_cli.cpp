// denovo_kmer_cli.cpp -- command-line driver over include/denovo_kmer.hpp.
//
// Stands in for the reference's Rust CLI on the hot path only: reads come from FASTA / FASTQ /
// one-sequence-per-line text (BAM/VCF I/O stays with the host tool, BASELINE.json north_star), the
// parents go into the GPU-resident filter batch by batch, the child is probed batch by batch, the
// per-batch tables are merged on the device and the child-only k-mers are written as TSV.
//
//   denovo_kmer_cli --k 31 --filter-log2 34 --parent p1.fq --parent p2.fq --child c.fq --out denovo.tsv
//                   [--hashes 4] [--seed N] [--min-count 2] [--batch-reads 2000000] [--mode auto|direct|bucketed]
//                   [--save-filter parents.dkbloom] [--load-filter parents.dkbloom] [--forward-only]
//                   [--exact]   parents held as an exact set of 2^filter-log2 bits instead of a Bloom filter
//
// Build: g++ -std=c++17 -O2 tools/denovo_kmer_cli.cpp -Ldenovo_kmer_amd -ldenovo_kmer -L/opt/rocm/lib -lamdhip64
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "../include/denovo_kmer.hpp"

namespace {

// sequential reader of FASTA ('>'), FASTQ ('@') or plain lines; returns false at end of file
class SeqReader {
public:
    explicit SeqReader(const std::string &path) : in_(path)
    {
        if (!in_) throw std::runtime_error("cannot open " + path);
        const int c = in_.peek();
        kind_ = c == '>' ? 'a' : c == '@' ? 'q' : 'p';
    }
    bool next(std::string &seq)
    {
        std::string line;
        if (kind_ == 'p') {
            while (std::getline(in_, line)) {
                strip(line);
                if (!line.empty()) { seq = line; return true; }
            }
            return false;
        }
        if (kind_ == 'q') {
            std::string plus, qual;
            if (!std::getline(in_, line)) return false;            // @name
            if (!std::getline(in_, seq)) return false;
            std::getline(in_, plus);
            std::getline(in_, qual);
            strip(seq);
            return true;
        }
        // FASTA: a record may span lines
        if (pending_.empty() && !std::getline(in_, pending_)) return false;
        seq.clear();
        while (std::getline(in_, line)) {
            if (!line.empty() && line[0] == '>') { pending_ = line; return true; }
            strip(line);
            seq += line;
        }
        pending_.clear();
        return !seq.empty() || in_.eof();
    }

private:
    static void strip(std::string &s)
    {
        while (!s.empty() && (s.back() == '\r' || s.back() == '\n' || s.back() == ' ')) s.pop_back();
    }
    std::ifstream in_;
    char kind_;
    std::string pending_;
};

struct Args {
    uint32_t k = 31, filter_log2 = 30, hashes = 4, min_count = 1, mode = DK_MODE_AUTO, windows = 1;
    uint64_t seed = 0x5EED, batch_reads = 2000000, accum_capacity = 0;        // capacity 0 = from the child file's size
    bool canonical = true, exact = false;
    std::vector<std::string> parents;
    std::string child, out, save_filter, load_filter;
};

[[noreturn]] void usage(const char *msg)
{
    std::fprintf(stderr, "%s\nusage: denovo_kmer_cli --k K --filter-log2 N --parent FILE [--parent FILE ...] --child FILE --out FILE\n"
                         "       [--hashes 4] [--seed N] [--min-count 1] [--batch-reads 2000000] [--mode auto|direct|bucketed]\n"
                         "       [--save-filter FILE] [--load-filter FILE] [--forward-only] [--exact]\n"
                         "       [--windows 1] [--accum-capacity N]   (child-only occurrences expected per hash window; default: 30 %% of\n"
                         "        the child file's bases / windows, at least 64 M; the child file is read once per window)\n"
                         "       k 1..64, filter-log2 20..40, windows a power of two <= 2^(filter-log2 - 20)\n", msg);
    std::exit(2);
}

Args parse(int argc, char **argv)
{
    Args a;
    for (int i = 1; i < argc; i++) {
        const std::string f = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) usage(("missing value for " + f).c_str());
            return argv[++i];
        };
        if (f == "--k") a.k = (uint32_t)std::stoul(val());
        else if (f == "--filter-log2") a.filter_log2 = (uint32_t)std::stoul(val());
        else if (f == "--hashes") a.hashes = (uint32_t)std::stoul(val());
        else if (f == "--seed") a.seed = std::stoull(val());
        else if (f == "--min-count") a.min_count = (uint32_t)std::stoul(val());
        else if (f == "--batch-reads") a.batch_reads = std::stoull(val());
        else if (f == "--windows") a.windows = (uint32_t)std::stoul(val());
        else if (f == "--accum-capacity") a.accum_capacity = std::stoull(val());
        else if (f == "--parent") a.parents.push_back(val());
        else if (f == "--child") a.child = val();
        else if (f == "--out") a.out = val();
        else if (f == "--save-filter") a.save_filter = val();
        else if (f == "--load-filter") a.load_filter = val();
        else if (f == "--forward-only") a.canonical = false;
        else if (f == "--exact") a.exact = true;
        else if (f == "--mode") {
            const std::string m = val();
            a.mode = m == "direct" ? DK_MODE_DIRECT : m == "bucketed" ? DK_MODE_BUCKETED : DK_MODE_AUTO;
        } else usage(("unknown flag " + f).c_str());
    }
    if (a.child.empty() || a.out.empty()) usage("--child and --out are required");
    if (a.parents.empty() && a.load_filter.empty()) usage("give --parent files or --load-filter");
    if (a.batch_reads == 0) usage("--batch-reads must be positive");
    if (a.windows == 0 || (a.windows & (a.windows - 1))) usage("--windows must be a power of two");
    if (a.k < 1 || a.k > 64) usage("--k must be 1..64");
    if (a.filter_log2 < 20 || a.filter_log2 > 40) usage("--filter-log2 must be 20..40");
    // every hash window covers at least two 64-KiB segments of the set (dk_accum_create)
    if ((uint64_t)a.windows > (1ULL << (a.filter_log2 - 20)))
        usage(("--windows " + std::to_string(a.windows) + ": a set of 2^" + std::to_string(a.filter_log2) + " bits takes at most " +
               std::to_string(1ULL << (a.filter_log2 - 20)) + " hash window(s)").c_str());
    return a;
}

std::string kmer_string(uint64_t hi, uint64_t lo, uint32_t k)
{
    std::string s(k, 'A');
    for (uint32_t i = 0; i < k; i++) {
        const uint32_t shift = 2 * (k - 1 - i);
        const uint64_t code = shift >= 64 ? (hi >> (shift - 64)) & 3 : (lo >> shift) & 3;
        s[i] = "ACGT"[code];
    }
    return s;
}

// feed a file to `fn` in batches of at most batch_reads sequences
template <class Fn>
uint64_t for_each_batch(const std::string &path, uint64_t batch_reads, Fn fn)
{
    SeqReader rd(path);
    std::vector<std::string> batch;
    std::string seq;
    uint64_t n = 0;
    while (rd.next(seq)) {
        batch.push_back(seq);
        n++;
        if (batch.size() == batch_reads) { fn(batch); batch.clear(); }
    }
    if (!batch.empty()) fn(batch);
    return n;
}

}  // namespace

int main(int argc, char **argv)
{
    const Args a = parse(argc, argv);
    try {
        dk_host::Config c;
        c.k = a.k;
        c.canonical = a.canonical;
        c.filter_log2_bits = a.filter_log2;
        c.n_hashes = a.hashes;
        c.seed = a.seed;
        c.min_count = 1;                 // thresholds apply to the merged counts
        c.mode = a.mode;
        c.set_kind = a.exact ? DK_SET_EXACT : DK_SET_BLOOM;
        dk_host::Engine eng(c);
        dk_host::KmerSet parents(eng);
        if (!a.load_filter.empty()) parents.load(a.load_filter);
        uint64_t parent_windows = 0;
        for (const std::string &p : a.parents) {
            const uint64_t n = for_each_batch(p, a.batch_reads, [&](const std::vector<std::string> &b) {
                parent_windows += parents.insert_sequences(b).n_windows;
            });
            std::fprintf(stderr, "parent %s: %llu reads\n", p.c_str(), (unsigned long long)n);
        }
        if (!a.save_filter.empty()) parents.save(a.save_filter);
        if (a.exact) std::fprintf(stderr, "exact parent set: %llu k-mers\n", (unsigned long long)parents.popcount());

        // the child's absent k-mer occurrences stay on the GPU across batches and are counted once per hash window
        // (what one hash window of this set geometry can count: 1024 units of 12288 (k > 32: 6144) records per 64-KiB segment)
        const uint64_t geometry_max = (1ULL << (a.filter_log2 - 19)) / a.windows * 1024 * (a.k > 32 ? 6144 : 12288);
        uint64_t capacity = a.accum_capacity;
        if (!capacity) {
            // not given: 30 % of the child's windows may be absent (reads with 0.5 % errors leave ~14 % at k = 31), spread
            // over the hash windows; the file's size bounds its bases (FASTQ spends as many bytes on qualities)
            std::ifstream probe(a.child, std::ios::binary | std::ios::ate);
            if (!probe) throw std::runtime_error("cannot open " + a.child);
            const uint64_t bytes = (uint64_t)probe.tellg();
            probe.seekg(0);
            const uint64_t bases = probe.peek() == '@' ? bytes / 2 : bytes;
            capacity = std::max<uint64_t>(64000000 / a.windows, bases * 3 / 10 / a.windows);
        }
        capacity = std::max<uint64_t>(1, std::min<uint64_t>(capacity, geometry_max));
        dk_host::ChildAccumulator acc(eng, &parents, capacity, a.windows);
        dk_host::KmerCounts res{};
        uint64_t n_child = 0, n_batches = 0;
        for (uint32_t w = 0; w < a.windows; w++) {
            acc.reset(w);
            n_batches = 0;
            n_child = for_each_batch(a.child, a.batch_reads, [&](const std::vector<std::string> &b) {
                dk_host::ReadBatch rb(eng, b);
                acc.add(rb);
                n_batches++;
            });
            acc.finish(a.min_count, res);
        }
        std::fprintf(stderr, "child %s: %llu reads in %llu batch(es), %u hash window(s); %zu child-only k-mers with count >= %u (%llu distinct)\n",
                     a.child.c_str(), (unsigned long long)n_child, (unsigned long long)n_batches, a.windows, res.size(), a.min_count,
                     (unsigned long long)res.stats.n_distinct);

        std::vector<size_t> order(res.size());
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            return res.hi[x] != res.hi[y] ? res.hi[x] < res.hi[y] : res.lo[x] < res.lo[y];
        });
        std::ofstream out(a.out);
        if (!out) throw std::runtime_error("cannot open " + a.out);
        out << "kmer\tcount\n";
        for (size_t i : order) out << kmer_string(res.hi[i], res.lo[i], a.k) << '\t' << res.count[i] << '\n';
    } catch (const dk_host::Error &e) {
        std::fprintf(stderr, "denovo_kmer error %d: %s\n", (int)e.status, e.what());
        if (e.status == DK_ERR_OVERFLOW)
            std::fprintf(stderr, "the child holds more absent k-mer occurrences per hash window than the accumulator was sized for: "
                                 "raise --accum-capacity (%s) or --windows (%u)\n",
                         a.accum_capacity ? std::to_string(a.accum_capacity).c_str() : "automatic", a.windows);
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
