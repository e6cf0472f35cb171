// denovo_kmer.hpp -- header-only C++ host mirror of the reference's k-mer API over the C ABI.
//
// The reference (jlanej/denovo_kmer, Rust) exposes `KmerCounter` and `KmerSet` from counter.rs and
// the extraction helpers from kmer.rs (both NOT IN MOUNT; there is no Rust toolchain
// in the build image, so the host side above the C ABI is C++ where the reference is compiled
// code).  The classes keep those names and argument meaning: construct with k, feed sequences,
// ask for membership / counts; errors surface as dk::Error (a Rust binding would return Result).
// Everything computes on the GPU through libdenovo_kmer.so; nothing here touches a k-mer.
#ifndef DENOVO_KMER_HPP
#define DENOVO_KMER_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "denovo_kmer.h"

namespace dk_host {

struct Error : std::runtime_error {
    dk_status status;
    Error(dk_status s, const std::string &msg) : std::runtime_error(msg), status(s) {}
};

inline void check(dk_status s, const dk_engine *e)
{
    if (s != DK_OK) throw Error(s, std::string(dk_status_string(s)) + ": " + dk_last_error(e));
}

struct Config {
    uint32_t k = 31;
    bool canonical = true;
    uint32_t filter_log2_bits = 30;
    uint32_t n_hashes = 4;
    uint64_t seed = 0x5EED;
    uint32_t min_count = 1;
    int32_t device_id = 0;
    uint32_t mode = DK_MODE_AUTO;
    uint32_t set_kind = DK_SET_BLOOM;      // DK_SET_EXACT: KmerSet is an exact set (HashSet semantics)
    void *stream = nullptr;
};

class Engine {
public:
    explicit Engine(const Config &c) : k_(c.k)
    {
        dk_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.k = c.k;
        cfg.canonical = c.canonical ? 1u : 0u;
        cfg.filter_log2_bits = c.filter_log2_bits;
        cfg.n_hashes = c.n_hashes;
        cfg.seed = c.seed;
        cfg.min_count = c.min_count;
        cfg.device_id = c.device_id;
        cfg.world_size = 1;
        cfg.mode = c.mode;
        cfg.set_kind = c.set_kind;
        cfg.stream = c.stream;
        check(dk_engine_create(&cfg, &e_), nullptr);
    }
    ~Engine() { dk_engine_destroy(e_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    dk_engine *get() const { return e_; }
    uint32_t k() const { return k_; }
    // one arena for everything the engine allocates later (the slow hipMalloc happens here, not inside the first batch)
    void reserve(uint64_t bytes) { check(dk_engine_reserve(e_, bytes), e_); }
    void set_option(const char *name, int64_t value) { check(dk_engine_set_option(e_, name, value), e_); }
    int64_t info(const char *name) const
    {
        int64_t v = 0;
        check(dk_engine_get_info(e_, name, &v), e_);
        return v;
    }

private:
    dk_engine *e_ = nullptr;
    uint32_t k_;
};

// a batch of reads resident on the GPU in the packed format
class ReadBatch {
public:
    ReadBatch(Engine &e, const std::vector<std::string> &reads) : e_(e)
    {
        std::string seq;
        std::vector<uint64_t> off(reads.size() + 1, 0);
        for (size_t i = 0; i < reads.size(); i++) {
            seq += reads[i];
            off[i + 1] = seq.size();
        }
        check(dk_reads_from_ascii(e.get(), reinterpret_cast<const uint8_t *>(seq.data()), off.data(), reads.size(), &r_),
              e.get());
    }
    ReadBatch(Engine &e, const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads) : e_(e)
    {
        check(dk_reads_from_ascii(e.get(), seq, offsets, n_reads, &r_), e.get());
    }
    ~ReadBatch() { dk_reads_destroy(r_); }
    // kmer.rs stand-in: canonical k-mer (and hash) of every stream position; not_kmer bit p (MSB first) = no k-mer at p
    struct Kmers {
        std::vector<uint64_t> lo, hi, hash, not_kmer;
        dk_stats stats{};
    };
    Kmers kmers(bool with_hashes = true) const
    {
        Kmers out;
        dk_stats st{};
        check(dk_reads_stats(r_, &st), e_.get());
        out.lo.resize(st.n_bases);
        if (e_.k() > 32) out.hi.resize(st.n_bases);
        if (with_hashes) out.hash.resize(st.n_bases);
        out.not_kmer.resize((st.n_bases + 63) / 64);
        check(dk_reads_kmers(e_.get(), r_, out.lo.data(), out.hi.empty() ? nullptr : out.hi.data(),
                             out.hash.empty() ? nullptr : out.hash.data(), out.not_kmer.data(), &out.stats), e_.get());
        return out;
    }
    ReadBatch(const ReadBatch &) = delete;
    ReadBatch &operator=(const ReadBatch &) = delete;
    dk_reads *get() const { return r_; }

private:
    Engine &e_;
    dk_reads *r_ = nullptr;
};

// k-mer -> count table (unordered, like the reference's HashMap)
struct KmerCounts {
    std::vector<uint64_t> lo, hi;
    std::vector<uint32_t> count;
    dk_stats stats{};
    size_t size() const { return lo.size(); }
};

// counter.rs `KmerSet`: insert sequences, test membership.  Config::set_kind picks what it holds:
// a blocked Bloom filter (no false negatives, tunable false positives) or an exact set
// (insert throws Error{DK_ERR_SET_FULL} when the table is too small; popcount() = k-mers held)
class KmerSet {
public:
    explicit KmerSet(Engine &e) : e_(e) { check(dk_set_create(e.get(), &s_), e.get()); }
    ~KmerSet() { dk_set_destroy(s_); }
    KmerSet(const KmerSet &) = delete;
    KmerSet &operator=(const KmerSet &) = delete;

    dk_stats insert(const ReadBatch &b)
    {
        dk_stats st{};
        check(dk_set_insert(s_, b.get(), &st), e_.get());
        return st;
    }
    dk_stats insert_sequences(const std::vector<std::string> &reads)
    {
        ReadBatch b(e_, reads);
        return insert(b);
    }
    std::vector<uint8_t> contains(const std::vector<uint64_t> &lo, const std::vector<uint64_t> &hi = {})
    {
        std::vector<uint8_t> out(lo.size());
        check(dk_set_contains(s_, lo.data(), e_.k() > 32 ? hi.data() : nullptr, lo.size(), out.data()), e_.get());
        return out;
    }
    uint64_t popcount()
    {
        uint64_t n = 0;
        check(dk_set_popcount(s_, &n), e_.get());
        return n;
    }
    void clear() { check(dk_set_clear(s_), e_.get()); }
    void save(const std::string &path) { check(dk_set_save(s_, path.c_str()), e_.get()); }
    void load(const std::string &path) { check(dk_set_load(s_, path.c_str()), e_.get()); }
    dk_set *get() const { return s_; }

private:
    Engine &e_;
    dk_set *s_ = nullptr;
};

// counter.rs `KmerCounter`: per-k-mer counts of a sample, and the child-only pass against a KmerSet
class KmerCounter {
public:
    explicit KmerCounter(Engine &e) : e_(e) {}
    KmerCounts count(const ReadBatch &b) { return run(nullptr, b); }
    KmerCounts child_only(const ReadBatch &child, const KmerSet &parents) { return run(parents.get(), child); }
    // device-resident tables for multi-batch samples: probe each batch, merge the handles, fetch once
    dk_result *child_only_device(const ReadBatch &child, const KmerSet &parents)
    {
        dk_result *res = nullptr;
        check(dk_probe(e_.get(), parents.get(), child.get(), &res, nullptr), e_.get());
        return res;
    }
    KmerCounts merge(const std::vector<dk_result *> &tables, uint32_t min_count)
    {
        KmerCounts out;
        dk_result *res = nullptr;
        check(dk_result_merge(e_.get(), tables.data(), (uint32_t)tables.size(), min_count, &res, &out.stats), e_.get());
        return fetch(res, out);
    }

private:
    KmerCounts run(dk_set *s, const ReadBatch &b)
    {
        KmerCounts out;
        dk_result *res = nullptr;
        check(dk_probe(e_.get(), s, b.get(), &res, &out.stats), e_.get());
        return fetch(res, out);
    }
    KmerCounts fetch(dk_result *res, KmerCounts &out)
    {
        uint64_t n = 0;
        dk_status st = dk_result_size(res, &n);
        if (st == DK_OK) {
            out.lo.resize(n);
            out.hi.resize(n);
            out.count.resize(n);
            st = dk_result_copy(res, out.lo.data(), out.hi.data(), out.count.data());
        }
        dk_result_destroy(res);
        check(st, e_.get());
        return out;
    }
    Engine &e_;
};

// A child that arrives in many batches (dk_accum_*): add() every batch, finish() once -- counts and min_count are
// exact over the whole sample.  windows > 1: stream the sample once per window, reset(w) before pass w.
class ChildAccumulator {
public:
    ChildAccumulator(Engine &e, const KmerSet *parents, uint64_t capacity_records, uint32_t window_count = 1) : e_(e)
    {
        check(dk_accum_create(e.get(), parents ? parents->get() : nullptr, 0, window_count, capacity_records, &a_), e.get());
    }
    ~ChildAccumulator() { dk_accum_destroy(a_); }
    ChildAccumulator(const ChildAccumulator &) = delete;
    ChildAccumulator &operator=(const ChildAccumulator &) = delete;
    dk_stats add(const ReadBatch &b)
    {
        dk_stats st;
        check(dk_accum_add(a_, b.get(), &st), e_.get());
        return st;
    }
    void reset(uint32_t window_index) { check(dk_accum_reset(a_, window_index), e_.get()); }
    // appends the window's table to `out` (k-mers of different windows are disjoint).  exchange = true (multi-GPU, collective,
    // after dk_comm_init on the engine): the ranks swap unit ranges first and `out` receives this rank's share of the hash space
    void finish(uint32_t min_count, KmerCounts &out, bool exchange = false)
    {
        dk_result *res = nullptr;
        dk_stats st;
        check(exchange ? dk_accum_exchange_finish(a_, min_count, &res, &st, nullptr) : dk_accum_finish(a_, min_count, &res, &st), e_.get());
        uint64_t n = 0;
        dk_status rc = dk_result_size(res, &n);
        const size_t at = out.lo.size();
        if (rc == DK_OK && n) {
            out.lo.resize(at + n);
            out.hi.resize(at + n);
            out.count.resize(at + n);
            rc = dk_result_copy(res, out.lo.data() + at, out.hi.data() + at, out.count.data() + at);
        }
        dk_result_destroy(res);
        check(rc, e_.get());
        out.stats.n_reads = st.n_reads;
        out.stats.n_bases = st.n_bases;
        out.stats.n_windows = st.n_windows;
        out.stats.n_valid = st.n_valid;
        out.stats.n_absent += st.n_absent;
        out.stats.n_distinct += st.n_distinct;
        out.stats.n_emitted += st.n_emitted;
    }

private:
    Engine &e_;
    dk_accum *a_ = nullptr;
};

}  // namespace dk_host
#endif
