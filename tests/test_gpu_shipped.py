"""GPU tests at the geometry bench.py ships for configs[2] / configs[3]: a 2^39-bit (64-GiB) parent filter, i.e. 2^20
segments -- 1024 x 1024 partition regions, taken slab by slab -- the whole-hash-space accumulator with 6-byte packed
records (window_count 1: configs[3]'s per-rank shape and this round's configs[2]) and the two-window pass of round 2.

The oracle checks a 2 M-read subset against the downloaded 64-GiB filter; the direct family -- an independent
implementation on the same filter -- checks the whole input.  Also: the engine's reserved arena and plan report, and the
overlapped upload of read batches.

PARITY UNPINNED vs the reference's Rust code (no source / fixtures in /root/reference); the oracle is the written spec of
DESIGN.md section 2."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


def dk():
    import denovo_kmer_amd
    return denovo_kmer_amd


def _checksum(res):
    """order-independent digest of a (k-mer, count) table"""
    hi, lo, cnt = res.to_host(sort=False)
    with np.errstate(over="ignore"):
        mixed = (lo ^ (lo >> np.uint64(29))) * np.uint64(0x9E3779B97F4A7C15)
        return (int(len(lo)), int(cnt.astype(np.uint64).sum()),
                int(mixed.sum(dtype=np.uint64)), int((mixed * cnt.astype(np.uint64)).sum(dtype=np.uint64)))


def _sum_checksums(cs):
    return (sum(c[0] for c in cs), sum(c[1] for c in cs), sum(c[2] for c in cs) % 2**64, sum(c[3] for c in cs) % 2**64)


@pytest.mark.timeout(1700)
def test_configs2_geometry_insert_and_accumulate_against_oracle_and_direct_family():
    import torch
    d = dk()
    k, log2_bits, nh, seed = 31, 39, 4, 20260313
    n_parent, n_child, n_sub = 4_000_000, 4_000_000, 2_000_000       # per batch: two parent batches, two child batches
    gcfg = d.synth_config(seed=seed, genome_len=64 << 20)
    ocfg = orc.synth_cfg(seed=seed, genome_len=64 << 20)
    words = (1 << log2_bits) // 64
    with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=seed, mode="bucketed") as eng:
        eng.reserve(60 << 30)                                         # workspace arena: no hipMalloc inside the calls below
        eng.set_option("slab_mb", 512)          # 4 M-read batches take several slabs, as the 48 M-read ones of bench.py do at 12 GiB
        fb = torch.zeros(words, dtype=torch.int64, device="cuda")
        fd = torch.zeros(words, dtype=torch.int64, device="cuda")
        ks = d.KmerSet(eng, device_ptr=fb.data_ptr(), keepalive=fb)
        kd = d.KmerSet(eng, device_ptr=fd.data_ptr(), keepalive=fd)
        # ---- parent build: bucketed (1024 x 1024 regions, slab-wise) against the direct family, bit for bit --------
        for smp in (0, 1):
            pb = d.ReadBatch.synth(eng, gcfg, smp, 0, n_parent)
            eng.set_option("mode", 2)
            eng.set_option("l2_packed", smp)               # the second parent through the packed-region kernels
            ks.insert_reads(pb)
            names = [n for n, _ in eng.timings()["stages"]]
            assert names[:3] == ["scan_part", "repart", "seg_insert"] and "overflow_redo" not in names, names
            assert (eng.info("plan_levels"), eng.info("plan_b1"), eng.info("plan_b2"), eng.info("plan_sbits")) == (2, 10, 10, 0)
            assert eng.info("plan_slabs") > 1 and eng.info("plan_segment_bits") == 20
            eng.set_option("mode", 1)
            kd.insert_reads(pb)
            assert [n for n, _ in eng.timings()["stages"]] == ["insert_direct"]
            pb.close()
        assert torch.equal(fb, fd), "bucketed and direct parent builds differ"
        pop = ks.popcount()
        assert 0 < pop < (1 << log2_bits) // 100
        kd.close()
        del fd, kd
        torch.cuda.empty_cache()
        eng.set_option("mode", 0)
        eng.set_option("l2_packed", 0)
        # ---- the whole child through the direct family (one dk_probe per batch, merged) = the reference table ------
        eng.set_option("mode", 1)
        parts = [d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, b * n_child, n_child), ks) for b in range(2)]
        for mc in (1, 2):
            merged = d.KmerCounter(eng).merge(parts, min_count=mc)
            want = (_checksum(merged), merged.stats["n_distinct"])
            merged.close()
            if mc == 1:
                want1 = want
            else:
                want2 = want
        n_absent_direct = sum(p.stats["n_absent"] for p in parts)
        for p in parts:
            p.close()
        eng.set_option("mode", 2)
        # ---- accumulate: one window (6-byte packed units, the shape of configs[2] now and of a configs[3] rank),
        #      two windows (round 2's pass), and one window over the 512 x 1024 + sub-segment-split layout of round 2 ----
        for windows, scan_bits, l2_packed in ((1, 0, 0), (2, 0, 0), (1, 9, 0), (1, 0, 1), (2, 9, 1)):
            eng.set_option("scan_bits", scan_bits)
            eng.set_option("repart_pieces", 2 if windows == 1 else 1)       # concatenated pieces / piece by piece
            eng.set_option("l2_packed", l2_packed)     # 6-byte records in the level-2 regions too (off by default: slower)
            acc = d.ChildAccumulator(eng, ks, capacity_records=int(0.2 * 2 * n_child * 120 / windows), window_count=windows)
            n_units, cap, rb = acc.geometry()
            assert rb == 6
            cs1, cs2, nd, na = [], [], 0, 0
            for w in range(windows):
                acc.reset(w)
                for b in range(2):
                    st = acc.add(d.ReadBatch.synth(eng, gcfg, 2, b * n_child, n_child))
                    names = [n for n, _ in eng.timings()["stages"]]
                    assert names[:3] == ["scan_part", "repart", "seg_probe"] and "overflow_redo" not in names, names
                    assert eng.info("plan_slabs") > 1
                    assert eng.info("plan_sbits") == (1 if scan_bits == 9 and windows == 1 else 0)
                    assert (eng.info("plan_b1"), eng.info("plan_b2")) == ((9, 10) if scan_bits == 9 or windows == 2 else (10, 10))
                    na += st["n_absent"]
                r1 = acc.finish(min_count=1)
                r2 = acc.finish(min_count=2)
                cs1.append(_checksum(r1))
                cs2.append(_checksum(r2))
                nd += r1.stats["n_distinct"]
                r1.close()
                r2.close()
            assert na == n_absent_direct
            assert (_sum_checksums(cs1), nd) == want1, (windows, scan_bits)
            assert _sum_checksums(cs2) == want2[0], (windows, scan_bits)
            acc.close()
        eng.set_option("scan_bits", 0)
        eng.set_option("l2_packed", 0)
        eng.set_option("repart_pieces", 0)
        # ---- the oracle on a subset, against the downloaded 64-GiB filter ------------------------------------------
        filt = ks.to_host()
        assert int(np.bitwise_count(filt[:1 << 20]).sum()) > 0
        cseq, coff = orc.synth_reads(ocfg, 2, 0, n_sub)
        km, cn, ost = orc.bloom_probe(filt, log2_bits, nh, seed, k, True, cseq, coff, 1, n_threads=16)
        del filt, cseq, coff
        acc = d.ChildAccumulator(eng, ks, capacity_records=int(0.2 * n_sub * 120))
        st = acc.add(d.ReadBatch.synth(eng, gcfg, 2, 0, n_sub))
        assert st["n_valid"] == ost["n_valid"] and st["n_absent"] == ost["n_absent"]
        res = acc.finish(min_count=1)
        hi, lo, cnt = res.to_host(sort=True)
        assert np.array_equal(lo, km["lo"]) and np.array_equal(hi, km["hi"]) and np.array_equal(cnt, cn)
        assert res.stats["n_distinct"] == ost["n_distinct"]
        res.close()
        acc.close()
        ks.close()


def test_reserved_arena_serves_every_allocation_and_reports():
    d = dk()
    rng = np.random.default_rng(3)
    alphabet = np.array(list("ACGT"))
    reads = ["".join(alphabet[rng.integers(0, 4, size=150)]) for _ in range(3000)]
    with d.Engine(k=31, filter_log2_bits=27, seed=2, mode="bucketed") as eng:
        assert eng.info("pool_bytes_reserved") == 0
        eng.reserve(1 << 30)
        assert eng.info("pool_bytes_reserved") == 1 << 30
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads[:1000])
        cb = d.ReadBatch.from_sequences(eng, reads)
        res = d.KmerCounter(eng).child_only(cb, ks)
        cb.close()
        f = orc.new_filter(27)
        pseq, poff = orc.concat_reads(reads[:1000])
        cseq, coff = orc.concat_reads(reads)
        orc.bloom_insert(f, 27, 4, 2, 31, True, pseq, poff)
        km, cn, ost = orc.bloom_probe(f, 27, 4, 2, 31, True, cseq, coff, 1)
        assert res.stats["n_absent"] == ost["n_absent"] >= 2000 * 120 - 20      # (a handful of false positives)
        assert np.array_equal(ks.to_host(), f)
        assert eng.info("pool_bytes_cached") == 0, "an allocation went past the arena"
        assert 0 < eng.info("pool_bytes_in_use") <= eng.info("pool_bytes_peak") <= 1 << 30
        with pytest.raises(d.DkError):
            eng.info("no_such_thing")
        res.close()
        ks.close()
        assert eng.info("pool_bytes_in_use") == 0
        eng.reserve(0)                                  # nothing lives in the arena: handed back
        assert eng.info("pool_bytes_reserved") == 0
        # a request beyond the arena falls through to the caching pool
        eng.reserve(1 << 20)
        ks = d.KmerSet(eng)
        assert eng.info("pool_bytes_cached") >= (1 << 27) // 8
        ks.close()


def test_batches_uploaded_while_the_previous_one_is_probed_give_the_oracle_result():
    """dk_reads_from_packed_async: batch i + 1 is copied from pinned host memory on the copy stream while batch i is
    accumulated on the engine's stream; every consumer waits for its batch on the device"""
    d = dk()
    from conftest import related_trio
    rng = np.random.default_rng(17)
    parents, child = related_trio(rng, genome_len=30000, n_reads=2400, read_len=150)
    k, log2_bits = 31, 27
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(log2_bits)
    orc.bloom_insert(f, log2_bits, 4, 5, k, True, pseq, poff)
    km, cn, ost = orc.bloom_probe(f, log2_bits, 4, 5, k, True, cseq, coff, 1)
    batches = [child[i:i + 400] for i in range(0, len(child), 400)]
    for mode in ("bucketed", "direct"):
        with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=5, mode=mode) as eng:
            ks = d.KmerSet(eng)
            ks.insert_sequences(parents)
            acc = d.ChildAccumulator(eng, ks, capacity_records=400_000)
            pinned, meta = [], []
            for b in batches:
                seq, off = orc.concat_reads(b)
                bases, mask, n_bases = orc.pack_reads(seq, off)
                pp = d.PinnedPacked(n_bases)
                pp.bases[:] = bases[:pp.n_bwords]
                pp.mask[:] = mask[:pp.n_mwords]
                pinned.append(pp)
                meta.append((n_bases, len(b), orc.n_windows(off, k)))
            # two uploads in flight ahead of the batch being accumulated
            inflight = [d.ReadBatch.from_packed_async(eng, pinned[i], *meta[i]) for i in range(2)]
            for i in range(len(batches)):
                rb = inflight.pop(0)
                if i + 2 < len(batches):
                    inflight.append(d.ReadBatch.from_packed_async(eng, pinned[i + 2], *meta[i + 2]))
                acc.add(rb)
                rb.wait()
                rb.close()
            res = acc.finish()
            hi, lo, cnt = res.to_host(sort=True)
            assert np.array_equal(lo, km["lo"]) and np.array_equal(hi, km["hi"]) and np.array_equal(cnt, cn)
            assert res.stats["n_absent"] == ost["n_absent"]
            # a batch that is still uploading can be downloaded, destroyed or never used
            rb = d.ReadBatch.from_packed_async(eng, pinned[0], *meta[0])
            gb, gm, gn = rb.download()
            assert np.array_equal(gb[:pinned[0].n_bwords], pinned[0].bases) and np.array_equal(gm[:pinned[0].n_mwords], pinned[0].mask)
            rb.close()
            d.ReadBatch.from_packed_async(eng, pinned[1], *meta[1]).close()
            res.close()
            acc.close()
            ks.close()
            for pp in pinned:
                pp.close()


def test_counting_a_sample_of_duplicates_does_not_fall_off_a_cliff():
    """every absent k-mer twice (a batch accumulated twice) against a 2^36-bit set: all records take the table path of
    seg_count and its crowded rounds split.  The split must spread the keys (until round 3 the round selector used hash
    bits that are constant inside a unit of a large set: the rounds multiplied 4096-fold and this count took seconds)"""
    d = dk()
    n = 2_000_000
    gcfg = d.synth_config(genome_len=64 << 20)
    with d.Engine(k=31, filter_log2_bits=36, n_hashes=4, seed=1, mode="bucketed") as eng:
        ks = d.KmerSet(eng)
        for smp in (0, 1):
            ks.insert_reads(d.ReadBatch.synth(eng, gcfg, smp, 0, n))
        rb = d.ReadBatch.synth(eng, gcfg, 2, 0, n)
        acc = d.ChildAccumulator(eng, ks, capacity_records=int(0.5 * n * 120))
        acc.add(rb)
        once = acc.finish(min_count=1)
        t_once = dict(eng.timings()["stages"])["seg_count"]
        acc.add(rb)
        twice = acc.finish(min_count=1)
        t_twice = dict(eng.timings()["stages"])["seg_count"]
        (h1, l1, c1), (h2, l2, c2) = once.to_host(), twice.to_host()
        assert np.array_equal(l1, l2) and np.array_equal(h1, h2) and np.array_equal(2 * c1, c2)
        assert t_twice < 40 * max(t_once, 0.5), (t_once, t_twice)
        for r in (once, twice):
            r.close()
        acc.close()
        ks.close()
