"""GPU tests of the child-only accumulator (dk_accum_*): a sample that arrives in many batches, optionally
streamed in several hash-window passes, must give exactly the table the oracle computes for the whole sample.

PARITY UNPINNED vs the reference's Rust code (no source / fixtures in /root/reference); the oracle is the
written spec of DESIGN.md section 2."""
import numpy as np
import pytest

from conftest import random_reads, related_trio
from oracle import orc

pytestmark = pytest.mark.gpu


def dk():
    import denovo_kmer_amd
    return denovo_kmer_amd


def table_of(res):
    hi, lo, cnt = res.to_host(sort=False)
    return {(int(h), int(l)): int(c) for h, l, c in zip(hi, lo, cnt)}


def oracle_table(km, cn):
    return {(int(h), int(l)): int(c) for h, l, c in zip(km["hi"], km["lo"], cn)}


def ragged_batches(reads, cuts):
    edges = [0] + [int(len(reads) * c) for c in cuts] + [len(reads)]
    return [reads[a:b] for a, b in zip(edges[:-1], edges[1:])]


@pytest.mark.parametrize("mode", ["direct", "bucketed"])
@pytest.mark.parametrize("k,set_kind,log2_bits", [(31, "bloom", 24), (45, "bloom", 24), (31, "exact", 26), (51, "exact", 26)])
@pytest.mark.parametrize("window_count,sub_split", [(1, 0), (4, 0), (1, 2), (2, 1)])
def test_accumulated_batches_equal_the_whole_sample(mode, k, set_kind, log2_bits, window_count, sub_split):
    d = dk()
    rng = np.random.default_rng(1234 + k)
    parents, child = related_trio(rng, genome_len=20000, n_reads=700, read_len=130)
    child = child + child[:200] + ["", "ACGT", "N" * 40]           # repeats (counts > 1), degenerate reads
    batches = ragged_batches(child, (0.1, 0.11, 0.4, 0.41, 0.8)) + [[]]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    if set_kind == "bloom":
        f = orc.new_filter(log2_bits)
        orc.bloom_insert(f, log2_bits, 4, 5, k, True, pseq, poff)
    with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=5, mode=mode, set_kind=set_kind) as eng:
        eng.set_option("sub_split", sub_split)          # segment kernels share partition regions (what 2^38-bit sets do)
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        acc = d.ChildAccumulator(eng, ks, capacity_records=200_000, window_count=window_count)
        for mc in (1, 2):
            if set_kind == "bloom":
                km, cn, ost = orc.bloom_probe(f, log2_bits, 4, 5, k, True, cseq, coff, mc)
            else:
                km, cn, ost = orc.exact_child_only(k, True, pseq, poff, cseq, coff, mc)
            got, n_absent, n_distinct = {}, 0, 0
            for w in range(window_count):
                acc.reset(w)
                for b in batches:
                    st = acc.add(d.ReadBatch.from_sequences(eng, b))
                    names = [n for n, _ in eng.timings()["stages"]]
                    if b and mode == "bucketed":
                        assert names[0] == "scan_part" and "overflow_redo" not in names, names
                res = acc.finish(min_count=mc)
                t = table_of(res)
                assert not (set(t) & set(got)), "hash windows overlap"
                got.update(t)
                assert res.stats["n_valid"] == ost["n_valid"] and res.stats["n_windows"] == ost["n_windows"]
                assert res.stats["n_emitted"] == len(t)
                n_absent += res.stats["n_absent"]
                n_distinct += res.stats["n_distinct"]
                res.close()
            assert got == oracle_table(km, cn)
            assert n_absent == ost["n_absent"] and n_distinct == ost["n_distinct"]
            assert max(got.values()) >= 2
        acc.close()
        ks.close()


@pytest.mark.parametrize("k,set_kind", [(31, "bloom"), (45, "bloom"), (31, "exact")])
@pytest.mark.parametrize("slabs,sub_split,window_count,min_u,plain", [(2, 0, 1, 8, 0), (2, 1, 1, 8, 1), (2, 0, 2, 9, 0), (0, 0, 1, 8, 0), (2, 0, 1, 0, 0)])
def test_slab_wise_accumulate_and_packed_units(k, set_kind, slabs, sub_split, window_count, min_u, plain):
    """the geometry of a whole-genome child pass on a small set: the level-1 bins are taken slab by slab (option "slabs"),
    with and without the sub-segment split, and the accumulator keeps 6-byte packed records (units of >= 16 prefix bits:
    option "accum_min_u"; "accum_plain" = 8-byte records); direct mode appends to the same units through global cursors"""
    d = dk()
    rng = np.random.default_rng(77 + k)
    parents, child = related_trio(rng, genome_len=30000, n_reads=900, read_len=140)
    child = child + child[:300] + ["", "ACGT", "N" * 40]
    batches = ragged_batches(child, (0.2, 0.21, 0.7)) + [[]]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    log2_bits = 27 if set_kind == "bloom" else 28        # 2^8 / 2^9 segments: 16 level-1 bins, two slabs of 8
    if set_kind == "bloom":
        f = orc.new_filter(log2_bits)
        orc.bloom_insert(f, log2_bits, 4, 5, k, True, pseq, poff)
        km, cn, ost = orc.bloom_probe(f, log2_bits, 4, 5, k, True, cseq, coff, 1)
    else:
        km, cn, ost = orc.exact_child_only(k, True, pseq, poff, cseq, coff, 1)
    for mode in ("bucketed", "direct"):
        with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=5, mode=mode, set_kind=set_kind) as eng:
            for name, val in (("slabs", slabs), ("sub_split", sub_split), ("accum_min_u", min_u), ("accum_plain", plain)):
                eng.set_option(name, val)
            ks = d.KmerSet(eng)
            ks.insert_sequences(parents)
            if set_kind == "bloom":
                assert np.array_equal(ks.to_host(), f)          # (the insert ran slab-wise too)
            acc = d.ChildAccumulator(eng, ks, capacity_records=400_000, window_count=window_count)
            n_units, cap, rb = acc.geometry()
            assert rb == (16 if k > 32 else 6 if (min_u >= 8 and not plain) else 8)
            got, n_absent = {}, 0
            for w in range(window_count):
                acc.reset(w)
                for b in batches:
                    acc.add(d.ReadBatch.from_sequences(eng, b))
                    names = [n for n, _ in eng.timings()["stages"]]
                    if b and mode == "bucketed":
                        assert names[:3] == ["scan_part", "repart", "seg_exact_probe" if set_kind == "exact" else "seg_probe"], names
                        assert "overflow_redo" not in names, names
                res = acc.finish(min_count=1)
                got.update(table_of(res))
                n_absent += res.stats["n_absent"]
                res.close()
            assert got == oracle_table(km, cn)
            assert n_absent == ost["n_absent"]
            acc.close()
            ks.close()


@pytest.mark.parametrize("seed", [5, 6, 7, 8])
def test_partition_failure_in_a_later_slab_is_redone_exactly(seed):
    """a slab-wise accumulate whose partition loses records in slab F (overflow list full: here 1000 entries against
    40 000 copies of poly-A, all in one region) has appended the slabs before F; the rest of the hash range is redone by
    the direct family and nothing is counted twice.  Two slabs: poly-A's slab is the top bit of its hash (seeds 5, 6: the
    second slab -- the first one stays; seeds 7, 8: the first -- everything is redone)."""
    d = dk()
    k = 21
    rng = np.random.default_rng(seed)
    parents = random_reads(rng, 300, 150, 151)
    # (one poly-A read in eleven: spread over the tiles, its records fit the level-1 pieces and overflow only their region)
    base = random_reads(rng, 3000, 150, 151) + parents[:100]
    child = []
    for i, rd in enumerate(base):
        child.append(rd)
        if i % 10 == 0:
            child.append("A" * 150)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(27)
    orc.bloom_insert(f, 27, 4, seed, k, True, pseq, poff)
    km, cn, ost = orc.bloom_probe(f, 27, 4, seed, k, True, cseq, coff, 1)
    in_second_slab = bool(orc.hash_kmer(0, 0, k, seed) >> 63)
    assert in_second_slab == (seed in (5, 6))
    for plain in (0, 1):
        with d.Engine(k=k, filter_log2_bits=27, n_hashes=4, seed=seed, mode="bucketed") as eng:
            for name, val in (("slabs", 2), ("ovf_cap", 1000), ("accum_min_u", 8), ("accum_plain", plain)):
                eng.set_option(name, val)
            ks = d.KmerSet(eng)
            ks.insert_sequences(parents)
            acc = d.ChildAccumulator(eng, ks, capacity_records=60_000_000)
            st = acc.add(d.ReadBatch.from_sequences(eng, child))
            names = [n for n, _ in eng.timings()["stages"]]
            assert "overflow_redo" in names and ("slab_partial" in names) == in_second_slab, names
            assert st["n_valid"] == ost["n_valid"] and st["n_absent"] == ost["n_absent"]
            res = acc.finish()
            assert table_of(res) == oracle_table(km, cn)
            res.close()
            acc.close()
            ks.close()


@pytest.mark.parametrize("mode", ["direct", "bucketed"])
@pytest.mark.parametrize("k", [31, 51])
def test_accumulating_kmer_counter_without_a_set(mode, k):
    """parents=None: every k-mer of every batch counts (KmerCounter over a sample in batches)"""
    d = dk()
    rng = np.random.default_rng(99)
    reads = random_reads(rng, 500, 100, 160, n_rate=0.002)
    reads = reads + reads[:120]
    seq, off = orc.concat_reads(reads)
    km, cn, ost = orc.count_reads(k, True, seq, off)
    with d.Engine(k=k, filter_log2_bits=24, seed=3, mode=mode) as eng:
        acc = d.ChildAccumulator(eng, None, capacity_records=200_000, window_count=2)
        got = {}
        for w in range(2):
            acc.reset(w)
            for b in ragged_batches(reads, (0.3, 0.31, 0.9)):
                acc.add(d.ReadBatch.from_sequences(eng, b))
            res = acc.finish()
            got.update(table_of(res))
            res.close()
        assert got == oracle_table(km, cn)
        acc.close()


def test_full_units_spill_to_the_overflow_list_and_a_full_list_fails_loudly():
    d = dk()
    rng = np.random.default_rng(5)
    parents = random_reads(rng, 20, 150, 151)
    child = random_reads(rng, 260, 150, 151)                       # ~31 K absent occurrences over 2 units of ~270
    child = child + child[:60]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(20)
    orc.bloom_insert(f, 20, 3, 12, 31, True, pseq, poff)
    km, cn, ost = orc.bloom_probe(f, 20, 3, 12, 31, True, cseq, coff, 1)
    for mode in ("direct", "bucketed"):
        with d.Engine(k=31, filter_log2_bits=20, n_hashes=3, seed=12, mode=mode) as eng:
            ks = d.KmerSet(eng)
            ks.insert_sequences(parents)
            acc = d.ChildAccumulator(eng, ks, capacity_records=1)   # units of a few hundred records, 65536 overflow entries
            for b in ragged_batches(child, (0.5,)):
                acc.add(d.ReadBatch.from_sequences(eng, b))
            res = acc.finish()
            assert "acc_ovf_sort" in [n for n, _ in eng.timings()["stages"]]
            assert table_of(res) == oracle_table(km, cn)
            assert res.stats["n_absent"] == ost["n_absent"]
            res.close()
            # a sample far beyond the capacity: the call that fills the overflow list fails, later calls refuse
            big = d.ReadBatch.from_sequences(eng, random_reads(rng, 700, 150, 151))
            with pytest.raises(d.DkError) as ei:
                acc.add(big)
            assert ei.value.status == 6 and "accumulator full" in str(ei.value)
            with pytest.raises(d.DkError):
                acc.add(big)
            with pytest.raises(d.DkError):
                acc.finish()
            acc.reset(0)
            acc.add(d.ReadBatch.from_sequences(eng, child[:50]))
            assert len(acc.finish()) > 0
            acc.close()
            ks.close()


def test_accumulator_arguments_are_checked():
    d = dk()
    with d.Engine(k=31, filter_log2_bits=24) as eng:          # 32 segments
        for kw in (dict(window_count=3), dict(window_count=64), dict(window_index=2, window_count=2), dict(capacity_records=0)):
            args = dict(capacity_records=1000, window_index=0, window_count=1)
            args.update(kw)
            with pytest.raises(d.DkError):
                d.ChildAccumulator(eng, None, **args)
        with pytest.raises(d.DkError):                        # far more records per window than its units can count
            d.ChildAccumulator(eng, None, capacity_records=10**12)
        acc = d.ChildAccumulator(eng, None, capacity_records=1000, window_count=2)
        with pytest.raises(d.DkError):
            acc.reset(2)
        assert acc.device_bytes() > 0 and len(acc.finish()) == 0
        acc.close()


def _checksum(res):
    import hashlib
    hi, lo, cnt = res.to_host(sort=True)
    h = hashlib.sha256()
    for a in (hi, lo, cnt):
        h.update(np.ascontiguousarray(a).tobytes())
    return len(lo), h.hexdigest()


@pytest.mark.timeout(1500)
def test_many_batches_against_a_2_to_the_36_bit_filter():
    """configs[2] in miniature: an 8-GiB parent filter (2^17 segments), parents inserted in 16 batches, the child
    streamed as 32 batches -- once in one window, once in two hash-window passes -- against (a) the oracle on the
    whole child, (b) one dk_probe call over the whole child (union of batches == whole), (c) the direct family."""
    d = dk()
    k, log2_bits, nh, seed = 31, 36, 4, 20260313
    n_parent, n_child, n_pb, n_cb = 262144, 1048576, 8, 32
    ocfg = orc.synth_cfg(genome_len=4 << 20)
    gcfg = d.synth_config(genome_len=4 << 20)
    f = orc.new_filter(log2_bits)
    for smp in (0, 1):
        seq, off = orc.synth_reads(ocfg, smp, 0, n_parent)
        orc.bloom_insert(f, log2_bits, nh, seed, k, True, seq, off)
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_child)
    want = {}
    for mc in (1, 3):
        km, cn, ost = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff, mc, n_threads=16)
        order = np.lexsort((km["lo"], km["hi"]))
        want[mc] = (km["hi"][order], km["lo"][order], cn[order], ost)
    del f, cseq, coff
    sums = {}
    for mode in ("bucketed", "direct"):
        with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=seed, mode=mode) as eng:
            eng.set_option("multiplicity_hint", 2)
            ks = d.KmerSet(eng)
            for smp in (0, 1):
                for b in range(n_pb):
                    ks.insert_reads(d.ReadBatch.synth(eng, gcfg, smp, b * (n_parent // n_pb), n_parent // n_pb))
            for windows in ((1, 2) if mode == "bucketed" else (1,)):
                acc = d.ChildAccumulator(eng, ks, capacity_records=int(0.2 * n_child * 120) // windows, window_count=windows)
                for mc in (1, 3):
                    parts = []
                    for w in range(windows):
                        acc.reset(w)
                        for b in range(n_cb):
                            acc.add(d.ReadBatch.synth(eng, gcfg, 2, b * (n_child // n_cb), n_child // n_cb))
                            if mode == "bucketed":
                                assert [n for n, _ in eng.timings()["stages"]][:2] == ["scan_part", "repart"]
                        parts.append(acc.finish(min_count=mc))
                    hi = np.concatenate([p.to_host(sort=False)[0] for p in parts])
                    lo = np.concatenate([p.to_host(sort=False)[1] for p in parts])
                    cnt = np.concatenate([p.to_host(sort=False)[2] for p in parts])
                    order = np.lexsort((lo, hi))
                    whi, wlo, wcn, ost = want[mc]
                    assert np.array_equal(hi[order], whi) and np.array_equal(lo[order], wlo) and np.array_equal(cnt[order], wcn)
                    assert sum(p.stats["n_absent"] for p in parts) == ost["n_absent"]
                    assert all(p.stats["n_valid"] == ost["n_valid"] for p in parts)
                    for p in parts:
                        p.close()
                acc.close()
            if mode == "bucketed":
                # union of batches == whole: one probe call over the whole child
                whole = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, n_child), ks)
                whi, wlo, wcn, ost = want[1]
                h2, l2, c2 = whole.to_host(sort=True)
                assert np.array_equal(h2, whi) and np.array_equal(l2, wlo) and np.array_equal(c2, wcn)
                whole.close()
            sums[mode] = ks.popcount()
            ks.close()
    assert sums["bucketed"] == sums["direct"]


def test_min_count_table_too_small_is_recounted_exactly():
    """with min_count > 1 the result table is sized for a sixteenth of the upper bound; a sample where nearly every
    k-mer passes the threshold overruns it and is counted a second time with the room the first run tallied"""
    d = dk()
    rng = np.random.default_rng(8)
    reads = random_reads(rng, 30000, 150, 151)
    seq, off = orc.concat_reads(reads + reads)
    km, cn, ost = orc.count_reads(31, True, seq, off)
    assert int(cn.min()) >= 2
    with d.Engine(k=31, filter_log2_bits=26, seed=1, mode="bucketed") as eng:
        acc = d.ChildAccumulator(eng, None, capacity_records=8_000_000)
        for _ in range(2):
            acc.add(d.ReadBatch.from_sequences(eng, reads))
        res = acc.finish(min_count=2)
        names = [n for n, _ in eng.timings()["stages"]]
        assert names[-2:] == ["seg_count", "seg_count_redo"], names
        hi, lo, cnt = res.to_host(sort=True)
        assert np.array_equal(lo, km["lo"]) and np.array_equal(hi, km["hi"]) and np.array_equal(cnt, cn)
        assert res.stats["n_distinct"] == ost["n_distinct"] == len(lo)
        res.close()
        acc.close()


@pytest.mark.parametrize("window_count", [1, 2])
def test_heavy_hitters_reach_the_accumulator_through_the_overflow_paths(window_count):
    """hundreds of thousands of copies of one absent k-mer overflow their partition region (handled one by one:
    ovf_probe -> ovf_append) and then their counting unit (the accumulator's overflow list, sorted in at finish);
    a k-mer that is present in the set overflows the region too and must not be counted"""
    d = dk()
    k = 21
    parents = ["A" * 150] * 2000 + ["ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGATTAGC" * 3] * 10
    child = ["A" * 150] * 3000 + ["C" * 150] * 3000 + parents[-10:] + ["GATTACA" * 20] * 5
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(24)
    orc.bloom_insert(f, 24, 4, 5, k, True, pseq, poff)
    with d.Engine(k=k, filter_log2_bits=24, n_hashes=4, seed=5, mode="bucketed") as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        # (the overflow list holds capacity / 64 entries: room for the 390 000 copies of poly-C beyond their unit)
        acc = d.ChildAccumulator(eng, ks, capacity_records=30_000_000, window_count=window_count)
        for mc in (1, 100):
            km, cn, ost = orc.bloom_probe(f, 24, 4, 5, k, True, cseq, coff, mc)
            got, seen_ovf = {}, False
            for w in range(window_count):
                acc.reset(w)
                for b in ragged_batches(child, (0.3, 0.7)):
                    acc.add(d.ReadBatch.from_sequences(eng, b))
                    names = [n for n, _ in eng.timings()["stages"]]
                    assert "overflow_redo" not in names, names
                    seen_ovf = seen_ovf or "ovf_append" in names
                res = acc.finish(min_count=mc)
                got.update(table_of(res))
                res.close()
            assert seen_ovf, "the poly-C batch was expected to overflow its partition region"
            assert got == oracle_table(km, cn)
            assert max(got.values()) == 3000 * 130
        acc.close()
        ks.close()
