"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Bit-exact everywhere (integer / bit work): filter words, k-mers, counts, statistics.

PARITY UNPINNED vs the reference's Rust code (no source / fixtures in /root/reference); the
oracle is the written spec of DESIGN.md section 2.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import random_reads, related_trio
from oracle import orc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "small_trios.json")
MODES = ["direct", "bucketed"]


def dk():
    import denovo_kmer_amd
    return denovo_kmer_amd


def make_engine(mode="direct", **kw):
    return dk().Engine(mode=mode, **kw)


def oracle_trio(parents, child, k, log2_bits, nh, seed, canonical=True, min_count=1):
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(log2_bits)
    ist = orc.bloom_insert(f, log2_bits, nh, seed, k, canonical, pseq, poff)
    km, cn, pst = orc.bloom_probe(f, log2_bits, nh, seed, k, canonical, cseq, coff, min_count)
    return f, ist, km, cn, pst


def assert_family_ran(eng):
    """the kernel family the engine was asked for is the one that produced the last result"""
    names = [n for n, _ in eng.timings()["stages"]]
    if not names:
        return
    if eng.mode == "bucketed":
        assert names[0] == "scan_part", names
        if eng.k >= 12:      # tiny k: a handful of distinct k-mers, bins legitimately overflow and are redone
            assert "overflow_redo" not in names, names
    elif eng.mode == "direct":
        assert names[0] in ("insert_direct", "probe_direct"), names


def gpu_trio(eng, parents, child):
    d = dk()
    ks = d.KmerSet(eng)
    pb = d.ReadBatch.from_sequences(eng, parents)
    ist = ks.insert_reads(pb)
    assert_family_ran(eng)
    cb = d.ReadBatch.from_sequences(eng, child)
    res = d.KmerCounter(eng).child_only(cb, ks)
    assert_family_ran(eng)
    return ks, ist, res


def assert_result_equals(res, km, cn):
    hi, lo, cnt = res.to_host(sort=True)
    assert len(lo) == len(km)
    assert np.array_equal(hi, km["hi"]) and np.array_equal(lo, km["lo"]) and np.array_equal(cnt, cn)


def assert_stats(gst, ost, keys):
    for key in keys:
        assert gst[key] == ost[key], (key, gst, ost)


# ---- packing ---------------------------------------------------------------------------------------

def test_gpu_pack_matches_oracle(rng):
    d = dk()
    reads = random_reads(rng, 300, 0, 260, n_rate=0.02, lower_rate=0.2) + ["", "", "N", "acgtn", "A" * 1000]
    seq, off = orc.concat_reads(reads)
    with d.Engine(k=31) as eng:
        b = d.ReadBatch.from_ascii(eng, seq, off)
        gb, gm, gn = b.download()
        ob, om, on = orc.pack_reads(seq, off)
        assert gn == on and np.array_equal(gb, ob) and np.array_equal(gm, om)
        st = b.stats()
        assert st["n_reads"] == len(reads) and st["n_windows"] == orc.n_windows(off, 31)
        # host-packed upload gives the same batch
        b2 = d.ReadBatch.from_packed(eng, ob, om, on, len(reads), st["n_windows"])
        gb2, gm2, _ = b2.download()
        assert np.array_equal(gb2, ob) and np.array_equal(gm2, om)


# ---- insert / probe parity over k, modes and geometry ---------------------------------------------

@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,log2_bits,nh,canonical", [
    (21, 20, 4, True), (31, 22, 3, True), (32, 20, 2, True), (31, 20, 1, False),
    (1, 20, 2, True), (2, 20, 4, True), (15, 21, 7, True), (27, 24, 16, True),
    (33, 20, 4, True), (51, 21, 3, True), (64, 20, 4, True), (47, 20, 2, False),
])
def test_trio_parity(rng, mode, k, log2_bits, nh, canonical):
    parents, child = related_trio(rng, genome_len=3000, n_reads=80, read_len=130)
    with make_engine(mode, k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=0xABCDEF12345, canonical=canonical) as eng:
        ks, ist, res = gpu_trio(eng, parents, child)
        f, oist, km, cn, pst = oracle_trio(parents, child, k, log2_bits, nh, 0xABCDEF12345, canonical)
        assert np.array_equal(ks.to_host(), f)
        assert_stats(ist, oist, ["n_reads", "n_windows", "n_valid"])
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_reads", "n_windows", "n_valid", "n_absent", "n_distinct"])
        assert res.stats["n_emitted"] == len(km)
        assert ks.popcount() == int(np.bitwise_count(f).sum())


@pytest.mark.parametrize("mode", MODES)
def test_ragged_and_degenerate_reads(rng, mode):
    k = 19
    parents = random_reads(rng, 200, 0, 150, n_rate=0.03, lower_rate=0.3) + ["", "N" * 50, "ACG", "A" * 19]
    child = random_reads(rng, 150, 0, 150, n_rate=0.03, lower_rate=0.3) + parents[:40] + ["", "T" * 19, "acgtnacgt"]
    with make_engine(mode, k=k, filter_log2_bits=20, n_hashes=4, seed=3) as eng:
        ks, ist, res = gpu_trio(eng, parents, child)
        f, oist, km, cn, pst = oracle_trio(parents, child, k, 20, 4, 3)
        assert np.array_equal(ks.to_host(), f)
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])


@pytest.mark.parametrize("mode", MODES)
def test_empty_and_all_invalid_batches(mode):
    d = dk()
    with make_engine(mode, k=31, filter_log2_bits=20) as eng:
        ks = d.KmerSet(eng)
        for reads in ([], [""], ["", "", ""], ["N" * 100], ["ACGT"], ["ACGT" * 7 + "AC"]):   # last: 30 bases < k
            st = ks.insert_sequences(reads)
            assert st["n_valid"] == 0 and st["n_reads"] == len(reads)
            b = d.ReadBatch.from_sequences(eng, reads)
            res = d.KmerCounter(eng).child_only(b, ks)
            assert len(res) == 0 and res.stats["n_absent"] == 0
        assert ks.popcount() == 0


@pytest.mark.parametrize("mode", MODES)
def test_min_count_threshold(rng, mode):
    parents, child = related_trio(rng, genome_len=1500, n_reads=60, read_len=100)
    child = child + child[:20]
    with make_engine(mode, k=25, filter_log2_bits=20, n_hashes=3, seed=11, min_count=2) as eng:
        ks, _, res = gpu_trio(eng, parents, child)
        _, _, km, cn, pst = oracle_trio(parents, child, 25, 20, 3, 11, True, 2)
        assert_result_equals(res, km, cn)
        assert res.stats["n_distinct"] == pst["n_distinct"] > len(km) > 0


@pytest.mark.parametrize("mode", MODES)
def test_loaded_filter_false_positives_and_exact_superset(rng, mode):
    # SURVEY H1 / spec A-6: the GPU set equals the Bloom oracle and is a subset of the exact set
    parents = random_reads(rng, 600, 250, 250)
    child = random_reads(rng, 60, 250, 250) + parents[:10]
    with make_engine(mode, k=25, filter_log2_bits=20, n_hashes=1, seed=424242) as eng:
        ks, _, res = gpu_trio(eng, parents, child)
        f, _, km, cn, pst = oracle_trio(parents, child, 25, 20, 1, 424242)
        assert np.array_equal(ks.to_host(), f)
        assert_result_equals(res, km, cn)
        pseq, poff = orc.concat_reads(parents)
        cseq, coff = orc.concat_reads(child)
        ekm, ecn, _ = orc.exact_child_only(25, True, pseq, poff, cseq, coff)
        hi, lo, cnt = res.to_host()
        got = set(zip(hi.tolist(), lo.tolist()))
        exact = set(zip(ekm["hi"].tolist(), ekm["lo"].tolist()))
        assert got <= exact and 0 < len(got) < len(exact)


# ---- golden vectors ---------------------------------------------------------------------------------

def golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_vectors_on_gpu(case, mode):
    with make_engine(mode, k=case["k"], filter_log2_bits=case["filter_log2_bits"], n_hashes=case["n_hashes"],
                     seed=case["seed"], canonical=case["canonical"], min_count=case["min_count"]) as eng:
        ks, ist, res = gpu_trio(eng, case["parents"], case["child"])
        assert hashlib.sha256(ks.to_host().tobytes()).hexdigest() == case["filter_sha256"]
        assert ist["n_valid"] == case["insert_stats"]["n_valid"]
        hi, lo, cnt = res.to_host()
        got = [[int(a), int(b), int(c)] for a, b, c in zip(hi, lo, cnt)]
        assert got == case["child_only"]
        for key in ("n_windows", "n_valid", "n_absent", "n_distinct"):
            assert res.stats[key] == case["probe_stats"][key]


# ---- KmerCounter / KmerSet API -----------------------------------------------------------------------

@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [17, 31, 45])
def test_kmer_counter_counts_all_kmers(rng, mode, k):
    d = dk()
    reads = random_reads(rng, 120, 20, 160, n_rate=0.01)
    reads = reads + reads[:50] + reads[:10]
    seq, off = orc.concat_reads(reads)
    with make_engine(mode, k=k, filter_log2_bits=20) as eng:
        res = d.KmerCounter(eng).count_sequences(reads)
        km, cn, st = orc.count_reads(k, True, seq, off)
        assert_result_equals(res, km, cn)
        assert res.stats["n_valid"] == st["n_valid"] and res.stats["n_distinct"] == st["n_distinct"]
        as_dict = res.as_dict()
        assert len(as_dict) == len(km) and all(len(s) == k for s in list(as_dict)[:5])


@pytest.mark.parametrize("k", [21, 31, 51])
def test_kmer_set_contains(rng, k):
    d = dk()
    reads = random_reads(rng, 50, 100, 100)
    seq, off = orc.concat_reads(reads)
    km, cn, _ = orc.count_reads(k, True, seq, off)
    other = random_reads(rng, 50, 100, 100)
    oseq, ooff = orc.concat_reads(other)
    okm, _, _ = orc.count_reads(k, True, oseq, ooff)
    with make_engine("direct", k=k, filter_log2_bits=24, n_hashes=4, seed=9) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads)
        assert ks.contains((km["hi"], km["lo"])).all()                    # no false negatives
        f = orc.new_filter(24)
        orc.bloom_insert(f, 24, 4, 9, k, True, seq, off)
        got = ks.contains((okm["hi"], okm["lo"]))
        exp = []
        for a in okm:
            blk, bits = orc.bloom_positions(orc.hash_kmer(int(a["hi"]), int(a["lo"]), k, 9), 24, 4)
            exp.append(all((int(f[8 * blk + (t >> 6)]) >> (t & 63)) & 1 for t in bits))
        assert got.tolist() == exp
        # string front-end
        first = d.kmer_to_str(int(km["hi"][0]), int(km["lo"][0]), k)
        assert ks.contains([first]).tolist() == [True]


def test_set_upload_download_clear_and_idempotent_insert(rng):
    d = dk()
    reads = random_reads(rng, 100, 80, 120)
    with d.Engine(k=31, filter_log2_bits=21, n_hashes=4) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads)
        w1 = ks.to_host()
        ks.insert_sequences(reads)                       # OR is idempotent
        assert np.array_equal(ks.to_host(), w1)
        ks.clear()
        assert ks.popcount() == 0
        ks.from_host(w1)
        assert np.array_equal(ks.to_host(), w1)
        # a set split over two inserts equals one insert of everything (OR is associative)
        ks2 = d.KmerSet(eng)
        ks2.insert_sequences(reads[:37])
        ks2.insert_sequences(reads[37:])
        assert np.array_equal(ks2.to_host(), w1)


def test_or_reduce_slices_and_attached_filter(rng):
    import torch
    d = dk()
    with d.Engine(k=31, filter_log2_bits=20, n_hashes=4) as eng:
        n_words = (1 << 20) // 64
        src = torch.from_numpy(rng.integers(0, 2**63, size=(5, n_words), dtype=np.int64)).cuda()
        dst = torch.from_numpy(rng.integers(0, 2**63, size=n_words, dtype=np.int64)).cuda()
        exp = dst.clone()
        for j in range(5):
            exp |= src[j]
        torch.cuda.synchronize()
        eng.or_reduce_slices(dst.data_ptr(), src.data_ptr(), 5, n_words * 8)
        assert torch.equal(dst, exp)
        # a KmerSet attached to torch-owned memory writes into that tensor
        buf = torch.zeros(n_words, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ks = d.KmerSet(eng, device_ptr=buf.data_ptr(), keepalive=buf)
        reads = random_reads(rng, 50, 100, 100)
        ks.insert_sequences(reads)
        seq, off = orc.concat_reads(reads)
        f = orc.new_filter(20)
        orc.bloom_insert(f, 20, 4, eng.seed, 31, True, seq, off)
        assert np.array_equal(buf.cpu().numpy().view(np.uint64), f)


# ---- synthetic generator ------------------------------------------------------------------------------

@pytest.mark.parametrize("sample", [0, 1, 2])
def test_device_synth_matches_oracle_generator(sample):
    d = dk()
    ocfg = orc.synth_cfg(genome_len=100_000, err_rate=0.02, n_rate=0.003, snv_rate=0.01, denovo_rate=0.001, xover_block=1 << 12)
    gcfg = d.synth_config(genome_len=100_000, err_rate=0.02, n_rate=0.003, snv_rate=0.01, denovo_rate=0.001, xover_log2=12)
    with d.Engine(k=31) as eng:
        b = d.ReadBatch.synth(eng, gcfg, sample, 1000, 777)
        gb, gm, gn = b.download()
        seq, off = orc.synth_reads(ocfg, sample, 1000, 777)
        ob, om, on = orc.pack_reads(seq, off)
        assert gn == on and np.array_equal(gm, om) and np.array_equal(gb, ob)
        assert b.stats()["n_windows"] == 777 * 120


# ---- larger seeded runs (oracle finishes in seconds) ---------------------------------------------------

@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,log2_bits", [(21, 24), (31, 26)])
def test_synthetic_trio_parity_config0_scale(mode, k, log2_bits):
    # BASELINE.json configs[0] shape: 10k synthetic 150 bp reads per trio member
    d = dk()
    n_reads = 10_000
    ocfg = orc.synth_cfg(genome_len=50_000)
    gcfg = d.synth_config(genome_len=50_000)
    with make_engine(mode, k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=20260313) as eng:
        ks = d.KmerSet(eng)
        f = orc.new_filter(log2_bits)
        for s in (0, 1):
            ist = ks.insert_reads(d.ReadBatch.synth(eng, gcfg, s, 0, n_reads))
            seq, off = orc.synth_reads(ocfg, s, 0, n_reads)
            oist = orc.bloom_insert(f, log2_bits, 4, 20260313, k, True, seq, off)
            assert ist["n_valid"] == oist["n_valid"] and ist["n_windows"] == oist["n_windows"]
        assert np.array_equal(ks.to_host(), f)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, n_reads), ks)
        cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
        km, cn, pst = orc.bloom_probe(f, log2_bits, 4, 20260313, k, True, cseq, coff)
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])
        assert eng.timings()["total_ms"] > 0


@pytest.mark.parametrize("mode,k", [("direct", 51), ("bucketed", 51), ("direct", 31), ("bucketed", 31),
                                    ("direct", 64), ("bucketed", 64), ("bucketed", 33)])
def test_long_reads_config4_shape(mode, k):
    # BASELINE.json configs[4] shape: ONT-style 10 kb reads with 5 % errors, both kernel families
    # (k > 32 uses 128-bit k-mers and 16-byte bucket records)
    d = dk()
    n_reads, L, log2_bits = 300, 10_000, 26
    ocfg = orc.synth_cfg(genome_len=400_000, read_len=L, err_rate=0.05)
    gcfg = d.synth_config(genome_len=400_000, read_len=L, err_rate=0.05)
    with make_engine(mode, k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=51) as eng:
        ks = d.KmerSet(eng)
        f = orc.new_filter(log2_bits)
        for s in (0, 1):
            ist = ks.insert_reads(d.ReadBatch.synth(eng, gcfg, s, 0, n_reads))
            assert_family_ran(eng)
            seq, off = orc.synth_reads(ocfg, s, 0, n_reads)
            oist = orc.bloom_insert(f, log2_bits, 4, 51, k, True, seq, off)
            assert ist["n_valid"] == oist["n_valid"] and ist["n_windows"] == n_reads * (L - k + 1)
        assert np.array_equal(ks.to_host(), f)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, n_reads), ks)
        assert_family_ran(eng)
        cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
        km, cn, pst = orc.bloom_probe(f, log2_bits, 4, 51, k, True, cseq, coff)
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])


@pytest.mark.parametrize("log2_bits", [29, 31])
def test_two_level_partition_large_filter(log2_bits):
    # filters above 2^9 segments take the two-level multisplit (scan_part + repart)
    d = dk()
    n_reads, k = 30_000, 31
    ocfg = orc.synth_cfg(genome_len=300_000)
    gcfg = d.synth_config(genome_len=300_000)
    with make_engine("bucketed", k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=77) as eng:
        ks = d.KmerSet(eng)
        f = orc.new_filter(log2_bits)
        for s in (0, 1):
            ist = ks.insert_reads(d.ReadBatch.synth(eng, gcfg, s, 0, n_reads))
            seq, off = orc.synth_reads(ocfg, s, 0, n_reads)
            oist = orc.bloom_insert(f, log2_bits, 4, 77, k, True, seq, off)
            assert ist["n_valid"] == oist["n_valid"]
        assert [n for n, _ in eng.timings()["stages"]] == ["scan_part", "repart", "seg_insert"]
        assert np.array_equal(ks.to_host(), f)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, n_reads), ks)
        assert [n for n, _ in eng.timings()["stages"]] == ["scan_part", "repart", "seg_probe", "seg_count"]
        cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
        km, cn, pst = orc.bloom_probe(f, log2_bits, 4, 77, k, True, cseq, coff)
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])


def test_heavy_hitter_overflow_is_handled_exactly():
    # one k-mer repeated far beyond its segment's capacity overflows its region; the overflow records
    # are OR-ed / probed one by one and counted with their segment, with identical results and
    # without redoing the batch on the direct family
    d = dk()
    k = 21
    reads = ["A" * 150] * 4000 + ["ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGATTAGC" * 3] * 10
    seq, off = orc.concat_reads(reads)
    with make_engine("bucketed", k=k, filter_log2_bits=24, n_hashes=4, seed=5) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads[:2000])
        names = [n for n, _ in eng.timings()["stages"]]
        assert "ovf_insert" in names and "overflow_redo" not in names
        f = orc.new_filter(24)
        s2, o2 = orc.concat_reads(reads[:2000])
        orc.bloom_insert(f, 24, 4, 5, k, True, s2, o2)
        assert np.array_equal(ks.to_host(), f)
        # KmerCounter over everything: 520 000 copies of poly-A land in one segment
        res = d.KmerCounter(eng).count_sequences(reads)
        names = [n for n, _ in eng.timings()["stages"]]
        assert "ovf_probe" in names and "overflow_redo" not in names
        km, cn, st = orc.count_reads(k, True, seq, off)
        assert_result_equals(res, km, cn)
        assert int(cn.max()) == 4000 * 130
        assert res.stats["n_absent"] == st["n_valid"] and res.stats["n_distinct"] == st["n_distinct"]
        # membership pass: the heavy hitter is present in the filter, a second one is not
        child = ["A" * 150] * 3000 + ["C" * 150] * 3000 + reads[-10:]
        cseq, coff = orc.concat_reads(child)
        res2 = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        names = [n for n, _ in eng.timings()["stages"]]
        assert "ovf_probe" in names and "overflow_redo" not in names
        km2, cn2, st2 = orc.bloom_probe(f, 24, 4, 5, k, True, cseq, coff)
        assert_result_equals(res2, km2, cn2)
        assert res2.stats["n_absent"] == st2["n_absent"] and int(cn2.max()) == 3000 * 130


def test_min_count_with_overflow_records():
    d = dk()
    k = 25
    child = ["G" * 120] * 2500 + ["ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGATTAGC" * 3] * 2 + ["TTGACCATGCAATGCATGCCGGATAGCTAGCATCG"]
    seq, off = orc.concat_reads(child)
    with make_engine("bucketed", k=k, filter_log2_bits=23, n_hashes=3, seed=8, min_count=2) as eng:
        ks = d.KmerSet(eng)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        f = orc.new_filter(23)
        km, cn, st = orc.bloom_probe(f, 23, 3, 8, k, True, seq, off, min_count=2)
        assert_result_equals(res, km, cn)
        assert res.stats["n_distinct"] == st["n_distinct"] > len(km)


@pytest.mark.parametrize("mode", MODES)
def test_result_device_view_is_dense(mode):
    # the bucketed count kernel fills several output regions; a device view must be one dense array
    d = dk()
    gcfg = d.synth_config(genome_len=150_000)
    with make_engine(mode, k=31, filter_log2_bits=27, n_hashes=4) as eng:
        ks = d.KmerSet(eng)
        ks.insert_reads(d.ReadBatch.synth(eng, gcfg, 0, 0, 15_000))
        res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, 15_000), ks)
        before = res.to_host()
        plo, phi, pcnt, n = res.device_view()
        assert n == len(before[1]) == res.stats["n_emitted"] and plo and pcnt
        res._host = None
        after = res.to_host()
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
        import torch
        buf = torch.empty(n, dtype=torch.int64, device="cuda")
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(plo), ctypes.c_size_t(n * 8), 3) == 0
        assert np.array_equal(np.sort(buf.cpu().numpy().view(np.uint64)), before[1])


@pytest.mark.parametrize("mode", MODES)
def test_repeat_runs_are_bitwise_identical(mode):
    # atomics commute (OR / integer add): the result must not depend on scheduling
    d = dk()
    gcfg = d.synth_config(genome_len=200_000)
    with make_engine(mode, k=31, filter_log2_bits=25, n_hashes=4) as eng:
        outs = []
        for _ in range(3):
            ks = d.KmerSet(eng)
            ks.insert_reads(d.ReadBatch.synth(eng, gcfg, 0, 0, 20_000))
            ks.insert_reads(d.ReadBatch.synth(eng, gcfg, 1, 0, 20_000))
            res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, 20_000), ks)
            hi, lo, cnt = res.to_host()
            outs.append((hashlib.sha256(ks.to_host().tobytes()).hexdigest(),
                         hashlib.sha256(lo.tobytes() + cnt.tobytes()).hexdigest()))
            ks.close()
        assert outs[0] == outs[1] == outs[2]


# ---- BASELINE.json configs[1] at FULL size: size-independent properties ---------------------------------

def _result_checksum(res):
    """order-independent digest of a (k-mer, count) table: wrapped sums of mixed keys"""
    hi, lo, cnt = res.to_host(sort=False)
    with np.errstate(over="ignore"):
        mixed = (lo ^ (lo >> np.uint64(29))) * np.uint64(0x9E3779B97F4A7C15)
        return (int(len(lo)), int(cnt.astype(np.uint64).sum()),
                int(mixed.sum(dtype=np.uint64)), int((mixed * cnt.astype(np.uint64)).sum(dtype=np.uint64)))


@pytest.mark.timeout(900)
def test_full_size_configs1_properties():
    """k=31, 12.8 M x 150 bp reads per sample, 2^34-bit filter (the bench workload).  The oracle cannot
    run this size in seconds, so the checks are properties: no false negatives, idempotence,
    union-of-shards == whole, and agreement of the two independent kernel families."""
    d = dk()
    n_reads, k, log2_bits = 12_800_000, 31, 34
    gcfg = d.synth_config(genome_len=64 << 20)
    with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=20260313, mode="bucketed") as eb:
        ks = d.KmerSet(eb)
        p0 = d.ReadBatch.synth(eb, gcfg, 0, 0, n_reads)
        st0 = ks.insert_reads(p0)
        assert [n for n, _ in eb.timings()["stages"]] == ["scan_part", "repart", "seg_insert"]
        assert st0["n_windows"] == n_reads * 120 and 0.99 * st0["n_windows"] < st0["n_valid"] < st0["n_windows"]
        pop0 = ks.popcount()
        # idempotence of OR: inserting the same batch again changes nothing
        ks.insert_reads(p0)
        assert ks.popcount() == pop0
        # union of shards == whole: the same parent inserted as two half batches into a second filter
        ks2 = d.KmerSet(eb)
        ks2.insert_reads(d.ReadBatch.synth(eb, gcfg, 0, 0, n_reads // 2))
        ks2.insert_reads(d.ReadBatch.synth(eb, gcfg, 0, n_reads // 2, n_reads - n_reads // 2))
        assert ks2.popcount() == pop0
        a, b = ks.to_host(), ks2.to_host()
        assert np.array_equal(a, b)
        del a, b
        ks2.close()
        # no false negatives: every k-mer of the inserted reads is present
        self_probe = d.KmerCounter(eb).child_only(p0, ks)
        assert self_probe.stats["n_absent"] == 0 and len(self_probe) == 0
        assert self_probe.stats["n_valid"] == st0["n_valid"]
        p0.close()
        ks.insert_reads(d.ReadBatch.synth(eb, gcfg, 1, 0, n_reads))
        pop = ks.popcount()
        assert pop0 < pop < 2 * pop0
        child = d.ReadBatch.synth(eb, gcfg, 2, 0, n_reads)
        rb = d.KmerCounter(eb).child_only(child, ks)
        assert [n for n, _ in eb.timings()["stages"]] == ["scan_part", "repart", "seg_probe", "seg_count"]
        sb, cb = rb.stats, _result_checksum(rb)
        assert cb[0] == sb["n_distinct"] == sb["n_emitted"] and cb[1] == sb["n_absent"]
        assert 0.05 * sb["n_valid"] < sb["n_absent"] < 0.3 * sb["n_valid"]     # error k-mers of the child
        rb.close()
        filt = ks.to_host()
        child.close()
        ks.close()
    # the direct family, an independent implementation, must give the same set and counts
    with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=20260313, mode="direct") as ed:
        kd = d.KmerSet(ed)
        kd.from_host(filt)
        del filt
        rd = d.KmerCounter(ed).child_only(d.ReadBatch.synth(ed, gcfg, 2, 0, n_reads), kd)
        assert [n for n, _ in ed.timings()["stages"]][0] == "probe_direct"
        for key in ("n_windows", "n_valid", "n_absent", "n_distinct", "n_emitted"):
            assert rd.stats[key] == sb[key], key
        assert _result_checksum(rd) == cb


# ---- multi-batch samples: device-side merge of per-batch tables ----------------------------------------

@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,merge_options", [(31, {}), (45, {}), (31, {"merge_pass_bits": 3}), (45, {"merge_pass_bits": 2, "merge_idx64": 1}),
                                             (31, {"merge_idx64": 1})])
def test_result_merge_equals_whole_sample(rng, mode, k, merge_options):
    """merge_pass_bits / merge_idx64 force the hash-range passes and the 64-bit candidate indices that tables of
    2^32 entries and more take"""
    d = dk()
    parents, child = related_trio(rng, genome_len=4000, n_reads=150, read_len=120)
    child = child + child[:40] + child[10:30]
    with make_engine(mode, k=k, filter_log2_bits=22, n_hashes=4, seed=77, min_count=1) as eng:
        for name, value in merge_options.items():
            eng.set_option(name, value)
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        kc = d.KmerCounter(eng)
        parts = [kc.child_only(d.ReadBatch.from_sequences(eng, child[a:b]), ks)
                 for a, b in ((0, 70), (70, 71), (71, 160), (160, len(child)))]
        pseq, poff = orc.concat_reads(parents)
        cseq, coff = orc.concat_reads(child)
        f = orc.new_filter(22)
        orc.bloom_insert(f, 22, 4, 77, k, True, pseq, poff)
        for mc in (1, 2, 3):
            merged = kc.merge(parts, min_count=mc)
            km, cn, st = orc.bloom_probe(f, 22, 4, 77, k, True, cseq, coff, min_count=mc)
            assert_result_equals(merged, km, cn)
            assert merged.stats["n_distinct"] == st["n_distinct"] and merged.stats["n_emitted"] == len(km)
        # merging nothing / a single table is the identity
        assert len(kc.merge([], min_count=1)) == 0
        one = kc.merge(parts[:1], min_count=1)
        a, b = one.to_host(), parts[0].to_host()
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("k", [31, 45])
@pytest.mark.parametrize("pass_bits,undersize", [(0, 0), (2, 0), (5, 0), (8, 0), (3, 6), (0, 2)])
def test_merging_the_tables_of_one_hash_window(k, pass_bits, undersize):
    """the tables an accumulator returns for ONE hash window share their top hash bits: merge passes and table slots come
    from a remix of the hash, so every pass takes its share (round 2's merge put such inputs into one pass, whose table
    then overflowed and the insert kernel span forever).  merge_undersize starts with tables too small: the bounded
    probe reports it and the merge is redone with larger ones ("merge_redo")."""
    d = dk()
    rng = np.random.default_rng(31)
    reads = random_reads(rng, 3000, 150, 151)
    reads = reads + reads[:700]
    seq, off = orc.concat_reads(reads)
    km, cn, ost = orc.count_reads(k, True, seq, off)
    hashes = np.array([orc.hash_kmer(int(h), int(l), k, 3) for h, l in zip(km["hi"], km["lo"])], dtype=np.uint64)
    with d.Engine(k=k, filter_log2_bits=24, seed=3, mode="bucketed") as eng:
        eng.set_option("merge_pass_bits", pass_bits)
        eng.set_option("merge_undersize", undersize)
        acc = d.ChildAccumulator(eng, None, capacity_records=600_000, window_count=8)
        total = 0
        for w in (0, 5):
            # the window's k-mers arrive as three tables (three finishes of three partial accumulations)
            parts = []
            for a, b in ((0, 1500), (1500, 1501), (1501, len(reads))):
                acc.reset(w)
                acc.add(d.ReadBatch.from_sequences(eng, reads[a:b]))
                parts.append(acc.finish(min_count=1))
            for mc in (1, 2):
                merged = d.KmerCounter(eng).merge(parts, min_count=mc)
                names = [n for n, _ in eng.timings()["stages"]]
                assert ("merge_redo" in names) == bool(undersize), names
                hi, lo, cnt = merged.to_host(sort=True)
                # the oracle's table, restricted to the window: top 3 bits of the k-mer's hash
                sel = ((hashes >> np.uint64(61)) == np.uint64(w)) & (cn >= mc)
                assert np.array_equal(lo, km["lo"][sel]) and np.array_equal(hi, km["hi"][sel]) and np.array_equal(cnt, cn[sel])
                if mc == 1:
                    total += len(lo)
                merged.close()
            for p in parts:
                p.close()
        assert total > 0
        acc.close()


def test_filter_save_and_load(tmp_path, rng):
    d = dk()
    reads = random_reads(rng, 200, 80, 140)
    path = tmp_path / "parents.dkbloom"
    with d.Engine(k=31, filter_log2_bits=23, n_hashes=4, seed=99) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads)
        words = ks.to_host()
        ks.save(path)
        assert path.stat().st_size == 64 + (1 << 23) // 8
        ks2 = d.KmerSet(eng)
        ks2.load(path)
        assert np.array_equal(ks2.to_host(), words)
    with d.Engine(k=31, filter_log2_bits=23, n_hashes=3, seed=99) as other:       # other geometry: refused
        ks3 = d.KmerSet(other)
        with pytest.raises(d.DkError) as ei:
            ks3.load(path)
        assert "geometry" in str(ei.value) and ks3.popcount() == 0
        with pytest.raises(d.DkError):
            ks3.load(tmp_path / "missing.dkbloom")


# ---- kernel geometries picked by size: every seg_count geometry, every scan_part tile shape -----------

@pytest.mark.parametrize("n_reads,k", [(30, 31), (140, 31), (330, 31), (700, 31), (20, 51), (50, 51), (90, 51), (200, 51), (420, 51)])
def test_every_seg_count_geometry_counts_exactly(rng, n_reads, k):
    # KmerCounter over a 2^22-bit geometry (8 segments): n_reads * ~120 records / 8 segments per segment
    # walks the <128>, <256>, <512> and <1024>-thread count kernels (thresholds in bucketed_probe_t)
    d = dk()
    reads = random_reads(rng, n_reads, 150, 151)
    reads = reads + reads[: n_reads // 3]                    # repeats: counts above 1
    seq, off = orc.concat_reads(reads)
    with make_engine("bucketed", k=k, filter_log2_bits=22, seed=99) as eng:
        res = d.KmerCounter(eng).count_sequences(reads)
        assert_family_ran(eng)
        km, cn, st = orc.count_reads(k, True, seq, off)
        assert_result_equals(res, km, cn)
        assert res.stats["n_distinct"] == st["n_distinct"] and int(cn.max()) >= 2


@pytest.mark.parametrize("k", [31, 51])
@pytest.mark.parametrize("count_seg", [300, 2500, 6000, 12000])
def test_seg_count_geometries_by_unit_size(k, count_seg):
    """count_seg sets the records per counting unit of a KmerCounter batch: a few hundred (128 threads), thousands (256 / 512)
    and 12 K (the 1024-thread kernel, units held in registers up to 16 K records; k > 32: two chunks per unit)"""
    d = dk()
    rng = np.random.default_rng(2024)
    reads = random_reads(rng, 1500, 150, 151)
    reads = reads + reads[:500]
    seq, off = orc.concat_reads(reads)
    km, cn, st = orc.count_reads(k, True, seq, off)
    with make_engine("bucketed", k=k, filter_log2_bits=22, seed=99) as eng:
        eng.set_option("count_seg", count_seg)
        res = d.KmerCounter(eng).count_sequences(reads)
        assert_family_ran(eng)
        assert_result_equals(res, km, cn)
        assert res.stats["n_distinct"] == st["n_distinct"] and int(cn.max()) >= 2


def _forced_geometry_trio(options, expect_stages):
    """small related trio through the bucketed family with kernel geometries forced by engine options
    (dk_engine_set_option test hooks), checked against the oracle"""
    d = dk()
    rng = np.random.default_rng(4242)
    parents, child = related_trio(rng, genome_len=30000, n_reads=900, read_len=140)
    for k in (31, 45):
        f, ist, km, cn, pst = oracle_trio(parents, child, k, 24, 4, 7)
        with d.Engine(k=k, filter_log2_bits=24, n_hashes=4, seed=7, mode="bucketed") as eng:
            for name, value in options.items():
                eng.set_option(name, value)
            ks = d.KmerSet(eng)
            ks.insert_sequences(parents)
            assert np.array_equal(ks.to_host(), f)
            res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
            names = [n for n, _ in eng.timings()["stages"]]
            assert names[:len(expect_stages)] == expect_stages and "overflow_redo" not in names, names
            assert_result_equals(res, km, cn)
            assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])


@pytest.mark.parametrize("scan_variant,repart_variant", [(6, 0), (1, 0), (3, 1), (4, 0), (5, 0), (2, 0)])
def test_scan_part_tile_shapes_agree_with_the_oracle(scan_variant, repart_variant):
    """The scan_part geometry is chosen from the number of segments (2 below 2^16 segments, 6 from there on);
    the engine options force each compiled shape on a small input so that all of them are checked"""
    _forced_geometry_trio({"scan_variant": scan_variant, "repart_variant": repart_variant}, ["scan_part", "repart"])


@pytest.mark.parametrize("sub_split", [1, 2, 3])
def test_sub_segment_split_of_the_insert_kernels(sub_split):
    """sub_split forces what 2^38 / 2^39-bit sets do by themselves: the partition stops 1-3 bits short of the 64-KiB
    segments and 2-8 workgroups share a region, each taking the records of its own segment"""
    _forced_geometry_trio({"sub_split": sub_split}, ["scan_part", "repart"])
    d = dk()
    rng = np.random.default_rng(77)
    parents, child = related_trio(rng, genome_len=20000, n_reads=500, read_len=120)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    km, cn, ost = orc.exact_child_only(45, True, pseq, poff, cseq, coff, 1)
    with d.Engine(k=45, filter_log2_bits=26, seed=7, mode="bucketed", set_kind="exact") as eng:
        eng.set_option("sub_split", sub_split)
        ks, _, res = gpu_trio(eng, parents, child)
        assert_result_equals(res, km, cn)


def test_repart_in_plain_block_order_agrees():
    """repart deals its tiles one bin per XCD by default; repart_plain restores plain block order (A/B runs)"""
    _forced_geometry_trio({"repart_plain": 1}, ["scan_part", "repart"])
    _forced_geometry_trio({"repart_plain": 1, "force_l3": 1}, ["scan_part", "repart", "repart3"])


@pytest.mark.parametrize("repart_pieces", [1, 2])
def test_repart_over_pieces_and_over_their_concatenation_agree(repart_pieces):
    """repart reads a level-1 bin piece by piece (1) or as the concatenation of its pieces (2: full tiles; what the
    whole-genome child step takes by itself); small batches leave pieces of a few records, many per tile -- with and
    without a third level and a forced scan shape of many small workgroups (hundreds of pieces per bin)"""
    _forced_geometry_trio({"repart_pieces": repart_pieces}, ["scan_part", "repart"])
    _forced_geometry_trio({"repart_pieces": repart_pieces, "force_l3": 1}, ["scan_part", "repart", "repart3"])
    _forced_geometry_trio({"repart_pieces": repart_pieces, "scan_variant": 5}, ["scan_part", "repart"])
    # the level-1 pieces workgroup-major instead of bin-major (option "l1_layout"): the same records at other addresses
    _forced_geometry_trio({"repart_pieces": repart_pieces, "l1_layout": 1}, ["scan_part", "repart"])
    _forced_geometry_trio({"repart_pieces": repart_pieces, "l1_layout": 1, "scan_variant": 5}, ["scan_part", "repart"])
    # other distances between the pieces of consecutive bins (option "l1_skew", bytes)
    _forced_geometry_trio({"repart_pieces": repart_pieces, "l1_skew": 4224}, ["scan_part", "repart"])
    _forced_geometry_trio({"repart_pieces": repart_pieces, "l1_skew": 16, "l1_layout": 1}, ["scan_part", "repart"])


def test_forced_scan_shape_with_fewer_threads_than_bins():
    """variants 4 / 5 have 256 / 128 threads; with more level-1 bins than threads the plan moves bits to level 2
    (2^33 bits = 2^14 segments: b1_up pushes the level-1 split to 9 bits = 512 bins)"""
    d = dk()
    rng = np.random.default_rng(7)
    parents, child = related_trio(rng, genome_len=20000, n_reads=500, read_len=120)
    f, ist, km, cn, pst = oracle_trio(parents, child, 31, 33, 4, 7)
    for variant in (4, 5):
        with d.Engine(k=31, filter_log2_bits=33, n_hashes=4, seed=7, mode="bucketed") as eng:
            eng.set_option("scan_variant", variant)
            eng.set_option("b1_up", 4)
            ks, _, res = gpu_trio(eng, parents, child)
            assert ks.popcount() == int(np.bitwise_count(f).sum())
            assert_result_equals(res, km, cn)
            ks.close()


def test_engine_options_are_validated():
    d = dk()
    with d.Engine(k=31, filter_log2_bits=24) as eng:
        for name, value in (("scan_variant", 7), ("scan_variant", -1), ("no_such_option", 1), ("force_l3", 2), ("count_seg", 5)):
            with pytest.raises(d.DkError) as ei:
                eng.set_option(name, value)
            assert ei.value.status == 1
        eng.set_option("multiplicity_hint", 2)
        eng.set_option("scan_variant", 0)


# ---- kmer.rs stand-in: per-position canonical k-mers and hashes (dk_reads_kmers) -----------------------

def _expected_kmers(reads, k, canonical, seed):
    n_bases = sum(len(r) for r in reads) + len(reads)
    lo = np.zeros(n_bases, dtype=np.uint64)
    hi = np.zeros(n_bases, dtype=np.uint64)
    hs = np.zeros(n_bases, dtype=np.uint64)
    notk = np.ones(n_bases, dtype=bool)
    p0 = 0
    for r in reads:
        km, valid = orc.read_kmers(r, k, canonical)
        for j in np.flatnonzero(valid):
            lo[p0 + j], hi[p0 + j] = km["lo"][j], km["hi"][j]
            hs[p0 + j] = orc.hash_kmer(int(km["hi"][j]), int(km["lo"][j]), k, seed)
            notk[p0 + j] = False
        p0 += len(r) + 1
    pad = (-n_bases) % 64
    bits = np.concatenate([notk, np.ones(pad, dtype=bool)]).astype(np.uint8)
    words = np.packbits(bits).view(">u8").astype(np.uint64)      # MSB-first 64-bit words
    return lo, hi, hs, words, int((~notk).sum())


@pytest.mark.parametrize("k,canonical", [(21, True), (31, True), (32, False), (1, True), (33, True), (51, True), (64, False)])
def test_per_position_kmers_match_the_oracle(rng, k, canonical):
    d = dk()
    import torch
    torch.cuda.init()            # torch's HIP runtime must come up before the engine's in a shared process
    reads = random_reads(rng, 60, 0, 180, n_rate=0.02, lower_rate=0.2) + ["", "N", "ACGT" * 20, "acgtn" * 15]
    seed = 0xC0FFEE12345
    lo, hi, hs, words, n_valid = _expected_kmers(reads, k, canonical, seed)
    with d.Engine(k=k, canonical=canonical, seed=seed) as eng:
        b = d.ReadBatch.from_sequences(eng, reads)
        out = b.kmers(hashes=True)
        assert np.array_equal(out["lo"], lo) and np.array_equal(out["hash"], hs)
        if k > 32:
            assert np.array_equal(out["hi"], hi)
        assert np.array_equal(out["not_kmer"], words)
        assert out["stats"]["n_valid"] == n_valid and out["stats"]["n_windows"] == b.stats()["n_windows"]
        # device pointers are written in place (here: a torch tensor), optional outputs may be left out
        n = len(lo)
        t_lo = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        t_hi = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        st = b.kmers(into={"lo": t_lo.data_ptr(), "hi": t_hi.data_ptr() if k > 32 else 0})["stats"]
        assert st["n_valid"] == n_valid
        assert np.array_equal(t_lo.cpu().numpy().view(np.uint64), lo)
        if k > 32:
            assert np.array_equal(t_hi.cpu().numpy().view(np.uint64), hi)
        # empty batch
        e = d.ReadBatch.from_sequences(eng, [])
        assert len(e.kmers()["lo"]) == 0


# ---- three partition levels: filters above 2^37 bits (2^19 segments and more) --------------------------

def test_three_level_partition_forced_on_small_inputs():
    """force_l3 sends every geometry with at least 8 segments through scan_part -> repart -> repart3 (the path
    that 2^38-bit and larger filters take), so the oracle can check it at small sizes"""
    _forced_geometry_trio({"force_l3": 1}, ["scan_part", "repart", "repart3"])


@pytest.mark.parametrize("set_kind,log2_bits", [("bloom", 38), ("exact", 38), ("bloom", 40)])
def test_partition_levels_of_very_large_sets(set_kind, log2_bits):
    """32-GiB and 128-GiB parent sets.  2^38 bits = 2^19 segments: two partition levels (512 x 1024 bins).  2^40 bits =
    2^21 segments: the insert takes two levels (1024 x 1024) and the sub-segment split, the per-batch probe three levels.
    The direct family, an independent implementation working on the same geometry, must agree on the set size and on
    every child-only k-mer and count (the oracle cannot hold such a filter in this test's time)."""
    d = dk()
    n_reads, k = 1_500_000, 31
    gcfg = d.synth_config(genome_len=8 << 20)
    levels = ["scan_part", "repart"] + (["repart3"] if log2_bits >= 40 else [])
    levels_insert = ["scan_part", "repart", "seg_exact_insert" if set_kind == "exact" else "seg_insert"]
    out = {}
    for mode in ("bucketed", "direct"):
        with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=20260313, mode=mode, set_kind=set_kind) as eng:
            ks = d.KmerSet(eng)
            for smp in (0, 1):
                ks.insert_reads(d.ReadBatch.synth(eng, gcfg, smp, 0, n_reads))
            names = [n for n, _ in eng.timings()["stages"]]
            if mode == "bucketed":
                assert names[:3] == levels_insert, names
            else:
                assert names == ["insert_direct"], names
            pop = ks.popcount()
            res = d.KmerCounter(eng).child_only(d.ReadBatch.synth(eng, gcfg, 2, 0, n_reads), ks)
            names = [n for n, _ in eng.timings()["stages"]]
            assert (names[:len(levels)] == levels) if mode == "bucketed" else names[0] == "probe_direct", names
            out[mode] = (pop, {key: res.stats[key] for key in ("n_windows", "n_valid", "n_absent", "n_distinct", "n_emitted")},
                         _result_checksum(res))
            res.close()
            ks.close()
    assert out["bucketed"] == out["direct"]
    assert 0.02 * out["direct"][1]["n_valid"] < out["direct"][1]["n_absent"] < 0.4 * out["direct"][1]["n_valid"]


@pytest.mark.parametrize("k", [31, 45])
def test_many_absent_records_per_segment_are_split_before_counting(rng, k):
    # 2 filter segments, ~70 K absent records: far more per segment than a count workgroup keeps in registers,
    # so the absent lists are split by further hash bits first (stage "count_split"); results must not change
    d = dk()
    parents = random_reads(rng, 20, 150, 151)
    child = random_reads(rng, 600, 150, 151)
    child = child + child[:150] + parents[:10]
    with make_engine("bucketed", k=k, filter_log2_bits=20, n_hashes=3, seed=12) as eng:
        ks, ist, res = gpu_trio(eng, parents, child)
        names = [n for n, _ in eng.timings()["stages"]]
        assert "count_split" in names and "overflow_redo" not in names, names
        f, oist, km, cn, pst = oracle_trio(parents, child, k, 20, 3, 12)
        assert np.array_equal(ks.to_host(), f)
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])
        assert int(cn.max()) >= 2


@pytest.mark.parametrize("k", [31, 45])
def test_high_absent_rate_sinks_into_finer_units_while_probing(rng, k):
    # 256 filter segments, ~4.5 M absent records (a child unrelated to its parents): the membership kernel samples the
    # absent rate on the first segments and then appends the absent records straight to finer counting units, so the
    # separate split of the absent lists (stage "count_split") is not needed; "sink_plain" brings it back.  Same results.
    d = dk()
    parents = random_reads(rng, 200, 150, 151)
    child = random_reads(rng, 40000, 150, 151)
    child = child + child[:3000] + parents[:50]
    f, oist, km, cn, pst = oracle_trio(parents, child, k, 27, 4, 99)
    got = {}
    for plain in (0, 1):
        with make_engine("bucketed", k=k, filter_log2_bits=27, n_hashes=4, seed=99) as eng:
            eng.set_option("sink_plain", plain)
            ks, ist, res = gpu_trio(eng, parents, child)
            names = [n for n, _ in eng.timings()["stages"]]
            assert ("count_split" in names) == bool(plain) and "overflow_redo" not in names, names
            if not plain:
                assert np.array_equal(ks.to_host(), f)
            assert_result_equals(res, km, cn)
            assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])
            got[plain] = res.stats["n_emitted"]
    assert got[0] == got[1] and int(cn.max()) >= 2


def test_sunk_unit_running_full_falls_back_to_the_plain_probe(rng):
    # as above, plus one read 4000 times over: its 120 k-mers (absent from the parents) each add 4000 records to one of the
    # finer units, more than the unit has room for -- the membership kernel reports the overflow and the batch is probed
    # again the plain way (whole-segment absent lists, split before counting where they fit); results equal the oracle's
    d = dk()
    k = 31
    parents = random_reads(rng, 200, 150, 151)
    child = random_reads(rng, 40000, 150, 151)
    heavy = random_reads(rng, 1, 150, 151)
    child = child + heavy * 4000 + parents[:50]
    f, oist, km, cn, pst = oracle_trio(parents, child, k, 27, 4, 5)
    with make_engine("bucketed", k=k, filter_log2_bits=27, n_hashes=4, seed=5) as eng:
        ks, ist, res = gpu_trio(eng, parents, child)
        names = [n for n, _ in eng.timings()["stages"]]
        # (marks of one name are summed into one stage: the abandoned attempt is renamed)
        assert "seg_probe_sunk" in names and "seg_probe" in names and "overflow_redo" not in names, names
        assert_result_equals(res, km, cn)
        assert_stats(res.stats, pst, ["n_windows", "n_valid", "n_absent", "n_distinct"])
        assert int(cn.max()) >= 4000


# ---- window-major scan: batches whose reads all have one length ---------------------------------------------

@pytest.mark.parametrize("k", [1, 5, 16, 21, 31, 32])
@pytest.mark.parametrize("read_len", [32, 47, 150, 151, 1000])
def test_uniform_read_lengths_take_the_window_major_scan_and_agree(k, read_len):
    """reads of ONE length are dealt to the scan's threads by windows, not by positions (dk_bucket_scan.h: WindowMajor);
    every k / read-length pairing -- parts that end exactly at the read's last window, parts that would run into the
    next read (k = 1), more threads per read than k (1000-bp reads) -- must count what the oracle counts, with the option
    off (position-major) and on, through ASCII input, packed upload and an attached device buffer"""
    d = dk()
    if read_len < k:
        pytest.skip("no windows")
    rng = np.random.default_rng(1000 * k + read_len)
    reads = random_reads(rng, max(40, 40000 // read_len), read_len, read_len, n_rate=0.004)
    reads = reads + reads[:7]
    seq, off = orc.concat_reads(reads)
    km, cn, st = orc.count_reads(k, True, seq, off)
    bases, mask, n_bases = orc.pack_reads(seq, off)
    for positions in (0, 1):
        with make_engine("bucketed", k=k, filter_log2_bits=22, seed=99) as eng:
            eng.set_option("scan_positions", positions)
            batches = [d.ReadBatch.from_sequences(eng, reads),
                       d.ReadBatch.from_packed(eng, bases, mask, n_bases, len(reads), orc.n_windows(off, k))]
            for b in batches:
                res = d.KmerCounter(eng).count_reads(b)
                assert_family_ran(eng)
                assert_result_equals(res, km, cn)
                assert res.stats["n_valid"] == st["n_valid"]
                res.close()


def test_streams_that_only_look_uniform_are_counted_exactly():
    """n_bases divides by n_reads but the reads differ in length: the device check clears the batch's flag and the scan
    stays position-major; and a stream whose flags at the positions = L mod L + 1 are all set although the reads are NOT
    of one length (an N sits where a separator would be, the real separators are elsewhere) may take the window-major
    mapping -- it depends on the flags only -- and still counts exactly"""
    d = dk()
    rng = np.random.default_rng(77)
    k = 21
    # (a) 60 reads of 100 and 60 of 200 bases: (100 + 1) * 60 + (200 + 1) * 60 = 120 * 151
    a = random_reads(rng, 60, 100, 100) + random_reads(rng, 60, 200, 200)
    rng.shuffle(a)
    # (b) as if 150-bp reads: position 150 of every 151 flagged -- by a separator or by an N inside a longer read
    stretch = "".join(np.array(list("ACGT"))[rng.integers(0, 4, size=151 * 40 - 1)])
    chars = list(stretch)
    for p in range(150, len(chars), 151):
        chars[p] = "N"
    b = ["".join(chars[:151 * 10 - 1]), "".join(chars[151 * 10:151 * 25 - 1]), "".join(chars[151 * 25:])]
    for reads in (a, b):
        seq, off = orc.concat_reads(reads)
        n_pos = int(off[-1]) + len(reads)
        km, cn, st = orc.count_reads(k, True, seq, off)
        bases, mask, n_bases = orc.pack_reads(seq, off)
        assert n_bases == n_pos
        with make_engine("bucketed", k=k, filter_log2_bits=22, seed=5) as eng:
            # (b): 3 reads, but a host that says 40 reads makes n_bases / n_reads = 151: the flags allow it
            n_say = len(reads) if reads is a else 40
            assert n_bases % n_say == 0
            batch = d.ReadBatch.from_packed(eng, bases, mask, n_bases, n_say, orc.n_windows(off, k))
            res = d.KmerCounter(eng).count_reads(batch)
            assert_result_equals(res, km, cn)
            assert res.stats["n_valid"] == st["n_valid"]
            res.close()
