"""GPU parity tests of the exact KmerSet (DK_SET_EXACT): the same C-ABI calls as
the Bloom path, but membership is exact -- what a HashSet of parent k-mers gives -- so the child-only
set must equal the oracle's exact set difference, k-mer for k-mer and count for count.

PARITY UNPINNED vs the reference's Rust code (no source / fixtures in /root/reference); the oracle is
orc_exact_child_only (oracle/dk_oracle.c), the written spec of DESIGN.md section 2.
"""
import numpy as np
import pytest

from conftest import random_reads, related_trio
from oracle import orc
from test_gpu_parity import _result_checksum, assert_result_equals, dk

pytestmark = pytest.mark.gpu

MODES = ["direct", "bucketed"]


def exact_engine(mode, **kw):
    return dk().Engine(mode=mode, set_kind="exact", **kw)


def stage_names(eng):
    return [n for n, _ in eng.timings()["stages"]]


def oracle_exact(parents, child, k, canonical=True, min_count=1):
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    return orc.exact_child_only(k, canonical, pseq, poff, cseq, coff, min_count)


def n_distinct(reads, k, canonical=True):
    seq, off = orc.concat_reads(reads)
    km, _, _ = orc.count_reads(k, canonical, seq, off)
    return len(km), km


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,log2_bits,canonical", [
    (21, 22, True), (31, 22, True), (32, 23, True), (31, 22, False), (1, 20, True), (2, 20, True),
    (15, 21, True), (33, 23, True), (51, 24, True), (64, 23, True), (47, 23, False),
])
def test_exact_trio_parity(rng, mode, k, log2_bits, canonical):
    d = dk()
    parents, child = related_trio(rng, genome_len=3000, n_reads=80, read_len=130)
    with exact_engine(mode, k=k, filter_log2_bits=log2_bits, seed=0xABCDEF12345, canonical=canonical) as eng:
        ks = d.KmerSet(eng)
        # the two parents go in as separate batches; the second repeats part of the first
        ist = ks.insert_sequences(parents[:80])
        assert stage_names(eng)[0] == ("scan_part" if mode == "bucketed" else "insert_direct")
        ks.insert_sequences(parents[60:])
        nd, pk = n_distinct(parents, k, canonical)
        assert ks.popcount() == nd
        s0, o0 = orc.concat_reads(parents[:80])
        assert ist["n_windows"] == orc.n_windows(o0, k)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        names = stage_names(eng)
        if mode == "bucketed":
            assert names[0] == "scan_part" and "seg_exact_probe" in names and "overflow_redo" not in names, names
        else:
            assert names[0] == "probe_direct", names
        km, cn, st = oracle_exact(parents, child, k, canonical)
        assert_result_equals(res, km, cn)
        for key in ("n_reads", "n_windows", "n_valid", "n_absent", "n_distinct"):
            assert res.stats[key] == st[key], (key, res.stats, st)
        assert res.stats["n_emitted"] == len(km)
        # KmerSet::contains is exact: every parent k-mer, none of the child-only ones
        assert ks.contains((pk["hi"], pk["lo"])).all()
        if len(km):
            assert not ks.contains((km["hi"], km["lo"])).any()


def _golden_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "small_trios.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_exact_golden_vectors_on_gpu(case, mode):
    d = dk()
    with exact_engine(mode, k=case["k"], filter_log2_bits=23, seed=case["seed"], canonical=case["canonical"],
                      min_count=case["min_count"]) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(case["parents"])
        assert ks.popcount() == case["parent_distinct"]
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, case["child"]), ks)
        hi, lo, cnt = res.to_host(sort=True)
        assert [[int(a), int(b), int(c)] for a, b, c in zip(hi, lo, cnt)] == case["exact_child_only"]
        for key in ("n_reads", "n_windows", "n_valid", "n_absent", "n_distinct"):
            assert res.stats[key] == case["exact_stats"][key], key


@pytest.mark.parametrize("mode", MODES)
def test_exact_is_a_superset_of_the_bloom_result(rng, mode):
    # a small, crowded Bloom filter loses child-only k-mers to false positives; the exact set loses none
    d = dk()
    k = 25
    parents, child = related_trio(rng, genome_len=20000, n_reads=600, read_len=140)
    km, cn, _ = oracle_exact(parents, child, k)
    with exact_engine(mode, k=k, filter_log2_bits=23, seed=11) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        assert_result_equals(res, km, cn)
    with d.Engine(mode=mode, k=k, filter_log2_bits=20, n_hashes=2, seed=11) as eng:
        kb = d.KmerSet(eng)
        kb.insert_sequences(parents)
        rb = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), kb)
        assert 0 < len(rb) < len(km)


@pytest.mark.parametrize("mode", MODES)
def test_exact_ragged_reads_and_min_count(rng, mode):
    d = dk()
    k = 19
    parents = random_reads(rng, 200, 0, 150, n_rate=0.03, lower_rate=0.3) + ["", "N" * 50, "ACG", "A" * 19]
    child = parents[:50] + random_reads(rng, 150, 0, 150, n_rate=0.03, lower_rate=0.3) + ["", "acgtn" * 9]
    child = child + child[40:90]
    with exact_engine(mode, k=k, filter_log2_bits=22, seed=3, min_count=2) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        km, cn, st = oracle_exact(parents, child, k, min_count=2)
        assert_result_equals(res, km, cn)
        assert res.stats["n_distinct"] == st["n_distinct"] and len(km) < st["n_distinct"]
        # empty batches are fine on both sides
        ks.insert_sequences([])
        assert len(d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, ["", "AC"]), ks)) == 0


@pytest.mark.parametrize("k", [21, 40])
def test_exact_heavy_hitters_take_the_overflow_path(k):
    # one k-mer repeated far beyond its segment region overflows it; the overflow records are inserted
    # into / looked up in the table one by one, with identical results
    d = dk()
    unit = "ACGTTGCATGCCGATAGCTAGCTAGGATCGATCGATTAGC" * 3
    parents = ["A" * 150] * 3000 + [unit] * 10
    child = ["A" * 150] * 3000 + ["C" * 150] * 3000 + [unit[7:] + "TTGACCA"] * 3
    with exact_engine("bucketed", k=k, filter_log2_bits=24, seed=5) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        names = stage_names(eng)
        assert "ovf_insert" in names and "overflow_redo" not in names, names
        assert ks.popcount() == n_distinct(parents, k)[0]
        res = d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), ks)
        names = stage_names(eng)
        assert "ovf_probe" in names and "overflow_redo" not in names, names
        km, cn, st = oracle_exact(parents, child, k)
        assert_result_equals(res, km, cn)
        assert res.stats["n_absent"] == st["n_absent"] and int(cn.max()) == 3000 * (150 - k + 1)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [31, 45])
def test_exact_set_full_is_reported(rng, mode, k):
    # 2^20 bits = 128 KiB = 16 Ki slots (8 Ki for k > 32); 40 000 distinct k-mers cannot fit
    d = dk()
    reads = random_reads(rng, 400, 130, 131)
    with exact_engine(mode, k=k, filter_log2_bits=20, seed=9) as eng:
        ks = d.KmerSet(eng)
        with pytest.raises(d.DkError) as ei:
            ks.insert_sequences(reads)
        assert ei.value.status == 7 and "filter_log2_bits" in str(ei.value)
        # what did fit is still a consistent subset: all slots taken, every held key is one of ours
        cap = (1 << 20) // 64 // (2 if k > 32 else 1)
        assert ks.popcount() == cap
        ks.clear()
        assert ks.popcount() == 0
        ks.insert_sequences(reads[:20])
        assert ks.popcount() == n_distinct(reads[:20], k)[0]


@pytest.mark.parametrize("k", [27, 55])
def test_exact_save_load_and_kind_mismatch(tmp_path, rng, k):
    d = dk()
    parents, child = related_trio(rng, genome_len=4000, n_reads=100, read_len=120)
    path = str(tmp_path / "parents.dkexact")
    km, cn, _ = oracle_exact(parents, child, k)
    with exact_engine("bucketed", k=k, filter_log2_bits=23, seed=77) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        ks.save(path)
        n_keys = ks.popcount()
    with exact_engine("direct", k=k, filter_log2_bits=23, seed=77) as eng:
        k2 = d.KmerSet(eng)
        k2.load(path)
        assert k2.popcount() == n_keys
        assert_result_equals(d.KmerCounter(eng).child_only(d.ReadBatch.from_sequences(eng, child), k2), km, cn)
    # a Bloom engine refuses the exact file, and the other way round
    with d.Engine(k=k, filter_log2_bits=23, seed=77) as eng:
        kb = d.KmerSet(eng)
        with pytest.raises(d.DkError):
            kb.load(path)
        bpath = str(tmp_path / "parents.dkbloom")
        kb.save(bpath)
    with exact_engine("direct", k=k, filter_log2_bits=23, seed=77) as eng:
        with pytest.raises(d.DkError):
            d.KmerSet(eng).load(bpath)


def test_exact_families_agree_at_scale():
    """2 M x 150 bp reads per sample, k=31, 2^34-bit exact set (~40 % load): properties the oracle cannot check at
    this size in seconds -- no false negatives, idempotence, batch-order independence, and agreement of
    the two independent kernel families (LDS tables vs. direct HBM lookups)."""
    d = dk()
    n_reads, k, log2_bits = 2_000_000, 31, 34
    gcfg = d.synth_config(genome_len=10 << 20)
    with exact_engine("bucketed", k=k, filter_log2_bits=log2_bits, seed=20260313) as eb:
        ks = d.KmerSet(eb)
        p0 = d.ReadBatch.synth(eb, gcfg, 0, 0, n_reads)
        p1 = d.ReadBatch.synth(eb, gcfg, 1, 0, n_reads)
        ks.insert_reads(p0)
        assert stage_names(eb) == ["scan_part", "repart", "seg_exact_insert"]
        n0 = ks.popcount()
        ks.insert_reads(p0)
        assert ks.popcount() == n0                        # idempotent
        self_probe = d.KmerCounter(eb).child_only(p0, ks)
        assert self_probe.stats["n_absent"] == 0 and len(self_probe) == 0      # no false negatives
        ks.insert_reads(p1)
        n01 = ks.popcount()
        assert n0 < n01 < 2 * n0
        k2 = d.KmerSet(eb)                                # other order, halves
        k2.insert_reads(p1)
        k2.insert_reads(d.ReadBatch.synth(eb, gcfg, 0, n_reads // 2, n_reads - n_reads // 2))
        k2.insert_reads(d.ReadBatch.synth(eb, gcfg, 0, 0, n_reads // 2))
        assert k2.popcount() == n01
        k2.close()
        # KmerCounter of the parents has exactly as many distinct k-mers as the set holds
        c0 = d.KmerCounter(eb).count_reads(p0)
        assert c0.stats["n_distinct"] == n0
        c0.close()
        child = d.ReadBatch.synth(eb, gcfg, 2, 0, n_reads)
        rb = d.KmerCounter(eb).child_only(child, ks)
        assert stage_names(eb) == ["scan_part", "repart", "seg_exact_probe", "seg_count"]
        sb, cb = rb.stats, _result_checksum(rb)
        assert cb[0] == sb["n_distinct"] == sb["n_emitted"] and cb[1] == sb["n_absent"]
        assert 0.05 * sb["n_valid"] < sb["n_absent"] < 0.3 * sb["n_valid"]
        table = ks.to_host()
    with exact_engine("direct", k=k, filter_log2_bits=log2_bits, seed=20260313) as ed:
        kd = d.KmerSet(ed)
        kd.from_host(table)
        rd = d.KmerCounter(ed).child_only(d.ReadBatch.synth(ed, gcfg, 2, 0, n_reads), kd)
        assert stage_names(ed)[0] == "probe_direct"
        for key in ("n_windows", "n_valid", "n_absent", "n_distinct", "n_emitted"):
            assert rd.stats[key] == sb[key], key
        assert _result_checksum(rd) == cb
        # a set built by the direct family holds the same keys
        kd.clear()
        kd.insert_reads(d.ReadBatch.synth(ed, gcfg, 0, 0, n_reads))
        assert stage_names(ed) == ["insert_direct"]
        assert kd.popcount() == n0
