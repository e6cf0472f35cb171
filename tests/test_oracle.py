"""CPU tests of the oracle: hand-derived known answers, agreement with the independent
pure-Python restatement, and the committed golden vectors.

PARITY UNPINNED: no reference fixture exists for this path -- these tests pin the
oracle to the written spec, not to the reference's Rust code.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import pyref
from conftest import random_reads, related_trio
from oracle import orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "small_trios.json")


def golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


# ---- hand-derived known answers -----------------------------------------------------------------

def test_fmix64_known_values():
    # murmur3 finaliser: 0 is a fixed point; fmix64(1) is the widely published value
    assert orc.fmix64(0) == 0
    assert orc.fmix64(1) == 0xB456BCFC34C2CB2C
    for x in (1, 2, 0xDEADBEEF, 2**64 - 1, 0x0123456789ABCDEF):
        assert orc.fmix64(x) == pyref.fmix64(x)


def test_encoding_and_canonical_by_hand():
    # A=0 C=1 G=2 T=3, first base most significant: ACG = 0b00_01_10 = 6, revcomp CGT = 27
    km, v = orc.read_kmers("ACG", 3)
    assert list(v) == [1] and km[0]["lo"] == 6 and km[0]["hi"] == 0
    km, v = orc.read_kmers("CGT", 3)
    assert km[0]["lo"] == 6
    km, v = orc.read_kmers("CGT", 3, canonical=False)
    assert km[0]["lo"] == 27
    # palindrome ACGT (its own reverse complement) = 0b00011011
    km, v = orc.read_kmers("ACGT", 4)
    assert km[0]["lo"] == 0b00011011
    # TTTT -> revcomp AAAA = 0 is smaller
    km, v = orc.read_kmers("TTTT", 4)
    assert km[0]["lo"] == 0
    # lowercase is the same base
    km, v = orc.read_kmers("acg", 3)
    assert list(v) == [1] and km[0]["lo"] == 6


def test_n_invalidates_every_covering_window():
    km, v = orc.read_kmers("ACGTNACGTAC", 3)
    assert list(v) == [1, 1, 0, 0, 0, 1, 1, 1, 1]
    assert len(km) == 9
    # any non-ACGT byte counts as N
    for ch in "NnRX-.*":
        _, v = orc.read_kmers("ACG" + ch + "ACG", 3)
        assert list(v) == [1, 0, 0, 0, 1]


def test_short_and_empty_reads():
    for r in ("", "A", "AC"):
        km, v = orc.read_kmers(r, 3)
        assert len(km) == 0 and len(v) == 0


def test_k32_and_wide_kmers_by_hand():
    km, v = orc.read_kmers("A" * 31 + "C", 32)
    assert km[0]["lo"] == 1 and km[0]["hi"] == 0             # revcomp G T^31 is larger
    km, v = orc.read_kmers("T" * 32, 32, canonical=False)
    assert km[0]["lo"] == 2**64 - 1
    # k = 33: C A^32 = 1 << 64 -> hi = 1, lo = 0 (revcomp T^32 G is larger)
    km, v = orc.read_kmers("C" + "A" * 32, 33)
    assert km[0]["hi"] == 1 and km[0]["lo"] == 0
    # k = 64, all T forward = 2^128 - 1; canonical = all A = 0
    km, v = orc.read_kmers("T" * 64, 64, canonical=False)
    assert km[0]["hi"] == 2**64 - 1 and km[0]["lo"] == 2**64 - 1
    km, v = orc.read_kmers("T" * 64, 64)
    assert km[0]["hi"] == 0 and km[0]["lo"] == 0


def test_bloom_positions_by_hand():
    # h = all ones, 2^20 bits -> 2^11 blocks: block = top 11 bits = 2047; a = 511, d = 511
    blk, bits = orc.bloom_positions(2**64 - 1, 20, 4)
    assert blk == 2047 and bits == [511, 510, 509, 508]
    # h with only bit 63 set: block = 1024 of 2048; a = 0, d = 1 (forced odd)
    blk, bits = orc.bloom_positions(1 << 63, 20, 3)
    assert blk == 1024 and bits == [0, 1, 2]
    blk, bits = orc.bloom_positions(0x123456789ABCDEF0, 30, 5)
    assert (blk, bits) == pyref.bloom_positions(0x123456789ABCDEF0, 30, 5)


def test_filter_word_layout_single_kmer():
    # one k-mer: the filter's set bits are exactly the predicted (block, bit) positions,
    # bit t of block b living in u64 word 8b + (t >> 6), bit t & 63
    seq, off = orc.concat_reads(["ACGTACGTACGTACGTACGTA"])
    f = orc.new_filter(20)
    orc.bloom_insert(f, 20, 4, 77, 21, True, seq, off)
    v = pyref.canonical_int("ACGTACGTACGTACGTACGTA")
    blk, bits = pyref.bloom_positions(pyref.hash_kmer(v, 21, 77), 20, 4)
    expect = {}
    for t in bits:
        expect[8 * blk + (t >> 6)] = expect.get(8 * blk + (t >> 6), 0) | (1 << (t & 63))
    nz = {int(i): int(f[i]) for i in np.nonzero(f)[0]}
    assert nz == expect


# ---- oracle vs the independent Python restatement -----------------------------------------------

@pytest.mark.parametrize("k", [1, 2, 5, 16, 21, 31, 32, 33, 40, 51, 63, 64])
@pytest.mark.parametrize("canonical", [True, False])
def test_read_kmers_match_pyref(rng, k, canonical):
    for read in random_reads(rng, 6, 0, 140, n_rate=0.02, lower_rate=0.1):
        km, v = orc.read_kmers(read, k, canonical)
        ref = pyref.read_kmers(read, k, canonical)
        assert len(km) == len(ref)
        for (ok, val), a, b in zip(ref, km, v):
            assert bool(b) == ok
            if ok:
                assert (int(a["hi"]) << 64) | int(a["lo"]) == val


@pytest.mark.parametrize("k,log2_bits,nh", [(21, 20, 4), (31, 21, 3), (51, 20, 2), (13, 20, 7)])
def test_child_only_matches_pyref(rng, k, log2_bits, nh):
    parents, child = related_trio(rng, genome_len=600, n_reads=12, read_len=90)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(log2_bits)
    orc.bloom_insert(f, log2_bits, nh, 31337, k, True, pseq, poff)
    km, cn, st = orc.bloom_probe(f, log2_bits, nh, 31337, k, True, cseq, coff)
    ref, bl = pyref.child_only(parents, child, k, log2_bits, nh, 31337)
    got = [((int(a["hi"]) << 64) | int(a["lo"]), int(c)) for a, c in zip(km, cn)]
    assert got == ref
    assert {int(i): int(f[i]) for i in np.nonzero(f)[0]} == bl.words()
    assert st["n_distinct"] == len(ref) and st["n_absent"] == sum(c for _, c in ref)


def test_pack_reads_matches_pyref(rng):
    reads = random_reads(rng, 20, 0, 100, n_rate=0.05, lower_rate=0.3) + ["", "", "NNN", "A"]
    seq, off = orc.concat_reads(reads)
    b, m, n = orc.pack_reads(seq, off)
    rb, rm, rn = pyref.pack_reads(reads)
    assert n == rn and list(map(int, b)) == rb and list(map(int, m)) == rm


def test_unpack_fixed_inverts_pack(rng):
    reads = random_reads(rng, 40, 75, 75, n_rate=0.05)
    seq, off = orc.concat_reads(reads)
    b, m, n = orc.pack_reads(seq, off)
    out, o2 = orc.unpack_fixed(b, m, 40, 75)
    assert bytes(out).decode() == "".join(reads).upper() and np.array_equal(o2, off)


def test_exact_is_superset_of_bloom(rng):
    # spec A-6 invariant: Bloom negatives are exact, so bloom child-only is a subset of exact child-only
    parents = random_reads(rng, 150, 200, 200)
    child = random_reads(rng, 20, 200, 200) + parents[:5]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(20)
    orc.bloom_insert(f, 20, 1, 5, 23, True, pseq, poff)
    km, cn, _ = orc.bloom_probe(f, 20, 1, 5, 23, True, cseq, coff)
    ekm, ecn, _ = orc.exact_child_only(23, True, pseq, poff, cseq, coff)
    b = {(int(a["hi"]), int(a["lo"])): int(c) for a, c in zip(km, cn)}
    e = {(int(a["hi"]), int(a["lo"])): int(c) for a, c in zip(ekm, ecn)}
    assert set(b) <= set(e) and len(b) < len(e)          # this load produces false positives
    assert all(e[x] == b[x] for x in b)


def test_multithreaded_probe_equals_single_thread(rng):
    parents, child = related_trio(rng, genome_len=3000, n_reads=200, read_len=100)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child + child[:50])
    f = orc.new_filter(20)
    orc.bloom_insert(f, 20, 3, 5, 27, True, pseq, poff)
    ref = orc.bloom_probe(f, 20, 3, 5, 27, True, cseq, coff, min_count=1, n_threads=1)
    for t in (2, 3, 8):
        got = orc.bloom_probe(f, 20, 3, 5, 27, True, cseq, coff, min_count=1, n_threads=t)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2]
    ref2 = orc.bloom_probe(f, 20, 3, 5, 27, True, cseq, coff, min_count=2, n_threads=1)
    got2 = orc.bloom_probe(f, 20, 3, 5, 27, True, cseq, coff, min_count=2, n_threads=4)
    assert np.array_equal(got2[0], ref2[0]) and np.array_equal(got2[1], ref2[1]) and len(ref2[0]) < len(ref[0])


def test_count_reads_is_probe_of_empty_filter(rng):
    reads = random_reads(rng, 10, 30, 80, n_rate=0.01) * 2
    seq, off = orc.concat_reads(reads)
    km, cn, st = orc.count_reads(17, True, seq, off)
    f = orc.new_filter(20)
    km2, cn2, st2 = orc.bloom_probe(f, 20, 4, 0, 17, True, seq, off)
    assert np.array_equal(km, km2) and np.array_equal(cn, cn2)
    assert cn.min() >= 2                                   # every read appears twice


# ---- golden vectors ------------------------------------------------------------------------------

@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_golden(case):
    pseq, poff = orc.concat_reads(case["parents"])
    cseq, coff = orc.concat_reads(case["child"])
    f = orc.new_filter(case["filter_log2_bits"])
    ist = orc.bloom_insert(f, case["filter_log2_bits"], case["n_hashes"], case["seed"], case["k"],
                           case["canonical"], pseq, poff)
    assert ist == case["insert_stats"]
    assert hashlib.sha256(f.tobytes()).hexdigest() == case["filter_sha256"]
    km, cn, pst = orc.bloom_probe(f, case["filter_log2_bits"], case["n_hashes"], case["seed"], case["k"],
                                  case["canonical"], cseq, coff, case["min_count"])
    assert pst == case["probe_stats"]
    got = [[int(a["hi"]), int(a["lo"]), int(c)] for a, c in zip(km, cn)]
    assert got == case["child_only"]


@pytest.mark.parametrize("case", [c for c in golden_cases() if c["name"] != "loaded_k25_h1"],
                         ids=lambda c: c["name"])
def test_pyref_reproduces_golden(case):
    # the independent restatement derives the same vectors without touching the C oracle
    ref, bl = pyref.child_only(case["parents"], case["child"], case["k"], case["filter_log2_bits"],
                               case["n_hashes"], case["seed"], case["canonical"], case["min_count"])
    assert [[v >> 64, v & pyref.M64, c] for v, c in ref] == case["child_only"]
    words = bl.words()
    assert sum(bin(w).count("1") for w in words.values()) == case["filter_popcount"]
    for i, w in case["filter_nonzero_words"]:
        assert words[i] == w


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_exact_set_golden_by_oracle_and_pyref(case):
    # DK_SET_EXACT expectations: C oracle and the Python set-difference restatement agree with the file
    pseq, poff = orc.concat_reads(case["parents"])
    cseq, coff = orc.concat_reads(case["child"])
    km, cn, st = orc.exact_child_only(case["k"], case["canonical"], pseq, poff, cseq, coff, case["min_count"])
    assert [[int(a["hi"]), int(a["lo"]), int(c)] for a, c in zip(km, cn)] == case["exact_child_only"]
    assert st == case["exact_stats"]
    assert len(orc.count_reads(case["k"], case["canonical"], pseq, poff)[0]) == case["parent_distinct"]
    if case["name"] != "loaded_k25_h1":            # 45 000 k-mers of 25 bases: too slow for the string-based restatement
        ref, n_parents = pyref.exact_child_only(case["parents"], case["child"], case["k"], case["canonical"], case["min_count"])
        assert [[v >> 64, v & pyref.M64, c] for v, c in ref] == case["exact_child_only"]
        assert n_parents == case["parent_distinct"]
    # Bloom result is a subset of the exact one (no false negatives)
    exact = {(hi, lo): c for hi, lo, c in case["exact_child_only"]}
    assert all(exact.get((hi, lo)) == c for hi, lo, c in case["child_only"])


# ---- synthetic generator --------------------------------------------------------------------------

def test_synth_reads_are_deterministic_and_related():
    cfg = orc.synth_cfg(genome_len=20_000, err_rate=0.0, n_rate=0.0, snv_rate=0.0, denovo_rate=0.0)
    a, off = orc.synth_reads(cfg, 0, 5, 4)
    b, _ = orc.synth_reads(cfg, 0, 5, 4)
    assert np.array_equal(a, b) and len(a) == 4 * 150
    # without SNVs / errors every child k-mer exists in the (identical) parent genome
    pseq, poff = orc.synth_reads(cfg, 0, 0, 4000)
    cseq, coff = orc.synth_reads(cfg, 2, 0, 200)
    ekm, _, st = orc.exact_child_only(21, True, pseq, poff, cseq, coff)
    assert st["n_valid"] == 200 * 130
    assert st["n_absent"] < 0.05 * st["n_valid"]          # only coverage gaps
    # with errors on, a sizeable fraction of child windows becomes child-only
    cfg2 = orc.synth_cfg(genome_len=20_000)
    cseq2, coff2 = orc.synth_reads(cfg2, 2, 0, 200)
    assert not np.array_equal(cseq, cseq2)
