"""Seeded differential sweep of the bucketed family against the oracle: random k, read-length mixes (uniform lengths take the
window-major scan), filter sizes, hash counts and the engine options that pick a kernel variant (slabs, sub-segment split,
level-1 bits, concatenated repart, packed accumulator units, hash windows), insert + accumulate + finish each time.

PARITY UNPINNED vs the reference's Rust code (no source / fixtures in /root/reference); the oracle is the written spec of
DESIGN.md section 2."""
import numpy as np
import pytest

from conftest import random_reads
from oracle import orc

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(90_000 + seed)
    k = int(rng.choice([1, 7, 15, 21, 27, 31, 32, 33, 40, 51, 64]))
    uniform = bool(rng.integers(0, 2))
    lo = int(rng.integers(max(k, 20), 260))
    hi = lo if uniform else lo + int(rng.integers(1, 120))
    log2_bits = int(rng.choice([24, 26, 27, 28]))
    opts = {"slabs": int(rng.choice([0, 2, 4])), "sub_split": int(rng.choice([0, 0, 1, 2])), "scan_bits": int(rng.choice([0, 0, 9])),
            "repart_pieces": int(rng.choice([0, 1, 2])), "accum_min_u": int(rng.choice([0, 4, 8, 10])),
            "accum_plain": int(rng.integers(0, 2)), "scan_positions": int(rng.integers(0, 2)),
            "scan_variant": int(rng.choice([0, 0, 2, 6])), "l2_packed": int(rng.integers(0, 2))}
    opts["l1_layout"] = seed % 2               # (added after the draws above: the cases of earlier rounds keep their inputs)
    return rng, k, lo, hi, log2_bits, int(rng.integers(1, 9)), int(rng.choice([1, 1, 2, 4])), opts


@pytest.mark.parametrize("seed", range(48))
def test_random_geometry_against_the_oracle(seed):
    import denovo_kmer_amd as d
    rng, k, lo, hi, log2_bits, nh, windows, opts = _case(seed)
    n_reads = max(200, 400000 // hi)
    parents = random_reads(rng, n_reads, lo, hi, n_rate=0.003)
    child = random_reads(rng, n_reads, lo, hi, n_rate=0.003) + parents[: n_reads // 2]
    child = child + child[: n_reads // 4]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(log2_bits)
    orc.bloom_insert(f, log2_bits, nh, seed, k, True, pseq, poff)
    mc = 1 + seed % 2
    km, cn, ost = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff, mc)
    with d.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=seed, mode="bucketed") as eng:
        if seed % 2:
            # every second case carves all its memory from a small reserved arena (first fit, splits, coalescing; what does
            # not fit falls through to the caching pool)
            eng.reserve((96 if seed % 4 == 1 else 640) << 20)
        for name, val in opts.items():
            eng.set_option(name, val)
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents[: n_reads // 3])
        ks.insert_sequences(parents[n_reads // 3:])
        assert np.array_equal(ks.to_host(), f), (seed, k, opts)
        acc = d.ChildAccumulator(eng, ks, capacity_records=int(1.5 * ost["n_valid"] / windows) + 1000, window_count=windows)
        got_hi, got_lo, got_cnt, n_absent = [], [], [], 0
        cut = len(child) // 3
        for w in range(windows):
            acc.reset(w)
            for part in (child[:cut], child[cut:cut + 1], child[cut + 1:]):
                acc.add(d.ReadBatch.from_sequences(eng, part))
                assert "overflow_redo" not in [n for n, _ in eng.timings()["stages"]]
            res = acc.finish(min_count=mc)
            h, l, c = res.to_host(sort=False)
            got_hi.append(h); got_lo.append(l); got_cnt.append(c)
            n_absent += res.stats["n_absent"]
            res.close()
        hi_a, lo_a, cnt_a = np.concatenate(got_hi), np.concatenate(got_lo), np.concatenate(got_cnt)
        order = np.lexsort((lo_a, hi_a))
        assert np.array_equal(lo_a[order], km["lo"]) and np.array_equal(hi_a[order], km["hi"]) and np.array_equal(cnt_a[order], cn), (seed, k, opts)
        assert n_absent == ost["n_absent"], (seed, k, opts)
        acc.close()
        ks.close()
        assert eng.info("pool_bytes_in_use") == 0, "something the engine handed out was never returned"
