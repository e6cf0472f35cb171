#!/usr/bin/env python3
"""Generate tests/golden/*.json from the CPU oracle (oracle/dk_oracle.c).

PARITY UNPINNED: the reference (/root/reference) ships no fixtures or golden vectors for this
path, so these vectors pin the build's own spec: they are produced by the C
oracle and re-derived independently by tests/pyref.py in tests/test_oracle.py.
Re-run:  python tools/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402
from conftest import random_reads, related_trio  # noqa: E402


def case(name, parents, child, k, log2_bits, n_hashes, seed, canonical=True, min_count=1):
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    filt = orc.new_filter(log2_bits)
    ist = orc.bloom_insert(filt, log2_bits, n_hashes, seed, k, canonical, pseq, poff)
    km, cn, pst = orc.bloom_probe(filt, log2_bits, n_hashes, seed, k, canonical, cseq, coff, min_count)
    ekm, ecn, est = orc.exact_child_only(k, canonical, pseq, poff, cseq, coff, min_count)
    nz = np.nonzero(filt)[0]
    return {
        "name": name, "k": k, "filter_log2_bits": log2_bits, "n_hashes": n_hashes, "seed": seed,
        "canonical": canonical, "min_count": min_count,
        "parents": parents, "child": child,
        "insert_stats": ist, "probe_stats": pst,
        "filter_sha256": hashlib.sha256(filt.tobytes()).hexdigest(),
        "filter_popcount": int(np.unpackbits(filt.view(np.uint8)).sum()),
        "filter_nonzero_words": [[int(i), int(filt[i])] for i in nz[:64]],
        "child_only": [[int(a["hi"]), int(a["lo"]), int(c)] for a, c in zip(km, cn)],
        "exact_child_only_n": int(len(ekm)),
        "exact_stats": est,
        # DK_SET_EXACT: the exact set difference and the number of distinct parent k-mers
        "exact_child_only": [[int(a["hi"]), int(a["lo"]), int(c)] for a, c in zip(ekm, ecn)],
        "parent_distinct": int(len(orc.count_reads(k, canonical, pseq, poff)[0])),
    }


def main():
    rng = np.random.default_rng(20260313)
    cases = []
    p, c = related_trio(rng, genome_len=1500, n_reads=30, read_len=80)
    cases.append(case("trio_k21", p, c, 21, 20, 4, 0x5EED))
    p, c = related_trio(rng, genome_len=1500, n_reads=30, read_len=80)
    cases.append(case("trio_k31", p, c, 31, 21, 3, 12345))
    p, c = related_trio(rng, genome_len=1500, n_reads=20, read_len=120)
    cases.append(case("trio_k51", p, c, 51, 20, 4, 99))
    p, c = related_trio(rng, genome_len=800, n_reads=20, read_len=70)
    cases.append(case("trio_k32_fwd_min2", p, c, 32, 20, 2, 7, canonical=False, min_count=2))
    p = random_reads(rng, 12, 0, 90, n_rate=0.03, lower_rate=0.2)
    c = random_reads(rng, 12, 0, 90, n_rate=0.03, lower_rate=0.2) + p[:3] + ["", "N" * 40, "ACGT"]
    cases.append(case("ragged_k15", p, c, 15, 20, 5, 1))
    p, c = related_trio(rng, genome_len=1000, n_reads=15, read_len=100)
    cases.append(case("trio_k64", p, c, 64, 20, 4, 2**63 + 5))
    # heavily loaded filter (one hash, ~5 % fill): Bloom false positives make the result differ
    # from the exact set, which is what spec A-6 / SURVEY H1 is about
    p = random_reads(rng, 200, 250, 250)
    c = random_reads(rng, 24, 250, 250) + p[:4]
    cases.append(case("loaded_k25_h1", p, c, 25, 20, 1, 424242))
    out = os.path.join(ROOT, "tests", "golden", "small_trios.json")
    with open(out, "w") as f:
        json.dump({"generator": "tools/make_golden.py", "oracle": "oracle/dk_oracle.c",
                   "parity": "unpinned (no reference fixtures exist)", "cases": cases}, f, indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
