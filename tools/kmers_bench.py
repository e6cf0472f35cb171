#!/usr/bin/env python3
"""Throughput of the extraction-only operation (dk_reads_kmers, the kmer.rs stand-in) at configs[1] scale with the
outputs in HBM: one JSON line per configuration with the kernel's HIP-event time and its roofline fraction.

    python tools/kmers_bench.py [--steps 5]          (profiled by: BENCH="tools/kmers_bench.py" tools/profile_round.sh r02_kmers)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import denovo_kmer_amd as dk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--reads", type=int, default=12_800_000)
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
args, _ = ap.parse_known_args()
gcfg = dk.synth_config(genome_len=64 << 20)
for k in (31, 51):
    with dk.Engine(k=k, seed=20260313) as eng:
        for ov in args.opt:
            name, _, val = ov.partition("=")
            eng.set_option(name, int(val))
        b = dk.ReadBatch.synth(eng, gcfg, 2, 0, args.reads)
        n = b.stats()["n_bases"]
        lo = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        hi = torch.zeros(n if k > 32 else 1, dtype=torch.int64, device="cuda:0")
        hs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        nk = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        for hashes in (False, True):
            into = {"lo": lo.data_ptr(), "hi": hi.data_ptr() if k > 32 else 0, "hash": hs.data_ptr() if hashes else 0,
                    "not_kmer": nk.data_ptr()}
            ms = []
            for i in range(args.warmup + args.steps):
                st = b.kmers(into=into)["stats"]
                if i >= args.warmup:
                    ms.append(dict(eng.timings()["stages"])["kmers"])
            t = sum(ms) / len(ms)
            algo = n * 3 / 8 + n * 8 * (1 + (k > 32) + hashes) + n / 8          # stream in; k-mer words (+ hash) and the not-a-k-mer bits out
            print(json.dumps({"op": "dk_reads_kmers", "k": k, "hashes": hashes, "reads": args.reads, "positions": n, "kernel_ms": t,
                              "gkmers_s": st["n_windows"] / t / 1e6, "algorithmic_bytes": algo, "achieved_gbs": algo / t / 1e6,
                              "frac_of_8TBs": algo / t / 1e6 / 8000.0}), flush=True)
