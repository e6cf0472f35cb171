"""Throughput of the extraction-only operation (dk_reads_kmers) at configs[1] scale, outputs in HBM."""
import sys
sys.path.insert(0, ".")
import torch
import denovo_kmer_amd as dk

n_reads = 12_800_000
gcfg = dk.synth_config(genome_len=64 << 20)
for k in (31, 51):
    with dk.Engine(k=k, seed=20260313) as eng:
        b = dk.ReadBatch.synth(eng, gcfg, 2, 0, n_reads)
        n = b.stats()["n_bases"]
        lo = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        hi = torch.zeros(n if k > 32 else 1, dtype=torch.int64, device="cuda:0")
        hs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        nk = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        for hashes in (False, True):
            into = {"lo": lo.data_ptr(), "hi": hi.data_ptr() if k > 32 else 0, "hash": hs.data_ptr() if hashes else 0,
                    "not_kmer": nk.data_ptr()}
            for _ in range(3):
                st = b.kmers(into=into)["stats"]
                ms = dict(eng.timings()["stages"])["kmers"]
            out_bytes = n * 8 * (1 + (k > 32) + hashes) + n / 8
            in_bytes = n * 3 / 8
            print(f"k={k} hashes={hashes}: {ms:.2f} ms, {st['n_windows'] / ms / 1e6:.1f} Gk-mers/s, "
                  f"{(in_bytes + out_bytes) / ms / 1e6:.0f} GB/s algorithmic")
