"""seg_count on a sample whose every absent k-mer occurs exactly twice (a batch accumulated twice)"""
import sys, time
sys.path.insert(0, ".")
import torch
import denovo_kmer_amd as dk

log2_bits, n = 36, 16_000_000
eng = dk.Engine(k=31, filter_log2_bits=log2_bits, n_hashes=4, seed=1, mode="bucketed")
eng.reserve(100 << 30)
gcfg = dk.synth_config(genome_len=1 << 30)
ks = dk.KmerSet(eng)
[ks.insert_reads(dk.ReadBatch.synth(eng, gcfg, s, b * n, n)) for s in (0, 1) for b in range(4)]
rb = [dk.ReadBatch.synth(eng, gcfg, 2, b * n, n) for b in range(4)]
for label, seq in (("4 different batches", [0, 1, 2, 3]), ("2 batches twice", [0, 1, 0, 1]), ("1 batch four times", [0, 0, 0, 0])):
    for cap_batches in (4, 25):
        acc = dk.ChildAccumulator(eng, ks, capacity_records=int(0.16 * cap_batches * n * 120))
        for b in seq:
            acc.add(rb[b])
        for mc in (1, 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = acc.finish(min_count=mc)
            torch.cuda.synchronize()
            print("%-22s capacity for %2d batches, min_count %d: %8.1f ms  %s  units %s emitted %d distinct %d" % (
                label, cap_batches, mc, (time.perf_counter() - t0) * 1e3, [(a, round(b, 1)) for a, b in eng.timings()["stages"]],
                acc.geometry()[:2], res.stats["n_emitted"], res.stats["n_distinct"]), flush=True)
            res.close()
        acc.close()
