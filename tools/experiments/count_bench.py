import sys, time
sys.path.insert(0, ".")
import denovo_kmer_amd as dk
n_reads = 12_800_000
gcfg = dk.synth_config(genome_len=64 << 20)
for log2 in (34, 36):
    with dk.Engine(k=31, filter_log2_bits=log2, seed=20260313, mode="bucketed") as eng:
        b = dk.ReadBatch.synth(eng, gcfg, 2, 0, n_reads)
        c = dk.KmerCounter(eng)
        for it in range(3):
            r = c.count_reads(b)
            t = eng.timings()
            st = r.stats
            r.close()
        print(log2, "KmerCounter: %.2f Gk-mers/s" % (st["n_windows"] / (t["total_ms"] * 1e-3) / 1e9), round(t["total_ms"], 2),
              [(n, round(ms, 2)) for n, ms in t["stages"]], "distinct", st["n_distinct"], "valid", st["n_valid"])
