import sys
sys.path.insert(0, ".")
import denovo_kmer_amd as dk
n_reads = 40_000_000
gcfg = dk.synth_config(genome_len=200 << 20)
with dk.Engine(k=31, filter_log2_bits=30, seed=20260313, mode="bucketed") as eng:
    b = dk.ReadBatch.synth(eng, gcfg, 2, 0, n_reads)
    c = dk.KmerCounter(eng)
    for it in range(2):
        r = c.count_reads(b)
        t = eng.timings()
        st = r.stats
        n = len(r)
        r.close()
    print("KmerCounter 40 M reads: %.2f Gk-mers/s" % (st["n_windows"] / (t["total_ms"] * 1e-3) / 1e9), round(t["total_ms"], 2),
          [(nm, round(ms, 2)) for nm, ms in t["stages"]], "distinct", st["n_distinct"], "emitted", n, "valid", st["n_valid"])
    assert st["n_distinct"] == n
