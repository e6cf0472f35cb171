"""How long does the counting pass of a whole-genome hash window take as a function of the accumulator's unit stride?
Fills an accumulator of the configs[2] geometry (2^39-bit set, one of two windows) with ~9.6 x 10^9 nearly-unique records
(no set attached: every k-mer of the batch counts as absent; a 2^40-base synthetic genome makes repeats rare, like the
sequencing-error k-mers of the real pass) and times dk_accum_finish.   python tools/experiments/accum_count.py [caps...]"""
import sys
import time
sys.path.insert(0, ".")
import denovo_kmer_amd as dk

caps = [int(a) for a in sys.argv[1:]] or [0, 11792, 12304, 12816, 13328, 16400]
n_reads, n_adds = 32_000_000, 5
gcfg = dk.synth_config(genome_len=1 << 40)
with dk.Engine(k=31, filter_log2_bits=39, seed=20260313, mode="bucketed") as eng:
    eng.set_option("multiplicity_hint", 2)
    batches = [dk.ReadBatch.synth(eng, gcfg, 2, i * n_reads, n_reads) for i in range(n_adds)]
    for cap in caps:
        eng.set_option("accum_unit_cap", cap)
        acc = dk.ChildAccumulator(eng, None, capacity_records=int(10.2e9), window_index=0, window_count=2)
        n_units, unit_cap, _ = acc.geometry()
        for b in batches:
            acc.add(b)
        t0 = time.perf_counter()
        res = acc.finish(min_count=2)
        wall = time.perf_counter() - t0
        t = eng.timings()
        st = acc.stats()
        print("unit_cap %6d (stride %8d B = %3d x 4 KiB + %4d B) units %d: finish %.1f ms device (%s), %.2f s wall; %d records, %d emitted"
              % (unit_cap, unit_cap * 8, unit_cap * 8 // 4096, unit_cap * 8 % 4096, n_units, t["total_ms"],
                 ", ".join("%s %.1f" % (n, ms) for n, ms in t["stages"]), wall, st["n_absent"], len(res)), flush=True)
        res.close()
        acc.close()
