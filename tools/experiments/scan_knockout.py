"""scan_part's time inside a parent insert of one 48 M-read batch (2^39-bit set), for builds of the library that knock one part
of the kernel out (-DDK_KO_STORE / DK_KO_ATOMIC / DK_KO_HASH: results are wrong on purpose; only the timing is looked at).
  DK_LIB_PATH=ab/lib_ko_STORE.so python tools/experiments/scan_knockout.py"""
import os
import sys
sys.path.insert(0, ".")
import denovo_kmer_amd as dk

n_reads = 48_000_000
gcfg = dk.synth_config(genome_len=6_000_000_000, read_len=150)
with dk.Engine(k=31, filter_log2_bits=39, n_hashes=4, seed=20260313, mode="bucketed") as eng:
    eng.set_option("multiplicity_hint", 3)
    ks = dk.KmerSet(eng)
    b = dk.ReadBatch.synth(eng, gcfg, 0, 0, n_reads)
    for _ in range(3):
        ks.insert_reads(b)
        st = {}
        for name, ms in eng.timings()["stages"]:
            st[name] = round(st.get(name, 0.0) + ms, 2)
    print(os.environ.get("DK_LIB_PATH", "default"), st, flush=True)
