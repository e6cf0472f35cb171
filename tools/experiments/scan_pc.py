"""Does scan_part's speed follow where its code was loaded?  (diagnostic build: -DDK_DEBUG_INFO records the kernel's PC)
  DK_LIB_PATH=ab/lib_dbginfo.so python tools/experiments/scan_pc.py [ont|wgs]"""
import sys
sys.path.insert(0, ".")
import denovo_kmer_amd as dk

which = sys.argv[1] if len(sys.argv) > 1 else "ont"
if which == "ont":
    k, L, n_reads, bits = 51, 10000, 192000, 35
    gcfg = dk.synth_config(genome_len=64 << 20, read_len=L, err_rate=0.05)
else:
    k, L, n_reads, bits = 31, 150, 24_000_000, 37
    gcfg = dk.synth_config(genome_len=3_000_000_000, read_len=L)
with dk.Engine(k=k, filter_log2_bits=bits, n_hashes=4, seed=20260313, mode="bucketed") as eng:
    for ov in sys.argv[2:]:
        name, _, val = ov.partition("=")
        eng.set_option(name, int(val))
    ks = dk.KmerSet(eng)
    b = dk.ReadBatch.synth(eng, gcfg, 0, 0, n_reads)
    for _ in range(3):
        ks.insert_reads(b)
        st = dict((n, round(ms, 2)) for n, ms in eng.timings()["stages"])
    pc = eng.info("dbg7")
    print(which, sys.argv[2:], st, "scan_part pc 0x%x  mod 64K 0x%x  mod 4K 0x%x" % (pc, pc & 0xFFFF, pc & 0xFFF), flush=True)
