"""where does an overlapped upload lose its time?  (diagnostic for bench.py's ingest section)"""
import sys, time
sys.path.insert(0, ".")
import torch
import denovo_kmer_amd as dk

log2_bits, n = 36, 16_000_000
eng = dk.Engine(k=31, filter_log2_bits=log2_bits, n_hashes=4, seed=1, mode="bucketed")
eng.reserve(60 << 30)
gcfg = dk.synth_config(genome_len=1 << 30)
ks = dk.KmerSet(eng)
ks.insert_reads(dk.ReadBatch.synth(eng, gcfg, 0, 0, n))
host, meta = [], []
for b in range(2):
    rb = dk.ReadBatch.synth(eng, gcfg, 2, b * n, n)
    bases, mask, nb = rb.download()
    pp = dk.PinnedPacked(nb)
    pp.bases[:] = bases
    pp.mask[:] = mask
    host.append(pp)
    meta.append((nb, n, rb.stats()["n_windows"]))
    rb.close()
acc = dk.ChildAccumulator(eng, ks, capacity_records=int(0.3 * 6 * n * 120))
def t(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print("%-40s %.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True); return r
r0 = t("upload + wait", lambda: (lambda x: (x.wait(), x)[1])(dk.ReadBatch.from_packed_async(eng, host[0], *meta[0])))
t("add (uploaded batch)", lambda: acc.add(r0)); print(eng.timings())
t("add again", lambda: acc.add(r0)); print(eng.timings())
def overlapped():
    nxt = dk.ReadBatch.from_packed_async(eng, host[1], *meta[1])
    acc.add(r0)
    return nxt
r1 = t("async upload of next + add", overlapped); print(eng.timings())
t("wait", lambda: r1.wait())
def fresh():
    x = dk.ReadBatch.from_packed_async(eng, host[0], *meta[0])
    acc.add(x)
    return x
r2 = t("async upload + add of the same batch", fresh); print(eng.timings())
