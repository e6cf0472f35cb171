"""ctypes front-end of the CPU oracle (oracle/dk_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: /root/reference holds no source for kmer.rs / counter.rs, so
the oracle restates the written spec, not reference code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcKmer(C.Structure):
    _fields_ = [("hi", C.c_uint64), ("lo", C.c_uint64)]


class OrcStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_windows", C.c_uint64), ("n_valid", C.c_uint64),
                ("n_absent", C.c_uint64), ("n_distinct", C.c_uint64)]

    def as_dict(self):
        return {f: int(getattr(self, f)) for f, _ in self._fields_}


class OrcSynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("read_len", C.c_uint32),
                ("snv_rate", C.c_double), ("denovo_rate", C.c_double), ("err_rate", C.c_double),
                ("n_rate", C.c_double), ("xover_block", C.c_uint64)]


KMER_DT = np.dtype([("hi", "<u8"), ("lo", "<u8")])


def build(force=False):
    so = os.path.join(_HERE, "libdk_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("dk_oracle.c", "dk_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libdk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.orc_fmix64.restype = C.c_uint64
        L.orc_fmix64.argtypes = [C.c_uint64]
        L.orc_hash_kmer.restype = C.c_uint64
        L.orc_hash_kmer.argtypes = [OrcKmer, C.c_int, C.c_uint64]
        L.orc_read_kmers.restype = C.c_uint64
        L.orc_read_kmers.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_bloom_positions.restype = None
        L.orc_bloom_positions.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p, u32p]
        L.orc_bloom_insert_reads.restype = C.c_int
        L.orc_bloom_insert_reads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcStats)]
        L.orc_bloom_probe_reads.restype = C.c_int64
        L.orc_bloom_probe_reads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcStats)]
        L.orc_last_phase_seconds.restype = None
        L.orc_last_phase_seconds.argtypes = [C.POINTER(C.c_double)]
        L.orc_bloom_probe_reads_mt.restype = C.c_int64
        L.orc_bloom_probe_reads_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                               C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcStats), C.c_int]
        L.orc_exact_child_only.restype = C.c_int64
        L.orc_exact_child_only.argtypes = [C.c_int, C.c_int, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcStats)]
        L.orc_count_reads.restype = C.c_int64
        L.orc_count_reads.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcStats)]
        L.orc_pack_reads.restype = C.c_uint64
        L.orc_pack_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_unpack_fixed.restype = None
        L.orc_unpack_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_synth_read.restype = None
        L.orc_synth_read.argtypes = [C.POINTER(OrcSynthCfg), C.c_int, C.c_uint64, C.c_void_p]
        L.orc_synth_mix.restype = C.c_uint64
        L.orc_synth_mix.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    return _LIB


# ---- helpers -------------------------------------------------------------------------------

def concat_reads(reads):
    """list of bytes/str -> (uint8 array, uint64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seq = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seq, offsets


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fmix64(x):
    return int(lib().orc_fmix64(C.c_uint64(x & (2**64 - 1))))


def hash_kmer(hi, lo, k, seed):
    return int(lib().orc_hash_kmer(OrcKmer(hi, lo), k, C.c_uint64(seed)))


def read_kmers(read, k, canonical=True):
    """-> (kmers structured array [hi,lo], valid uint8 array) for one read"""
    b = read.encode() if isinstance(read, str) else bytes(read)
    n = max(0, len(b) - k + 1)
    out = np.zeros(max(n, 1), dtype=KMER_DT)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    buf = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, np.uint8)
    nw = lib().orc_read_kmers(_ptr(buf), len(b), k, int(canonical), _ptr(out), _ptr(valid))
    assert nw == n
    return out[:n], valid[:n]


def bloom_positions(h, log2_bits, n_hashes):
    blk = C.c_uint64()
    bits = (C.c_uint32 * n_hashes)()
    lib().orc_bloom_positions(C.c_uint64(h), log2_bits, n_hashes, C.byref(blk), bits)
    return int(blk.value), [int(b) for b in bits]


def new_filter(log2_bits):
    return np.zeros((1 << log2_bits) // 64, dtype=np.uint64)


def bloom_insert(filt, log2_bits, n_hashes, seed, k, canonical, seq, offsets):
    st = OrcStats()
    rc = lib().orc_bloom_insert_reads(_ptr(filt), log2_bits, n_hashes, C.c_uint64(seed), k, int(canonical),
                                      _ptr(seq), _ptr(offsets), len(offsets) - 1, C.byref(st))
    assert rc == 0
    return st.as_dict()


def n_windows(offsets, k):
    ln = np.diff(offsets.astype(np.int64))
    return int(np.maximum(ln - k + 1, 0).sum())


def bloom_probe(filt, log2_bits, n_hashes, seed, k, canonical, seq, offsets, min_count=1, n_threads=1):
    cap = max(1, n_windows(offsets, k))
    km = np.zeros(cap, dtype=KMER_DT)
    cn = np.zeros(cap, dtype=np.uint32)
    st = OrcStats()
    n = lib().orc_bloom_probe_reads_mt(_ptr(filt), log2_bits, n_hashes, C.c_uint64(seed), k, int(canonical),
                                       min_count, _ptr(seq), _ptr(offsets), len(offsets) - 1,
                                       _ptr(km), _ptr(cn), cap, C.byref(st), n_threads)
    assert n >= 0
    return km[:n].copy(), cn[:n].copy(), st.as_dict()


def last_phase_seconds():
    """(probe, sort, merge) wall seconds of the last bloom_probe call"""
    out = (C.c_double * 3)()
    lib().orc_last_phase_seconds(out)
    return float(out[0]), float(out[1]), float(out[2])


def exact_child_only(k, canonical, pseq, poff, cseq, coff, min_count=1):
    cap = max(1, n_windows(coff, k))
    km = np.zeros(cap, dtype=KMER_DT)
    cn = np.zeros(cap, dtype=np.uint32)
    st = OrcStats()
    n = lib().orc_exact_child_only(k, int(canonical), min_count, _ptr(pseq), _ptr(poff), len(poff) - 1,
                                   _ptr(cseq), _ptr(coff), len(coff) - 1, _ptr(km), _ptr(cn), cap, C.byref(st))
    assert n >= 0
    return km[:n].copy(), cn[:n].copy(), st.as_dict()


def count_reads(k, canonical, seq, offsets):
    cap = max(1, n_windows(offsets, k))
    km = np.zeros(cap, dtype=KMER_DT)
    cn = np.zeros(cap, dtype=np.uint32)
    st = OrcStats()
    n = lib().orc_count_reads(k, int(canonical), _ptr(seq), _ptr(offsets), len(offsets) - 1,
                              _ptr(km), _ptr(cn), cap, C.byref(st))
    assert n >= 0
    return km[:n].copy(), cn[:n].copy(), st.as_dict()


def pack_reads(seq, offsets):
    n_reads = len(offsets) - 1
    total = int(offsets[-1]) + n_reads
    bases = np.zeros((total + 31) // 32 + 1, dtype=np.uint64)
    mask = np.zeros((total + 63) // 64 + 1, dtype=np.uint64)
    t = lib().orc_pack_reads(_ptr(seq), _ptr(offsets), n_reads, _ptr(bases), _ptr(mask))
    assert t == total
    return bases[: (total + 31) // 32], mask[: (total + 63) // 64], total


def unpack_fixed(bases, mask, n_reads, read_len):
    """packed dk_reads words of fixed-length reads -> (uint8 ASCII [n_reads*read_len], offsets)"""
    out = np.zeros(n_reads * read_len, dtype=np.uint8)
    lib().orc_unpack_fixed(_ptr(bases), _ptr(mask), n_reads, read_len, _ptr(out))
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)
    return out, offsets


def synth_cfg(seed=20260313, genome_len=50_000, read_len=150, snv_rate=1e-3, denovo_rate=None,
              err_rate=5e-3, n_rate=1e-4, xover_block=1 << 20):
    if denovo_rate is None:
        denovo_rate = 100.0 / (64 << 20)
    return OrcSynthCfg(seed, genome_len, read_len, snv_rate, denovo_rate, err_rate, n_rate, xover_block)


def synth_reads(cfg, sample, first, count):
    """-> (uint8 seq [count*L], offsets) for reads first..first+count of a sample"""
    L = cfg.read_len
    seq = np.zeros(count * L, dtype=np.uint8)
    base = seq.ctypes.data
    f = lib().orc_synth_read
    for i in range(count):
        f(C.byref(cfg), sample, first + i, C.c_void_p(base + i * L))
    offsets = (np.arange(count + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint64)
    return seq, offsets
