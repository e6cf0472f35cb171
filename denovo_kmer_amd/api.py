"""Host-side mirror of the reference's k-mer API over the C ABI.

The reference (jlanej/denovo_kmer) exposes `KmerCounter` and `KmerSet` from counter.rs and the
extraction helpers from kmer.rs (both NOT IN MOUNT -- SURVEY.md 0.1); the classes here keep those
names and the argument meaning BASELINE.json describes (k, sequences in, k-mer -> count out,
set membership), and delegate every computation to libdenovo_kmer.so.  Nothing here computes
k-mers on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DkConfig, DkStats, DkSynthConfig, DkTimings, check

# kernel families smoke() and the GPU tests sweep
SMOKE_MODES = ("direct", "bucketed")

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_BASE = "ACGT"


def kmer_from_str(s):
    """'ACGT' -> (hi, lo) under spec A-1 (first base most significant).  Host utility only."""
    v = 0
    for ch in s.upper():
        v = (v << 2) | _CODE[ch]
    return v >> 64, v & (2**64 - 1)


def kmer_to_str(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return "".join(_BASE[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def _concat(reads):
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seq = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seq, offsets


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Engine:
    """One GPU + one stream + the k-mer / filter geometry (dk_engine)."""

    def __init__(self, k=31, canonical=True, filter_log2_bits=30, n_hashes=4, seed=0x5EED,
                 min_count=1, device_id=0, mode="auto", rank=0, world_size=1, stream=None, set_kind="bloom"):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        mode_id = {"auto": _lib.MODE_AUTO, "direct": _lib.MODE_DIRECT, "bucketed": _lib.MODE_BUCKETED}[mode]
        kind_id = {"bloom": _lib.SET_BLOOM, "exact": _lib.SET_EXACT}[set_kind]
        cfg = DkConfig(C.sizeof(DkConfig), k, int(bool(canonical)), filter_log2_bits, n_hashes,
                       seed & (2**64 - 1), min_count, device_id, rank, world_size, mode_id, kind_id, stream)
        check(self._lib.dk_engine_create(C.byref(cfg), C.byref(self._h)))
        self.k = k
        self.canonical = bool(canonical)
        self.filter_log2_bits = filter_log2_bits
        self.n_hashes = n_hashes
        self.seed = seed & (2**64 - 1)
        self.min_count = min_count
        self.device_id = device_id
        self.mode = mode
        self.set_kind = set_kind          # "bloom": blocked Bloom filter; "exact": exact set (HashSet semantics)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("engine is closed")
        return self._h

    def check(self, status):
        check(status, self._h)

    def synchronize(self):
        self.check(self._lib.dk_engine_synchronize(self.handle))

    def timings(self):
        t = DkTimings()
        self.check(self._lib.dk_engine_timings(self.handle, C.byref(t)))
        return t.as_dict()

    def set_option(self, name, value):
        """run-time option of the engine (dk_engine_set_option): "multiplicity_hint" for capacity planning, and the
        validated test hooks that force a kernel geometry ("scan_variant", "force_l3", ...)"""
        self.check(self._lib.dk_engine_set_option(self.handle, name.encode(), int(value)))

    def trim(self):
        """free the engine's cached workspace blocks now (dk_engine_trim) -> bytes freed"""
        n = C.c_uint64()
        self.check(self._lib.dk_engine_trim(self.handle, C.byref(n)))
        return int(n.value)

    def reserve(self, n_bytes):
        """allocate one arena of n_bytes now; every later workspace / batch / set / accumulator of the engine is carved
        from it (dk_engine_reserve); 0 hands back the arenas nothing lives in"""
        self.check(self._lib.dk_engine_reserve(self.handle, int(n_bytes)))

    def info(self, name):
        """what the engine did / holds (dk_engine_get_info): "plan_slabs", "plan_sbits", "pool_bytes_in_use", ..."""
        v = C.c_int64()
        self.check(self._lib.dk_engine_get_info(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (dk_comm_unique_id): one rank creates it, the host hands it to the others"""
        buf = (C.c_uint8 * 128)()
        check(_lib.load().dk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world_size):
        """collective: join this engine to the RCCL communicator identified by unique_id (dk_comm_init)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self.check(self._lib.dk_comm_init(self.handle, buf, rank, world_size))

    def comm_finalize(self):
        self.check(self._lib.dk_comm_finalize(self.handle))

    def or_reduce_slices(self, dst_ptr, src_ptr, n_slices, slice_bytes):
        """dst |= OR of n_slices slices at src (device pointers): local step of the OR-all-reduce."""
        self.check(self._lib.dk_or_reduce_slices(self.handle, C.c_void_p(dst_ptr), C.c_void_p(src_ptr),
                                                 n_slices, slice_bytes))

    def union_slices(self, dst_ptr, src_ptr, n_slices, slice_bytes, first_segment):
        """exact sets: insert the keys of n_slices table slices at src into the slice at dst (device
        pointers; all cover the 64-KiB segments from first_segment): local step of the union-all-reduce."""
        self.check(self._lib.dk_union_slices(self.handle, C.c_void_p(dst_ptr), C.c_void_p(src_ptr),
                                             n_slices, slice_bytes, first_segment))

    def close(self):
        if self._h:
            self._lib.dk_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_config(seed=20260313, genome_len=50_000, read_len=150, snv_rate=1e-3, denovo_rate=None,
                 err_rate=5e-3, n_rate=1e-4, xover_log2=20):
    """Synthetic-trio parameters; rates become 64-bit thresholds."""
    if denovo_rate is None:
        denovo_rate = 100.0 / (64 << 20)

    def thr(rate):
        return 0 if rate <= 0 else (2**64 - 1 if rate >= 1 else int(rate * 2.0**64))

    return DkSynthConfig(C.sizeof(DkSynthConfig), seed, genome_len, read_len, xover_log2,
                         thr(snv_rate), thr(denovo_rate), thr(err_rate), thr(n_rate))


class PinnedPacked:
    """Pinned host memory (dk_host_alloc) holding one packed batch -- bases words then mask words -- for
    ReadBatch.from_packed_async; `bases` / `mask` are numpy views of it"""

    def __init__(self, n_bases):
        self._lib = _lib.load()
        self.n_bwords, self.n_mwords = (n_bases + 31) // 32, (n_bases + 63) // 64
        self._p = C.c_void_p()
        check(self._lib.dk_host_alloc((self.n_bwords + self.n_mwords) * 8, C.byref(self._p)))
        buf = (C.c_uint64 * (self.n_bwords + self.n_mwords)).from_address(self._p.value)
        whole = np.frombuffer(buf, dtype=np.uint64)
        self.bases, self.mask = whole[:self.n_bwords], whole[self.n_bwords:]
        self.bases_ptr, self.mask_ptr = self._p.value, self._p.value + self.n_bwords * 8

    def close(self):
        if self._p:
            self.bases = self.mask = None
            self._lib.dk_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadBatch:
    """A device-resident packed read batch (dk_reads)."""

    def __init__(self, engine, handle, keepalive=None):
        self.engine = engine
        self._h = handle
        self._keep = keepalive

    @classmethod
    def from_sequences(cls, engine, reads):
        seq, offsets = _concat(reads)
        return cls.from_ascii(engine, seq, offsets)

    @classmethod
    def from_ascii(cls, engine, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        engine.check(engine._lib.dk_reads_from_ascii(engine.handle, _vp(seq), _vp(offsets),
                                                     len(offsets) - 1, C.byref(h)))
        return cls(engine, h)

    @classmethod
    def from_packed(cls, engine, bases, mask, n_bases, n_reads, n_windows):
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        mask = np.ascontiguousarray(mask, dtype=np.uint64)
        assert bases.size >= (n_bases + 31) // 32 and mask.size >= (n_bases + 63) // 64
        h = C.c_void_p()
        engine.check(engine._lib.dk_reads_from_packed(engine.handle, _vp(bases), _vp(mask), n_bases,
                                                      n_reads, n_windows, C.byref(h)))
        return cls(engine, h)

    @classmethod
    def from_packed_async(cls, engine, pinned, n_bases, n_reads, n_windows):
        """overlapped upload (dk_reads_from_packed_async) from a PinnedPacked host buffer, which must stay untouched
        until wait() returns; whatever consumes the batch waits for the copy on the device"""
        h = C.c_void_p()
        engine.check(engine._lib.dk_reads_from_packed_async(engine.handle, C.c_void_p(pinned.bases_ptr), C.c_void_p(pinned.mask_ptr),
                                                            n_bases, n_reads, n_windows, C.byref(h)))
        return cls(engine, h, pinned)

    def wait(self):
        """block until an asynchronous upload of this batch has finished (dk_reads_wait)"""
        self.engine.check(self.engine._lib.dk_reads_wait(self._h))

    @classmethod
    def attach_device(cls, engine, d_bases, d_mask, n_bases, n_reads, n_windows, keepalive=None):
        h = C.c_void_p()
        engine.check(engine._lib.dk_reads_attach_device(engine.handle, C.c_void_p(d_bases), C.c_void_p(d_mask),
                                                        n_bases, n_reads, n_windows, C.byref(h)))
        return cls(engine, h, keepalive)

    @classmethod
    def synth(cls, engine, cfg, sample, first_read, n_reads):
        h = C.c_void_p()
        engine.check(engine._lib.dk_reads_synth(engine.handle, C.byref(cfg), sample, first_read, n_reads, C.byref(h)))
        return cls(engine, h)

    def stats(self):
        st = DkStats()
        self.engine.check(self.engine._lib.dk_reads_stats(self._h, C.byref(st)))
        return st.as_dict()

    def download(self):
        n = self.stats()["n_bases"]
        bases = np.zeros((n + 31) // 32, dtype=np.uint64)
        mask = np.zeros((n + 63) // 64, dtype=np.uint64)
        self.engine.check(self.engine._lib.dk_reads_download(self._h, _vp(bases), _vp(mask)))
        return bases, mask, n

    def kmers(self, hashes=True, into=None):
        """kmer.rs stand-in: canonical k-mer (and hash) of every stream position (dk_reads_kmers).

        -> dict(lo, hi, hash, not_kmer, stats); arrays are indexed by stream position (read i starts at
        offsets[i] + i), not_kmer is the MSB-first bit mask of positions where no k-mer starts.
        into: optional dict of device pointers (ints) {"lo", "hi", "hash", "not_kmer"} to write to instead
        of host arrays (the caller owns n_bases * 8 bytes each, (n_bases + 63) // 64 * 8 for the mask)."""
        n = self.stats()["n_bases"]
        wide = self.engine.k > 32
        out = {}
        if into is None:
            out["lo"] = np.zeros(n, dtype=np.uint64)
            out["hi"] = np.zeros(n, dtype=np.uint64) if wide else None
            out["hash"] = np.zeros(n, dtype=np.uint64) if hashes else None
            out["not_kmer"] = np.zeros((n + 63) // 64, dtype=np.uint64)
            ptrs = [_vp(out[key]) if out[key] is not None else None for key in ("lo", "hi", "hash", "not_kmer")]
        else:
            ptrs = [C.c_void_p(into[key]) if into.get(key) else None for key in ("lo", "hi", "hash", "not_kmer")]
        ds = DkStats()
        self.engine.check(self.engine._lib.dk_reads_kmers(self.engine.handle, self._h, *ptrs, C.byref(ds)))
        out["stats"] = ds.as_dict()
        return out

    def close(self):
        if self._h:
            if self.engine._h:
                self.engine._lib.dk_reads_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_ascii_host(seq, offsets):
    """CPU packer of the C ABI (dk_pack_ascii_host): for hosts that pack while reading BAM."""
    lib = _lib.load()
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_reads = len(offsets) - 1
    total = int(offsets[-1]) + n_reads if n_reads else 0
    bases = np.zeros((total + 31) // 32 + 1, dtype=np.uint64)
    mask = np.zeros((total + 63) // 64 + 1, dtype=np.uint64)
    t = lib.dk_pack_ascii_host(_vp(seq), _vp(offsets), n_reads, _vp(bases), _vp(mask))
    assert t == total
    return bases[: (total + 31) // 32], mask[: (total + 63) // 64], total


class KmerCounts:
    """k-mer -> count table returned by the GPU (dk_result).  Unordered, like a HashMap."""

    def __init__(self, engine, handle, stats):
        self.engine = engine
        self._h = handle
        self.stats = stats
        self._host = None
        self._keep = None

    def __len__(self):
        n = C.c_uint64()
        self.engine.check(self.engine._lib.dk_result_size(self._h, C.byref(n)))
        return int(n.value)

    def to_host(self, sort=True):
        """-> (hi, lo, counts) numpy arrays; sorted by (hi, lo) when sort=True."""
        if self._host is None:
            n = len(self)
            lo = np.zeros(n, dtype=np.uint64)
            hi = np.zeros(n, dtype=np.uint64)
            cnt = np.zeros(n, dtype=np.uint32)
            if n:
                self.engine.check(self.engine._lib.dk_result_copy(self._h, _vp(lo), _vp(hi), _vp(cnt)))
            self._host = (hi, lo, cnt)
        hi, lo, cnt = self._host
        if sort and len(lo):
            order = np.lexsort((lo, hi))
            return hi[order], lo[order], cnt[order]
        return hi, lo, cnt

    @classmethod
    def from_device(cls, engine, lo_ptr, hi_ptr, cnt_ptr, n, keepalive=None):
        """wrap caller-owned device arrays (e.g. slices of gathered torch tensors) as a table (dk_result_attach)"""
        h = C.c_void_p()
        engine.check(engine._lib.dk_result_attach(engine.handle, C.c_void_p(lo_ptr), C.c_void_p(hi_ptr) if hi_ptr else None,
                                                  C.c_void_p(cnt_ptr), n, C.byref(h)))
        t = cls(engine, h, {})
        t._keep = keepalive
        return t

    def copy_to_device(self, lo_ptr, hi_ptr, cnt_ptr):
        """dense copy of the table into caller-owned device (or host) memory: len(self) entries per array"""
        self.engine.check(self.engine._lib.dk_result_copy(self._h, C.c_void_p(lo_ptr), C.c_void_p(hi_ptr) if hi_ptr else None,
                                                          C.c_void_p(cnt_ptr)))

    def device_view(self):
        plo, phi, pc, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.engine.check(self.engine._lib.dk_result_device_view(self._h, C.byref(plo), C.byref(phi), C.byref(pc), C.byref(n)))
        return plo.value, phi.value, pc.value, int(n.value)

    def as_dict(self):
        hi, lo, cnt = self.to_host(sort=False)
        k = self.engine.k
        return {kmer_to_str(h, l, k): int(c) for h, l, c in zip(hi, lo, cnt)}

    def close(self):
        if self._h:
            if self.engine._h:
                self.engine._lib.dk_result_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerSet:
    """Parent k-mer set resident in HBM (dk_set).  Mirrors counter.rs `KmerSet` (insert / contains).

    Engine(set_kind="bloom") (default): blocked Bloom filter -- no false negatives, false positives at
    the filter's rate.  Engine(set_kind="exact"): exact set (HashSet semantics)
    held as per-segment open-addressing tables in the same 2^filter_log2_bits bits; insert raises
    DkError(status 7) when a segment is full, popcount() is the number of k-mers held."""

    def __init__(self, engine, device_ptr=None, keepalive=None):
        self.engine = engine
        self._h = C.c_void_p()
        self._keep = keepalive
        if device_ptr is None:
            engine.check(engine._lib.dk_set_create(engine.handle, C.byref(self._h)))
        else:
            engine.check(engine._lib.dk_set_attach(engine.handle, C.c_void_p(device_ptr), C.byref(self._h)))
        self.n_bytes = (1 << engine.filter_log2_bits) // 8
        self.last_stats = None

    def insert_reads(self, batch):
        st = DkStats()
        self.engine.check(self.engine._lib.dk_set_insert(self._h, batch._h, C.byref(st)))
        self.last_stats = st.as_dict()
        return self.last_stats

    def insert_sequences(self, reads):
        b = ReadBatch.from_sequences(self.engine, reads)
        try:
            return self.insert_reads(b)
        finally:
            b.close()

    def contains(self, kmers):
        """kmers: iterable of str, or (hi, lo) uint64 arrays.  -> bool array"""
        if isinstance(kmers, tuple):
            hi = np.ascontiguousarray(kmers[0], dtype=np.uint64)
            lo = np.ascontiguousarray(kmers[1], dtype=np.uint64)
        else:
            pairs = [kmer_from_str(s) for s in kmers]
            hi = np.array([p[0] for p in pairs], dtype=np.uint64)
            lo = np.array([p[1] for p in pairs], dtype=np.uint64)
        out = np.zeros(len(lo), dtype=np.uint8)
        self.engine.check(self.engine._lib.dk_set_contains(self._h, _vp(lo), _vp(hi) if self.engine.k > 32 else None,
                                                           len(lo), _vp(out)))
        return out.astype(bool)

    def clear(self):
        self.engine.check(self.engine._lib.dk_set_clear(self._h))

    def allreduce_or(self):
        """collective over the engine's communicator (Engine.comm_init): combine the ranks' sets in place -- OR for a
        Bloom filter, union for an exact set (dk_set_allreduce_or).  -> bytes sent by this rank"""
        n = C.c_uint64()
        self.engine.check(self.engine._lib.dk_set_allreduce_or(self._h, C.byref(n)))
        return int(n.value)

    def popcount(self):
        n = C.c_uint64()
        self.engine.check(self.engine._lib.dk_set_popcount(self._h, C.byref(n)))
        return int(n.value)

    def save(self, path):
        """write the filter (64-byte geometry header + bits) to `path`"""
        self.engine.check(self.engine._lib.dk_set_save(self._h, str(path).encode()))

    def load(self, path):
        """read a filter written by save(); the file's geometry must match the engine's"""
        self.engine.check(self.engine._lib.dk_set_load(self._h, str(path).encode()))

    @property
    def device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self.engine.check(self.engine._lib.dk_set_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value

    def to_host(self):
        words = np.zeros(self.n_bytes // 8, dtype=np.uint64)
        self.engine.check(self.engine._lib.dk_set_download(self._h, _vp(words)))
        return words

    def from_host(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.size == self.n_bytes // 8
        self.engine.check(self.engine._lib.dk_set_upload(self._h, _vp(words)))

    def close(self):
        if self._h:
            if self.engine._h:
                self.engine._lib.dk_set_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChildAccumulator:
    """Child-only k-mers of a sample that arrives in many batches (dk_accum): the absent occurrences of every
    batch are kept on the GPU and counted once, so counts and `min_count` are exact over the whole sample.

    window_count > 1: the sample is streamed window_count times, pass w (`reset(w)`) keeping the k-mers whose hash
    lies in the w-th of window_count equal ranges -- for samples whose absent occurrences do not fit in HBM beside
    the set.  parents=None counts every k-mer (KmerCounter over batches)."""

    def __init__(self, engine, parents, capacity_records, window_index=0, window_count=1):
        self.engine = engine
        self._h = C.c_void_p()
        self._parents = parents
        self.window_count = window_count
        engine.check(engine._lib.dk_accum_create(engine.handle, parents._h if parents is not None else None,
                                                 window_index, window_count, int(capacity_records), C.byref(self._h)))

    def add(self, batch):
        st = DkStats()
        self.engine.check(self.engine._lib.dk_accum_add(self._h, batch._h, C.byref(st)))
        return st.as_dict()

    def finish(self, min_count=1):
        h = C.c_void_p()
        st = DkStats()
        self.engine.check(self.engine._lib.dk_accum_finish(self._h, min_count, C.byref(h), C.byref(st)))
        return KmerCounts(self.engine, h, st.as_dict())

    def reset(self, window_index=0):
        self.engine.check(self.engine._lib.dk_accum_reset(self._h, window_index))

    def geometry(self):
        """-> (units of the window, records per unit, bytes per record)"""
        n, cap, rb = C.c_uint64(), C.c_uint32(), C.c_uint32()
        self.engine.check(self.engine._lib.dk_accum_geometry(self._h, C.byref(n), C.byref(cap), C.byref(rb)))
        return int(n.value), int(cap.value), int(rb.value)

    def device_view(self):
        """-> (store pointer, fill pointer, overflow-list pointer, overflow entries); synchronises the engine"""
        st, fl, ov, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.engine.check(self.engine._lib.dk_accum_device_view(self._h, C.byref(st), C.byref(fl), C.byref(ov), C.byref(n)))
        return st.value, fl.value, ov.value, int(n.value)

    def finish_pieces(self, stores_ptr, fills_ptr, n_pieces, first_unit, n_units, extra_ptr=0, n_extra=0, min_count=1, keepalive=None):
        """multi-GPU: count units [first_unit, first_unit + n_units) from the piece-major slices the ranks exchanged"""
        h = C.c_void_p()
        st = DkStats()
        self.engine.check(self.engine._lib.dk_accum_finish_pieces(
            self._h, C.c_void_p(stores_ptr), C.c_void_p(fills_ptr), n_pieces, first_unit, n_units,
            C.c_void_p(extra_ptr) if extra_ptr else None, n_extra, min_count, C.byref(h), C.byref(st)))
        res = KmerCounts(self.engine, h, st.as_dict())
        res._keep = keepalive            # (the result does not reference the pieces, but callers may want them to live as long)
        return res

    def exchange_finish(self, min_count=1):
        """multi-GPU, collective (dk_accum_exchange_finish): the ranks swap unit ranges on the engine's communicator, in
        place, and each counts its share of the hash space; the accumulator is consumed (reset before reuse).
        -> KmerCounts; .bytes_sent = what this rank sent"""
        h = C.c_void_p()
        st = DkStats()
        sent = C.c_uint64()
        self.engine.check(self.engine._lib.dk_accum_exchange_finish(self._h, min_count, C.byref(h), C.byref(st), C.byref(sent)))
        res = KmerCounts(self.engine, h, st.as_dict())
        res.bytes_sent = int(sent.value)
        return res

    def stats(self):
        st = DkStats()
        self.engine.check(self.engine._lib.dk_accum_stats(self._h, C.byref(st)))
        return st.as_dict()

    def device_bytes(self):
        n = C.c_uint64()
        self.engine.check(self.engine._lib.dk_accum_device_bytes(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._h:
            if self.engine._h:
                self.engine._lib.dk_accum_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerCounter:
    """Mirrors counter.rs `KmerCounter`: per-k-mer occurrence counts of a sample, and the
    child-only ("de-novo") pass against a parent `KmerSet`."""

    def __init__(self, engine):
        self.engine = engine

    def _probe(self, kset, batch):
        h = C.c_void_p()
        st = DkStats()
        e = self.engine
        e.check(e._lib.dk_probe(e.handle, kset._h if kset is not None else None, batch._h, C.byref(h), C.byref(st)))
        return KmerCounts(e, h, st.as_dict())

    def count_reads(self, batch):
        """all canonical k-mers of the batch with their counts"""
        return self._probe(None, batch)

    def count_sequences(self, reads):
        b = ReadBatch.from_sequences(self.engine, reads)
        try:
            return self.count_reads(b)
        finally:
            b.close()

    def child_only(self, batch, parents):
        """k-mers of the (child) batch absent from the parent set, with their child counts"""
        return self._probe(parents, batch)

    def merge(self, tables, min_count=1):
        """sum several KmerCounts by k-mer (batches of one sample, or shards); the inputs should come
        from an engine with min_count=1, the threshold is applied to the summed counts here"""
        e = self.engine
        arr = (C.c_void_p * len(tables))(*[t._h for t in tables])
        h = C.c_void_p()
        st = DkStats()
        e.check(e._lib.dk_result_merge(e.handle, arr, len(tables), min_count, C.byref(h), C.byref(st)))
        return KmerCounts(e, h, st.as_dict())
