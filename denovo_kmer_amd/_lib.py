"""ctypes binding of libdenovo_kmer.so (include/denovo_kmer.h).

The library is the product: there is no Python or CPU stand-in.  `load()` raises if the shared
object is missing, and every engine call raises `DkError` if the HIP side fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DK_LIB_PATH: load another build of the same library (e.g. the DK_STAMPS diagnostic build)
LIB_PATH = os.environ.get("DK_LIB_PATH") or os.path.join(_HERE, "libdenovo_kmer.so")

DK_OK = 0
DK_ERR_INVALID_ARG = 1
DK_ERR_NO_DEVICE = 2
DK_ERR_HIP = 3
DK_ERR_OOM = 4
DK_ERR_UNSUPPORTED = 5
DK_ERR_OVERFLOW = 6

MODE_AUTO, MODE_DIRECT, MODE_BUCKETED = 0, 1, 2
SET_BLOOM, SET_EXACT = 0, 1
ERR_SET_FULL = 7
MAX_STAGES = 12
ABI_VERSION = 3


class DkError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"denovo_kmer status {status}: {message}")
        self.status = status


class DkConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("k", C.c_uint32), ("canonical", C.c_uint32),
                ("filter_log2_bits", C.c_uint32), ("n_hashes", C.c_uint32), ("seed", C.c_uint64),
                ("min_count", C.c_uint32), ("device_id", C.c_int32), ("rank", C.c_uint32),
                ("world_size", C.c_uint32), ("mode", C.c_uint32), ("set_kind", C.c_uint32),
                ("stream", C.c_void_p)]


class DkStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_windows", C.c_uint64),
                ("n_valid", C.c_uint64), ("n_absent", C.c_uint64), ("n_distinct", C.c_uint64),
                ("n_emitted", C.c_uint64)]

    def as_dict(self):
        return {f: int(getattr(self, f)) for f, _ in self._fields_}


class DkTimings(C.Structure):
    _fields_ = [("n_stages", C.c_uint32), ("total_ms", C.c_float),
                ("stage_ms", C.c_float * MAX_STAGES), ("stage_name", (C.c_char * 24) * MAX_STAGES)]

    def as_dict(self):
        d = {"total_ms": float(self.total_ms), "stages": []}
        for i in range(self.n_stages):
            d["stages"].append((bytes(self.stage_name[i]).split(b"\0")[0].decode(), float(self.stage_ms[i])))
        return d


class DkSynthConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint64), ("seed", C.c_uint64), ("genome_len", C.c_uint64),
                ("read_len", C.c_uint32), ("xover_log2", C.c_uint32), ("snv_thr", C.c_uint64),
                ("denovo_thr", C.c_uint64), ("err_thr", C.c_uint64), ("n_thr", C.c_uint64)]


# every symbol include/denovo_kmer.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_U64 = C.c_uint64
_PU64 = C.POINTER(C.c_uint64)
SYMBOLS = {
    "dk_abi_version": (C.c_int32, []),
    "dk_status_string": (C.c_char_p, [C.c_int32]),
    "dk_engine_create": (C.c_int32, [C.POINTER(DkConfig), _PP]),
    "dk_engine_destroy": (None, [_P]),
    "dk_last_error": (C.c_char_p, [_P]),
    "dk_engine_synchronize": (C.c_int32, [_P]),
    "dk_engine_timings": (C.c_int32, [_P, C.POINTER(DkTimings)]),
    "dk_engine_config": (C.c_int32, [_P, C.POINTER(DkConfig)]),
    "dk_engine_set_option": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "dk_engine_trim": (C.c_int32, [_P, _PU64]),
    "dk_engine_get_info": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "dk_engine_reserve": (C.c_int32, [_P, _U64]),
    "dk_reads_from_packed_async": (C.c_int32, [_P, _P, _P, _U64, _U64, _U64, _PP]),
    "dk_reads_wait": (C.c_int32, [_P]),
    "dk_host_alloc": (C.c_int32, [_U64, _PP]),
    "dk_host_free": (None, [_P]),
    "dk_reads_from_ascii": (C.c_int32, [_P, _P, _P, _U64, _PP]),
    "dk_reads_from_packed": (C.c_int32, [_P, _P, _P, _U64, _U64, _U64, _PP]),
    "dk_reads_attach_device": (C.c_int32, [_P, _P, _P, _U64, _U64, _U64, _PP]),
    "dk_reads_synth": (C.c_int32, [_P, C.POINTER(DkSynthConfig), C.c_int32, _U64, _U64, _PP]),
    "dk_reads_stats": (C.c_int32, [_P, C.POINTER(DkStats)]),
    "dk_reads_download": (C.c_int32, [_P, _P, _P]),
    "dk_reads_kmers": (C.c_int32, [_P, _P, _P, _P, _P, _P, C.POINTER(DkStats)]),
    "dk_reads_destroy": (None, [_P]),
    "dk_pack_ascii_host": (_U64, [_P, _P, _U64, _P, _P]),
    "dk_set_create": (C.c_int32, [_P, _PP]),
    "dk_set_attach": (C.c_int32, [_P, _P, _PP]),
    "dk_set_clear": (C.c_int32, [_P]),
    "dk_set_insert": (C.c_int32, [_P, _P, C.POINTER(DkStats)]),
    "dk_set_contains": (C.c_int32, [_P, _P, _P, _U64, _P]),
    "dk_set_device_ptr": (C.c_int32, [_P, _PP, _PU64]),
    "dk_set_download": (C.c_int32, [_P, _P]),
    "dk_set_upload": (C.c_int32, [_P, _P]),
    "dk_set_popcount": (C.c_int32, [_P, _PU64]),
    "dk_set_save": (C.c_int32, [_P, C.c_char_p]),
    "dk_set_load": (C.c_int32, [_P, C.c_char_p]),
    "dk_or_reduce_slices": (C.c_int32, [_P, _P, _P, _U64, _U64]),
    "dk_union_slices": (C.c_int32, [_P, _P, _P, _U64, _U64, _U64]),
    "dk_set_destroy": (None, [_P]),
    "dk_comm_unique_id": (C.c_int32, [_P]),
    "dk_comm_init": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32]),
    "dk_comm_finalize": (C.c_int32, [_P]),
    "dk_set_allreduce_or": (C.c_int32, [_P, _PU64]),
    "dk_comm_layout": (C.c_int32, [_U64, _U64, C.c_uint32, C.c_uint32, _U64, _PU64, _PU64]),
    "dk_probe": (C.c_int32, [_P, _P, _P, _PP, C.POINTER(DkStats)]),
    "dk_result_size": (C.c_int32, [_P, _PU64]),
    "dk_result_copy": (C.c_int32, [_P, _P, _P, _P]),
    "dk_result_device_view": (C.c_int32, [_P, _PP, _PP, _PP, _PU64]),
    "dk_result_merge": (C.c_int32, [_P, _PP, C.c_uint32, C.c_uint32, _PP, C.POINTER(DkStats)]),
    "dk_result_attach": (C.c_int32, [_P, _P, _P, _P, _U64, _PP]),
    "dk_result_destroy": (None, [_P]),
    "dk_accum_create": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _U64, _PP]),
    "dk_accum_add": (C.c_int32, [_P, _P, C.POINTER(DkStats)]),
    "dk_accum_finish": (C.c_int32, [_P, C.c_uint32, _PP, C.POINTER(DkStats)]),
    "dk_accum_reset": (C.c_int32, [_P, C.c_uint32]),
    "dk_accum_stats": (C.c_int32, [_P, C.POINTER(DkStats)]),
    "dk_accum_geometry": (C.c_int32, [_P, _PU64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "dk_accum_device_view": (C.c_int32, [_P, _PP, _PP, _PP, _PU64]),
    "dk_accum_finish_pieces": (C.c_int32, [_P, _P, _P, C.c_uint32, _U64, _U64, _P, _U64, C.c_uint32, _PP, C.POINTER(DkStats)]),
    "dk_accum_exchange_finish": (C.c_int32, [_P, C.c_uint32, _PP, C.POINTER(DkStats), _PU64]),
    "dk_accum_device_bytes": (C.c_int32, [_P, _PU64]),
    "dk_accum_destroy": (None, [_P]),
}

_lib = None


def load():
    """Load libdenovo_kmer.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C denovo_kmer_amd/csrc` "
            "(or __graft_entry__.build()); there is no pure-Python path")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dk_abi_version() != ABI_VERSION:
        raise ImportError(f"libdenovo_kmer ABI {lib.dk_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, engine_handle=None):
    if status != DK_OK:
        lib = load()
        msg = lib.dk_last_error(engine_handle)
        msg = msg.decode() if msg else ""
        name = lib.dk_status_string(status).decode()
        raise DkError(status, f"{name}: {msg}")
