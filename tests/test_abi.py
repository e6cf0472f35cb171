"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/denovo_kmer.h declares, refuses to run without a GPU (no CPU fallback), and its one
host-only utility (the packer) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, random_reads
from denovo_kmer_amd import _lib
from oracle import orc

HEADER = os.path.join(ROOT, "include", "denovo_kmer.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_integration_md_rust_block_matches_the_header():
    """INTEGRATION.md's `extern "C"` block is generated from the header (tools/gen_rust_extern.py); a Rust toolchain
    is absent, so this comparison -- every function, every argument name and type -- is the guard against drift"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_extern", os.path.join(ROOT, "tools", "gen_rust_extern.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = md.index('extern "C" {')
    block = md[a:md.index("\n}\n", a) + 2]
    assert block.split() == gen.rust_block().split()
    names = re.findall(r"pub fn (dk_[a-z0-9_]+)\(", block)
    assert sorted(names) == declared_symbols() == sorted(_lib.SYMBOLS)
    # arity of the ctypes binding against the header
    for name, ret, params in gen.declarations():
        assert len(_lib.SYMBOLS[name][1]) == len(params), name


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.dk_abi_version() == _lib.ABI_VERSION
    assert lib.dk_status_string(0) == b"ok"


def test_struct_sizes_match_header():
    # field-by-field layout is plain C; these sizes are what a Rust #[repr(C)] mirror must have
    assert C.sizeof(_lib.DkConfig) == 64
    assert C.sizeof(_lib.DkStats) == 56
    assert C.sizeof(_lib.DkSynthConfig) == 64
    assert C.sizeof(_lib.DkTimings) == 8 + 4 * 12 + 24 * 12


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_create_fails_loudly_without_gpu():
    from denovo_kmer_amd import DkError, Engine
    with pytest.raises(DkError) as ei:
        Engine(k=31)
    assert ei.value.status == _lib.DK_ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value)


def test_engine_create_rejects_bad_config():
    lib = _lib.load()
    h = C.c_void_p()
    for field, value in (("k", 0), ("k", 65), ("filter_log2_bits", 19), ("filter_log2_bits", 41),
                         ("n_hashes", 0), ("n_hashes", 17), ("min_count", 0), ("mode", 9), ("set_kind", 2),
                         ("struct_size", 8)):
        cfg = _lib.DkConfig(C.sizeof(_lib.DkConfig), 31, 1, 24, 4, 1, 1, 0, 0, 1, 0, 0, None)
        setattr(cfg, field, value)
        assert lib.dk_engine_create(C.byref(cfg), C.byref(h)) == _lib.DK_ERR_INVALID_ARG, field
        assert not h.value
        assert lib.dk_last_error(None)
    assert lib.dk_engine_create(None, C.byref(h)) == _lib.DK_ERR_INVALID_ARG


def test_host_packer_matches_oracle(rng):
    from denovo_kmer_amd import pack_ascii_host
    reads = random_reads(rng, 50, 0, 200, n_rate=0.03, lower_rate=0.2) + ["", "N", "acgt", ""]
    seq, off = orc.concat_reads(reads)
    b, m, n = pack_ascii_host(seq, off)
    ob, om, on = orc.pack_reads(seq, off)
    assert n == on and np.array_equal(b, ob) and np.array_equal(m, om)


def test_product_code_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under denovo_kmer_amd/ or include/ may reference it
    bad = []
    for base in ("denovo_kmer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|dk_oracle|orc_", text):
                        bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


@pytest.mark.gpu
def test_malformed_read_batches_are_rejected():
    """ASCII offsets must start at 0; a packed stream (uploaded or attached) must end with a flagged separator
    position -- the kernels judge a window that would run past the end by its mask flags"""
    import denovo_kmer_amd as d
    from denovo_kmer_amd.api import pack_ascii_host
    with d.Engine(k=21, filter_log2_bits=22) as eng:
        seq = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTAAAA", dtype=np.uint8).copy()
        with pytest.raises(d.DkError) as ei:
            d.ReadBatch.from_ascii(eng, seq, np.array([4, 28], dtype=np.uint64))
        assert ei.value.status == _lib.DK_ERR_INVALID_ARG and "offsets[0]" in str(ei.value)
        with pytest.raises(d.DkError):
            d.ReadBatch.from_ascii(eng, seq, np.array([0, 20, 10], dtype=np.uint64))        # not monotonic
        bases, mask, n = pack_ascii_host(seq, np.array([0, 28], dtype=np.uint64))
        ok = d.ReadBatch.from_packed(eng, bases, mask, n, 1, 8)
        assert ok.stats()["n_bases"] == 29
        bad_mask = mask.copy()
        bad_mask[(n - 1) >> 6] &= ~np.uint64(1 << (63 - ((n - 1) & 63)))                     # clear the final separator flag
        with pytest.raises(d.DkError) as ei:
            d.ReadBatch.from_packed(eng, bases, bad_mask, n, 1, 8)
        assert "separator" in str(ei.value)
        tb = torch.from_numpy(bases.view(np.int64)).cuda()
        tm = torch.from_numpy(bad_mask.view(np.int64)).cuda()
        with pytest.raises(d.DkError) as ei:
            d.ReadBatch.attach_device(eng, tb.data_ptr(), tm.data_ptr(), n, 1, 8, keepalive=(tb, tm))
        assert "separator" in str(ei.value)
        tm2 = torch.from_numpy(mask.view(np.int64)).cuda()
        att = d.ReadBatch.attach_device(eng, tb.data_ptr(), tm2.data_ptr(), n, 1, 8, keepalive=(tb, tm2))
        ks = d.KmerSet(eng)
        assert ks.insert_reads(att)["n_valid"] == 8


def test_exchange_layout_arithmetic_for_two_to_eight_ranks():
    """dk_comm_layout is the host-side arithmetic of the native piece-wise exchanges (dk_set_allreduce_or,
    dk_accum_exchange_finish): for every rank of P = 1..8 the peers' pieces must fill distinct staging slots 0..P-2 in
    rank order (the order torch's all_to_all_single delivers them in, own slice left out), a piece must be a whole number
    of granules that fits the staging buffer P - 1 times, and the pieces must tile the slice."""
    lib = _lib.load()
    for world in range(1, 9):
        for rank in range(world):
            for staging, sl, gran in ((1 << 30, 8 << 30, 65536), (1 << 20, 12345 * 6, 256), (4096, 100, 4), (1 << 30, 1 << 20, 65536)):
                piece = C.c_uint64()
                slots = (C.c_uint64 * world)()
                assert lib.dk_comm_layout(staging, sl, rank, world, gran, C.byref(piece), slots) == 0
                peers = [q for q in range(world) if q != rank] or [0]
                got = [int(slots[q]) for q in peers]
                assert got == list(range(len(peers))), (world, rank, got)
                if world > 1:
                    assert int(slots[rank]) == 2**64 - 1
                p = int(piece.value)
                assert p > 0 and p * len(peers) <= staging and (p % gran == 0 or p == sl) and p <= sl
                # the pieces tile the slice: ceil(sl / p) groups, the last one shorter
                assert sum(min(p, sl - off) for off in range(0, sl, p)) == sl
    assert lib.dk_comm_layout(1 << 20, 1 << 20, 3, 2, 4, C.byref(piece), slots) != 0     # rank outside the world


def test_a_named_rccl_library_that_does_not_load_is_an_error_not_a_fallback():
    """DK_RCCL_LIBRARY replaces the search for librccl (the multi-rank GPU tests point it at tests/rccl_shim): a wrong path
    must surface as DK_ERR_UNSUPPORTED from dk_comm_unique_id rather than silently picking the system's library"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import denovo_kmer_amd as d\n"
            "try:\n"
            "    d.Engine.comm_unique_id()\n"
            "except d.DkError as exc:\n"
            "    print('refused:', exc)\n"
            "else:\n"
            "    print('loaded')\n" % ROOT)
    env = dict(os.environ, DK_RCCL_LIBRARY="/nonexistent/librccl_shim.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "refused:" in out.stdout and "cannot load librccl" in out.stdout
