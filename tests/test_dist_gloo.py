"""CPU tests of the multi-GPU host logic with world_size 2 over gloo: the composed OR-all-reduce,
read sharding, and the count merge.  The local OR step is injected (torch on CPU here, the HIP
kernel dk_or_reduce_slices on the GPU box); k-mer work is done by the oracle as the checker."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, related_trio
from denovo_kmer_amd.dist import merge_counts, or_allreduce_, shard_range
from oracle import orc


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def cpu_or_fn(dst, src, n_slices):
    sl = dst.numel()
    for j in range(n_slices):
        dst |= src[j * sl:(j + 1) * sl]


def _worker(rank, world, port, parents, child, k, log2_bits, nh, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. raw collective: random words
        g = torch.Generator().manual_seed(100 + rank)
        t = torch.randint(-2**62, 2**62, (4096,), generator=g, dtype=torch.int64)
        mine = t.clone()
        sent = or_allreduce_(t, cpu_or_fn)
        alls = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(alls, mine)
        exp = alls[0].clone()
        for a in alls[1:]:
            exp |= a
        ok_collective = bool(torch.equal(t, exp)) and sent == 2 * (world - 1) * (4096 // world) * 8

        # 2. sharded trio: partial parent filters -> OR-all-reduce -> local child probe -> merge
        plo, phi = shard_range(len(parents), rank, world)
        clo, chi = shard_range(len(child), rank, world)
        pseq, poff = orc.concat_reads(parents[plo:phi])
        f = orc.new_filter(log2_bits)
        orc.bloom_insert(f, log2_bits, nh, seed, k, True, pseq, poff)
        ft = torch.from_numpy(f.view(np.int64))
        or_allreduce_(ft, cpu_or_fn)
        cseq, coff = orc.concat_reads(child[clo:chi])
        km, cn, _ = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff, 1)
        mhi, mlo, mcn = merge_counts(km["hi"], km["lo"], cn, min_count=2)
        q.put((rank, ok_collective, f.copy(), mhi, mlo, mcn))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_or_allreduce_and_sharded_trio_world2():
    rng = np.random.default_rng(7)
    parents, child = related_trio(rng, genome_len=2000, n_reads=60, read_len=100)
    child = child + child[:15]
    k, log2_bits, nh, seed, world = 25, 20, 3, 99, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, parents, child, k, log2_bits, nh, seed, q))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # whole-input oracle
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(log2_bits)
    orc.bloom_insert(f, log2_bits, nh, seed, k, True, pseq, poff)
    km, cn, _ = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff, 2)
    assert len(km) > 0
    for rank, ok_collective, filt, mhi, mlo, mcn in outs:
        assert ok_collective
        assert np.array_equal(filt, f)                       # union of shards == whole
        assert np.array_equal(mhi, km["hi"]) and np.array_equal(mlo, km["lo"]) and np.array_equal(mcn, cn)


# ---- exact sets: the same three collective steps with a key union as the local reduction ------------
SEG_WORDS = 8192            # one 64-KiB table segment = 8192 u64 slots (k <= 32)


def _cpu_union_fn(T):
    """CPU stand-in for dk_union_slices (k <= 32): slot values whose top T bits equal the segment index
    are keys, anything else is an empty slot; keys of the source slices go into free slots of dst."""
    def fn(dst, src, n_slices):
        sl = dst.numel()
        assert sl % SEG_WORDS == 0
        first_segment = dist.get_rank() * (sl // SEG_WORDS)
        d = dst.numpy().view(np.uint64)
        s = src.numpy().view(np.uint64)
        for seg in range(sl // SEG_WORDS):
            prefix = np.uint64(first_segment + seg)
            dseg = d[seg * SEG_WORDS:(seg + 1) * SEG_WORDS]
            have = set(int(x) for x in dseg[(dseg >> np.uint64(64 - T)) == prefix])
            free = [i for i in range(SEG_WORDS) if (int(dseg[i]) >> (64 - T)) != int(prefix)]
            for j in range(n_slices):
                sseg = s[j * sl + seg * SEG_WORDS:j * sl + (seg + 1) * SEG_WORDS]
                for x in sseg[(sseg >> np.uint64(64 - T)) == prefix]:
                    if int(x) not in have:
                        have.add(int(x))
                        dseg[free.pop(0)] = x
    return fn


def _union_worker(rank, world, port, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_seg = 1 << T
        rng = np.random.default_rng(500 + rank)
        table = np.zeros(n_seg * SEG_WORDS, dtype=np.uint64)
        keys = []
        for seg in range(n_seg):
            empty = np.uint64((seg ^ 1) << (64 - T))
            table[seg * SEG_WORDS:(seg + 1) * SEG_WORDS] = empty
            low = rng.choice(1 << 20, size=300, replace=False).astype(np.uint64) * np.uint64(rank % 2 + 1)   # overlapping key sets
            ks = (np.uint64(seg) << np.uint64(64 - T)) | low
            slots = rng.choice(SEG_WORDS, size=len(ks), replace=False)
            table[seg * SEG_WORDS + slots] = ks
            keys.append(ks)
        t = torch.from_numpy(table.view(np.int64))
        or_allreduce_(t, _cpu_union_fn(T))
        q.put((rank, table.copy(), np.concatenate(keys)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_union_allreduce_of_exact_tables_world2():
    world, T = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_union_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=200) for _ in range(world)), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.unique(np.concatenate([o[2] for o in outs]))
    assert np.array_equal(outs[0][1], outs[1][1])                 # every rank ends with the same table
    table = outs[0][1]
    seg_of_slot = np.arange(len(table)) // SEG_WORDS
    is_key = (table >> np.uint64(64 - T)) == seg_of_slot.astype(np.uint64)
    assert np.array_equal(np.sort(table[is_key]), want)           # exactly the union, each key once


def test_local_reduce_fn_follows_the_set_kind():
    from denovo_kmer_amd import dist as dkdist

    class FakeEngine:
        set_kind = "bloom"
    assert dkdist.local_reduce_fn(FakeEngine()).__qualname__.startswith("hip_or_fn")
    FakeEngine.set_kind = "exact"
    assert dkdist.local_reduce_fn(FakeEngine()).__qualname__.startswith("hip_union_fn")
