"""Multi-rank GPU path rehearsed on ONE GPU: two processes (gloo rendezvous, collectives staged
through the host because gloo has no device all-to-all) each build a partial parent filter with the
HIP kernels, OR-all-reduce it with the HIP slice-OR kernel, probe their child shard, and merge the
counts.  The result must equal the oracle on the whole input.  On an 8-GPU node the same code runs
with backend nccl (RCCL) and no host staging -- that run belongs to the driver."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import orc

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n_reads, k, log2_bits, nh, seed, mode, set_kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import denovo_kmer_amd as dk
    from denovo_kmer_amd.dist import local_reduce_fn, merge_counts, merge_counts_device, or_allreduce_, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        gcfg = dk.synth_config(genome_len=100_000)
        eng = dk.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=seed, device_id=0, mode=mode,
                        rank=rank, world_size=world, set_kind=set_kind)
        filt = torch.zeros((1 << log2_bits) // 64, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        ks = dk.KmerSet(eng, device_ptr=filt.data_ptr(), keepalive=filt)
        if set_kind == "exact":
            ks.clear()                       # an empty exact set is not all-zero memory
        lo, hi = shard_range(n_reads, rank, world)
        for s in (0, 1):
            ks.insert_reads(dk.ReadBatch.synth(eng, gcfg, s, lo, hi - lo))
        or_allreduce_(filt, local_reduce_fn(eng), stage_through_cpu=True)
        n_keys = ks.popcount()
        res = dk.KmerCounter(eng).child_only(dk.ReadBatch.synth(eng, gcfg, 2, lo, hi - lo), ks)
        khi, klo, kcnt = res.to_host()
        mhi, mlo, mcnt = merge_counts(khi, klo, kcnt, min_count=1)
        # the device-side merge (gathered tensors attached as tables, summed on the GPU) must agree with it
        dm = merge_counts_device(eng, res, min_count=1, stage_through_cpu=True)
        dhi, dlo, dcnt = dm.to_host(sort=True)
        assert np.array_equal(dhi, mhi) and np.array_equal(dlo, mlo) and np.array_equal(dcnt, mcnt)
        dm2 = merge_counts_device(eng, res, min_count=2, stage_through_cpu=True)
        keep = mcnt >= 2
        d2 = dm2.to_host(sort=True)
        assert np.array_equal(d2[1], mlo[keep]) and np.array_equal(d2[2], mcnt[keep])
        q.put((rank, filt.cpu().numpy().view(np.uint64).copy(), mhi, mlo, mcnt, n_keys))
        eng.close()
    finally:
        dist.destroy_process_group()


def _run(mode, set_kind, n_reads, k, log2_bits, nh, seed, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, k, log2_bits, nh, seed, mode, set_kind, q))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return outs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["direct", "bucketed"])
def test_two_ranks_one_gpu_match_whole_input_oracle(mode):
    n_reads, k, log2_bits, nh, seed, world = 6000, 31, 24, 4, 31337, 2
    outs = _run(mode, "bloom", n_reads, k, log2_bits, nh, seed, world)
    ocfg = orc.synth_cfg(genome_len=100_000)
    f = orc.new_filter(log2_bits)
    for smp in (0, 1):
        seq, off = orc.synth_reads(ocfg, smp, 0, n_reads)
        orc.bloom_insert(f, log2_bits, nh, seed, k, True, seq, off)
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
    km, cn, _ = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff)
    for rank, filt, mhi, mlo, mcnt, _ in outs:
        assert np.array_equal(filt, f)
        assert np.array_equal(mhi, km["hi"]) and np.array_equal(mlo, km["lo"]) and np.array_equal(mcnt, cn)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,k", [("bucketed", 31), ("direct", 41)])
def test_two_ranks_exact_set_union_matches_whole_input_oracle(mode, k):
    """exact sets: the per-rank tables are united segment by segment (dk_union_slices) between the
    all-to-all and the all-gather; every rank ends with the same table and the exact child-only set"""
    n_reads, log2_bits, seed, world = 6000, 27, 31337, 2
    outs = _run(mode, "exact", n_reads, k, log2_bits, 4, seed, world)
    ocfg = orc.synth_cfg(genome_len=100_000)
    pseq, poff = orc.synth_reads(ocfg, 0, 0, n_reads)
    p1seq, p1off = orc.synth_reads(ocfg, 1, 0, n_reads)
    allseq = np.concatenate([pseq, p1seq])
    alloff = np.concatenate([poff, p1off[1:] + poff[-1]])
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
    km, cn, _ = orc.exact_child_only(k, True, allseq, alloff, cseq, coff)
    pk, _, _ = orc.count_reads(k, True, allseq, alloff)
    assert np.array_equal(outs[0][1], outs[1][1])                 # same table on both ranks
    for rank, table, mhi, mlo, mcnt, n_keys in outs:
        assert n_keys == len(pk)
        assert np.array_equal(mhi, km["hi"]) and np.array_equal(mlo, km["lo"]) and np.array_equal(mcnt, cn)


def _accum_worker(rank, world, port, n_reads, k, log2_bits, nh, seed, mode, cap, q, opts=()):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import denovo_kmer_amd as dk
    from denovo_kmer_amd.dist import accum_exchange_finish, local_reduce_fn, or_allreduce_, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        gcfg = dk.synth_config(genome_len=100_000)
        eng = dk.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=nh, seed=seed, device_id=0, mode=mode, rank=rank, world_size=world)
        for name, val in opts:
            eng.set_option(name, val)
        filt = torch.zeros((1 << log2_bits) // 64, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        ks = dk.KmerSet(eng, device_ptr=filt.data_ptr(), keepalive=filt)
        lo, hi = shard_range(n_reads, rank, world)
        for s in (0, 1):
            ks.insert_reads(dk.ReadBatch.synth(eng, gcfg, s, lo, hi - lo))
        or_allreduce_(filt, local_reduce_fn(eng), stage_through_cpu=True)
        out = {}
        if opts:
            probe = dk.ChildAccumulator(eng, ks, capacity_records=cap)
            assert probe.geometry()[2] == (6 if dict(opts).get("accum_min_u", 0) >= 8 and k <= 32 else 16 if k > 32 else 8)
            probe.close()
        for windows in (1, 2):
            acc = dk.ChildAccumulator(eng, ks, capacity_records=cap, window_count=windows)
            for mc in (1, 2):
                parts = []
                for w in range(windows):
                    acc.reset(w)
                    half = (hi - lo) // 2                     # the rank's shard in two batches
                    acc.add(dk.ReadBatch.synth(eng, gcfg, 2, lo, half))
                    acc.add(dk.ReadBatch.synth(eng, gcfg, 2, lo + half, hi - lo - half))
                    res = accum_exchange_finish(acc, min_count=mc, stage_through_cpu=True)
                    parts.append(res.to_host(sort=False))
                    res.close()
                out[(windows, mc)] = tuple(np.concatenate([p[i] for p in parts]) for i in range(3))
            acc.close()
        q.put((rank, out))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,k,cap,log2_bits,opts", [("bucketed", 31, 400_000, 24, ()), ("direct", 45, 400_000, 24, ()), ("bucketed", 31, 1, 24, ()),
                                                       ("bucketed", 31, 400_000, 27, (("accum_min_u", 8),)),
                                                       ("direct", 31, 1, 27, (("accum_min_u", 8),))])
def test_two_ranks_exchange_accumulators_for_exact_counts_across_shards(mode, k, cap, log2_bits, opts):
    """every rank accumulates its child shard; the ranks swap unit ranges and count their own share of the hash space
    from both ranks' pieces: the union of the ranks' tables must be the oracle's table of the WHOLE child, counts and
    min_count included (cap = 1: nearly everything travels through the all-gathered overflow lists; accum_min_u = 8 on a
    2^27-bit set: packed 6-byte units, counted from TWO piece-major pieces -- what a native exchange between GPUs delivers)"""
    n_reads, nh, seed, world = 6000, 4, 31337, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, n_reads, k, log2_bits, nh, seed, mode, cap, q, opts)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ocfg = orc.synth_cfg(genome_len=100_000)
    f = orc.new_filter(log2_bits)
    for smp in (0, 1):
        seq, off = orc.synth_reads(ocfg, smp, 0, n_reads)
        orc.bloom_insert(f, log2_bits, nh, seed, k, True, seq, off)
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
    for mc in (1, 2):
        km, cn, _ = orc.bloom_probe(f, log2_bits, nh, seed, k, True, cseq, coff, mc)
        for windows in (1, 2):
            hi = np.concatenate([outs[r][(windows, mc)][0] for r in range(world)])
            lo = np.concatenate([outs[r][(windows, mc)][1] for r in range(world)])
            cnt = np.concatenate([outs[r][(windows, mc)][2] for r in range(world)])
            order = np.lexsort((lo, hi))
            assert np.array_equal(hi[order], km["hi"]) and np.array_equal(lo[order], km["lo"]) and np.array_equal(cnt[order], cn)
            assert all(len(outs[r][(windows, mc)][1]) > 0 for r in range(world))      # both ranks hold a share
    assert int(cn.max()) >= 2


def test_native_allreduce_world_one_through_the_abi():
    """dk_comm_init / dk_set_allreduce_or with one rank: the communicator is accepted, the all-reduce is the
    identity, a second communicator is refused.  (World sizes above one need one GPU per rank -- RCCL refuses two
    ranks on one device -- so the native path beyond this runs for the first time in the driver's multi-GPU bench;
    the slice arithmetic it shares with the torch path is covered by tests/test_dist_gloo.py.)"""
    import denovo_kmer_amd as d
    rng = np.random.default_rng(3)
    reads = ["".join(rng.choice(list("ACGT"), size=120)) for _ in range(200)]
    for set_kind, bits in (("bloom", 24), ("exact", 26)):
        with d.Engine(k=31, filter_log2_bits=bits, seed=5, set_kind=set_kind) as eng:
            ks = d.KmerSet(eng)
            ks.insert_sequences(reads)
            before = ks.to_host()
            assert ks.allreduce_or() == 0                     # no communicator: a no-op
            eng.comm_init(None, 0, 1)
            assert ks.allreduce_or() == 0
            assert np.array_equal(ks.to_host(), before)
            with pytest.raises(d.DkError):
                eng.comm_init(None, 0, 1)
            eng.comm_finalize()
            eng.comm_init(None, 0, 1)
            with pytest.raises(d.DkError):
                eng.comm_init(None, 1, 1)
            ks.close()


def test_rccl_collectives_on_library_memory_with_one_rank():
    """The multi-GPU exchange hands torch.distributed (backend nccl = RCCL) tensors that wrap the LIBRARY's device memory
    (dist._device_bytes over __cuda_array_interface__: the accumulator's store and fill counters), not memory of torch's
    allocator.  One rank is all a single GPU allows, but it runs the same calls on the same kind of tensor: all_to_all_single,
    all_reduce and all_gather_into_tensor must take the wrapped memory and leave what the library put there."""
    import torch
    import torch.distributed as dist
    import denovo_kmer_amd as d
    from denovo_kmer_amd.dist import _device_bytes
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    rng = np.random.default_rng(11)
    reads = ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(3000)]
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        with d.Engine(k=31, filter_log2_bits=24, seed=5, mode="bucketed") as eng:
            dev = torch.device("cuda", eng.device_id)
            acc = d.ChildAccumulator(eng, None, capacity_records=1 << 20, window_index=0, window_count=1)
            acc.add(d.ReadBatch.from_sequences(eng, reads))
            n_units, cap, rb = acc.geometry()
            sp, fp, op, n_ovf = acc.device_view()
            store = _device_bytes(sp, n_units * cap * rb, dev).view(1, n_units * cap * rb)
            fill = _device_bytes(fp, n_units * 4, dev).view(torch.int32).view(1, n_units)
            assert int(fill.sum().item()) + n_ovf == acc.stats()["n_absent"] > 300000
            recv_store, recv_fill = torch.empty_like(store), torch.empty_like(fill)
            dist.all_to_all_single(recv_fill, fill)
            dist.all_to_all_single(recv_store, store)
            gathered = torch.empty(n_units, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(gathered, fill.view(-1))
            total = fill.sum().to(torch.int64).view(1)
            dist.all_reduce(total)
            torch.cuda.synchronize()
            assert torch.equal(recv_fill, fill) and torch.equal(recv_store, store) and torch.equal(gathered, fill.view(-1))
            assert int(total.item()) == int(fill.sum().item())
            # counting from the received copy gives what counting in place gives
            ref = acc.finish(min_count=1)
            got = acc.finish_pieces(recv_store.data_ptr(), recv_fill.data_ptr(), 1, 0, n_units, 0, 0, 1,
                                    keepalive=(recv_store, recv_fill))
            (ahi, alo, acnt), (bhi, blo, bcnt) = ref.to_host(), got.to_host()          # sorted by k-mer
            assert len(alo) == len(blo) > 0
            assert np.array_equal(alo, blo) and np.array_equal(acnt, bcnt)
            ref.close()
            got.close()
            acc.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("set_kind,bits", [("bloom", 26), ("exact", 26)])
def test_native_rccl_allreduce_with_one_rank_as_its_own_peer(set_kind, bits):
    """A one-rank RCCL communicator (dk_comm_unique_id + dk_comm_init(id, 0, 1)) sends the rank's slice to itself through
    ncclSend / ncclRecv, combines it with itself in the OR / union kernel and all-gathers in place: every RCCL entry point
    the library loads runs on the engine's stream, and the set must come out unchanged.  The staging buffer is 1 GiB, the
    set 8 MiB, so this is one piece; two GPUs and more are the driver's run."""
    import denovo_kmer_amd as d
    rng = np.random.default_rng(4)
    reads = ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(2000)]
    with d.Engine(k=31, filter_log2_bits=bits, seed=9, set_kind=set_kind) as eng:
        ks = d.KmerSet(eng)
        ks.insert_sequences(reads)
        before = ks.to_host()
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        assert ks.allreduce_or() == 0                  # nothing leaves the GPU
        assert np.array_equal(ks.to_host(), before)
        child = d.ReadBatch.from_sequences(eng, reads[:50] + ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(50)])
        res = d.KmerCounter(eng).child_only(child, ks)
        assert res.stats["n_absent"] == 50 * 120       # the 50 parent reads are all found, the 50 new ones are not
        res.close()
        eng.comm_finalize()
        ks.close()


@pytest.mark.parametrize("k,min_u,plain", [(31, 8, 0), (31, 0, 0), (45, 0, 0), (31, 8, 1)])
def test_native_accumulator_exchange_with_one_rank_as_its_own_peer(k, min_u, plain):
    """dk_accum_exchange_finish on a one-rank RCCL communicator: the header all-gather, the in-place all-to-all of fills
    and store (slice 0 goes through ncclSend / ncclRecv into the staging buffer and is copied back over itself) and the
    all-gather of the overflow list all run on the engine's stream, and counting from the exchanged store must give what
    dk_accum_finish gave before -- packed 6-byte units, plain 8-byte ones and 16-byte ones.  The accumulator is consumed
    afterwards.  Two GPUs and more are the driver's run (N > 1 remains unverified on hardware)."""
    import denovo_kmer_amd as d
    rng = np.random.default_rng(21)
    alphabet = np.array(list("ACGT"))
    parents = ["".join(alphabet[rng.integers(0, 4, size=150)]) for _ in range(300)]
    child = ["".join(alphabet[rng.integers(0, 4, size=150)]) for _ in range(1500)] + parents[:100]
    child = child + child[:400]
    with d.Engine(k=k, filter_log2_bits=27, n_hashes=4, seed=11, mode="bucketed") as eng:
        eng.set_option("accum_min_u", min_u)
        eng.set_option("accum_plain", plain)
        ks = d.KmerSet(eng)
        ks.insert_sequences(parents)
        # a capacity far below the sample: most units run full, the overflow list is exchanged too
        acc = d.ChildAccumulator(eng, ks, capacity_records=60_000 if min_u else 3_000_000)
        for part in (child[:700], child[700:]):
            acc.add(d.ReadBatch.from_sequences(eng, part))
        ref = acc.finish(min_count=2)
        rhi, rlo, rcnt = ref.to_host()
        assert len(rlo) > 0 and int(rcnt.max()) >= 2
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        got = acc.exchange_finish(min_count=2)
        ghi, glo, gcnt = got.to_host()
        assert np.array_equal(glo, rlo) and np.array_equal(ghi, rhi) and np.array_equal(gcnt, rcnt)
        assert got.stats["n_absent"] == ref.stats["n_absent"] and got.stats["n_distinct"] == ref.stats["n_distinct"]
        assert got.bytes_sent > 0
        assert "acc_exchange" in [n for n, _ in eng.timings()["stages"]]
        with pytest.raises(d.DkError):                 # consumed: the store is no longer in unit order
            acc.add(d.ReadBatch.from_sequences(eng, child[:10]))
        with pytest.raises(d.DkError):
            acc.finish()
        acc.reset(0)
        acc.add(d.ReadBatch.from_sequences(eng, child[:700]))
        again = acc.exchange_finish(min_count=1)
        assert len(again) > 0
        for r in (ref, got, again):
            r.close()
        eng.comm_finalize()
        acc.close()
        ks.close()


# ---- the native collectives with MORE than one rank, on one GPU, over tests/rccl_shim ------------------------------------
def _shim_path():
    """build tests/rccl_shim/shim.cpp (a shared-memory stand-in for librccl, test infrastructure only) on first use"""
    import subprocess
    src = os.path.join(ROOT, "tests", "rccl_shim", "shim.cpp")
    out_dir = os.path.join(ROOT, "tests", "rccl_shim", "_build")
    out = os.path.join(out_dir, "librccl_shim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-o", out, src, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def _native_worker(rank, world, shim, idq, n_reads, k, log2_bits, set_kind, staging_kb, cap, opts, q):
    os.environ["DK_RCCL_LIBRARY"] = shim
    sys.path.insert(0, ROOT)
    import denovo_kmer_amd as dk
    from denovo_kmer_amd.dist import shard_range
    gcfg = dk.synth_config(genome_len=100_000)
    eng = dk.Engine(k=k, filter_log2_bits=log2_bits, n_hashes=4, seed=31337, device_id=0, mode="bucketed",
                    rank=rank, world_size=world, set_kind=set_kind)
    eng.set_option("comm_staging_kb", staging_kb)
    for name, val in opts:
        eng.set_option(name, val)
    if rank == 0:
        uid = eng.comm_unique_id()
        for _ in range(world - 1):
            idq.put(uid)
    else:
        uid = idq.get(timeout=300)
    ks = dk.KmerSet(eng)
    lo, hi = shard_range(n_reads, rank, world)
    for s in (0, 1):
        ks.insert_reads(dk.ReadBatch.synth(eng, gcfg, s, lo, hi - lo))
    eng.comm_init(uid, rank, world)
    sent = ks.allreduce_or()
    table = ks.to_host().copy()
    out = {"table": table, "sent": sent, "n_keys": ks.popcount()}
    child = dk.ReadBatch.synth(eng, gcfg, 2, lo, hi - lo)
    res = dk.KmerCounter(eng).child_only(child, ks)
    out["local"] = res.to_host()
    res.close()
    if set_kind == "bloom":
        for mc in (1, 2):
            acc = dk.ChildAccumulator(eng, ks, capacity_records=cap)
            half = (hi - lo) // 2
            acc.add(dk.ReadBatch.synth(eng, gcfg, 2, lo, half))
            acc.add(dk.ReadBatch.synth(eng, gcfg, 2, lo + half, hi - lo - half))
            out[("geometry", mc)] = acc.geometry()
            got = acc.exchange_finish(min_count=mc)
            out[("accum", mc)] = got.to_host(sort=False)
            out[("accum_sent", mc)] = got.bytes_sent
            got.close()
            acc.close()
    q.put((rank, out))
    eng.comm_finalize()
    ks.close()
    eng.close()


def _run_native(world, n_reads, k, log2_bits, set_kind, staging_kb, cap, opts=()):
    shim = _shim_path()
    ctx = mp.get_context("spawn")
    q, idq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, shim, idq, n_reads, k, log2_bits, set_kind, staging_kb, cap, opts, q))
             for r in range(world)]
    for p in procs:
        p.start()
    try:
        outs = dict(q.get(timeout=500) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return outs


def _sum_tables(parts):
    """per-rank (hi, lo, count) lists of the ranks' own child shards -> one table with the counts summed per k-mer"""
    hi = np.concatenate([p[0] for p in parts])
    lo = np.concatenate([p[1] for p in parts])
    cnt = np.concatenate([p[2] for p in parts]).astype(np.uint64)
    order = np.lexsort((lo, hi))
    hi, lo, cnt = hi[order], lo[order], cnt[order]
    first = np.ones(len(lo), dtype=bool)
    first[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
    return hi[first], lo[first], np.add.reduceat(cnt, np.flatnonzero(first)) if len(lo) else cnt


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,k,log2_bits,staging_kb,cap,opts", [
    (2, 31, 26, 0, 400_000, ()),                          # one piece per slice
    (2, 31, 26, 1024, 400_000, ()),                       # 4-MiB slices through 1 MiB of staging: four pieces
    (4, 31, 27, 1536, 400_000, (("accum_min_u", 8),)),    # four ranks, three peers of 512 KiB each; packed 6-byte units
    (4, 45, 26, 640, 1, ()),                              # 16-byte records, nearly everything through the overflow lists
])
def test_native_collectives_between_ranks_through_the_c_abi(world, k, log2_bits, staging_kb, cap, opts):
    """dk_comm_init / dk_set_allreduce_or / dk_accum_exchange_finish with 2 and 4 ranks -- the calls a host without torch
    makes -- on one GPU: tests/rccl_shim stands in for librccl (which refuses two ranks on a device) and moves every
    ncclSend / ncclRecv / ncclAllGather through shared memory, so peer indices, staging slots, piece loops and the in-place
    exchange run as they do between GPUs.  Every rank's set must equal the oracle's filter of the WHOLE parent input, and
    the union of the ranks' count tables the oracle's table of the WHOLE child.  RCCL itself between GPUs stays the
    driver's run."""
    n_reads = 6000
    outs = _run_native(world, n_reads, k, log2_bits, "bloom", staging_kb, cap, opts)
    ocfg = orc.synth_cfg(genome_len=100_000)
    f = orc.new_filter(log2_bits)
    for smp in (0, 1):
        seq, off = orc.synth_reads(ocfg, smp, 0, n_reads)
        orc.bloom_insert(f, log2_bits, 4, 31337, k, True, seq, off)
    for r in range(world):
        assert np.array_equal(outs[r]["table"].view(np.uint64), f.view(np.uint64)), "rank %d holds a different set" % r
        assert outs[r]["sent"] == 2 * (world - 1) * (1 << log2_bits) // 8 // world
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
    km, cn, _ = orc.bloom_probe(f, log2_bits, 4, 31337, k, True, cseq, coff, 1)
    hi, lo, cnt = _sum_tables([outs[r]["local"] for r in range(world)])
    assert np.array_equal(hi, km["hi"]) and np.array_equal(lo, km["lo"]) and np.array_equal(cnt, cn.astype(np.uint64))
    for mc in (1, 2):
        km, cn, _ = orc.bloom_probe(f, log2_bits, 4, 31337, k, True, cseq, coff, mc)
        parts = [outs[r][("accum", mc)] for r in range(world)]
        hi, lo, cnt = (np.concatenate([p[i] for p in parts]) for i in range(3))
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], km["hi"]) and np.array_equal(lo[order], km["lo"]) and np.array_equal(cnt[order], cn)
        assert all(len(p[1]) > 0 for p in parts)             # every rank counted a share of the hash space
        assert all(outs[r][("accum_sent", mc)] > 0 for r in range(world))
        want_bytes = 6 if dict(opts).get("accum_min_u", 0) >= 8 and k <= 32 else 16 if k > 32 else 8
        assert all(outs[r][("geometry", mc)][2] == want_bytes for r in range(world))
    assert int(cn.max()) >= 2


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,k,staging_kb", [(2, 31, 512), (4, 45, 0)])
def test_native_union_of_exact_sets_between_ranks(world, k, staging_kb):
    """the exact set's all-reduce (union_slices_kernel on received table slices) with 2 and 4 ranks over tests/rccl_shim"""
    n_reads, log2_bits = 4000, 27
    outs = _run_native(world, n_reads, k, log2_bits, "exact", staging_kb, 0)
    ocfg = orc.synth_cfg(genome_len=100_000)
    seqs = [orc.synth_reads(ocfg, s, 0, n_reads) for s in (0, 1)]
    allseq = np.concatenate([s[0] for s in seqs])
    alloff = np.concatenate([seqs[0][1], seqs[1][1][1:] + seqs[0][1][-1]])
    cseq, coff = orc.synth_reads(ocfg, 2, 0, n_reads)
    km, cn, _ = orc.exact_child_only(k, True, allseq, alloff, cseq, coff)
    pk, _, _ = orc.count_reads(k, True, allseq, alloff)
    for r in range(1, world):
        assert np.array_equal(outs[r]["table"], outs[0]["table"])
    assert all(outs[r]["n_keys"] == len(pk) for r in range(world))
    hi, lo, cnt = _sum_tables([outs[r]["local"] for r in range(world)])
    assert np.array_equal(hi, km["hi"]) and np.array_equal(lo, km["lo"]) and np.array_equal(cnt, cn.astype(np.uint64))
