"""Multi-GPU host logic: read sharding and the OR- (Bloom) / union- (exact set) all-reduce of the parent set.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  The hot path shards
by reads with one exchange step: every rank builds a partial parent filter of the FULL size from
its parent-read shard, the partial filters are OR-combined, then every rank probes its child
shard locally.  RCCL has no bitwise-OR reduction (ncclRedOp_t = sum/prod/max/min/avg), so the
all-reduce is composed:

    all_to_all_single   rank r receives slice r of every rank's filter      (P-1)/P * M bytes out
    local OR            HIP kernel dk_or_reduce_slices over the P slices
    all_gather          every rank receives every reduced slice            (P-1)/P * M bytes in

On a fully connected xGMI node each phase uses all 7 links of a GPU at once, which is why this
shape is used instead of a ring.  torch is plumbing only (device memory + collectives).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """contiguous shard [lo, hi) of n_items for `rank`; shards differ by at most one item"""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def hip_or_fn(engine):
    """local OR step on the GPU through the C ABI (dk_or_reduce_slices)"""
    def fn(dst, src, n_slices):
        torch.cuda.synchronize()            # collectives ran on torch's stream, the kernel on the engine's
        engine.or_reduce_slices(dst.data_ptr(), src.data_ptr(), n_slices, dst.numel() * dst.element_size())
    return fn


def hip_union_fn(engine, group=None):
    """local step for exact sets (Engine(set_kind="exact")): the slices are runs of 64-KiB table
    segments and are merged key by key through the C ABI (dk_union_slices)"""
    def fn(dst, src, n_slices):
        torch.cuda.synchronize()
        slice_bytes = dst.numel() * dst.element_size()
        assert slice_bytes % 65536 == 0, "an exact set splits across ranks in whole 64-KiB segments"
        first_segment = dist.get_rank(group) * (slice_bytes // 65536)     # rank r reduces slice r
        engine.union_slices(dst.data_ptr(), src.data_ptr(), n_slices, slice_bytes, first_segment)
    return fn


def local_reduce_fn(engine, group=None):
    """the local reduction step that matches what the engine's sets hold"""
    return hip_union_fn(engine, group) if engine.set_kind == "exact" else hip_or_fn(engine)


def or_allreduce_(filt, or_fn, group=None, stage_through_cpu=False):
    """In-place all-reduce of the parent set `filt` (1-D int64 tensor, one per rank, equal sizes):
    bitwise OR for a Bloom filter (or_fn = hip_or_fn), key union for an exact set (hip_union_fn).
    On return the result is complete in device memory for every stream (the call ends with a device
    synchronize: the collectives run on torch's stream, the engine probes on its own).

    or_fn(dst, src, n_slices): dst <- dst combined with the n_slices contiguous slices in src.
    stage_through_cpu: move the payload through host memory around each collective (for rehearsing
    the GPU path over the gloo backend, which has no device all-to-all); the OR step still runs
    wherever `filt` lives.
    Returns bytes sent per rank (for bandwidth reporting)."""
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    assert filt.dim() == 1 and filt.is_contiguous() and filt.dtype == torch.int64
    n = filt.numel()
    assert n % (2 * world) == 0, "filter words must split into 16-byte-aligned slices per rank"
    sl = n // world
    if stage_through_cpu:
        send = filt.cpu()
        recv_h = torch.empty_like(send)
        dist.all_to_all_single(recv_h, send, group=group)
        recv = recv_h.to(filt.device)
    else:
        recv = torch.empty_like(filt)
        dist.all_to_all_single(recv, filt, group=group)      # recv[j*sl:(j+1)*sl] = rank j's slice `rank`
    reduced = recv[:sl]
    or_fn(reduced, recv[sl:], world - 1)
    if stage_through_cpu:
        out_h = torch.empty(n, dtype=filt.dtype)
        dist.all_gather_into_tensor(out_h, reduced.cpu(), group=group)
        filt.copy_(out_h)
    else:
        dist.all_gather_into_tensor(filt, reduced, group=group)
        # the gather is ordered only against torch's current stream; the engine's stream (hipStreamNonBlocking)
        # reads `filt` next, so the in-place contract needs the device to be idle here
        if filt.is_cuda:
            torch.cuda.synchronize()
        del recv, reduced
    return 2 * (world - 1) * sl * filt.element_size()


class _DeviceMemory:
    """library-owned device memory seen by torch through the CUDA array interface (no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _device_bytes(ptr, nbytes, device):
    return torch.as_tensor(_DeviceMemory(ptr, nbytes), device=device)


def accum_exchange_finish(acc, min_count=1, group=None, stage_through_cpu=False):
    """Multi-GPU end of a child pass: every rank has accumulated ITS child reads (ChildAccumulator.add) over the same
    hash window; the ranks now swap unit ranges -- all_to_all_single of the stores and of the fill counters: rank r
    receives units [r u/P, (r+1) u/P) of every rank -- and each rank counts its own range from the P pieces
    (dk_accum_finish_pieces).  The returned KmerCounts holds the child-only k-mers of this rank's share of the hash
    space with counts summed over ALL ranks' reads, min_count applied to the sums; the ranks' tables are disjoint and
    their union is the answer, so nothing is ever gathered on one GPU.  The (rare) overflow lists are all-gathered.
    stage_through_cpu: move the payloads through host memory (gloo rehearsal on one GPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return acc.finish(min_count=min_count)
    rank = dist.get_rank(group)
    eng = acc.engine
    dev = torch.device("cuda", eng.device_id)
    n_units, cap, rb = acc.geometry()
    assert n_units % world == 0, "the units of a window split evenly over a power-of-two number of ranks"
    # a rank whose accumulator is unusable (it lost records in an earlier call) must not leave the others inside the
    # collectives below: its state travels with the geometry check, and every rank raises when any rank failed
    view, failure = None, None
    try:
        view = acc.device_view()
    except Exception as exc:                               # noqa: BLE001 -- reported on every rank below
        failure = exc
    # the slices are interpreted with THIS rank's geometry: every rank must have created its accumulator alike
    geo = torch.tensor([n_units, -n_units, cap, -cap, rb, -rb, 1 if failure else 0], dtype=torch.int64,
                       device="cpu" if stage_through_cpu else dev)
    dist.all_reduce(geo, op=dist.ReduceOp.MAX, group=group)
    if int(geo[6]):
        raise RuntimeError("accumulator exchange abandoned on every rank: %s" % (failure or "another rank's accumulator failed"))
    if any(int(geo[i]) != -int(geo[i + 1]) for i in (0, 2, 4)):
        raise ValueError("accumulators differ between ranks (units, unit capacity or record size): create them with the "
                         "same capacity_records and window_count on every rank")
    upr = n_units // world
    sp, fp, op, n_ovf = view
    store = _device_bytes(sp, n_units * cap * rb, dev).view(world, upr * cap * rb)
    fill = _device_bytes(fp, n_units * 4, dev).view(torch.int32).view(world, upr)

    def a2a(t):
        if stage_through_cpu:
            send = t.cpu()
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
            return recv.to(dev)
        recv = torch.empty_like(t)
        dist.all_to_all_single(recv, t, group=group)
        return recv

    recv_fill = a2a(fill)
    recv_store = a2a(store)
    # overflow lists: every rank gets all of them and keeps the records of its own unit range
    sizes = torch.zeros(world, dtype=torch.int64, device="cpu" if stage_through_cpu else dev)
    sizes[rank] = n_ovf
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = [int(x) for x in sizes.cpu()]
    extra, n_extra = None, sum(sizes)
    if n_extra:
        ob = 16 if rb == 16 else 8          # the overflow list holds plain records (packed units: 6-byte records, 8 here)
        pad = max(sizes)
        mine = torch.zeros(pad * ob, dtype=torch.uint8, device=dev)
        if n_ovf:
            mine[: n_ovf * ob] = _device_bytes(op, n_ovf * ob, dev)
        if stage_through_cpu:
            g = torch.empty(world * pad * ob, dtype=torch.uint8)
            dist.all_gather_into_tensor(g, mine.cpu(), group=group)
            g = g.to(dev)
        else:
            g = torch.empty(world * pad * ob, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(g, mine, group=group)
        extra = torch.cat([g[r * pad * ob: (r * pad + sizes[r]) * ob] for r in range(world) if sizes[r]])
    torch.cuda.synchronize()
    res = acc.finish_pieces(recv_store.data_ptr(), recv_fill.data_ptr(), world, rank * upr, upr,
                            extra.data_ptr() if extra is not None else 0, n_extra, min_count)
    res._keep = (recv_store, recv_fill, extra)
    return res


def comm_init_from_torch(engine, group=None):
    """Join `engine` to an RCCL communicator of its own behind the C ABI (dk_comm_init), using torch.distributed
    only to hand rank 0's 128-byte id to the other ranks.  After this KmerSet.allreduce_or() runs the composed
    all-reduce natively on the engine's stream -- the path a Rust host takes (it has no torch)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    box = [None]
    if rank == 0 and world > 1:
        try:
            box = [engine.comm_unique_id()]
        except Exception as exc:            # noqa: BLE001 -- librccl missing: every rank must learn it, none may wait for an id
            box = [str(exc)]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError("no RCCL communicator: %s" % (box[0],))
    engine.comm_init(box[0], rank, world)


def filter_digest(filt, chunk_words=1 << 24):
    """position-sensitive 64-bit digest of a set's words, sum of w[i] * (2 i + 1) mod 2^64 (as a signed int),
    computed in chunks on the device: equal filters <=> equal digests up to a 2^-64 accident, unlike a bit count"""
    assert filt.dim() == 1 and filt.dtype == torch.int64
    acc = torch.zeros((), dtype=torch.int64, device=filt.device)
    n = filt.numel()
    for lo in range(0, n, chunk_words):
        c = filt[lo:lo + chunk_words]
        idx = torch.arange(lo, lo + c.numel(), dtype=torch.int64, device=filt.device)
        acc += (c * (2 * idx + 1)).sum()
    return int(acc.item())


def merge_counts(hi, lo, cnt, min_count=1, group=None):
    """Gather per-rank (k-mer, count) lists to every rank and sum counts per k-mer.

    A child k-mer can be seen by several ranks (reads are sharded, not k-mers), so the per-rank
    child-only tables are concatenated and reduced by key.  Ranks must run their engines with
    min_count=1; the threshold is applied here, after the sum.  Inputs/outputs are numpy arrays;
    the lists are small next to the reads."""
    import numpy as np
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (np.asarray(hi), np.asarray(lo), np.asarray(cnt)), group=group)
    ahi = np.concatenate([p[0] for p in parts]).astype(np.uint64)
    alo = np.concatenate([p[1] for p in parts]).astype(np.uint64)
    acnt = np.concatenate([p[2] for p in parts]).astype(np.uint64)
    if len(alo) == 0:
        return ahi, alo, acnt.astype(np.uint32)
    order = np.lexsort((alo, ahi))
    ahi, alo, acnt = ahi[order], alo[order], acnt[order]
    new = np.ones(len(alo), dtype=bool)
    new[1:] = (ahi[1:] != ahi[:-1]) | (alo[1:] != alo[:-1])
    idx = np.flatnonzero(new)
    sums = np.add.reduceat(acnt, idx)
    keep = sums >= min_count
    return ahi[idx][keep], alo[idx][keep], np.minimum(sums[keep], 0xFFFFFFFF).astype(np.uint32)


def merge_counts_device(engine, table, min_count=1, group=None, stage_through_cpu=False):
    """Device-side counterpart of merge_counts for tables too large to pickle: every rank's (k-mer, count)
    arrays are gathered with all_gather_into_tensor (padded to the largest table), attached as tables in
    place (dk_result_attach) and summed by k-mer with dk_result_merge.  Returns the merged KmerCounts
    (identical on every rank).  `table` must come from an engine with min_count=1.
    stage_through_cpu: gloo rehearsal (no device collectives)."""
    from .api import KmerCounter, KmerCounts
    world = dist.get_world_size(group)
    dev = torch.device("cuda", engine.device_id)
    wide = engine.k > 32
    n = len(table)
    sizes = torch.zeros(world, dtype=torch.int64, device="cpu" if stage_through_cpu else dev)
    sizes[dist.get_rank(group)] = n
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = [int(x) for x in sizes.cpu()]
    cap = max(max(sizes), 1)
    lo = torch.zeros(cap, dtype=torch.int64, device=dev)
    hi = torch.zeros(cap if wide else 1, dtype=torch.int64, device=dev)
    cnt = torch.zeros(cap, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if n:
        table.copy_to_device(lo.data_ptr(), hi.data_ptr() if wide else 0, cnt.data_ptr())
    parts = []
    for t in ([lo, cnt, hi] if wide else [lo, cnt]):
        if stage_through_cpu:
            g = torch.empty(world * cap, dtype=t.dtype)
            dist.all_gather_into_tensor(g, t.cpu(), group=group)
            g = g.to(dev)
        else:
            g = torch.empty(world * cap, dtype=t.dtype, device=dev)
            dist.all_gather_into_tensor(g, t, group=group)
        parts.append(g)
    torch.cuda.synchronize()
    glo, gcnt = parts[0], parts[1]
    ghi = parts[2] if wide else None
    tables = []
    for r in range(world):
        if sizes[r] == 0:
            continue
        tables.append(KmerCounts.from_device(engine, glo[r * cap:].data_ptr(), ghi[r * cap:].data_ptr() if wide else 0,
                                             gcnt[r * cap:].data_ptr(), sizes[r], keepalive=(glo, ghi, gcnt)))
    merged = KmerCounter(engine).merge(tables, min_count=min_count)
    for t in tables:
        t.close()
    return merged
