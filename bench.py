#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: Gk-mers/s, child reads vs parent Bloom, k=31.

    python bench.py --gpus N --steps K --warmup W

A "step" is one membership pass of the hot path over one resident batch of synthetic child reads
(BASELINE.json configs[1]: k=31, chr20-scale 30x trio of synthetic 150 bp reads = 12.8 M reads per
sample per GPU) against the parent filter resident in HBM: k-mer extraction, canonicalisation,
hashing, filter probe, and counting of the child-only k-mers into the output table.  Inputs are in
HBM before the timed region.  The parent build (insert of both parents, and for N > 1 the
OR-all-reduce of the filter over RCCL) is run once before the timed steps and reported in the same
JSON line under "parent_build"; it is not part of `value` (the metric is the probe pass).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- every rank owns 12.8 M
reads of each sample of a genome N times larger, builds a partial filter, the partials are
OR-all-reduced, and each rank probes its own child shard with no collective in the timed step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def stage_algorithmic_bytes(stage, st, filter_bytes, read_len, k, geom=None):
    """Algorithmic HBM bytes one launch of `stage` must move.
    st = dk_stats of the pass (n_bases, n_windows, n_valid, n_absent, n_distinct)."""
    nb, nv, na, nd = st["n_bases"], st["n_valid"], st["n_absent"], st["n_distinct"]
    stream = nb * 3 / 8.0                                   # 2-bit bases + 1-bit mask
    table = {
        # direct family
        "probe_direct": stream + 64.0 * nv + 8.0 * na,     # one 64-B filter block per k-mer, absent k-mers appended
        "insert_direct": stream + 128.0 * nv,              # block fetched and written back
        "count_insert": 8.0 * na + 8.0 * na,               # candidate read + slot/count update
        "count_emit": 12.0 * nd,
        # bucketed family (records are 8-byte hashes)
        "scan_part": stream + 8.0 * nv,                    # read stream, write one record per valid k-mer
        "repart": 16.0 * nv,                               # read + write every record once
        "seg_probe": 8.0 * nv + filter_bytes + 8.0 * na,   # records + one sweep of the filter + absent records out
        "seg_insert": 8.0 * nv + 2.0 * filter_bytes,       # records + filter read and written back
        "seg_count": 8.0 * na + 12.0 * nd,                 # absent records in, (k-mer, count) out
        # exact set (--set-kind exact): the segments are hash tables, swept exactly like the filter
        "seg_exact_probe": 8.0 * nv + filter_bytes + 8.0 * na,
        "seg_exact_insert": 8.0 * nv + 2.0 * filter_bytes,
    }
    return table.get(stage)


def cpu_baseline(dk, eng, kset, gcfg, args, sample_reads):
    """Time the CPU oracle ("port": the build's C restatement, OpenMP over reads on the host cores
    this process may use) on a bounded sample of the same child workload against the same filter,
    and cross-check the GPU on that sample."""
    import numpy as np
    from oracle import orc
    filt = kset.to_host()
    sb = dk.ReadBatch.synth(eng, gcfg, 2, 0, sample_reads)
    bases, mask, _ = sb.download()
    seq, off = orc.unpack_fixed(bases, mask, sample_reads, args.read_len)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box grants this job a 16-core CPU share whatever the affinity mask says
    cores = max(1, min(cores, args.cpu_threads if args.cpu_threads > 0 else 16))
    t0 = time.perf_counter()
    km, cn, st = orc.bloom_probe(filt, args.log2_bits, args.n_hashes, args.seed, args.k, True, seq, off, n_threads=cores)
    dt = time.perf_counter() - t0
    res = dk.KmerCounter(eng).child_only(sb, kset)
    hi, lo, cnt = res.to_host()
    ok = bool(np.array_equal(lo, km["lo"]) and np.array_equal(hi, km["hi"]) and np.array_equal(cnt, cn))
    res.close()
    sb.close()
    return {"value": st["n_windows"] / dt / 1e9, "unit": "Gk-mers/s", "cores": cores, "kind": "port",
            "sample": f"first {sample_reads} child reads of the same workload ({st['n_windows']} windows, "
                      f"{dt:.1f} s), oracle/dk_oracle.c with {cores} OpenMP thread(s), same {len(filt) * 8 >> 20} MiB filter",
            "gpu_matches_oracle_on_sample": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--reads", type=int, default=12_800_000, help="reads per sample per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=64 << 20, help="genome length per GPU (bases)")
    ap.add_argument("--log2-bits", type=int, default=0, help="filter size; 0 = 34 + ceil(log2(gpus))")
    ap.add_argument("--n-hashes", type=int, default=4)
    ap.add_argument("--seed", type=int, default=20260313)
    ap.add_argument("--mode", default="auto", choices=["auto", "direct", "bucketed"])
    ap.add_argument("--set-kind", default="bloom", choices=["bloom", "exact"],
                    help="bloom: the headline metric's parent Bloom filter; exact: exact parent set (side measurement, "
                         "--log2-bits then defaults to 36 + ceil(log2(gpus)))")
    ap.add_argument("--cpu-sample-reads", type=int, default=6_000_000)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline; 0 = min(cores this process may use, 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (collectives staged through the host)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import denovo_kmer_amd as dk
    from denovo_kmer_amd.dist import local_reduce_fn, or_allreduce_

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    on_host = args.backend == "gloo"          # reductions of scalars go through host tensors under gloo

    if args.log2_bits == 0:
        args.log2_bits = (36 if args.set_kind == "exact" else 34) + max(0, (world - 1).bit_length())
    genome_len = args.genome * world
    gcfg = dk.synth_config(seed=args.seed, genome_len=genome_len, read_len=args.read_len)
    eng = dk.Engine(k=args.k, filter_log2_bits=args.log2_bits, n_hashes=args.n_hashes, seed=args.seed,
                    device_id=local_rank, mode=args.mode, rank=rank, world_size=world, set_kind=args.set_kind)
    filter_bytes = (1 << args.log2_bits) // 8
    filt = torch.zeros(filter_bytes // 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    kset = dk.KmerSet(eng, device_ptr=filt.data_ptr(), keepalive=filt)
    if args.set_kind == "exact":
        kset.clear()                         # an empty exact set is not all-zero memory
    first = rank * args.reads

    # ---- parent build (once; reported, not part of `value`) ---------------------------------
    # (the first insert also grows the engine's workspace pool; the rate is taken from the second parent)
    insert_ms, insert_windows, insert_stages = 0.0, 0, {}
    for s in (0, 1):
        pb = dk.ReadBatch.synth(eng, gcfg, s, first, args.reads)
        st = kset.insert_reads(pb)
        t = eng.timings()
        insert_ms, insert_windows = t["total_ms"], st["n_windows"]
        insert_stages = {name: ms for name, ms in t["stages"]}
        pb.close()
    allreduce_ms, allreduce_bytes = 0.0, 0
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        allreduce_bytes = or_allreduce_(filt, local_reduce_fn(eng), stage_through_cpu=on_host)
        torch.cuda.synchronize()
        dist.barrier()
        allreduce_ms = (time.perf_counter() - t0) * 1e3
    # every rank must now hold the same filter: compare bit counts across ranks
    popc = kset.popcount()
    filter_consistent = True
    if world > 1:
        pc = torch.tensor([popc, -popc], dtype=torch.int64, device="cpu" if on_host else dev)
        dist.all_reduce(pc, op=dist.ReduceOp.MAX)
        filter_consistent = bool(int(pc[0].item()) == -int(pc[1].item()))

    # ---- child membership pass: warmup + K timed steps -----------------------------------------
    child = dk.ReadBatch.synth(eng, gcfg, 2, first, args.reads)
    counter = dk.KmerCounter(eng)
    stats = None
    for _ in range(args.warmup):
        r = counter.child_only(child, kset)
        stats = r.stats
        r.close()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    stage_sum, total_dev_ms = {}, 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = counter.child_only(child, kset)
        stats = r.stats
        t = eng.timings()                       # HIP events on the engine's stream, this step
        total_dev_ms += t["total_ms"]
        for name, ms in t["stages"]:
            stage_sum[name] = stage_sum.get(name, 0.0) + ms
        r.close()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        sdev = "cpu" if on_host else dev
        tt = torch.tensor([elapsed], dtype=torch.float64, device=sdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ww = torch.tensor([stats["n_windows"]], dtype=torch.int64, device=sdev)
        dist.all_reduce(ww, op=dist.ReduceOp.SUM)
        windows_all = int(ww.item())
    else:
        windows_all = stats["n_windows"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = windows_all * args.steps / elapsed / 1e9
        stages = {n: ms / args.steps for n, ms in stage_sum.items()}
        dom = max(stages, key=stages.get)
        dom_bytes = stage_algorithmic_bytes(dom, stats, filter_bytes, args.read_len, args.k)
        achieved = dom_bytes / (stages[dom] * 1e-3) / 1e9 if dom_bytes else None
        # HBM bytes of the dominant kernel from the committed PMC passes (profiles/traffic.json:
        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs), same configuration only
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                tj = json.load(open(prof))
                if tj.get("reads") == args.reads and tj.get("log2_bits") == args.log2_bits and world == 1:
                    traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Gk-mers/sec (child reads vs parent %s), k=%d" % ("Bloom" if args.set_kind == "bloom" else "exact set", args.k),
            "value": value, "unit": "Gk-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: k=%d, chr20-scale 30x synthetic trio, %d x %d bp reads per sample per GPU, "
                                   "genome %d Mb, parent %s 2^%d bits resident in HBM"
                                   % (args.k, args.reads, args.read_len, genome_len >> 20,
                                      "Bloom" if args.set_kind == "bloom" else "exact set (open-addressing tables)", args.log2_bits),
                       "k": args.k, "reads_per_gpu": args.reads, "read_len": args.read_len,
                       "filter_log2_bits": args.log2_bits, "n_hashes": args.n_hashes, "mode": args.mode,
                       "set_kind": args.set_kind,
                       "parallelism": "reads sharded x%d, %s-all-reduce of parent set" % (world, "OR" if args.set_kind == "bloom" else "union")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms": stages[dom]},
            "stages_ms": stages,
            "device_ms_per_step": total_dev_ms / args.steps,
            "pass_stats": stats,
            "parent_build": {"insert_gkmers_s": insert_windows / (insert_ms * 1e-3) / 1e9 if insert_ms else None,
                             "insert_ms": insert_ms, "insert_stages_ms": insert_stages,
                             "or_allreduce_ms": allreduce_ms, "or_allreduce_bytes_per_rank": allreduce_bytes,
                             "filter_bits_set": popc, "filter_identical_on_all_ranks": filter_consistent},
        }
        if args.set_kind == "exact":
            out["parent_build"]["exact_set_load"] = popc / (filter_bytes / (16 if args.k > 32 else 8))
        if world == 1 and not args.no_cpu_baseline and args.set_kind == "bloom":
            out["cpu_baseline"] = cpu_baseline(dk, eng, kset, gcfg, args, min(args.cpu_sample_reads, args.reads))
        print(json.dumps(out), flush=True)

    child.close()
    kset.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
