#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: Gk-mers/s, child reads vs parent Bloom, k=31.

    python bench.py --gpus N --steps K --warmup W [--workload wgs|chr20|ont]

Workloads (BASELINE.json `configs`):
  wgs   (default) configs[2]: k=31, full 30x whole-genome trio, ~1.2 B synthetic 150-bp reads per sample on one
        MI355X.  The parents are inserted batch by batch into a 2^39-bit (64 GiB) Bloom filter resident in HBM
        (reported under "parent_build"); the child is streamed in batches through dk_accum_add, which keeps the
        absent k-mer occurrences on the GPU so that counts and min_count are exact over the whole sample.  With
        6-byte packed accumulator records and the slab-wise partition (one scan per batch, level 2 slab by slab)
        the occurrences of the whole hash space fit beside the filter: ONE pass over the child (`hash_windows` 1;
        --windows 2 restores round 2's two passes, each covering half of the hash space and of the filter).
        A "step" = one child batch through one pass; it completes 1/hash_windows of the batch's membership work,
        so `value` counts n_windows / hash_windows per step.  The steps walk the resident batches; whenever every
        resident batch has been added once, and after the last step, what was accumulated is counted
        (dk_accum_finish) inside the timed region.  After the timed steps the whole child is run end
        to end (all passes, all batches, finish) and reported under "end_to_end", and -- N = 1 -- the other two
        single-GPU workloads run a few steps each and are attached under "other_workloads".
  chr20 configs[1]: k=31, chr20-scale 30x trio = 12.8 M reads per sample per GPU, 2^34-bit filter; a step is one
        dk_probe call over the resident child batch (round 1's headline).
  ont   configs[4]: k=51, 10-kb reads with 5 % errors, 192 k reads per sample, 2^35-bit filter; step as chr20.

Inputs are resident in HBM before the timed region.  N > 1 (launched by torch.distributed.run, one rank per GPU):
every rank owns 1/N of the reads of each sample (wgs) or its own 12.8 M reads of an N times larger genome (chr20,
ont), builds a partial filter, the partials are OR-all-reduced (RCCL), and each rank probes its own child shard.
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
XGMI_LINK_GBS = 153.0          # per link, 7 links per GPU

WORKLOADS = {
    "wgs": dict(k=31, read_len=150, reads=1_200_000_000, batch=64_000_000, parent_batch=128_000_000, log2_bits=39, err=5e-3,
                min_count=2, windows=1, cfg="configs[2]"),
    "chr20": dict(k=31, read_len=150, reads=12_800_000, batch=12_800_000, parent_batch=12_800_000, log2_bits=34, err=5e-3,
                  min_count=1, windows=1, cfg="configs[1]"),
    "ont": dict(k=51, read_len=10_000, reads=192_000, batch=192_000, parent_batch=192_000, log2_bits=35, err=5e-2, min_count=1,
                windows=1, cfg="configs[4]"),
}


def stage_algorithmic_bytes(stage, st, filter_bytes, rec_bytes, windows=1, absent_rec_bytes=None):
    """Algorithmic HBM bytes one launch of `stage` must move.
    st: n_bases, n_valid (all valid windows), n_absent (absent ones inside the hash window), n_distinct, n_emitted;
    windows: hash windows of the pass (records, filter share and absent records are 1/windows of the batch's);
    absent_rec_bytes: bytes of an absent record where it differs from a partition record (packed accumulator units: 6)."""
    nb, na = st["n_bases"], st["n_absent"]
    nv = st["n_valid"] / windows                            # records of this pass
    nd, ne = st.get("n_distinct", 0), st.get("n_emitted", 0)
    fb = filter_bytes / windows
    R = float(rec_bytes)
    RA = float(absent_rec_bytes or rec_bytes)
    stream = nb * 3 / 8.0                                   # 2-bit bases + 1-bit mask
    table = {
        # direct family
        "probe_direct": stream + 64.0 * nv + R * na,       # one 64-B filter block per k-mer, absent k-mers appended
        "insert_direct": stream + 128.0 * nv,              # block fetched and written back
        "count_insert": 2 * R * na,                         # candidate read + slot/count update
        "count_emit": (R + 4) * nd,
        # bucketed family (records: 8-byte hashes, 16 bytes for k > 32)
        "scan_part": stream + R * nv,                       # read stream, write one record per k-mer of the window
        "repart": 2 * R * nv,                               # read + write every record once
        "repart3": 2 * R * nv,
        "seg_probe": R * nv + fb + RA * na,                 # records + one sweep of the filter share + absent records out
        "seg_insert": R * nv + 2.0 * fb,                    # records + filter read and written back
        "seg_count": RA * na + (R + 4) * ne,                # absent records in, (k-mer, count) out
        "seg_count_dry": R * na,                            # sizing run of a min_count > 1 finish: records in, nothing out
        "count_split": 2 * R * na,
        # exact set (--set-kind exact): the segments are hash tables, swept exactly like the filter
        "seg_exact_probe": R * nv + fb + RA * na,
        "seg_exact_insert": R * nv + 2.0 * fb,
    }
    return table.get(stage)


def roofline_of(stage_ms, stage_bytes, traffic_by_kernel, traffic_source, launches=None):
    """roofline object of the pass: the dominant kernel on top, every stage under "stages", the pass total.
    launches: stage -> kernel launches per step (slab-wise stages: one per level-2 slab); bytes and time are those of the
    stage's launches of one step together, and achieved = bytes / time is the same per launch or per step"""
    stages = {}
    for name, ms in stage_ms.items():
        b = stage_bytes.get(name)
        gbs = b / (ms * 1e-3) / 1e9 if b and ms > 0 else None
        stages[name] = {"algorithmic_bytes": b, "ms": ms, "achieved": gbs,
                        "frac": gbs / HBM_PEAK_GBS if gbs else None}
    timed = {n: v for n, v in stages.items() if v["achieved"]}
    dom = max(timed, key=lambda n: timed[n]["ms"]) if timed else None
    tot_b = sum(v["algorithmic_bytes"] for v in timed.values())
    tot_ms = sum(v["ms"] for v in timed.values())
    out = {"bound": "hbm", "kernel": dom, "achieved": timed[dom]["achieved"] if dom else None, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": timed[dom]["frac"] if dom else None,
           "traffic": ((traffic_by_kernel or {}).get(dom) or 0) / (launches or {}).get(dom, 1) or None,
           "traffic_per_step": (traffic_by_kernel or {}).get(dom), "traffic_source": traffic_source,
           "launches_per_step": (launches or {}).get(dom, 1) if dom else None,
           "algorithmic_bytes_per_launch": timed[dom]["algorithmic_bytes"] / (launches or {}).get(dom, 1) if dom else None,
           "kernel_ms_per_launch": timed[dom]["ms"] / (launches or {}).get(dom, 1) if dom else None,
           "algorithmic_bytes_per_step": timed[dom]["algorithmic_bytes"] if dom else None,
           "kernel_ms": timed[dom]["ms"] if dom else None,
           "stages": stages,
           "pass": {"algorithmic_bytes": tot_b, "ms": tot_ms, "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else None,
                    "frac": tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tot_ms else None}}
    return out


def committed_traffic(workload, reads, log2_bits, world, reads_per_step=None):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json: FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, separate --pmc runs by tools/profile_round.sh), for the same workload only.  Static, not measured
    in this run: the JSON says so in roofline.traffic_source."""
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(prof):
        return None, None
    try:
        tj = json.load(open(prof))
        entry = tj.get("workloads", {}).get(workload) or (tj if workload == "chr20" else None)
        if entry and entry.get("reads") == reads and entry.get("log2_bits") == log2_bits:
            # (a slab-wise stage launches its kernel once per slab: the stage's traffic is that of all its launches of a step)
            per_step = {k: v.get("hbm_bytes_per_step", v.get("hbm_bytes_per_launch")) for k, v in entry.get("kernels", {}).items()}
            src = "profiles/traffic.json (static: committed rocprofv3 --pmc passes of this workload, not this run)"
            prof_rps = entry.get("reads_per_step")
            if reads_per_step and prof_rps and abs(reads_per_step - prof_rps) > 0.005 * prof_rps:
                # the box of this run fits another batch than the profiled one: everything a step moves follows the batch
                # but the sweep of the set by the membership kernels (2^log2_bits / 8 bytes, read; the insert writes it back too)
                sweep = {"seg_probe": (1 << log2_bits) / 8.0, "seg_exact_probe": (1 << log2_bits) / 8.0, "seg_insert": 2.0 * (1 << log2_bits) / 8.0}
                ratio = reads_per_step / float(prof_rps)
                per_step = {k: (max(v - sweep.get(k, 0.0), 0.0) * ratio + min(sweep.get(k, 0.0), v)) if v else v for k, v in per_step.items()}
                src += "; scaled from the profiled batch of %d reads per step to this run's %d" % (prof_rps, reads_per_step)
            return per_step, src
    except Exception:
        pass
    return None, None


def cpu_baseline(dk, eng, kset, gcfg, args, wl, sample_reads, accum_windows=0, slabs=0):
    """Time the CPU oracle ("port": the build's C restatement, OpenMP over reads on the host cores
    this process may use) on a bounded sample of the same child workload against the same filter,
    and cross-check the GPU on that sample.  accum_windows > 0 (wgs): the sample goes through the path that produced
    `value` -- dk_accum_add with the bucketed family forced (a sample this small would take the direct family in AUTO
    mode), the same number of level-2 slabs, packed accumulator units, every hash window, then dk_accum_finish."""
    import numpy as np
    from oracle import orc
    filt = kset.to_host()
    sb = dk.ReadBatch.synth(eng, gcfg, 2, 0, sample_reads)
    bases, mask, _ = sb.download()
    seq, off = orc.unpack_fixed(bases, mask, sample_reads, wl["read_len"])
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box grants this job a 16-core CPU share whatever the affinity mask says
    cores = max(1, min(cores, args.cpu_threads if args.cpu_threads > 0 else 16))
    t0 = time.perf_counter()
    km, cn, st = orc.bloom_probe(filt, args.log2_bits, args.n_hashes, args.seed, wl["k"], True, seq, off,
                                 min_count=1, n_threads=cores)
    dt = time.perf_counter() - t0
    t_probe, t_sort, t_merge = orc.last_phase_seconds()
    if accum_windows:
        eng.set_option("mode", 2)
        eng.set_option("slabs", slabs)
        acc = dk.ChildAccumulator(eng, kset, capacity_records=int(1.3 * st["n_absent"] / accum_windows) + 4096,
                                  window_count=accum_windows)
        his, los, cnts, path = [], [], [], []
        for w in range(accum_windows):
            acc.reset(w)
            acc.add(sb)
            path = [n for n, _ in eng.timings()["stages"]] + ["slabs=%d" % eng.info("plan_slabs"), "record_bytes=%d" % acc.geometry()[2]]
            r = acc.finish(min_count=1)
            h, l, c = r.to_host(sort=False)
            his.append(h)
            los.append(l)
            cnts.append(c)
            r.close()
        acc.close()
        eng.set_option("mode", 0)
        eng.set_option("slabs", 0)
        hi, lo, cnt = np.concatenate(his), np.concatenate(los), np.concatenate(cnts)
        order = np.lexsort((lo, hi))
        hi, lo, cnt = hi[order], lo[order], cnt[order]
    else:
        res = dk.KmerCounter(eng).child_only(sb, kset)
        path = [n for n, _ in eng.timings()["stages"]]
        hi, lo, cnt = res.to_host()
        res.close()
    ok = bool(np.array_equal(lo, km["lo"]) and np.array_equal(hi, km["hi"]) and np.array_equal(cnt, cn))
    sb.close()
    del filt
    return {"value": st["n_windows"] / t_probe / 1e9, "unit": "Gk-mers/s", "cores": cores, "kind": "port",
            "sample": f"first {sample_reads} child reads of the same workload ({st['n_windows']} windows), oracle/dk_oracle.c with "
                      f"{cores} OpenMP thread(s) against the same {kset.n_bytes >> 20} MiB filter; value = windows / {t_probe:.1f} s "
                      f"of extraction + hashing + filter probe (parallel); the oracle's exact counting of the absent k-mers took "
                      f"{t_sort:.1f} s (parallel sort) + {t_merge:.1f} s (serial merge) more, {dt:.1f} s for the whole call",
            "probe_seconds": t_probe, "sort_seconds": t_sort, "merge_seconds": t_merge, "whole_call_seconds": dt,
            "value_including_counting": st["n_windows"] / dt / 1e9,
            "gpu_matches_oracle_on_sample": ok, "gpu_path_of_the_sample": path}


T_START = time.perf_counter()


def progress(msg):
    """phase marks on stderr (a full-genome run is silent for minutes otherwise)"""
    print("[bench %7.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def run_probe_workload(dk, torch, name, args, steps, warmup):
    """One of the single-GPU probe workloads (chr20 = configs[1], ont = configs[4]) on cuda:0 with an engine of its own:
    parents inserted, the child resident in HBM, `warmup` + `steps` dk_probe calls timed as the default run times its
    steps.  -> the entry bench.py attaches under "other_workloads" when the default (wgs) run has finished."""
    wl = WORKLOADS[name]
    k, L, reads = wl["k"], wl["read_len"], wl["reads"]
    rec_bytes = 16 if k > 32 else 8
    genome_len = max(reads * L // 30, 4 * L)
    gcfg = dk.synth_config(seed=args.seed, genome_len=genome_len, read_len=L, err_rate=wl["err"])
    filter_bytes = (1 << wl["log2_bits"]) // 8
    with dk.Engine(k=k, filter_log2_bits=wl["log2_bits"], n_hashes=args.n_hashes, seed=args.seed, device_id=torch.cuda.current_device(),
                   mode=args.mode) as eng:
        kset = dk.KmerSet(eng)
        insert = {}
        for smp in (0, 1):
            pb = dk.ReadBatch.synth(eng, gcfg, smp, 0, reads)
            ist = kset.insert_reads(pb)
            t = eng.timings()
            insert = {"ms": t["total_ms"], "stages_ms": {n: ms for n, ms in t["stages"]}, "gkmers_s": ist["n_windows"] / (t["total_ms"] * 1e-3) / 1e9}
            pb.close()
        child = dk.ReadBatch.synth(eng, gcfg, 2, 0, reads)
        counter = dk.KmerCounter(eng)
        for _ in range(warmup):
            counter.child_only(child, kset).close()
        torch.cuda.synchronize()
        stage_sum, windows, st = {}, 0, None
        t0 = time.perf_counter()
        for _ in range(steps):
            r = counter.child_only(child, kset)
            st = r.stats
            windows += st["n_windows"]
            for n, ms in eng.timings()["stages"]:
                stage_sum[n] = stage_sum.get(n, 0.0) + ms
            r.close()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        stages = {n: ms / steps for n, ms in stage_sum.items()}
        sb = {n: stage_algorithmic_bytes(n, st, filter_bytes, rec_bytes, 1) for n in stages}
        traffic, tsrc = committed_traffic(name, reads, wl["log2_bits"], 1)
        out = {"config": "%s: k=%d, %d x %d bp reads per sample, parent Bloom 2^%d bits, min_count %d"
                         % (wl["cfg"], k, reads, L, wl["log2_bits"], wl["min_count"]),
               "value": windows / elapsed / 1e9, "unit": "Gk-mers/s", "steps": steps, "warmup": warmup,
               "ms_per_step": elapsed / steps * 1e3, "stages_ms": stages, "roofline": roofline_of(stages, sb, traffic, tsrc),
               "pass_stats": st, "parent_insert": insert}
        child.close()
        kset.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="wgs", choices=sorted(WORKLOADS))
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--reads", type=int, default=0, help="reads per sample (wgs: of the whole job; chr20/ont: per GPU)")
    ap.add_argument("--batch", type=int, default=0, help="child reads per batch (one dk_accum_add / dk_probe call)")
    ap.add_argument("--parent-batch", type=int, default=0, help="parent reads per batch (one dk_set_insert call)")
    ap.add_argument("--log2-bits", type=int, default=0, help="filter size; 0 = the workload's (chr20/ont: + ceil(log2(gpus)))")
    ap.add_argument("--windows", type=int, default=0, help="wgs: hash-window passes over the child")
    ap.add_argument("--min-count", type=int, default=0)
    ap.add_argument("--n-hashes", type=int, default=4)
    ap.add_argument("--seed", type=int, default=20260313)
    ap.add_argument("--mode", default="auto", choices=["auto", "direct", "bucketed"])
    ap.add_argument("--set-kind", default="bloom", choices=["bloom", "exact"],
                    help="bloom: the headline metric's parent Bloom filter; exact: exact parent set (side measurement, chr20 only)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline; 0 = min(cores this process may use, 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="wgs: skip the full child run after the timed steps")
    ap.add_argument("--no-reserve", action="store_true", help="wgs: no arena (dk_engine_reserve): the grow-only pool allocates inside the first batches")
    ap.add_argument("--no-other-workloads", action="store_true", help="wgs, one GPU: skip the short chr20 / ont runs attached under other_workloads")
    ap.add_argument("--other-steps", type=int, default=4, help="timed steps of each of the other workloads")
    ap.add_argument("--no-ingest", action="store_true", help="wgs, one GPU: skip the PCIe-inclusive steps (batches uploaded from pinned host memory)")
    ap.add_argument("--ingest-steps", type=int, default=6, help="steps of the PCIe-inclusive measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (collectives staged through the host)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option for A/B runs (dk_engine_set_option), e.g. --opt repart_plain=1")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    import denovo_kmer_amd as dk
    from denovo_kmer_amd.dist import accum_exchange_finish, comm_init_from_torch, merge_counts_device, filter_digest, local_reduce_fn, or_allreduce_

    wl = dict(WORKLOADS[args.workload])
    for key, val in (("k", args.k), ("reads", args.reads), ("batch", args.batch), ("parent_batch", args.parent_batch),
                     ("windows", args.windows), ("min_count", args.min_count)):
        if val:
            wl[key] = val
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
    sdev = "cpu" if on_host else dev

    wgs = args.workload == "wgs"
    k, L = wl["k"], wl["read_len"]
    rec_bytes = 16 if k > 32 else 8
    if args.set_kind == "exact" and args.workload != "chr20":
        raise SystemExit("--set-kind exact is a chr20 side measurement")
    if args.log2_bits == 0:
        args.log2_bits = wl["log2_bits"] if wgs else (36 if args.set_kind == "exact" else wl["log2_bits"]) + max(0, (world - 1).bit_length())
    # wgs: the job is one trio whose reads are dealt to the ranks; chr20 / ont: every rank brings its own reads of an N x genome
    reads_total = wl["reads"] if wgs else wl["reads"] * world
    reads_rank = reads_total // world
    genome_len = max(reads_total * L // 30, 4 * L)
    gcfg = dk.synth_config(seed=args.seed, genome_len=genome_len, read_len=L, err_rate=wl["err"])
    eng = dk.Engine(k=k, filter_log2_bits=args.log2_bits, n_hashes=args.n_hashes, seed=args.seed,
                    device_id=local_rank, mode=args.mode, rank=rank, world_size=world, set_kind=args.set_kind)
    for ov in args.opt:
        name, _, val = ov.partition("=")
        eng.set_option(name, int(val))
    # ---- device memory: one arena for everything the engine allocates (dk_engine_reserve) ----------------------------------
    # The filter is a torch tensor (the torch.distributed fallback of the all-reduce needs it to be); everything else --
    # workspaces, read batches, the accumulator, result tables -- is carved from ONE arena reserved now, so that the one slow
    # hipMalloc of the run (seconds for ~200 GB) is paid here, where a real host would be opening its input files, and not
    # inside the first parent batch.  --no-reserve restores round 2's behaviour (grow-only pool, first batch pays).
    filter_bytes = (1 << args.log2_bits) // 8
    filt = torch.zeros(filter_bytes // 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    reserve_s, reserved = 0.0, 0
    native_possible = world == 1 or (args.backend == "nccl" and not args.single_device)
    if wgs and not args.no_reserve and native_possible:
        free, _ = torch.cuda.mem_get_info(dev)
        # torch keeps a few GB for digests / scalars; with several ranks two RCCL communicators (torch's for the scalars,
        # the library's for the set and the accumulator) bring their own channel buffers
        reserved = max(0, free - ((5 if world == 1 else 20) << 30))
        t0 = time.perf_counter()
        try:
            eng.reserve(reserved)
            reserve_s = time.perf_counter() - t0
            progress("arena of %.1f GB reserved in %.2f s" % (reserved / 1e9, reserve_s))
        except dk.DkError as exc:                            # one block of that size was refused: the grow-only pool takes over
            progress("arena of %.1f GB refused (%s): allocating as needed" % (reserved / 1e9, exc))
            reserved = 0

    def free_bytes():
        if reserved:
            return eng.info("pool_bytes_reserved") - eng.info("pool_bytes_in_use")
        return torch.cuda.mem_get_info(dev)[0]

    def fit_batch(want, bytes_per_read, share, reserve=0):
        """reads per batch that the free device memory allows (the partition workspace and the resident reads scale with the
        batch; a larger batch amortises the sweep of the set better): `want` unless `share` of the free bytes (less
        `reserve`, set aside for later allocations) is less"""
        fit = int(share * max(free_bytes() - reserve, 0) / bytes_per_read) // 1_000_000 * 1_000_000
        return max(4_000_000, min(want, fit)) if wgs else want

    # parent insert, slab-wise: the level-1 pieces (9.2 bytes per window) + the overflow list + the reads themselves
    win_per_read = max(L - k + 1, 1)
    ws_per_read = 9.8 * win_per_read * (rec_bytes / 8.0) + 57
    pbatch = min(fit_batch(wl["parent_batch"], ws_per_read, 0.9), reads_rank)
    batch = min(wl["batch"], reads_rank)
    n_batches = (reads_rank + batch - 1) // batch
    n_pbatches = (reads_rank + pbatch - 1) // pbatch

    def set_hint(b):
        # a batch covers the genome b*L/genome_len times: capacity planning of the bucket regions (exact for any value)
        eng.set_option("multiplicity_hint", max(2, int(2 * b * L / genome_len) + 1) if b < reads_rank else 0)
    set_hint(pbatch)
    kset = dk.KmerSet(eng, device_ptr=filt.data_ptr(), keepalive=filt)
    if args.set_kind == "exact":
        kset.clear()                         # an empty exact set is not all-zero memory
    first = rank * reads_rank

    def batch_range(b, size=None):
        size = size or batch
        lo = first + b * size
        return lo, min(size, first + reads_rank - lo)

    # ---- parent build (reported, not part of `value`): every batch of both parents ------------------
    insert_ms, insert_windows, insert_stages, t_par = 0.0, 0, {}, time.perf_counter()
    parent_windows, parent_dev_ms, parent_gen_s, parent_batch_ms = 0, 0.0, 0.0, []
    progress("parent build: 2 x %d batches of %d reads into a 2^%d-bit set" % (n_pbatches, pbatch, args.log2_bits))
    for s in (0, 1):
        for b in range(n_pbatches):
            lo, n = batch_range(b, pbatch)
            tg = time.perf_counter()
            pb = dk.ReadBatch.synth(eng, gcfg, s, lo, n)
            parent_gen_s += time.perf_counter() - tg
            st = kset.insert_reads(pb)
            t = eng.timings()
            parent_windows += st["n_windows"]
            parent_dev_ms += t["total_ms"]
            parent_batch_ms.append(round(t["total_ms"], 2))
            if n == pbatch:                  # rate of a full batch, not the first one (which also grows the workspace pool)
                insert_ms, insert_windows = t["total_ms"], st["n_windows"]
                insert_stages = {name: ms for name, ms in t["stages"]}
                insert_stats = st
            pb.close()
    torch.cuda.synchronize()
    parent_seconds = time.perf_counter() - t_par
    progress("parent build done in %.1f s" % parent_seconds)
    freed = eng.trim()                       # the child pass uses other workspace shapes: hand the parents' back first
    progress("workspace of the parent build returned: %.1f GB" % (freed / 1e9))
    set_hint(batch)
    allreduce_ms, allreduce_bytes, allreduce_path = 0.0, 0, None
    native = False
    # a library named by DK_RCCL_LIBRARY is NOT librccl (tests/rccl_shim in the one-GPU rehearsal): the labels say so
    rccl_note = (" [DK_RCCL_LIBRARY=%s stands in for librccl]" % os.path.basename(os.environ["DK_RCCL_LIBRARY"])) if os.environ.get("DK_RCCL_LIBRARY") else ""
    if world > 1:
        # native path: the library's own RCCL communicator (dk_comm_init + dk_set_allreduce_or), as a host without torch
        # would run it; the gloo rehearsal (several ranks on one GPU) and any failure to set it up use the
        # torch.distributed composition of the same three steps
        # (DK_RCCL_LIBRARY set in a single-device rehearsal: the library's collectives over tests/rccl_shim, several ranks on one GPU)
        if (not on_host and not args.single_device) or (args.single_device and os.environ.get("DK_RCCL_LIBRARY")):
            try:
                comm_init_from_torch(eng)
                native = True
            except Exception as exc:                       # noqa: BLE001 -- reported, the torch path takes over
                progress("native RCCL communicator unavailable (%s): torch.distributed path" % exc)
        flags = torch.tensor([1 if native else 0], dtype=torch.int64, device=sdev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        native = bool(int(flags.item()))
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if native:
            allreduce_bytes = kset.allreduce_or()
            allreduce_path = "dk_set_allreduce_or (RCCL send/recv + HIP OR kernel + ncclAllGather on the engine stream)" + rccl_note
        else:
            allreduce_bytes = or_allreduce_(filt, local_reduce_fn(eng), stage_through_cpu=on_host)
            allreduce_path = "torch.distributed all_to_all_single + dk_or_reduce_slices + all_gather_into_tensor" + (" staged through the host (gloo rehearsal)" if on_host else "")
        torch.cuda.synchronize()
        dist.barrier()
        allreduce_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.empty_cache()             # the torch path stages a second copy of the set: hand it back to the device
    # every rank must now hold the same filter: compare a digest of the words (XOR/sum of mixed words) across ranks
    popc = kset.popcount()
    digest = filter_digest(filt)
    filter_consistent = True
    if world > 1:
        dg = torch.tensor([digest, -digest, popc, -popc], dtype=torch.int64, device=sdev)
        dist.all_reduce(dg, op=dist.ReduceOp.MAX)
        filter_consistent = bool(int(dg[0]) == -int(dg[1]) and int(dg[2]) == -int(dg[3]))

    # ---- child membership pass: warmup + K timed steps -----------------------------------------------
    stage_sum, total_dev_ms, step_stats = {}, 0.0, None
    finish_stats, windows_timed, n_flushes, step_ms = None, 0, 0, []
    if wgs:
        # hash-window passes: one -- the occurrences of the whole hash space fit beside the filter as 6-byte records (and the
        # multi-GPU exchange is in place, so N > 1 needs no second copy); --windows 2 = round 2's two passes
        R = args.windows or wl["windows"]
        # expected absent occurrences per pass: the absent rate of the child's first 4 M reads (what a host sees after its
        # first batch), 3 % on top; units have 5 sigma of room above that and spill to an overflow list of capacity / 64
        # records.  (The analytic bound -- windows with >= 1 error base -- is 7.6 % above what this generator yields: 7 GB
        # of accumulator that the batch, i.e. the amortisation of the filter sweep, can use.)
        pn = min(4_000_000, reads_rank)
        pb = dk.ReadBatch.synth(eng, gcfg, 2, first, pn)
        pr = dk.KmerCounter(eng).child_only(pb, kset)
        absent_rate = pr.stats["n_absent"] / max(pr.stats["n_windows"], 1)
        pr.close()
        pb.close()
        if world > 1:
            ar = torch.tensor([absent_rate], dtype=torch.float64, device=sdev)
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)        # every rank must build the same accumulator
            absent_rate = float(ar.item())
        p_err = 1.0 - (1.0 - wl["err"]) ** k
        cap = int(min(1.03 * absent_rate, 1.0 * p_err + 0.01) * reads_rank * (L - k + 1) / R)
        acc = dk.ChildAccumulator(eng, kset, capacity_records=cap, window_index=0, window_count=R)
        # the counting pass's result table comes from the same memory (sized from the capacity: a sixteenth of cap / min_count
        # entries of 12 bytes, + slack); the torch.distributed fallback of the exchange receives a second copy of the store
        table_bytes = int(cap / max(wl["min_count"], 1) / (16 if wl["min_count"] > 1 else 1) * 1.13 * 12) + (64 << 20)
        later = table_bytes + (acc.device_bytes() * 1.05 if world > 1 and not native_possible else 0)
        # child batch: level-1 pieces + overflow list per read of the batch (windowed passes: 1 / R of the records), plus the
        # reads of the resident batches (57 bytes each).  At least `n_keep` batches stay resident for the timed steps; more
        # if they fit (up to one per step: steps beyond that revisit the resident batches after a counting pass + reset)
        n_keep = max(1, min(8, args.warmup + args.steps))
        slab_room = 9 << 30                  # one slab of level-2 regions (the engine keeps it below 12 GiB)
        batch = min(fit_batch(wl["batch"], (ws_per_read - 57) / R + 57 * n_keep, 0.97, reserve=later + slab_room), reads_rank)
        if world > 1:
            # the counting passes inside the loops below are collective: every rank must walk the same number of batches
            bt = torch.tensor([batch], dtype=torch.int64, device=sdev)
            dist.all_reduce(bt, op=dist.ReduceOp.MIN)
            batch = int(bt.item())
        n_batches = (reads_rank + batch - 1) // batch
        batch = (reads_rank + n_batches - 1) // n_batches      # equal batches (no short last one: 150 M reads per rank = 4 x 37.5 M)
        set_hint(batch)
        room = free_bytes() - later - slab_room - batch * (ws_per_read - 57) / R
        n_res = int(max(n_keep, min(n_batches, args.warmup + args.steps, room // (batch * 57))))
        n_res = min(n_res, n_batches)
        if world > 1:
            nr = torch.tensor([n_res], dtype=torch.int64, device=sdev)
            dist.all_reduce(nr, op=dist.ReduceOp.MIN)
            n_res = int(nr.item())
        resident = []
        for b in range(n_res):
            lo, n = batch_range(b)
            resident.append(dk.ReadBatch.synth(eng, gcfg, 2, lo, n))
        progress("child: %d resident batches, accumulator of %.1f GB" % (n_res, acc.device_bytes() / 1e9))
        finish_stages, n_flushes = {}, 0

        exchange_bytes = [0]

        def count_accumulated():
            # counting of what was accumulated is part of the job; N > 1: the ranks first swap unit ranges, so that counts
            # and min_count are exact across the read shards (each rank ends with its share of the hash space) -- natively
            # (dk_accum_exchange_finish: in place on the library's communicator) or, in the gloo rehearsal, over torch.distributed
            if world > 1 and native:
                res = acc.exchange_finish(min_count=wl["min_count"])
                exchange_bytes[0] += res.bytes_sent
                return res
            return accum_exchange_finish(acc, min_count=wl["min_count"], stage_through_cpu=on_host)

        def flush():
            res = count_accumulated()
            st, t = res.stats, eng.timings()
            res.close()
            return st, t

        # The steps walk the rank's child batches like a real pass does: when the pass is complete (every resident batch
        # added once: only with few batches per rank, i.e. many GPUs) it is counted and the accumulator starts over
        for j in range(args.warmup):
            acc.add(resident[j % n_res])
            if (j + 1) % n_res == 0:
                flush()
                acc.reset(0)
        flush()                              # warm-up of the counting stage too (its table comes from the workspace pool)
        acc.reset(0)                         # the timed steps start on an empty accumulator, at the batch after the warm-up's last
        since = 0                            # batches added since the last reset
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            j = args.warmup + i
            st = acc.add(resident[j % n_res])
            step_stats = st if step_stats is None else {key: step_stats[key] + st[key] for key in st}
            windows_timed += st["n_windows"]
            t = eng.timings()
            total_dev_ms += t["total_ms"]
            step_ms.append([round(ms, 2) for _, ms in t["stages"]])
            for name, ms in t["stages"]:
                stage_sum[name] = stage_sum.get(name, 0.0) + ms
            since += 1
            if since == n_res or i == args.steps - 1:
                since = 0
                fst, t = flush()
                n_flushes += 1
                finish_stats = fst if finish_stats is None else {key: finish_stats[key] + fst[key] for key in fst}
                total_dev_ms += t["total_ms"]
                for name, ms in t["stages"]:
                    finish_stages[name] = finish_stages.get(name, 0.0) + ms
                if i != args.steps - 1:
                    acc.reset(0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        windows_done = windows_timed / R                     # a step completes 1/R of its batch's membership work
        step_stats = {key: v / args.steps for key, v in step_stats.items()}      # the average step (batches may differ in size)
        for rb in resident:
            rb.close()
    else:
        R = 1
        child = dk.ReadBatch.synth(eng, gcfg, 2, first, reads_rank)
        counter = dk.KmerCounter(eng)
        for _ in range(args.warmup):
            r = counter.child_only(child, kset)
            r.close()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = counter.child_only(child, kset)
            step_stats = r.stats
            windows_timed += r.stats["n_windows"]
            t = eng.timings()                       # HIP events on the engine's stream, this step
            total_dev_ms += t["total_ms"]
            for name, ms in t["stages"]:
                stage_sum[name] = stage_sum.get(name, 0.0) + ms
            r.close()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        windows_done = windows_timed
        finish_stages = {}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=sdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ww = torch.tensor([windows_done], dtype=torch.float64, device=sdev)
        dist.all_reduce(ww, op=dist.ReduceOp.SUM)
        windows_all = float(ww.item())
    else:
        windows_all = windows_done

    # ---- wgs: the whole child end to end (all passes, all batches, counting), after the timed steps ----------
    e2e = None
    progress("timed steps done")
    if wgs and not args.no_end_to_end:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_child_only, n_absent_all, child_windows, gen_s = 0, 0, 0, 0.0
        add_dev_ms, finish_dev_ms, finish_stage_ms, finish_wall_s = 0.0, 0.0, {}, 0.0
        for w in range(R):
            acc.reset(w)
            for b in range(n_batches):
                lo, n = batch_range(b)
                tg = time.perf_counter()
                cb = dk.ReadBatch.synth(eng, gcfg, 2, lo, n)      # generated in place (a real host would upload packed reads here)
                gen_s += time.perf_counter() - tg
                st = acc.add(cb)
                add_dev_ms += eng.timings()["total_ms"]
                if w == 0:
                    child_windows += st["n_windows"]
                cb.close()
            tc = time.perf_counter()
            res = count_accumulated()
            tf = eng.timings()
            finish_dev_ms += tf["total_ms"]
            finish_wall_s += time.perf_counter() - tc
            for name, ms in tf["stages"]:
                finish_stage_ms[name] = finish_stage_ms.get(name, 0.0) + ms
            progress("end to end: pass %d of %d counted (%.0f ms on the device, %.0f ms wall)"
                     % (w + 1, R, tf["total_ms"], (time.perf_counter() - tc) * 1e3))
            n_child_only += len(res)
            n_absent_all += res.stats["n_absent"]
            res.close()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        child_seconds = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([child_seconds, gen_s, parent_seconds], dtype=torch.float64, device=sdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            child_seconds, gen_s, parent_seconds = (float(x) for x in tt)
            cc = torch.tensor([child_windows, n_child_only, n_absent_all, parent_windows], dtype=torch.float64, device=sdev)
            dist.all_reduce(cc, op=dist.ReduceOp.SUM)
            child_windows, n_child_only, n_absent_all, parent_windows = (float(x) for x in cc)
        e2e = {"child_reads": reads_total, "child_windows": child_windows, "child_seconds": child_seconds,
               "child_seconds_without_read_generation": child_seconds - gen_s,
               "child_gkmers_s": child_windows / (child_seconds - gen_s) / 1e9,
               "hash_windows": R, "batches_per_pass": n_batches, "min_count": wl["min_count"],
               "device_seconds_adds": add_dev_ms * 1e-3, "device_seconds_counting": finish_dev_ms * 1e-3,
               "counting_stages_ms": finish_stage_ms, "wall_seconds_counting": finish_wall_s,
               "absent_occurrences": n_absent_all, "child_only_kmers": n_child_only,
               "trio_seconds": reserve_s + parent_seconds + allreduce_ms * 1e-3 + child_seconds,
               "trio_gkmers_s": (parent_windows + child_windows) / (reserve_s + parent_seconds + allreduce_ms * 1e-3 + child_seconds) / 1e9,
               "arena_reserve_seconds_included": reserve_s,
               "parent_reads": 2 * reads_total, "parent_windows": parent_windows, "parent_seconds": parent_seconds,
               "note": "read generation (synthetic, on the GPU) is inside child_seconds / parent_seconds and subtracted for child_gkmers_s; "
                       + ("the ranks swap accumulator unit ranges before counting (dist.accum_exchange_finish): counts are exact across "
                          "the read shards, every rank keeps its share of the hash space" if world > 1 else "one GPU")}
    ingest = None
    if wgs:
        acc_rec_bytes = acc.geometry()[2]
        timed_slabs = eng.info("plan_slabs")
        if world == 1 and not args.no_ingest and args.ingest_steps > 0:
            # PCIe-inclusive rate (reported beside `value`, never instead of it): the same steps with every batch arriving from
            # pinned host memory -- the upload of batch i + 1 (dk_reads_from_packed_async, copy stream) overlaps the kernels of
            # batch i.  Every step brings a different batch (one pinned host buffer each: a batch met twice would make every
            # absent k-mer a duplicate, which is not what a sample looks like); the accumulator starts empty.
            args.ingest_steps = min(args.ingest_steps, n_batches)
            progress("ingest from the host: %d steps" % args.ingest_steps)
            n_host = args.ingest_steps
            host, meta = [], []
            for b in range(n_host):
                lo, n = batch_range(b)
                rb = dk.ReadBatch.synth(eng, gcfg, 2, lo, n)
                bases, mask, n_bases = rb.download()
                pp = dk.PinnedPacked(n_bases)
                pp.bases[:] = bases
                pp.mask[:] = mask
                host.append(pp)
                meta.append((n_bases, n, rb.stats()["n_windows"]))
                rb.close()
                del bases, mask
            acc.reset(0)
            # one untimed step (the copy stream and its buffers come up), then the timed ones
            w0 = dk.ReadBatch.from_packed_async(eng, host[0], *meta[0])
            acc.add(w0)
            w0.close()
            acc.reset(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nxt = dk.ReadBatch.from_packed_async(eng, host[0], *meta[0])
            up_bytes, in_windows, dev_ms = 0, 0, 0.0
            for i in range(args.ingest_steps):
                cur = nxt
                if i + 1 < args.ingest_steps:
                    nxt = dk.ReadBatch.from_packed_async(eng, host[(i + 1) % n_host], *meta[(i + 1) % n_host])
                st = acc.add(cur)
                dev_ms += eng.timings()["total_ms"]
                ingest_stages = {nm: round(ms, 2) for nm, ms in eng.timings()["stages"]}
                progress("ingest step %d: %s" % (i, ingest_stages))
                cur.close()
                in_windows += st["n_windows"]
                up_bytes += (host[i % n_host].n_bwords + host[i % n_host].n_mwords) * 8
            res = acc.finish(min_count=wl["min_count"])
            dev_ms += eng.timings()["total_ms"]
            res.close()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # the same uploads alone (nothing to overlap with): what the link delivers
            t1 = time.perf_counter()
            for i in range(min(3, args.ingest_steps)):
                rb = dk.ReadBatch.from_packed_async(eng, host[i % n_host], *meta[i % n_host])
                rb.wait()
                rb.close()
            link_gbs = min(3, args.ingest_steps) * up_bytes / args.ingest_steps / (time.perf_counter() - t1) / 1e9
            ingest = {"gkmers_s_pcie_inclusive": in_windows / R / dt / 1e9, "steps": args.ingest_steps, "ms_per_step": dt / args.ingest_steps * 1e3,
                      "device_ms_per_step": dev_ms / args.ingest_steps, "bytes_uploaded_per_step": up_bytes / args.ingest_steps,
                      "upload_alone_gbs": link_gbs, "reads_per_step": batch, "stages_ms_of_the_last_step": ingest_stages,
                      "note": "packed reads (2-bit bases + 1-bit flags, 56.6 bytes per 150-bp read) in pinned host memory, uploaded with "
                              "dk_reads_from_packed_async one batch ahead of the batch being accumulated; counting included; "
                              "hash_windows > 1 would upload every batch once per window"}
            for pp in host:
                pp.close()
        acc.close()
    elif world > 1:
        # chr20 / ont over several GPUs, end to end: one more child pass per rank + the cross-rank sum of the per-rank
        # tables on the device (every rank ends with the whole child-only table)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = counter.child_only(child, kset)
        merged = merge_counts_device(eng, r, min_count=wl["min_count"], stage_through_cpu=on_host)
        torch.cuda.synchronize()
        dist.barrier()
        child_seconds = time.perf_counter() - t0
        tt = torch.tensor([child_seconds, parent_seconds], dtype=torch.float64, device=sdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        child_seconds, parent_seconds = (float(x) for x in tt)
        cc = torch.tensor([r.stats["n_windows"], parent_windows], dtype=torch.float64, device=sdev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        child_windows, parent_windows = (float(x) for x in cc)
        e2e = {"child_windows": child_windows, "child_seconds_probe_plus_cross_rank_merge": child_seconds,
               "child_only_kmers_after_merge": len(merged), "parent_windows": parent_windows, "parent_seconds": parent_seconds,
               "trio_seconds": parent_seconds + allreduce_ms * 1e-3 + child_seconds,
               "trio_gkmers_s": (parent_windows + child_windows) / (parent_seconds + allreduce_ms * 1e-3 + child_seconds) / 1e9,
               "note": "parent insert (incl. synthetic read generation) + set all-reduce + child pass + merge_counts_device over all ranks"}
        merged.close()
        r.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = windows_all / elapsed / 1e9
        stages = {n: ms / args.steps for n, ms in stage_sum.items()}
        arb = acc_rec_bytes if wgs else None               # accumulate mode: the absent records are the accumulator's (6 bytes packed)
        sb = {n: stage_algorithmic_bytes(n, step_stats, filter_bytes, rec_bytes, R, arb) for n in stages}
        if finish_stats:
            for n, ms in finish_stages.items():            # once per K steps: amortised per step like its time
                stages[n] = ms / args.steps
                b = stage_algorithmic_bytes(n, finish_stats, filter_bytes, rec_bytes, R, arb)
                sb[n] = b / args.steps if b else None
        traffic, tsrc = committed_traffic(args.workload, wl["reads"], args.log2_bits, world, batch if wgs else None)
        slab_launches = {n: timed_slabs for n in ("repart", "seg_probe", "seg_exact_probe")} if wgs else None
        rl = roofline_of(stages, sb, traffic, tsrc, slab_launches)
        desc = {
            "wgs": "configs[2]: k=%d, full 30x WGS synthetic trio, %d x %d bp reads per sample (genome %.2f Gb), parent Bloom 2^%d bits "
                   "(%d GiB) resident in HBM, child streamed in %d-read batches through %d hash-window passes (dk_accum_add), "
                   "min_count %d" % (k, reads_total, L, genome_len / 1e9, args.log2_bits, filter_bytes >> 30, batch, R, wl["min_count"]),
            "chr20": "configs[1]: k=%d, chr20-scale 30x synthetic trio, %d x %d bp reads per sample per GPU, genome %d Mb, parent %s 2^%d "
                     "bits resident in HBM" % (k, reads_rank, L, genome_len >> 20,
                                               "Bloom" if args.set_kind == "bloom" else "exact set (open-addressing tables)", args.log2_bits),
            "ont": "configs[4]: k=%d, ONT-style synthetic trio, %d x %d bp reads per sample per GPU (%.0f %% errors), genome %d Mb, parent "
                   "Bloom 2^%d bits resident in HBM" % (k, reads_rank, L, 100 * wl["err"], genome_len >> 20, args.log2_bits),
        }[args.workload]
        out = {
            "metric": "Gk-mers/sec (child reads vs parent %s), k=%d" % ("Bloom" if args.set_kind == "bloom" else "exact set", k),
            "value": value, "unit": "Gk-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            # wgs deals ONE trio to the ranks (the total work is fixed); chr20 / ont bring N times the genome
            "scaling": "strong" if wgs else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "k": k, "reads_per_sample": reads_total, "reads_per_gpu": reads_rank,
                       "reads_per_step": batch, "read_len": L, "filter_log2_bits": args.log2_bits, "n_hashes": args.n_hashes,
                       "mode": args.mode, "set_kind": args.set_kind, "hash_windows": R, "engine_options": args.opt,
                       "windows_counted_per_step": "n_windows / hash_windows",
                       "counting_passes_inside_timed_region": n_flushes if wgs else None,
                       "resident_child_batches": n_res if wgs else 1,
                       "level2_slabs": timed_slabs if wgs else eng.info("plan_slabs"), "level1_bits": eng.info("plan_b1"), "level2_bits": eng.info("plan_b2"),
                       "sub_segment_split_bits": eng.info("plan_sbits"),
                       "accumulator_record_bytes": acc_rec_bytes if wgs else None,
                       "accumulator_capacity_records": cap if wgs else None, "absent_rate_of_first_reads": absent_rate if wgs else None,
                       "arena_bytes": reserved, "arena_reserve_seconds": reserve_s,
                       "device_bytes_peak_of_the_engine": eng.info("pool_bytes_peak"),
                       "parallelism": "reads sharded x%d, %s-all-reduce of parent set" % (world, "OR" if args.set_kind == "bloom" else "union")},
            "roofline": rl,
            "stages_ms": stages,
            "device_ms_per_step": total_dev_ms / args.steps, "stage_ms_of_each_step": step_ms,
            "pass_stats": step_stats,
            "parent_build": {"insert_gkmers_s": insert_windows / (insert_ms * 1e-3) / 1e9 if insert_ms else None,
                             "insert_ms_per_batch": insert_ms, "insert_stages_ms": insert_stages, "batches": 2 * n_pbatches,
                             "reads_per_batch": pbatch,
                             "seconds_all_batches_incl_read_generation": parent_seconds,
                             "arena_reserve_seconds": reserve_s,
                             "seconds_all_batches_plus_arena_reserve": parent_seconds + reserve_s,
                             "read_generation_seconds": parent_gen_s, "device_ms_of_each_batch": parent_batch_ms, "device_seconds_all_batches": parent_dev_ms * 1e-3,
                             "gkmers_s_all_batches_device_time": parent_windows / (parent_dev_ms * 1e-3) / 1e9 if parent_dev_ms else None,
                             "or_allreduce_ms": allreduce_ms, "or_allreduce_bytes_per_rank": allreduce_bytes,
                             "or_allreduce_path": allreduce_path,
                             "or_allreduce_gbs_per_rank": allreduce_bytes / (allreduce_ms * 1e-3) / 1e9 if allreduce_ms else None,
                             "xgmi_peak_gbs_per_rank": 7 * XGMI_LINK_GBS,
                             "filter_bits_set": popc, "filter_digest": "%016x" % (digest & (2**64 - 1)),
                             "filter_identical_on_all_ranks": filter_consistent},
        }
        if insert_ms:
            ib = {n: stage_algorithmic_bytes(n, insert_stats, filter_bytes, rec_bytes, 1) for n in insert_stages}
            out["parent_build"]["roofline"] = roofline_of(insert_stages, ib, None, None)
        if finish_stats:
            out["finish_stats"] = finish_stats
        if e2e:
            out["end_to_end"] = e2e
        if ingest:
            out["ingest_host"] = ingest
        if world > 1 and wgs:
            out["config"]["accumulator_exchange"] = ("dk_accum_exchange_finish (RCCL, in place)" + rccl_note if native else "torch.distributed (rehearsal / fallback)")
            out["config"]["accumulator_exchange_bytes_sent_rank0"] = exchange_bytes[0]
        if args.set_kind == "exact":
            out["parent_build"]["exact_set_load"] = popc / (filter_bytes / (16 if k > 32 else 8))
        if world == 1 and not args.no_cpu_baseline and args.set_kind == "bloom":
            default_sample = {"wgs": 3_000_000, "chr20": 6_000_000, "ont": 12_000}[args.workload]
            progress("CPU baseline (oracle on a bounded sample, filter copied to the host)")
            out["cpu_baseline"] = cpu_baseline(dk, eng, kset, gcfg, args, wl, min(args.cpu_sample_reads or default_sample, reads_rank),
                                               accum_windows=R if wgs else 0, slabs=timed_slabs if wgs else 0)

    if not wgs:
        child.close()
    kset.close()
    eng.close()
    if rank == 0:
        if wgs and world == 1 and not args.no_other_workloads:
            # the other two single-GPU configurations, a few steps each, in the same process (the driver times one command)
            del filt, kset
            torch.cuda.empty_cache()
            out["other_workloads"] = {}
            for name in ("chr20", "ont"):
                progress("other workload: %s" % name)
                out["other_workloads"][name] = run_probe_workload(dk, torch, name, args, args.other_steps, 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
