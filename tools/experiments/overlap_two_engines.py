#!/usr/bin/env python3
"""Experiment: does running the VALU-bound scan of one batch beside the HBM-bound level-2 / membership kernels of another pay?

Two engines (two HIP streams) share ONE parent filter (attached device memory); each accumulates its own child batches from
its own host thread, so the kernels of the two pipelines overlap however the hardware schedules them.  Compared with one
engine walking the same batches alone.  configs[1] geometry (2^39-bit filter, k = 31, 150-bp reads), smaller batches so that
two partition workspaces fit.  Prints one JSON line.  

  python tools/experiments/overlap_two_engines.py [--batch 24000000] [--steps 6] [--opt name=value ...]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import denovo_kmer_amd as dk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-bits", type=int, default=39)
    ap.add_argument("--reads", type=int, default=1_200_000_000)
    ap.add_argument("--parent-batch", type=int, default=128_000_000)
    ap.add_argument("--batch", type=int, default=24_000_000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--resident", type=int, default=3)
    ap.add_argument("--delay-ms", type=float, default=0.0, help="the second engine's thread starts this much later")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--opt-b", action="append", default=None, help="options of the second engine (default: the first's)")
    args = ap.parse_args()
    k, L, seed = 31, 150, 20260313
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    genome_len = args.reads * L // 30
    gcfg = dk.synth_config(seed=seed, genome_len=genome_len, read_len=L, err_rate=5e-3)

    def engine(opts):
        e = dk.Engine(k=k, filter_log2_bits=args.log2_bits, n_hashes=4, seed=seed, device_id=0, mode="bucketed")
        for ov in opts:
            name, _, val = ov.partition("=")
            e.set_option(name, int(val))
        return e

    t00 = time.perf_counter()
    eng_a = engine(args.opt)
    filt = torch.zeros((1 << args.log2_bits) // 64, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    set_a = dk.KmerSet(eng_a, device_ptr=filt.data_ptr(), keepalive=filt)
    eng_a.set_option("multiplicity_hint", max(2, int(2 * args.parent_batch * L / genome_len) + 1))
    for s in (0, 1):
        for lo in range(0, args.reads, args.parent_batch):
            pb = dk.ReadBatch.synth(eng_a, gcfg, s, lo, min(args.parent_batch, args.reads - lo))
            set_a.insert_reads(pb)
            pb.close()
    eng_a.trim()
    print("[overlap %.1f s] parents inserted" % (time.perf_counter() - t00), file=sys.stderr, flush=True)
    eng_b = engine(args.opt_b if args.opt_b is not None else args.opt)
    set_b = dk.KmerSet(eng_b, device_ptr=filt.data_ptr(), keepalive=filt)
    hint = max(2, int(2 * args.batch * L / genome_len) + 1)
    eng_a.set_option("multiplicity_hint", hint)
    eng_b.set_option("multiplicity_hint", hint)

    # absent rate from a small probe, as bench.py does
    pb = dk.ReadBatch.synth(eng_a, gcfg, 2, 0, 4_000_000)
    pr = dk.KmerCounter(eng_a).child_only(pb, set_a)
    absent_rate = pr.stats["n_absent"] / pr.stats["n_windows"]
    pr.close()
    pb.close()
    n_total = args.steps + 1
    cap = int(1.06 * absent_rate * args.batch * (L - k + 1) * n_total) + (1 << 22)

    def make(eng, kset, first):
        acc = dk.ChildAccumulator(eng, kset, capacity_records=cap)
        res = [dk.ReadBatch.synth(eng, gcfg, 2, first + i * args.batch, args.batch) for i in range(args.resident)]
        return acc, res

    acc_a, res_a = make(eng_a, set_a, 0)
    acc_b, res_b = make(eng_b, set_b, 600_000_000)

    def run(acc, res, eng, out, key, delay=0.0):
        stages = {}
        acc.add(res[0])                                  # warm-up: workspace from the pool
        if delay:
            time.sleep(delay)
        out[key + "_t0"] = time.perf_counter()
        n_abs = 0
        for i in range(args.steps):
            st = acc.add(res[(i + 1) % len(res)])
            n_abs += st["n_absent"]
            for name, ms in eng.timings()["stages"]:
                stages[name] = stages.get(name, 0.0) + ms
        out[key + "_t1"] = time.perf_counter()
        out[key + "_stages"] = {n: round(ms / args.steps, 2) for n, ms in stages.items()}
        out[key + "_absent"] = n_abs

    windows_step = args.batch * (L - k + 1)
    out = {}
    # one engine alone
    run(acc_a, res_a, eng_a, out, "alone")
    alone_s = out["alone_t1"] - out["alone_t0"]
    acc_a.reset(0)
    print("[overlap %.1f s] alone: %.1f ms per step" % (time.perf_counter() - t00, alone_s / args.steps * 1e3), file=sys.stderr, flush=True)
    # two engines, one thread each
    ta = threading.Thread(target=run, args=(acc_a, res_a, eng_a, out, "a"))
    tb = threading.Thread(target=run, args=(acc_b, res_b, eng_b, out, "b", args.delay_ms * 1e-3))
    ta.start()
    tb.start()
    ta.join()
    tb.join()
    both_s = max(out["a_t1"], out["b_t1"]) - min(out["a_t0"], out["b_t0"])
    line = {
        "batch": args.batch, "steps": args.steps, "delay_ms": args.delay_ms, "options": args.opt, "options_b": args.opt_b,
        "alone_ms_per_step": round(alone_s / args.steps * 1e3, 2), "alone_gkmers_s": round(windows_step * args.steps / alone_s / 1e9, 2),
        "alone_stages_ms": out["alone_stages"],
        "two_engines_wall_ms_per_pair_of_steps": round(both_s / args.steps * 1e3, 2),
        "two_engines_gkmers_s": round(2 * windows_step * args.steps / both_s / 1e9, 2),
        "a_stages_ms": out["a_stages"], "b_stages_ms": out["b_stages"],
        "a_wall_ms_per_step": round((out["a_t1"] - out["a_t0"]) / args.steps * 1e3, 2),
        "b_wall_ms_per_step": round((out["b_t1"] - out["b_t0"]) / args.steps * 1e3, 2),
        "absent_rate": absent_rate, "plan_a": {n: eng_a.info(n) for n in ("plan_b1", "plan_b2", "plan_sbits", "plan_slabs", "plan_scan_variant")},
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
