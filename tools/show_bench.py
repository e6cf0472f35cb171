#!/usr/bin/env python3
"""print the headline fields of a bench.py JSON line: tools/show_bench.py gpurun_out/x.json"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads([l for l in open(path) if l.startswith("{")][0])
    c = d["config"]
    print("%s: %.2f Gk-mers/s, %.2f ms/step, %d reads/step, %s resident, windows %s, slabs %s" % (
        path, d["value"], d["ms_per_step"], c["reads_per_step"], c.get("resident_child_batches"), c.get("hash_windows"), c.get("level2_slabs")))
    print("  stages", {k: round(v, 2) for k, v in d["stages_ms"].items()})
    rl = d["roofline"]
    print("  dominant %s frac %.3f; stage fracs %s; pass frac %.3f" % (
        rl["kernel"], rl["frac"], {k: round(v["frac"], 3) for k, v in rl["stages"].items() if v["frac"]}, rl["pass"]["frac"]))
    pb = d["parent_build"]
    print("  parent: %.1f ms per %d-read batch %s, %.1f Gk-mers/s; all batches %.2f s (+ arena %.2f s); first batches %s" % (
        pb["insert_ms_per_batch"], pb["reads_per_batch"], {k: round(v, 1) for k, v in pb["insert_stages_ms"].items()},
        pb["insert_gkmers_s"] or 0, pb["seconds_all_batches_incl_read_generation"], pb.get("arena_reserve_seconds", 0),
        pb["device_ms_of_each_batch"][:3]))
    e = d.get("end_to_end")
    if e:
        print("  end to end: child %.2f Gk-mers/s (%.2f s), trio %.2f s = %.1f Gk-mers/s, absent %d, child-only %d" % (
            e["child_gkmers_s"], e["child_seconds"], e["trio_seconds"], e["trio_gkmers_s"], e["absent_occurrences"], e["child_only_kmers"]))
    cb = d.get("cpu_baseline")
    if cb:
        print("  cpu %.3f Gk-mers/s on %d cores; sample matches: %s via %s" % (cb["value"], cb["cores"], cb["gpu_matches_oracle_on_sample"], cb.get("gpu_path_of_the_sample")))
    for k, v in d.get("other_workloads", {}).items():
        print("  other %s: %.2f Gk-mers/s, %.2f ms/step, %s, dominant %s %.3f" % (
            k, v["value"], v["ms_per_step"], {a: round(b, 2) for a, b in v["stages_ms"].items()}, v["roofline"]["kernel"], v["roofline"]["frac"]))
