"""CPU tests of bench.py's bookkeeping: the algorithmic-byte formulas and the
roofline object the driver reads.  No GPU, no engine: pure arithmetic on recorded statistics."""
import importlib.util
import os

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

# configs[1] as measured (BENCH_r01.json pass_stats)
CHR20 = {"n_reads": 12800000, "n_bases": 1932800000, "n_windows": 1536000000, "n_valid": 1531254708,
         "n_absent": 207567189, "n_distinct": 204545270, "n_emitted": 204545270}
F_CHR20 = (1 << 34) // 8


def gb(x):
    return round(x / 1e9, 2)


def test_chr20_byte_table_matches_baseline_md():
    b = {s: bench.stage_algorithmic_bytes(s, CHR20, F_CHR20, 8, 1) for s in ("scan_part", "repart", "seg_probe", "seg_count", "seg_insert")}
    assert gb(b["scan_part"]) == 12.97            # 0.725 + 12.25 (VERDICT round 1 recomputed the same figure)
    assert gb(b["repart"]) == 24.50
    assert gb(b["seg_probe"]) == 16.06
    assert gb(b["seg_count"]) == 4.12
    per_window = sum(b[s] for s in ("scan_part", "repart", "seg_probe", "seg_count")) / CHR20["n_windows"]
    assert 37.0 < per_window < 38.0               # 57.6 GB per pass; the 45.6 B of BASELINE.md counts distinct at 12 B and v = 1
    assert gb(b["seg_insert"]) == gb(8 * CHR20["n_valid"] + 2 * F_CHR20)


def test_hash_windows_divide_records_filter_and_keep_the_stream():
    st = dict(CHR20)
    one = bench.stage_algorithmic_bytes("scan_part", st, F_CHR20, 8, 1)
    two = bench.stage_algorithmic_bytes("scan_part", st, F_CHR20, 8, 2)
    stream = st["n_bases"] * 3 / 8
    assert abs((one - stream) - 2 * (two - stream)) < 1            # records halve, the stream does not
    p1 = bench.stage_algorithmic_bytes("seg_probe", st, F_CHR20, 8, 1)
    p2 = bench.stage_algorithmic_bytes("seg_probe", st, F_CHR20, 8, 2)
    assert abs(p2 - (8 * st["n_valid"] / 2 + F_CHR20 / 2 + 8 * st["n_absent"])) < 1      # n_absent is already the window's
    assert p2 < p1
    assert bench.stage_algorithmic_bytes("repart", st, F_CHR20, 16, 1) == 2 * bench.stage_algorithmic_bytes("repart", st, F_CHR20, 8, 1)
    assert bench.stage_algorithmic_bytes("no_such_stage", st, F_CHR20, 8, 1) is None


def test_roofline_object_names_the_longest_kernel_and_lists_every_stage():
    ms = {"scan_part": 4.9, "repart": 4.8, "seg_probe": 3.3, "seg_count": 1.5, "ovf_probe": 0.01}
    by = {s: bench.stage_algorithmic_bytes(s, CHR20, F_CHR20, 8, 1) for s in ms}
    rl = bench.roofline_of(ms, by, {"scan_part": 14.4e9}, "profiles/traffic.json (static)")
    assert rl["kernel"] == "scan_part" and rl["bound"] == "hbm" and rl["peak"] == 8000.0 and rl["unit"] == "GB/s"
    assert abs(rl["achieved"] - by["scan_part"] / 4.9e-3 / 1e9) < 1e-6 and abs(rl["frac"] - rl["achieved"] / 8000.0) < 1e-12
    assert rl["traffic"] == 14.4e9 and "static" in rl["traffic_source"]
    assert set(rl["stages"]) == set(ms) and rl["stages"]["ovf_probe"]["frac"] is None      # no formula: listed, not rated
    timed = [s for s in ms if s != "ovf_probe"]
    assert abs(rl["pass"]["ms"] - sum(ms[s] for s in timed)) < 1e-9
    assert abs(rl["pass"]["algorithmic_bytes"] - sum(by[s] for s in timed)) < 1
    assert 0.3 < rl["pass"]["frac"] < 0.7


def test_committed_traffic_is_per_workload_and_only_for_the_same_geometry():
    t, src = bench.committed_traffic("chr20", 12_800_000, 34, 1)
    assert t and src and "static" in src and t["scan_part"] > 1e10
    assert bench.committed_traffic("chr20", 12_800_000, 35, 1) == (None, None)
    assert bench.committed_traffic("chr20", 12_800_000, 34, 2) == (None, None)
    t, _ = bench.committed_traffic("wgs", 1_200_000_000, 39, 1)
    assert t and t["seg_probe"] > 4e10
