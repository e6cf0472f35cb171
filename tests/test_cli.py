"""tools/denovo_kmer_cli.cpp: FASTA / FASTQ / text in, child-only k-mers out (TSV), multi-batch with
device-side merge, filter save / load.  Compiles on CPU; runs against the oracle on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, related_trio
from oracle import orc

LIBDIR = os.path.join(ROOT, "denovo_kmer_amd")
SRC = os.path.join(ROOT, "tools", "denovo_kmer_cli.cpp")


def build(tmp_path):
    exe = str(tmp_path / "denovo_kmer_cli")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", exe, SRC,
                           "-L" + LIBDIR, "-ldenovo_kmer", "-Wl,-rpath," + LIBDIR,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def kmer_str(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def test_cli_compiles_and_prints_usage(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, "--child", "x", "--out", "y"], capture_output=True, text=True)
    assert r.returncode == 2 and "--parent" in r.stderr
    # geometry the accumulator cannot take is a usage error, not an arithmetic accident further down
    ok = ["--parent", "p", "--child", "x", "--out", "y"]
    for bad, word in ((["--filter-log2", "19"], "20..40"), (["--filter-log2", "41"], "20..40"), (["--k", "65"], "1..64"),
                      (["--filter-log2", "22", "--windows", "8"], "at most 4 hash window"), (["--windows", "3"], "power of two")):
        r = subprocess.run([exe] + ok + bad, capture_output=True, text=True)
        assert r.returncode == 2 and word in r.stderr, (bad, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("k,mode", [(31, "bucketed"), (31, "direct"), (45, "auto")])
def test_cli_matches_oracle(tmp_path, rng, k, mode):
    exe = build(tmp_path)
    parents, child = related_trio(rng, genome_len=3000, n_reads=90, read_len=110)
    child = child + child[:30]
    p1, p2 = parents[:90], parents[90:]
    # three input formats
    (tmp_path / "p1.fa").write_text("".join(f">r{i}\n{s[:60]}\n{s[60:]}\n" for i, s in enumerate(p1)))
    (tmp_path / "p2.fq").write_text("".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(p2)))
    (tmp_path / "c.txt").write_text("\n".join(child) + "\n")
    out, flt = tmp_path / "out.tsv", tmp_path / "parents.dkbloom"
    base = [exe, "--k", str(k), "--filter-log2", "22", "--hashes", "4", "--seed", "4242", "--min-count", "2",
            "--batch-reads", "37", "--mode", mode]
    subprocess.run(base + ["--parent", str(tmp_path / "p1.fa"), "--parent", str(tmp_path / "p2.fq"),
                           "--child", str(tmp_path / "c.txt"), "--out", str(out), "--save-filter", str(flt)],
                   check=True, capture_output=True, text=True)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(22)
    orc.bloom_insert(f, 22, 4, 4242, k, True, pseq, poff)
    km, cn, _ = orc.bloom_probe(f, 22, 4, 4242, k, True, cseq, coff, min_count=2)
    expect = ["kmer\tcount"] + [f"{kmer_str(a['hi'], a['lo'], k)}\t{int(c)}" for a, c in zip(km, cn)]
    assert out.read_text().strip().split("\n") == expect and len(expect) > 5
    # the saved filter reproduces the run without the parents -- here with the child streamed in two hash-window passes
    out2 = tmp_path / "out2.tsv"
    subprocess.run(base + ["--load-filter", str(flt), "--child", str(tmp_path / "c.txt"), "--out", str(out2), "--windows", "2",
                           "--accum-capacity", "100000"],
                   check=True, capture_output=True, text=True)
    assert out2.read_text() == out.read_text()
    assert np.array_equal(np.fromfile(flt, dtype=np.uint64, offset=64), f)


@pytest.mark.gpu
@pytest.mark.parametrize("k,mode", [(31, "bucketed"), (45, "direct")])
def test_cli_exact_set_matches_exact_oracle(tmp_path, rng, k, mode):
    exe = build(tmp_path)
    parents, child = related_trio(rng, genome_len=3000, n_reads=90, read_len=110)
    child = child + child[:30]
    (tmp_path / "p.txt").write_text("\n".join(parents) + "\n")
    (tmp_path / "c.txt").write_text("\n".join(child) + "\n")
    out, flt = tmp_path / "out.tsv", tmp_path / "parents.dkexact"
    base = [exe, "--k", str(k), "--filter-log2", "23", "--seed", "4242", "--min-count", "2", "--batch-reads", "41",
            "--mode", mode, "--exact"]
    r = subprocess.run(base + ["--parent", str(tmp_path / "p.txt"), "--child", str(tmp_path / "c.txt"), "--out", str(out),
                               "--save-filter", str(flt)], check=True, capture_output=True, text=True)
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    km, cn, _ = orc.exact_child_only(k, True, pseq, poff, cseq, coff, min_count=2)
    expect = ["kmer\tcount"] + [f"{kmer_str(a['hi'], a['lo'], k)}\t{int(c)}" for a, c in zip(km, cn)]
    assert out.read_text().strip().split("\n") == expect and len(expect) > 5
    pk, _, _ = orc.count_reads(k, True, pseq, poff)
    assert f"exact parent set: {len(pk)} k-mers" in r.stderr
    out2 = tmp_path / "out2.tsv"
    subprocess.run(base + ["--load-filter", str(flt), "--child", str(tmp_path / "c.txt"), "--out", str(out2)],
                   check=True, capture_output=True, text=True)
    assert out2.read_text() == out.read_text()
