"""The C++ host mirror (include/denovo_kmer.hpp): compiles and links against the C ABI on CPU, and on
the GPU gives the oracle's result through a compiled-language driver (tests/cpp/trio_cli.cpp)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, related_trio
from oracle import orc

LIBDIR = os.path.join(ROOT, "denovo_kmer_amd")
SRC = os.path.join(ROOT, "tests", "cpp", "trio_cli.cpp")


def build_cli(tmp_path):
    exe = str(tmp_path / "trio_cli")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", exe, SRC,
                           "-L" + LIBDIR, "-ldenovo_kmer", "-Wl,-rpath," + LIBDIR,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_cpp_host_compiles_and_links(tmp_path):
    exe = build_cli(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_cpp_host_reports_no_device(tmp_path):
    exe = build_cli(tmp_path)
    (tmp_path / "p.txt").write_text("ACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([exe, "21", "20", "4", "1", "0", str(tmp_path / "p.txt"), str(tmp_path / "p.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("k,mode", [(31, 1), (31, 2), (45, 1), (45, 2)])
def test_cpp_host_matches_oracle(tmp_path, rng, k, mode):
    exe = build_cli(tmp_path)
    parents, child = related_trio(rng, genome_len=2500, n_reads=70, read_len=120)
    (tmp_path / "p.txt").write_text("\n".join(parents) + "\n")
    (tmp_path / "c.txt").write_text("\n".join(child) + "\n")
    r = subprocess.run([exe, str(k), "22", "4", "4242", str(mode), str(tmp_path / "p.txt"), str(tmp_path / "c.txt")],
                       capture_output=True, text=True, check=True)
    rows = [ln.split() for ln in r.stdout.strip().split("\n")]
    stats = [int(x) for x in rows[-1][1:]]
    got = sorted((int(a), int(b), int(c)) for a, b, c in rows[:-1])
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(22)
    orc.bloom_insert(f, 22, 4, 4242, k, True, pseq, poff)
    km, cn, st = orc.bloom_probe(f, 22, 4, 4242, k, True, cseq, coff)
    assert got == [(int(a["hi"]), int(a["lo"]), int(c)) for a, c in zip(km, cn)]
    assert stats == [st["n_windows"], st["n_valid"], st["n_absent"], st["n_distinct"]]
