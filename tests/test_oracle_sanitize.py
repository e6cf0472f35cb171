"""The C oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool).
The oracle is the checker of every parity test, so its own memory safety and freedom from undefined
behaviour (shifts by 64, signed overflow, out-of-bounds on ragged reads) is worth a test."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r"""
import ctypes as C, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from oracle import orc
orc._LIB = None
orc.build = lambda: {lib!r}                      # load the sanitizer build instead of the optimised one
from conftest import random_reads, related_trio
rng = np.random.default_rng(5)
for k in (1, 2, 15, 31, 32, 33, 47, 64):
    parents, child = related_trio(rng, genome_len=1200, n_reads=25, read_len=90)
    parents += random_reads(rng, 20, 0, 70, n_rate=0.05, lower_rate=0.3) + ["", "N", "ACGT" * 40]
    child += random_reads(rng, 20, 0, 70, n_rate=0.05, lower_rate=0.3) + ["", "n" * 70]
    pseq, poff = orc.concat_reads(parents)
    cseq, coff = orc.concat_reads(child)
    f = orc.new_filter(20)
    orc.bloom_insert(f, 20, 4, 99, k, True, pseq, poff)
    a = orc.bloom_probe(f, 20, 4, 99, k, True, cseq, coff, 1)
    b = orc.bloom_probe(f, 20, 4, 99, k, True, cseq, coff, 1, n_threads=3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    orc.exact_child_only(k, True, pseq, poff, cseq, coff, 2)
    orc.count_reads(k, False, cseq, coff)
    bases, mask, n = orc.pack_reads(cseq, coff)
cfg = orc.synth_cfg(genome_len=5000)
seq, off = orc.synth_reads(cfg, 2, 3, 50)
bases, mask, n = orc.pack_reads(seq, off)
s2, o2 = orc.unpack_fixed(bases, mask, 50, 150)
assert np.array_equal(o2, off)
print("sanitized run ok")
"""


@pytest.mark.timeout(600)
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    lib = os.path.join(ROOT, "oracle", "libdk_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdk_oracle_asan.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="3")
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, lib=lib)], env=env, capture_output=True, text=True,
                       timeout=500)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
