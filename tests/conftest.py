import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """PyTorch's wheel carries its own HIP runtime; in a process shared with the engine it has to come
    up first, or torch.cuda.is_available() stays false for the tests that hand torch tensors to the
    engine.  No-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield


def random_reads(rng, n_reads, min_len, max_len, n_rate=0.0, lower_rate=0.0):
    """ragged random reads over ACGT with optional N / lowercase"""
    reads = []
    for _ in range(n_reads):
        ln = int(rng.integers(min_len, max_len + 1))
        a = rng.integers(0, 4, size=ln)
        s = np.array(list("ACGT"))[a]
        if n_rate > 0:
            s[rng.random(ln) < n_rate] = "N"
        if lower_rate > 0:
            m = rng.random(ln) < lower_rate
            s[m] = np.char.lower(s[m])
        reads.append("".join(s))
    return reads


def related_trio(rng, genome_len=4000, n_reads=120, read_len=100, err=0.01, n_rate=0.002, denovo=3):
    """tiny trio with shared sequence so that both present and absent child k-mers occur"""
    g = rng.integers(0, 4, size=genome_len)
    child_g = g.copy()
    pos = rng.choice(genome_len, size=denovo, replace=False)
    child_g[pos] = (child_g[pos] + 1 + rng.integers(0, 3, size=denovo)) % 4

    def sample(genome, n):
        out = []
        for _ in range(n):
            s = int(rng.integers(0, genome_len - read_len + 1))
            r = genome[s:s + read_len].copy()
            e = rng.random(read_len) < err
            r[e] = (r[e] + 1 + rng.integers(0, 3, size=int(e.sum()))) % 4
            if rng.random() < 0.5:
                r = (3 - r)[::-1]
            st = np.array(list("ACGT"))[r]
            st[rng.random(read_len) < n_rate] = "N"
            out.append("".join(st))
        return out

    return sample(g, n_reads) + sample(g, n_reads), sample(child_g, n_reads)


@pytest.fixture
def rng():
    return np.random.default_rng(20260313)
