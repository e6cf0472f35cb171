"""denovo_kmer_amd -- MI355X-native k-mer extraction / parent-set membership engine.

Python is the host-side plumbing over the C ABI of libdenovo_kmer.so (include/denovo_kmer.h);
all k-mer work runs in hand-written HIP kernels for gfx950.
"""
from ._lib import DkError, LIB_PATH, load  # noqa: F401
from .api import (ChildAccumulator, Engine, KmerCounter, KmerCounts, KmerSet, PinnedPacked, ReadBatch,  # noqa: F401
                  kmer_from_str, kmer_to_str, pack_ascii_host, synth_config)

__all__ = ["ChildAccumulator", "Engine", "KmerCounter", "KmerCounts", "KmerSet", "PinnedPacked", "ReadBatch", "DkError",
           "kmer_from_str", "kmer_to_str", "pack_ascii_host", "synth_config", "load", "LIB_PATH"]
