"""Independent pure-Python restatement of the spec for SMALL cases.

Written on strings (slice, reverse-complement by translation table, int(..., 4)) so that it
shares no code shape with either the C oracle (rolling u128) or the HIP kernels (bit-field
extraction).  Test infrastructure only.
"""
M64 = (1 << 64) - 1
_COMP = str.maketrans("ACGT", "TGCA")
_DIGIT = str.maketrans("ACGT", "0123")


def fmix64(x):
    x &= M64
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & M64
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & M64
    x ^= x >> 33
    return x


def kmer_int(s):
    return int(s.translate(_DIGIT), 4)


def canonical_int(s, canonical=True):
    f = kmer_int(s)
    if not canonical:
        return f
    r = kmer_int(s.translate(_COMP)[::-1])
    return min(f, r)


def read_kmers(read, k, canonical=True):
    """-> list of (valid, value) per window"""
    read = read.upper()
    out = []
    for i in range(len(read) - k + 1):
        w = read[i:i + k]
        if all(c in "ACGT" for c in w):
            out.append((True, canonical_int(w, canonical)))
        else:
            out.append((False, 0))
    return out


def hash_kmer(v, k, seed):
    hi, lo = v >> 64, v & M64
    t = seed & M64
    if k > 32:
        t ^= fmix64((hi + 0x9E3779B97F4A7C15) & M64)
    return fmix64(lo ^ t)


def bloom_positions(h, log2_bits, n_hashes):
    lb = log2_bits - 9
    block = h >> (64 - lb) if lb > 0 else 0
    a = h & 511
    d = ((h >> 9) & 511) | 1
    return block, [(a + j * d) & 511 for j in range(n_hashes)]


class Bloom:
    def __init__(self, log2_bits, n_hashes, seed, k):
        self.log2_bits, self.n_hashes, self.seed, self.k = log2_bits, n_hashes, seed, k
        self.bits = set()          # absolute bit indices: block * 512 + bit

    def _pos(self, v):
        b, bits = bloom_positions(hash_kmer(v, self.k, self.seed), self.log2_bits, self.n_hashes)
        return [b * 512 + t for t in bits]

    def insert(self, v):
        self.bits.update(self._pos(v))

    def contains(self, v):
        return all(p in self.bits for p in self._pos(v))

    def words(self):
        """dict word_index -> u64 value (little-endian u64 words, bit t of a block in word t>>6)"""
        w = {}
        for p in self.bits:
            w[p >> 6] = w.get(p >> 6, 0) | (1 << (p & 63))
        return w


def child_only(parent_reads, child_reads, k, log2_bits, n_hashes, seed, canonical=True, min_count=1):
    bl = Bloom(log2_bits, n_hashes, seed, k)
    for r in parent_reads:
        for ok, v in read_kmers(r, k, canonical):
            if ok:
                bl.insert(v)
    counts = {}
    for r in child_reads:
        for ok, v in read_kmers(r, k, canonical):
            if ok and not bl.contains(v):
                counts[v] = counts.get(v, 0) + 1
    return sorted((v, c) for v, c in counts.items() if c >= min_count), bl


def exact_child_only(parent_reads, child_reads, k, canonical=True, min_count=1):
    """DK_SET_EXACT semantics: plain set difference with counts.
    -> (sorted [(kmer, count)], number of distinct parent k-mers)"""
    parents = set()
    for r in parent_reads:
        for ok, v in read_kmers(r, k, canonical):
            if ok:
                parents.add(v)
    counts = {}
    for r in child_reads:
        for ok, v in read_kmers(r, k, canonical):
            if ok and v not in parents:
                counts[v] = counts.get(v, 0) + 1
    return sorted((v, c) for v, c in counts.items() if c >= min_count), len(parents)


def pack_reads(reads):
    """-> (bases words, mask words, n_bases) in the dk_reads format"""
    codes, flags = [], []
    for r in reads:
        for ch in r.upper():
            if ch in "ACGT":
                codes.append("ACGT".index(ch)); flags.append(0)
            else:
                codes.append(0); flags.append(1)
        codes.append(0); flags.append(1)
    n = len(codes)
    bases = [0] * ((n + 31) // 32)
    mask = [0] * ((n + 63) // 64)
    for i, (c, f) in enumerate(zip(codes, flags)):
        if f:
            mask[i // 64] |= 1 << (63 - i % 64)
        else:
            bases[i // 32] |= c << (62 - 2 * (i % 32))
    return bases, mask, n
