import numpy as np
rng = np.random.default_rng(1)
def fmix32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x85ebca6b) & 0xFFFFFFFF; x ^= x >> 13; x = (x * 0xc2b2ae35) & 0xFFFFFFFF; x ^= x >> 16
    return x
def stats(k, m, L=150, n_reads=4000, chunk=None):
    w = k - m + 1
    tot_k = 0; tot_rec = 0; maxlen = 0
    pos_global = 0
    for r in range(n_reads):
        s = rng.integers(0, 4, L)
        # m-mers
        nm = L - m + 1
        f = np.zeros(nm, dtype=np.uint64); rc = np.zeros(nm, dtype=np.uint64)
        for i in range(m):
            f = (f << np.uint64(2)) | s[i:i+nm].astype(np.uint64)
            rc = rc | ((3 - s[i:i+nm]).astype(np.uint64) << np.uint64(2*i))
        can = np.minimum(f, rc)
        h = fmix32(can)
        nk = L - k + 1
        # minimizer position per k-mer (leftmost min)
        win = np.lib.stride_tricks.sliding_window_view(h, w)
        mpos = np.arange(nk) + np.argmin(win, axis=1)
        cut = np.ones(nk, dtype=bool)
        cut[1:] = mpos[1:] != mpos[:-1]
        if chunk:
            gp = pos_global + np.arange(nk)
            cut |= (gp % chunk) == 0
        starts = np.flatnonzero(cut)
        lens = np.diff(np.append(starts, nk))
        tot_k += nk; tot_rec += len(starts); maxlen = max(maxlen, lens.max())
        pos_global += L + 1
    return tot_k / tot_rec, maxlen
for k, m in ((31, 15), (31, 13), (31, 11), (21, 11), (27, 11)):
    for chunk in (None, 32, 64):
        a, mx = stats(k, m, chunk=chunk)
        print(f"k={k} m={m} w={k-m+1} chunk={chunk}: {a:.2f} k-mers per record, max {mx}")
