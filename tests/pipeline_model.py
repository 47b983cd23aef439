"""Executable specification (numpy) of the device construction pipeline's HOST LOGIC.

Every step below is one kernel (or one device sort) of suffixarray_amd/csrc/sa_build.hip; the
model replaces each kernel by the equivalent numpy expression and the device radix sort by a
stable argsort, so the round structure (alphabet compaction, packed initial keys, chunk rounds,
ISA-based doubling rounds, head flags, active-set compaction, truncated mode) can be checked
against the oracle on CPU.  Test infrastructure only.
"""
import numpy as np


def _bits_for(count):
    """smallest b with 2**b >= count (count distinct values 0..count-1)"""
    b = 0
    while (1 << b) < count:
        b += 1
    return b


def _pack_chars(codes_padded, starts, nchars, b, top_bit):
    """key bits [top_bit - b*nchars, top_bit) <- codes at starts+0..nchars-1 (uint64)."""
    key = np.zeros(starts.size, dtype=np.uint64)
    for j in range(nchars):
        c = codes_padded[starts + j].astype(np.uint64)
        key |= c << np.uint64(top_bit - b * (j + 1))
    return key


def build_sa_model(text, max_suffix_length=0, chunk_rounds_before_doubling=2, k0_override=None, stats=None):
    t = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    n = t.size
    if n == 0:
        return np.zeros(0, np.uint32)
    L = max_suffix_length if max_suffix_length else 0  # 0 = full suffix array
    # k_byte_hist + host: alphabet compaction. code 0 = past the end of the text.
    freq = np.bincount(t, minlength=256)
    present = np.nonzero(freq)[0]
    sigma = present.size
    code = np.zeros(256, dtype=np.uint8 if sigma < 255 else np.uint16)
    code = code.astype(np.uint16)
    code[present] = np.arange(1, sigma + 1)
    b = _bits_for(sigma + 1)
    kmax = 64 // b
    k0 = kmax if k0_override is None else min(kmax, k0_override)
    if L:
        k0 = min(k0, L)
    codes = np.concatenate([code[t], np.zeros(80, np.uint16)])
    # k_keygen
    idx = np.arange(n, dtype=np.int64)
    key = _pack_chars(codes, idx, k0, b, 64)
    # device sort #0
    order = np.argsort(key, kind="stable")
    key = key[order]
    sa = idx[order].astype(np.uint32)
    head = np.ones(n, dtype=bool)
    head[1:] = key[1:] != key[:-1]
    h = k0
    isa = None
    rounds = []

    def active_of(headflags):
        nxt = np.ones(headflags.size, dtype=bool)
        nxt[:-1] = headflags[1:]
        return ~(headflags & nxt)

    act = active_of(head)
    apos = np.nonzero(act)[0]
    ahead = head[apos]
    chunk_done = 0
    m_prev = 0
    while apos.size and (L == 0 or h < L):
        aidx = sa[apos].astype(np.int64)
        gid = (np.cumsum(ahead) - 1).astype(np.uint64)
        ngroups = int(gid[-1]) + 1
        gb = _bits_for(ngroups)
        # doubling (inverse suffix array: one scatter over the whole text) only once the active set has stopped
        # halving from round to round (sa_build.hpp, adaptive_doubling)
        shrinking = m_prev != 0 and 2 * apos.size <= m_prev
        use_chunk = bool(L) or chunk_done < chunk_rounds_before_doubling or (isa is None and shrinking)
        if use_chunk:
            kc = (64 - gb) // b
            if L:
                kc = min(kc, L - h)
            if kc == 0:  # cannot happen for n < 2**32 and b <= 9, guarded anyway
                use_chunk = False
        if use_chunk:
            starts = np.minimum(aidx + h, n + 8)
            key = (gid << np.uint64(64 - gb)) if gb else np.zeros(apos.size, np.uint64)
            key = key | _pack_chars(codes, starts, kc, b, 64 - gb)
            h_next = h + kc
            chunk_done += 1
            kind = "chunk"
        else:
            if isa is None:
                # k_isa_build: max-scan of head positions, scattered through SA
                gstart = np.maximum.accumulate(np.where(head, np.arange(n), 0))
                isa = np.empty(n, dtype=np.int64)
                isa[sa] = gstart
            rb = _bits_for(n + 1)
            assert gb + rb <= 64
            p = aidx + h
            key2 = np.where(p < n, isa[np.minimum(p, n - 1)] + 1, 0).astype(np.uint64)
            key = (gid << np.uint64(rb)) | key2
            h_next = 2 * h
            kind = "double"
        order = np.argsort(key, kind="stable")
        key = key[order]
        sidx = aidx[order]
        # k_writeback
        sa[apos] = sidx.astype(np.uint32)
        nh = np.ones(apos.size, dtype=bool)
        nh[1:] = key[1:] != key[:-1]
        head[apos] = nh
        if isa is not None:
            gs = np.maximum.accumulate(np.where(nh, apos, 0))
            isa[sidx] = gs
        rounds.append((kind, h, h_next, int(apos.size), ngroups))
        m_prev = int(apos.size)
        h = h_next
        keep = active_of(nh)
        apos = apos[keep]
        ahead = nh[keep]
    if stats is not None:
        stats.update(dict(sigma=sigma, b=b, k0=k0, rounds=rounds))
    return sa


# ---- the three-pass plan of the narrow sort (csrc/radix_split.hpp): host logic + what its two passes must leave ----

def split_levels(keys, lo_bits, hb):
    """split_hist_kernel + split_levels_kernel: keys = top digit (8 bits) above lo_bits narrow bits; the largest group per
    level k = 0..hb when the records of a bucket are grouped by the top k of the narrow key's first hb bits."""
    fine = (keys >> np.uint64(lo_bits - hb)).astype(np.int64)   # (bucket << hb) | fine bin
    hist = np.bincount(fine, minlength=256 << hb)
    return [int(hist.reshape(-1, 1 << (hb - k)).sum(axis=1).max()) for k in range(hb + 1)]


def choose_split_level(levels, lo_bits, cap=8192, cap_big=16384):
    """radix_sort_narrow's rule: the SMALLEST level whose groups all fit the local pass, its large form when none fits the
    small one (needs 12 key bits below the sub-bucket's), None: the plan is declined (LSD passes).  -> (rb, big)"""
    hb = len(levels) - 1
    for k in range(1, hb + 1):
        if levels[k] <= cap:
            return k, False
    for k in range(1, hb + 1):
        if levels[k] <= cap_big and lo_bits - k >= 12:
            return k, True
    return None, False


def three_pass_model(keys, lo_bits, dbits, cap=8192, cap_big=16384, seed=0):
    """What the plan must leave for (key, suffix) records with suffix = position: the order of a stable sort, the query
    directory and the staged tied slots -- derived the way the kernels derive them: the split pass groups the records of a
    bucket by rb key bits in ANY order (here: shuffled), the local pass orders a sub-bucket by (key, suffix), reads the
    sub-bucket's directory slice off its bin starts and finds the tied slots from neighbouring keys INSIDE the sub-bucket.
    keys: uint64, top digit in bits [lo_bits, lo_bits + 8).  -> None (declined) or dict."""
    n = keys.size
    hb = min(10, lo_bits - 11)
    if hb < 1:
        return None
    levels = split_levels(keys, lo_bits, hb)
    rb, big = choose_split_level(levels, lo_bits, cap, cap_big)
    if rb is None:
        return None
    bb = 12 if (big or lo_bits - rb >= 12) else 11
    g2 = dbits - 8 - rb
    if g2 < 0 or g2 > bb or dbits - 8 > lo_bits:
        return None   # (the fused flags work is declined; not modelled)
    rng = np.random.default_rng(seed)
    sub = (keys >> np.uint64(lo_bits - rb)).astype(np.int64)          # sub-bucket = (bucket << rb) | group
    nsub = 256 << rb
    counts = np.bincount(sub, minlength=nsub)
    starts = np.concatenate([[0], np.cumsum(counts)])
    # split pass: any order inside a sub-bucket
    shuffled = rng.permutation(n)
    order = shuffled[np.argsort(sub[shuffled], kind="stable")]
    out_keys = np.empty(n, np.uint64)
    out_sa = np.empty(n, np.int64)
    directory = np.empty((1 << dbits) + 1, np.int64)
    directory[1 << dbits] = n
    staged = []
    for s_i in range(nsub):
        lo, hi = starts[s_i], starts[s_i + 1]
        nd = 1 << g2
        if lo == hi:
            directory[s_i << g2:(s_i + 1) << g2] = lo
            continue
        idx = order[lo:hi]
        k = keys[idx]
        local = np.lexsort((idx, k))                      # the local pass: (key, suffix)
        idx, k = idx[local], k[local]
        out_keys[lo:hi], out_sa[lo:hi] = k, idx
        # bins: the bb key bits below the sub-bucket's; the directory slice is the bin-start table, subsampled
        rest = lo_bits - rb
        bins = ((k >> np.uint64(rest - bb)) & np.uint64((1 << bb) - 1)).astype(np.int64)
        bin_start = np.concatenate([[0], np.cumsum(np.bincount(bins, minlength=1 << bb))])
        directory[s_i << g2:(s_i + 1) << g2] = lo + bin_start[(np.arange(nd) << (bb - g2))]
        # tied slots: slot p stages itself when it starts a group and its successor always (pairs inside the sub-bucket only)
        same_next = np.flatnonzero(k[1:] == k[:-1])
        for p in same_next:
            if p == 0 or k[p - 1] != k[p]:
                staged.append((lo + p, True))
            staged.append((lo + p + 1, False))
    return {"rb": rb, "big": big, "levels": levels, "keys": out_keys, "sa": out_sa, "dir": directory, "staged": sorted(staged)}
