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
