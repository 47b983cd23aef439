"""GPU: BASELINE.json's full sizes.  Config 2 (N = 1e8, 32-bit SA) is compared bit for bit with a
CPU run (reference libsais when oracle/_ref travelled, else the oracle port); config 3 (N = 1e9,
libsais64 layout + 1M batched 16-byte queries) is checked through size-independent properties:
the on-device sufcheck (SA is unique, so verified == bit-exact), widening consistency, and query
ranges against the oracle restatement on the downloaded SA."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cpu_sa(text, oracle):
    import os
    from oracle.oracle import Ref
    if Ref.available():
        from oracle.oracle import usable_threads
        threads = usable_threads()   # the GPU box's CPU share
        return Ref().libsais(text, threads=threads).astype(np.uint32), f"reference libsais_omp({threads})"
    return oracle.sais(text).astype(np.uint32), "oracle port"


def test_config2_n1e8_bit_exact(gpu, oracle):
    from suffixarray_amd import synth
    t = synth.d1_uniform27(100_000_000)
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        assert idx.verify() == 0
        sa = idx.sa_u32()
        st = idx.build_stats()
    exp, how = _cpu_sa(t, oracle)
    assert np.array_equal(sa, exp), (how, st)


def test_dropin_libsais_calls_n1e8_memcmp_equal(gpu, oracle):
    """The libsais-call-compatible entry points (host pointers in, host suffix array out; libsais.h:84, libsais64.h:61)
    at config 2's size: int64 and int32 results equal the CPU run's byte for byte; freq = the byte histogram
    (libsais.c:1363-1371); the second call reuses the process-level workspace; the breakdown adds up."""
    from suffixarray_amd import synth
    t = synth.d1_uniform27(100_000_000)
    exp, how = _cpu_sa(t, oracle)
    gpu.release_workspace()
    sa64, f64 = gpu.libsais64(t, want_freq=True)
    cold = gpu.last_call_breakdown()
    assert sa64.dtype == np.int64 and np.array_equal(sa64, exp.astype(np.int64)), how
    assert np.array_equal(f64, np.bincount(t, minlength=256))
    sa32, f32 = gpu.libsais(t, want_freq=True)
    warm = gpu.last_call_breakdown()
    assert sa32.dtype == np.int32 and np.array_equal(sa32, exp.view(np.int32)), how
    assert np.array_equal(f32, np.bincount(t, minlength=256))
    assert cold["workspace_reused"] == 0 and warm["workspace_reused"] == 1 and warm["n"] == t.size
    for b in (cold, warm):
        parts = b["workspace_ms"] + b["upload_ms"] + b["build_ms"] + b["download_ms"]
        assert parts <= b["total_ms"] * 1.02 + 1.0 and b["build_device_ms"] <= b["build_ms"] + 0.5, b
    # a smaller text on the same workspace, a truncated build through the engine-compatible call, and an empty text
    small = synth.d2_words(3_000_000)
    assert np.array_equal(gpu.libsais(small), oracle.sais(small).astype(np.int32))
    assert np.array_equal(gpu.construct_truncated_suffix_array(small, 16), oracle.truncated_sa(small, 16))
    assert gpu.libsais64(np.zeros(0, np.uint8)).size == 0
    gpu.release_workspace()
    print("drop-in at n = 1e8: libsais64 cold %.0f ms (workspace %.0f), libsais warm %.0f ms = upload %.0f + build %.0f (device %.1f) + download %.0f" % (
        cold["total_ms"], cold["workspace_ms"], warm["total_ms"], warm["upload_ms"], warm["build_ms"], warm["build_device_ms"], warm["download_ms"]))


def test_words_n1e8_verified_and_bit_exact(gpu, oracle):
    from suffixarray_amd import synth
    t = synth.d2_words(100_000_000)
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        assert idx.verify() == 0, idx.build_stats()
        sa = idx.sa_u32()
    exp, how = _cpu_sa(t, oracle)
    assert np.array_equal(sa, exp), how


def test_verify_detects_corruption(gpu, oracle):
    from suffixarray_amd import synth
    t = synth.d1_uniform27(1_000_000)
    sa = oracle.sais(t).astype(np.uint32)
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.load(t, sa, 0)
        assert idx.verify() == 0
        bad = sa.copy()
        bad[[1000, 1001]] = bad[[1001, 1000]]
        idx.load(t, bad, 0)
        assert idx.verify() > 0
        dup = sa.copy()
        dup[5] = dup[6]
        idx.load(t, dup, 0)
        assert idx.verify() > 0
        tsa = oracle.truncated_sa(t, 6)
        idx.load(t, tsa, 6)
        assert idx.verify() == 0
        idx.load(t, tsa, 0)          # a truncated order is not the full suffix array
        assert idx.verify() > 0


def test_config3_n1e9_properties(gpu, oracle):
    from suffixarray_amd import synth
    n, q, m = 1_000_000_000, 1_000_000, 16
    t = synth.d1_uniform27(n)
    buf, off = synth.query_batch(t, q, m)
    with gpu.DeviceIndex(n, 0) as idx:
        idx.build(t)
        st = idx.build_stats()
        assert idx.verify() == 0, st                       # suffix array verified on the device
        # the headline's plan (round 4): three passes over the records, sub-buckets of at most 8192 records at level 10
        assert st["split_plan"] == 10 and 0 < st["split_max"] <= 8192 and st["radix_passes"] == 3, st
        got = idx.query_batch((buf, off))
        sa = idx.sa_u32()
        # libsais64 layout: the widened copy equals the 32-bit array (checked in slabs)
        sa64 = idx.sa_i64()
    for lo in range(0, n, 1 << 27):
        hi = min(n, lo + (1 << 27))
        assert np.array_equal(sa64[lo:hi], sa[lo:hi].astype(np.int64))
    del sa64
    # ranges: every one of the 1M against the oracle restatement of get_substring_positions
    exp = oracle.query_batch(t, sa, 0xFFFFFFFF, (buf, off))
    assert np.array_equal(got, exp)
    # and directly against the text for a sample: hits carry the pattern, the neighbours do not
    pats = buf.reshape(q, m)
    rng = np.random.default_rng(3)
    for i in rng.integers(0, q, 2000):
        f, s = int(got["first"][i]), int(got["second"][i])
        p = pats[i].tobytes()
        if f == 0xFFFFFFFF:
            assert bytes(t[sa[n - 1]:sa[n - 1] + m]) < p
            continue
        for slot in range(f, min(s, f + 20) + 1):
            assert bytes(t[sa[slot]:sa[slot] + m]) == p
        if f > 0:
            assert bytes(t[sa[f - 1]:sa[f - 1] + m]) < p
        nxt = s + 1 if s >= f else f
        if nxt < n:
            assert bytes(t[sa[nxt]:sa[nxt] + m]) > p


def test_d1_n1p3e9_large_local_form_verified(gpu):
    """Beyond the headline size the sub-buckets of the three-pass plan outgrow 8192 records at the finest level (D1 at n = 1.3e9:
    10 294): the plan then takes the large form of its local pass (sub-buckets of up to 16 384 records, one workgroup of 1024
    threads per CU) instead of falling back to five passes.  Size-independent properties: the plan taken, the suffix
    array verified on the device (unique => bit-exact), a 64-bit build's int64 copy equal to the u32 array, hits of a query batch
    carry their pattern."""
    import torch
    from suffixarray_amd import synth
    n, q, m = 1_300_000_000, 100_000, 16
    t = synth.d1_uniform27(n)
    buf, off = synth.query_batch(t, q, m)
    with gpu.DeviceIndex(n, 0) as idx:
        idx.build(t)
        out = torch.empty(n, dtype=torch.int64, device="cuda:0")
        idx.build_device64(idx.text_dev, n, out.data_ptr(), 0)
        idx.sync()
        st = idx.build_stats()
        assert st["split_plan"] == 10 and 8192 < st["split_max"] <= 16384 and st["radix_passes"] == 3 and st["widen_fused"] == 1, st
        assert idx.verify() == 0, st
        got = idx.query_batch((buf, off))
        sa = idx.sa_u32()
    for lo in range(0, n, 1 << 27):
        hi = min(n, lo + (1 << 27))
        assert np.array_equal(out[lo:hi].cpu().numpy(), sa[lo:hi].astype(np.int64))
    del out
    pats = buf.reshape(q, m)
    rng = np.random.default_rng(4)
    hits = 0
    for i in rng.integers(0, q, 2000):
        f, s2 = int(got["first"][i]), int(got["second"][i])
        p = pats[i].tobytes()
        if f == 0xFFFFFFFF or s2 < f:
            continue
        hits += 1
        assert bytes(t[sa[f]:sa[f] + m]) == p and bytes(t[sa[s2]:sa[s2] + m]) == p
        if f > 0:
            assert bytes(t[sa[f - 1]:sa[f - 1] + m]) < p
        if s2 + 1 < n:
            assert bytes(t[sa[s2 + 1]:sa[s2 + 1] + m]) > p
    assert hits > 200


def test_d2_words_n1e9_properties(gpu, oracle):
    """SURVEY 8(d)'s realistic text at the headline size: D2 words, N = 1e9 -- the 10-byte-record sort, the in-LDS group
    finisher and the global rounds on ~5e8 tied suffixes (D1 needs none of them).  Size-independent properties: the suffix array
    verified on the device (unique => bit-exact), the int64 copy of a 64-bit build equal to the u32 array, query ranges against
    the oracle's restatement over the downloaded array, hits and their neighbours against the text."""
    import torch
    from suffixarray_amd import synth
    n, q = 1_000_000_000, 200_000
    t = synth.d2_words_parts(n)
    rng = np.random.default_rng(11)
    pos = rng.integers(0, n - 40, q)
    lens = rng.integers(1, 33, q)
    pats = [bytes(t[p:p + l]) for p, l in zip(pos[: q // 2], lens[: q // 2])]
    pats += [bytes(rng.integers(97, 123, l, dtype=np.uint8)) for l in lens[q // 2:]]          # mostly misses
    pats = [p.replace(b"\n", b" ") for p in pats]
    sa64_t = torch.empty(n, dtype=torch.int64, device="cuda:0")
    with gpu.DeviceIndex(n, 0) as idx:
        idx.build(t)
        idx.build_device64(idx.text_dev, n, sa64_t.data_ptr(), 0)
        st = idx.build_stats()
        assert st["narrow48"] == 1 and st["finisher_resolved"] > n // 10 and st["rounds"] >= 1, st
        assert idx.verify() == 0, st
        got = idx.query_batch(pats)
        sa = idx.sa_u32()
    sa64 = sa64_t.cpu().numpy()
    del sa64_t
    for lo in range(0, n, 1 << 27):
        hi = min(n, lo + (1 << 27))
        assert np.array_equal(sa64[lo:hi], sa[lo:hi].astype(np.int64))
    del sa64
    exp = oracle.query_batch(t, sa, 0xFFFFFFFF, pats)
    assert np.array_equal(got, exp)
    hits = ((exp["second"].astype(np.int64) - exp["first"].astype(np.int64) + 1) & 0xFFFFFFFF) > 0
    assert 0.3 < hits.mean() < 0.9
    for i in rng.integers(0, q, 1500):
        f, s = int(got["first"][i]), int(got["second"][i])
        p = pats[i]
        m = len(p)
        if f == 0xFFFFFFFF:
            assert bytes(t[sa[n - 1]:sa[n - 1] + m]) < p
            continue
        for slot in range(f, min(s, f + 10) + 1):
            assert bytes(t[sa[slot]:sa[slot] + m]) == p
        if f > 0:
            assert bytes(t[sa[f - 1]:sa[f - 1] + m]) < p
        nxt = s + 1 if s >= f else f
        if nxt < n:
            assert bytes(t[sa[nxt]:sa[nxt] + m]) > p
    print("D2 words n=1e9: build %.1f ms, k0=%d, rounds %d, active_total %d, finisher resolved %d" % (
        st["total_ms"], st["initial_chars"], st["rounds"], st["active_total"], st["finisher_resolved"]))


def test_long_repeats_67m_many_doubling_rounds(gpu, monkeypatch):
    """A 1 MiB random block repeated 64 times: ~24 refinement rounds over 6.7e7 active records each.
    Regression for a lost-LDS-atomic race (bare s_barrier after ds_add on a loop path, see
    common.hpp sync_lds) that corrupted one per-chunk histogram in about one sort out of twenty.
    Then the same text with the periodic-run shortcut (period_finish.hpp): no doubling round at all, the same array."""
    rng = np.random.default_rng(1)
    t = np.tile(rng.integers(97, 123, 1 << 20, dtype=np.uint8), 64)
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "0")
    with gpu.DeviceIndex(t.size, 0) as idx:
        for _ in range(2):
            idx.build(t)
            st = idx.build_stats()
            assert st["doubling_rounds"] >= 20, st
            assert idx.verify() == 0, st
        slow = idx.sa_u32().copy()
        slow_ms = st["total_ms"]
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "1")
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        idx.build_device(idx.text_dev, t.size, 0)
        st = idx.build_stats()
        assert st["doubling_rounds"] == 0 and st["period_resolved"] > t.size // 2, st
        assert idx.verify() == 0, st
        assert np.array_equal(idx.sa_u32(), slow)
        print("1 MiB block x 64 (n = 6.7e7): %.1f ms through the doubling rounds, %.1f ms with the periodic-run shortcut" % (slow_ms, st["total_ms"]))


def test_beyond_int32_max_3e9_verified(gpu, monkeypatch):
    """n = 3e9 > INT32_MAX: past the reach of 32-bit libsais (the reference switches to the true
    64-bit libsais64 there, libsais64.c:6684); the device pipeline keeps unsigned 32-bit suffix
    indices up to n = 2^32 - 2.  Checked by the on-device sufcheck and text spot checks of queries."""
    from suffixarray_amd import synth
    n = 3_000_000_000
    t = synth.d1_uniform27(n)
    # both sort plans beyond 2^31 records: the build's own choice (40-bit keys in narrow records: positions and
    # bucket offsets above INT32_MAX), then the 12-byte-record plan
    buf, off = synth.query_batch(t, 20000, 16)
    with gpu.DeviceIndex(n, 0) as idx:
        idx.build(t)
        st = idx.build_stats()
        assert st["pass_launches"][2] > 0 and st["narrow_k"] == 1 and idx.verify() == 0, st
        got_narrow = idx.query_batch((buf, off))   # u32 narrow key array, 27-bit directory, slots beyond 2^31
    monkeypatch.setenv("SA_HIP_NARROW", "0")
    with gpu.DeviceIndex(n, 0) as idx:
        idx.build(t)
        st = idx.build_stats()
        assert st["pass_launches"][0] > 0 and idx.verify() == 0, st
        got = idx.query_batch((buf, off))
        assert np.array_equal(got, got_narrow)
        pats = buf.reshape(-1, 16)
        for i in np.random.default_rng(0).integers(0, 20000, 200):
            f, s = int(got["first"][i]), int(got["second"][i])
            p = pats[i].tobytes()
            if f == 0xFFFFFFFF:
                continue
            if s >= f:
                for x in idx.sa_range(f, min(s - f + 1, 4)):
                    assert bytes(t[int(x):int(x) + 16]) == p
            else:
                assert bytes(t[int(idx.sa_range(f, 1)[0]):][:16]) > p


def test_config5_full_size_csv_mode(gpu, oracle, tmp_path):
    """BASELINE config 5 at its stated size: 50M-row synthetic company_name CSV (1.5 GB file, 9.16e8 column characters),
    max_suffix_length = 32, build + query_records.  Checked through size-independent properties: the on-device
    sufcheck of the truncated order, 200 sampled names whose ranges equal the oracle's binary search over the
    downloaded SA, whose returned rows all contain the name (at most k, each row once), and -- for 12 of them -- whose
    hit count equals a memmem count over the extracted column."""
    import time
    from suffixarray_amd import SuffixArray
    from csv_ingest import extract_column
    rows = 50_000_000
    path = tmp_path / "companies.csv"
    gpu.synth_csv(str(path), rows, 1)
    t0 = time.time()
    s = SuffixArray(csv_file=str(path), search_column="company_name", max_suffix_length=32)
    t_index = time.time() - t0
    idx = s._index
    st = idx.build_stats()
    assert idx.verify() == 0, st
    col = extract_column(str(path), "company_name")
    text = col.text_array
    assert idx.n == text.size and len(col.text_row_starts) == rows
    rng = np.random.default_rng(5)
    picks = rng.integers(0, rows, 200)
    starts = col.text_row_starts
    names = [bytes(text[int(starts[i]):int(starts[i + 1]) - 1 if i + 1 < rows else text.size - 1]) for i in picks]
    got = s.query_ranges([n.decode().upper() for n in names])   # case-insensitive like the reference (pyx:228)
    sa = idx.sa_u32()
    exp = oracle.query_batch(text, sa, 32, names)
    assert np.array_equal(got["first"], exp["first"]) and np.array_equal(got["second"], exp["second"])
    del sa
    hits = (got["second"].astype(np.int64) - got["first"].astype(np.int64) + 1)
    assert (hits >= 1).all()
    blob = text.tobytes()
    for j in range(12):
        assert blob.count(names[j][:32]) >= 1
        # occurrences may overlap only for periodic names; count non-overlapping <= hits, and equal for the usual name
        c = blob.count(names[j][:32])
        assert c <= hits[j] and (c == hits[j] or len(set(names[j])) < 3), (names[j], c, hits[j])
    del blob
    lat = []
    for n in names[:100]:
        t0 = time.perf_counter()
        recs = s.query_records(n.decode(), k=50)
        lat.append(time.perf_counter() - t0)
        assert 1 <= len(recs) <= 50
        assert len({r["id"] for r in recs}) == len(recs)
        for r in recs:
            assert n[:32] in r["company_name"].lower().encode(), (n, r)
    print("config 5: SuffixArray(csv_file) %.2f s, device build %.1f ms, query_records median %.0f us" % (
        t_index, st["total_ms"], float(np.median(lat)) * 1e6))
    s.close()
