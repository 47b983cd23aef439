"""GPU: suffix-array construction through the C ABI, bit-exact against the oracle and the
golden vectors produced by the reference (libsais 2.8.4)."""
import hashlib
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN
from test_oracle import check_truncated_order

pytestmark = pytest.mark.gpu


def test_small_cases_bit_exact(gpu, oracle):
    texts = cases.small_texts()
    nmax = max(t.size for t in texts.values())
    with gpu.DeviceIndex(nmax, 0) as idx:
        for name, t in texts.items():
            idx.build(t)
            got = idx.sa_u32()
            exp = oracle.sais(t).astype(np.uint32)
            assert np.array_equal(got, exp), (name, idx.build_stats())
            f = idx.freq()
            assert np.array_equal(f, np.bincount(t, minlength=256).astype(np.uint64)), name


def test_golden_small(gpu):
    g = np.load(os.path.join(GOLDEN, "golden_small.npz"), allow_pickle=False)
    with gpu.DeviceIndex(1 << 17, 0) as idx:
        for name in g["names"]:
            t, sa = g[f"text__{name}"], g[f"sa__{name}"]
            idx.build(t)
            assert np.array_equal(idx.sa_u32(), sa.astype(np.uint32)), name
            assert np.array_equal(idx.sa_i64(), sa.astype(np.int64)), name


def test_golden_1mb_digest(gpu):
    g = np.load(os.path.join(GOLDEN, "golden_1mb.npz"), allow_pickle=False)
    with gpu.DeviceIndex(1 << 20, 0) as idx:
        for name in g["names"]:
            t = g[f"text__{name}"]
            idx.build(t)
            d = hashlib.sha256(idx.sa_u32().astype("<i4").tobytes()).hexdigest()
            assert d == str(g[f"sa_sha256__{name}"][0]), (name, idx.build_stats())


def test_empty_and_tiny(gpu):
    with gpu.DeviceIndex(16, 0) as idx:
        idx.build(np.zeros(0, np.uint8))
        assert idx.sa_u32().size == 0
        idx.build(np.frombuffer(b"q", np.uint8))
        assert idx.sa_u32().tolist() == [0]
        idx.build(np.frombuffer(b"ba", np.uint8))
        assert idx.sa_u32().tolist() == [1, 0]


def test_libsais_compatible_wrappers(gpu, oracle):
    t = cases.small_texts()["d2_300k"]
    sa, freq = gpu.libsais(t, want_freq=True)
    exp = oracle.sais(t)
    assert sa.dtype == np.int32 and np.array_equal(sa, exp)
    assert np.array_equal(freq, np.bincount(t, minlength=256).astype(np.int32))
    sa64 = gpu.libsais64(t)
    assert sa64.dtype == np.int64 and np.array_equal(sa64, exp.astype(np.int64))
    # libsais argument validation: -1 on bad arguments (libsais.c:6620-6623)
    lib = gpu.lib()
    buf = np.zeros(4, np.int32)
    assert lib.sa_hip_libsais(None, buf.ctypes.data, 4, 0, None) == -1
    assert lib.sa_hip_libsais(t.ctypes.data, None, 4, 0, None) == -1
    assert lib.sa_hip_libsais(t.ctypes.data, buf.ctypes.data, -1, 0, None) == -1
    assert lib.sa_hip_libsais(t.ctypes.data, buf.ctypes.data, 4, -1, None) == -1
    assert lib.sa_hip_libsais_omp(t.ctypes.data, buf.ctypes.data, 4, 0, None, -1) == -1
    assert lib.sa_hip_libsais_omp(t.ctypes.data, buf.ctypes.data, 4, 0, None, 3) == 0
    assert np.array_equal(buf, oracle.sais(t[:4]))


def test_truncated_mode_bit_exact_vs_oracle(gpu, oracle):
    texts = cases.small_texts()
    nmax = max(t.size for t in texts.values())
    with gpu.DeviceIndex(nmax, 0) as idx:
        for name, t in texts.items():
            for L in (1, 5, 32, 64):
                idx.build(t, L)
                got = idx.sa_u32()
                assert np.array_equal(got, oracle.truncated_sa(t, L)), (name, L, idx.build_stats())
    t = texts["d2_300k"]
    sa = gpu.construct_truncated_suffix_array(t, 32)
    check_truncated_order(t, sa, 32)


def test_doubling_path_is_exercised(gpu, oracle, monkeypatch):
    """Long repeats must go through chunk rounds AND doubling rounds (with the periodic-run shortcut switched off: it would
    order these groups in one step, test_period_finisher_matches_doubling_rounds)."""
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "0")
    t = cases.small_texts()["repeat_block"]
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        st = idx.build_stats()
        assert st["doubling_rounds"] >= 1 and st["chunk_rounds"] >= 1, st
        assert np.array_equal(idx.sa_u32(), oracle.sais(t).astype(np.uint32))


def test_medium_d1_and_words(gpu, oracle):
    from suffixarray_amd import synth
    for t in (synth.d1_uniform27(10_000_000), synth.d2_words(10_000_000)):
        with gpu.DeviceIndex(t.size, 0) as idx:
            idx.build(t)
            got = idx.sa_u32()
            assert np.array_equal(got, oracle.sais(t).astype(np.uint32)), idx.build_stats()


def test_rebuild_same_handle_is_idempotent(gpu):
    from suffixarray_amd import synth
    t = synth.d1_uniform27(1_000_000)
    with gpu.DeviceIndex(t.size, 0) as idx:
        a = idx.build(t).sa_u32().copy()
        b = idx.build(t).sa_u32().copy()
        assert np.array_equal(a, b)


def test_narrow_record_sort_matches_plain_sort(gpu, oracle, monkeypatch):
    """Initial keys of <= 40 bits are sorted in 8-byte records (top digit first, then LSD passes inside
    the 256 buckets, radix_narrow.hpp); the result must be the plain 12-byte-record sort's, bit for bit:
    uniform text, a skewed alphabet (one huge bucket, most buckets empty), two symbols, word text; forced
    key lengths exercise last passes of 1..8 bits; truncated mode keeps ties in text order."""
    from suffixarray_amd import synth
    rng = np.random.default_rng(11)
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 6_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    two = rng.choice(np.array([97, 122], dtype=np.uint8), 5_000_000)
    runs = [(synth.d1_uniform27(4_500_001), 0, 0), (synth.d1_uniform27(6_000_000), 7, 0), (synth.d1_uniform27(5_000_000), 0, 32),
            (skew, 13, 0), (skew, 9, 0), (two, 20, 0), (two, 9, 6), (synth.d2_words(8_000_000), 8, 0),
            # 256 symbols: 9-bit codes, the top-digit pass reads u64 keys (radix_onesweep_kernel<512,0,true>), not the text
            (rng.integers(0, 256, 5_000_000).astype(np.uint8), 0, 4)]
    for t, k0, L in runs:
        if k0:
            monkeypatch.setenv("SA_HIP_INITIAL_CHARS", str(k0))
        else:
            monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_NARROW", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st = idx.build_stats()
                assert (st["pass_launches"][2] + st["pass_launches"][3] > 0) == (mode == "1"), st
                assert idx.verify() == 0, st
                got[mode] = idx.sa_u32().copy()
        assert np.array_equal(got["1"], got["0"]), (t.size, k0, L)
    t = runs[0][0]
    monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
    monkeypatch.setenv("SA_HIP_NARROW", "1")
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        assert np.array_equal(idx.sa_u32(), oracle.sais(t).astype(np.uint32))


def test_split_plan_matches_lsd_passes(gpu, oracle, monkeypatch):
    """The three-pass plan of the narrow sort (radix_split.hpp: top digit, split pass by the next rb key bits, every sub-bucket
    ordered completely in LDS) against the five-pass plan (SA_HIP_SPLIT=0: LSD passes inside the buckets): the same suffix
    array bit for bit, verified on the device, the oracle's on two texts.  Three ways: "F" = the plan with the build's first
    flags pass folded into the local pass (directory + staged active records come out of the sub-buckets), "S" = the plan
    with the flags pass on its own, "0" = LSD passes.  The query ranges of a batch (hits, misses, edge patterns) must agree
    in all three -- they go through the bucket directory -- and with the oracle on two texts.  Uniform text at several sizes
    and forced key lengths (levels rb = 2..8 by lowering the bound on a sub-bucket), DNA-like text, truncated builds whose ties
    must stay in text order (k0 = L), the fused int64 copy; skewed text must DECLINE the plan and still come out right."""
    from suffixarray_amd import synth
    import cases
    rng = np.random.default_rng(14)
    dna = rng.choice(np.frombuffer(b"acgt", np.uint8), 5_000_000)
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 6_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    d1 = synth.d1_uniform27(6_000_000)
    runs = [("d1", synth.d1_uniform27(4_500_001), 0, 0, 0, True), ("d1_k7", d1, 7, 0, 0, True), ("d1_L8", synth.d1_uniform27(5_000_000), 0, 8, 0, True),
            ("d1_cap2048", d1, 0, 0, 2048, True), ("d1_cap300", d1, 0, 0, 300, True), ("d1_k6_cap1000", d1, 6, 0, 1000, True),
            ("dna_k13", dna, 13, 0, 0, True), ("d1_L8_cap500", synth.d1_uniform27(5_000_000), 0, 8, 500, True),
            # four characters: 20-bit keys, nearly every slot tied -- the staging rows of the fused flags work overflow (more than 256 tied
            # slots per sub-bucket) and the build repeats that work as a pass of its own; sub-buckets beyond 8192: the large local form
            ("d1_k4_ties", d1, 4, 0, 0, None), ("d1_k5_ties", d1, 5, 0, 0, None),
            ("skew", skew, 13, 0, 0, False), ("words_k8", synth.d2_words(8_000_000), 8, 0, 0, None)]
    levels = set()
    failures = []
    lsd = {}
    for name, t, k0, L, cap, taken in runs:
        if k0:
            monkeypatch.setenv("SA_HIP_INITIAL_CHARS", str(k0))
        else:
            monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
        if cap:
            monkeypatch.setenv("SA_HIP_SPLIT_CAP", str(cap))
        else:
            monkeypatch.delenv("SA_HIP_SPLIT_CAP", raising=False)
        pats = cases.query_patterns(t, 3000, np.random.default_rng(5), maxlen=24)
        got, ranges = {}, {}
        for mode in ("F", "S", "0"):
            monkeypatch.setenv("SA_HIP_SPLIT", "0" if mode == "0" else "1")
            monkeypatch.setenv("SA_HIP_SPLIT_FLAGS", "1" if mode == "F" else "0")
            with gpu.DeviceIndex(t.size, 0) as idx:
                try:
                    idx.build(t, L)
                except Exception as e:
                    failures.append((name, mode, str(e), {k: v for k, v in idx.build_stats().items() if k.startswith("split") or k == "initial_chars"}))
                    got[mode] = None
                    continue
                st = idx.build_stats()
                assert st["narrow_k"] == 1, (name, st)
                if mode == "0":
                    assert st["split_plan"] == 0 and st["split_max"] == 0, (name, st)
                else:
                    assert st["split_max"] > 0 and (st["split_plan"] > 0) == (st["split_max"] <= (cap or 16384)), (name, st)
                    assert taken is None or (st["split_plan"] > 0) == taken, (name, st)
                    levels.add(st["split_plan"])
                assert idx.verify() == 0, (name, mode, st)
                got[mode] = idx.sa_u32().copy()
                ranges[mode] = idx.query_batch(pats).copy()
        if any(got.get(m) is None for m in ("F", "S", "0")):
            continue
        assert np.array_equal(got["F"], got["0"]) and np.array_equal(got["S"], got["0"]), name
        assert np.array_equal(ranges["F"], ranges["0"]) and np.array_equal(ranges["S"], ranges["0"]), name
        lsd[name] = (got["0"], ranges["0"], pats)
        if name in ("d1", "d1_L8_cap500"):
            ref = oracle.sais(t).astype(np.uint32) if L == 0 else oracle.truncated_sa(t, L).astype(np.uint32)
            assert np.array_equal(got["F"], ref), name
            assert np.array_equal(ranges["F"], oracle.query_batch(t, ref, L if L else 0xFFFFFFFF, pats)), name
    assert not failures, failures
    assert len(levels - {0}) >= 3, levels
    # the large form of the local pass (sub-buckets of up to 16384 records, one workgroup of 1024 threads per CU: what a text
    # takes whose sub-buckets outgrow 8192 at the finest level), and the split pass with published counts + look-back
    monkeypatch.delenv("SA_HIP_SPLIT_CAP", raising=False)
    monkeypatch.setenv("SA_HIP_SPLIT", "1")
    monkeypatch.setenv("SA_HIP_SPLIT_FLAGS", "1")
    # ... and the top-digit pass in its stable form (the default with this plan: claims by global atomics), and with LDS-atomic ranks
    for env, val in (("SA_HIP_LOCAL_BIG", "1"), ("SA_HIP_SPLIT_ATOMIC", "0"), ("SA_HIP_TOP_CLAIMS", "0"), ("SA_HIP_TOP_ARANKS", "1")):
        monkeypatch.setenv(env, val)
        for name, t, k0, L, cap, taken in runs[:3]:
            if k0:
                monkeypatch.setenv("SA_HIP_INITIAL_CHARS", str(k0))
            else:
                monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st = idx.build_stats()
                assert st["split_plan"] > 0 and idx.verify() == 0, (env, name, st)
                assert np.array_equal(idx.sa_u32(), lsd[name][0]), (env, name)
                assert np.array_equal(idx.query_batch(lsd[name][2]), lsd[name][1]), (env, name)
        monkeypatch.delenv(env)
    # the int64 copy leaves the local pass with the suffixes
    monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
    import torch
    t = runs[0][1]
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.build(t)
        out = torch.full((t.size,), -7, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        idx.build_device64(idx.text_dev, t.size, out.data_ptr(), 0)
        idx.sync()
        st = idx.build_stats()
        assert st["split_plan"] > 0 and st["widen_fused"] == 1, st
        assert np.array_equal(out.cpu().numpy(), oracle.sais(t).astype(np.int64))


def test_narrow48_record_sort_matches_wide_sort(gpu, oracle, monkeypatch):
    """Initial keys of 41..56 bits are sorted as 10-byte records (radix_narrow48.hpp: top digit from the text, the 48-bit
    remainder as u32 + u16, two passes ranked by the u16 part, three or four by the u32 part, u64 keys rebuilt by the last
    pass).  Same suffix array as the 12-byte-record sort (SA_HIP_NARROW48=0), verified on the device, equal to the oracle's;
    forced key lengths cover every split of the remainder (33..48 bits: last passes of 1..8 bits, three and four upper
    passes); a skewed alphabet (one huge bucket), two symbols, DNA-like text, word and name text with the pilot's key;
    truncated builds keep ties in text order; queries over the rebuilt key array answer like the oracle."""
    from suffixarray_amd import synth
    rng = np.random.default_rng(12)
    words = synth.d2_words(6_000_000)
    d1 = synth.d1_uniform27(4_700_000)
    dna = rng.choice(np.frombuffer(b"acgt", np.uint8), 5_000_000)
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 6_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    two = rng.choice(np.array([97, 122], dtype=np.uint8), 5_000_000)
    sym60 = rng.integers(60, 120, 4_500_000).astype(np.uint8)
    runs = [("words", words, 0, 0), ("words_L20", words, 0, 20), ("words_k9", words, 9, 0), ("words_k10", words, 10, 0),
            ("d1_k9", d1, 9, 0), ("d1_k11", d1, 11, 0), ("dna", dna, 0, 0), ("dna_k14", dna, 14, 0), ("dna_k16", dna, 16, 0),
            ("skew_k17", skew, 17, 0), ("two_k24", two, 24, 0), ("two_k21_L30", two, 21, 30), ("sym60_k8", sym60, 8, 0), ("sym60_k9", sym60, 9, 7)]
    for name, t, k0, L in runs:
        if k0:
            monkeypatch.setenv("SA_HIP_INITIAL_CHARS", str(k0))
        else:
            monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
        got, stats = {}, {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_NARROW48", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st = stats[mode] = idx.build_stats()
                assert bool(st["narrow48"]) == (mode == "1"), (name, st)
                assert idx.verify() == 0, (name, mode, st)
                got[mode] = idx.sa_u32().copy()
                if mode == "1" and name in ("words", "words_L20", "dna", "two_k24"):
                    pats = cases.query_patterns(t, 1500, rng)
                    exp = oracle.query_batch(t, got[mode], L if L else 0xFFFFFFFF, pats)
                    assert np.array_equal(idx.query_batch(pats), exp), name
        assert np.array_equal(got["1"], got["0"]), (name, stats)
        bits = stats["1"]["initial_chars"] * stats["1"]["bits_per_symbol"]
        assert 40 < bits <= 56, (name, stats["1"])
        if name in ("words", "dna", "d1_k11"):
            assert np.array_equal(got["1"], oracle.sais(t).astype(np.uint32)), name
        if L:
            assert np.array_equal(got["1"], oracle.truncated_sa(t, L)), name
    monkeypatch.delenv("SA_HIP_INITIAL_CHARS", raising=False)
    monkeypatch.delenv("SA_HIP_NARROW48", raising=False)


def test_period_finisher_matches_doubling_rounds(gpu, oracle, monkeypatch):
    """Long repeats (period_finish.hpp): tied groups whose members form an arithmetic progression inside one periodic run are
    ordered by ONE comparison instead of ~log2(n) doubling rounds over everything.  Same suffix array as with the shortcut
    switched off (SA_HIP_PERIOD_FINISH=0), verified on the device, equal to the oracle's: all-'a', periods 2 / 3 / 7, a random
    block repeated (two alphabets), a Fibonacci string (several differences), a block repeat with one mutated character
    (chains cut by the end of a run: left to the rounds), two periodic regions with different periods, a word text with a
    60 000-character run, near-random text (nothing to do), and the int64 copy of a fused 64-bit build on top of it."""
    import torch
    from suffixarray_amd import synth
    rng = np.random.default_rng(21)
    blk26 = np.tile(rng.integers(97, 123, 300_000, dtype=np.uint8), 16)
    blk4 = np.tile(rng.integers(97, 101, 70_000, dtype=np.uint8), 40)
    mut = blk26.copy()
    mut[2_345_678] = ord("!")
    two = np.concatenate([synth.periodic(1_500_000, 2), np.frombuffer(b"zq", np.uint8), synth.periodic(1_600_000, 7), synth.all_same(900_000, 99)])
    run = synth.d2_words(5_000_000).copy()
    run[2_000_000:2_060_000] = ord("q")
    runs = [("all_a", synth.all_same(3_000_000), True), ("p2", synth.periodic(2_000_000, 2), True), ("p3", synth.periodic(2_000_001, 3), True),
            ("p7", synth.periodic(1_000_003, 7), True), ("blk26", blk26, True), ("blk4", blk4, True), ("fib", synth.fibonacci(2_178_309), None),
            ("mut", mut, None), ("two", two, True), ("run", run, True), ("small_blk", cases.small_texts()["repeat_block"], True),
            ("all_a_70000", cases.small_texts()["all_a_70000"], True), ("d1", synth.d1_uniform27(4_400_000), False)]
    for name, t, expect in runs:
        got, stats = {}, {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_PERIOD_FINISH", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t)
                stats[mode] = idx.build_stats()
                assert idx.verify() == 0, (name, mode, stats[mode])
                got[mode] = idx.sa_u32().copy()
                if mode == "1" and name in ("blk26", "run", "fib"):
                    out = torch.full((t.size,), -7, dtype=torch.int64, device="cuda:0")
                    torch.cuda.synchronize()
                    idx.build_device64(idx.text_dev, t.size, out.data_ptr(), 0)
                    idx.sync()
                    assert np.array_equal(out.cpu().numpy(), got[mode].astype(np.int64)), name
        assert np.array_equal(got["1"], got["0"]), (name, stats)
        assert (expect is None or (stats["1"]["period_resolved"] > 0) == expect) and stats["0"]["period_resolved"] == 0, (name, stats["1"])
        if t.size <= 5_000_000:
            assert np.array_equal(got["1"], oracle.sais(t).astype(np.uint32)), name
        if name in ("all_a", "blk26", "blk4", "p2"):
            assert stats["1"]["rounds"] + 4 < stats["0"]["rounds"], (name, stats["1"]["rounds"], stats["0"]["rounds"])
    # truncated builds never take the shortcut
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "1")
    with gpu.DeviceIndex(blk4.size, 0) as idx:
        idx.build(blk4, 40)
        assert idx.build_stats()["period_resolved"] == 0 and idx.verify() == 0
        assert np.array_equal(idx.sa_u32(), oracle.truncated_sa(blk4, 40))


def test_rounds_sorted_in_lds_match_global_sort(gpu, oracle, monkeypatch):
    """Refinement rounds are sorted group-wise in LDS (round_sort.hpp: tiles of whole groups, 12-bit local group ids,
    packed and unpacked record form, groups too large for a tile through the global sort as a compact list).  Same
    suffix array as with SA_HIP_LOCAL_ROUNDS=0 (every round through the global 8-pass sort), verified on the device:
    word text (millions of small groups, later rounds with few groups: the unpacked form), names with a truncation depth,
    a block repeated 12 times (doubling rounds, groups of 12), a text with one 60 000-character run (a group far larger
    than a tile next to small ones), a skewed alphabet (dense active set, groups of every size)."""
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "0")   # long repeats through the rounds here; the periodic-run shortcut has its own test
    from suffixarray_amd import synth
    rng = np.random.default_rng(31)
    words = synth.d2_words(6_000_000)
    run = synth.d2_words(3_000_000).copy()
    run[1_000_000:1_060_000] = ord("q")
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 3_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    blk = np.tile(rng.integers(97, 101, 200_000, dtype=np.uint8), 12)
    runs = [("words", words, 0), ("words_L20", words, 20), ("run", run, 0), ("skew", skew, 0), ("blocks", blk, 0),
            ("d2_300k", cases.small_texts()["d2_300k"], 0), ("repeat_block", cases.small_texts()["repeat_block"], 0)]
    monkeypatch.setenv("SA_HIP_GROUP_FINISH", "0")   # every group through the rounds (the finisher has its own test below)
    for name, t, L in runs:
        got, stats = {}, {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_LOCAL_ROUNDS", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                stats[mode] = idx.build_stats()
                assert idx.verify() == 0, (name, mode, stats[mode])
                got[mode] = idx.sa_u32().copy()
        assert stats["1"]["rounds"] >= 1, (name, stats["1"])
        assert np.array_equal(got["1"], got["0"]), (name, stats)
        if name in ("words", "d2_300k", "repeat_block"):
            assert np.array_equal(got["1"], oracle.sais(t).astype(np.uint32)), name
        if name == "words_L20":
            assert np.array_equal(got["1"], oracle.truncated_sa(t, 20)), name
        # the global sort launches fewer passes when the rounds go through LDS
        if name in ("words", "blocks"):
            assert stats["1"]["radix_passes"] < stats["0"]["radix_passes"], (name, stats)


def test_load_refuses_suffix_array_with_out_of_range_entries(gpu, oracle):
    """An adopted SA is range-checked on the device: an entry >= n would send the query kernel's text reads out of
    bounds, so sa_hip_index_load returns SA_HIP_EINVAL instead of adopting it."""
    t = cases.small_texts()["d2_300k"]
    sa = oracle.sais(t).astype(np.uint32)
    with gpu.DeviceIndex(t.size, 0) as idx:
        idx.load(t, sa, 0)
        assert idx.verify() == 0
        for bad_value in (t.size, t.size + 12345, 0xFFFFFFFF):
            bad = sa.copy()
            bad[t.size // 2] = bad_value
            with pytest.raises(gpu.SaHipError) as e:
                idx.load(t, bad, 0)
            assert e.value.code == -1
            with pytest.raises(gpu.SaHipError):   # and the handle holds no index afterwards
                idx.query_batch([b"the"])
        idx.load(t, sa, 0)   # a good array is adopted again
        assert idx.query_batch([b"a"])["first"][0] != 0xFFFFFFFF


def test_widen_device_is_the_libsais64_layout(gpu, oracle):
    """sa_hip_index_widen_device (the pass bench.py times as part of the 64-bit SA build): int64[n] == (int64)SA,
    for sizes that exercise the 16-byte body and the element-wise tail."""
    import torch
    texts = cases.small_texts()
    for name in ("d2_300k", "banana", "len1"):
        t = texts[name] if name in texts else np.frombuffer(name.encode(), np.uint8)
        for cut in (0, 1, 3):
            tt = t[:t.size - cut] if t.size > cut else t
            with gpu.DeviceIndex(max(tt.size, 1), 0) as idx:
                idx.build(tt)
                out = torch.full((tt.size + 4,), -7, dtype=torch.int64, device="cuda:0")
                torch.cuda.synchronize()
                idx.widen_device(out.data_ptr())
                idx.sync()
                got = out.cpu().numpy()
                assert np.array_equal(got[:tt.size], oracle.sais(tt).astype(np.int64)), (name, cut)
                assert (got[tt.size:] == -7).all()
                assert idx.build_stats()["widen_ms"] >= 0.0


def test_group_finisher_matches_global_rounds(gpu, oracle, monkeypatch):
    """Groups that fit a tile are refined to the end by one workgroup in LDS (group_finish.hpp: fetch characters, stable
    radix sort by (group, characters), split, drop singletons, until nothing is tied; deferred writes; groups it cannot
    finish -- larger than a tile, or still tied after its round limit -- stay on the global path untouched).  Same suffix
    array as with SA_HIP_GROUP_FINISH=0, verified on the device and against the oracle: word text at several
    truncation depths (the finisher stops at depth L, ties in text order), short initial keys (dense active set, groups
    of every size), a 60 000-character run (one group far larger than a tile), blocks repeated 12 times (groups of 12
    that no bounded number of rounds separates: the finisher must give up and leave them to the doubling rounds), a
    skewed alphabet, near-random text without the tiny-group finisher (pairs), texts that end inside a group."""
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "0")   # long repeats through the rounds here; the periodic-run shortcut has its own test
    from suffixarray_amd import synth
    rng = np.random.default_rng(77)
    words = synth.d2_words(5_000_000)
    run = synth.d2_words(3_000_000).copy()
    run[1_000_000:1_060_000] = ord("q")
    skew = rng.choice(np.array([97, 98, 99, 100, 122], dtype=np.uint8), 3_000_000, p=[0.9, 0.04, 0.03, 0.02, 0.01])
    blk = np.tile(rng.integers(97, 101, 200_000, dtype=np.uint8), 12)
    tail = np.concatenate([synth.d2_words(400_000), np.frombuffer(b"abcabcabcabcabcabcabcabcabcabcabcabcabc" * 50, np.uint8)])
    st = cases.small_texts()
    runs = [("words", words, 0, {}), ("words_L1", words, 1, {}), ("words_L9", words, 9, {}), ("words_L20", words, 20, {}),
            ("words_L33", words, 33, {}), ("words_k4", words, 0, {"SA_HIP_INITIAL_CHARS": "4"}),
            ("words_nopilot", words, 0, {"SA_HIP_PILOT": "0"}), ("words_nopilot_L32", words, 32, {"SA_HIP_PILOT": "0"}),
            ("run", run, 0, {}), ("skew", skew, 0, {}), ("skew_L7", skew, 7, {}), ("blocks", blk, 0, {}), ("tail", tail, 0, {}),
            ("d1_pairs", synth.d1_uniform27(4_500_000), 0, {"SA_HIP_TINY": "0"}),
            ("d1_pairs_L10", synth.d1_uniform27(4_500_000), 10, {"SA_HIP_TINY": "0"}),
            ("d2_300k", st["d2_300k"], 0, {}), ("repeat_block", st["repeat_block"], 0, {}), ("fib", st["fib"], 0, {}),
            ("all_a_70000", st["all_a_70000"], 0, {}), ("period7", st["period7"], 0, {}), ("r2_30000", st["r2_30000"], 0, {}),
            ("with_nul", st["with_nul"], 0, {}), ("highbit", st["highbit"], 0, {}),
            # every byte value occurs: 9-bit codes (code 256 does not fit a byte)
            ("bytes256", np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 3000, dtype=np.uint8)] * 3), 0, {}),
            ("perm256", st["perm256"], 0, {}), ("r256_30000", st["r256_30000"], 0, {})]
    for name, t, L, env in runs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got, stats = {}, {}
        for mode in ("1", "v2", "0"):
            # "v2": round 4's restructured finisher (group_finish2_kernel: per-group depth, large groups split apart by one
            # wave each, finals straight to their SA slots, failed groups restored) -- measured no faster, kept for A/B
            monkeypatch.setenv("SA_HIP_GROUP_FINISH", "0" if mode == "0" else "1")
            monkeypatch.setenv("SA_HIP_FIN_V2", "1" if mode == "v2" else "0")
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                stats[mode] = idx.build_stats()
                assert idx.verify() == 0, (name, mode, stats[mode])
                got[mode] = idx.sa_u32().copy()
        monkeypatch.delenv("SA_HIP_FIN_V2", raising=False)
        for k in env:
            monkeypatch.delenv(k, raising=False)
        assert np.array_equal(got["1"], got["0"]), (name, stats)
        assert np.array_equal(got["v2"], got["0"]), (name, stats)
        assert stats["v2"]["finisher_runs"] > 0 or stats["1"]["finisher_runs"] == 0, (name, stats)
        assert stats["0"]["finisher_runs"] == 0
        if name.startswith("words") and L not in (1, 9):   # (L <= the initial key length: nothing to refine)
            assert stats["1"]["finisher_resolved"] > 0 and stats["1"]["active_total"] < stats["0"]["active_total"], (name, stats)
        if name in ("d1_pairs", "d1_pairs_L10"):
            assert stats["1"]["finisher_resolved"] > 0 and stats["1"]["rounds"] == 0, (name, stats["1"])
        if name == "blocks":   # nothing to finish: groups of 12 with common prefixes of 200 000 characters
            assert stats["1"]["finisher_resolved"] < blk.size // 100 and stats["1"]["doubling_rounds"] > 0, stats["1"]
        if L == 0 and t.size <= 5_000_000:
            assert np.array_equal(got["1"], oracle.sais(t).astype(np.uint32)), name
        elif L:
            assert np.array_equal(got["1"], oracle.truncated_sa(t, L)), name


def test_build_device64_is_the_libsais64_layout(gpu, oracle, monkeypatch):
    """sa_hip_index_build_device64 (bench.py's 64-bit build): the int64 array equals (int64)SA whether it comes out of
    the narrow sort's last pass + the patch of the refined slots (near-random text with tied pairs, word text with
    millions of refined slots through finisher and rounds, a truncated build) or from the widening pass at the end
    (texts too short for the narrow plan, 12-byte-record plan)."""
    monkeypatch.setenv("SA_HIP_PERIOD_FINISH", "0")   # long repeats through the rounds here; the periodic-run shortcut has its own test
    import torch
    from suffixarray_amd import synth
    runs = [("d1", synth.d1_uniform27(4_500_000), 0, {}, True), ("d1_L12", synth.d1_uniform27(4_500_000), 12, {}, True),
            ("words", synth.d2_words(5_000_000), 0, {"SA_HIP_PILOT": "0"}, True), ("words_L20", synth.d2_words(5_000_000), 20, {"SA_HIP_PILOT": "0"}, True),
            ("words_plain", synth.d2_words(5_000_000), 0, {"SA_HIP_NARROW": "0"}, False), ("small", cases.small_texts()["d2_300k"], 0, {}, False),
            ("banana", cases.small_texts()["banana"], 0, {}, False), ("len1", cases.small_texts()["len1"], 0, {}, False)]
    # fused AND long repeats: the finisher gives up (fin_useful cleared), doubling rounds swap the slot lists many times
    # while the FIRST list (the slots to patch) has to stay intact
    rng = np.random.default_rng(9)
    blocks = np.tile(rng.integers(97, 123, 400_000, dtype=np.uint8), 12)
    run = synth.d2_words(5_000_000).copy()
    run[2_000_000:2_060_000] = ord("q")
    runs += [("blocks", blocks, 0, {"SA_HIP_PILOT": "0"}, True), ("run", run, 0, {"SA_HIP_PILOT": "0"}, True)]
    # the 10-byte-record plan (pilot's key: 11 characters) writes the int64 copy in its last pass too
    runs += [("words48", synth.d2_words(5_000_000), 0, {}, True), ("words48_L20", synth.d2_words(5_000_000), 20, {}, True),
             ("blocks48", blocks, 0, {}, True)]
    for name, t, L, env, fused in runs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with gpu.DeviceIndex(t.size, 0) as idx:
            idx.build(t, L)   # uploads the text
            out = torch.full((t.size + 2,), -7, dtype=torch.int64, device="cuda:0")
            torch.cuda.synchronize()
            idx.build_device64(idx.text_dev, t.size, out.data_ptr(), L)
            idx.sync()
            st = idx.build_stats()
            assert bool(st["widen_fused"]) == fused, (name, st)
            if name in ("blocks", "run", "blocks48"):
                assert st["doubling_rounds"] > 0, (name, st)
            assert bool(st["narrow48"]) == name.endswith(("48", "48_L20")), (name, st)
            assert idx.verify() == 0, name
            got = out.cpu().numpy()
            assert np.array_equal(got[:t.size], idx.sa_u32().astype(np.int64)), name
            assert (got[t.size:] == -7).all()
            exp = oracle.truncated_sa(t, L) if L else oracle.sais(t)
            assert np.array_equal(got[:t.size], exp.astype(np.int64)), name
        for k in env:
            monkeypatch.delenv(k, raising=False)


def test_wide_sort_from_the_text_matches_key_array(gpu, oracle, monkeypatch):
    """Keys of more than 40 bits are sorted as 12-byte records; pass 0 of that sort reads the text itself
    (text_low_pass_kernel: keys assembled in registers, ranked by the LOWEST digit, next-pass histogram on the way)
    instead of a key array written by keygen_kernel (SA_HIP_WIDE_TEXT_PASS=0): same suffix array bit for bit, verified,
    equal to the oracle -- word text (12 five-bit characters), forced key lengths (9..12 characters: last passes of 5, 2,
    7 and 4 bits), a four-letter alphabet (21 characters of 3 bits), 200 symbols (8-bit codes), truncated builds, texts
    whose last tile is partial, and a 256-symbol text (9-bit codes: the text pass does not apply, both runs use keygen)."""
    from suffixarray_amd import synth
    rng = np.random.default_rng(123)
    words = synth.d2_words(5_000_003)
    dna = (rng.integers(0, 4, 3_000_000).astype(np.uint8) + 97)
    sym200 = rng.choice(np.arange(20, 220, dtype=np.uint8), 2_500_000)
    monkeypatch.setenv("SA_HIP_NARROW48", "0")   # this test is about the 12-byte-record plan (the 10-byte plan has its own above)
    runs = [("words", words, 0, {}), ("words_L30", words, 30, {}), ("words_k9", words, 0, {"SA_HIP_INITIAL_CHARS": "9"}),
            ("words_k10", words, 0, {"SA_HIP_INITIAL_CHARS": "10"}), ("words_k11", words, 0, {"SA_HIP_INITIAL_CHARS": "11"}),
            ("d1_plain", synth.d1_uniform27(4_600_000), 0, {"SA_HIP_NARROW": "0"}), ("d1_plain_k12", synth.d1_uniform27(300_000), 0, {"SA_HIP_INITIAL_CHARS": "12"}),
            ("dna", dna, 0, {}), ("sym200", sym200, 0, {"SA_HIP_INITIAL_CHARS": "7"}),
            ("bytes256", rng.integers(0, 256, 2_000_000).astype(np.uint8), 0, {"SA_HIP_INITIAL_CHARS": "6"}),
            ("d2_300k", cases.small_texts()["d2_300k"], 0, {}), ("r27_65537", cases.small_texts()["r27_65537"], 0, {"SA_HIP_INITIAL_CHARS": "12"})]
    for name, t, L, env in runs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got, st = {}, {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SA_HIP_WIDE_TEXT_PASS", mode)
            with gpu.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                st[mode] = idx.build_stats()
                assert idx.verify() == 0, (name, mode, st[mode])
                got[mode] = idx.sa_u32().copy()
                pats = cases.query_patterns(t, 500, rng)
                assert np.array_equal(idx.query_batch(pats), oracle.query_batch(t, got[mode], L if L else 0xFFFFFFFF, pats)), (name, mode)
        for k in env:
            monkeypatch.delenv(k, raising=False)
        assert np.array_equal(got["1"], got["0"]), (name, st)
        assert st["0"]["pass_launches"][0] > 0 and st["0"]["text_top_pass"] == 0, (name, st["0"])     # the 12-byte-record plan ran
        if name != "bytes256":
            assert st["1"]["text_top_pass"] == 1, (name, st["1"])
            assert st["1"]["radix_bytes"] < st["0"]["radix_bytes"], name
        exp = oracle.truncated_sa(t, L) if L else oracle.sais(t).astype(np.uint32)
        assert np.array_equal(got["1"], exp), name
    monkeypatch.delenv("SA_HIP_WIDE_TEXT_PASS", raising=False)
    monkeypatch.delenv("SA_HIP_NARROW48", raising=False)
