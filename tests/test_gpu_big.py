"""Texts beyond 2^32 - 2 bytes: the 64-bit-index build (csrc/big_build.hpp; the counterpart of libsais64's true 64-bit
path, libsais64.c:6684 -> libsais64_main).  At small sizes the same code is compared bit for bit with the oracle (the
entry point accepts any n); at n = 4.4e9 -- past every 32-bit index -- through the size-independent properties: the
on-device sufcheck with 64-bit indices and text spot checks."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _build_on_device(gpu, t):
    import torch
    n = int(t.size)
    text_d = torch.from_numpy(np.ascontiguousarray(t)).to("cuda:0") if n else torch.empty(16, dtype=torch.uint8, device="cuda:0")
    sa_d = torch.empty(max(n, 1), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    st = gpu.libsais64_device(text_d.data_ptr(), sa_d.data_ptr(), n)
    bad = gpu.sufcheck64_device(text_d.data_ptr(), sa_d.data_ptr(), n)
    return sa_d[:n].cpu().numpy(), st, bad


def test_big_path_matches_oracle_at_small_sizes(gpu, oracle):
    """Every small text of the suite (classic, adversarial: all-a, Fibonacci, block repeats, two symbols, 256 symbols, NUL and
    high-bit bytes) + word text + a text with a long run: the 64-bit-index build gives the oracle's suffix array, its own
    sufcheck agrees, the doubling rounds run where the text has long repeats."""
    from suffixarray_amd import synth
    st = cases.small_texts()
    texts = dict(st)
    texts["words_2m"] = synth.d2_words(2_000_000)
    texts["d1_3m"] = synth.d1_uniform27(3_000_000)
    run = synth.d2_words(1_500_000).copy()
    run[400_000:470_000] = ord("q")
    texts["run"] = run
    rng = np.random.default_rng(5)
    texts["blocks"] = np.tile(rng.integers(97, 101, 50_000, dtype=np.uint8), 9)
    texts["one"] = np.frombuffer(b"x", np.uint8)
    texts["two_equal"] = np.frombuffer(b"aa", np.uint8)
    seen_rounds = 0
    for name, t in texts.items():
        if t.size == 0:
            continue
        sa, stats, bad = _build_on_device(gpu, t)
        assert bad == 0, (name, stats)
        assert np.array_equal(sa, oracle.sais(t).astype(np.int64)), (name, stats)
        seen_rounds += stats["rounds"]
    assert seen_rounds > 10   # all-a / Fibonacci / block repeats went through prefix doubling


def test_sufcheck64_sees_a_wrong_array(gpu, oracle):
    import torch
    from suffixarray_amd import synth
    t = synth.d2_words(300_000)
    sa = oracle.sais(t).astype(np.int64)
    text_d = torch.from_numpy(t).to("cuda:0")
    for kind in ("swap", "dup", "range"):
        bad = sa.copy()
        if kind == "swap":
            bad[1000], bad[1001] = bad[1001], bad[1000]
        elif kind == "dup":
            bad[5] = bad[6]
        else:
            bad[77] = t.size + 3
        sa_d = torch.from_numpy(bad).to("cuda:0")
        torch.cuda.synchronize()
        assert gpu.sufcheck64_device(text_d.data_ptr(), sa_d.data_ptr(), t.size) > 0, kind
    sa_d = torch.from_numpy(sa).to("cuda:0")
    torch.cuda.synchronize()
    assert gpu.sufcheck64_device(text_d.data_ptr(), sa_d.data_ptr(), t.size) == 0


def test_beyond_uint32_4p4e9_verified(gpu):
    """n = 4.4e9 > 2^32 - 2: D1 text generated on the host, built on the device with 64-bit suffix indices (41 bytes of HBM
    per character: 180 GB), checked by the on-device sufcheck (the suffix array is unique: verified <=> bit-exact) and by
    comparing sampled neighbours in the text on the host; entries beyond 2^32 occur."""
    import torch
    from suffixarray_amd import synth
    n = 4_400_000_000
    free, _ = torch.cuda.mem_get_info(0)
    if free < 46 * n:
        pytest.skip("needs %d GB of free HBM" % (46 * n >> 30))
    t = synth.d1_uniform27(n)
    text_d = torch.from_numpy(t).to("cuda:0")
    sa_d = torch.empty(n, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    st = gpu.libsais64_device(text_d.data_ptr(), sa_d.data_ptr(), n)
    assert st["initial_chars"] == 12 and st["sort_passes"] >= 8, st
    assert gpu.sufcheck64_device(text_d.data_ptr(), sa_d.data_ptr(), n) == 0, st
    assert int(sa_d.max().item()) == n - 1 and int((sa_d > 0xFFFFFFFF).sum().item()) == n - (1 << 32)
    rng = np.random.default_rng(1)
    js = torch.from_numpy(rng.integers(1, n, 2000)).to("cuda:0")
    a = sa_d[js - 1].cpu().numpy(); b = sa_d[js].cpu().numpy()
    for x, y in zip(a, b):
        assert bytes(t[int(x):int(x) + 64]) < bytes(t[int(y):int(y) + 64]) or int(x) + 64 > n
    print("4.4e9 characters, 64-bit indices: %.1f ms on the device (%s)" % (st["total_ms"], st))
    # the drop-in call (host text in, host int64 array out): the same array -- every 100 003rd entry and both ends compared
    # (the device array was verified above and a suffix array is unique), the byte histogram beside it
    import time
    t0 = time.time()
    sa_h, freq = gpu.libsais64(t, want_freq=True)
    print("sa_hip_libsais64(T, SA, 4.4e9) host to host: %.1f s" % (time.time() - t0))
    assert sa_h.dtype == np.int64 and sa_h.size == n
    assert np.array_equal(sa_h[::100_003], sa_d[::100_003].cpu().numpy())
    assert np.array_equal(sa_h[:4096], sa_d[:4096].cpu().numpy()) and np.array_equal(sa_h[-4096:], sa_d[-4096:].cpu().numpy())
    assert int(freq.sum()) == n and int((freq > 0).sum()) == 27
