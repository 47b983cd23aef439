import sys, os
import numpy as np
cnt, shift, mask, p, npass = [int(x) for x in open("/tmp/sa_pass_meta.txt").read().split()]
kin = np.fromfile("/tmp/sa_pass_kin.bin", dtype=np.uint64); vin = np.fromfile("/tmp/sa_pass_vin.bin", dtype=np.uint32)
kout = np.fromfile("/tmp/sa_pass_kout.bin", dtype=np.uint64); vout = np.fromfile("/tmp/sa_pass_vout.bin", dtype=np.uint32)
TILE = 8192
tiles = (cnt + TILE - 1) // TILE; tpc = (tiles + 7) // 8
d = ((kin >> np.uint64(shift)) & np.uint64(mask)).astype(np.int64)
dout = ((kout >> np.uint64(shift)) & np.uint64(mask)).astype(np.int64)
print("output digit-sorted?", bool(np.all(np.diff(dout) >= 0)))
# source position of every output record via its (unique) value
nmax = int(max(vin.max(), vout.max())) + 1
pos_of = np.full(nmax, -1, dtype=np.int64); pos_of[vin] = np.arange(cnt)
src = pos_of[vout]
print("outputs with unknown value:", int((src < 0).sum()))
ok = src >= 0
print("key matches source key:", int((kout[ok] == kin[src[ok]]).sum()), "of", int(ok.sum()))
# per digit region: is src increasing?
hist = np.bincount(d, minlength=256); base = np.concatenate([[0], np.cumsum(hist)])
hout = np.bincount(dout, minlength=256)
print("digit histogram in == out:", bool(np.array_equal(hist, hout)), "diff digits", np.nonzero(hist != hout)[0][:10], (hout - hist)[np.nonzero(hist != hout)[0][:10]])
for dg in np.nonzero(hist)[0]:
    seg = src[base[dg]:base[dg + 1]]
    dec = np.nonzero(np.diff(seg) < 0)[0]
    if dec.size:
        print("digit", dg, "region [%d,%d) non-monotone at %d places; first at slot %d: src %d (tile %d chunk %d) -> %d (tile %d chunk %d)" % (
            base[dg], base[dg + 1], dec.size, base[dg] + dec[0], seg[dec[0]], seg[dec[0]] // TILE, min(7, seg[dec[0]] // TILE // tpc),
            seg[dec[0] + 1], seg[dec[0] + 1] // TILE, min(7, seg[dec[0] + 1] // TILE // tpc)))
# count per (chunk, digit) true
chunk_of_pos = np.minimum(7, (np.arange(cnt) // TILE) // tpc)
true = np.zeros((8, 256), dtype=np.int64)
np.add.at(true, (chunk_of_pos, d), 1)
# observed layout: for each digit region, the sequence of chunks of sources and their counts
for dg in np.nonzero(hist)[0][:6]:
    seg = src[base[dg]:base[dg + 1]]
    ch = np.minimum(7, seg // TILE // tpc)
    obs = np.bincount(ch[seg >= 0], minlength=8)
    print("digit", dg, "true per-chunk", true[:, dg].tolist(), "observed per-chunk", obs.tolist())
# missing source records (never appear in output)
seen = np.zeros(cnt, dtype=bool); seen[src[ok]] = True
miss = np.nonzero(~seen)[0]
print("missing source records:", miss.size, miss[:30], "tiles", (miss[:30] // TILE), "offs", (miss[:30] % TILE), "digits", d[miss[:30]])
dup = np.nonzero(np.bincount(src[ok], minlength=cnt) > 1)[0]
print("duplicated source records:", dup.size, dup[:30], "tiles", dup[:30] // TILE, "offs", dup[:30] % TILE, "digits", d[dup[:30]])
small = np.fromfile("/tmp/sa_pass_small.bin", dtype=np.uint32)
tick = small[:64].reshape(8, 8)
hist = small[64:64 + 8 * 8 * 256].reshape(8, 8, 256)
basearr = small[64 + 8 * 8 * 256:].reshape(8, 256)
print("tickets of this pass", tick[p].tolist(), "chunk tile counts", [min(tpc, max(0, tiles - c * tpc)) for c in range(8)])
dh = hist[p].astype(np.int64) - true
bad = np.argwhere(dh != 0)
print("device hist(pass) vs true per-chunk counts: mismatches", bad.shape[0])
for c, dg in bad[:20]:
    print("   chunk", c, "digit", dg, "device", hist[p][c, dg], "true", true[c, dg])
