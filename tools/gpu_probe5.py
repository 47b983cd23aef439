import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi
rng = np.random.default_rng(1)
blk = rng.integers(97, 123, 1 << 20, dtype=np.uint8)
t = np.tile(blk, 64)
for attempt in range(3):
    with _capi.DeviceIndex(t.size, 0) as idx:
        try:
            idx.build(t)
            print("build ok", idx.verify(), flush=True)
        except Exception as e:
            print("build FAIL", e, flush=True)
    if os.path.exists("/tmp/sa_pass_meta.txt"):
        break
if not os.path.exists("/tmp/sa_pass_meta.txt"):
    print("no dump"); sys.exit(0)
cnt, shift, mask, p, npass = [int(x) for x in open("/tmp/sa_pass_meta.txt").read().split()]
kin = np.fromfile("/tmp/sa_pass_kin.bin", dtype=np.uint64); vin = np.fromfile("/tmp/sa_pass_vin.bin", dtype=np.uint32)
kout = np.fromfile("/tmp/sa_pass_kout.bin", dtype=np.uint64); vout = np.fromfile("/tmp/sa_pass_vout.bin", dtype=np.uint32)
print("pass dump: cnt", cnt, "shift", shift, "mask", mask, "pass", p, "/", npass, flush=True)
d = ((kin >> np.uint64(shift)) & np.uint64(mask)).astype(np.int64)
order = np.argsort(d, kind="stable")
ek, ev = kin[order], vin[order]
badk = np.nonzero(kout != ek)[0]; badv = np.nonzero(vout != ev)[0]
print("bad key slots", badk.size, "bad val slots", badv.size, flush=True)
TILE = 8192
tiles = (cnt + TILE - 1) // TILE; tpc = (tiles + 7) // 8
print("tiles", tiles, "tpc", tpc)
inv = np.empty(cnt, dtype=np.int64); inv[order] = np.arange(cnt)   # source position -> expected dest
bad = np.union1d(badk, badv)
# group bad dest slots into runs
runs = np.split(bad, np.nonzero(np.diff(bad) != 1)[0] + 1) if bad.size else []
print("runs", len(runs))
for r in runs[:40]:
    s0 = r[0]
    src = order[s0]   # expected source position of the record that should be at dest s0
    print("dest run [%d,%d) len %d dest_tile %d dest_chunk %d | expected from src pos %d (src tile %d, chunk %d, off %d) digit %d | got key==expected? %s val got %d exp %d" % (
        r[0], r[-1] + 1, r.size, s0 // TILE, min(7, (s0 // TILE) // tpc), src, src // TILE, min(7, (src // TILE) // tpc), src % TILE, d[src],
        kout[s0] == ek[s0], vout[s0], ev[s0]))
# where did the expected records go? search their values in vout
if bad.size:
    lost_vals = ev[bad[:2000]]
    pos_of = {}
    idxs = np.nonzero(np.isin(vout, lost_vals))[0]
    print("occurrences of the first lost values in vout:", idxs.size)
    for v in lost_vals[:10]:
        print("  val", v, "expected dest", int(np.nonzero(ev == v)[0][0]), "found at", np.nonzero(vout == v)[0][:4])
