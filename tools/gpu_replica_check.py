"""What bench.py --gpus N does after its timed steps, on one GPU: the index of a build (narrow key array, fused directory)
against an adopted copy of its text + SA (sa_hip_index_load_device: u64 key array, directory by binary search), the
same 1M-query batch through both, ranges must be identical (bench line: replica_query_ok)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from suffixarray_amd import _capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
Q = 1_000_000
text = synth.d1_uniform27(N)
buf, off = synth.query_batch(text, Q, 16)
dev = torch.device("cuda:0")
pat_t = torch.from_numpy(np.concatenate([buf, np.zeros(64, np.uint8)])).to(dev)
off_t = torch.from_numpy(off.view(np.int64)).to(dev)
out_a = torch.zeros(Q * 2, dtype=torch.int32, device=dev)
out_b = torch.zeros(Q * 2, dtype=torch.int32, device=dev)
idx = _capi.DeviceIndex(N, 0)
idx.build(text)
print("build", idx.build_stats()["total_ms"], "ms narrow_k", idx.build_stats()["narrow_k"], flush=True)
idx.query_batch_device(pat_t.data_ptr(), off_t.data_ptr(), Q, out_a.data_ptr()); idx.sync()
print("own query", idx.query_stats()["kernel_ms"], "ms", flush=True)
tx_t = torch.from_numpy(text).to(dev)
sa_t = torch.from_numpy(idx.sa_u32().view(np.int32)).to(dev)
rep = _capi.DeviceIndex(N, 0)
t0 = time.time(); rep.load_device(tx_t.data_ptr(), sa_t.data_ptr(), N, 0); rep.sync(); print("adopt %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
rep.query_batch_device(pat_t.data_ptr(), off_t.data_ptr(), Q, out_b.data_ptr()); rep.sync()
print("replica query", rep.query_stats()["kernel_ms"], "ms", flush=True)
ok = bool(torch.equal(out_a, out_b))
print("replica_query_ok", ok)
sys.exit(0 if ok else 1)
