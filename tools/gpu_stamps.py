"""When do the tail workgroups of the fused launch actually run?  Wall-clock stamps (100 MHz) of every workgroup of the
last fused launch (development hook rq_debug_stamps, not part of include/rq.h)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_uq_amd import _native as nat
N = int(os.environ.get("RQ_N", 1_000_000)); dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0); idx.reserve(N)
for c in range(0, N, 125_000):
    n = min(125_000, N - c)
    idx.add_f16_device(torch.nn.functional.normalize(torch.randn((n, 768), device=dev), dim=1).half().contiguous(), n)
lib = nat.load_library()
idx.set_option("pipeline", 2)
q = torch.randn((64, 768), device=dev)
sc = torch.empty((64, 10), device=dev); rw = torch.empty((64, 10), device=dev, dtype=torch.int64); st = torch.empty((64,), device=dev, dtype=torch.int32)
s = torch.cuda.Stream(device=dev)
lib.rq_debug_stamps(idx._h, 1, None, 0)
for opt in filter(None, os.environ.get("RQ_OPTS", "").split(",")):
    name, val = opt.split("="); idx.set_option(name, float(val))
for _ in range(8):                                     # warm-up: workspaces, code objects
    idx.search_device(q, 64, 10, 0, sc, rw, None, st, s.cuda_stream)
torch.cuda.synchronize()
idx.set_option("profile", 1); idx.reset_timing()
for _ in range(12):
    idx.search_device(q, 64, 10, 0, sc, rw, None, st, s.cuda_stream)
torch.cuda.synchronize()
tm = idx.timing()
print(f"options [{os.environ.get('RQ_OPTS', '')}]: HIP-event duration of the scan launches: {tm['scan_ms'] * 1e3 / tm['scan_launches']:.1f} us (avg of {tm['scan_launches']})")
out = np.zeros((8192, 4), dtype=np.uint64)
lib.rq_debug_stamps(idx._h, 1, out.ctypes.data, 8192)
idx.search_flush_device(s.cuda_stream); torch.cuda.synchronize()
live = out[:, 1] > 0
t = out[live][:, :2].astype(np.int64); ids = np.nonzero(live)[0]
hw = out[live][:, 2].astype(np.int64); xcc = out[live][:, 3].astype(np.int64) & 0xf
t0 = t[:, 0].min()
us = (t - t0) / 100.0
scan = ids < 512
print(f"{live.sum()} workgroups stamped; scan workgroups: start {us[scan, 0].min():.1f}..{us[scan, 0].max():.1f} us, end {us[scan, 1].min():.1f}..{us[scan, 1].max():.1f} us, mean end {us[scan, 1].mean():.1f}")
tl = ~scan
if tl.any():
    print(f"tail workgroups ({tl.sum()}): start {us[tl, 0].min():.1f}..{us[tl, 0].max():.1f} us (median {np.median(us[tl, 0]):.1f}), end {us[tl, 1].min():.1f}..{us[tl, 1].max():.1f} us (median {np.median(us[tl, 1]):.1f}), duration median {np.median(us[tl, 1] - us[tl, 0]):.1f} us")
# who is slow?  by XCD, by CU (se/sh/cu bits of HW_ID), by number of quads
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
nquads = (N + 63) // 64
nloc = np.array([(b + 1) * nquads // 512 - b * nquads // 512 for b in ids[scan]])
e = us[scan, 1]
print("scan end by quads per workgroup:", {int(n): (round(float(e[nloc == n].mean()), 1), int((nloc == n).sum())) for n in np.unique(nloc)})
print("scan end by XCD:", {int(x): round(float(e[xcc[scan] == x].mean()), 1) for x in np.unique(xcc[scan])})
per_cu = {}
for c_, e_ in zip(cuid[scan], e): per_cu.setdefault(int(c_), []).append(float(e_))
cnt = np.array([len(v) for v in per_cu.values()])
print("scan workgroups per CU: histogram", {int(k): int((cnt == k).sum()) for k in np.unique(cnt)}, "CUs used", len(per_cu))
for k in np.unique(cnt):
    print(f"  CUs with {k} scan workgroup(s): mean end {np.mean([np.mean(v) for v in per_cu.values() if len(v) == k]):.1f} us")
tcu = {}
for c_ in cuid[tl]: tcu[int(c_)] = tcu.get(int(c_), 0) + 1
withtail = np.array([tcu.get(int(c_), 0) for c_ in cuid[scan]])
print("scan end vs tail workgroups hosted by the same CU:", {int(k): round(float(e[withtail == k].mean()), 1) for k in np.unique(withtail)})
