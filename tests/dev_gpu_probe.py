"""Development probe, kept under tests/ because it checks against the oracle (run on the GPU box): parity on small shapes, then scan-variant timings at 1M rows."""
import importlib.util, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rag_uq_amd  # noqa
from rag_uq_amd import _native as nat
from oracle import dense_oracle as orc

out = {}
def log(*a):
    print(*a, flush=True)

def check(idx, x16, q, k, metric=0, tag=""):
    s, r = idx.search(q, k, metric)
    gs, gr = orc.dense_topk(q, x16, k, metric)
    ok_rows = bool((r == gr).all())
    ds = float(np.abs(s - gs).max()) if s.size else 0.0
    t = idx.timing()
    log(f"[parity] {tag} N={len(x16)} B={len(q)} k={k} metric={metric}: rows_equal={ok_rows} max|ds|={ds:.2e} widened={t['widened']} exact={t['exact_scans']}")
    if not ok_rows:
        bad = np.argwhere(r != gr)[:5]
        for b, j in bad:
            log("   mismatch q", b, "rank", j, "got", r[b, j], s[b, j], "want", gr[b, j], gs[b, j])
    return ok_rows and ds <= 1e-5

allok = True
log("devices:", nat.device_count(), nat.load_library().rq_version())
for (n, dim, fast) in [(1000, 768, 1), (37, 32, 1), (5000, 384, 1), (100000, 768, 1), (100000, 768, 0), (1000, 768, 0)]:
    x16 = orc.synthetic_corpus(n, dim, seed=1234)
    idx = nat.NativeIndex(dim, 0)
    idx.set_option("fast_tail", fast)
    idx.add_f16(x16[: n // 2]); idx.add_f16(x16[n // 2:])
    assert len(idx) == n
    back = idx.get_rows_f16(0, n)
    assert (back.view(np.uint16) == x16.view(np.uint16)).all(), "stored rows differ"
    for B, k in [(1, 10), (5, 1), (64, 10), (70, 50), (3, 100)]:
        q = orc.synthetic_queries(B, dim, seed=4321 + B)
        allok &= check(idx, x16, q, k, 0, f"dim={dim} fast={fast}")
    q = orc.synthetic_queries(8, dim, seed=99)
    allok &= check(idx, x16, q, 10, 1, f"dim={dim} ip")
    idx.close()
# f32 add path
x32 = np.random.default_rng(7).standard_normal((3000, 768)).astype(np.float32) * 3.0
idx = nat.NativeIndex(768, 0); idx.add_f32(x32, True)
want = orc.prepare_rows_f32(x32, True)
got = idx.get_rows_f16(0, 3000)
log("[add_f32] stored rows bit-equal:", bool((got.view(np.uint16) == want.view(np.uint16)).all()), "mismatches", int((got.view(np.uint16) != want.view(np.uint16)).sum()))
allok &= check(idx, got, orc.synthetic_queries(16, 768, 5), 10, 0, "add_f32")
idx.close()
# forced fallbacks: eps huge -> certificate always fails -> widen -> exact scan
x16 = orc.synthetic_corpus(20000, 768, seed=3)
idx = nat.NativeIndex(768, 0); idx.add_f16(x16); idx.set_option("eps", 10.0)
allok &= check(idx, x16, orc.synthetic_queries(4, 768, 11), 10, 0, "eps=10 (exact-scan ladder)")
idx.close()
# duplicates + zero query + zero rows
x16 = orc.synthetic_corpus(4096, 768, seed=5); x16[100:400] = x16[7]; x16[1000:1010] = 0
idx = nat.NativeIndex(768, 0); idx.add_f16(x16)
q = orc.synthetic_queries(6, 768, 13); q[2] = 0; q[3] = x16[7].astype(np.float32)
allok &= check(idx, x16, q, 20, 0, "dups/zero")
idx.close()
log("PARITY_ALL_OK" if allok else "PARITY_FAILED")
out["parity_ok"] = bool(allok)

# ---------------------------------------------------------------- perf at 1M x 768
import torch
dev = torch.device("cuda:0")
N = 1_000_000
g = torch.Generator(device=dev); g.manual_seed(1235)
idx = nat.NativeIndex(768, 0); idx.reserve(N)
for lo in range(0, N, 250_000):
    x = torch.randn((250_000, 768), device=dev, generator=g, dtype=torch.float32)
    x = torch.nn.functional.normalize(x, dim=1).half().contiguous()
    idx.add_f16_device(x, x.shape[0])
del x
torch.cuda.synchronize()
B, k = 64, 10
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
sc = torch.empty((B, k), device=dev); rows = torch.empty((B, k), device=dev, dtype=torch.int64)
keys = torch.empty((B, k), device=dev, dtype=torch.int64); st = torch.empty((B,), device=dev, dtype=torch.int32)
res = []
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [(torch.empty((B, 128), device=dev), torch.empty((B, 128), device=dev, dtype=torch.int64),
         torch.empty((B, 128), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32)) for _ in range(2)]
def run(tag, iters=40, nstreams=1, k=10):
    idx.set_option("profile", 1); idx.reset_timing()
    def go(i):
        j = i % nstreams
        o = outs[j]
        idx.search_device(qs[i % 8], B, k, 0, o[0], o[1], o[2], o[3], streams[j].cuda_stream)
    for i in range(6):
        go(i)
    torch.cuda.synchronize(); idx.reset_timing()
    t0 = time.perf_counter()
    for i in range(iters):
        go(i)
    for st_ in streams: idx.search_flush_device(st_.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    st = outs[0][3]
    t = idx.timing()
    scan_us = t["scan_ms"] / max(t["scan_launches"], 1) * 1e3
    gbs = N * 1536 / (scan_us * 1e-6) / 1e9
    bad = int(st.sum().item())
    log(f"[perf] {tag}: scan {scan_us:8.1f} us  {gbs:7.1f} GB/s ({gbs/8000:.1%} of 8 TB/s)   end-to-end {dt*1e6:8.1f} us/batch  {B/dt:9.0f} q/s  uncertified={bad}")
    res.append(dict(tag=tag, scan_us=scan_us, gbs=gbs, e2e_us=dt * 1e6, qps=B / dt, uncertified=bad))

for ring, pf, wg in [(4, 1, 3), (4, 4, 2), (6, 12, 2), (5, 6, 2)]:
    idx.set_option("kstage", 2); idx.set_option("ring", ring); idx.set_option("prefetch", pf); idx.set_option("wg_per_cu", wg); idx.set_option("nt", 1)
    for ns, pipe in ((1, 0), (2, 0), (1, 1), (2, 1)):
        idx.set_option("pipeline", pipe)
        run(f"ring={ring} pf={pf} wg/cu={wg} streams={ns} pipeline={pipe} k=10", nstreams=ns)
        for st_ in streams: idx.search_flush_device(st_.cuda_stream)
        torch.cuda.synchronize()
idx.set_option("kstage", 1); idx.set_option("ring", 2); idx.set_option("prefetch", 1); idx.set_option("wg_per_cu", 2); idx.set_option("pipeline", 1)
for kk in (1, 50, 100):
    run(f"default pipeline streams=1 k={kk}", nstreams=1, k=kk)
for st_ in streams: idx.search_flush_device(st_.cuda_stream)
torch.cuda.synchronize(); idx.set_option("pipeline", 0)
idx.set_option("fast_tail", 0); run("generic tail streams=2 k=10", nstreams=2); idx.set_option("fast_tail", 1)
out["perf"] = res
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
# spot-check the 1M result against the oracle on a 2-query slice (oracle over 1M rows in chunks)
q_host = qs[0][:3].cpu().numpy()
xs = idx.get_rows_f16(0, N)
for kk in (10, 100):
    s_host, r_host = idx.search(q_host, kk, 0)
    gs, gr = orc.dense_topk(q_host, xs, kk, 0)
    log(f"[1M parity k={kk}] rows_equal=", bool((r_host == gr).all()), "max|ds|=", float(np.abs(s_host - gs).max()), idx.timing())
