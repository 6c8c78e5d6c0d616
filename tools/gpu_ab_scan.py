"""Same-box A/B of two builds of the library (RQ_LIB_PATH): scan-kernel HIP-event time, alternating processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, rag_uq_amd
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0"); N = int(os.environ.get("RQ_N", 1000000)); B = 64; k = 10
idx = nat.NativeIndex(768, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, N, 125000):
    n = min(125000, N - lo)
    x = torch.nn.functional.normalize(torch.randn((n, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, n)
qs = [torch.randn((B, 768), device=dev, generator=g) for _ in range(8)]
o = (torch.empty((B, k), device=dev), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B,), device=dev, dtype=torch.int32))
idx.set_option("profile", 1)
for i in range(5): idx.search_device(qs[i %% 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
torch.cuda.synchronize(); idx.reset_timing(); t0 = time.perf_counter()
for i in range(40): idx.search_device(qs[i %% 8], B, k, 0, o[0], o[1], o[2], o[3], 0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
t = idx.timing()
print("scan %%.1f us  e2e(single stream, profiled) %%.1f us" %% (t["scan_ms"] * 1e3 / t["scan_launches"], dt * 1e6))
''' % ROOT
import glob
libs = {"current build": ""}
for f in sorted(glob.glob(os.path.join(ROOT, "_ab", "librq_*.so"))):      # alternative builds dropped into _ab/
    libs[os.path.basename(f)[6:-3]] = f
for rnd in range(3):
    for name, path in libs.items():
        env = dict(os.environ)
        if path: env["RQ_LIB_PATH"] = path
        else: env.pop("RQ_LIB_PATH", None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"round {rnd} {name}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
