"""Fused int8 loop (1M x 768, 64 queries, k = 10): per-batch time and launch duration against option settings.
usage: python tools/gpu_scan8_fused.py "name=value,..;name=value,.." (one run per ';'-separated option set)"""
import sys, time
import numpy as np, torch
import os, pathlib
sys.path.insert(0, ".")
from rag_uq_amd import _native as nat
if os.environ.get("RQ_AB_LIB"):          # development: time another build of the library (e.g. tools/ab/librq_hip_old.so)
    nat.LIB_PATH = pathlib.Path(os.environ["RQ_AB_LIB"]).resolve()

n, k = 1_000_000, 10
sets = (sys.argv[1] if len(sys.argv) > 1 else "thr_mult8=1.25").split(";")
dev = torch.device("cuda:0")
idx = nat.NativeIndex(768, 0)
idx.reserve(n)
for c in range(8):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000); del x
gq = torch.Generator(device=dev); gq.manual_seed(4321)
qs = [torch.randn((64, 768), device=dev, generator=gq) for _ in range(16)]
st = torch.cuda.Stream(device=dev)
idx.set_option("pipeline", 2)
outs = [(torch.empty((64, k), device=dev), torch.empty((64, k), device=dev, dtype=torch.int64), torch.zeros((64,), device=dev, dtype=torch.int32)) for _ in range(16)]
def loop(steps):
    with torch.cuda.stream(st):
        for i in range(steps):
            j = i % 16
            if os.environ.get("RQ_NOHINT") != "1":      # RQ_NOHINT=1: every call prepares its own queries (no prep workgroups in the fused launch)
                idx.search_hint_next_device(qs[(i + 1) % 16], 64, st.cuda_stream)
            idx.search_device(qs[j], 64, k, 0, outs[j][0], outs[j][1], None, outs[j][2], st.cuda_stream)
        idx.search_flush_device(st.cuda_stream)
    torch.cuda.synchronize()
for rep in range(2):
    for opts in sets:
        pairs = [o.split("=") for o in opts.split(",") if o]
        idx.set_option("scan8", 1); idx.set_option("thr_mult8", 1.25); idx.set_option("tail_stop", 0); idx.set_option("pipeline", 2)   # defaults, then the set
        for name, v in pairs:
            idx.set_option(name, float(v))
        idx.set_option("profile", 0)
        loop(64)
        idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 4)
        t0 = time.perf_counter(); loop(400); dt = (time.perf_counter() - t0) / 400
        t = idx.timing()
        unc = int(sum(int(o[2].sum()) for o in outs))
        print(f"{opts:40s}: {dt * 1e6:7.1f} us/batch  launch {t['scan_ms'] / max(t['scan_launches'], 1) * 1e3:6.1f} us  uncertified {unc}", flush=True)
        idx.set_option("tail_stop", 0)
        loop(16)
idx.close()
