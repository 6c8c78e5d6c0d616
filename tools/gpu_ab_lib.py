"""Interleaved A/B of two builds of librq_hip.so in ONE process (the boxes drift by several percent within a minute, so
separate runs cannot resolve a 2 % difference): the fused int8 loop at 1M x 768, 64 queries, k = 10, alternating 100
batches of build A and 100 of build B.  usage: python tools/gpu_ab_lib.py <other .so> [rounds] [k]"""
import importlib.util, pathlib, sys, time
import torch
sys.path.insert(0, ".")
import rag_uq_amd
from rag_uq_amd import _native as nat_new

other = pathlib.Path(sys.argv[1]).resolve()
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
spec = importlib.util.spec_from_file_location("rq_native_other", nat_new.__file__)
nat_old = importlib.util.module_from_spec(spec); spec.loader.exec_module(nat_old)
nat_old.LIB_PATH = other
dev = torch.device("cuda:0")
n = 1_000_000
gq = torch.Generator(device=dev); gq.manual_seed(4321)
qs = [torch.randn((64, 768), device=dev, generator=gq) for _ in range(16)]
st = torch.cuda.Stream(device=dev)
sides = {}
order = (("other", nat_old), ("this", nat_new))
if len(sys.argv) > 4 and sys.argv[4] == "swap":      # which index is allocated first (placement in HBM differs)
    order = order[::-1]
for name, nat in order:
    idx = nat.NativeIndex(768, 0); idx.reserve(n)
    for c in range(8):
        g = torch.Generator(device=dev); g.manual_seed(1235 + c)
        x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
        idx.add_f16_device(x, 125_000); del x
    idx.set_option("pipeline", 2); idx.set_option("scan8", 1)
    outs = [(torch.empty((64, k), device=dev), torch.empty((64, k), device=dev, dtype=torch.int64), torch.zeros((64,), device=dev, dtype=torch.int32)) for _ in range(16)]
    sides[name] = (idx, outs, [], [])
def loop(idx, outs, steps):
    with torch.cuda.stream(st):
        for i in range(steps):
            j = i % 16
            idx.search_hint_next_device(qs[(i + 1) % 16], 64, st.cuda_stream)
            idx.search_device(qs[j], 64, k, 0, outs[j][0], outs[j][1], None, outs[j][2], st.cuda_stream)
        idx.search_flush_device(st.cuda_stream)
    torch.cuda.synchronize()
for name in sides:
    loop(sides[name][0], sides[name][1], 64)
for r in range(rounds):
    for name in (("other", "this") if r % 2 == 0 else ("this", "other")):
        idx, outs, per_batch, per_launch = sides[name]
        idx.reset_timing(); idx.set_option("profile", 1); idx.set_option("profile_stride", 4)
        t0 = time.perf_counter(); loop(idx, outs, 100); dt = (time.perf_counter() - t0) / 100
        t = idx.timing(); idx.set_option("profile", 0)
        per_batch.append(dt * 1e6); per_launch.append(t["scan_ms"] / max(t["scan_launches"], 1) * 1e3)
for name, (idx, outs, pb, pl) in sides.items():
    unc = int(sum(int(o[2].sum()) for o in outs))
    print(f"{name:6s}: per batch {sum(pb) / len(pb):7.1f} us (min {min(pb):6.1f})  launch {sum(pl) / len(pl):6.1f} us (min {min(pl):6.1f})  uncertified {unc}   rounds: " + " ".join(f"{v:.1f}" for v in pl))
same = all(torch.equal(a[1], b[1]) for a, b in zip(sides["other"][1], sides["this"][1]))
print("rows identical between the two builds:", same)
