"""Round 3 experiment: the encoder's four library GEMMs per layer (configs[3] shape: 256 x 68 tokens, 12 layers, fp16) with the library's
default algorithm choice against PyTorch's TunableOp (every rocBLAS / hipBLASLt solution timed once per shape, the best kept)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd  # noqa: F401
from rag_uq_amd.embedders import NomicBertEmbedder

torch.manual_seed(0)
emb = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256)
queries = [f"question {i}: what is known about topic {i * 7919 % 1000} and the river number {i % 13}?" for i in range(256)]

def timed(tag, reps=20):
    for _ in range(3): emb.embed_device(queries)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): v = emb.embed_device(queries)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{tag}: forward {dt * 1e3:.3f} ms per 256 texts", flush=True)
    return v

v0 = timed("default algorithms")
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True)
tun.set_max_tuning_duration(8); tun.set_max_tuning_iterations(5)
tun.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03_tunableop_gfx950.csv"))
t0 = time.perf_counter(); emb.embed_device(queries); torch.cuda.synchronize()
print(f"tuning pass: {time.perf_counter() - t0:.1f} s", flush=True)
tun.tuning_enable(False)
v1 = timed("tuned algorithms")
tun.write_file()
cos = torch.nn.functional.cosine_similarity(v0, v1, dim=1).min().item()
print(f"min cosine default vs tuned: {cos:.6f}", flush=True)
for r in tun.get_results(): print(r, flush=True)
