"""Where configs[3]'s encode time goes: tokenisation (host), forward (eager / hipGraph replay), pooling + D2H."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_uq_amd
from rag_uq_amd.embedders import NomicBertEmbedder
torch.manual_seed(0)
emb = NomicBertEmbedder(random_init=True, num_layers=12, device="cuda:0", dtype="float16", batch_size=256)
queries = [f"question {i}: what is known about topic {i * 7919 % 1000} and the river number {i % 13}?" for i in range(256)]
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
ms_tok, (ids, mask, _lens) = t(lambda: emb._tokenize(queries))
print(f"tokenise (host + H2D): {ms_tok:.2f} ms   ids {tuple(ids.shape)}")
with torch.inference_mode():
    ms_fwd, out = t(lambda: emb.model(input_ids=ids, attention_mask=mask).last_hidden_state)
    print(f"forward eager: {ms_fwd:.2f} ms")
    def pool():
        h = out.float(); m = mask.unsqueeze(-1).float()
        return ((h * m).sum(1) / m.sum(1).clamp_min(1.0)).cpu().numpy()
    ms_pool, _ = t(pool)
    print(f"pool + D2H: {ms_pool:.2f} ms")
ms_all, _ = t(lambda: emb.embed(queries))
print(f"embed() total (fused path {'on' if emb.fused is not None else 'off'}): {ms_all:.2f} ms")
if emb.fused is not None:
    with torch.inference_mode():
        ms_f, fo = t(lambda: emb.fused(ids, mask))
        h = emb.model(input_ids=ids, attention_mask=mask).last_hidden_state.float(); m_ = mask.unsqueeze(-1).float()
        so = (h * m_).sum(1) / m_.sum(1).clamp_min(1.0)
    cos = torch.nn.functional.cosine_similarity(fo, so, dim=1)
    print(f"forward fused (4 GEMMs + 4 kernels per layer, incl. embeddings + pooling): {ms_f:.2f} ms   min cosine vs stock {float(cos.min()):.6f}")
# hipGraph replay of the forward at this shape
try:
    sids, smask = ids.clone(), mask.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.inference_mode():
        for _ in range(3): emb.model(input_ids=sids, attention_mask=smask)
    torch.cuda.current_stream().wait_stream(s)
    with torch.inference_mode(), torch.cuda.graph(g):
        gout = emb.model(input_ids=sids, attention_mask=smask).last_hidden_state
    ms_g, _ = t(lambda: g.replay())
    print(f"forward hipGraph replay: {ms_g:.2f} ms   max|diff| vs eager {float((gout.float() - out.float()).abs().max()):.3e}")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
