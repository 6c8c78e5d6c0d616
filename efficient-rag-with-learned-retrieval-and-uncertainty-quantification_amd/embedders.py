"""Text -> vector front ends for DenseIndex (replaces reference rag_uq/streaming_index.py:267-288).

The reference asks an Ollama server for `nomic-embed-text` one text per HTTP request (:276-279),
falls back to a SHA-256 pseudo embedding when the `ollama` module is absent (:269-273) and returns a
768-d zero vector when the request fails (:281-284).  Here an embedder is an in-process object with

    dim: int
    embed(texts: Sequence[str]) -> np.ndarray [len(texts), dim] float32     (may raise)

DenseIndex keeps the zero-vector-on-failure rule around it.
"""
from __future__ import annotations

import hashlib
import logging
import os
from typing import Callable, Optional, Sequence

import numpy as np

logger = logging.getLogger(__name__)


class HashEmbedder:
    """The reference's no-Ollama fallback, bit for bit: bytes of SHA-256(text) / 255.

    streaming_index.py:272-273 slices the 32-byte digest with [:384], so the vector is 32-d."""

    dim = 32

    def embed(self, texts: Sequence[str]) -> np.ndarray:
        out = np.empty((len(texts), self.dim), dtype=np.float32)
        for i, t in enumerate(texts):
            digest = hashlib.sha256(t.encode()).digest()
            out[i] = [float(b) / 255.0 for b in digest]
        return out


class RandomProjectionEmbedder:
    """Deterministic offline stand-in for a text encoder: every lower-cased whitespace token owns a fixed Gaussian
    direction (seeded by SHA-256 of the token), a text is the sum of its tokens' directions.  Cosine similarity then
    follows token overlap, which is what synthetic retrieval workloads need when no model weights are available
    (BASELINE.json configs[4] stand-in; HashEmbedder has no such structure).  Not a language model."""

    def __init__(self, dim: int = 768):
        self.dim = int(dim)
        self._cache = {}

    def _token(self, tok: str) -> np.ndarray:
        v = self._cache.get(tok)
        if v is None:
            seed = int.from_bytes(hashlib.sha256(tok.encode()).digest()[:8], "little")
            v = np.random.default_rng(seed).standard_normal(self.dim).astype(np.float32)
            self._cache[tok] = v
        return v

    def embed(self, texts: Sequence[str]) -> np.ndarray:
        out = np.zeros((len(texts), self.dim), dtype=np.float32)
        for i, t in enumerate(texts):
            for tok in t.lower().split():
                out[i] += self._token(tok)
        return out


class CallableEmbedder:
    """Wrap any `fn(list_of_texts) -> array[n, dim]` (e.g. a client of an embedding service)."""

    def __init__(self, fn: Callable[[Sequence[str]], np.ndarray], dim: int):
        self.fn = fn
        self.dim = int(dim)

    def embed(self, texts: Sequence[str]) -> np.ndarray:
        v = np.asarray(self.fn(list(texts)), dtype=np.float32)
        if v.ndim != 2 or v.shape != (len(texts), self.dim):
            raise ValueError(f"embedder returned shape {v.shape}, expected {(len(texts), self.dim)}")
        return v


class FusedNomicBertForward:
    """The NomicBert encoder stack with the memory-bound work fused into four gfx950 kernels (csrc/rq_encoder.hip, include/rq.h
    "encoder pieces"); the GEMMs stay with the framework (hipBLASLt through `torch.nn.functional.linear`).

    Built from a `transformers` NomicBertModel (its weights are re-used, q/k/v and gate/up concatenated so that a layer is four
    GEMMs instead of seven).  Per layer: qkv GEMM -> rq_nb_attention_f16 (rotary + softmax(QK^T/8) V on the matrix cores,
    prefix mask from the sequence lengths) -> o GEMM -> rq_nb_add_layernorm_f16 -> gate|up GEMM -> rq_nb_swiglu_f16 -> down GEMM
    -> rq_nb_add_layernorm_f16; then rq_nb_mean_pool_f16.  Preconditions (checked by `usable`): fp16 on a GPU, head_dim 64,
    hidden size <= 1536, right-padded batches of at most 512 tokens.  Architecture as in
    transformers/models/nomic_bert/modeling_nomic_bert.py of the installed package (post-LN, rotate-half rotary, SwiGLU, no
    projection biases).  rocprofv3 at configs[3]'s shape (256 x 68 tokens, 12 layers): stock forward 10.3 ms, see DESIGN.md 6.
    """

    MAX_SEQ = 512
    PACK_BELOW_PERCENT = 97        # batches with less than 3 % padding rows keep the padded layout

    def __init__(self, model):
        import torch
        from . import _native
        self.torch, self.nat = torch, _native
        cfg = model.config
        self.heads = int(cfg.num_attention_heads)
        self.hidden = int(cfg.hidden_size)
        self.inter = int(cfg.intermediate_size)
        self.eps = float(cfg.layer_norm_eps)
        rope = getattr(cfg, "rope_parameters", None) or {}
        self.theta = float(rope.get("rope_theta", getattr(cfg, "rope_theta", 10000.0)))
        self.ok = (getattr(cfg, "head_dim", self.hidden // self.heads) == 64 and self.heads * 64 == self.hidden and self.hidden <= 1536
                   and self.hidden % 8 == 0 and self.inter % 8 == 0 and rope.get("rope_type", "default") == "default"
                   and cfg.hidden_act in ("silu", "swish"))
        self.emb = model.embeddings
        self.layers = []
        self._rope = None          # rotary table [MAX_SEQ][64] fp32, built on first use
        if not self.ok:
            return
        for lyr in model.encoder.layers if hasattr(model, "encoder") else model.layers:
            a, m = lyr.self_attn, lyr.mlp
            if any(x.bias is not None for x in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj)):
                self.ok = False
                return
            self.layers.append(dict(
                wqkv=torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous(),
                wo=a.o_proj.weight.contiguous(),
                wgu=torch.cat([m.gate_proj.weight, m.up_proj.weight], 0).contiguous(),
                wd=m.down_proj.weight.contiguous(),
                g1=lyr.post_attention_layernorm.weight.contiguous(), b1=lyr.post_attention_layernorm.bias.contiguous(),
                g2=lyr.post_mlp_layernorm.weight.contiguous(), b2=lyr.post_mlp_layernorm.bias.contiguous()))

    def usable(self, ids, mask, lengths_host=None) -> bool:
        """lengths_host: the sequence lengths as a host array when the caller KNOWS the batch is right-padded (the tokenizer built the
        mask on the host): the device-side check of the mask -- and its synchronisation -- is skipped."""
        torch = self.torch
        if not (self.ok and ids.is_cuda and ids.shape[1] <= self.MAX_SEQ and self.layers and self.layers[0]["wqkv"].dtype == torch.float16):
            return False
        if lengths_host is not None:
            return True
        lengths = mask.sum(1)
        return bool((mask == (torch.arange(mask.shape[1], device=mask.device)[None, :] < lengths[:, None])).all())   # valid tokens first

    def __call__(self, ids, mask, lengths_host=None):
        """[B][L] token ids + right-padded attention mask -> [B][hidden] fp32 mean-pooled embeddings (device tensor).  lengths_host
        (numpy, the same lengths the mask encodes): nothing is read back from the device before the first kernel."""
        torch, nat = self.torch, self.nat
        F = torch.nn.functional
        B, L = ids.shape
        H = self.hidden
        st = torch.cuda.current_stream(ids.device).cuda_stream
        if lengths_host is not None:
            lens_np = np.ascontiguousarray(lengths_host, dtype=np.int32)
            lengths = torch.from_numpy(lens_np).to(ids.device)
            total = int(lens_np.sum())
        else:
            lengths = mask.sum(1).to(torch.int32).contiguous()
            lens_np = None
            total = int(lengths.sum())
        pos = torch.arange(L, device=ids.device)[None, :]
        # Ragged batches run PACKED: the token rows of all sequences back to back (no padding rows), an offset table instead of the
        # lengths -- GEMMs, LayerNorm and SwiGLU then work on sum(lengths) rows instead of B * longest.  (Batches that are all but full
        # keep the padded layout: the gather below is not free.)
        packed = total * 100 <= B * L * self.PACK_BELOW_PERCENT
        if packed:
            keep = mask.bool()
            if lens_np is not None:
                offs_np = np.zeros((B + 1,), np.int32)
                np.cumsum(lens_np, out=offs_np[1:])
                offs = torch.from_numpy(offs_np).to(ids.device)
            else:
                offs = torch.zeros((B + 1,), device=ids.device, dtype=torch.int32)
                offs[1:] = torch.cumsum(lengths, 0)
            T = total
            h = self.emb(input_ids=ids[keep][None, :], position_ids=pos.expand(B, L)[keep][None, :]).reshape(T, H).contiguous()
        else:
            T = B * L
            h = self.emb(input_ids=ids, position_ids=pos).reshape(T, H).contiguous()
        if self._rope is None or self._rope.device != ids.device:
            self._rope = torch.empty((self.MAX_SEQ, 64), device=ids.device, dtype=torch.float32)
            nat.nb_rope_table(self._rope, self.MAX_SEQ, self.theta, st)
        ctx = torch.empty((T, H), device=ids.device, dtype=torch.float16)
        act = torch.empty((T, self.inter), device=ids.device, dtype=torch.float16)
        for w in self.layers:
            qkv = F.linear(h, w["wqkv"])
            if packed:
                nat.nb_attention_packed(qkv, offs, self._rope, ctx, B, L, self.heads, st)
            else:
                nat.nb_attention(qkv, lengths, self._rope, ctx, B, L, self.heads, st)
            o = F.linear(ctx, w["wo"])
            nat.nb_add_layernorm(o, h, w["g1"], w["b1"], h, T, H, self.eps, st)
            gu = F.linear(h, w["wgu"])
            nat.nb_swiglu(gu, act, T, self.inter, st)
            d = F.linear(act, w["wd"])
            nat.nb_add_layernorm(d, h, w["g2"], w["b2"], h, T, H, self.eps, st)
        out = torch.empty((B, H), device=ids.device, dtype=torch.float32)
        if packed:
            nat.nb_mean_pool_packed(h, offs, out, B, L, H, st)
        else:
            nat.nb_mean_pool(h, lengths, out, B, L, H, st)
        return out


class NomicBertEmbedder:
    """nomic-embed-text forward pass on PyTorch-ROCm (BASELINE.json configs[3]).

    Architecture from the locally installed `transformers.models.nomic_bert` (12 layers, 768 hidden,
    12 heads, SwiGLU, rotary); mean pooling over the attention mask; raw text, no task prefix,
    exactly what the reference sends (:276-279).  Weights and tokenizer come ONLY from `model_path`
    (a local directory): nothing is ever downloaded.  With `random_init=True` the architecture is
    instantiated with random weights and a byte-level stand-in tokenizer -- a smoke path that
    exercises shapes and throughput, not retrieval quality.
    """

    dim = 768

    def __init__(self, model_path: Optional[str] = None, device: str = "cuda:0", dtype: str = "float16",
                 max_length: int = 512, batch_size: int = 256, random_init: bool = False, num_layers: Optional[int] = None,
                 fused: bool = True):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.max_length = int(max_length)
        self.batch_size = int(batch_size)
        self.dtype = getattr(torch, dtype)
        self.tokenizer = None
        if model_path is None and not random_init:
            raise FileNotFoundError(
                "nomic-embed-text weights are not present: pass model_path=<local directory> "
                "(HF_HUB_OFFLINE: nothing is downloaded) or random_init=True for a smoke run")
        from transformers import AutoConfig
        if random_init:
            from transformers.models.nomic_bert import NomicBertConfig, NomicBertModel
            cfg = NomicBertConfig()
            if num_layers is not None:
                cfg.num_hidden_layers = int(num_layers)
            self.model = NomicBertModel(cfg)
            self.vocab = int(cfg.vocab_size)
        else:
            if not os.path.isdir(model_path):
                raise FileNotFoundError(f"model_path {model_path!r} is not a local directory")
            from transformers import AutoModel, AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
            self.model = AutoModel.from_pretrained(model_path, local_files_only=True)
            self.vocab = int(AutoConfig.from_pretrained(model_path, local_files_only=True).vocab_size)
        self.model = self.model.to(self.device, self.dtype).eval()
        self.dim = int(self.model.config.hidden_size)
        # fp16 on a GPU: the fused gfx950 forward (batches it cannot take -- more than 512 tokens, left padding -- go through
        # the stock module, same weights)
        self.fused = None
        if fused and self.device.type == "cuda" and self.dtype == torch.float16:
            with torch.inference_mode():
                f = FusedNomicBertForward(self.model)
            self.fused = f if f.ok else None

    def _tokenize(self, texts: Sequence[str]):
        torch = self.torch
        if self.tokenizer is not None:
            enc = self.tokenizer(list(texts), padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
            m = enc["attention_mask"]
            lens = m.sum(1)
            right_padded = bool((m == (torch.arange(m.shape[1])[None, :] < lens[:, None])).all())       # (on the host: the tokenizer's own tensors)
            return enc["input_ids"].to(self.device), m.to(self.device), (lens.numpy().astype(np.int32) if right_padded else None)
        # stand-in: utf-8 bytes shifted into the vocabulary (random-init smoke path only)
        raw = [np.frombuffer(t.encode()[: self.max_length], dtype=np.uint8) for t in texts]
        L = max(max((len(r) for r in raw), default=1), 1)
        ids = np.zeros((len(raw), L), dtype=np.int64)
        mask = np.zeros((len(raw), L), dtype=np.int64)
        for i, r in enumerate(raw):
            if len(r) == 0:
                ids[i, 0] = 5; mask[i, 0] = 1                      # an empty text is one token
            else:
                ids[i, : len(r)] = r.astype(np.int64) % (self.vocab - 10) + 5
                mask[i, : len(r)] = 1
        return torch.from_numpy(ids).to(self.device), torch.from_numpy(mask).to(self.device), mask.sum(1).astype(np.int32)

    def embed_device(self, texts: Sequence[str]):
        """[len(texts)][dim] fp32 embeddings as ONE tensor on the embedder's device, rows in the caller's order (what
        DenseIndex.search_batch hands to rq_search_device: the query matrix never leaves HBM).  May raise."""
        torch = self.torch
        texts = list(texts)
        if not texts:
            return torch.zeros((0, self.dim), device=self.device, dtype=torch.float32)
        # batches of similar length: a batch is padded to its longest text, so ragged inputs (passages) are sorted by length
        # first and the rows are put back in the caller's order afterwards
        order = sorted(range(len(texts)), key=lambda i: len(texts[i])) if len(texts) > self.batch_size else list(range(len(texts)))
        single = len(order) <= self.batch_size
        out = None if single else torch.empty((len(texts), self.dim), device=self.device, dtype=torch.float32)
        with torch.inference_mode():
            for lo in range(0, len(order), self.batch_size):
                sel = order[lo: lo + self.batch_size]
                ids, mask, lens_host = self._tokenize([texts[i] for i in sel])
                if self.fused is not None and self.fused.usable(ids, mask, lens_host):
                    v = self.fused(ids, mask, lens_host)
                else:
                    h = self.model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
                    m = mask.unsqueeze(-1).float()
                    v = (h * m).sum(1) / m.sum(1).clamp_min(1.0)
                if single:
                    return v          # (one batch in the caller's order: no scatter)
                out[torch.as_tensor(sel, device=self.device)] = v
        return out

    def embed(self, texts: Sequence[str]) -> np.ndarray:
        if not len(texts):
            return np.zeros((0, self.dim), np.float32)
        return self.embed_device(texts).cpu().numpy()


def default_embedder(embedding_model: str = "nomic-embed-text"):
    """RAG_UQ_EMBEDDER_PATH=<local nomic-embed-text directory> selects the real model; otherwise the
    reference's own fallback (HashEmbedder) is used, with the same warning the reference logs (:51)."""
    path = os.environ.get("RAG_UQ_EMBEDDER_PATH")
    if path:
        return NomicBertEmbedder(model_path=path)
    logger.warning("no local %s weights configured (RAG_UQ_EMBEDDER_PATH). Using fallback embeddings.", embedding_model)
    return HashEmbedder()
