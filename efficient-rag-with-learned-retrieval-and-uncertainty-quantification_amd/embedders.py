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
                 max_length: int = 512, batch_size: int = 256, random_init: bool = False, num_layers: Optional[int] = None):
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

    def _tokenize(self, texts: Sequence[str]):
        torch = self.torch
        if self.tokenizer is not None:
            enc = self.tokenizer(list(texts), padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
            return enc["input_ids"].to(self.device), enc["attention_mask"].to(self.device)
        # stand-in: utf-8 bytes shifted into the vocabulary (random-init smoke path only)
        rows = [[(b % (self.vocab - 10)) + 5 for b in t.encode()[: self.max_length]] or [5] for t in texts]
        L = max(len(r) for r in rows)
        ids = torch.zeros((len(rows), L), dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            mask[i, : len(r)] = 1
        return ids.to(self.device), mask.to(self.device)

    def embed(self, texts: Sequence[str]) -> np.ndarray:
        torch = self.torch
        outs = []
        with torch.inference_mode():
            for lo in range(0, len(texts), self.batch_size):
                ids, mask = self._tokenize(texts[lo: lo + self.batch_size])
                h = self.model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
                m = mask.unsqueeze(-1).float()
                outs.append(((h * m).sum(1) / m.sum(1).clamp_min(1.0)).cpu().numpy())
        return np.concatenate(outs, axis=0).astype(np.float32) if outs else np.zeros((0, self.dim), np.float32)


def default_embedder(embedding_model: str = "nomic-embed-text"):
    """RAG_UQ_EMBEDDER_PATH=<local nomic-embed-text directory> selects the real model; otherwise the
    reference's own fallback (HashEmbedder) is used, with the same warning the reference logs (:51)."""
    path = os.environ.get("RAG_UQ_EMBEDDER_PATH")
    if path:
        return NomicBertEmbedder(model_path=path)
    logger.warning("no local %s weights configured (RAG_UQ_EMBEDDER_PATH). Using fallback embeddings.", embedding_model)
    return HashEmbedder()
