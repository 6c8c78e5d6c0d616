"""Drop-in mirror of reference `rag_uq/streaming_index.py` with the dense path on MI355X.

Same module-level names, constructor signatures, defaults and return conventions as the reference
(Document :54-77, RetrievalResult :80-89, BM25Index :92-225, DenseIndex :228-373, HybridRetriever
:376-560, StreamingIndex :563-686), so `experiments/run_evaluation.py:165-167`,
`experiments/run_router_training.py:80-82` and `data/preprocessing/build_chroma_index.py:55-125`
run unchanged against it.  What differs underneath:

  * DenseIndex keeps the passage vectors as an fp16 matrix in HBM and answers `search` with an exact
    cosine top-k computed by hand-written HIP kernels (librq_hip.so, include/rq.h) instead of
    ChromaDB/HNSW; embeddings come from an in-process embedder instead of one Ollama HTTP request
    per text.  There is no CPU fallback for this class.
  * BM25Index restates rank-bm25's BM25Okapi (not installed here) over an inverted index, so adding
    documents no longer rebuilds the whole model.  It stays on the CPU (BASELINE.json configs[4]).
  * Batched entry points are added next to the single-query ones (`search_batch`,
    `hybrid_search_batch`, `get_scores_for_router_batch`); the originals are thin wrappers.
"""
from __future__ import annotations

import gc
import io
import json
import logging
import math
import os
import pickle
from collections import Counter
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .embedders import default_embedder

logger = logging.getLogger(__name__)


def _gpu_backend_available() -> bool:
    """True when librq_hip.so loads and sees a device.  A missing library raises (loud), a missing
    GPU only disables dense retrieval, like the reference's HAS_CHROMA switch (:31-37, :412-420)."""
    return _native.device_count() > 0


# =============================================================================================
# records (reference :54-89)
# =============================================================================================
@dataclass
class Document:
    """A document for indexing."""
    id: str
    text: str
    title: Optional[str] = None
    metadata: Optional[Dict[str, Any]] = None

    def to_dict(self) -> Dict[str, Any]:
        return {"id": self.id, "text": self.text, "title": self.title or "", "metadata": self.metadata or {}}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "Document":
        return cls(id=data["id"], text=data["text"], title=data.get("title"), metadata=data.get("metadata"))


@dataclass
class RetrievalResult:
    """Result from hybrid retrieval."""
    doc_id: str
    text: str
    bm25_score: float
    dense_score: float
    hybrid_score: Optional[float] = None
    title: Optional[str] = None
    metadata: Optional[Dict[str, Any]] = None


class _PlainUnpickler(pickle.Unpickler):
    """The BM25 snapshot holds dicts, lists, strings and floats only (reference :189-201), so loading it needs no global
    at all: refusing every class lookup keeps the file format and removes pickle's code-execution surface."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"BM25 snapshot must not reference {module}.{name}")


def _load_plain_pickle(path) -> Any:
    with open(path, "rb") as f:
        return _PlainUnpickler(io.BytesIO(f.read())).load()


def _atomic_write(path: Path, data: bytes) -> None:
    tmp = Path(str(path) + ".tmp")
    with open(tmp, "wb") as f:
        f.write(data)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


# =============================================================================================
# sparse side (reference :92-225).  CPU, host logic.
# =============================================================================================
class BM25Index:
    """BM25Okapi over an inverted index; same API, pickle layout and scores as the reference class.

    Scoring restates rank-bm25 0.2.2 `BM25Okapi` (requirements.txt:8; not vendored, not installed):
      idf(t)  = ln(N - n_t + 0.5) - ln(n_t + 0.5); negative idfs are replaced by 0.25 * mean(idf)
      score   = sum over query tokens t (with repetition) of
                idf(t) * f * (k1 + 1) / (f + k1 * (1 - b + b * dl / avgdl))
    `search` keeps the reference's selection verbatim (:172-177): argsort, reversed, first top_k,
    score > 0 only.

    Persistence.  The reference rebuilds BM25Okapi and re-pickles the WHOLE index on every add_documents (:141-146,
    :185-201): O(N^2) bytes over an indexing run (10 000 full dumps for 1M passages in batches of 100).  Here an add
    appends its documents to `<persist_path>.log.jsonl`; the snapshot at `persist_path` keeps the reference's pickle
    layout and is rewritten (atomically) when the log has grown as large as the snapshot (so total snapshot bytes stay
    linear), on `save()` and on `close()`.  Loading = snapshot + replay of the log.  `snapshot_every=1` restores the
    reference's "pickle on every add".
    """

    EPSILON = 0.25
    SNAPSHOT_MIN_DOCS = 10_000

    def __init__(self, persist_path: Optional[str] = None, k1: float = 1.5, b: float = 0.75, *, snapshot_every: Optional[int] = None):
        self.persist_path = Path(persist_path) if persist_path else None
        self.snapshot_every = snapshot_every
        self._snapshot_docs = 0        # documents covered by the snapshot file
        self._log_file = None
        self.k1 = k1
        self.b = b
        self.documents: Dict[str, Document] = {}
        self.doc_ids: List[str] = []
        self.tokenized_corpus: List[List[str]] = []
        # inverted index: token -> (doc rows, term frequencies), appended in row order
        self._post_rows: Dict[str, List[int]] = {}
        self._post_tf: Dict[str, List[int]] = {}
        self._doc_len: List[int] = []
        self._total_len = 0
        self._idf: Optional[Dict[str, float]] = None   # None = stale
        if self.persist_path and (self.persist_path.exists() or self._log_path().exists()):
            self._load()

    @property
    def bm25(self):
        """The reference exposes the BM25Okapi object; callers only test it for None (:165)."""
        return self if self.doc_ids else None

    def _tokenize(self, text: str) -> List[str]:
        return text.lower().split()

    def _index_tokens(self, row: int, tokens: List[str]) -> None:
        for tok, tf in Counter(tokens).items():
            self._post_rows.setdefault(tok, []).append(row)
            self._post_tf.setdefault(tok, []).append(tf)
        self._doc_len.append(len(tokens))
        self._total_len += len(tokens)
        self._idf = None

    def add_documents(self, documents: List[Document]) -> int:
        new_count = 0
        for doc in documents:
            if doc.id in self.documents:
                continue
            self.documents[doc.id] = doc
            self.doc_ids.append(doc.id)
            toks = self._tokenize(doc.text)
            self.tokenized_corpus.append(toks)
            self._index_tokens(len(self.doc_ids) - 1, toks)
            new_count += 1
        if new_count:
            logger.info(f"Added {new_count} documents to BM25 index. Total: {len(self.doc_ids)}")
        if self.persist_path and new_count:
            self._persist_new(len(self.doc_ids) - new_count)
        return new_count

    def _ensure_idf(self) -> Dict[str, float]:
        if self._idf is not None:
            return self._idf
        n_docs = len(self.doc_ids)
        idf: Dict[str, float] = {}
        idf_sum = 0.0
        negative: List[str] = []
        for tok, rows in self._post_rows.items():   # insertion order = first-seen order, as BM25Okapi's nd dict
            n_t = len(rows)
            v = math.log(n_docs - n_t + 0.5) - math.log(n_t + 0.5)
            idf[tok] = v
            idf_sum += v
            if v < 0:
                negative.append(tok)
        if idf:
            eps = self.EPSILON * (idf_sum / len(idf))
            for tok in negative:
                idf[tok] = eps
        self._idf = idf
        return idf

    def _np_doc_len(self) -> np.ndarray:
        cache = getattr(self, "_doc_len_np", None)
        if cache is None or len(cache) != len(self._doc_len):
            cache = self._doc_len_np = np.asarray(self._doc_len, dtype=np.float64)
        return cache

    def _np_postings(self, tok: str, rows: List[int]) -> Tuple[np.ndarray, np.ndarray]:
        """numpy views of a token's posting list, rebuilt when the list has grown (the lists are append-only).  Converting the
        Python lists on every query was 70 % of a search over 50 k passages (cProfile: 4.9 of 7.0 ms per query)."""
        cache = self.__dict__.setdefault("_post_np", {})
        hit = cache.get(tok)
        if hit is None or len(hit[0]) != len(rows):
            hit = cache[tok] = (np.asarray(rows, dtype=np.int64), np.asarray(self._post_tf[tok], dtype=np.float64))
        return hit

    def _token_contribution(self, tok: str, rows: List[int], idf: Dict[str, float], avgdl: float) -> Tuple[np.ndarray, np.ndarray]:
        """(rows, idf * tf-saturation per row) of one token for the CURRENT corpus: the same expression the per-query loop used
        to evaluate, kept until the next add (idf and avgdl change with every added document)."""
        n_docs = len(self.doc_ids)
        cache = self.__dict__.setdefault("_contrib", {})
        if cache.get("#docs") != n_docs:
            cache.clear()
            cache["#docs"] = n_docs
        hit = cache.get(tok)
        if hit is None:
            r, f = self._np_postings(tok, rows)
            w = idf.get(tok) or 0
            hit = cache[tok] = (r, w * (f * (self.k1 + 1) / (f + self.k1 * (1 - self.b + self.b * self._np_doc_len()[r] / avgdl))))
        return hit

    def get_scores(self, tokenized_query: List[str]) -> np.ndarray:
        n_docs = len(self.doc_ids)
        scores = np.zeros(n_docs)
        if n_docs == 0:
            return scores
        idf = self._ensure_idf()
        avgdl = self._total_len / n_docs
        for tok in tokenized_query:
            rows = self._post_rows.get(tok)
            if not rows:
                continue
            r, c = self._token_contribution(tok, rows, idf, avgdl)
            scores[r] += c
        return scores

    @staticmethod
    def _select_topk(scores: np.ndarray, top_k: int) -> np.ndarray:
        """Rows of the reference's selection (:172-177: argsort, reversed, first top_k, score > 0 only) without sorting the whole
        score vector (80 % of a search over 50 k passages).  Exact ties -- which numpy's default argsort orders in an unspecified,
        build-dependent way -- are ordered as `np.argsort(scores, kind="stable")[::-1]` would: by DESCENDING row.  The same rule as
        librq_bm25.so (include/rq_bm25.h), so `search` and `search_batch` agree to the last bit."""
        pos = np.flatnonzero(scores > 0)
        if top_k <= 0 or pos.size == 0:
            return pos[:0]
        if pos.size > top_k:
            vals = scores[pos]
            kth = np.partition(vals, pos.size - top_k)[pos.size - top_k]
            pos = pos[vals >= kth]                     # every row tied with the k-th score takes part in the tie-break
        order = np.lexsort((-pos, -scores[pos]))
        return pos[order[:top_k]]

    SINGLE_NATIVE_AFTER: Optional[int] = 4   # one-query searches in a row over an unchanged corpus before they pay for the batch scorer's arrays (None: never)

    def search(self, query: str, top_k: int = 10) -> List[Tuple[str, float]]:
        if self.bm25 is None or not self.doc_ids:
            return []
        # One query per call is the reference's evaluation loop (experiments/run_evaluation.py:157-206).  The path below allocates and
        # selects over one float64 per PASSAGE per query (20 ms at 1M passages); the batch scorer (librq_bm25.so) walks the same posting
        # lists in 3.4 ms -- same additions, same bits -- but needs the CSR form of the index, which every add invalidates (2.3 s to
        # rebuild at 1M passages).  So: through the batch scorer when its arrays are current, or once a few searches in a row have
        # seen the same corpus (an evaluation loop); a build loop that alternates adds and searches stays on this path.
        n = len(self.doc_ids)
        if top_k > 0 and self.SINGLE_NATIVE_AFTER is not None and _native.bm25_available():
            c = self.__dict__.get("_csr_cache")
            current = c is not None and c["n_docs"] == n
            if not current:
                streak = self.__dict__.get("_single_streak", (0, -1))
                streak = (streak[0] + 1, n) if streak[1] == n else (1, n)
                self.__dict__["_single_streak"] = streak
            if current or streak[0] > self.SINGLE_NATIVE_AFTER:
                return self.search_batch([query], top_k, n_threads=1)[0]
        scores = self.get_scores(self._tokenize(query))
        return [(self.doc_ids[i], float(scores[i])) for i in self._select_topk(scores, top_k).tolist()]

    # ---- a whole batch of queries at once (extension; configs[4]: 500 questions per call) ---------------------------------
    def _csr(self) -> Dict[str, Any]:
        """The inverted index as CSR arrays for the CURRENT corpus (rebuilt after an add: idf and avgdl change with every document):
        token ids in first-seen order, postings concatenated, and every posting's contribution idf * tf-saturation evaluated with
        the SAME float64 expression as _token_contribution -- the batch path then only adds them, in query-token order."""
        n_docs = len(self.doc_ids)
        c = self.__dict__.get("_csr_cache")
        if c is not None and c["n_docs"] == n_docs:
            return c
        if c is not None and c.get("handle") is not None:
            _native.bm25_destroy(c["handle"])
        from itertools import chain
        idf = self._ensure_idf()
        avgdl = self._total_len / n_docs
        toks = list(self._post_rows)
        lens = np.fromiter((len(self._post_rows[t]) for t in toks), np.int64, len(toks))
        indptr = np.zeros(len(toks) + 1, np.int64)
        np.cumsum(lens, out=indptr[1:])
        nnz = int(indptr[-1])
        rows = np.fromiter(chain.from_iterable(self._post_rows.values()), np.int32, nnz)
        f = np.fromiter(chain.from_iterable(self._post_tf.values()), np.float64, nnz)
        w = np.repeat(np.fromiter(((idf.get(t) or 0) for t in toks), np.float64, len(toks)), lens)
        contrib = w * (f * (self.k1 + 1) / (f + self.k1 * (1 - self.b + self.b * self._np_doc_len()[rows] / avgdl)))
        c = {"n_docs": n_docs, "tid": {t: i for i, t in enumerate(toks)}, "indptr": indptr, "rows": rows, "contrib": contrib, "handle": None}
        if _native.bm25_available():
            c["handle"] = _native.bm25_create(indptr, rows, contrib, len(toks), n_docs)
        self.__dict__["_csr_cache"] = c
        return c

    def search_batch_rows(self, queries: Sequence[str], top_k: int = 10, *, n_threads: int = 0, use_native: Optional[bool] = None
                          ) -> Tuple[np.ndarray, np.ndarray]:
        """The selection of `search` for a whole batch, in ROW space: (rows int32 [B][top_k], -1 padded; scores float64 [B][top_k]).
        librq_bm25.so (include/rq_bm25.h) walks the posting lists of every query on the host cores -- term at a time, one float64
        accumulator per thread, the contributions added in query order (the same additions as get_scores: identical bits), top-k by a
        heap with the tie rule of _select_topk.  Without the library (`use_native=False`, or it is not built) the same arrays are
        scored with numpy, one query at a time."""
        B, k = len(queries), max(int(top_k), 1)
        if self.bm25 is None or not self.doc_ids or top_k <= 0 or B == 0:
            return np.full((B, k), -1, np.int32), np.zeros((B, k))
        c = self._csr()
        tid = c["tid"]
        q_tok: List[int] = []
        q_ptr = [0]
        for q in queries:
            for t in self._tokenize(q):
                j = tid.get(t)
                if j is not None:
                    q_tok.append(j)
            q_ptr.append(len(q_tok))
        if use_native and c["handle"] is None:
            raise _native.RqError("librq_bm25.so is not built: `make -C csrc` (or use_native=False for the numpy path)")
        native = c["handle"] is not None if use_native is None else (use_native and c["handle"] is not None)
        if native:
            return _native.bm25_topk(c["handle"], np.asarray(q_ptr, np.int64), np.asarray(q_tok, np.int32), B, k, n_threads)
        indptr, rows, contrib = c["indptr"], c["rows"], c["contrib"]
        out_rows, out_scores = np.full((B, k), -1, np.int32), np.zeros((B, k))
        for b in range(B):
            scores = np.zeros(len(self.doc_ids))
            for j in q_tok[q_ptr[b]: q_ptr[b + 1]]:
                lo, hi = indptr[j], indptr[j + 1]
                scores[rows[lo:hi]] += contrib[lo:hi]
            sel = self._select_topk(scores, k)
            out_rows[b, :len(sel)] = sel
            out_scores[b, :len(sel)] = scores[sel]
        return out_rows, out_scores

    def _ids_array(self) -> np.ndarray:
        cache = self.__dict__.get("_ids_np")
        if cache is None or len(cache) != len(self.doc_ids):
            cache = np.empty(len(self.doc_ids), dtype=object)
            cache[:] = self.doc_ids
            self.__dict__["_ids_np"] = cache
        return cache

    def search_batch(self, queries: Sequence[str], top_k: int = 10, *, n_threads: int = 0, use_native: Optional[bool] = None
                     ) -> List[List[Tuple[str, float]]]:
        """`[self.search(q, top_k) for q in queries]`, scored as one job (search_batch_rows)."""
        if self.bm25 is None or not self.doc_ids or top_k <= 0:
            return [[] for _ in queries]
        out_rows, out_scores = self.search_batch_rows(queries, top_k, n_threads=n_threads, use_native=use_native)
        idl = self._ids_array()[np.where(out_rows >= 0, out_rows, 0)].tolist()
        scl = out_scores.tolist()
        have = (out_rows >= 0).sum(axis=1).tolist()
        return [list(zip(idl[b][:have[b]], scl[b][:have[b]])) for b in range(len(have))]

    def _drop_csr(self) -> None:
        c = self.__dict__.pop("_csr_cache", None)
        if c is not None and c.get("handle") is not None:
            _native.bm25_destroy(c["handle"])

    def __del__(self):
        try:
            self._drop_csr()
        except Exception:
            pass

    def get_document(self, doc_id: str) -> Optional[Document]:
        return self.documents.get(doc_id)

    # ---- persistence: append-only log + occasional snapshot in the reference's pickle layout ----------------
    def _log_path(self) -> Path:
        return Path(str(self.persist_path) + ".log.jsonl")

    def _persist_new(self, first_new: int) -> None:
        n = len(self.doc_ids)
        logged = n - self._snapshot_docs
        every = self.snapshot_every
        if (every is not None and logged >= every) or (every is None and logged >= max(self.SNAPSHOT_MIN_DOCS, self._snapshot_docs)):
            self._save()
            return
        if self._log_file is None:
            self.persist_path.parent.mkdir(parents=True, exist_ok=True)
            self._log_file = open(self._log_path(), "a")
        for i in range(first_new, n):
            self._log_file.write(json.dumps(self.documents[self.doc_ids[i]].to_dict()) + "\n")
        self._log_file.flush()

    def _save(self):
        """Snapshot in the reference's layout (:189-201), written atomically; the log restarts empty."""
        if self.persist_path is None:
            return
        self.persist_path.parent.mkdir(parents=True, exist_ok=True)
        data = {
            "documents": {k: v.to_dict() for k, v in self.documents.items()},
            "doc_ids": self.doc_ids,
            "tokenized_corpus": self.tokenized_corpus,
            "k1": self.k1,
            "b": self.b,
        }
        _atomic_write(self.persist_path, pickle.dumps(data))
        self._snapshot_docs = len(self.doc_ids)
        if self._log_file is not None:
            self._log_file.close()
            self._log_file = None
        if self._log_path().exists():
            self._log_path().unlink()

    def save(self) -> None:
        self._save()

    def close(self) -> None:
        if self.persist_path is not None and len(self.doc_ids) != self._snapshot_docs:
            self._save()
        if self._log_file is not None:
            self._log_file.close()
            self._log_file = None

    def _load(self):
        if self.persist_path is None:
            return
        if self.persist_path.exists():
            data = _load_plain_pickle(self.persist_path)   # plain containers only: no class is ever resolved
            self.documents = {k: Document.from_dict(v) for k, v in data["documents"].items()}
            self.doc_ids = data["doc_ids"]
            self.tokenized_corpus = data["tokenized_corpus"]
            self.k1 = data["k1"]
            self.b = data["b"]
        self._post_rows, self._post_tf, self._doc_len, self._total_len = {}, {}, [], 0
        self.__dict__.pop("_post_np", None)              # numpy views of the posting lists (get_scores): rebuilt on demand
        self.__dict__.pop("_doc_len_np", None)
        self.__dict__.pop("_contrib", None)
        self._drop_csr()
        for row, toks in enumerate(self.tokenized_corpus):
            self._index_tokens(row, toks)
        self._snapshot_docs = len(self.doc_ids)
        if self._log_path().exists():                      # documents added since the snapshot
            # Replay stops at the first line that is not a complete record (a torn tail: the process died inside a write).  The
            # torn bytes must not stay in the file: the next add reopens the log in append mode, its first record would be glued
            # onto them, and every document acknowledged from then on would be unreadable at the following load.  So the log
            # is cut back to the end of the last good record (and a record that lost only its newline gets it back).
            good_end, needs_newline = 0, False
            with open(self._log_path(), "rb") as f:
                for raw in f:
                    try:
                        doc = Document.from_dict(json.loads(raw.decode("utf-8")))
                    except (json.JSONDecodeError, UnicodeDecodeError, KeyError, TypeError, AttributeError):
                        break
                    good_end += len(raw)
                    needs_newline = not raw.endswith(b"\n")
                    if doc.id in self.documents:
                        continue
                    self.documents[doc.id] = doc
                    self.doc_ids.append(doc.id)
                    toks = self._tokenize(doc.text)
                    self.tokenized_corpus.append(toks)
                    self._index_tokens(len(self.doc_ids) - 1, toks)
            if good_end != self._log_path().stat().st_size or needs_newline:
                logger.warning(f"BM25 log {self._log_path()}: dropping a torn tail after {good_end} bytes")
                with open(self._log_path(), "r+b") as f:
                    f.truncate(good_end)
                    if needs_newline:
                        f.seek(0, os.SEEK_END)
                        f.write(b"\n")
        logger.info(f"Loaded BM25 index with {len(self.doc_ids)} documents")

    def __len__(self) -> int:
        return len(self.doc_ids)


# =============================================================================================
# dense side (reference :228-373) -- the hot path
# =============================================================================================
def _read_meta(meta_path: Path) -> Optional[Dict[str, int]]:
    """<collection>.meta: 'rq-index 1 / dim D / rows N / dtype f16' (what rq_load parses) + an optional 'docs_bytes B' line."""
    if not meta_path.exists():
        return None
    kv = dict(line.split(None, 1) for line in meta_path.read_text().splitlines() if " " in line)
    if kv.get("rq-index", "").strip() != "1" or kv.get("dtype", "").strip() != "f16":
        raise RuntimeError(f"{meta_path} is not an rq-index v1 meta file")
    out = {"dim": int(kv["dim"]), "rows": int(kv["rows"])}
    if "docs_bytes" in kv:
        out["docs_bytes"] = int(kv["docs_bytes"])
    return out


def repair_persisted_collection(base: Path, docs_path: Path, truncate: bool = True) -> Optional[Dict[str, int]]:
    """Bring the three files of a persisted collection back to the last COMMITTED state.

    An add appends rows to <base>.f16, records to <docs_path>, and only then replaces <base>.meta (the commit point,
    written atomically).  A process killed in between leaves data files LONGER than the meta says; they are cut back
    here, so that the collection reopens with exactly the committed rows (Chroma's PersistentClient, which the reference
    uses at :257, is transactional on add).  Data files SHORTER than the commit are real corruption and raise.
    `truncate=False` only READS the commit (what a process that merely opens the collection does: bytes beyond the commit may
    belong to a writer that sits between appending its data and replacing the meta -- cutting them off would leave that
    writer's next meta describing more bytes than exist); the writer path truncates before it appends.
    Returns {'dim', 'rows', 'docs_bytes'} or None when nothing was ever committed."""
    meta = _read_meta(Path(str(base) + ".meta"))
    if meta is None:
        return None
    f16_path = Path(str(base) + ".f16")
    want = meta["rows"] * meta["dim"] * 2
    have = f16_path.stat().st_size if f16_path.exists() else 0
    if have < want:
        raise RuntimeError(f"persisted index is inconsistent: {f16_path} holds {have} bytes, the commit record needs {want}")
    if have > want and truncate:
        with open(f16_path, "r+b") as f:
            f.truncate(want)
    docs_have = docs_path.stat().st_size if docs_path.exists() else 0
    if "docs_bytes" in meta:
        docs_want = meta["docs_bytes"]
    else:                                   # a meta written by rq_save alone: the commit is the first `rows` lines
        docs_want, seen = 0, 0
        if docs_path.exists():
            with open(docs_path, "rb") as f:
                for line in f:
                    if seen == meta["rows"] or not line.endswith(b"\n"):
                        break
                    docs_want += len(line)
                    seen += 1
        if seen < meta["rows"]:
            raise RuntimeError(f"persisted index is inconsistent: {seen} ids vs {meta['rows']} rows")
    if docs_have < docs_want:
        raise RuntimeError(f"persisted index is inconsistent: {docs_path} holds {docs_have} bytes, the commit record needs {docs_want}")
    if docs_have > docs_want and truncate:
        with open(docs_path, "r+b") as f:
            f.truncate(docs_want)
    meta["docs_bytes"] = docs_want
    return meta

class DenseIndex:
    """Exact cosine top-k over fp16 passage vectors resident in MI355X HBM.

    Signature and defaults follow reference :236-243.  `chroma_host` / `chroma_port` are accepted for
    call compatibility and ignored (there is no service).  Keyword-only extras: `embedder`, `device`,
    `metric`, `backend_options`.  Construction raises when the GPU backend is unavailable, as the reference raises
    ImportError without chromadb (:248-249).
    """

    def __init__(self, collection_name: str = "rag_documents", persist_directory: str = "./data/chroma_db",
                 embedding_model: str = "nomic-embed-text", chroma_host: Optional[str] = None, chroma_port: int = 8000,
                 *, embedder=None, device: int = 0, devices: Optional[Sequence[int]] = None, metric: str = "cosine",
                 load_persisted: bool = True, auto_persist: bool = True, backend_options: Optional[Dict[str, float]] = None):
        self.collection_name = collection_name
        # options of the C library (include/rq.h rq_set_option), applied when the index is created or loaded -- e.g.
        # {"scan8": 0} keeps the fp16 rows as the only copy of the corpus in HBM (no int8 image for the scan)
        self.backend_options = dict(backend_options or {})
        self.persist_directory = persist_directory
        self.embedding_model = embedding_model
        if chroma_host:
            logger.info("chroma_host=%s ignored: the dense index is in-process on GPU %d", chroma_host, device)
        if not _gpu_backend_available():
            raise ImportError("a gfx950 GPU (librq_hip.so backend) is required for DenseIndex")
        self.embedder = embedder if embedder is not None else default_embedder(embedding_model)
        self.device = int(device)
        self.devices = [int(d) for d in devices] if devices else None     # several GPUs in this process: rows are sharded
        self.metric = _native.METRIC_IP if metric in ("ip", "inner_product") else _native.METRIC_COSINE
        self.auto_persist = bool(auto_persist) and bool(persist_directory)   # Chroma's PersistentClient is durable on add (:257)
        self.dim: Optional[int] = None
        self._index: Optional[_native.NativeIndex] = None
        self._ids: List[str] = []
        self._row_of: Dict[str, int] = {}
        self._texts: List[str] = []
        self._metas: List[Dict[str, Any]] = []
        if load_persisted and persist_directory and self._files()[0].exists():
            self._load()
        logger.info(f"Initialized DenseIndex with collection '{collection_name}'")

    # ---- embedding (reference :267-288) -----------------------------------------------------------
    def _zero(self) -> List[float]:
        return [0.0] * (self.dim or getattr(self.embedder, "dim", 768) or 768)

    def _get_embedding(self, text: str) -> List[float]:
        try:
            return [float(v) for v in self.embedder.embed([text])[0]]
        except Exception as e:   # reference :281-284: log, zero vector
            logger.error(f"Embedding failed: {e}")
            return self._zero()

    def _get_embeddings_batch(self, texts: List[str]) -> List[List[float]]:
        return self._embed_matrix(texts).tolist()

    def _embed_matrix(self, texts: Sequence[str]) -> np.ndarray:
        """One embedder call for the whole batch; on failure fall back to per-text calls so a bad text
        degrades to a zero row instead of failing its neighbours."""
        if not texts:
            return np.zeros((0, self.dim or getattr(self.embedder, "dim", 768)), np.float32)
        try:
            v = np.asarray(self.embedder.embed(list(texts)), dtype=np.float32)
            if v.ndim == 2 and v.shape[0] == len(texts):
                return v
            raise ValueError(f"embedder returned shape {v.shape}")
        except Exception as e:
            logger.error(f"Batch embedding failed ({e}); retrying one text at a time")
            return np.asarray([self._get_embedding(t) for t in texts], dtype=np.float32)

    # ---- build (reference :290-336) -----------------------------------------------------------------
    def _ensure_index(self, dim: int) -> None:
        if self._index is None:
            self.dim = int(dim)
            if self.devices and len(self.devices) > 1:
                from .distributed import MultiDeviceIndex
                self._index = MultiDeviceIndex(self.dim, self.devices)
            else:
                self._index = _native.NativeIndex(self.dim, self.devices[0] if self.devices else self.device)
            self._apply_backend_options()
        elif dim != self.dim:
            raise ValueError(f"embedding dimension {dim} does not match the index ({self.dim})")

    def _apply_backend_options(self) -> None:
        for name, value in self.backend_options.items():
            if name == "stripe_rows" and len(self._index):      # the stripe layout of a multi-device index is fixed by its first rows
                continue
            self._index.set_option(name, float(value))

    def add_documents(self, documents: List[Document], batch_size: int = 100) -> int:
        # the reference fetches every stored id per call (:306); a hash table does the same filter
        seen = set()
        new_docs = []
        for d in documents:
            if d.id in self._row_of or d.id in seen:
                continue
            seen.add(d.id)
            new_docs.append(d)
        if not new_docs:
            logger.info("No new documents to add")
            return 0
        total_added = 0
        for i in range(0, len(new_docs), batch_size):
            batch = new_docs[i:i + batch_size]
            vecs = self._embed_matrix([d.text for d in batch])
            self.add_vectors([d.id for d in batch], vecs, [d.text for d in batch],
                             [{"title": d.title or "", **(d.metadata or {})} for d in batch])
            total_added += len(batch)
            logger.info(f"Indexed batch {i // batch_size + 1}, total: {total_added}/{len(new_docs)}")
        return total_added

    def add_vectors(self, ids: Sequence[str], vectors: np.ndarray, texts: Optional[Sequence[str]] = None,
                    metadatas: Optional[Sequence[Dict[str, Any]]] = None) -> int:
        """Append pre-computed embeddings (the `collection.add(embeddings=...)` of reference :326-331)."""
        vectors = np.asarray(vectors, dtype=np.float32)
        if vectors.ndim != 2 or vectors.shape[0] != len(ids):
            raise ValueError("vectors must be [len(ids), dim]")
        self._ensure_index(vectors.shape[1])
        # cosine index: rows are unit-normalised before fp16 rounding (cosine is invariant, fp16 range is safe)
        self._index.add_f32(vectors, normalize=self.metric == _native.METRIC_COSINE)
        first_new = len(self._ids)
        for j, doc_id in enumerate(ids):
            self._row_of[doc_id] = len(self._ids)
            self._ids.append(doc_id)
            self._texts.append(texts[j] if texts is not None else "")
            self._metas.append(dict(metadatas[j]) if metadatas is not None else {})
        if self.auto_persist and hasattr(self._index, "get_rows_f16"):
            self._persist_append(first_new)
        return len(ids)

    def _persist_append(self, first_new: int) -> None:
        """Append the rows [first_new, len) to the rq_save layout (<collection>.f16 / .meta) and their records to
        <collection>.docs.jsonl, so that a later process finds the index (reference: Chroma persists on add).
        Order: data files first, then the meta file replaces the old one atomically -- the commit point.  Whatever an
        interrupted earlier add left behind the commit is cut off before appending (repair_persisted_collection)."""
        docs_path, base = self._files()
        docs_path.parent.mkdir(parents=True, exist_ok=True)
        n = len(self._ids)
        f16_path, meta_path = Path(str(base) + ".f16"), Path(str(base) + ".meta")
        committed = repair_persisted_collection(base, docs_path) if first_new else None
        if committed is None or committed["rows"] != first_new or committed["dim"] != self.dim:
            first_new = 0                         # nothing usable on disk: (re)write from scratch
            docs_bytes = 0
            if meta_path.exists():                # ... and say so first: an old commit over half-rewritten files could never be reopened
                meta_path.unlink()
        else:
            docs_bytes = committed["docs_bytes"]
        rows = self._index.get_rows_f16(first_new, n - first_new)
        with open(f16_path, "ab" if first_new else "wb") as f:
            f.write(np.ascontiguousarray(rows).view(np.uint16).tobytes())
        with open(docs_path, "ab" if first_new else "wb") as f:
            for i in range(first_new, n):
                line = (json.dumps({"id": self._ids[i], "text": self._texts[i], "metadata": self._metas[i]}) + "\n").encode()
                f.write(line)
                docs_bytes += len(line)
        _atomic_write(meta_path, f"rq-index 1\ndim {self.dim}\nrows {n}\ndtype f16\ndocs_bytes {docs_bytes}\n".encode())

    # ---- query (reference :338-370) -------------------------------------------------------------------
    @classmethod
    def from_native(cls, index: "_native.NativeIndex", ids: Sequence[str], texts: Optional[Sequence[str]] = None, *,
                    embedder=None, metric: str = "cosine") -> "DenseIndex":
        """Wrap a shard that already sits in HBM (built through the C ABI / `add_f16_device`) as a DenseIndex: row i of `index`
        answers as `ids[i]`.  Nothing is persisted (extension; the reference has no counterpart)."""
        if len(ids) != len(index):
            raise ValueError(f"{len(ids)} ids for {len(index)} rows")
        self = cls.__new__(cls)
        self.collection_name, self.persist_directory, self.embedding_model = "rag_documents", "", "nomic-embed-text"
        self.backend_options = {}
        self.embedder = embedder if embedder is not None else default_embedder(self.embedding_model)
        self.device, self.devices = index.device, None
        self.metric = _native.METRIC_IP if metric in ("ip", "inner_product") else _native.METRIC_COSINE
        self.auto_persist = False
        self.dim = index.dim
        self._index = index
        self._ids = list(ids)
        self._row_of = {d: i for i, d in enumerate(self._ids)}
        self._texts = list(texts) if texts is not None else [""] * len(self._ids)
        self._metas = [{} for _ in self._ids] if len(self._ids) < 100_000 else [{}] * len(self._ids)
        return self

    def _id_text_arrays(self) -> Tuple[np.ndarray, np.ndarray]:
        """object arrays of the ids / texts (one fancy-index gather per search instead of a Python loop over B * k results)"""
        cache = self.__dict__.get("_idtext_np")
        if cache is None or len(cache[0]) != len(self._ids):
            ids = np.empty(len(self._ids), dtype=object)
            ids[:] = self._ids
            texts = None                       # a collection that stores no passage text (add_vectors without texts): nothing to gather
            if any(self._texts):
                texts = np.empty(len(self._texts), dtype=object)
                texts[:] = self._texts
            cache = self.__dict__["_idtext_np"] = (ids, texts)
        return cache

    def _assemble(self, scores: np.ndarray, rows: np.ndarray) -> List[List[Tuple[str, float, str]]]:
        """(scores [B][k], rows [B][k], -1 padded) -> the reference's result lists (:361-368): (doc_id, float(score), text), best first"""
        ids, texts = self._id_text_arrays()
        ok = rows >= 0
        all_ok = bool(ok.all())
        safe = rows if all_ok else np.where(ok, rows, 0)
        # B * k tuples + B lists of (str, float, str): nothing here can form a cycle, but every 700th container allocation starts a
        # young-generation collection -- 40 % of this function at 256 x 10 (370 -> 220 us on the build container) -- so the collector
        # is paused for these few lines
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            idl, scl = ids[safe].tolist(), scores.astype(np.float64).tolist()
            txl = texts[safe].tolist() if texts is not None else [[""] * rows.shape[1]] * rows.shape[0]
            if all_ok:
                return [list(zip(i, s, t)) for i, s, t in zip(idl, scl, txl)]
            okl = ok.tolist()
            return [[(i, s, t) for i, s, t, v in zip(ib, sb, tb, vb) if v] for ib, sb, tb, vb in zip(idl, scl, txl, okl)]
        finally:
            if gc_was_on:
                gc.enable()

    def search_vectors(self, vectors: np.ndarray, top_k: int = 10) -> List[List[Tuple[str, float, str]]]:
        vectors = np.atleast_2d(np.asarray(vectors, dtype=np.float32))
        if self._index is None or len(self._ids) == 0 or top_k <= 0:
            return [[] for _ in range(vectors.shape[0])]
        if vectors.shape[1] != self.dim:
            raise ValueError(f"query dimension {vectors.shape[1]} does not match the index ({self.dim})")
        k = min(int(top_k), len(self._ids), _native.MAX_K)
        scores, rows = self._index.search(vectors, k, self.metric)
        return self._assemble(scores, rows)

    def search_device_vectors(self, d_vectors, top_k: int = 10) -> List[List[Tuple[str, float, str]]]:
        """Queries that are already in HBM (a CUDA tensor [B][dim] fp32, e.g. `NomicBertEmbedder.embed_device`): searched where they
        are (rq_search_device on torch's current stream), repaired if a certificate failed, ONE device-to-host copy of rows + scores."""
        B = int(d_vectors.shape[0])
        if self._index is None or len(self._ids) == 0 or top_k <= 0 or B == 0:
            return [[] for _ in range(B)]
        return self._assemble(*self._search_device_rows(d_vectors, top_k))

    def _search_device_rows(self, d_vectors, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(scores float32 [B][k], rows int64 [B][k], -1 padded), k = min(top_k, rows of the index)"""
        import torch
        B = int(d_vectors.shape[0])
        if int(d_vectors.shape[1]) != self.dim:
            raise ValueError(f"query dimension {int(d_vectors.shape[1])} does not match the index ({self.dim})")
        k = min(int(top_k), len(self._ids), _native.MAX_K)
        if not hasattr(self._index, "search_device") or len(getattr(self._index, "devices", [0])) > 1:
            return self._index.search(d_vectors.float().cpu().numpy(), k, self.metric)       # multi-device parent: host-buffer calls only
        dev = d_vectors.device
        q = d_vectors.to(torch.float32).contiguous()
        buf = self.__dict__.get("_dev_out")
        if buf is None or buf[0] < B or buf[1] < k or buf[2].device != dev:
            nb, nk = max(B, buf[0] if buf else 0), max(k, buf[1] if buf else 0)
            block = torch.empty((nb * nk * 3 + nb,), device=dev, dtype=torch.int32)          # [rows int64 | scores fp32 | status int32]
            buf = self.__dict__["_dev_out"] = (nb, nk, block, torch.empty((nb * nk * 3 + nb,), dtype=torch.int32).pin_memory())
        block, pinned = buf[2], buf[3]
        rows_t = block[: 2 * B * k].view(torch.int64).view(B, k)
        scores_t = block[2 * B * k: 3 * B * k].view(torch.float32).view(B, k)
        status_t = block[3 * B * k: 3 * B * k + B]
        stream = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            self._index.search_device(q, B, k, self.metric, scores_t, rows_t, None, status_t, stream.cuda_stream)
            self._index.search_flush_device(stream.cuda_stream)
            n = 3 * B * k + B
            pinned[:n].copy_(block[:n], non_blocking=True)
            stream.synchronize()
            if bool(pinned[3 * B * k: n].any()):        # rare: a certificate failed -> exact repair on the device, fetch again
                self._index.search_fixup_device(q, B, k, self.metric, scores_t, rows_t, None, status_t, stream.cuda_stream)
                pinned[:n].copy_(block[:n], non_blocking=True)
                stream.synchronize()
        host = pinned.numpy()
        rows = host[: 2 * B * k].view(np.int64).reshape(B, k)
        scores = host[2 * B * k: 3 * B * k].view(np.float32).reshape(B, k)
        return scores, rows

    def search_rows_batch(self, queries: Sequence[str], top_k: int = 10) -> Tuple[np.ndarray, np.ndarray]:
        """`search_batch` without the (doc_id, score, text) tuples: (scores float32 [B][k], rows int64 [B][k], -1 padded; row r is
        `self._ids[r]`), k = min(top_k, len(self)).  What HybridRetriever's batched fusion consumes (extension)."""
        B = len(queries)
        if self._index is None or len(self._ids) == 0 or top_k <= 0 or B == 0:
            return np.zeros((B, 0), np.float32), np.zeros((B, 0), np.int64)
        if hasattr(self.embedder, "embed_device"):
            try:
                d_q = self.embedder.embed_device(list(queries))
            except Exception as e:
                logger.error(f"Device embedding failed ({e}); falling back to the host path")
                d_q = None
            if d_q is not None:
                return self._search_device_rows(d_q, top_k)
        k = min(int(top_k), len(self._ids), _native.MAX_K)
        return self._index.search(self._embed_matrix(list(queries)), k, self.metric)

    def search_batch(self, queries: Sequence[str], top_k: int = 10) -> List[List[Tuple[str, float, str]]]:
        if not queries:
            return []
        # an embedder that can leave its output in HBM (embedders.NomicBertEmbedder.embed_device) feeds the search directly: no
        # device -> host -> numpy -> pinned -> device round trip of the query matrix
        if hasattr(self.embedder, "embed_device") and self._index is not None and len(self._ids):
            try:
                d_q = self.embedder.embed_device(list(queries))
            except Exception as e:     # reference :281-284 semantics live in _embed_matrix (per-text retry, zero vector)
                logger.error(f"Device embedding failed ({e}); falling back to the host path")
                d_q = None
            if d_q is not None:
                return self.search_device_vectors(d_q, top_k)
        return self.search_vectors(self._embed_matrix(list(queries)), top_k)

    def search(self, query: str, top_k: int = 10) -> List[Tuple[str, float, str]]:
        """List of (doc_id, score, text), best first; score = cosine similarity (= 1 - Chroma distance)."""
        return self.search_batch([query], top_k)[0]

    def __len__(self) -> int:
        return len(self._ids)

    # ---- persistence (the Chroma persist directory of the reference) ------------------------------------
    def _files(self) -> Tuple[Path, Path]:
        base = Path(self.persist_directory) / self.collection_name
        return Path(str(base) + ".docs.jsonl"), base

    def save(self) -> None:
        """rows -> <dir>/<collection>.f16/.meta (rq_save), ids/texts/metadata -> <collection>.docs.jsonl.  Everything is written
        under temporary names first; the data files are then moved into place and the meta replaces the old one LAST (the commit
        point): a process killed anywhere in between leaves either the old commit over files that still contain it as a prefix, or
        the new one."""
        docs_path, base = self._files()
        docs_path.parent.mkdir(parents=True, exist_ok=True)
        if self._index is None:
            return
        docs_tmp, base_tmp = Path(str(docs_path) + ".tmp"), Path(str(base) + ".saving")
        docs_bytes = 0
        with open(docs_tmp, "wb") as f:
            for i, doc_id in enumerate(self._ids):
                line = (json.dumps({"id": doc_id, "text": self._texts[i], "metadata": self._metas[i]}) + "\n").encode()
                f.write(line)
                docs_bytes += len(line)
            f.flush()
            os.fsync(f.fileno())
        self._index.save(str(base_tmp))              # rows (+ a meta of its own, discarded) through rq_save
        os.replace(str(base_tmp) + ".f16", str(base) + ".f16")
        os.replace(docs_tmp, docs_path)
        Path(str(base_tmp) + ".meta").unlink(missing_ok=True)
        _atomic_write(Path(str(base) + ".meta"), f"rq-index 1\ndim {self.dim}\nrows {len(self._ids)}\ndtype f16\ndocs_bytes {docs_bytes}\n".encode())

    def _load(self) -> None:
        docs_path, base = self._files()
        # read-only: bytes beyond the commit are ignored, not cut off (another process may be in the middle of an add; this
        # process truncates only when it appends itself, _persist_append)
        committed = repair_persisted_collection(base, docs_path, truncate=False)
        if committed is None:
            return
        self._index = _native.NativeIndex.load(str(base), self.device, devices=self.devices)     # (rq_load reads `rows` rows, no more)
        self._apply_backend_options()
        self.dim = self._index.dim
        with open(docs_path, "rb") as f:
            for line in f.read(committed["docs_bytes"]).splitlines():
                rec = json.loads(line)
                self._row_of[rec["id"]] = len(self._ids)
                self._ids.append(rec["id"])
                self._texts.append(rec.get("text", ""))
                self._metas.append(rec.get("metadata", {}))
        if len(self._ids) != len(self._index):
            raise RuntimeError(f"persisted index is inconsistent: {len(self._ids)} ids vs {len(self._index)} rows")
        logger.info(f"Loaded DenseIndex with {len(self._ids)} documents")


# =============================================================================================
# hybrid fusion (reference :376-560) -- pure host logic, kept semantically identical
# =============================================================================================
class HybridRetriever:
    """Unified hybrid retrieval combining BM25 and dense retrieval (reference :376-560).

    Extra keyword-only arguments: `dense_index` (inject a ready index), `embedder`, `device`.
    """

    def __init__(self, bm25_persist_path: str = "./data/bm25_index.pkl", chroma_persist_path: str = "./data/chroma_db",
                 chroma_host: Optional[str] = None, embedding_model: str = "nomic-embed-text",
                 *, dense_index=None, embedder=None, device: int = 0):
        self.bm25_index = BM25Index(persist_path=bm25_persist_path)
        if dense_index is not None:
            self.dense_index = dense_index
        elif _gpu_backend_available():
            self.dense_index = DenseIndex(persist_directory=chroma_persist_path,
                                          chroma_host=chroma_host or os.environ.get("CHROMA_HOST"),
                                          embedding_model=embedding_model, embedder=embedder, device=device)
        else:
            self.dense_index = None
            logger.warning("Dense retrieval disabled")   # reference :418-420
        # The reference starts with an empty document store (:422-423) although both indexes persist, so a fresh
        # process over a persisted index answers [].  The BM25 pickle holds every Document:
        # restore the store from it (same signatures, DESIGN.md "deviations").
        self.documents: Dict[str, Document] = dict(self.bm25_index.documents)

    def add_documents(self, documents: List[Document], batch_size: int = 100) -> Dict[str, int]:
        for doc in documents:
            self.documents[doc.id] = doc
        stats: Dict[str, int] = {}
        # The reference tests `if self.bm25_index:` / `if self.dense_index:` (:442, :445); both classes
        # define __len__, so an EMPTY index is falsy and never receives its first documents.  That is a
        # defect, not a contract: here the test is `is not None` (DESIGN.md "deviations").
        if self.bm25_index is not None:
            stats["bm25_added"] = self.bm25_index.add_documents(documents)
        if self.dense_index is not None:
            stats["dense_added"] = self.dense_index.add_documents(documents, batch_size)
        stats["total_documents"] = len(self.documents)
        return stats

    def bm25_search(self, query: str, top_k: int = 20) -> List[Tuple[str, float]]:
        if self.bm25_index is None:
            return []
        return self.bm25_index.search(query, top_k)

    def dense_search(self, query: str, top_k: int = 20) -> List[Tuple[str, float]]:
        if self.dense_index is None:
            return []
        return [(doc_id, score) for doc_id, score, _ in self.dense_index.search(query, top_k)]

    def dense_search_batch(self, queries: Sequence[str], top_k: int = 20) -> List[List[Tuple[str, float]]]:
        if self.dense_index is None:
            return [[] for _ in queries]
        if hasattr(self.dense_index, "search_batch"):
            res = self.dense_index.search_batch(list(queries), top_k)
        else:
            res = [self.dense_index.search(q, top_k) for q in queries]
        return [[(d, s) for d, s, _ in r] for r in res]

    def _fuse_columns(self, bm25_list: List[Tuple[str, float]], dense_list: List[Tuple[str, float]], top_k: int):
        """Reference :485-523 on parallel lists: union of both pools, ids unknown to `self.documents` dropped, missing score 0.0,
        hybrid = (bm25/max_bm25 + dense/max_dense)/2 with `max(...) or 1` over ALL candidates, stable sort desc, first top_k.
        The reference walks a Python set (arbitrary order); here the union is walked in first-seen order (BM25 pool, then dense
        pool), one admissible instance of that order.  Returns (ids, bm25 scores, dense scores, hybrid scores) of the top_k --
        the RetrievalResult objects (2 us each, 200 candidates per question) are only built for what is returned."""
        bm25_results = dict(bm25_list)
        dense_results = dict(dense_list)
        documents = self.documents
        ids = [d for d in dict.fromkeys(list(bm25_results) + list(dense_results)) if d in documents]
        if not ids:
            return [], [], [], []
        b = [bm25_results.get(d, 0.0) for d in ids]
        de = [dense_results.get(d, 0.0) for d in ids]
        max_bm25 = max(b) or 1
        max_dense = max(de) or 1
        h = [(x / max_bm25 + y / max_dense) / 2 for x, y in zip(b, de)]
        order = sorted(range(len(ids)), key=lambda i: h[i] or 0, reverse=True)[:top_k]      # stable, like list.sort(reverse=True)
        return [ids[i] for i in order], [b[i] for i in order], [de[i] for i in order], [h[i] for i in order]

    def _fuse(self, bm25_list: List[Tuple[str, float]], dense_list: List[Tuple[str, float]], top_k: int) -> List[RetrievalResult]:
        ids, b, de, h = self._fuse_columns(bm25_list, dense_list, top_k)
        out = []
        for doc_id, bs, ds, hs in zip(ids, b, de, h):
            doc = self.documents[doc_id]
            out.append(RetrievalResult(doc_id=doc_id, text=doc.text, bm25_score=bs, dense_score=ds, hybrid_score=hs, title=doc.title, metadata=doc.metadata))
        return out

    def hybrid_search(self, query: str, top_k: int = 10, retrieval_pool_size: int = 50) -> List[RetrievalResult]:
        return self._fuse(self.bm25_search(query, retrieval_pool_size), self.dense_search(query, retrieval_pool_size), top_k)

    def hybrid_search_batch(self, queries: Sequence[str], top_k: int = 10, retrieval_pool_size: int = 50) -> List[List[RetrievalResult]]:
        """All dense pools from one GPU batch, all BM25 pools from one pass over the posting lists on the host cores (librq_bm25.so);
        fusion per query."""
        fused = self._fuse_batch_rows(queries, top_k, retrieval_pool_size)
        if fused is not None:
            ids, b, de, hy, count = fused
            idl, bl, dl, hl = ids.tolist(), b.tolist(), de.tolist(), hy.tolist()
            out = []
            for q in range(len(queries)):
                res = []
                for i in range(int(count[q])):
                    doc = self.documents[idl[q][i]]
                    res.append(RetrievalResult(doc_id=doc.id, text=doc.text, bm25_score=bl[q][i], dense_score=dl[q][i], hybrid_score=hl[q][i],
                                               title=doc.title, metadata=doc.metadata))
                out.append(res)
            return out
        dense, sparse = self._pools_batch(queries, retrieval_pool_size)
        return [self._fuse(sparse[i], dense[i], top_k) for i in range(len(queries))]

    def _pools_batch(self, queries: Sequence[str], retrieval_pool_size: int):
        dense = self.dense_search_batch(queries, retrieval_pool_size)
        if self.bm25_index is None:
            sparse = [[] for _ in queries]
        elif hasattr(self.bm25_index, "search_batch"):
            sparse = self.bm25_index.search_batch(list(queries), retrieval_pool_size)
        else:
            sparse = [self.bm25_search(q, retrieval_pool_size) for q in queries]
        return dense, sparse

    @staticmethod
    def _router_arrays(results: List[RetrievalResult], num_passages: int):
        bm25_scores = [r.bm25_score for r in results]
        dense_scores = [r.dense_score for r in results]
        doc_ids = [r.doc_id for r in results]
        texts = [r.text for r in results]
        pad = num_passages - len(results)
        if pad > 0:
            bm25_scores += [0.0] * pad
            dense_scores += [0.0] * pad
            doc_ids += [""] * pad
            texts += [""] * pad
        return bm25_scores, dense_scores, doc_ids, texts

    def get_scores_for_router(self, query: str, num_passages: int = 20, *, retrieval_pool_size: int = 50
                              ) -> Tuple[List[float], List[float], List[str], List[str]]:
        """Reference :525-557 (its pools are always 50, :537); `retrieval_pool_size` is a keyword-only extension for
        BASELINE.json configs[4] (top-100 pools)."""
        return self._router_arrays(self.hybrid_search(query, top_k=num_passages, retrieval_pool_size=retrieval_pool_size), num_passages)

    def get_scores_for_router_batch(self, queries: Sequence[str], num_passages: int = 20, *, retrieval_pool_size: int = 50):
        """`[get_scores_for_router(q, ...) for q in queries]` with the pools of the whole batch computed at once.  When both sides are
        this module's own indexes the fusion itself runs on the whole batch in row space (`_fuse_batch_rows`); otherwise per query on the
        fused columns (no RetrievalResult objects in between)."""
        fused = self._fuse_batch_rows(queries, num_passages, retrieval_pool_size)
        if fused is not None:
            ids, b, de, _, count = fused
            ids[np.arange(num_passages)[None, :] >= count[:, None]] = ""
            idl = ids.tolist()
            documents = self.documents                      # (texts are read from the store at call time, as the per-query path does)
            texts = [[documents[d].text if d else "" for d in row] for row in idl]
            return list(zip(b.tolist(), de.tolist(), idl, texts))
        dense, sparse = self._pools_batch(queries, retrieval_pool_size)
        documents = self.documents
        out = []
        for i in range(len(queries)):
            ids, b, de, _ = self._fuse_columns(sparse[i], dense[i], num_passages)
            texts = [documents[d].text for d in ids]
            pad = num_passages - len(ids)
            if pad > 0:
                b = b + [0.0] * pad; de = de + [0.0] * pad; ids = ids + [""] * pad; texts = texts + [""] * pad
            out.append((b, de, ids, texts))
        return out

    # ---- batched fusion in row space (extension; configs[4]: 500 questions x two pools of 100) ------------------------------------
    def _key_space(self):
        """One integer key per document either index can return: BM25 row r -> r, dense row j -> the BM25 row of the same id if BM25
        holds it, else n_bm25 + j.  Per key: the id and whether `self.documents` knows it at all (reference :491-493 skips ids it does
        not).  Rebuilt when any of the three stores has grown."""
        bm, dn = self.bm25_index, self.dense_index
        stamp = (len(bm.doc_ids), len(dn._ids), len(self.documents))
        c = self.__dict__.get("_keys_cache")
        if c is not None and c["stamp"] == stamp:
            return c
        nb = len(bm.doc_ids)
        row_of = {d: i for i, d in enumerate(bm.doc_ids)}
        dense_key = np.fromiter((row_of.get(d, nb + j) for j, d in enumerate(dn._ids)), np.int64, len(dn._ids))
        all_ids = list(bm.doc_ids) + list(dn._ids)
        ids = np.empty(len(all_ids), dtype=object)
        ids[:] = all_ids
        docs = self.documents
        known = np.fromiter((d in docs for d in all_ids), np.bool_, len(all_ids))
        c = self.__dict__["_keys_cache"] = {"stamp": stamp, "nb": nb, "dense_key": dense_key, "ids": ids, "known": known}
        return c

    def _fuse_batch_rows(self, queries: Sequence[str], top_k: int, retrieval_pool_size: int):
        """`_fuse_columns` for every query of the batch at once, on integer keys: same candidates (union of both pools in first-seen
        order -- BM25 pool, then dense pool -- ids unknown to `self.documents` dropped), same float64 arithmetic (`max(...) or 1` over all
        candidates, hybrid = (b/max_b + d/max_d)/2), same stable descending sort, first top_k.  Returns (ids [B][top_k] object,
        bm25 scores, dense scores, hybrid scores, count [B]) -- entries beyond count[b] are padding -- or None when one of the two sides
        is not this module's own index class (then the per-query path runs)."""
        bm, dn = self.bm25_index, self.dense_index
        if not (isinstance(bm, BM25Index) and isinstance(dn, DenseIndex)) or len(queries) == 0 or top_k <= 0:
            return None
        B = len(queries)
        ks = self._key_space()
        sp_rows, sp_scores = bm.search_batch_rows(list(queries), retrieval_pool_size)
        de_scores, de_rows = dn.search_rows_batch(list(queries), retrieval_pool_size)
        K1, K2 = sp_rows.shape[1], de_rows.shape[1]
        sp_keys = sp_rows.astype(np.int64)
        de_keys = np.where(de_rows >= 0, ks["dense_key"][np.where(de_rows >= 0, de_rows, 0)], -1) if K2 else np.zeros((B, 0), np.int64)
        keys = np.concatenate([sp_keys, de_keys], axis=1)
        bsc = np.concatenate([sp_scores, np.zeros((B, K2))], axis=1)
        dsc = np.concatenate([np.zeros((B, K1)), de_scores.astype(np.float64)], axis=1)
        valid = keys >= 0
        valid &= ks["known"][np.where(valid, keys, 0)]
        # a document in both pools: its dense score moves to the BM25 entry (first seen), the dense entry goes
        order = np.argsort(keys, axis=1, kind="stable")
        sk = np.take_along_axis(keys, order, 1)
        r, j = np.nonzero((sk[:, 1:] == sk[:, :-1]) & (sk[:, 1:] >= 0))
        first, second = order[r, j], order[r, j + 1]
        dsc[r, first] = dsc[r, second]
        valid[r, second] = False
        neg_inf = -np.inf
        max_b = np.where(valid, bsc, neg_inf).max(axis=1, initial=neg_inf)
        max_d = np.where(valid, dsc, neg_inf).max(axis=1, initial=neg_inf)
        max_b = np.where((max_b == 0) | ~np.isfinite(max_b), 1.0, max_b)          # `max(...) or 1` (a row without candidates: anything)
        max_d = np.where((max_d == 0) | ~np.isfinite(max_d), 1.0, max_d)
        h = (bsc / max_b[:, None] + dsc / max_d[:, None]) / 2
        sort_key = np.where(valid, -np.where(h == 0, 0.0, h), np.inf)            # `key = hybrid or 0`, descending, stable; dropped entries last
        sel = np.argsort(sort_key, axis=1, kind="stable")[:, :top_k]
        if sel.shape[1] < top_k:                                                  # pools smaller than top_k: pad the columns
            sel = np.concatenate([sel, np.zeros((B, top_k - sel.shape[1]), np.int64)], axis=1)
        count = np.minimum(valid.sum(axis=1), top_k)
        live = np.arange(top_k)[None, :] < count[:, None]
        out_keys = np.where(live, np.take_along_axis(keys, sel, 1), 0)
        b = np.where(live, np.take_along_axis(bsc, sel, 1), 0.0)
        de = np.where(live, np.take_along_axis(dsc, sel, 1), 0.0)
        hy = np.where(live, np.take_along_axis(h, sel, 1), 0.0)
        return ks["ids"][out_keys], b, de, hy, count

    def close(self) -> None:
        """Write the BM25 snapshot if documents were added since the last one (extension; the reference has no close)."""
        if self.bm25_index is not None and hasattr(self.bm25_index, "close"):
            self.bm25_index.close()

    def __len__(self) -> int:
        return len(self.documents)


# =============================================================================================
# streaming indexer + checkpoint (reference :563-686)
# =============================================================================================
class StreamingIndex:
    """Resumable JSONL -> retriever feeder with the reference's checkpoint schema
    {'last_offset', 'total_indexed', 'files_completed'} (:593-604)."""

    def __init__(self, retriever: HybridRetriever, checkpoint_path: str = "./data/index_checkpoint.json", batch_size: int = 100):
        self.retriever = retriever
        self.checkpoint_path = Path(checkpoint_path)
        self.batch_size = batch_size
        self.progress = self._load_checkpoint()

    def _load_checkpoint(self) -> Dict[str, Any]:
        if self.checkpoint_path.exists():
            with open(self.checkpoint_path) as f:
                return json.load(f)
        return {"last_offset": 0, "total_indexed": 0, "files_completed": []}

    def _save_checkpoint(self):
        self.checkpoint_path.parent.mkdir(parents=True, exist_ok=True)
        with open(self.checkpoint_path, "w") as f:
            json.dump(self.progress, f)

    def _commit(self, batch: List[Document], offset: int) -> int:
        self.retriever.add_documents(batch)
        self.progress["last_offset"] = offset
        self.progress["total_indexed"] += len(batch)
        self._save_checkpoint()
        return len(batch)

    def stream_from_jsonl(self, jsonl_path: str, resume: bool = True) -> Iterator[int]:
        path = Path(jsonl_path)
        if not path.exists():
            raise FileNotFoundError(f"Corpus file not found: {jsonl_path}")
        start_offset = self.progress["last_offset"] if resume else 0
        with open(path) as f:
            for _ in range(start_offset):
                f.readline()
            batch: List[Document] = []
            offset = start_offset
            for line in f:
                try:
                    data = json.loads(line.strip())
                    batch.append(Document(id=data["id"], text=data["text"], title=data.get("title"), metadata=data.get("metadata")))
                except (json.JSONDecodeError, KeyError) as e:
                    logger.warning(f"Skipping invalid line at offset {offset}: {e}")
                offset += 1
                if len(batch) >= self.batch_size:
                    n = self._commit(batch, offset)
                    logger.info(f"Indexed batch: {n} docs, total: {self.progress['total_indexed']}")
                    yield n
                    batch = []
            if batch:
                yield self._commit(batch, offset)
        if jsonl_path not in self.progress["files_completed"]:
            self.progress["files_completed"].append(jsonl_path)
            self._save_checkpoint()
        logger.info(f"Completed indexing {jsonl_path}")

    def get_progress(self) -> Dict[str, Any]:
        return {**self.progress, "retriever_size": len(self.retriever)}
