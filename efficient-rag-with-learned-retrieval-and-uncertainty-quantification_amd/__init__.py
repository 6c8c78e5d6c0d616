"""MI355X-native dense retrieval behind the reference's streaming_index API.

Re-exports, lazily, the names of the reference's `rag_uq/__init__.py:11-24` that live on the dense hot path
(`HybridRetriever`, `StreamingIndex`) plus the records and index classes of `rag_uq/streaming_index.py`; everything else
the reference exports there (router, confidence, evaluation) is a consumer of this API and stays where it is.
Lazy: `import rag_uq_amd` must not load librq_hip.so or torch.
"""
__version__ = "0.2.0"

__all__ = ["HybridRetriever", "StreamingIndex", "DenseIndex", "BM25Index", "Document", "RetrievalResult"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return getattr(importlib.import_module(__name__ + ".streaming_index"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
