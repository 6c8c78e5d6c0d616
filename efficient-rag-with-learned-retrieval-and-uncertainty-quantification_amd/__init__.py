"""MI355X-native dense retrieval behind the reference's streaming_index API.

Mirrors `rag_uq/__init__.py:11-24` of the reference for the names on the dense hot path
(HybridRetriever, StreamingIndex and the records they use); everything else of the reference
(router, confidence, evaluation) is a consumer of this API and stays where it is.
"""
__version__ = "0.1.0"
