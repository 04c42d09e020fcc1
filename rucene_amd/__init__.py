"""rucene_amd — MI355X-native query evaluation for Rucene (Lucene50 postings -> BM25 -> top-k on gfx950).

The product is the C-ABI shared library `librucene_gpu.so` (include/rucene_gpu.h); this package is the thin
Python host layer over it (ctypes) used by tests and bench.py, plus the synthetic index generator binding.
There is NO CPU fallback: importing works anywhere, but every compute call needs the HIP library and a GPU
and fails loudly otherwise.
"""
import os as _os
# multi-process GPU work on this pool's hosts needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); a caller's own
# setting wins. Must be in the environment before the HSA runtime starts, i.e. before the library's first HIP call.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from ._lib import (OP_AND, OP_OR, OP_TERM, Context, RgpuError, Segment, bm25_compute_weight, bm25_encode_norm,  # noqa: F401
                   norms_from_lucene53, live_docs_from_lucene50, field_infos_from_lucene60, segment_info_from_lucene62, commit_from_segments_file, compound_files_from_lucene50, TermDictionary, lib, lib_path, QUERY_DTYPE, QUERY_TERM_DTYPE, TERM_STATE_DTYPE, HIT_DTYPE)
from .searcher import (BM25Similarity, BooleanQuery, CollectionStatistics, GpuIndexSearcher, LeafReader,  # noqa: F401
                       PhraseQuery, TermQuery, TopDocs, TopDocsCollector, open_directory)
