"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by rucene_amd/; tests/test_layout.py enforces that).

CPU restatement of the reference's algorithms, each function citing the /root/reference file:line it follows:

    store.hpp          DataInput/DataOutput grammar, codec headers / footers, CRC32
    packed.hpp         BP128 (SIMD128Packer), legacy Packed / PackedSingleBlock, ForUtil sizing     [pinned by reference KATs]
    postings.hpp       .doc: ForUtil block framing, skip writer/reader, postings writer, BlockDocIterator
    search.hpp         SmallFloat, BM25, TermScorer, Conjunction / DisjunctionSum / ReqNot / ReqOpt scorers, BulkScorer,
                       TopDocsCollector (Rust BinaryHeap emulation + canonical order), IndexSearcher        [scorers pinned by KATs]
    norms.hpp          Lucene53 norms (.nvm/.nvd), Lucene50 live docs (.liv)
    fst.hpp            FST<ByteSequenceOutput> builder + reader (the .tip terms index)                        [pinned by the reference's test_fst]
    blocktree.hpp      BlockTreeTermsWriter, BlockTreeTermsReader::seek_exact (.tim/.tip)
    field_infos.hpp    Lucene60FieldInfosFormat (.fnm)
    segment_infos.hpp  Lucene62SegmentInfoFormat (.si), SegmentInfos commit point (segments_N)
    positions.hpp      .pos, skip entries with position pointers, BlockPostingIterator
    phrase.hpp         PhraseWeight + ExactPhraseScorer (slop 0)
    oracle_capi.cpp    extern "C" surface; binding.py is its ctypes wrapper

Everything not marked pinned is "parity unpinned": the reference holds no test for it and cannot be built here (no Rust
toolchain), so the source text is the only authority — see DESIGN.md §6.
"""
