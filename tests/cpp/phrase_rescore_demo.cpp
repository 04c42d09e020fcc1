// The C++ host mirror's PhraseQuery and QueryRescorer paths (rucene_amd/csrc/host/gpu_index_searcher.hpp) over a positions
// field handed over as raw files: <dir>/{doc,pos,norms,terms,tpos}.bin (terms = rgpu_term_state[], tpos =
// rgpu_term_positions[]) and "<max_doc> <doc_count> <sum_total_term_freq>" on the command line. Prints
//   phrase <i> <total_hits> <doc>:<score-bits> ...      for a fixed list of phrases, k = 10
//   rescore <i> <doc>:<score-bits> ...                  the TERM top-10 of term i re-ranked by an OR / AND / TERM query
// tests/test_gpu_parity.py::test_cpp_host_mirror_phrases_and_rescoring compares the lines with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../rucene_amd/csrc/host/gpu_index_searcher.hpp"

static std::vector<uint8_t> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
static void print_docs(const rucene::TopDocs& top) {
  for (const rucene::ScoreDoc& d : top.score_docs()) {
    uint32_t bits;
    std::memcpy(&bits, &d.score, 4);
    std::printf(" %d:%08x", d.doc, bits);
  }
  std::printf("\n");
}

int main(int argc, char** argv) {
  using namespace rucene;
  if (argc != 5) return 1;
  try {
    const std::string dir = argv[1];
    const std::vector<uint8_t> doc = slurp(dir + "/doc.bin"), pos = slurp(dir + "/pos.bin"), norms = slurp(dir + "/norms.bin"),
                               terms = slurp(dir + "/terms.bin"), tpos = slurp(dir + "/tpos.bin");
    LeafReader leaf;
    leaf.index_options = 3;
    leaf.doc_bytes = doc.data();
    leaf.doc_len = doc.size();
    leaf.pos_bytes = pos.data();
    leaf.pos_len = pos.size();
    leaf.norms = norms.data();
    leaf.max_doc = std::atoi(argv[2]);
    leaf.doc_count = std::atoll(argv[3]);
    leaf.sum_total_term_freq = std::atoll(argv[4]);
    leaf.terms = reinterpret_cast<const rgpu_term_state*>(terms.data());
    leaf.n_terms = static_cast<int64_t>(terms.size() / sizeof(rgpu_term_state));
    leaf.term_positions = reinterpret_cast<const rgpu_term_positions*>(tpos.data());
    GpuIndexSearcher searcher({leaf});

    std::vector<PhraseQuery> phrases;
    phrases.emplace_back(std::vector<TermQuery>{TermQuery(0), TermQuery(1)});
    phrases.emplace_back(std::vector<TermQuery>{TermQuery(3), TermQuery(3)});
    phrases.emplace_back(std::vector<TermQuery>{TermQuery(4), TermQuery(5), TermQuery(6)});
    phrases.emplace_back(std::vector<TermQuery>{TermQuery(2), TermQuery(7)}, std::vector<int32_t>{0, 2});  // a gap
    phrases.emplace_back(std::vector<TermQuery>{TermQuery(1), TermQuery(leaf.n_terms - 1)});               // an absent term
    std::vector<const PhraseQuery*> pq;
    for (const PhraseQuery& p : phrases) pq.push_back(&p);
    const std::vector<TopDocs> got = searcher.search_phrases(pq, 10);
    for (size_t i = 0; i < got.size(); ++i) {
      std::printf("phrase %zu %lld", i, (long long)got[i].total_hits());
      print_docs(got[i]);
    }

    // first pass: TERM queries 0..2; second pass: an OR, an AND and a TERM query, three RescoreModes
    std::vector<std::unique_ptr<Query>> first, second;
    for (int t = 0; t < 3; ++t) first.emplace_back(new TermQuery(t));
    second.push_back(BooleanQuery::build({}, {TermQuery(4), TermQuery(5)}));
    second.push_back(BooleanQuery::build({TermQuery(0), TermQuery(2)}, {}));
    second.emplace_back(new TermQuery(9));
    std::vector<const Query*> fq;
    for (auto& q : first) fq.push_back(q.get());
    const std::vector<TopDocs> pass1 = searcher.search_many(fq, 10);
    std::vector<RescoreRequest> reqs(3);
    reqs[0].query = second[0].get(); reqs[0].mode = RGPU_RESCORE_TOTAL; reqs[0].rescore_weight = 2.0f;
    reqs[1].query = second[1].get(); reqs[1].mode = RGPU_RESCORE_MAX; reqs[1].window_size = 5;
    reqs[2].query = second[2].get(); reqs[2].mode = RGPU_RESCORE_MULTIPLY; reqs[2].query_weight = 0.5f;
    const std::vector<TopDocs> pass2 = searcher.rescore(pass1, reqs, 10);
    for (size_t i = 0; i < pass2.size(); ++i) {
      std::printf("rescore %zu", i);
      print_docs(pass2[i]);
    }
  } catch (const rucene::Error& e) {
    std::fprintf(stderr, "rucene::Error kind=%d: %s\n", e.kind, e.what());
    return 2;
  }
  return 0;
}
