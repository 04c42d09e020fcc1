// Host-only use of the C++ mirror: rucene::IndexDirectory::open over a Rucene-style directory (no GPU involved until a
// GpuIndexSearcher is built on the leaves). Prints, per leaf: max_doc doc_base doc_count sum_ttf sum_df field_number live? and
// the doc_freq / doc_start_fp of the terms given on the command line ("-" for an absent term).
//   argv: directory field term...
#include <cstdio>
#include <string>

#include "../../rucene_amd/csrc/host/gpu_index_searcher.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  try {
    auto dir = rucene::IndexDirectory::open(argv[1], argv[2]);
    for (const rucene::LeafReader& leaf : dir->leaves()) {
      std::printf("leaf %d %d %lld %lld %lld %d %d", leaf.max_doc, leaf.doc_base, (long long)leaf.doc_count, (long long)leaf.sum_total_term_freq,
                  (long long)leaf.sum_doc_freq, leaf.field_number, leaf.live_docs ? 1 : 0);
      for (int i = 3; i < argc; ++i) {
        rgpu_term_state st;
        if (leaf.term_state(rucene::TermQuery(std::string(argv[i])), &st)) std::printf(" %d@%lld", st.doc_freq, (long long)st.doc_start_fp);
        else std::printf(" -");
      }
      std::printf("\n");
    }
  } catch (const rucene::Error& e) {
    std::printf("error %d %s\n", e.kind, e.what());
    return 1;
  }
  return 0;
}
