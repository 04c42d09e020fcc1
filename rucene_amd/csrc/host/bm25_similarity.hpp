// Host-side mirror of Rucene's BM25Similarity / SimWeight for the GPU path (C++: the reference is
// compiled Rust and no Rust toolchain exists in this image — see DESIGN.md §2).
// The device never derives idf or the norm cache: the host computes them exactly like the reference and
// ships {weight, cache[256]} inside rgpu_query_term / rgpu_sim_table_upload.
//
// Mirrors (paths relative to /root/reference/src/core):
//   util/small_float.rs:16-36                     SmallFloat::{float_to_byte315, byte315_to_float}
//   search/similarity/bm25_similarity.rs:33-43    NORM_TABLE
//   search/similarity/bm25_similarity.rs:45-63    BM25Similarity::new / default (k1 = 1.2, b = 0.75)
//   search/similarity/bm25_similarity.rs:72-114   avg_field_length, encode_norm_value, idf
//   search/similarity/bm25_similarity.rs:151-177  compute_weight
//   search/statistics.rs                          CollectionStatistics, TermStatistics
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rucene {

struct SmallFloat {
  static uint8_t float_to_byte315(float f) {
    int32_t bits;
    std::memcpy(&bits, &f, 4);
    const int32_t zero_point = (63 - 15) << 3;
    int32_t small = bits >> 21;
    if (small <= zero_point) return bits <= 0 ? 0 : 1;
    if (small >= zero_point + 0x100) return 255;
    return static_cast<uint8_t>(small - zero_point);
  }
  static float byte315_to_float(uint8_t b) {
    if (b == 0) return 0.0f;
    uint32_t bits = (static_cast<uint32_t>(b) << 21) + (static_cast<uint32_t>(63 - 15) << 24);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
  }
};

struct CollectionStatistics {
  int32_t doc_base = 0;
  int64_t max_doc = 0;
  int64_t doc_count = -1;
  int64_t sum_total_term_freq = -1;
  int64_t sum_doc_freq = -1;
};

struct TermStatistics {
  int64_t doc_freq = 0;
  int64_t total_term_freq = -1;
};

// What SimWeight carries to the scorer: weight = idf * boost and the 256-entry length-norm cache.
struct BM25SimWeight {
  float k1 = 0, b = 0, idf = 0, boost = 1, weight = 0, avg_dl = 0;
  std::array<float, 256> cache{};
};

class BM25Similarity {
 public:
  static constexpr float DEFAULT_BM25_K1 = 1.2f;
  static constexpr float DEFAULT_BM25_B = 0.75f;
  BM25Similarity() : k1_(DEFAULT_BM25_K1), b_(DEFAULT_BM25_B) {}
  BM25Similarity(float k1, float b) : k1_(k1), b_(b) {}
  float k1() const { return k1_; }
  float b() const { return b_; }

  static const std::array<float, 256>& norm_table() {
    static const std::array<float, 256> table = [] {
      std::array<float, 256> t{};
      for (int i = 1; i < 256; ++i) {
        float f = SmallFloat::byte315_to_float(static_cast<uint8_t>(i));
        t[i] = 1.0f / (f * f);
      }
      t[0] = 1.0f / t[255];
      return t;
    }();
    return table;
  }
  static float avg_field_length(const CollectionStatistics& cs) {
    if (cs.sum_total_term_freq <= 0) return 1.0f;
    int64_t dc = cs.doc_count == -1 ? cs.max_doc : cs.doc_count;
    return static_cast<float>(static_cast<double>(cs.sum_total_term_freq) / static_cast<double>(dc));
  }
  static uint8_t encode_norm_value(float boost, int32_t field_length) {
    return SmallFloat::float_to_byte315(boost / std::sqrt(static_cast<float>(field_length)));
  }
  static float idf(const TermStatistics* terms, size_t n, const CollectionStatistics& cs) {
    float sum = 0.0f;
    const double dc = static_cast<double>(cs.doc_count == -1 ? cs.max_doc : cs.doc_count);
    for (size_t i = 0; i < n; ++i) {
      const double df = static_cast<double>(terms[i].doc_freq);
      sum += static_cast<float>(std::log(1.0 + (dc - df + 0.5) / (df + 0.5)));
    }
    return sum;
  }
  BM25SimWeight compute_weight(const CollectionStatistics& cs, const TermStatistics* terms, size_t n, float boost) const {
    BM25SimWeight w;
    w.k1 = k1_;
    w.b = b_;
    w.avg_dl = avg_field_length(cs);
    w.idf = idf(terms, n, cs);
    const auto& nt = norm_table();
    for (int i = 0; i < 256; ++i) w.cache[i] = k1_ * ((1.0f - b_) + b_ * (nt[i] / w.avg_dl));
    w.boost = boost;
    w.weight = w.idf * boost;
    return w;
  }

 private:
  float k1_, b_;
};

}  // namespace rucene
