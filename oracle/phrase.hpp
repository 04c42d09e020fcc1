// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's exact PhraseQuery evaluation (slop 0): PhraseWeight's BM25 weight over all the phrase's
// terms, the conjunction over the terms' position-bearing postings, and ExactPhraseScorer::phrase_freq.
// Groundwork for SURVEY.md §8(f)3; no product code evaluates phrases yet. SloppyPhraseScorer (slop > 0) is not restated.
//
// PARITY UNPINNED: the reference's only phrase test (query/phrase_query.rs:511) needs a live index directory and asserts
// nothing about scores; the source text is the only authority. tests/test_positions.py checks phrase_freq against brute
// force over the input positions and the scores against the BM25 formula.
//
// Follows (paths relative to /root/reference/src/core/search):
//   query/phrase_query.rs:136-186        PhraseQuery::create_weight: term statistics of every term, compute_weight(.., boost 1.0)
//   query/phrase_query.rs:262-330        PhraseWeight::create_scorer: None if any term is absent; postings with POSITIONS;
//                                        slop == 0 -> postings_freqs.sort() (by phrase position) -> ExactPhraseScorer
//   scorer/phrase_scorer.rs:29-57        PostingsIterAsScorer
//   scorer/phrase_scorer.rs:59-120       PostingsAndFreq ordering (position, then term count, then term bytes)
//   scorer/phrase_scorer.rs:122-243      ExactPhraseScorer::{new, advance_position, phrase_freq, do_next}
//   scorer/phrase_scorer.rs:245-294      score = sim.score(doc, phrase freq); next / advance
//   scorer/phrase_scorer.rs:296-317      PostingsAndPosition
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "positions.hpp"
#include "search.hpp"

namespace orc {

// phrase_scorer.rs:29-57: a postings iterator seen as a (never scored) Scorer so that ConjunctionScorer can drive it
struct PostingsIterAsScorer : Scorer {
  BlockPostingIterator* it;  // owned by the phrase scorer
  explicit PostingsIterAsScorer(BlockPostingIterator* i) : it(i) {}
  float score() override { throw OracleError(E_ILLEGAL_STATE, "PostingsIterAsScorer::score is unreachable"); }
  int32_t doc_id() const override { return it->doc_id(); }
  int32_t next() override { return it->next(); }
  int32_t advance(int32_t t) override { return it->advance(t); }
  size_t cost() const override { return it->cost(); }
};

struct ExactPhraseScorer : Scorer {
  struct PostingsAndPosition {  // phrase_scorer.rs:296-317
    BlockPostingIterator* postings;
    int32_t pos = -1, offset = 0, freq = 0, up_to = 1;
  };
  std::vector<std::unique_ptr<BlockPostingIterator>> iterators;
  std::vector<PostingsAndPosition> postings;  // in phrase-position order (the caller sorted them)
  std::unique_ptr<ConjunctionScorer> conjunction;
  int32_t freq_ = 0;
  bool needs_scores;
  const BM25Weight* weight;
  const uint8_t* norms;

  // phrase_scorer.rs:131-160. `its[i]` belongs to the term at phrase position `offsets[i]`, already sorted.
  ExactPhraseScorer(std::vector<std::unique_ptr<BlockPostingIterator>> its, const std::vector<int32_t>& offsets, const BM25Weight* w,
                    const uint8_t* norms_, bool needs_scores_)
      : iterators(std::move(its)), needs_scores(needs_scores_), weight(w), norms(norms_) {
    std::vector<ScorerBox> as_scorers;
    for (size_t i = 0; i < iterators.size(); i++) {
      as_scorers.emplace_back(new PostingsIterAsScorer(iterators[i].get()));
      PostingsAndPosition pp;
      pp.postings = iterators[i].get();
      pp.offset = offsets[i];
      postings.push_back(pp);
    }
    conjunction.reset(new ConjunctionScorer(std::move(as_scorers)));
  }
  // phrase_scorer.rs:166-177
  static bool advance_position(PostingsAndPosition& p, int32_t target) {
    while (p.pos < target) {
      if (p.up_to == p.freq) return false;
      p.pos = p.postings->next_position();
      p.up_to++;
    }
    return true;
  }
  // phrase_scorer.rs:179-229
  int32_t phrase_freq() {
    for (PostingsAndPosition& pp : postings) {
      pp.freq = pp.postings->freq();
      pp.pos = pp.postings->next_position();
      pp.up_to = 1;
    }
    int32_t freq = 0;
    PostingsAndPosition& lead = postings[0];
    bool done = false;
    while (!done) {  // 'advanceHead
      const int32_t phrase_pos = lead.pos - lead.offset;
      bool restart = false;
      for (size_t i = 1; i < postings.size(); i++) {
        PostingsAndPosition& posting = postings[i];
        const int32_t expected_pos = phrase_pos + posting.offset;
        if (!advance_position(posting, expected_pos)) { done = true; break; }
        if (posting.pos != expected_pos) {
          const int32_t target = posting.pos - posting.offset + lead.offset;
          if (advance_position(lead, target)) restart = true; else done = true;
          break;
        }
      }
      if (done) break;
      if (restart) continue;
      freq++;
      if (!needs_scores) break;
      if (lead.up_to == lead.freq) break;
      lead.pos = lead.postings->next_position();
      lead.up_to++;
    }
    freq_ = freq;
    return freq_;
  }
  // phrase_scorer.rs:231-243
  int32_t do_next(int32_t doc) {
    while (true) {
      if (doc == NO_MORE_DOCS) return NO_MORE_DOCS;
      if (phrase_freq() > 0) return doc;
      doc = conjunction->next();
    }
  }
  int32_t freq() const { return freq_; }
  float score() override {  // phrase_scorer.rs:246-251 -> BM25SimScorer::score(doc, freq)
    const int32_t d = conjunction->doc_id();
    return bm25_compute_score(weight->weight, weight->k1, (float)freq_, norms != nullptr, norms ? weight->cache[norms[d] & 0xFF] : 0.0f);
  }
  int32_t doc_id() const override { return conjunction->doc_id(); }
  int32_t next() override { return do_next(conjunction->next()); }
  int32_t advance(int32_t target) override { return do_next(conjunction->advance(target)); }
  size_t cost() const override { return conjunction->cost(); }
};

}  // namespace orc
