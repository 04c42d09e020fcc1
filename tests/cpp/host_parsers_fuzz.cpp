// Mutation fuzzer for the host-side file parsers (header-only, no HIP): every reader that takes bytes from an index
// directory must reject damaged input with a status — never read out of bounds, overflow, loop or crash. Built with
// -fsanitize=address,undefined by tests/test_host_parsers_fuzz.py, which hands it valid files written by the oracle:
//   argv: iterations seed doc tim tip nvm nvd liv fnm si segments [tim_pos tip_pos [cfe cfs]]
// (tim_pos / tip_pos: a second dictionary whose field 1 is indexed with positions, offsets and payloads: longs_size 3)
// Each iteration takes one valid file set, damages one or more files (bit flips, byte overwrites, truncation, insertion,
// block duplication) and runs every parser. Prints a tally; exit code 0 unless a sanitizer aborts the process.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../rucene_amd/csrc/host/compound_format.hpp"
#include "../../rucene_amd/csrc/host/doc_format.hpp"
#include "../../rucene_amd/csrc/host/field_infos_format.hpp"
#include "../../rucene_amd/csrc/host/norms_format.hpp"
#include "../../rucene_amd/csrc/host/segment_infos_format.hpp"
#include "../../rucene_amd/csrc/host/term_dict.hpp"

using Bytes = std::vector<uint8_t>;

static Bytes slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return Bytes(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

struct Rng {
  uint64_t s;
  uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  size_t below(size_t n) { return n ? (size_t)(next() % n) : 0; }
};

static void refoot(Bytes& b) {  // keep the CRC valid so that damage reaches the parsers behind the checksum
  if (b.size() < 16) return;
  const uint32_t crc = rucene::detail::crc32_ieee(b.data(), b.size() - 8);
  for (int i = 0; i < 4; ++i) b[b.size() - 8 + i] = 0;
  for (int i = 0; i < 4; ++i) b[b.size() - 4 + i] = (uint8_t)(crc >> (24 - 8 * i));
}

static void mutate(Bytes& b, Rng& r) {
  if (b.empty()) return;
  const int kind = (int)r.below(6);
  const size_t body = b.size() > 16 ? b.size() - 16 : b.size();
  switch (kind) {
    case 0: b[r.below(body)] ^= (uint8_t)(1u << r.below(8)); break;
    case 1: b[r.below(body)] = (uint8_t)r.next(); break;
    case 2: b.resize(r.below(b.size())); break;                                         // truncate
    case 3: b.insert(b.begin() + (long)r.below(body), (uint8_t)r.next()); break;         // insert
    case 4: { const size_t at = r.below(body); const size_t n = 1 + r.below(9); for (size_t i = 0; i < n && at + i < body; ++i) b[at + i] = 0xff; break; }  // vint bombs
    case 5: { const size_t at = r.below(body), n = 1 + r.below(64); Bytes blk(b.begin() + (long)at, b.begin() + (long)std::min(body, at + n)); b.insert(b.begin() + (long)r.below(body), blk.begin(), blk.end()); break; }
  }
  if (r.below(4) != 0) refoot(b);
}

int main(int argc, char** argv) {
  if (argc != 12 && argc != 14 && argc != 16) { std::fprintf(stderr, "usage: %s iterations seed doc tim tip nvm nvd liv fnm si segments [tim_pos tip_pos]\n", argv[0]); return 2; }
  const long iterations = std::atol(argv[1]);
  Rng rng{(uint64_t)std::atoll(argv[2])};
  const Bytes doc0 = slurp(argv[3]), tim0 = slurp(argv[4]), tip0 = slurp(argv[5]), nvm0 = slurp(argv[6]), nvd0 = slurp(argv[7]),
              liv0 = slurp(argv[8]), fnm0 = slurp(argv[9]), si0 = slurp(argv[10]), seg0 = slurp(argv[11]);
  const Bytes ptim0 = argc >= 14 ? slurp(argv[12]) : Bytes(), ptip0 = argc >= 14 ? slurp(argv[13]) : Bytes();
  const Bytes cfe0 = argc == 16 ? slurp(argv[14]) : Bytes(), cfs0 = argc == 16 ? slurp(argv[15]) : Bytes();
  const int32_t max_doc = 20000;
  long ok[9] = {0}, bad[9] = {0};
  auto tally = [&](int which, int rc) { (rc == 0 ? ok : bad)[which]++; };
  std::string why;
  for (long it = 0; it <= iterations; ++it) {
    Bytes doc = doc0, tim = tim0, tip = tip0, nvm = nvm0, nvd = nvd0, liv = liv0, fnm = fnm0, si = si0, seg = seg0;
    if (it > 0) {  // iteration 0 checks the undamaged set
      Bytes* all[] = {&doc, &tim, &tip, &nvm, &nvd, &liv, &fnm, &si, &seg};
      const int n_mut = 1 + (int)rng.below(3);
      for (int m = 0; m < n_mut; ++m) mutate(*all[rng.below(9)], rng);
      if (rng.below(3) == 0) mutate(tim, rng);  // the term dictionary is the largest parser: hit it more often
    }
    rucene::DocFileInfo dfi;
    tally(0, rucene::parse_doc_file(doc.data(), doc.size(), &dfi, &why));
    {
      const rucene::TermFieldInfo infos[2] = {{1, 2, 0}, {0, 1, 0}};
      std::unique_ptr<rucene::TermDictionary> dict;
      const int rc = rucene::TermDictionary::open(tim.data(), tim.size(), tip.data(), tip.size(), infos, 2, max_doc, &dict, &why);
      tally(1, rc);
      if (rc == 0) {
        rucene::TermState st;
        const uint8_t probe[6] = {'w', '0', '0', '0', '0', '1'};
        dict->lookup(1, probe, 6, &st);
        dict->lookup(0, probe, 0, &st);
        const int64_t offs[3] = {0, 6, 6};
        rucene::TermState sts[2];
        uint8_t found[2];
        dict->lookup_batch(1, probe, offs, 2, sts, found);
        rucene::TermPositions pos[2];
        dict->lookup_batch(1, probe, offs, 2, sts, found, pos);
        dict->lookup(1, probe, 6, &st, pos);
      }
    }
    if (!ptim0.empty()) {  // positions field: longs_size 3, last_pos_block_offset, pay pointers
      Bytes ptim = ptim0, ptip = ptip0;
      if (it > 0) { mutate(ptim, rng); if (rng.below(4) == 0) mutate(ptip, rng); }
      const rucene::TermFieldInfo infos[1] = {{1, 4, 1}};
      std::unique_ptr<rucene::TermDictionary> dict;
      const int rc = rucene::TermDictionary::open(ptim.data(), ptim.size(), ptip.data(), ptip.size(), infos, 1, max_doc, &dict, &why);
      tally(7, rc);
      if (rc == 0) {
        rucene::TermState st;
        rucene::TermPositions pos;
        const uint8_t probe[6] = {'w', '0', '0', '0', '0', '1'};
        dict->lookup(1, probe, 6, &st, &pos);
      }
    }
    if (!cfe0.empty()) {
      Bytes cfe = cfe0, cfs = cfs0;
      if (it > 0) { mutate(cfe, rng); if (rng.below(3) == 0) mutate(cfs, rng); }
      std::vector<rucene::CompoundEntry> entries;
      tally(8, rucene::read_lucene50_compound_entries(cfe.data(), cfe.size(), cfs.data(), cfs.size(), nullptr, &entries, &why));
    }
    std::vector<uint8_t> norms((size_t)max_doc);
    tally(2, rucene::read_lucene53_norms(nvm.data(), nvm.size(), nvd.data(), nvd.size(), 1, max_doc, norms.data(), &why));
    std::vector<uint64_t> words((size_t)(max_doc + 63) / 64);
    tally(3, rucene::read_lucene50_live_docs(liv.data(), liv.size(), max_doc, -1, words.data(), &why));
    std::vector<rucene::FieldInfoEntry> fis;
    tally(4, rucene::read_lucene60_field_infos(fnm.data(), fnm.size(), &fis, &why));
    rucene::SegmentInfoEntry sie;
    tally(5, rucene::read_lucene62_segment_info(si.data(), si.size(), nullptr, &sie, &why));
    std::vector<rucene::CommitSegmentEntry> segs;
    tally(6, rucene::read_segments_file(seg.data(), seg.size(), 2, &segs, &why));
    if (it == 0) for (int i = 0; i < 9; ++i) if (bad[i]) { std::fprintf(stderr, "parser %d rejects the undamaged file: %s\n", i, why.c_str()); return 3; }
  }
  const char* names[9] = {"doc", "tim/tip", "nvm/nvd", "liv", "fnm", "si", "segments_N", "tim/tip+pos", "cfe/cfs"};
  for (int i = 0; i < (ptim0.empty() ? 7 : cfe0.empty() ? 8 : 9); ++i) std::printf("%-10s accepted %ld rejected %ld\n", names[i], ok[i], bad[i]);
  return 0;
}
