// ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.
// CPU restatement of Rucene's FST<ByteSequenceOutput> (the terms-index automaton of the block-tree term
// dictionary): builder with suffix sharing, the on-disk node/arc encoding, and the arc-walking reader.
//
// Pinned by the reference's own tests: `test_fst` (util/fst/fst_reader.rs:1079-1109, "cat … dogs"),
// ByteSequenceOutput prefix/cat/subtract/read-write (util/fst/bytes_output.rs:250-308) and the reverse
// bytes reader (util/fst/bytes_store.rs:654-670) — ported in tests/test_oracle_kat.py.
// PARITY UNPINNED for the exact byte image of a saved FST (no reference test holds one); the encoding below
// follows the source text line by line so that a Rucene-written .tip is readable and vice versa.
//
// Follows (paths relative to /root/reference/src/core/util/fst):
//   bytes_output.rs:47-118,133-200   ByteSequenceOutput prefix/cat/subtract, factory common/subtract/add, write/read
//   bytes_store.rs                   reverse reader (position decrements), reverse(), skip/copy used by add_node
//   fst_builder.rs:38-150            FstBuilder::build/init/compile_node
//   fst_builder.rs:152-245           freeze_tail (min_suffix_count1 = min_suffix_count2 = 0 at every call site on
//                                    this path, blocktree_writer.rs:947-957, so the prune arms are dead)
//   fst_builder.rs:247-340           add
//   fst_builder.rs:342-372           finish
//   fst_builder.rs:425-600           NodeHash (find-or-insert on full node equality)
//   fst_builder.rs:610-760           UnCompiledNode
//   fst_reader.rs:25-53              flag bits, ARCS_AS_FIXED_ARRAY, FIXED_ARRAY_* thresholds, versions
//   fst_reader.rs:205-275            FST::from_input
//   fst_reader.rs:385-405            root_arc
//   fst_reader.rs:413-520            find_target_arc
//   fst_reader.rs:545-700            read_first_real_arc / read_next_real_arc / seek_to_next_node
//   fst_reader.rs:715-880            add_node / should_expand
//   fst_reader.rs:890-960            finish / save
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "store.hpp"

namespace orc {

using Bytes = std::vector<uint8_t>;

// ---- ByteSequenceOutput algebra (bytes_output.rs) ------------------------------------------------------------------
inline Bytes bso_common(const Bytes& a, const Bytes& b) {  // factory.common == ByteSequenceOutput::prefix
  Bytes r;
  for (size_t i = 0; i < std::min(a.size(), b.size()) && a[i] == b[i]; i++) r.push_back(a[i]);
  return r;
}
inline Bytes bso_subtract(const Bytes& o1, const Bytes& o2) {  // o1 minus its prefix o2
  if (o2.empty()) return o1;
  if (o1.size() < o2.size() || !std::equal(o2.begin(), o2.end(), o1.begin()))
    throw OracleError(E_ILLEGAL_STATE, "subtract: not a prefix");
  return Bytes(o1.begin() + o2.size(), o1.end());
}
inline Bytes bso_add(const Bytes& prefix, const Bytes& output) {  // cat
  Bytes r(prefix);
  r.insert(r.end(), output.begin(), output.end());
  return r;
}

constexpr uint8_t FST_BIT_FINAL_ARC = 1, FST_BIT_LAST_ARC = 1 << 1, FST_BIT_TARGET_NEXT = 1 << 2,
                  FST_BIT_STOP_NODE = 1 << 3, FST_BIT_ARC_HAS_OUTPUT = 1 << 4, FST_BIT_ARC_HAS_FINAL_OUTPUT = 1 << 5;
constexpr uint8_t FST_ARCS_AS_FIXED_ARRAY = FST_BIT_ARC_HAS_FINAL_OUTPUT;
constexpr int FST_FIXED_ARRAY_SHALLOW_DISTANCE = 3, FST_FIXED_ARRAY_NUM_ARCS_SHALLOW = 5, FST_FIXED_ARRAY_NUM_ARCS_DEEP = 10;
constexpr int32_t FST_VERSION_PACKED = 3, FST_VERSION_VINT_TARGET = 4, FST_VERSION_NO_NODE_ARC_COUNTS = 5,
                  FST_VERSION_PACKED_REMOVED = 6, FST_VERSION_CURRENT = 6;
constexpr int64_t FST_FINAL_END_NODE = -1, FST_NON_FINAL_END_NODE = 0;
constexpr int FST_END_LABEL = -1;

// bytes_store.rs reverse reader: read_byte returns b[pos] and decrements.
struct RevReader {
  const uint8_t* b;
  int64_t n;
  int64_t pos;
  uint8_t read_byte() {
    if (pos < 0 || pos >= n) throw OracleError(E_UNEXPECTED_EOF, "fst: read past end");
    return b[pos--];
  }
  void skip_bytes(int64_t k) { pos -= k; }
  int32_t read_vint() {
    uint32_t v = 0;
    for (int shift = 0; shift < 35; shift += 7) {
      uint8_t x = read_byte();
      v |= (uint32_t)(x & 0x7f) << shift;
      if (!(x & 0x80)) return (int32_t)v;
    }
    throw OracleError(E_CORRUPT_INDEX, "Invalid vInt detected");
  }
  int64_t read_vlong() {
    uint64_t v = 0;
    for (int shift = 0; shift < 63; shift += 7) {
      uint8_t x = read_byte();
      v |= (uint64_t)(x & 0x7f) << shift;
      if (!(x & 0x80)) return (int64_t)v;
    }
    throw OracleError(E_CORRUPT_INDEX, "Invalid vLong detected");
  }
  Bytes read_output() {  // bytes_output.rs read(): vint length + bytes
    int32_t len = read_vint();
    Bytes r((size_t)len);
    for (int32_t i = 0; i < len; i++) r[i] = read_byte();
    return r;
  }
  void skip_output() { int32_t len = read_vint(); skip_bytes(len); }
};

struct FstArc {
  uint8_t flags = 0;
  int label = 0;
  Bytes output;             // empty == None / NO_OUTPUT
  Bytes next_final_output;  // empty == None / NO_OUTPUT
  int64_t next_arc = 0;
  int64_t target = 0;
  int64_t arc_start_position = 0;
  int64_t bytes_per_arc = 0;
  int64_t arc_index = 0;
  int64_t num_arcs = 0;
  bool is_last() const { return flags & FST_BIT_LAST_ARC; }
  bool is_final() const { return flags & FST_BIT_FINAL_ARC; }
};

struct Fst {
  bool has_empty_output = false;
  Bytes empty_output;
  int64_t start_node = -1;
  int32_t version = FST_VERSION_CURRENT;
  Bytes bytes;  // builder: starts with one 0 byte (fst_reader.rs:190-192) so that address 0 is never a node

  RevReader reader() const { return RevReader{bytes.data(), (int64_t)bytes.size(), 0}; }

  // fst_reader.rs:205-275
  static Fst from_input(ByteIn& in) {
    Fst f;
    int32_t magic = in.read_int();
    if (magic != CODEC_MAGIC) throw OracleError(E_CORRUPT_INDEX, "fst: codec header mismatch");
    if (in.read_string() != "FST") throw OracleError(E_CORRUPT_INDEX, "fst: codec mismatch");
    f.version = in.read_int();
    if (f.version < FST_VERSION_PACKED || f.version > FST_VERSION_CURRENT)
      throw OracleError(E_CORRUPT_INDEX, "fst: version out of range");
    if (f.version < FST_VERSION_PACKED_REMOVED && in.read_byte() == 1)
      throw OracleError(E_CORRUPT_INDEX, "Cannot read packed FSTs anymore");
    if (in.read_byte() == 1) {
      int32_t num_bytes = in.read_vint();
      Bytes tmp((size_t)num_bytes);
      in.read_exact(tmp.data(), tmp.size());
      RevReader r{tmp.data(), (int64_t)tmp.size(), num_bytes > 0 ? num_bytes - 1 : 0};
      f.empty_output = r.read_output();
      f.has_empty_output = true;
    }
    uint8_t t = in.read_byte();
    if (t != 0) throw OracleError(t <= 2 ? E_UNSUPPORTED : E_ILLEGAL_STATE, "fst: only BYTE1 inputs are restated");
    f.start_node = in.read_vlong();
    if (f.version < FST_VERSION_NO_NODE_ARC_COUNTS) { in.read_vlong(); in.read_vlong(); in.read_vlong(); }
    int64_t num_bytes = in.read_vlong();
    f.bytes.resize((size_t)num_bytes);
    in.read_exact(f.bytes.data(), f.bytes.size());
    return f;
  }

  // fst_reader.rs:890-960 (codec_util write_header = magic, name, version)
  void save(ByteOut& out) const {
    if (start_node == -1) throw OracleError(E_ILLEGAL_STATE, "call finish first!");
    out.write_int(CODEC_MAGIC);
    out.write_string("FST");
    out.write_int(FST_VERSION_CURRENT);
    if (has_empty_output) {
      out.write_byte(1);
      ByteOut tmp;
      tmp.write_vint((int32_t)empty_output.size());
      tmp.write_bytes(empty_output.data(), empty_output.size());
      std::reverse(tmp.buf.begin(), tmp.buf.end());
      out.write_vint((int32_t)tmp.buf.size());
      out.write_bytes(tmp.buf.data(), tmp.buf.size());
    } else {
      out.write_byte(0);
    }
    out.write_byte(0);  // InputType::Byte1
    out.write_vlong(start_node);
    out.write_vlong((int64_t)bytes.size());
    out.write_bytes(bytes.data(), bytes.size());
  }

  // fst_reader.rs:385-405
  FstArc root_arc() const {
    FstArc arc;
    if (has_empty_output) {
      arc.flags = FST_BIT_FINAL_ARC | FST_BIT_LAST_ARC;
      arc.next_final_output = empty_output;
      if (!empty_output.empty()) arc.flags |= FST_BIT_ARC_HAS_FINAL_OUTPUT;
    } else {
      arc.flags = FST_BIT_LAST_ARC;
    }
    arc.target = start_node;
    return arc;
  }

  static bool target_has_arc(int64_t target) { return target > 0; }

  int64_t read_unpacked_node(RevReader& r) const {
    if (version < FST_VERSION_VINT_TARGET) {  // big-endian int through the reverse reader
      uint32_t v = 0;
      for (int i = 0; i < 4; i++) v = (v << 8) | r.read_byte();
      return (int32_t)v;
    }
    return r.read_vlong();
  }

  void read_array_header(RevReader& r, FstArc& arc) const {
    arc.num_arcs = r.read_vint();
    if (version >= FST_VERSION_VINT_TARGET) {
      arc.bytes_per_arc = r.read_vint();
    } else {
      uint32_t v = 0;
      for (int i = 0; i < 4; i++) v = (v << 8) | r.read_byte();
      arc.bytes_per_arc = (int32_t)v;
    }
    arc.arc_start_position = r.pos;
  }

  // fst_reader.rs:655-685
  void seek_to_next_node(RevReader& r) const {
    for (;;) {
      uint8_t flags = r.read_byte();
      r.read_byte();  // label (BYTE1)
      if (flags & FST_BIT_ARC_HAS_OUTPUT) r.skip_output();
      if (flags & FST_BIT_ARC_HAS_FINAL_OUTPUT) r.skip_output();
      if (!(flags & FST_BIT_STOP_NODE) && !(flags & FST_BIT_TARGET_NEXT)) read_unpacked_node(r);
      if (flags & FST_BIT_LAST_ARC) return;
    }
  }

  // fst_reader.rs:605-653
  void read_next_real_arc(FstArc& arc, RevReader& r) const {
    if (arc.bytes_per_arc > 0) {
      r.pos = arc.arc_start_position;
      r.skip_bytes(arc.arc_index * arc.bytes_per_arc);
      arc.arc_index++;
    } else {
      r.pos = arc.next_arc;
    }
    arc.flags = r.read_byte();
    arc.label = r.read_byte();
    arc.output = (arc.flags & FST_BIT_ARC_HAS_OUTPUT) ? r.read_output() : Bytes();
    arc.next_final_output = (arc.flags & FST_BIT_ARC_HAS_FINAL_OUTPUT) ? r.read_output() : Bytes();
    if (arc.flags & FST_BIT_STOP_NODE) {
      arc.target = FST_FINAL_END_NODE;
      arc.next_arc = r.pos;
    } else if (arc.flags & FST_BIT_TARGET_NEXT) {
      arc.next_arc = r.pos;
      if (!(arc.flags & FST_BIT_LAST_ARC)) {
        if (arc.bytes_per_arc > 0) {
          r.pos = arc.arc_start_position;
          r.skip_bytes(arc.bytes_per_arc * arc.num_arcs);
        } else {
          seek_to_next_node(r);
        }
      }
      arc.target = r.pos;
    } else {
      arc.target = read_unpacked_node(r);
      arc.next_arc = r.pos;
    }
  }

  // fst_reader.rs:545-567
  FstArc read_first_real_arc(int64_t node, RevReader& r) const {
    r.pos = node;
    FstArc arc;
    if (r.read_byte() == FST_ARCS_AS_FIXED_ARRAY) {
      read_array_header(r, arc);
      arc.arc_index = 0;
    } else {
      arc.next_arc = node;
    }
    read_next_real_arc(arc, r);
    return arc;
  }

  // fst_reader.rs:413-520 (labels >= 0 only; END_LABEL handling is used by the enumerator below instead)
  bool find_target_arc(int label, const FstArc& incoming, FstArc& out, RevReader& r) const {
    if (!target_has_arc(incoming.target)) return false;
    r.pos = incoming.target;
    FstArc arc;
    if (r.read_byte() == FST_ARCS_AS_FIXED_ARRAY) {
      read_array_header(r, arc);
      int64_t low = 0, high = arc.num_arcs - 1;
      while (low <= high) {
        int64_t mid = (low + high) >> 1;
        r.pos = arc.arc_start_position;
        r.skip_bytes(arc.bytes_per_arc * mid + 1);
        int cur = r.read_byte();
        int cmp = cur - label;
        if (cmp < 0) {
          low = mid + 1;
        } else if (cmp > 0) {
          if (mid == 0) break;
          high = mid - 1;
        } else {
          arc.arc_index = mid;
          read_next_real_arc(arc, r);
          out = arc;
          return true;
        }
      }
      return false;
    }
    arc = read_first_real_arc(incoming.target, r);
    for (;;) {
      if (arc.label == label) { out = arc; return true; }
      if (arc.label > label || arc.is_last()) return false;
      read_next_real_arc(arc, r);
    }
  }

  // fst_reader.rs:283-312 FST::get
  bool get(const Bytes& input, Bytes& result) const {
    FstArc arc = root_arc();
    Bytes output;
    RevReader r = reader();
    for (uint8_t label : input) {
      FstArc next;
      if (!find_target_arc(label, arc, next, r)) return false;
      arc = next;
      output = bso_add(output, arc.output);
    }
    if (!arc.is_final()) return false;
    result = bso_add(output, arc.next_final_output);
    return true;
  }

  // fst_iteartor.rs BytesRefFSTIterator::next, restated as a depth-first walk: every accepted input in
  // byte order with its total output.
  void enumerate(const std::function<void(const Bytes&, const Bytes&)>& visit) const {
    if (has_empty_output) visit(Bytes(), empty_output);
    if (!target_has_arc(start_node)) return;
    Bytes input;
    walk(start_node, input, Bytes(), visit);
  }

 private:
  void walk(int64_t node, Bytes& input, const Bytes& output,
            const std::function<void(const Bytes&, const Bytes&)>& visit) const {
    RevReader r = reader();
    FstArc arc = read_first_real_arc(node, r);
    for (;;) {
      input.push_back((uint8_t)arc.label);
      Bytes out = bso_add(output, arc.output);
      if (arc.is_final()) visit(input, bso_add(out, arc.next_final_output));
      if (target_has_arc(arc.target)) walk(arc.target, input, out, visit);
      input.pop_back();
      if (arc.is_last()) break;
      read_next_real_arc(arc, r);
    }
  }
};

// ---- builder ----------------------------------------------------------------------------------------------------------

struct FstBuilder {
  struct BArc {
    int label = 0;
    bool compiled = false;
    int64_t target = 0;  // compiled address, or frontier index while uncompiled
    bool is_final = false;
    Bytes output, next_final_output;
  };
  struct UNode {  // fst_builder.rs:610-760
    std::vector<BArc> arcs;  // only [0, num_arcs) is live
    size_t num_arcs = 0;
    Bytes output;
    bool is_final = false;
    int64_t input_count = 0;
    int depth = 0;
    void clear() { num_arcs = 0; is_final = false; output.clear(); input_count = 0; }
    BArc& last(int label_to_match) {
      if (num_arcs == 0 || arcs[num_arcs - 1].label != label_to_match) throw OracleError(E_ILLEGAL_STATE, "fst: last arc mismatch");
      return arcs[num_arcs - 1];
    }
    void add_arc(int label, int64_t frontier_index) {
      if (label <= 0) throw OracleError(E_ILLEGAL_ARGUMENT, "fst: label must be > 0");  // fst_builder.rs:690
      if (num_arcs && label <= arcs[num_arcs - 1].label) throw OracleError(E_ILLEGAL_STATE, "fst: arcs out of order");
      BArc a;
      a.label = label;
      a.target = frontier_index;
      if (num_arcs == arcs.size()) arcs.push_back(a); else arcs[num_arcs] = a;
      num_arcs++;
    }
    void prepend_output(const Bytes& prefix) {
      for (size_t i = 0; i < num_arcs; i++) arcs[i].output = bso_add(prefix, arcs[i].output);
      if (is_final) output = bso_add(prefix, output);
    }
  };

  Fst fst;
  bool do_share_suffix, do_share_non_singleton_nodes, allow_array_arcs;
  uint32_t share_max_tail_length;
  std::vector<int> last_input;
  std::vector<UNode> frontier;
  int64_t last_frozen_node = 0;
  std::map<std::string, int64_t> dedup;  // NodeHash: full-equality find-or-insert (fst_builder.rs:425-560)

  // fst_builder.rs:38-98. FstBuilder::new == (true, true); the block-tree index uses (true, false).
  FstBuilder(bool share_suffix, bool share_non_singleton, uint32_t max_tail = INT32_MAX, bool array_arcs = true)
      : do_share_suffix(share_suffix), do_share_non_singleton_nodes(share_non_singleton), allow_array_arcs(array_arcs),
        share_max_tail_length(max_tail) {
    fst.bytes.push_back(0);
    for (int i = 0; i < 10; i++) { frontier.emplace_back(); frontier.back().depth = i; }
  }

  bool should_expand(const UNode& node) const {  // fst_reader.rs:872-878
    return allow_array_arcs && ((node.depth <= FST_FIXED_ARRAY_SHALLOW_DISTANCE &&
                                 node.num_arcs >= (size_t)FST_FIXED_ARRAY_NUM_ARCS_SHALLOW) ||
                                node.num_arcs >= (size_t)FST_FIXED_ARRAY_NUM_ARCS_DEEP);
  }

  static void put_vint(Bytes& b, int32_t v) {
    uint32_t u = (uint32_t)v;
    while (u & ~0x7fu) { b.push_back((uint8_t)((u & 0x7f) | 0x80)); u >>= 7; }
    b.push_back((uint8_t)u);
  }
  static void put_vlong(Bytes& b, int64_t v) {
    uint64_t u = (uint64_t)v;
    while (u & ~0x7full) { b.push_back((uint8_t)((u & 0x7f) | 0x80)); u >>= 7; }
    b.push_back((uint8_t)u);
  }
  static void put_output(Bytes& b, const Bytes& o) { put_vint(b, (int32_t)o.size()); b.insert(b.end(), o.begin(), o.end()); }

  // fst_reader.rs:715-870
  int64_t add_node(const UNode& node) {
    if (node.num_arcs == 0) return node.is_final ? FST_FINAL_END_NODE : FST_NON_FINAL_END_NODE;
    Bytes& bs = fst.bytes;
    const size_t start_address = bs.size();
    const bool do_fixed_array = should_expand(node);
    std::vector<size_t> bytes_per_arc(node.num_arcs, 0);
    size_t last_arc_start = bs.size(), max_bytes_per_arc = 0;
    for (size_t idx = 0; idx < node.num_arcs; idx++) {
      const BArc& arc = node.arcs[idx];
      const int64_t target = arc.target;
      uint8_t flags = 0;
      if (idx == node.num_arcs - 1) flags += FST_BIT_LAST_ARC;
      if (last_frozen_node == target && !do_fixed_array) flags += FST_BIT_TARGET_NEXT;
      if (arc.is_final) {
        flags += FST_BIT_FINAL_ARC;
        if (!arc.next_final_output.empty()) flags += FST_BIT_ARC_HAS_FINAL_OUTPUT;
      } else if (!arc.next_final_output.empty()) {
        throw OracleError(E_ILLEGAL_STATE, "fst: final output on a non-final arc");
      }
      const bool target_has_arcs = target > 0;
      if (!target_has_arcs) flags += FST_BIT_STOP_NODE;
      if (!arc.output.empty()) flags += FST_BIT_ARC_HAS_OUTPUT;
      bs.push_back(flags);
      if (arc.label <= 0 || arc.label > 255) throw OracleError(E_ILLEGAL_ARGUMENT, "fst: BYTE1 label out of range");
      bs.push_back((uint8_t)arc.label);
      if (!arc.output.empty()) put_output(bs, arc.output);
      if (!arc.next_final_output.empty()) put_output(bs, arc.next_final_output);
      if (target_has_arcs && !(flags & FST_BIT_TARGET_NEXT)) put_vlong(bs, target);
      if (do_fixed_array) {
        bytes_per_arc[idx] = bs.size() - last_arc_start;
        last_arc_start = bs.size();
        max_bytes_per_arc = std::max(max_bytes_per_arc, bytes_per_arc[idx]);
      }
    }
    if (do_fixed_array) {
      Bytes header;
      header.push_back(FST_ARCS_AS_FIXED_ARRAY);
      put_vint(header, (int32_t)node.num_arcs);
      put_vint(header, (int32_t)max_bytes_per_arc);
      const size_t fixed_array_start = start_address + header.size();
      size_t src_pos = bs.size();
      size_t dest_pos = fixed_array_start + node.num_arcs * max_bytes_per_arc;
      if (dest_pos > src_pos) {
        bs.resize(dest_pos, 0);  // skip_bytes
        for (size_t i = 0; i < node.num_arcs; i++) {
          size_t arc_idx = node.num_arcs - 1 - i;
          dest_pos -= max_bytes_per_arc;
          src_pos -= bytes_per_arc[arc_idx];
          if (src_pos != dest_pos) std::memmove(&bs[dest_pos], &bs[src_pos], bytes_per_arc[arc_idx]);
        }
      }
      std::memcpy(&bs[start_address], header.data(), header.size());
    }
    const size_t this_node_address = bs.size() - 1;
    std::reverse(bs.begin() + start_address, bs.begin() + this_node_address + 1);
    return (int64_t)this_node_address;
  }

  static std::string node_key(const UNode& node) {
    std::string k;
    for (size_t i = 0; i < node.num_arcs; i++) {
      const BArc& a = node.arcs[i];
      k.append((const char*)&a.label, sizeof a.label);
      k.append((const char*)&a.target, sizeof a.target);
      k.push_back(a.is_final ? 1 : 0);
      uint32_t n1 = (uint32_t)a.output.size(), n2 = (uint32_t)a.next_final_output.size();
      k.append((const char*)&n1, 4);
      k.append((const char*)a.output.data(), n1);
      k.append((const char*)&n2, 4);
      k.append((const char*)a.next_final_output.data(), n2);
    }
    return k;
  }

  // fst_builder.rs:105-150
  int64_t compile_node(size_t node_index, uint32_t tail_length) {
    UNode& n = frontier[node_index];
    const size_t bytes_pos_start = fst.bytes.size();
    int64_t node;
    if (do_share_suffix && (do_share_non_singleton_nodes || n.num_arcs <= 1) && tail_length <= share_max_tail_length) {
      if (n.num_arcs == 0) {
        node = add_node(n);
        last_frozen_node = node;
      } else {
        std::string key = node_key(n);
        auto it = dedup.find(key);
        if (it != dedup.end()) {
          node = it->second;
        } else {
          node = add_node(n);
          dedup.emplace(std::move(key), node);
        }
      }
    } else {
      node = add_node(n);
    }
    if (fst.bytes.size() != bytes_pos_start) last_frozen_node = node;
    n.clear();
    return node;
  }

  // fst_builder.rs:152-245 with min_suffix_count1 == min_suffix_count2 == 0: never prune, always compile
  void freeze_tail(size_t prefix_len_plus1) {
    const size_t down_to = std::max<size_t>(1, prefix_len_plus1);
    if (last_input.size() < down_to) return;
    for (size_t i = 0; i <= last_input.size() - down_to; i++) {
      const size_t idx = last_input.size() - i;
      Bytes next_final_output = frontier[idx].output;
      const bool is_final = frontier[idx].is_final || frontier[idx].num_arcs == 0;
      const uint32_t tail_len = (uint32_t)(1 + last_input.size() - idx);
      int64_t compiled = compile_node(idx, tail_len);
      BArc& arc = frontier[idx - 1].last(last_input[idx - 1]);  // replace_last
      arc.compiled = true;
      arc.target = compiled;
      arc.next_final_output = std::move(next_final_output);
      arc.is_final = is_final;
    }
  }

  // fst_builder.rs:247-340
  void add(const Bytes& input_bytes, Bytes output) {
    std::vector<int> input(input_bytes.begin(), input_bytes.end());
    if (!last_input.empty() && !(input > last_input)) throw OracleError(E_ILLEGAL_STATE, "fst: inputs out of order");
    while (frontier.size() < input.size() + 2) { frontier.emplace_back(); frontier.back().depth = (int)frontier.size() - 1; }
    if (input.empty()) {
      frontier[0].input_count++;
      frontier[0].is_final = true;
      if (fst.has_empty_output) throw OracleError(E_UNSUPPORTED, "ByteSequenceOutput merge");  // set_empty_output -> merge
      fst.has_empty_output = true;
      fst.empty_output = output;
      return;
    }
    size_t pos1 = 0;
    const size_t pos1_stop = std::min(last_input.size(), input.size());
    for (;;) {
      frontier[pos1].input_count++;
      if (pos1 >= pos1_stop || last_input[pos1] != input[pos1]) break;
      pos1++;
    }
    const size_t prefix_len_plus1 = pos1 + 1;
    freeze_tail(prefix_len_plus1);
    for (size_t i = prefix_len_plus1; i <= input.size(); i++) {
      frontier[i - 1].add_arc(input[i - 1], (int64_t)i);
      frontier[i].input_count++;
    }
    UNode& last_node = frontier[input.size()];
    if (last_input.size() != input.size() || prefix_len_plus1 != input.size() + 1) {
      last_node.is_final = true;
      last_node.output.clear();
    }
    for (size_t i = 1; i < prefix_len_plus1; i++) {
      BArc& parent_arc = frontier[i - 1].last(input[i - 1]);
      Bytes last_output = parent_arc.output;
      Bytes common_prefix;
      if (!last_output.empty()) {
        common_prefix = bso_common(output, last_output);
        frontier[i].prepend_output(bso_subtract(last_output, common_prefix));
      }
      output = bso_subtract(output, common_prefix);
      if (!last_output.empty()) parent_arc.output = common_prefix;
    }
    if (last_input.size() == input.size() && prefix_len_plus1 == input.size() + 1) {
      throw OracleError(E_UNSUPPORTED, "ByteSequenceOutput merge");  // same input twice -> outputs().merge
    } else {
      frontier[prefix_len_plus1 - 1].last(input[prefix_len_plus1 - 1]).output = output;
    }
    last_input = input;
  }

  // fst_builder.rs:342-372 + fst_reader.rs:880-900. Returns false for the "no FST" case (None).
  bool finish() {
    freeze_tail(0);
    if (frontier[0].num_arcs == 0 && !fst.has_empty_output) return false;
    int64_t node = compile_node(0, (uint32_t)last_input.size());
    if (fst.start_node != -1) throw OracleError(E_ILLEGAL_STATE, "already finished");
    fst.start_node = node == FST_FINAL_END_NODE ? 0 : node;
    return true;
  }
};

}  // namespace orc
