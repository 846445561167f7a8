// scan_filter.cpp -- pushed-down predicate trees -> conjunctive normal form over normalised leaves.
//
// The reference pushes no filters (filter_pushdown = false, src/scanner/read_arrow.cpp:47-48); what is accepted here is
// what DuckDB's TableFilterSet can hand a scan: constant comparisons, IS [NOT] NULL, IN-lists,
// AND / OR trees.  On integers every comparison is an inclusive range (v < c is [MIN, c-1], v <> c is NOT [c, c]), so
// the kernel knows four leaf forms only and adjacent range conjuncts on one column intersect into one leaf.
#include <algorithm>
#include <limits>

#include "scan_operator.hpp"

namespace miarrow {

namespace {
constexpr int64_t kMin = std::numeric_limits<int64_t>::min(), kMax = std::numeric_limits<int64_t>::max();
constexpr size_t kMaxLeaves = static_cast<size_t>(device::kMaxFilterLeaves);

FilterLeaf LeafOf(const mi_filter_node& n) {
  if (!n.column || !*n.column) throw InvalidInputException("filter leaf without a column name");
  FilterLeaf l;
  l.column = n.column;
  l.op = device::kLeafRange;
  l.lo = kMin;
  l.hi = kMax;
  const int64_t c = n.value;
  auto closed = [&](int64_t lo, int64_t hi) { l.lo = lo; l.hi = hi; l.lo_open = l.hi_open = false; };
  if (n.str_value || n.str_values) {   // byte-string constants: a VARCHAR / BLOB column
    l.is_string = true;
    l.op = device::kLeafStrIn;
    auto add = [&](const char* p, int32_t len) {
      if (!p || len < 0) throw InvalidInputException("string filter constant without bytes");
      l.str_values.emplace_back(p, static_cast<size_t>(len));
    };
    switch (n.op) {
      case MI_F_EQ: add(n.str_value, n.str_len); break;
      case MI_F_NE: add(n.str_value, n.str_len); l.negate = true; break;
      case MI_F_IN:
        if (n.n_values < 0 || (n.n_values > 0 && (!n.str_values || !n.str_lens))) throw InvalidInputException("IN filter without values");
        if (n.n_values > 256) throw NotImplementedException("IN-list with more than 256 values is not pushed down");
        for (int32_t k = 0; k < n.n_values; k++) add(n.str_values[k], n.str_lens[k]);
        break;
      case MI_F_LT: case MI_F_LE: case MI_F_GT: case MI_F_GE: case MI_F_STARTS_WITH: {
        // ordering and prefix tests: one range leaf [lower, upper] with open / closed ends (byte-wise order)
        if (!n.str_value || n.str_len < 0) throw InvalidInputException("string filter constant without bytes");
        const std::string c(n.str_value, static_cast<size_t>(n.str_len));
        l.op = device::kLeafStrRange;
        l.str_values.assign(2, std::string());
        l.lo_open = l.hi_open = true;
        if (n.op == MI_F_LT || n.op == MI_F_LE) {
          l.str_values[1] = c;
          l.hi_open = false;
          l.hi_incl = n.op == MI_F_LE;
        } else if (n.op == MI_F_GT || n.op == MI_F_GE) {
          l.str_values[0] = c;
          l.lo_open = false;
          l.lo_incl = n.op == MI_F_GE;
        } else {
          // begins with c  <=>  c <= row < successor(c), the successor being c with its last byte that is not 0xFF
          // incremented and everything behind it dropped (all 0xFF or empty: no upper bound)
          l.str_values[0] = c;
          l.lo_open = false;
          l.lo_incl = true;
          std::string up = c;
          while (!up.empty() && static_cast<unsigned char>(up.back()) == 0xFF) up.pop_back();
          if (!up.empty()) {
            up.back() = static_cast<char>(static_cast<unsigned char>(up.back()) + 1);
            l.str_values[1] = up;
            l.hi_open = false;
            l.hi_incl = false;
          }
        }
        return l;
      }
      default:
        throw NotImplementedException("this comparison is not pushed down on VARCHAR / BLOB columns (column '" + l.column + "')");
    }
    std::sort(l.str_values.begin(), l.str_values.end());
    l.str_values.erase(std::unique(l.str_values.begin(), l.str_values.end()), l.str_values.end());
    return l;
  }
  switch (n.op) {
    case MI_F_EQ: closed(c, c); break;
    case MI_F_NE: closed(c, c); l.negate = true; break;
    case MI_F_LT: if (c == kMin) closed(1, 0); else { l.hi = c - 1; l.hi_open = false; } break;   // nothing is < MIN: an empty range
    case MI_F_LE: l.hi = c; l.hi_open = false; break;
    case MI_F_GT: if (c == kMax) closed(1, 0); else { l.lo = c + 1; l.lo_open = false; } break;
    case MI_F_GE: l.lo = c; l.lo_open = false; break;
    case MI_F_IS_NULL: l.op = device::kLeafIsNull; break;
    case MI_F_IS_NOT_NULL: l.op = device::kLeafIsNotNull; break;
    case MI_F_IN:
      if (n.n_values < 0 || (n.n_values > 0 && !n.values)) throw InvalidInputException("IN filter without values");
      if (n.n_values > 256) throw NotImplementedException("IN-list with more than 256 values is not pushed down");
      l.in_values.assign(n.values, n.values + n.n_values);
      std::sort(l.in_values.begin(), l.in_values.end());
      l.in_values.erase(std::unique(l.in_values.begin(), l.in_values.end()), l.in_values.end());
      if (l.in_values.empty()) closed(1, 0);                                  // IN () keeps nothing
      else if (l.in_values.size() == 1) { closed(l.in_values[0], l.in_values[0]); l.in_values.clear(); }
      else l.op = device::kLeafIn;
      break;
    default: throw InvalidInputException("unknown filter op " + std::to_string(n.op));
  }
  return l;
}

size_t LeafCount(const FilterCnf& cnf) {
  size_t n = 0;
  for (auto& c : cnf) n += c.size();
  return n;
}

FilterCnf ToCnf(const mi_filter_node* nodes, int32_t n_nodes, int32_t at, int depth) {
  if (at < 0 || at >= n_nodes) throw InvalidInputException("filter node index out of range");
  if (depth > 32) throw InvalidInputException("filter tree too deep");
  const mi_filter_node& n = nodes[at];
  if (n.op != MI_F_AND && n.op != MI_F_OR) return FilterCnf{{LeafOf(n)}};
  if (n.n_children <= 0 || n.first_child < 0 || n.first_child > n_nodes - n.n_children) throw InvalidInputException("AND / OR filter node without children");
  FilterCnf out;
  if (n.op == MI_F_AND) {
    for (int32_t k = 0; k < n.n_children; k++) {
      FilterCnf c = ToCnf(nodes, n_nodes, n.first_child + k, depth + 1);
      out.insert(out.end(), c.begin(), c.end());
    }
  } else {
    // (A1 & A2) | (B1 & B2) = (A1|B1) & (A1|B2) & (A2|B1) & (A2|B2): distribute child by child
    out = FilterCnf{{}};
    for (int32_t k = 0; k < n.n_children; k++) {
      FilterCnf c = ToCnf(nodes, n_nodes, n.first_child + k, depth + 1);
      FilterCnf next;
      for (auto& left : out)
        for (auto& right : c) {
          next.push_back(left);
          next.back().insert(next.back().end(), right.begin(), right.end());
          if (LeafCount(next) > 4 * kMaxLeaves) throw NotImplementedException("filter is too complex to push into the scan");
        }
      out.swap(next);
    }
  }
  return out;
}
}  // namespace

FilterCnf NormaliseFilter(const mi_filter_node* nodes, int32_t n_nodes, int32_t root) {
  if (!nodes || n_nodes <= 0) throw InvalidInputException("empty filter");
  FilterCnf cnf = ToCnf(nodes, n_nodes, root, 0);
  // single-leaf range clauses on one column intersect (lo <= v AND v < hi -> one leaf)
  FilterCnf merged;
  for (auto& clause : cnf) {
    bool folded = false;
    if (clause.size() == 1 && clause[0].op == device::kLeafRange && !clause[0].negate) {
      for (auto& m : merged) {
        if (m.size() == 1 && m[0].op == device::kLeafRange && !m[0].negate && m[0].column == clause[0].column) {
          if (!clause[0].lo_open) { m[0].lo = m[0].lo_open ? clause[0].lo : std::max(m[0].lo, clause[0].lo); m[0].lo_open = false; }
          if (!clause[0].hi_open) { m[0].hi = m[0].hi_open ? clause[0].hi : std::min(m[0].hi, clause[0].hi); m[0].hi_open = false; }
          folded = true;
          break;
        }
      }
    }
    if (!folded) merged.push_back(std::move(clause));
  }
  if (LeafCount(merged) > kMaxLeaves)
    throw NotImplementedException("filter needs " + std::to_string(LeafCount(merged)) + " leaves in conjunctive normal form, at most " +
                                  std::to_string(kMaxLeaves) + " are pushed into the scan");
  return merged;
}

}  // namespace miarrow
