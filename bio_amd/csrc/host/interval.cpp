// interval.cpp -- BED interval unions and region strings (reference package `interval`,
// interval/bedunion.go, interval/endpoint_index.go), as far as pileup/snp uses them.
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>

#include "bio_host.h"

namespace bio {

namespace {

// strconv.Atoi on a token: optional sign, decimal digits only, fits in int.
bool Atoi(const std::string& s, long long* out) {
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') {
    neg = s[0] == '-';
    i = 1;
  }
  if (i == s.size()) return false;
  long long v = 0;
  for (; i < s.size(); i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
    if (v > (1ll << 40)) return false;
  }
  *out = neg ? -v : v;
  return true;
}

std::string AtoiErr(const std::string& tok) { return "strconv.Atoi: parsing \"" + tok + "\": invalid syntax"; }

// Shared tail of scanBEDUnion / NewBEDUnionFromEntries: add one (ref, start, end) in input
// order, merging touching/overlapping intervals and dropping empty ones
// (bedunion.go:470-526, :671-720).
struct UnionBuilder {
  std::map<std::string, std::vector<PosType>> name_map;
  std::string prev_ref;
  PosType prev_start = 0, prev_end = 0;
  std::vector<PosType> cur;
  bool started = false;
  // scanBEDUnion keeps the (-1,-1) placeholder of a contig whose first line is empty when a
  // later interval or the end of input flushes it (bedunion.go:512, :533); the Entries
  // loader does not (bedunion.go:706-708, :723-725).  The placeholder holds no positions.
  bool keep_placeholder = false;

  std::string Add(const std::string& ref, PosType start, PosType end, const char* who) {
    if (!started || prev_ref != ref) {
      if (started) {
        if (prev_end != -1) { cur.push_back(prev_start); cur.push_back(prev_end); }
        name_map[prev_ref] = cur;
      }
      started = true;
      prev_ref = ref;
      if (name_map.count(prev_ref))
        return std::string(who) + ": unsorted input (split reference " + ref + ")";
      cur.clear();
      if (end == start) {
        prev_start = -1;
        prev_end = -1;
      } else {
        prev_start = start;
        prev_end = end;
      }
      return "";
    }
    if (end == start) return "";
    if (start > prev_end) {
      if (prev_end != -1 || keep_placeholder) { cur.push_back(prev_start); cur.push_back(prev_end); }
      prev_start = start;
      prev_end = end;
    } else {
      if (start < prev_start) return std::string(who) + ": unsorted input";
      if (end > prev_end) prev_end = end;
    }
    return "";
  }
  void Finish() {
    if (started) {
      if (prev_end != -1 || keep_placeholder) { cur.push_back(prev_start); cur.push_back(prev_end); }
      name_map[prev_ref] = cur;
    }
  }
};

// nameToIDData (bedunion.go:343-364)
void NameToID(const std::map<std::string, std::vector<PosType>>& name_map,
              const std::vector<std::string>& ref_names, BEDUnion* out) {
  out->ref_names = ref_names;
  out->id_map.assign(ref_names.size(), {});
  out->present.assign(ref_names.size(), 0);
  for (size_t i = 0; i < ref_names.size(); i++) {
    auto it = name_map.find(ref_names[i]);
    if (it != name_map.end()) {
      out->id_map[i] = it->second;
      out->present[i] = 1;
    }
  }
}

}  // namespace

std::string ParseRegionString(const std::string& region, Entry* out) {
  if (region.empty()) return "interval.ParseRegionString: empty region string";
  size_t colon = region.find(':');
  if (colon == std::string::npos) {
    out->ref_name = region;
    out->start0 = 0;
    out->end = kPosTypeMax - 1;
    return "";
  }
  if (colon == 0) return "interval.ParseRegionString: empty contig ID";
  out->ref_name = region.substr(0, colon);
  std::string range = region.substr(colon + 1);
  size_t dash = range.find('-');
  long long v = 0;
  if (dash == std::string::npos) {
    if (!Atoi(range, &v) || v > kPosTypeMax) return "strconv.ParseInt: parsing \"" + range + "\": invalid syntax";
    if (v <= 0) return "interval.ParseRegionString: position " + range + " in region string out of range";
    out->start0 = (PosType)(v - 1);
    out->end = (PosType)v;
    return "";
  }
  std::string s1 = range.substr(0, dash), s2 = range.substr(dash + 1);
  long long start1 = 0, end0 = 0;
  if (!Atoi(s1, &start1)) return AtoiErr(s1);
  if (start1 <= 0) return "interval.ParseRegionString: position " + s1 + " in region string out of range";
  if (!Atoi(s2, &end0)) return AtoiErr(s2);
  if (end0 <= start1 || end0 >= kPosTypeMax) return "interval.ParseRegionString: invalid range string " + range;
  out->start0 = (PosType)(start1 - 1);
  out->end = (PosType)end0;
  return "";
}

std::string BEDUnion::FromPath(const std::string& path, const std::vector<std::string>& ref_names,
                               BEDUnion* out) {
  gzFile f = gzopen(path.c_str(), "rb");  // plain or gzip (fileio.DetermineType, bedunion.go:571-576)
  if (!f) return "open " + path + ": no such file or directory";
  gzbuffer(f, 1 << 20);
  UnionBuilder ub;
  ub.keep_placeholder = true;
  std::string line;
  std::vector<char> buf(1 << 16);
  int line_idx = 0;
  std::string err;
  bool eof = false;
  while (!eof && err.empty()) {
    line.clear();
    for (;;) {  // one full line of any length
      if (!gzgets(f, buf.data(), (int)buf.size())) { eof = true; break; }
      line += buf.data();
      if (!line.empty() && line.back() == '\n') break;
    }
    if (eof && line.empty()) break;
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    line_idx++;
    // getTokens (bedunion.go:373-409): any byte <= ' ' separates; first three tokens
    std::string tok[3];
    int nt = 0;
    size_t p = 0;
    while (nt < 3) {
      while (p < line.size() && (unsigned char)line[p] <= ' ') p++;
      if (p >= line.size()) break;
      size_t e = p;
      while (e < line.size() && (unsigned char)line[e] > ' ') e++;
      tok[nt++] = line.substr(p, e - p);
      p = e;
    }
    if (nt != 3) {
      if (nt == 0) continue;
      err = "interval.scanBEDUnion: line " + std::to_string(line_idx) + " has fewer tokens than expected";
      break;
    }
    long long ps = 0, pe = 0;
    if (!Atoi(tok[1], &ps)) { err = AtoiErr(tok[1]); break; }
    if (ps < 0) {
      err = "interval.scanBEDUnion: negative start coordinate " + tok[1] + " on line " + std::to_string(line_idx);
      break;
    }
    if (!Atoi(tok[2], &pe)) { err = AtoiErr(tok[2]); break; }
    if (pe < ps || pe >= kPosTypeMax) {
      err = "interval.scanBEDUnion: invalid coordinate pair on line " + std::to_string(line_idx);
      break;
    }
    err = ub.Add(tok[0], (PosType)ps, (PosType)pe, "interval.scanBEDUnion");
  }
  gzclose(f);
  if (!err.empty()) return err;
  ub.Finish();
  NameToID(ub.name_map, ref_names, out);
  return "";
}

std::string BEDUnion::FromEntries(const std::vector<Entry>& entries,
                                  const std::vector<std::string>& ref_names, BEDUnion* out) {
  UnionBuilder ub;
  for (const Entry& e : entries) {
    if (e.start0 < 0) return "interval.NewBEDUnionFromEntries: negative start coordinate";
    if (e.end < e.start0 || e.end >= kPosTypeMax)
      return "interval.NewBEDUnionFromEntry: invalid coordinate pair [" + std::to_string(e.start0) + ", " +
             std::to_string(e.end) + ")";
    std::string err = ub.Add(e.ref_name, e.start0, e.end, "interval.NewBEDUnionFromEntries");
    if (!err.empty()) return err;
  }
  ub.Finish();
  NameToID(ub.name_map, ref_names, out);
  return "";
}

bool BEDUnion::IntersectsByID(int ref, PosType start, PosType limit) const {
  if (ref < 0 || (size_t)ref >= id_map.size() || !present[(size_t)ref]) return false;
  const std::vector<PosType>& e = id_map[(size_t)ref];
  size_t idx = (size_t)(std::lower_bound(e.begin(), e.end(), start + 1) - e.begin());
  if (idx & 1) return true;
  return idx < e.size() && e[idx] < limit;
}

BEDUnion BEDUnion::Subset(int sref, PosType spos, int lref, PosType lpos) const {
  // Positions of the union inside [(sref,spos), (lref,lpos)).  Same position set as
  // bedunion.go:753-852 (which can additionally leave an empty [x,x) interval behind; empty
  // intervals carry no positions and are not reproduced).
  BEDUnion out;
  out.ref_names = ref_names;
  out.id_map.assign(id_map.size(), {});
  out.present.assign(id_map.size(), 0);
  for (int r = sref; r <= lref && (size_t)r < id_map.size(); r++) {
    if (!present[(size_t)r]) continue;
    const PosType lo = r == sref ? spos : 0;
    const PosType hi = r == lref ? lpos : kPosTypeMax;
    std::vector<PosType>& o = out.id_map[(size_t)r];
    const std::vector<PosType>& e = id_map[(size_t)r];
    for (size_t k = 0; k + 1 < e.size(); k += 2) {
      PosType s = std::max(e[k], lo), t = std::min(e[k + 1], hi);
      if (s < t) { o.push_back(s); o.push_back(t); }
    }
    if (!o.empty()) out.present[(size_t)r] = 1;
  }
  return out;
}

int64_t BEDUnion::NumPositions() const {
  int64_t n = 0;
  for (const auto& e : id_map)
    for (size_t k = 0; k + 1 < e.size(); k += 2) n += (int64_t)e[k + 1] - e[k];
  return n;
}

}  // namespace bio
