// bamindex.cpp -- .bai and .gbai readers (reference: encoding/bamprovider/bamprovider.go:67-100
// picks the index by suffix; .gbai layout and lookup: encoding/bam/gindex.go:35-121; .bai is the
// SAM specification's, read there by biogo/hts).  Used only to skip parts of the BAM no job
// needs; every offset handed out is at or before the first record wanted, and the reader
// filters by coordinate afterwards, as bamIterator.Scan does (bamprovider.go:417-435).
#include <zlib.h>

#include <algorithm>
#include <cstring>

#include "bio_host.h"

namespace bio {

namespace {
const uint8_t kGbaiMagic[16] = {'G', 'B', 'A', 'I', 0x01, 0xf1, 0x78, 0x5c,
                                0x7b, 0xcb, 0xc1, 0xba, 0x08, 0x23, 0xb1, 0x19};  // gindex.go:76-79

std::string ReadAll(const std::string& path, bool gz, std::vector<uint8_t>* out) {
  out->clear();
  if (gz) {
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) return "open " + path + ": no such file or directory";
    uint8_t tmp[1 << 16];
    for (;;) {
      int n = gzread(g, tmp, sizeof(tmp));
      if (n < 0) { gzclose(g); return "read " + path + ": gzip error"; }
      if (n == 0) break;
      out->insert(out->end(), tmp, tmp + n);
    }
    gzclose(g);
    return "";
  }
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "open " + path + ": no such file or directory";
  uint8_t tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) out->insert(out->end(), tmp, tmp + n);
  fclose(f);
  return "";
}

template <typename T>
bool Take(const std::vector<uint8_t>& b, size_t* p, T* v) {
  if (*p + sizeof(T) > b.size()) return false;
  memcpy(v, b.data() + *p, sizeof(T));
  *p += sizeof(T);
  return true;
}
}  // namespace

std::string BamIndex::Load(const std::string& path) {
  loaded_ = false;
  lin_.clear();
  g_.clear();
  std::vector<uint8_t> b;
  gbai_ = path.size() >= 5 && path.compare(path.size() - 5, 5, ".gbai") == 0;  // bamprovider.go:86
  std::string err = ReadAll(path, gbai_, &b);
  if (!err.empty()) return err;
  if (gbai_) {
    if (b.size() < 16 || memcmp(b.data(), kGbaiMagic, 16) != 0) return path + ": not a .gbai index (bad magic)";
    if ((b.size() - 16) % 20 != 0) return path + ": truncated .gbai entry";
    size_t p = 16;
    while (p < b.size()) {
      GEntry e;
      Take(b, &p, &e.ref);
      Take(b, &p, &e.pos);
      Take(b, &p, &e.seq);
      Take(b, &p, &e.voff);
      g_.push_back(e);
    }
    if (g_.empty()) return path + ": .gbai index has no entries";
    loaded_ = true;
    return "";
  }
  size_t p = 0;
  int32_t n_ref = 0;
  if (b.size() < 8 || memcmp(b.data(), "BAI\1", 4) != 0) return path + ": not a .bai index (bad magic)";
  p = 4;
  if (!Take(b, &p, &n_ref) || n_ref < 0) return path + ": truncated .bai index";
  lin_.resize((size_t)n_ref);
  for (int32_t r = 0; r < n_ref; r++) {
    int32_t n_bin = 0, n_intv = 0;
    if (!Take(b, &p, &n_bin) || n_bin < 0) return path + ": truncated .bai index";
    for (int32_t k = 0; k < n_bin; k++) {
      uint32_t bin;
      int32_t n_chunk = 0;
      if (!Take(b, &p, &bin) || !Take(b, &p, &n_chunk) || n_chunk < 0 || p + 16 * (size_t)n_chunk > b.size())
        return path + ": truncated .bai index";
      p += 16 * (size_t)n_chunk;
    }
    if (!Take(b, &p, &n_intv) || n_intv < 0 || p + 8 * (size_t)n_intv > b.size())
      return path + ": truncated .bai index";
    lin_[(size_t)r].resize((size_t)n_intv);
    if (n_intv) memcpy(lin_[(size_t)r].data(), b.data() + p, 8 * (size_t)n_intv);
    p += 8 * (size_t)n_intv;
  }
  loaded_ = true;
  return "";
}

bool BamIndex::Offset(int32_t ref, int32_t pos, uint64_t* voffset) const {
  if (!loaded_ || ref < 0) return false;
  if (pos < 0) pos = 0;
  if (gbai_) {
    // GIndex.RecordOffset(ref, pos, 0) (gindex.go:94-115): the entry at the target, else the
    // one before it.  Unmapped entries (ref -1) sort last.
    auto key = [](const GEntry& e) { return ((uint64_t)(uint32_t)(e.ref < 0 ? 0x7fffffff : e.ref) << 32) | (uint32_t)e.pos; };
    const uint64_t t = ((uint64_t)(uint32_t)ref << 32) | (uint32_t)pos;
    size_t x = (size_t)(std::lower_bound(g_.begin(), g_.end(), t, [&](const GEntry& e, uint64_t v) { return key(e) < v; }) -
                        g_.begin());
    if (x == g_.size()) x = g_.size() - 1;
    else if ((key(g_[x]) > t || g_[x].seq > 0) && x > 0) x--;
    *voffset = g_[x].voff;
    return true;
  }
  // Linear index: ioffset[w] is the smallest virtual offset of a record overlapping the
  // 16 kb window w.  The nearest filled window at or before the target's is at or before the
  // first record wanted; with none, the first filled window after it is that record itself.
  uint64_t best = 0;
  for (int32_t r = ref; r >= 0 && best == 0; r--) {
    if ((size_t)r >= lin_.size()) continue;
    const std::vector<uint64_t>& v = lin_[(size_t)r];
    if (v.empty()) continue;
    int64_t w = r == ref ? std::min<int64_t>((int64_t)pos >> 14, (int64_t)v.size() - 1) : (int64_t)v.size() - 1;
    for (; w >= 0; w--)
      if (v[(size_t)w]) { best = v[(size_t)w]; break; }
  }
  if (best) { *voffset = best; return true; }
  for (size_t r = (size_t)ref; r < lin_.size(); r++) {
    const std::vector<uint64_t>& v = lin_[r];
    size_t w = r == (size_t)ref ? std::min<size_t>((size_t)pos >> 14, v.size()) : 0;
    for (; w < v.size(); w++)
      if (v[w]) { *voffset = v[w]; return true; }
  }
  return false;
}

}  // namespace bio
