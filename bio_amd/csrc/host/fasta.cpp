// fasta.cpp -- eager FASTA loader with Seq8 encoding and the BAM-header consistency check
// (reference: encoding/fasta/fasta.go:132-175, biosimd/biosimd_amd64.go:471-511,
// pileup/common.go:169-225).
#include <zlib.h>

#include <cstring>

#include "bio_host.h"

namespace bio {

int Fasta::Find(const std::string& name) const {
  // Go keeps a map keyed by name, so a repeated name resolves to the LAST sequence stored.
  for (size_t i = names.size(); i-- > 0;)
    if (names[i] == name) return (int)i;
  return -1;
}

std::string LoadFaSeq8(const std::string& path, Fasta* out) {
  gzFile f = gzopen(path.c_str(), "rb");  // compress.NewReader sniffs gzip (common.go:179)
  if (!f) return "open " + path + ": no such file or directory";
  gzbuffer(f, 4 << 20);
  uint8_t tab[256];
  memset(tab, 15, sizeof(tab));
  tab['A'] = tab['a'] = 1;
  tab['C'] = tab['c'] = 2;
  tab['G'] = tab['g'] = 4;
  tab['T'] = tab['t'] = 8;

  out->names.clear();
  out->seqs.clear();
  std::string seq_name, seq_buf, line;
  std::vector<char> buf(1 << 20);
  bool eof = false;
  while (!eof) {
    line.clear();
    for (;;) {
      if (!gzgets(f, buf.data(), (int)buf.size())) { eof = true; break; }
      line += buf.data();
      if (!line.empty() && line.back() == '\n') break;
    }
    if (eof && line.empty()) break;
    if (!line.empty() && line.back() == '\n') line.pop_back();
    if (!line.empty() && line.back() == '\r') line.pop_back();  // bufio.ScanLines drops one trailing \r
    if (line.empty()) continue;
    if (line[0] == '>') {
      if (!seq_buf.empty()) {
        if (seq_name.empty()) { gzclose(f); return "malformed FASTA file"; }
        out->names.push_back(seq_name);
        out->seqs.push_back(seq_buf);
        seq_buf.clear();
      }
      size_t sp = line.find(' ', 1);
      seq_name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
    } else {
      size_t o = seq_buf.size();
      seq_buf.resize(o + line.size());
      for (size_t i = 0; i < line.size(); i++) seq_buf[o + i] = (char)tab[(uint8_t)line[i]];
    }
  }
  gzclose(f);
  out->names.push_back(seq_name);
  out->seqs.push_back(seq_buf);
  return "";
}

std::string FaToStringSlice(const Fasta& fa, const BamHeader& h, std::vector<const std::string*>* out,
                            std::string* warnings) {
  out->assign(h.ref_names.size(), nullptr);
  int missing_from_fa = 0;
  for (size_t i = 0; i < h.ref_names.size(); i++) {
    int k = fa.Find(h.ref_names[i]);
    if (k < 0) { missing_from_fa++; continue; }
    const std::string& s = fa.seqs[(size_t)k];
    if (s.size() != (uint64_t)(uint32_t)h.ref_lens[i])
      return "pileup.FaToStringSlice: inconsistent lengths for contig " + h.ref_names[i] + " (" +
             std::to_string(h.ref_lens[i]) + " in BAM/PAM header, " + std::to_string(s.size()) + " in .fa)";
    if (s.empty()) return "start must be less than end";  // fa.Get(name, 0, 0) (fasta.go:183-185)
    (*out)[i] = &s;
  }
  if (warnings) {
    if (missing_from_fa)
      *warnings += "pileup.FaToStringSlice: warning: " + std::to_string(missing_from_fa) +
                   " reference(s) present in BAM/PAM header but missing from .fa\n";
    long missing_from_xam = (long)fa.names.size() + missing_from_fa - (long)h.ref_names.size();
    if (missing_from_xam)
      *warnings += "pileup.FaToStringSlice: warning: " + std::to_string(missing_from_xam) +
                   " reference(s) present in .fa but missing from BAM/PAM header\n";
  }
  return "";
}

std::string ParseCols(const std::string& cols, int default_bitset, int* out) {
  static const struct { const char* name; int bit; } kMap[] = {
      {"dpref", kColDpRef}, {"dpalt", kColDpAlt}, {"enddists", kColEndDists}, {"quals", kColQuals},
      {"fraglens", kColFraglens}, {"strands", kColStrands}, {"highq", kColHighQ}, {"lowq", kColLowQ}};
  auto lookup = [&](const std::string& s) {
    for (const auto& e : kMap)
      if (s == e.name) return e.bit;
    return 0;
  };
  if (cols.empty()) { *out = default_bitset; return ""; }
  std::vector<std::string> parts;
  size_t p = 0;
  for (;;) {
    size_t c = cols.find(',', p);
    parts.push_back(cols.substr(p, c == std::string::npos ? std::string::npos : c - p));
    if (c == std::string::npos) break;
    p = c + 1;
  }
  const char* mixed = "parseCols: either all terms in column set descriptor must be preceded by +/-, or none can be";
  // An empty term makes the reference index part[0] out of range (a Go panic); report it.
  for (const auto& s : parts)
    if (s.empty()) return "parseCols: empty term in column set descriptor";
  int bits = 0;
  if (parts[0][0] == '+' || parts[0][0] == '-') {
    bits = default_bitset;
    for (const auto& s : parts) {
      if (s[0] != '+' && s[0] != '-') return mixed;
      int v = lookup(s.substr(1));
      if (!v) return "parseCols: " + s.substr(1) + " not found";
      if (s[0] == '+') bits |= v; else bits &= ~v;
    }
  } else {
    for (const auto& s : parts) {
      if (s[0] == '+' || s[0] == '-') return mixed;
      int v = lookup(s);
      if (!v) return "parseCols: " + s + " not found";
      bits |= v;
    }
  }
  *out = bits;
  return "";
}

}  // namespace bio
