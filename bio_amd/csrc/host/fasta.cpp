// fasta.cpp -- eager FASTA loader with Seq8 encoding and the BAM-header consistency check
// (reference: encoding/fasta/fasta.go:132-175, biosimd/biosimd_amd64.go:471-511,
// pileup/common.go:169-225).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>

#include "../par.h"
#include "bio_host.h"

namespace bio {

int Fasta::Find(const std::string& name) const {
  // Go keeps a map keyed by name, so a repeated name resolves to the LAST sequence stored.
  for (size_t i = names.size(); i-- > 0;)
    if (names[i] == name) return (int)i;
  return -1;
}

namespace {

// The common case - an uncompressed file that starts with a header line - on all host threads: the
// file is mapped, header lines are found block by block, and every sequence is converted in
// blocks whose output offsets come from a count of the bytes that are not line ends.  (One thread
// walking a chromosome line by line was 0.25 s of a 1.8 s run, all of it in front of the first
// record read.)  Same results as the line-by-line loader below, which takes every other file:
// a '\n' and a '\r' directly in front of one (or at the end of the file) are dropped, empty lines
// with them; a header's name ends at the first space; a header without sequence lines is replaced
// by the next one.  Returns false when the file is not of this kind (nothing stored yet).
bool LoadPlainParallel(const std::string& path, const uint8_t* tab, Fasta* out) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) { close(fd); return false; }
  const size_t size = (size_t)sb.st_size;
  void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return false;
  const char* const d = (const char*)map;
  if (d[0] != '>') { munmap(map, size); return false; }  // (a gzip file starts 1f 8b; anything odd goes the slow way)
  const int nt = HostThreads();
  const size_t kBlock = 4u << 20;
  // header lines: '>' at the start of a line
  const size_t nblk = (size + kBlock - 1) / kBlock;
  std::vector<std::vector<size_t>> found(nblk);
  bp_par::parallel_for(nblk, nt, [&](size_t b) {
    const size_t lo = b * kBlock, hi = std::min(size, lo + kBlock);
    for (const char* p = d + lo; p < d + hi;) {
      const char* q = (const char*)memchr(p, '>', (size_t)(d + hi - p));
      if (!q) break;
      if (q == d || q[-1] == '\n') found[b].push_back((size_t)(q - d));
      p = q + 1;
    }
  });
  std::vector<size_t> hdr;
  for (const auto& v : found) hdr.insert(hdr.end(), v.begin(), v.end());
  auto line_end = [&](size_t i) { return i + 1 < size && d[i] == '\r' ? d[i + 1] == '\n' : (d[i] == '\r' && i + 1 == size); };
  std::string name;
  for (size_t k = 0; k < hdr.size(); k++) {
    const size_t hs = hdr[k];
    const char* nl = (const char*)memchr(d + hs, '\n', size - hs);
    size_t he = nl ? (size_t)(nl - d) : size;          // the header line is d[hs, he)
    const size_t body0 = nl ? he + 1 : size, body1 = k + 1 < hdr.size() ? hdr[k + 1] : size;
    if (he > hs && d[he - 1] == '\r') he--;
    const char* sp = (const char*)memchr(d + hs + 1, ' ', he - hs - 1);
    name.assign(d + hs + 1, sp ? (size_t)(sp - d) - hs - 1 : he - hs - 1);
    // bytes of the body that are bases: neither '\n' nor a '\r' in front of one
    const size_t nb = (body1 - body0 + kBlock - 1) / kBlock;
    std::vector<size_t> cnt(nb + 1, 0);
    bp_par::parallel_for(nb, nt, [&](size_t b) {
      const size_t lo = body0 + b * kBlock, hi = std::min(body1, lo + kBlock);
      size_t drop = 0;
      for (const char* p = d + lo; p < d + hi;) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(d + hi - p));
        if (!q) break;
        drop++;
        if (q > d + body0 && q[-1] == '\r') drop++;  // (the '\r' may lie in the block before: it is counted here)
        p = q + 1;
      }
      cnt[b + 1] = hi - lo - drop;
    });
    // ... a '\r' counted by the block after it must not count in its own; simplest exact way: fix up
    // the block ends, and the last line of a file that ends in '\r' without a newline
    for (size_t b = 0; b < nb; b++) {
      const size_t hi = std::min(body1, body0 + (b + 1) * kBlock);
      if (hi < body1 && d[hi - 1] == '\r' && d[hi] == '\n') { cnt[b + 1]--; cnt[b + 2]++; }
    }
    if (body1 == size && body1 > body0 && d[size - 1] == '\r') cnt[nb]--;
    for (size_t b = 0; b < nb; b++) cnt[b + 1] += cnt[b];
    const size_t total = cnt[nb];
    if (total == 0 && k + 1 < hdr.size()) continue;  // a header without sequence lines: the next one replaces it
    std::string seq(total, '\0');
    bp_par::parallel_for(nb, nt, [&](size_t b) {
      const size_t lo = body0 + b * kBlock, hi = std::min(body1, lo + kBlock);
      char* o = &seq[0] + cnt[b];
      for (size_t i = lo; i < hi; i++) {
        const char c = d[i];
        if (c == '\n' || (c == '\r' && line_end(i))) continue;
        *o++ = (char)tab[(uint8_t)c];
      }
    });
    out->names.push_back(name);
    out->seqs.push_back(std::move(seq));
  }
  munmap(map, size);
  return true;
}

}  // namespace

std::string LoadFaSeq8(const std::string& path, Fasta* out) {
  {
    uint8_t tab0[256];
    memset(tab0, 15, sizeof(tab0));
    tab0['A'] = tab0['a'] = 1;
    tab0['C'] = tab0['c'] = 2;
    tab0['G'] = tab0['g'] = 4;
    tab0['T'] = tab0['t'] = 8;
    out->names.clear();
    out->seqs.clear();
    if (LoadPlainParallel(path, tab0, out)) return "";
  }
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
  // The file in 8 MiB blocks, lines found with memchr: a chromosome is millions of 60-byte lines.
  std::string seq_name, seq_buf, carry;
  std::vector<char> buf(8 << 20);
  std::string bad;
  auto take_line = [&](const char* p, size_t n) {  // one line without its '\n'
    if (n && p[n - 1] == '\r') n--;  // bufio.ScanLines drops one trailing \r
    if (n == 0) return;
    if (p[0] == '>') {
      if (!seq_buf.empty()) {
        if (seq_name.empty()) { bad = "malformed FASTA file"; return; }
        out->names.push_back(seq_name);
        seq_buf.shrink_to_fit();
        out->seqs.push_back(std::move(seq_buf));
        seq_buf.clear();
      }
      const char* sp = (const char*)memchr(p + 1, ' ', n - 1);
      seq_name.assign(p + 1, sp ? (size_t)(sp - p - 1) : n - 1);
    } else {
      const size_t o = seq_buf.size();
      if (seq_buf.capacity() < o + n) seq_buf.reserve(std::max<size_t>(2 * seq_buf.capacity(), (size_t)64 << 20));
      seq_buf.resize(o + n);
      char* d = &seq_buf[o];
      for (size_t i = 0; i < n; i++) d[i] = (char)tab[(uint8_t)p[i]];
    }
  };
  for (;;) {
    const int got = gzread(f, buf.data(), (unsigned)buf.size());
    if (got < 0) { gzclose(f); return "read " + path + ": corrupt compressed data"; }
    if (got == 0) break;
    const char* p = buf.data();
    const char* const end = p + got;
    while (p < end && bad.empty()) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      if (!nl) {
        carry.append(p, (size_t)(end - p));
        break;
      }
      if (!carry.empty()) {
        carry.append(p, (size_t)(nl - p));
        take_line(carry.data(), carry.size());
        carry.clear();
      } else {
        take_line(p, (size_t)(nl - p));
      }
      p = nl + 1;
    }
    if (!bad.empty()) { gzclose(f); return bad; }
  }
  if (!carry.empty()) take_line(carry.data(), carry.size());  // last line without a newline
  gzclose(f);
  if (!bad.empty()) return bad;
  out->names.push_back(seq_name);
  seq_buf.shrink_to_fit();
  out->seqs.push_back(std::move(seq_buf));
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
