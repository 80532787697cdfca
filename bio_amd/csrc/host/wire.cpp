// wire.cpp -- the record-level wire formats of pileup/snp that are specified inside the
// reference tree (the recordio container around them is not: it lives in grailbio/base):
//   MarshalPileupRow / unmarshalPileupRow    pileup/snp/row.go:113-227
//   marshalBaseStrand / UnmarshalBaseStrand  pileup/snp/basestrand.go:103-135, :183-201
//   baseStrandsRioTrailer / parse...Trailer  pileup/snp/basestrand.go:77-101
//   WriteBaseStrandToTSV, WriteBaseStrandTsv pileup/snp/basestrand.go:204-231, :282-290
//   ReadBaseStrandTsv, ReadSingleStrandBaseStrandTsv, ChrId   basestrand.go:262-279, :309-367, :401-422
// All little-endian, plain host code; declared in include/bio_pileup_host.h.
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/bio_pileup_host.h"

namespace {

inline void Put32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
inline void Put16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void Put64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
inline uint32_t Get32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint32_t Get16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint64_t Get64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

constexpr uint32_t kFieldCounts = 1, kFieldPerReadA = 2, kFieldPerReadAny = 2 | 4 | 8 | 16;  // row.go:33-40
constexpr int64_t kTrailerVersion = 1;                                                       // basestrand.go:36

void AppendUint(std::string& s, uint64_t v) {
  char t[24];
  int n = 24;
  do { t[--n] = (char)('0' + v % 10); v /= 10; } while (v);
  s.append(t + n, (size_t)(24 - n));
}

// ---- the TSV readers of basestrand.go.  grailbio/base's tsv.Reader (absent from the tree) is a
// csv.Reader with Comma '\t': empty lines are skipped, a line starting with the Comment byte is
// skipped, a trailing '\r' is dropped, and integer columns go through strconv.ParseInt, whose
// error text ("strconv.ParseInt: parsing \"A\": invalid syntax") the reference's test looks for.

int SetErr(char* err, size_t errlen, const std::string& m) {
  if (err && errlen) { size_t k = m.size() < errlen - 1 ? m.size() : errlen - 1; memcpy(err, m.data(), k); err[k] = 0; }
  return -1;
}

struct TsvLines {
  const char* p;
  const char* end;
  char comment;  // 0: none
  // next non-empty, non-comment line as fields (views into the text); false at the end
  bool Next(std::vector<std::pair<const char*, size_t>>& f) {
    while (p < end) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* le = nl ? nl : end;
      const char* b = p;
      p = nl ? nl + 1 : end;
      if (le > b && le[-1] == '\r') le--;
      if (le == b) continue;
      if (comment && *b == comment) continue;
      f.clear();
      for (;;) {
        const char* t = (const char*)memchr(b, '\t', (size_t)(le - b));
        const char* fe = t ? t : le;
        f.emplace_back(b, (size_t)(fe - b));
        if (!t) break;
        b = t + 1;
      }
      return true;
    }
    return false;
  }
};

// strconv.ParseInt(s, 10, 64) with its error wording
bool ParseInt64(const char* s, size_t n, int64_t* out, std::string* why) {
  auto bad = [&](const char* what) {
    *why = "strconv.ParseInt: parsing \"" + std::string(s, n) + "\": " + what;
    return false;
  };
  size_t i = 0;
  bool neg = false;
  if (n && (s[0] == '+' || s[0] == '-')) { neg = s[0] == '-'; i = 1; }
  if (i == n) return bad("invalid syntax");
  uint64_t v = 0;
  const uint64_t lim = neg ? (uint64_t)1 << 63 : ((uint64_t)1 << 63) - 1;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return bad("invalid syntax");
    const uint64_t d = (uint64_t)(s[i] - '0');
    if (v > (lim - d) / 10) return bad("value out of range");
    v = v * 10 + d;
  }
  *out = neg ? (int64_t)(0 - v) : (int64_t)v;
  return true;
}

bool PutStr(char* dst, size_t cap, const char* s, size_t n) {
  if (n >= cap) return false;
  memcpy(dst, s, n);
  memset(dst + n, 0, cap - n);
  return true;
}

}  // namespace

extern "C" {

int64_t bio_read_basestrand_tsv(const char* text, size_t len, bio_basestrand_tsv_row* rows, int64_t cap, char* err,
                                size_t errlen) {
  // ReadBaseStrandTsv (basestrand.go:262-279): no header row is declared, Comment = '#' hides the
  // "#CHROM ..." line, columns map to BaseStrandTsvRow's fields by position
  if (!text && len) return SetErr(err, errlen, "bio_read_basestrand_tsv: null text");
  TsvLines L{text, text + len, '#'};
  std::vector<std::pair<const char*, size_t>> f;
  int64_t n = 0;
  std::string why;
  while (L.Next(f)) {
    if (f.size() != 11) return SetErr(err, errlen, "record on line " + std::to_string((long long)n + 1) + ": wrong number of fields");
    bio_basestrand_tsv_row r;
    if (!PutStr(r.chr, sizeof r.chr, f[0].first, f[0].second) || !PutStr(r.ref, sizeof r.ref, f[2].first, f[2].second))
      return SetErr(err, errlen, "bio_read_basestrand_tsv: CHROM or REF longer than the row holds");
    if (!ParseInt64(f[1].first, f[1].second, &r.pos, &why)) return SetErr(err, errlen, why);
    for (int k = 0; k < 8; k++)
      if (!ParseInt64(f[3 + k].first, f[3 + k].second, &r.counts[k], &why)) return SetErr(err, errlen, why);
    if (rows && n < cap) rows[n] = r;
    n++;
  }
  if (err && errlen) err[0] = 0;
  return n;
}

int64_t bio_read_single_strand_basestrand_tsv(const char* fwd, size_t fwd_len, const char* rev, size_t rev_len,
                                              bio_basestrand_tsv_row* rows, int64_t cap, char* err, size_t errlen) {
  // ReadSingleStrandBaseStrandTsv (basestrand.go:309-367): both files have a header row and the
  // columns of singleStrandBaseStrandTsvRow (:293-305) are found by NAME; rows are read in
  // lockstep until either file ends; CHROM/POS and REF must agree
  static const char* const kCols[11] = {"CHROM", "POS", "DEPTH", "REF", "A", "C", "G", "T", "N", "INS", "DEL"};
  struct Side {
    TsvLines L;
    int col[11];
    std::vector<std::pair<const char*, size_t>> f;
    std::string chr, ref;
    int64_t v[11];
  } S[2] = {{{fwd, fwd + fwd_len, 0}, {}, {}, {}, {}, {}}, {{rev, rev + rev_len, 0}, {}, {}, {}, {}, {}}};
  if ((!fwd && fwd_len) || (!rev && rev_len)) return SetErr(err, errlen, "bio_read_single_strand_basestrand_tsv: null text");
  bool has_header[2];
  for (int s = 0; s < 2; s++) {
    has_header[s] = S[s].L.Next(S[s].f);
    if (!has_header[s]) continue;  // an empty file: EOF at the first Read, no rows
    for (int c = 0; c < 11; c++) {
      S[s].col[c] = -1;
      for (size_t j = 0; j < S[s].f.size(); j++)
        if (S[s].f[j].second == strlen(kCols[c]) && !memcmp(S[s].f[j].first, kCols[c], S[s].f[j].second)) { S[s].col[c] = (int)j; break; }
      if (S[s].col[c] < 0) return SetErr(err, errlen, std::string("column ") + kCols[c] + " does not appear in the header");
    }
  }
  int64_t n = 0;
  std::string why;
  for (;;) {
    bool eof = false;
    for (int s = 0; s < 2 && !eof; s++) {
      Side& d = S[s];
      if (!has_header[s] || !d.L.Next(d.f)) { eof = true; break; }
      const size_t ncol = d.f.size();
      for (int c = 0; c < 11; c++) {
        if ((size_t)d.col[c] >= ncol) return SetErr(err, errlen, "record on line " + std::to_string((long long)n + 2) + ": wrong number of fields");
        const auto& x = d.f[(size_t)d.col[c]];
        if (c == 0) d.chr.assign(x.first, x.second);
        else if (c == 3) d.ref.assign(x.first, x.second);
        else if (!ParseInt64(x.first, x.second, &d.v[c], &why)) return SetErr(err, errlen, why);
      }
    }
    if (eof) break;
    if (S[0].chr != S[1].chr || S[0].v[1] != S[1].v[1])
      return SetErr(err, errlen, "expected equal chromosome position but got fwd: " + S[0].chr + " " + std::to_string((long long)S[0].v[1]) +
                                     " and rev: " + S[1].chr + " " + std::to_string((long long)S[1].v[1]));
    if (S[0].ref != S[1].ref)
      return SetErr(err, errlen, "expected equal reference alleles but got fwd: " + S[0].ref + " and rev: " + S[1].ref);
    bio_basestrand_tsv_row r;
    if (!PutStr(r.chr, sizeof r.chr, S[0].chr.data(), S[0].chr.size()) || !PutStr(r.ref, sizeof r.ref, S[0].ref.data(), S[0].ref.size()))
      return SetErr(err, errlen, "bio_read_single_strand_basestrand_tsv: CHROM or REF longer than the row holds");
    r.pos = S[0].v[1];
    for (int b = 0; b < 4; b++) { r.counts[2 * b] = S[0].v[4 + b]; r.counts[2 * b + 1] = S[1].v[4 + b]; }
    if (rows && n < cap) rows[n] = r;
    n++;
  }
  if (err && errlen) err[0] = 0;
  return n;
}

int bio_chr_id(const char* chr, int32_t* id, char* err, size_t errlen) {
  // ChrId (basestrand.go:401-422): the name behind its 3-byte "chr" prefix (whatever those bytes
  // are: the reference slices chr[3:] without looking, and panics on a shorter string)
  if (!chr || strlen(chr) < 3) return SetErr(err, errlen, "ChrId: name shorter than its \"chr\" prefix");
  const std::string t(chr + 3);
  int v = 0;
  if (t == "X") v = 23;
  else if (t == "Y") v = 24;
  else if (t == "M") v = 25;
  else {
    int64_t i = 0;
    std::string why;
    if (!ParseInt64(t.data(), t.size(), &i, &why)) {
      const bool range = why.find("out of range") != std::string::npos;
      return SetErr(err, errlen, "strconv.Atoi: parsing \"" + t + "\": " + (range ? "value out of range" : "invalid syntax"));
    }
    if (i <= 0 || i >= 23) return SetErr(err, errlen, std::to_string((long long)i) + " is not a valid chromosome id");
    v = (int)i;
  }
  if (id) *id = v;
  if (err && errlen) err[0] = 0;
  return 0;
}

uint32_t bio_pileup_row_fields(const uint32_t* vals, const uint32_t list_len[4], int32_t per_read_needed) {
  // flushToInternal (pileup.go:518-570): a zero-depth row carries no field; otherwise the
  // counts, and with perReadNeeded every non-empty list
  if (!vals || vals[0] == 0) return 0;
  uint32_t f = kFieldCounts;
  if (per_read_needed && list_len)
    for (int b = 0; b < 4; b++)
      if (list_len[b]) f |= kFieldPerReadA << b;
  return f;
}

int64_t bio_marshal_pileup_row(uint32_t fields_present, uint32_t ref_id, uint32_t pos, const uint32_t* vals,
                               const uint32_t list_len[4], const bp_per_read* const lists[4], uint8_t* out,
                               size_t cap) {
  if (!vals) return -1;
  size_t need = 16;
  if (fields_present & kFieldCounts) need += 40;
  if (fields_present & kFieldPerReadAny) {
    if (!list_len) return -1;
    for (int b = 0; b < 4; b++)
      if (fields_present & (kFieldPerReadA << b)) {
        if (list_len[b] && (!lists || !lists[b])) return -1;
        need += 4 + 6 * (size_t)list_len[b];
      }
  }
  if (!out || cap < need) return (int64_t)need;  // size query
  uint8_t* p = out;
  Put32(p, fields_present); Put32(p + 4, ref_id); Put32(p + 8, pos); Put32(p + 12, vals[0]);
  p += 16;
  if (fields_present & kFieldCounts) {
    for (int k = 0; k < 10; k++) Put32(p + 4 * k, vals[1 + k]);  // A+ A- C+ C- G+ G- T+ T- X+ X-
    p += 40;
  }
  if (fields_present & kFieldPerReadAny)
    for (int b = 0; b < 4; b++)
      if (fields_present & (kFieldPerReadA << b)) {
        Put32(p, list_len[b]);
        p += 4;
        for (uint32_t i = 0; i < list_len[b]; i++) {
          const bp_per_read& f = lists[b][i];
          Put16(p, f.dist5p); Put16(p + 2, f.fraglen); p[4] = f.qual; p[5] = f.strand;
          p += 6;
        }
      }
  return (int64_t)need;
}

int64_t bio_unmarshal_pileup_row(const uint8_t* in, size_t len, uint32_t* fields_present, uint32_t* ref_id,
                                 uint32_t* pos, uint32_t* vals, uint32_t list_len[4], bp_per_read* feats,
                                 size_t feats_cap) {
  if (!in || len < 16 || !vals) return -1;
  const uint32_t fp = Get32(in);
  if (fields_present) *fields_present = fp;
  if (ref_id) *ref_id = Get32(in + 4);
  if (pos) *pos = Get32(in + 8);
  for (int k = 0; k < BP_ROW_U32; k++) vals[k] = 0;
  vals[0] = Get32(in + 12);
  size_t o = 16;
  if (fp & kFieldCounts) {
    if (len < o + 40) return -1;
    for (int k = 0; k < 10; k++) vals[1 + k] = Get32(in + o + 4 * k);
    o += 40;
  }
  size_t nf = 0;
  for (int b = 0; b < 4; b++) {
    uint32_t n = 0;
    if (fp & (kFieldPerReadA << b)) {
      if (len < o + 4) return -1;
      n = Get32(in + o);
      o += 4;
      if (len < o + 6 * (size_t)n) return -1;
      for (uint32_t i = 0; i < n; i++, o += 6) {
        if (feats && nf + i < feats_cap) {
          bp_per_read& f = feats[nf + i];
          f.dist5p = (uint16_t)Get16(in + o); f.fraglen = (uint16_t)Get16(in + o + 2);
          f.qual = in[o + 4]; f.strand = in[o + 5];
        }
      }
      nf += n;
    }
    if (list_len) list_len[b] = n;
  }
  return (int64_t)nf;  // entries in all lists (the caller's feats holds min(nf, feats_cap))
}

void bio_marshal_basestrand(int64_t n, const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, uint8_t* out) {
  for (int64_t i = 0; i < n; i++) {
    uint8_t* p = out + 40 * i;
    Put32(p, (uint32_t)ref_id[i]);
    Put32(p + 4, (uint32_t)pos[i]);
    for (int k = 0; k < 8; k++) Put32(p + 8 + 4 * k, vals[i * BP_ROW_U32 + 1 + k]);  // A+ A- ... T+ T-
  }
}

void bio_unmarshal_basestrand(int64_t n, const uint8_t* in, int32_t* ref_id, int32_t* pos, uint32_t* counts8) {
  for (int64_t i = 0; i < n; i++) {
    const uint8_t* p = in + 40 * i;
    if (ref_id) ref_id[i] = (int32_t)Get32(p);
    if (pos) pos[i] = (int32_t)Get32(p + 4);
    if (counts8)
      for (int k = 0; k < 8; k++) counts8[8 * i + k] = Get32(p + 8 + 4 * k);
  }
}

void bio_basestrand_rio_trailer(int64_t num_piles, uint8_t out[16]) {
  Put64(out, (uint64_t)kTrailerVersion);
  Put64(out + 8, (uint64_t)num_piles);
}

int bio_parse_basestrand_rio_trailer(const uint8_t* trailer, size_t len, int64_t* num_piles, char* err, size_t errlen) {
  auto fail = [&](const std::string& m) {
    if (err && errlen) { size_t k = m.size() < errlen - 1 ? m.size() : errlen - 1; memcpy(err, m.data(), k); err[k] = 0; }
    return -1;
  };
  if (!trailer || len < 8) return fail("unexpected EOF");
  const int64_t v = (int64_t)Get64(trailer);
  if (v != kTrailerVersion) return fail("unrecognized trailer version: got " + std::to_string((long long)v) + ", want 1");
  if (len < 16) return fail("unexpected EOF");
  if (num_piles) *num_piles = (int64_t)Get64(trailer + 8);
  if (err && errlen) err[0] = 0;
  return 0;
}

int64_t bio_basestrand_piles_to_tsv(int64_t n, const int32_t* ref_id, const int32_t* pos, const uint32_t* counts8,
                                    int32_t n_ref, const char* const* ref_names, const char* ref_bases, char* out,
                                    size_t cap) {
  // ref_bases == NULL: WriteBaseStrandToTSV (no REF column, no '#', POS as stored);
  // else WriteBaseStrandTsv over BaseStrandTsvRow{Chr, Pos, Ref, ...} with ref_bases[i] as REF
  std::string s = ref_bases ? "#CHROM\tPOS\tREF\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\n" : "CHROM\tPOS\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\n";
  for (int64_t i = 0; i < n; i++) {
    if (ref_id[i] < 0 || ref_id[i] >= n_ref || !ref_names[ref_id[i]]) return -1;
    s += ref_names[ref_id[i]];
    s.push_back('\t');
    AppendUint(s, (uint32_t)pos[i]);
    s.push_back('\t');
    if (ref_bases) { s.push_back(ref_bases[i]); s.push_back('\t'); }
    for (int k = 0; k < 8; k++) {
      AppendUint(s, counts8[8 * i + k]);
      s.push_back(k == 7 ? '\n' : '\t');
    }
  }
  if (out && cap >= s.size()) memcpy(out, s.data(), s.size());
  return (int64_t)s.size();
}

}  // extern "C"
