// capi.cpp -- extern "C" surface of the host layers (include/bio_pileup_host.h).
#include <cstring>

#include "../../../include/bio_pileup_host.h"
#include "bio_host.h"

namespace {

int Fail(const std::string& msg, char* err, size_t errlen) {
  if (err && errlen) {
    size_t n = std::min(msg.size(), errlen - 1);
    memcpy(err, msg.data(), n);
    err[n] = 0;
  }
  return -1;
}
int Ok(char* err, size_t errlen) {
  if (err && errlen) err[0] = 0;
  return 0;
}
std::string Str(const char* s) { return s ? s : ""; }

}  // namespace

static_assert(sizeof(bio_pileup_opts) == 96, "bio_pileup_opts layout is part of the ABI");

struct bio_bed {
  bio::BEDUnion u;
};

extern "C" {

void bio_pileup_default_opts(bio_pileup_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  bio::Opts d;
  o->clip = d.Clip;
  o->flag_exclude = d.FlagExclude;
  o->mapq = d.Mapq;
  o->max_read_len = d.MaxReadLen;
  o->max_read_span = d.MaxReadSpan;
  o->min_bag_depth = d.MinBagDepth;
  o->min_base_qual = d.MinBaseQual;
}

int bio_pileup_run(const char* xampath, const char* fapath, const char* format, const char* out_prefix,
                   const bio_pileup_opts* opts, char* err, size_t errlen) {
  if (!xampath || !fapath || !format || !out_prefix || !opts) return Fail("bio_pileup_run: null argument", err, errlen);
  bio::Opts o;
  o.BedPath = Str(opts->bed_path);
  o.Region = Str(opts->region);
  o.BamIndexPath = Str(opts->bam_index_path);
  o.Clip = opts->clip;
  o.Cols = Str(opts->cols);
  o.FlagExclude = opts->flag_exclude;
  o.Mapq = opts->mapq;
  o.MaxReadLen = opts->max_read_len;
  o.MaxReadSpan = opts->max_read_span;
  o.MinBagDepth = opts->min_bag_depth;
  o.MinBaseQual = opts->min_base_qual;
  o.Parallelism = opts->parallelism;
  o.PerStrand = opts->per_strand != 0;
  o.RemoveSq = opts->remove_sq != 0;
  o.Stitch = opts->stitch != 0;
  o.TempDir = Str(opts->temp_dir);
  o.Device = opts->device;
  o.Devices = opts->devices;
  std::string e = bio::Pileup(xampath, fapath, format, out_prefix, o);
  return e.empty() ? Ok(err, errlen) : Fail(e, err, errlen);
}

int bio_pileup_main(int argc, char** argv) { return bio::BioPileupMain(argc, argv); }

int bio_parse_region(const char* region, char* ref_name, size_t ref_name_len, int32_t* start0, int32_t* end,
                     char* err, size_t errlen) {
  bio::Entry e;
  std::string m = bio::ParseRegionString(Str(region), &e);
  if (!m.empty()) return Fail(m, err, errlen);
  if (ref_name && ref_name_len) {
    size_t n = std::min(e.ref_name.size(), ref_name_len - 1);
    memcpy(ref_name, e.ref_name.data(), n);
    ref_name[n] = 0;
  }
  if (start0) *start0 = e.start0;
  if (end) *end = e.end;
  return Ok(err, errlen);
}

int bio_parse_cols(const char* cols, int32_t default_bitset, int32_t* bitset, char* err, size_t errlen) {
  int b = 0;
  std::string m = bio::ParseCols(Str(cols), default_bitset, &b);
  if (!m.empty()) return Fail(m, err, errlen);
  if (bitset) *bitset = b;
  return Ok(err, errlen);
}

int bio_bed_load(const char* path, int32_t n_ref, const char* const* ref_names, const int32_t* subset,
                 bio_bed** out, char* err, size_t errlen) {
  if (!path || !out || n_ref < 0 || (n_ref && !ref_names)) return Fail("bio_bed_load: null argument", err, errlen);
  std::vector<std::string> names;
  for (int32_t i = 0; i < n_ref; i++) names.push_back(Str(ref_names[i]));
  std::unique_ptr<bio_bed> b(new bio_bed);
  std::string m = bio::BEDUnion::FromPath(path, names, &b->u);
  if (!m.empty()) return Fail(m, err, errlen);
  if (subset) b->u = b->u.Subset(subset[0], subset[1], subset[2], subset[3]);
  *out = b.release();
  return Ok(err, errlen);
}

int32_t bio_bed_endpoints(const bio_bed* b, int32_t ref_id, int32_t* endpoints, int32_t cap) {
  if (!b || ref_id < 0 || (size_t)ref_id >= b->u.id_map.size()) return -1;
  const std::vector<bio::PosType>& e = b->u.id_map[(size_t)ref_id];
  if (!b->u.present[(size_t)ref_id]) return -1;  // reference the BED never mentions (nil in Go)
  for (int32_t i = 0; i < cap && (size_t)i < e.size(); i++) endpoints[i] = e[(size_t)i];
  return (int32_t)e.size();
}

int bio_bed_intersects(const bio_bed* b, int32_t ref_id, int32_t start, int32_t limit) {
  return b && b->u.IntersectsByID(ref_id, start, limit) ? 1 : 0;
}

void bio_bed_free(bio_bed* b) { delete b; }

int64_t bio_fasta_seq8(const char* path, const char* name, uint8_t* seq8, int64_t cap, char* err, size_t errlen) {
  bio::Fasta fa;
  std::string m = bio::LoadFaSeq8(Str(path), &fa);
  if (!m.empty()) return Fail(m, err, errlen);
  int k = fa.Find(Str(name));
  if (k < 0) return Fail("sequence not found: " + Str(name), err, errlen);
  const std::string& s = fa.seqs[(size_t)k];
  for (int64_t i = 0; i < cap && (size_t)i < s.size(); i++) seq8[i] = (uint8_t)s[(size_t)i];
  Ok(err, errlen);
  return (int64_t)s.size();
}

int bio_bam_scan(const char* path, int64_t* n_records, int64_t* sum_l_seq, int64_t* sum_n_cigar, int32_t* n_ref,
                 char* err, size_t errlen) {
  bio::BamReader rd;
  std::string m = rd.Open(Str(path));
  if (!m.empty()) return Fail(m, err, errlen);
  int64_t n = 0, sl = 0, sc = 0;
  bio::BamReader::Raw r;
  int rc;
  while ((rc = rd.Next(&r)) == 1) {
    n++;
    sl += r.l_seq;
    sc += r.n_cigar;
  }
  if (rc < 0) return Fail(rd.err(), err, errlen);
  if (n_records) *n_records = n;
  if (sum_l_seq) *sum_l_seq = sl;
  if (sum_n_cigar) *sum_n_cigar = sc;
  if (n_ref) *n_ref = (int32_t)rd.header().ref_names.size();
  return Ok(err, errlen);
}

int bio_bam_scan_chunks(const char* path, int64_t* n_records, int64_t* n_chunks, uint64_t* hash_records,
                        uint64_t* hash_chunks, int32_t* sorted, int32_t* fields_ok, int64_t* n_rewalked, char* err,
                        size_t errlen) {
  auto mix = [](uint64_t h, const bio::BamReader::Raw& r) {
    const uint64_t v[3] = {((uint64_t)(uint32_t)r.ref_id << 32) | (uint32_t)r.pos, ((uint64_t)r.l_seq << 32) | r.n_cigar,
                           ((uint64_t)r.flags << 32) | r.l_name};
    for (uint64_t x : v) h = (h ^ x) * 0x100000001b3ull;
    return h;
  };
  // record by record
  uint64_t h1 = 0xcbf29ce484222325ull, h2 = h1;
  int64_t n1 = 0, n2 = 0, nc = 0, nrw = 0;
  {
    bio::BamReader rd;
    std::string m = rd.Open(Str(path));
    if (!m.empty()) return Fail(m, err, errlen);
    bio::BamReader::Raw r;
    int rc;
    while ((rc = rd.Next(&r)) == 1) {
      h1 = mix(h1, r);
      n1++;
    }
    if (rc < 0) return Fail(rd.err(), err, errlen);
  }
  // a chunk at a time (stretches walked in parallel from guessed starts, guesses checked)
  bool all_sorted = true, all_ok = true;
  {
    bio::BamReader rd;
    std::string m = rd.Open(Str(path));
    if (!m.empty()) return Fail(m, err, errlen);
    bio::BamReader::Chunk c;
    int rc;
    uint64_t last = 0;
    while ((rc = rd.NextChunk(&c)) == 1) {
      nc++;
      nrw += c.rewalked;
      bool srt = true, ok = true;
      for (size_t k = 0; k < c.off.size(); k++) {
        bio::BamReader::Raw r;
        if (!bio::BamReader::Parse(c.data + c.off[k], &r)) ok = false;
        h2 = mix(h2, r);
        n2++;
        const uint64_t key = ((uint64_t)(uint32_t)r.ref_id << 32) | (uint32_t)r.pos;
        if (k && key < last) srt = false;
        last = key;
      }
      // the facts the walk reports are the facts of the records
      if (srt != c.sorted || ok != c.fields_ok) return Fail("bio_bam_scan_chunks: chunk facts disagree with its records", err, errlen);
      all_sorted &= srt;
      all_ok &= ok;
    }
    if (rc < 0) return Fail(rd.err(), err, errlen);
  }
  if (n1 != n2) return Fail("bio_bam_scan_chunks: record counts differ", err, errlen);
  if (n_records) *n_records = n1;
  if (n_chunks) *n_chunks = nc;
  if (hash_records) *hash_records = h1;
  if (hash_chunks) *hash_chunks = h2;
  if (sorted) *sorted = all_sorted ? 1 : 0;
  if (fields_ok) *fields_ok = all_ok ? 1 : 0;
  if (n_rewalked) *n_rewalked = nrw;
  return Ok(err, errlen);
}

int bio_bam_seek_scan(const char* bam_path, const char* index_path, int32_t ref_id, int32_t pos, int32_t* found,
                      int32_t* first_ref, int32_t* first_pos, int64_t* n_skipped, int64_t* n_at_or_after, char* err,
                      size_t errlen) {
  bio::BamReader rd;
  std::string m = rd.Open(Str(bam_path));
  if (!m.empty()) return Fail(m, err, errlen);
  bio::BamIndex ix;
  m = ix.Load(Str(index_path));
  if (!m.empty()) return Fail(m, err, errlen);
  uint64_t voff = 0;
  const bool ok = ix.Offset(ref_id, pos, &voff);
  if (found) *found = ok ? 1 : 0;
  int64_t skipped = 0, after = 0;
  int32_t fr = -1, fp = -1;
  if (ok) {
    if ((voff >> 16) > rd.BlockOffset()) {
      m = rd.Seek(voff);
      if (!m.empty()) return Fail(m, err, errlen);
    }
    bio::BamReader::Raw r;
    int rc;
    bool first = true;
    while ((rc = rd.Next(&r)) == 1) {
      if (first) { fr = r.ref_id; fp = r.pos; first = false; }
      const bool before = r.ref_id >= 0 && (r.ref_id < ref_id || (r.ref_id == ref_id && r.pos < pos));
      if (before) skipped++; else after++;
    }
    if (rc < 0) return Fail(rd.err(), err, errlen);
  }
  if (first_ref) *first_ref = fr;
  if (first_pos) *first_pos = fp;
  if (n_skipped) *n_skipped = skipped;
  if (n_at_or_after) *n_at_or_after = after;
  return Ok(err, errlen);
}

int bio_pileup_write_rows(const char* format, const char* out_prefix, const char* cols, int32_t n_ref,
                          const char* const* ref_names, const uint8_t* const* ref_seq8, const int64_t* ref_seq_len,
                          int64_t n_rows, const int32_t* ref_id, const int32_t* pos, const uint32_t* vals,
                          const uint64_t* off, const bp_per_read* feats, char* err, size_t errlen) {
  if (!format || !out_prefix || n_ref < 0 || n_rows < 0 || (n_ref && (!ref_names || !ref_seq8 || !ref_seq_len)) ||
      (n_rows && (!ref_id || !pos || !vals)))
    return Fail("bio_pileup_write_rows: null argument", err, errlen);
  const std::string fmt = format;
  const bool tsv = fmt == "tsv" || fmt == "tsv-bgz", bs = fmt == "basestrand-tsv" || fmt == "basestrand-tsv-bgz";
  if (!tsv && !bs) return Fail("Pileup: unrecognized format= argument", err, errlen);
  const bool bgz = fmt == "tsv-bgz" || fmt == "basestrand-tsv-bgz";
  const int cols_default = bio::kColDpRef | bio::kColHighQ | bio::kColLowQ;  // pileup.go:1015
  int bits = cols_default;
  if (cols && *cols) {
    std::string m = bio::ParseCols(cols, cols_default, &bits);
    if (!m.empty()) return Fail(m, err, errlen);
  }
  std::vector<std::string> names, seqs((size_t)n_ref);
  std::vector<const std::string*> seq_ptrs((size_t)n_ref, nullptr);
  for (int32_t i = 0; i < n_ref; i++) {
    names.push_back(Str(ref_names[i]));
    if (ref_seq8[i]) {
      seqs[(size_t)i].assign((const char*)ref_seq8[i], (size_t)ref_seq_len[i]);
      seq_ptrs[(size_t)i] = &seqs[(size_t)i];
    }
  }
  for (int64_t i = 0; i < n_rows; i++)
    if (ref_id[i] < 0 || ref_id[i] >= n_ref) return Fail("bio_pileup_write_rows: row with reference id out of range", err, errlen);
  std::string m;
  std::unique_ptr<bio::RowWriter> w = tsv ? bio::NewTsvWriter(out_prefix, bits, bgz, names, seq_ptrs, &m)
                                          : bio::NewBasestrandTsvWriter(out_prefix, bits, bgz, names, seq_ptrs, &m);
  if (!m.empty()) return Fail(m, err, errlen);
  if (n_rows) m = w->Write(ref_id, pos, vals, n_rows, off, feats);
  const std::string c = w->Close();
  if (m.empty()) m = c;
  return m.empty() ? Ok(err, errlen) : Fail(m, err, errlen);
}

int bio_bam_write(const char* path, int32_t n_ref, const char* const* ref_names, const int32_t* ref_lens,
                  const bp_record_batch* batch, int32_t level, int32_t threads, char* err, size_t errlen) {
  if (!path || !batch || n_ref < 0) return Fail("bio_bam_write: bad argument", err, errlen);
  std::vector<std::string> names;
  std::vector<int32_t> lens;
  for (int32_t i = 0; i < n_ref; i++) {
    names.push_back(Str(ref_names[i]));
    lens.push_back(ref_lens[i]);
  }
  std::string m = bio::WriteBam(path, names, lens, *batch, level, threads);
  if (!m.empty()) return Fail(m, err, errlen);
  return Ok(err, errlen);
}

}  // extern "C"
