// bam.cpp -- sequential BGZF/BAM reader feeding bp_record_batch (reference: encoding/bam
// unmarshal.go:126-224 for the record layout, encoding/bamprovider/provider.go:141-166 for the
// iterator contract).  BGZF blocks are inflated a batch at a time on a few host threads.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <thread>

#include "bio_host.h"

namespace bio {

void RecordBatchBuilder::clear() {
  ref_id.clear(); pos.clear(); mate_ref_id.clear(); mate_pos.clear();
  flags.clear(); mapq.clear(); n_cigar.clear(); l_seq.clear();
  cigar_off.clear(); seq_off.clear(); qual_off.clear(); name_off.clear();
  cigar.clear(); seq.clear(); qual.clear(); name.clear();
}

bp_record_batch RecordBatchBuilder::view() {
  bp_record_batch b;
  memset(&b, 0, sizeof(b));
  b.n = size();
  b.ref_id = ref_id.data(); b.pos = pos.data();
  b.mate_ref_id = mate_ref_id.data(); b.mate_pos = mate_pos.data();
  b.flags = flags.data(); b.mapq = mapq.data();
  b.n_cigar = n_cigar.data(); b.l_seq = l_seq.data();
  b.cigar_off = cigar_off.data(); b.cigar = cigar.data();
  b.seq_off = seq_off.data(); b.seq = seq.data();
  b.qual_off = qual_off.data(); b.qual = qual.data();
  if (keep_names) {
    if (name_off.size() == (size_t)b.n) name_off.push_back(name.size());
    b.name_off = name_off.data();
    b.name = name.data();
  }
  return b;
}

namespace {
inline uint16_t U16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t U32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline int32_t I32(const uint8_t* p) { return (int32_t)U32(p); }

// libdeflate, when the system has its runtime library (this image ships libdeflate.so.0 without
// headers, so the four entry points used are declared here and looked up with dlopen): about
// twice zlib's inflate rate and a several times faster CRC-32, and inflating is what the input
// side of a run costs (the reference links libdeflate for BGZF too, encoding/bgzf).  Without it
// zlib does the same work.  BIO_PILEUP_ZLIB=1 forces zlib.
struct Deflate {
  void* (*alloc)() = nullptr;
  int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
  void (*free_)(void*) = nullptr;
  uint32_t (*crc32_)(uint32_t, const void*, size_t) = nullptr;
  bool ok = false;
  Deflate() {
    if (const char* e = getenv("BIO_PILEUP_ZLIB"); e && *e && *e != '0') return;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
    decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
    free_ = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
    crc32_ = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    ok = alloc && decompress && free_ && crc32_;
  }
};
const Deflate& GetDeflate() {
  static const Deflate d;
  return d;
}

constexpr int kBatchBlocks = 256;  // up to 16 MiB inflated per refill (8 and 64 MiB are slower)
// A batch's inflated bytes start kHead bytes into its buffer: the unconsumed tail of the previous
// batch (normally part of one record) is copied in front of them and the buffers are swapped,
// instead of appending 16 MiB to the old one.
constexpr size_t kHead = 1 << 20;
}  // namespace

struct BamReader::Impl {
  FILE* f = nullptr;
  bool file_eof = false;     // the batch reader met the end of the file
  std::vector<uint8_t, NoInitAlloc<uint8_t>> buf;  // inflated bytes not yet consumed: [rd, buf.size())
  size_t rd = 0;
  uint64_t file_pos = 0;  // file offset of the next byte fread returns
  double wait_s = 0;      // time the consumer waited for inflated batches
  // (offset in buf, file offset) of the blocks buf holds bytes of, ascending
  std::vector<std::pair<size_t, uint64_t>> marks;

  // One batch of BGZF blocks, read and inflated off the main thread while the records of the
  // previous batch are decoded.
  struct Batch {
    std::vector<uint8_t, NoInitAlloc<uint8_t>> data;  // (sized, then filled by the inflaters: not zeroed first)
    std::vector<std::pair<size_t, uint64_t>> marks;  // (offset in data, file offset)
    bool eof = false;
    uint64_t end_pos = 0;  // file offset behind the batch
    std::string err;
  };
  std::future<std::unique_ptr<Batch>> next;

  ~Impl() {
    if (next.valid()) next.wait();
    if (f) fclose(f);
  }

  std::unique_ptr<Batch> ReadBatch(uint64_t at) {
    std::unique_ptr<Batch> out(new Batch);
    Batch& B = *out;
    struct Blk { size_t off, clen, isize, out; };
    std::vector<Blk> blks;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> raw;
    size_t out_total = 0;
    uint64_t pos = at;
    while ((int)blks.size() < kBatchBlocks) {
      uint8_t hd[18];
      const uint64_t block_at = pos;
      size_t got = fread(hd, 1, 18, f);
      if (got == 0) { B.eof = true; break; }
      if (got != 18 || hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4)) {
        B.err = "bgzf: invalid block header";
        return out;
      }
      unsigned xlen = U16(hd + 10);
      // The BC subfield is first in every BGZF writer in use; handle the general case too.
      std::vector<uint8_t> extra(xlen);
      memcpy(extra.data(), hd + 12, std::min<size_t>(6, xlen));
      if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) { B.err = "bgzf: truncated block"; return out; }
      int bsize = -1;
      for (size_t p = 0; p + 4 <= extra.size();) {
        unsigned slen = U16(extra.data() + p + 2);
        if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= extra.size())
          bsize = U16(extra.data() + p + 4);
        p += 4 + slen;
      }
      if (bsize < 0) { B.err = "bgzf: missing BC subfield"; return out; }
      long total = (long)bsize + 1;
      long remaining = total - 12 - (long)xlen;  // deflate data + crc32 + isize
      if (xlen < 6 || remaining < 8) { B.err = "bgzf: invalid block size"; return out; }
      size_t off = raw.size();
      raw.resize(off + (size_t)remaining);
      if (fread(raw.data() + off, 1, (size_t)remaining, f) != (size_t)remaining) { B.err = "bgzf: truncated block"; return out; }
      size_t isize = U32(raw.data() + off + (size_t)remaining - 4);
      if (isize > 65536) { B.err = "bgzf: invalid uncompressed size"; return out; }
      blks.push_back({off, (size_t)remaining - 8, isize, kHead + out_total});
      if (isize) B.marks.emplace_back(kHead + out_total, block_at);
      pos = block_at + (uint64_t)total;
      out_total += isize;
    }
    B.end_pos = pos;
    B.data.resize(kHead + out_total);
    if (blks.empty()) return out;
    int nt = std::min<int>(HostThreads(), (int)blks.size());
    std::vector<int> bad((size_t)nt, 0);
    const Deflate& ld = GetDeflate();
    auto work = [&](int t) {
      z_stream zs;
      void* dec = ld.ok ? ld.alloc() : nullptr;
      struct FreeDec { const Deflate& d; void* p; ~FreeDec() { if (p) d.free_(p); } } free_dec{ld, dec};
      for (size_t i = (size_t)t; i < blks.size(); i += (size_t)nt) {
        const Blk& b = blks[i];
        if (b.isize == 0) continue;
        if (dec) {
          size_t got = 0;
          if (ld.decompress(dec, raw.data() + b.off, b.clen, B.data.data() + b.out, b.isize, &got) != 0 || got != b.isize ||
              ld.crc32_(0, B.data.data() + b.out, b.isize) != U32(raw.data() + b.off + b.clen)) {
            bad[(size_t)t] = 1;
            return;
          }
          continue;
        }
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { bad[(size_t)t] = 1; return; }
        zs.next_in = raw.data() + b.off;
        zs.avail_in = (uInt)b.clen;
        zs.next_out = B.data.data() + b.out;
        zs.avail_out = (uInt)b.isize;
        int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0) { bad[(size_t)t] = 1; return; }
        uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), B.data.data() + b.out, (uInt)b.isize);
        if (crc != U32(raw.data() + b.off + b.clen)) { bad[(size_t)t] = 1; return; }
      }
    };
    if (nt == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
    for (int b : bad)
      if (b) { B.err = "bgzf: corrupt block (inflate or CRC failure)"; break; }
    return out;
  }

  // Appends the next batch behind the unconsumed tail and starts reading the one after it.
  std::string Refill() {
    if (!file_eof && buf.size() - rd <= kHead) {
      const auto w0 = std::chrono::steady_clock::now();
      std::unique_ptr<Batch> b = next.valid() ? next.get() : ReadBatch(file_pos);
      wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      if (!b->err.empty()) return b->err;
      const size_t tail = buf.size() - rd, at = kHead - tail;
      if (tail) memcpy(b->data.data() + at, buf.data() + rd, tail);
      // the mark of the block the tail starts in, and later ones, move with it
      std::vector<std::pair<size_t, uint64_t>> nm;
      if (tail) {
        size_t k = 0;
        while (k + 1 < marks.size() && marks[k + 1].first <= rd) k++;
        for (; k < marks.size(); k++) nm.emplace_back((marks[k].first > rd ? marks[k].first - rd : 0) + at, marks[k].second);
      }
      nm.insert(nm.end(), b->marks.begin(), b->marks.end());
      marks.swap(nm);
      buf.swap(b->data);
      rd = at;
      file_pos = b->end_pos;
      file_eof = b->eof;
      if (!file_eof) {
        const uint64_t pos = file_pos;
        next = std::async(std::launch::async, [this, pos] { return ReadBatch(pos); });
      }
      return "";
    }
    if (rd > 0) {
      buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)rd);
      // keep the mark of the block the tail starts in (its offset becomes 0) and later ones
      size_t k = 0;
      while (k + 1 < marks.size() && marks[k + 1].first <= rd) k++;
      marks.erase(marks.begin(), marks.begin() + (ptrdiff_t)k);
      for (auto& mk : marks) mk.first = mk.first > rd ? mk.first - rd : 0;
      rd = 0;
    }
    if (file_eof) return "";
    std::unique_ptr<Batch> b = next.valid() ? next.get() : ReadBatch(file_pos);
    if (!b->err.empty()) return b->err;
    const size_t base = buf.size();  // (a record longer than kHead: the slow way)
    buf.insert(buf.end(), b->data.begin() + (ptrdiff_t)std::min(kHead, b->data.size()), b->data.end());
    for (const auto& mk : b->marks) marks.emplace_back(base + mk.first - kHead, mk.second);
    file_pos = b->end_pos;
    file_eof = b->eof;
    if (!file_eof) {
      const uint64_t at = file_pos;
      next = std::async(std::launch::async, [this, at] { return ReadBatch(at); });
    }
    return "";
  }

  // Makes at least n unconsumed bytes available; returns false at a clean or short EOF.
  bool Need(size_t n, std::string* err) {
    while (buf.size() - rd < n) {
      if (file_eof) return false;
      *err = Refill();
      if (!err->empty()) return false;
    }
    return true;
  }
};

BamReader::BamReader() : impl_(new Impl) {}
BamReader::~BamReader() {}

std::string BamReader::Open(const std::string& path) {
  impl_->f = fopen(path.c_str(), "rb");
  if (!impl_->f) return "open " + path + ": no such file or directory";
  setvbuf(impl_->f, nullptr, _IOFBF, 4 << 20);
  std::string err;
  Impl& m = *impl_;
  if (!m.Need(12, &err)) return err.empty() ? "bam: truncated header" : err;
  if (memcmp(m.buf.data() + m.rd, "BAM\1", 4) != 0) return "bam: magic number mismatch";
  uint32_t l_text = U32(m.buf.data() + m.rd + 4);
  if (!m.Need(12 + (size_t)l_text, &err)) return err.empty() ? "bam: truncated header" : err;
  header_.text.assign((const char*)m.buf.data() + m.rd + 8, l_text);
  while (!header_.text.empty() && header_.text.back() == '\0') header_.text.pop_back();
  uint32_t n_ref = U32(m.buf.data() + m.rd + 8 + l_text);
  m.rd += 12 + (size_t)l_text;
  for (uint32_t i = 0; i < n_ref; i++) {
    if (!m.Need(4, &err)) return err.empty() ? "bam: truncated header" : err;
    uint32_t l_name = U32(m.buf.data() + m.rd);
    if (!m.Need(8 + (size_t)l_name, &err)) return err.empty() ? "bam: truncated header" : err;
    std::string nm((const char*)m.buf.data() + m.rd + 4, l_name ? l_name - 1 : 0);
    header_.ref_names.push_back(nm);
    header_.ref_lens.push_back(I32(m.buf.data() + m.rd + 4 + l_name));
    m.rd += 8 + (size_t)l_name;
  }
  return "";
}

std::string BamReader::Seek(uint64_t voffset) {
  Impl& m = *impl_;
  const uint64_t coff = voffset >> 16;
  if (m.next.valid()) m.next.get();  // the reader thread owns the file until its batch is done
  if (fseeko(m.f, (off_t)coff, SEEK_SET) != 0) return "bgzf: seek failed";
  m.file_pos = coff;
  m.file_eof = false;
  m.buf.clear();
  m.rd = 0;
  m.marks.clear();
  std::string err = m.Refill();
  if (!err.empty()) return err;
  if ((voffset & 0xffff) > m.buf.size() - m.rd) return "bgzf: virtual offset past the end of its block";
  m.rd += (size_t)(voffset & 0xffff);
  return "";
}

uint64_t BamReader::BlockOffset() const {
  const Impl& m = *impl_;
  uint64_t at = m.file_pos;
  for (const auto& mk : m.marks) {
    if (mk.first > m.rd) break;
    at = mk.second;
  }
  return at;
}

double BamReader::WaitSeconds() const { return impl_->wait_s; }

bool BamReader::NextRefills() const {
  const Impl& m = *impl_;
  const size_t have = m.buf.size() - m.rd;
  if (have < 4) return true;
  return have < 4 + (size_t)U32(m.buf.data() + m.rd);
}

int BamReader::Next(Raw* r) {
  Impl& m = *impl_;
  std::string err;
  if (!m.Need(4, &err)) {
    if (!err.empty()) { err_ = err; return -1; }
    if (m.buf.size() != m.rd) { err_ = "bam: truncated record"; return -1; }
    return 0;
  }
  uint32_t block_size = U32(m.buf.data() + m.rd);
  if (block_size < 32) { err_ = "bam: invalid record block size"; return -1; }
  if (!m.Need(4 + (size_t)block_size, &err)) {
    err_ = err.empty() ? "bam: truncated record" : err;
    return -1;
  }
  const uint8_t* p = m.buf.data() + m.rd + 4;
  m.rd += 4 + (size_t)block_size;
  // The walk is a pointer chase (a record's length says where the next one starts) through
  // bytes another core has just inflated: pull the lines a few records ahead in now.
  for (int k = 0; k < 5; k++) __builtin_prefetch(p + 1024 + 64 * k);
  r->ref_id = I32(p);
  r->pos = I32(p + 4);
  uint32_t l_read_name = p[8];
  r->mapq = p[9];
  r->n_cigar = U16(p + 12);
  r->flags = U16(p + 14);
  r->l_seq = U32(p + 16);
  r->mate_ref_id = I32(p + 20);
  r->mate_pos = I32(p + 24);
  size_t need = 32 + (size_t)l_read_name + 4 * (size_t)r->n_cigar + ((size_t)r->l_seq + 1) / 2 + r->l_seq;
  if (need > block_size) { err_ = "bam: record fields exceed block size"; return -1; }
  r->name = (const char*)p + 32;
  r->l_name = l_read_name ? l_read_name - 1 : 0;
  r->cigar = (const uint32_t*)(p + 32 + l_read_name);  // read with memcpy by callers
  r->seq = p + 32 + l_read_name + 4 * (size_t)r->n_cigar;
  r->qual = r->seq + ((size_t)r->l_seq + 1) / 2;
  r->aux = r->qual + r->l_seq;
  r->l_aux = (uint32_t)(block_size - need);
  return 1;
}

bool AuxInt(const uint8_t* aux, uint32_t l_aux, const char tag[2], int64_t* out) {
  size_t p = 0;
  while (p + 3 <= l_aux) {
    const bool hit = aux[p] == (uint8_t)tag[0] && aux[p + 1] == (uint8_t)tag[1];
    const char ty = (char)aux[p + 2];
    p += 3;
    size_t sz = 0;
    int64_t v = 0;
    bool isint = true;
    switch (ty) {
      case 'A': case 'C': if (p + 1 > l_aux) return false; sz = 1; v = aux[p]; isint = ty == 'C'; break;
      case 'c': if (p + 1 > l_aux) return false; sz = 1; v = (int8_t)aux[p]; break;
      case 'S': if (p + 2 > l_aux) return false; sz = 2; v = U16(aux + p); break;
      case 's': if (p + 2 > l_aux) return false; sz = 2; v = (int16_t)U16(aux + p); break;
      case 'I': if (p + 4 > l_aux) return false; sz = 4; v = U32(aux + p); break;
      case 'i': if (p + 4 > l_aux) return false; sz = 4; v = I32(aux + p); break;
      case 'f': sz = 4; isint = false; break;
      case 'Z': case 'H': {
        size_t e = p;
        while (e < l_aux && aux[e]) e++;
        sz = e - p + 1;
        isint = false;
        break;
      }
      case 'B': {
        if (p + 5 > l_aux) return false;
        char st = (char)aux[p];
        size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        sz = 5 + es * (size_t)U32(aux + p + 1);
        isint = false;
        break;
      }
      default: return false;
    }
    if (hit) {
      if (!isint) return false;
      *out = v;
      return true;
    }
    p += sz;
  }
  return false;
}

}  // namespace bio
