// output.cpp -- final-format writers (reference: pileup/snp/output.go).  The reference renders
// PileupRow temp files; here rows come straight from the device in (ref,pos) order.
//
// tsv.Writer (grailbio/base, not in the reference tree) is restated from its observable
// behaviour in cmd/bio-pileup/testdata/*.expected: every Write* appends the field and a tab,
// EndLine turns the trailing tab into a newline.
#include <dlfcn.h>
#include <stdlib.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "../par.h"
#include "bio_host.h"

namespace bio {

namespace {

const char kSeq8ToASCII[16] = {'=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};  // common.go:72
const uint8_t kSeq8ToEnum[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};                                 // common.go:66
const char kEnumToASCII[5] = {'A', 'C', 'G', 'T', 'N'};                                                           // common.go:69
const char kStrandToASCII[3] = {'.', '+', '-'};                                                                   // common.go:91
constexpr int kPerReadMask = kColEndDists | kColQuals | kColFraglens | kColStrands;                               // pileup.go:146

// libdeflate for the BGZF members when the system has its runtime library (see bam.cpp: looked
// up with dlopen, zlib otherwise, BIO_PILEUP_ZLIB=1 forces zlib).
struct DeflateOut {
  void* (*alloc)(int) = nullptr;
  size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
  void (*free_)(void*) = nullptr;
  uint32_t (*crc32_)(uint32_t, const void*, size_t) = nullptr;
  bool ok = false;
  DeflateOut() {
    if (const char* e = getenv("BIO_PILEUP_ZLIB"); e && *e && *e != '0') return;
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
    compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
    free_ = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
    crc32_ = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    ok = alloc && compress && free_ && crc32_;
  }
};
const DeflateOut& GetDeflateOut() {
  static const DeflateOut d;
  return d;
}

// Byte sink: plain file, or BGZF (public SAM-spec container: <=64 KiB gzip members with a
// BC extra subfield and an empty EOF member), compressed a batch of blocks at a time.
class Sink {
 public:
  explicit Sink(bool bgzf, int level = Z_DEFAULT_COMPRESSION, int threads = 4)
      : bgzf_(bgzf), level_(level), threads_(threads < 1 ? 1 : threads) {}
  ~Sink() {
    StopThread();
    if (f_) fclose(f_);
  }
  std::string Open(const std::string& path) {
    path_ = path;
    f_ = fopen(path.c_str(), "wb");
    if (!f_) return "open " + path + ": cannot create";
    setvbuf(f_, nullptr, _IOFBF, 4 << 20);
    th_ = std::thread([this] { Run(); });
    return "";
  }
  // Text is handed to a thread of the sink's own (file writes and BGZF compression then
  // overlap the rendering of the next rows); at most kMaxQueued bytes wait.  An error of an
  // earlier write comes back from a later call or from Close.
  std::string Write(TextString&& text) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return queued_ < kMaxQueued || !err_.empty(); });
    if (!err_.empty()) return err_;
    queued_ += text.size();
    q_.push_back(std::move(text));
    cv_.notify_all();
    return "";
  }
  std::string Write(const char* p, size_t n) { return Write(TextString(p, n)); }
  // Complete BGZF members (compressed elsewhere: on the device) go to the file as they are, behind
  // whatever text is still waiting to be compressed here.
  std::string WriteMembers(TextString&& members) {
    if (!bgzf_) return "write " + path_ + ": BGZF members for a plain file";
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return queued_ < kMaxQueued || !err_.empty(); });
    if (!err_.empty()) return err_;
    queued_ += members.size();
    q_.push_back(std::move(members));
    raw_.push_back(q_.size() - 1 + popped_);
    cv_.notify_all();
    return "";
  }
  TextString Spare() {
    std::unique_lock<std::mutex> lk(mu_);
    if (spare_.empty()) return TextString();
    TextString s = std::move(spare_.back());
    spare_.pop_back();
    return s;
  }
  std::string Close() {
    StopThread();
    std::string err = err_;
    if (bgzf_ && f_) {
      if (err.empty()) err = Compress(true);
      static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                      0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (err.empty() && fwrite(eof, 1, 28, f_) != 28) err = "write " + path_ + ": short write";
    }
    if (f_ && fclose(f_) != 0 && err.empty()) err = "close " + path_ + ": failed";
    f_ = nullptr;
    return err;
  }

 private:
  static constexpr size_t kMaxQueued = 256u << 20;
  void StopThread() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_ = true;
      cv_.notify_all();
    }
    if (th_.joinable()) th_.join();
  }
  void Run() {
    for (;;) {
      TextString text;
      bool raw = false;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        text = std::move(q_.front());
        q_.pop_front();
        if (!raw_.empty() && raw_.front() == popped_) {
          raw = true;
          raw_.pop_front();
        }
        popped_++;
      }
      std::string e = raw ? WriteMembersSync(text.data(), text.size()) : WriteSync(text.data(), text.size());
      std::unique_lock<std::mutex> lk(mu_);
      queued_ -= text.size();
      if (spare_.size() < 4 && text.capacity() >= (1u << 20)) {  // large buffers go round again
        // (with their old contents: whoever takes one sets its size, and growing a string fills it -
        // 34 MB of zeros per chunk of rows were a third of a closer's time)
        spare_.push_back(std::move(text));
      }
      if (err_.empty() && !e.empty()) err_ = e;
      cv_.notify_all();
    }
  }
  std::string WriteSync(const char* p, size_t n) {
    if (!bgzf_) {
      // Plain text goes to the file with pwrite at its known offset.  BIO_PILEUP_WRITE_THREADS > 1
      // cuts large pieces over that many threads; the default is one: buffered writes to one file
      // are serialised by its inode lock (tools/micro/pwrite_bench.c: 8-11 GB/s on the GPU box's
      // ext4 from 1 thread or 16), and on chr1 30x (8.5 GB of text) one thread measured 1.59-1.66 s
      // wall for the whole run against 1.76-2.01 s with 8, ten runs on one box.
      const int fd = fileno(f_);
      auto put = [&](const char* q, size_t m, off_t at) {
        while (m) {
          const ssize_t w = pwrite(fd, q, m, at);
          if (w <= 0) return false;
          q += w; m -= (size_t)w; at += w;
        }
        return true;
      };
      bool ok = true;
      const size_t kSlice = 4u << 20;
      static const size_t kMaxT = [] { const char* e = getenv("BIO_PILEUP_WRITE_THREADS"); const long v = e && *e ? atol(e) : 0; return (size_t)(v >= 1 && v <= 64 ? v : 1); }();
      const size_t nt = std::min<size_t>({(size_t)std::max(1, threads_), kMaxT, n / kSlice});
      if (nt <= 1) {
        ok = put(p, n, (off_t)off_);
      } else {
        std::atomic<int> bad(0);
        bp_par::parallel_for(nt, (int)nt, [&](size_t t) {
          const size_t a = n * t / nt, b = n * (t + 1) / nt;
          if (!put(p + a, b - a, (off_t)(off_ + a))) bad = 1;
        });
        ok = !bad;
      }
      off_ += n;
      if (!ok) return "write " + path_ + ": short write";
      return "";
    }
    pend_.append(p, n);
    if (pend_.size() >= kBlock * kBatch * (size_t)threads_) return Compress(false);
    return "";
  }
  std::string WriteMembersSync(const char* p, size_t n) {
    if (!pend_.empty()) {
      std::string err = Compress(true);
      if (!err.empty()) return err;
    }
    if (fwrite(p, 1, n, f_) != n) return "write " + path_ + ": short write";
    return "";
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<TextString> q_;
  std::deque<size_t> raw_;  // which of the queued items (counted from the first ever queued) are BGZF members
  size_t popped_ = 0;
  std::vector<TextString> spare_;
  size_t queued_ = 0;
  bool done_ = false;
  std::string err_;
  std::thread th_;

 private:
  static constexpr size_t kBlock = 0xff00;
  static constexpr size_t kBatch = 8;  // blocks per thread per batch

  std::string Compress(bool all) {
    size_t nblk = all ? (pend_.size() + kBlock - 1) / kBlock : pend_.size() / kBlock;
    if (nblk == 0) return "";
    std::vector<std::string> out(nblk);
    std::vector<int> bad(nblk, 0);
    const DeflateOut& ld = GetDeflateOut();
    auto work = [&](size_t t, size_t nt) {
      void* comp = ld.ok ? ld.alloc(level_ == Z_DEFAULT_COMPRESSION ? 6 : level_) : nullptr;
      struct FreeComp { const DeflateOut& d; void* p; ~FreeComp() { if (p) d.free_(p); } } free_comp{ld, comp};
      for (size_t b = t; b < nblk; b += nt) {
        const size_t off = b * kBlock, len = std::min(kBlock, pend_.size() - off);
        std::string& o = out[b];
        // (64 bytes beyond zlib's bound: libdeflate wants its own, 9 bytes of end padding among them, and
        // gives up on a short incompressible block - a header line by itself - without them)
        o.resize(18 + compressBound((uLong)len) + 64 + 8);
        size_t clen = 0;
        if (comp) {
          clen = ld.compress(comp, pend_.data() + off, len, &o[18], o.size() - 18 - 8);
          if (clen == 0) { bad[b] = 1; continue; }
        } else {
          z_stream zs;
          memset(&zs, 0, sizeof(zs));
          if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad[b] = 1; continue; }
          zs.next_in = (Bytef*)pend_.data() + off;
          zs.avail_in = (uInt)len;
          zs.next_out = (Bytef*)o.data() + 18;
          zs.avail_out = (uInt)(o.size() - 18);
          int rc = deflate(&zs, Z_FINISH);
          clen = zs.total_out;
          deflateEnd(&zs);
          if (rc != Z_STREAM_END) { bad[b] = 1; continue; }
        }
        if (18 + clen + 8 > 65536) { bad[b] = 1; continue; }
        uint8_t* h = (uint8_t*)o.data();
        const uint8_t hd[12] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0};
        memcpy(h, hd, 12);
        h[12] = 'B'; h[13] = 'C'; h[14] = 2; h[15] = 0;
        const size_t bsize = 18 + clen + 8 - 1;
        h[16] = (uint8_t)(bsize & 0xff); h[17] = (uint8_t)(bsize >> 8);
        uint32_t crc = comp ? ld.crc32_(0, pend_.data() + off, len)
                            : (uint32_t)crc32(crc32(0, nullptr, 0), (const Bytef*)pend_.data() + off, (uInt)len);
        uint8_t* tl = h + 18 + clen;
        for (int i = 0; i < 4; i++) tl[i] = (uint8_t)(crc >> (8 * i));
        for (int i = 0; i < 4; i++) tl[4 + i] = (uint8_t)((uint32_t)len >> (8 * i));
        o.resize(18 + clen + 8);
      }
    };
    size_t nt = std::min<size_t>((size_t)threads_, nblk);
    if (nt == 1) {
      work(0, 1);
    } else {
      std::vector<std::thread> th;
      for (size_t t = 0; t < nt; t++) th.emplace_back(work, t, nt);
      for (auto& x : th) x.join();
    }
    for (size_t b = 0; b < nblk; b++) {
      if (bad[b]) return "bgzf: deflate failed";
      if (fwrite(out[b].data(), 1, out[b].size(), f_) != out[b].size()) return "write " + path_ + ": short write";
    }
    pend_.erase(0, std::min(pend_.size(), nblk * kBlock));
    return "";
  }

  bool bgzf_;
  int level_, threads_;
  FILE* f_ = nullptr;
  uint64_t off_ = 0;  // plain mode: the file offset of the next write
  std::string path_, pend_;
};

// tsv.Writer restatement: the formatting half works on a string, so that rows can be rendered
// on several threads and appended to the file in order.
class TsvBuf {
 public:
  void WriteString(const char* s, size_t n) { buf_.append(s, n); buf_.push_back('\t'); }
  void WriteString(const std::string& s) { WriteString(s.data(), s.size()); }
  void WriteByte(char c) { buf_.push_back(c); buf_.push_back('\t'); }
  void WriteUint32(uint32_t v) { AppendUint(v); buf_.push_back('\t'); }
  void WritePartial(const char* s, size_t n) { buf_.append(s, n); }
  void WriteCsvUint32(uint32_t v) { AppendUint(v); buf_.push_back(','); }
  void WriteCsvByte(char c) { buf_.push_back(c); buf_.push_back(','); }
  void EndCsv() { if (!buf_.empty() && buf_.back() == ',') buf_.back() = '\t'; }
  void EndLine() { if (!buf_.empty() && buf_.back() == '\t') buf_.back() = '\n'; else buf_.push_back('\n'); }
  std::string& str() { return buf_; }

 private:
  void AppendUint(uint32_t v) {
    char t[12];
    int n = 12;
    do { t[--n] = (char)('0' + v % 10); v /= 10; } while (v);
    buf_.append(t + n, (size_t)(12 - n));
  }
  std::string buf_;
};

// One output file: header lines go through line(), rendered row text through Append().
class Tsv {
 public:
  Tsv(bool bgzf) : sink_(bgzf, Z_DEFAULT_COMPRESSION, HostThreads()) {}
  std::string Open(const std::string& path) { return sink_.Open(path); }
  TsvBuf& line() { return line_; }
  std::string EndLine() {
    line_.EndLine();
    std::string err = sink_.Write(line_.str().data(), line_.str().size());
    line_.str().clear();
    return err;
  }
  std::string Append(TextString&& text) { return sink_.Write(std::move(text)); }
  std::string Append(std::string&& text) { return sink_.Write(text.data(), text.size()); }
  std::string AppendMembers(TextString&& members) { return sink_.WriteMembers(std::move(members)); }
  TextString Spare() { return sink_.Spare(); }
  std::string Close() { return sink_.Close(); }

 private:
  Sink sink_;
  TsvBuf line_;
};

// Renders rows [0, n) in tasks of kRenderRows on HostThreads() threads; `render(first, last,
// task_slot)` fills the slot's buffers, `emit(task_slot)` appends them to the files in order.
constexpr int64_t kRenderRows = 16384;
template <class Render, class Emit>
std::string RenderParallel(int64_t n, int n_slots, Render render, Emit emit) {
  const int64_t n_tasks = (n + kRenderRows - 1) / kRenderRows;
  for (int64_t t0 = 0; t0 < n_tasks; t0 += n_slots) {
    const int m = (int)std::min<int64_t>(n_slots, n_tasks - t0);
    std::vector<std::string> errs((size_t)m);
    auto work = [&](int k) {
      const int64_t first = (t0 + k) * kRenderRows;
      errs[(size_t)k] = render(first, std::min(n, first + kRenderRows), k);
    };
    if (m == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (int k = 0; k < m; k++) th.emplace_back(work, k);
      for (auto& x : th) x.join();
    }
    for (int k = 0; k < m; k++) {
      if (!errs[(size_t)k].empty()) return errs[(size_t)k];
      std::string e = emit(k);
      if (!e.empty()) return e;
    }
  }
  return "";
}

struct RefLookup {
  const std::vector<std::string>& names;
  const std::vector<const std::string*>& seqs;
  // Seq8 of (ref,pos); the reference indexes refSeqs[refID][pos] directly and would panic on a
  // contig the .fa lacks (output.go:158), here that is an error.
  std::string Get(int32_t ref, int32_t pos, uint8_t* out) const {
    if (ref < 0 || (size_t)ref >= seqs.size() || !seqs[(size_t)ref] || (size_t)pos >= seqs[(size_t)ref]->size())
      return "pileup output: no reference base for " +
             (ref >= 0 && (size_t)ref < names.size() ? names[(size_t)ref] : std::string("?")) + ":" +
             std::to_string((long long)pos + 1) + " (contig missing from .fa, or position past its end)";
    *out = (uint8_t)(*seqs[(size_t)ref])[(size_t)pos];
    return "";
  }
};

class TsvWriter : public RowWriter {
 public:
  TsvWriter(int cols, bool bgz, const std::vector<std::string>& names, const std::vector<const std::string*>& seqs)
      : cols_(cols), ref_(bgz), alt_(bgz), lk_{names, seqs} {}
  std::string Open(const std::string& main_path, bool bgz) {
    std::string err = ref_.Open(main_path + ".ref.tsv" + (bgz ? ".gz" : ""));
    if (!err.empty()) return err;
    err = alt_.Open(main_path + ".alt.tsv" + (bgz ? ".gz" : ""));
    if (!err.empty()) return err;
    ref_.line().WriteString("#CHROM\tPOS\tREF");
    alt_.line().WriteString("#CHROM\tPOS\tREF\tALT");
    if (cols_ & kColDpRef) ref_.line().WriteString("DP");
    if (cols_ & kColDpAlt) alt_.line().WriteString("DP");
    if (cols_ & kColEndDists) {  // output.go:93-114
      ref_.line().WriteString("5P_DISTS\t3P_DISTS"); alt_.line().WriteString("5P_DISTS\t3P_DISTS");
      empty_ += ".\t.\t";
    }
    if (cols_ & kColQuals) { ref_.line().WriteString("QUALS"); alt_.line().WriteString("QUALS"); empty_ += ".\t"; }
    if (cols_ & kColFraglens) { ref_.line().WriteString("FRAGLENS"); alt_.line().WriteString("FRAGLENS"); empty_ += ".\t"; }
    if (cols_ & kColStrands) { ref_.line().WriteString("STRANDS"); alt_.line().WriteString("STRANDS"); empty_ += ".\t"; }
    if (cols_ & kColHighQ) { ref_.line().WriteString("ref_depth_tier1"); alt_.line().WriteString("alt_depth_tier1"); }
    if (cols_ & kColLowQ) { ref_.line().WriteString("ref_depth_tier2"); alt_.line().WriteString("alt_depth_tier2"); }
    err = ref_.EndLine();
    if (!err.empty()) return err;
    return alt_.EndLine();
  }
  // per-read columns of one feature list (output.go:173-200, :235-262)
  void WriteLists(TsvBuf& w, const bp_per_read* f, uint64_t m) const {
    if (cols_ & kColEndDists) {
      for (uint64_t k = 0; k < m; k++) w.WriteCsvUint32(f[k].dist5p);
      w.EndCsv();
      for (uint64_t k = 0; k < m; k++) w.WriteCsvUint32((uint16_t)(f[k].fraglen - 1 - f[k].dist5p));
      w.EndCsv();
    }
    if (cols_ & kColQuals) {
      for (uint64_t k = 0; k < m; k++) w.WriteCsvUint32(f[k].qual);
      w.EndCsv();
    }
    if (cols_ & kColFraglens) {
      for (uint64_t k = 0; k < m; k++) w.WriteCsvUint32(f[k].fraglen);
      w.EndCsv();
    }
    if (cols_ & kColStrands) {
      for (uint64_t k = 0; k < m; k++) w.WriteCsvByte(kStrandToASCII[f[k].strand < 3 ? f[k].strand : 0]);
      w.EndCsv();
    }
  }
  TextString SpareText(int which) override { return which == 0 ? ref_.Spare() : alt_.Spare(); }
  std::string WriteText(int which, TextString&& text) override {
    if (cols_ & kPerReadMask) return "pileup output: per-read columns are rendered on the host";
    return which == 0 ? ref_.Append(std::move(text)) : alt_.Append(std::move(text));
  }
  std::string WriteBgzf(int which, TextString&& members) override {
    if (cols_ & kPerReadMask) return "pileup output: per-read columns are rendered on the host";
    return which == 0 ? ref_.AppendMembers(std::move(members)) : alt_.AppendMembers(std::move(members));
  }
  std::string Write(const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, int64_t n,
                    const uint64_t* off, const bp_per_read* feats) override {
    const bool per_read = (cols_ & kPerReadMask) != 0;
    if (per_read && (!off || (!feats && off[4 * n] != off[0]))) return "pileup output: per-read columns without feature lists";
    const int n_slots = HostThreads();
    std::vector<TsvBuf> rb((size_t)n_slots), ab((size_t)n_slots);
    return RenderParallel(
        n, n_slots,
        [&](int64_t first, int64_t last, int k) {
          rb[(size_t)k].str().clear();
          ab[(size_t)k].str().clear();
          rb[(size_t)k].str().reserve((size_t)(last - first) * 40);
          return Render(ref_id, pos, vals, first, last, off, feats, rb[(size_t)k], ab[(size_t)k]);
        },
        [&](int k) {
          std::string e = ref_.Append(std::move(rb[(size_t)k].str()));
          return e.empty() ? alt_.Append(std::move(ab[(size_t)k].str())) : e;
        });
  }
  std::string Render(const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, int64_t first, int64_t last,
                     const uint64_t* off, const bp_per_read* feats, TsvBuf& ref_, TsvBuf& alt_) const {
    std::string err;
    const bool per_read = (cols_ & kPerReadMask) != 0;
    for (int64_t i = first; i < last; i++) {
      const uint32_t* v = vals + i * BP_ROW_U32;
      uint8_t b8;
      err = lk_.Get(ref_id[i], pos[i], &b8);
      if (!err.empty()) return err;
      const std::string& name = lk_.names[(size_t)ref_id[i]];
      const char ref_char = kSeq8ToASCII[b8 & 15];
      const int ref_base = kSeq8ToEnum[b8 & 15];
      ref_.WriteString(name);
      ref_.WriteUint32((uint32_t)pos[i] + 1);
      ref_.WriteByte(ref_char);
      if (cols_ & kColDpRef) ref_.WriteUint32(v[0]);
      if (per_read) {
        const uint64_t m = ref_base == 4 ? 0 : off[4 * i + ref_base + 1] - off[4 * i + ref_base];
        if (m == 0) ref_.WritePartial(empty_.data(), empty_.size());
        else WriteLists(ref_, feats + (off[4 * i + ref_base] - off[0]), m);
      }
      if (cols_ & kColHighQ) ref_.WriteUint32(v[1 + 2 * ref_base] + v[2 + 2 * ref_base]);
      if (cols_ & kColLowQ) ref_.WriteByte('0');
      ref_.EndLine();
      for (int alt = 0; alt < 5; alt++) {
        if (alt == ref_base) continue;
        const uint32_t c = v[1 + 2 * alt] + v[2 + 2 * alt];
        if (!c) continue;
        alt_.WriteString(name);
        alt_.WriteUint32((uint32_t)pos[i] + 1);
        alt_.WriteByte(ref_char);
        alt_.WriteByte(kEnumToASCII[alt]);
        if (cols_ & kColDpAlt) alt_.WriteUint32(v[0]);
        if (per_read) {
          if (alt == 4) alt_.WritePartial(empty_.data(), empty_.size());
          else WriteLists(alt_, feats + (off[4 * i + alt] - off[0]), off[4 * i + alt + 1] - off[4 * i + alt]);
        }
        if (cols_ & kColHighQ) alt_.WriteUint32(c);
        if (cols_ & kColLowQ) alt_.WriteByte('0');
        alt_.EndLine();
      }
    }
    return "";
  }
  std::string Close() override {
    std::string e1 = ref_.Close(), e2 = alt_.Close();
    return e1.empty() ? e2 : e1;
  }

 private:
  int cols_;
  Tsv ref_, alt_;
  RefLookup lk_;
  std::string empty_;  // emptyPerReadStats (output.go:92-113)
};

class BasestrandTsvWriter : public RowWriter {
 public:
  BasestrandTsvWriter(int cols, bool bgz, const std::vector<std::string>& names,
                      const std::vector<const std::string*>& seqs)
      : cols_(cols), w_(bgz), lk_{names, seqs} {}
  std::string Open(const std::string& main_path, bool bgz) {
    std::string err = w_.Open(main_path + ".basestrand.tsv" + (bgz ? ".gz" : ""));
    if (!err.empty()) return err;
    w_.line().WriteString("#CHROM\tPOS\tREF\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-");
    // The per-read header pieces start with a tab AND go through WriteString (output.go:
    // 405-418), which appends one: restated literally, so the header carries an empty column
    // before each piece.  grailbio/base/tsv is not in the reference tree: parity unpinned.
    static const char* kBS[8] = {"A+", "A-", "C+", "C-", "G+", "G-", "T+", "T-"};
    auto piece = [&](const char* a, const char* b) {
      std::string h;
      for (const char* pre : {a, b}) {
        if (!pre) continue;
        for (int k = 0; k < 8; k++) h += std::string("\t") + pre + "_" + kBS[k];
      }
      w_.line().WriteString(h);
      for (const char* pre : {a, b})
        if (pre) for (int k = 0; k < 8; k++) empty_ += ".\t";
    };
    if (cols_ & kColEndDists) piece("5P_DISTS", "3P_DISTS");
    if (cols_ & kColQuals) piece("QUALS", nullptr);
    if (cols_ & kColFraglens) piece("FRAGLENS", nullptr);
    if (cols_ & kColStrands) piece("STRANDS", nullptr);
    return w_.EndLine();
  }
  // one per-read column group over the four bases (output.go:463-552): values of forward-strand
  // reads go to the + column, all others to the - column
  template <class F>
  static void WriteGroup(TsvBuf& w_, std::string& plus_, std::string& minus_, const uint64_t* off4,
                         const bp_per_read* feats, uint64_t base0, F value) {
    for (int b = 0; b < 4; b++) {
      const uint64_t m = off4[b + 1] - off4[b];
      if (m == 0) { w_.WriteString(".\t."); continue; }
      const bp_per_read* f = feats + (off4[b] - base0);
      plus_.clear(); minus_.clear();
      for (uint64_t k = 0; k < m; k++) value(f[k], f[k].strand == 1 ? plus_ : minus_);
      for (std::string* sb : {&plus_, &minus_}) {  // flushPlusAndMinusBuf (output.go:359-374)
        if (sb->empty()) { w_.WriteByte('.'); } else { w_.WritePartial(sb->data(), sb->size()); w_.EndCsv(); }
      }
    }
  }
  static void AppendUint(std::string& s, uint32_t v) { s += std::to_string(v); s.push_back(','); }
  TextString SpareText(int) override { return w_.Spare(); }
  std::string WriteText(int, TextString&& text) override {
    if (cols_ & kPerReadMask) return "pileup output: per-read columns are rendered on the host";
    return w_.Append(std::move(text));
  }
  std::string WriteBgzf(int, TextString&& members) override {
    if (cols_ & kPerReadMask) return "pileup output: per-read columns are rendered on the host";
    return w_.AppendMembers(std::move(members));
  }
  std::string Write(const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, int64_t n,
                    const uint64_t* off, const bp_per_read* feats) override {
    const bool per_read = (cols_ & kPerReadMask) != 0;
    if (per_read && (!off || (!feats && off[4 * n] != off[0]))) return "pileup output: per-read columns without feature lists";
    const int n_slots = HostThreads();
    std::vector<TsvBuf> tb((size_t)n_slots);
    return RenderParallel(
        n, n_slots,
        [&](int64_t first, int64_t last, int k) {
          tb[(size_t)k].str().clear();
          tb[(size_t)k].str().reserve((size_t)(last - first) * 48);
          return Render(ref_id, pos, vals, first, last, off, feats, tb[(size_t)k]);
        },
        [&](int k) { return w_.Append(std::move(tb[(size_t)k].str())); });
  }
  std::string Render(const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, int64_t first, int64_t last,
                     const uint64_t* off, const bp_per_read* feats, TsvBuf& w_) const {
    std::string err, plus_, minus_;
    const bool per_read = (cols_ & kPerReadMask) != 0;
    for (int64_t i = first; i < last; i++) {
      const uint32_t* v = vals + i * BP_ROW_U32;
      uint8_t b8;
      err = lk_.Get(ref_id[i], pos[i], &b8);
      if (!err.empty()) return err;
      w_.WriteString(lk_.names[(size_t)ref_id[i]]);
      w_.WriteUint32((uint32_t)pos[i] + 1);
      w_.WriteByte(kSeq8ToASCII[b8 & 15]);
      for (int k = 0; k < 8; k++) w_.WriteUint32(v[1 + k]);
      if (per_read) {
        if (v[0] == 0) {
          w_.WritePartial(empty_.data(), empty_.size());
        } else {
          const uint64_t* o4 = off + 4 * i;
          if (cols_ & kColEndDists) {
            WriteGroup(w_, plus_, minus_, o4, feats, off[0], [](const bp_per_read& f, std::string& s) { AppendUint(s, f.dist5p); });
            WriteGroup(w_, plus_, minus_, o4, feats, off[0], [](const bp_per_read& f, std::string& s) {
              AppendUint(s, (uint16_t)(f.fraglen - 1 - f.dist5p));
            });
          }
          if (cols_ & kColQuals)
            WriteGroup(w_, plus_, minus_, o4, feats, off[0], [](const bp_per_read& f, std::string& s) { AppendUint(s, f.qual); });
          if (cols_ & kColFraglens)
            WriteGroup(w_, plus_, minus_, o4, feats, off[0], [](const bp_per_read& f, std::string& s) { AppendUint(s, f.fraglen); });
          if (cols_ & kColStrands)
            WriteGroup(w_, plus_, minus_, o4, feats, off[0], [](const bp_per_read& f, std::string& s) {
              s += f.strand == 1 ? "+," : "-,";
            });
        }
      }
      w_.EndLine();
    }
    return "";
  }
  std::string Close() override { return w_.Close(); }

 private:
  int cols_;
  Tsv w_;
  RefLookup lk_;
  std::string empty_;
};

}  // namespace

static int g_host_threads = 0;
int HostThreads() {
  if (g_host_threads > 0) return g_host_threads;
  unsigned hc = std::thread::hardware_concurrency();
  return (int)std::max<unsigned>(1, std::min<unsigned>(hc ? hc : 1, 16));
}
void SetHostThreads(int n) { g_host_threads = n > 0 ? std::min(n, 256) : 0; }

// BAM writer for tools and tests (SAM specification 4.2; the record block gbam.Unmarshal
// decodes, encoding/bam/unmarshal.go:126-224): header, then the records of `b` in order.
std::string WriteBam(const std::string& path, const std::vector<std::string>& ref_names,
                     const std::vector<int32_t>& ref_lens, const bp_record_batch& b, int level, int threads) {
  Sink out(true, level, threads);
  std::string err = out.Open(path);
  if (!err.empty()) return err;
  std::string buf;
  auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) buf.push_back((char)(v >> (8 * i))); };
  auto put16 = [&](uint32_t v) { buf.push_back((char)v); buf.push_back((char)(v >> 8)); };
  std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
  for (size_t i = 0; i < ref_names.size(); i++)
    text += "@SQ\tSN:" + ref_names[i] + "\tLN:" + std::to_string(ref_lens[i]) + "\n";
  buf += "BAM\1";
  put32((uint32_t)text.size());
  buf += text;
  put32((uint32_t)ref_names.size());
  for (size_t i = 0; i < ref_names.size(); i++) {
    put32((uint32_t)ref_names[i].size() + 1);
    buf += ref_names[i];
    buf.push_back('\0');
    put32((uint32_t)ref_lens[i]);
  }
  auto reg2bin = [](int64_t beg, int64_t end) -> uint32_t {
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
  };
  char tmp[32];
  for (int64_t i = 0; i < b.n; i++) {
    const char* nm = tmp;
    size_t nl;
    if (b.name && b.name_off) {
      nm = (const char*)b.name + b.name_off[i];
      nl = (size_t)(b.name_off[i + 1] - b.name_off[i]);
    } else {
      nl = (size_t)snprintf(tmp, sizeof tmp, "r%lld", (long long)i);
    }
    if (nl > 254) return "WriteBam: read name longer than 254 bytes";
    const uint32_t nc = b.n_cigar[i], ls = b.l_seq[i];
    const uint32_t* cg = b.cigar + b.cigar_off[i];
    int64_t span = 0;
    for (uint32_t k = 0; k < nc; k++) {
      const uint32_t op = cg[k] & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cg[k] >> 4;
    }
    const uint32_t body = 32 + (uint32_t)nl + 1 + 4 * nc + (ls + 1) / 2 + ls;
    put32(body);
    put32((uint32_t)b.ref_id[i]);
    put32((uint32_t)b.pos[i]);
    buf.push_back((char)(nl + 1));
    buf.push_back((char)b.mapq[i]);
    put16(reg2bin(b.pos[i], b.pos[i] + (span > 0 ? span : 1)));
    put16(nc);
    put16(b.flags[i]);
    put32(ls);
    put32((uint32_t)b.mate_ref_id[i]);
    put32((uint32_t)b.mate_pos[i]);
    put32(0);
    buf.append(nm, nl);
    buf.push_back('\0');
    buf.append((const char*)cg, 4 * (size_t)nc);
    buf.append((const char*)b.seq + b.seq_off[i], (ls + 1) / 2);
    buf.append((const char*)b.qual + b.qual_off[i], ls);
    if (buf.size() >= (1u << 22)) {
      err = out.Write(buf.data(), buf.size());
      if (!err.empty()) return err;
      buf.clear();
    }
  }
  err = out.Write(buf.data(), buf.size());
  if (!err.empty()) return err;
  return out.Close();
}

std::unique_ptr<RowWriter> NewTsvWriter(const std::string& main_path, int col_bitset, bool bgzip,
                                        const std::vector<std::string>& ref_names,
                                        const std::vector<const std::string*>& ref_seqs, std::string* err) {
  std::unique_ptr<TsvWriter> w(new TsvWriter(col_bitset, bgzip, ref_names, ref_seqs));
  *err = w->Open(main_path, bgzip);
  if (!err->empty()) return nullptr;
  return std::unique_ptr<RowWriter>(w.release());
}

std::unique_ptr<RowWriter> NewBasestrandTsvWriter(const std::string& main_path, int col_bitset, bool bgzip,
                                                  const std::vector<std::string>& ref_names,
                                                  const std::vector<const std::string*>& ref_seqs,
                                                  std::string* err) {
  std::unique_ptr<BasestrandTsvWriter> w(new BasestrandTsvWriter(col_bitset, bgzip, ref_names, ref_seqs));
  *err = w->Open(main_path, bgzip);
  if (!err->empty()) return nullptr;
  return std::unique_ptr<RowWriter>(w.release());
}

}  // namespace bio
