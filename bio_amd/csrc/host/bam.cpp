// bam.cpp -- sequential BGZF/BAM reader (reference: encoding/bam unmarshal.go:126-224 for the
// record layout, encoding/bamprovider/provider.go:141-166 for the iterator contract).  BGZF blocks
// are read and inflated a batch ahead on the host threads; records are handed out one at a time
// (Next) or as chunks of whole records as they lie in the inflated stream (NextChunk), which the
// driver passes to bp_submit_bam.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <zlib.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

#include "../../../include/bio_pileup.h"
#include "../par.h"
#include "bio_host.h"

namespace bio {

constexpr size_t kTextPinFrom = 8u << 20;  // (the host renderer's pieces stay below it)
void* TextAlloc(size_t n) {
  void* p = nullptr;
  if (n >= kTextPinFrom) {
    if (hipHostMalloc(&p, n, hipHostMallocPortable) == hipSuccess) return p;
    (void)hipGetLastError();
    // (no device, or no pinned memory left: plain memory does; TextFree tells the two apart)
    p = nullptr;
  }
  p = malloc(n + (n >= kTextPinFrom ? 16 : 0));
  if (!p) throw std::bad_alloc();
  if (n >= kTextPinFrom) {
    // a large block that is NOT pinned carries a mark in front
    memcpy(p, "notpinned-block", 16);
    return (char*)p + 16;
  }
  return p;
}
void TextFree(void* p, size_t n) {
  if (!p) return;
  if (n >= kTextPinFrom) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost) {
      (void)hipHostFree(p);
      return;
    }
    (void)hipGetLastError();
    free((char*)p - 16);
    return;
  }
  free(p);
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

constexpr int kBatchBlocksMax = 256;  // up to 16 MiB inflated per refill (8 and 64 MiB are slower)
// With the blocks inflated on the device a batch is 4,096 blocks (256 MiB inflated): a block is
// one wavefront's work for a few milliseconds, and the device holds thousands of wavefronts.
constexpr int kDeviceBatchBlocks = 4096;
constexpr int kDeviceParts = 4;  // a batch goes to the device in this many parts, each on a stream of its own
int BatchBlocks(bool device = false) {  // BIO_PILEUP_BATCH_BLOCKS exists for tests of the chunk hand-over and the seeks
  const char* e = getenv("BIO_PILEUP_BATCH_BLOCKS");
  long v = e && *e ? atol(e) : 0;
  const int mx = device ? kDeviceBatchBlocks : kBatchBlocksMax;
  return v >= 1 && v <= mx ? (int)v : mx;
}
// A batch's inflated bytes start kHead bytes into its buffer: the unconsumed tail of the previous
// batch (normally part of one record) is copied in front of them, instead of appending 16 MiB to
// the old buffer.
constexpr size_t kHead = 1 << 20;
constexpr size_t kStdCap = kHead + (size_t)kBatchBlocksMax * 65536 + 64;
constexpr size_t kDeviceStdCap = kHead + (size_t)kDeviceBatchBlocks * 65536 + 64;

// Inflated bytes live in buffers of one size that go back to a pool when the last chunk that
// points into them is dropped: fresh 17 MiB allocations would be page-faulted in over and over
// (as much CPU time as inflating them), and pinned ones (hipHostMalloc, so the device can copy
// records straight out of them: bp_submit_bam) are expensive to make.
struct Buf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
};

class BufPool {
 public:
  static BufPool& Get() {
    static BufPool* g = new BufPool();  // never destroyed: chunks may outlive every reader
    return *g;
  }
  // `big`: the device-inflate batch size (its buffers are pooled apart from the 17 MiB ones)
  std::shared_ptr<Buf> Take(size_t need, bool pinned, bool big = false) {
    Buf* b = nullptr;
    const size_t std_cap = big ? kDeviceStdCap : kStdCap;
    if (need <= std_cap) {
      std::lock_guard<std::mutex> lk(mu_);
      std::vector<Buf*>& f = free_[(pinned ? 1 : 0) + (big ? 2 : 0)];
      if (!f.empty()) {
        b = f.back();
        f.pop_back();
      }
    }
    if (!b) {
      b = new Buf();
      b->cap = std::max(need, std_cap);
      // only standard buffers are worth pinning (the others hold one over-long record)
      if (pinned && need <= std_cap) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, b->cap, hipHostMallocPortable) == hipSuccess) {
          b->p = (uint8_t*)hp;
          b->pinned = true;
        } else {
          (void)hipGetLastError();
        }
      }
      if (!b->p) b->p = (uint8_t*)malloc(b->cap);
      if (!b->p) {
        delete b;
        return nullptr;
      }
    }
    return std::shared_ptr<Buf>(b, [this](Buf* x) { Give(x); });
  }

 private:
  void Give(Buf* b) {
    if (b->cap == kStdCap || b->cap == kDeviceStdCap) {
      const bool big = b->cap == kDeviceStdCap;
      std::lock_guard<std::mutex> lk(mu_);
      std::vector<Buf*>& f = free_[(b->pinned ? 1 : 0) + (big ? 2 : 0)];
      if (f.size() < (big ? 6u : 24u)) {
        f.push_back(b);
        return;
      }
    }
    if (b->pinned) (void)hipHostFree(b->p);
    else free(b->p);
    delete b;
  }
  std::mutex mu_;
  std::vector<Buf*> free_[4];
};
}  // namespace

namespace {

// What one walk leaves: the records it passed and what their fixed fields say.
struct Walk {
  size_t start = SIZE_MAX;  // where it began (SIZE_MAX: no start found)
  size_t exit = 0;          // where it stopped: the first record start it did not take
  bool stopped = false;     // ... because that record is cut off by the end of the buffer or has an invalid block size
  std::vector<uint32_t> off;
  bool fields_ok = true, sorted = true, one_ref = true;
  int32_t r0 = 0, p0 = 0, r1 = 0, p1 = 0;
};

inline uint64_t OrderKey(int32_t ref, int32_t pos) { return ((uint64_t)(uint32_t)ref << 32) | (uint32_t)pos; }

// Takes whole records from p on while they start before `stop`.
void WalkFrom(const uint8_t* base, size_t p, size_t stop, size_t end, Walk* w) {
  w->start = p;
  w->off.clear();
  w->stopped = false;
  w->fields_ok = w->sorted = w->one_ref = true;
  uint64_t last = 0;
  while (p < stop) {
    if (end - p < 4) { w->stopped = true; break; }
    uint32_t f[6];
    if (end - p >= 24) memcpy(f, base + p, 24);
    else f[0] = U32(base + p);
    const uint32_t bs = f[0];
    if (bs < 32 || end - p - 4 < bs) { w->stopped = true; break; }
    __builtin_prefetch(base + p + 2048);
    const uint64_t need = 32 + (uint64_t)(f[3] & 0xffu) + 4 * (uint64_t)(f[4] & 0xffffu) + ((uint64_t)f[5] + 1) / 2 + f[5];
    if (need > bs) w->fields_ok = false;
    const int32_t ref = (int32_t)f[1], pos = (int32_t)f[2];
    const uint64_t key = OrderKey(ref, pos);  // (unplaced records, ref -1, sort last)
    if (w->off.empty()) {
      w->r0 = ref;
      w->p0 = pos;
    } else {
      if (key < last) w->sorted = false;
      if (ref != w->r0) w->one_ref = false;
    }
    last = key;
    w->r1 = ref;
    w->p1 = pos;
    w->off.push_back((uint32_t)p);
    p += 4 + (size_t)bs;
  }
  w->exit = p;
}

// Could a record start at p?  Only ever a guess (see NextChunk): block size, reference ids,
// positions and name length in range, the fields inside the block, the name NUL-terminated.
bool Plausible(const uint8_t* base, size_t p, size_t end, int32_t n_ref) {
  if (end - p < 36) return false;
  uint32_t f[8];
  memcpy(f, base + p, 32);
  const uint32_t bs = f[0], l_name = f[3] & 0xffu, n_cigar = f[4] & 0xffffu, l_seq = f[5];
  if (bs < 32 || bs > (1u << 26)) return false;
  const int32_t ref = (int32_t)f[1], pos = (int32_t)f[2], mref = (int32_t)f[6], mpos = (int32_t)f[7];
  if (ref < -1 || ref >= n_ref || mref < -1 || mref >= n_ref || pos < -1 || mpos < -1 || l_name == 0) return false;
  const uint64_t need = 32 + (uint64_t)l_name + 4 * (uint64_t)n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (need > bs) return false;
  const size_t nul = p + 36 + l_name - 1;
  return nul >= end || base[nul] == 0;
}

}  // namespace

namespace {

// ---- resident mode: the inflated stream stays on the device ------------------------------------
struct DevStream {  // one batch's stream on the device
  uint8_t* p = nullptr;
  size_t cap = 0;
  int dev = -1;
};
struct ColStore {  // the scan's columns of one batch, pinned host memory
  bp_bam_columns c;
  void* block = nullptr;
  size_t cap = 0;
  size_t names_cap = 0;  // bytes of read names it has room for (0: none wanted, stitching off)
};
class ResidentPool {
 public:
  static ResidentPool& Get() {
    static ResidentPool* g = new ResidentPool();
    return *g;
  }
  std::shared_ptr<DevStream> TakeDev(int dev, size_t need) {
    DevStream* d = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t k = 0; k < devs_.size(); k++)
        if (devs_[k]->dev == dev && devs_[k]->cap >= need) {
          d = devs_[k];
          devs_.erase(devs_.begin() + (ptrdiff_t)k);
          break;
        }
    }
    if (!d) {
      d = new DevStream();
      d->dev = dev;
      d->cap = std::max(need, kDeviceStdCap + 4096);
      if (hipSetDevice(dev) != hipSuccess || hipMalloc((void**)&d->p, d->cap) != hipSuccess) {
        (void)hipGetLastError();
        delete d;
        return nullptr;
      }
    }
    return std::shared_ptr<DevStream>(d, [this](DevStream* x) {
      std::unique_lock<std::mutex> lk(mu_);
      if (devs_.size() < 6 && x->cap <= 2 * kDeviceStdCap) {
        devs_.push_back(x);
        return;
      }
      lk.unlock();
      (void)hipSetDevice(x->dev);
      (void)hipFree(x->p);
      delete x;
    });
  }
  std::shared_ptr<ColStore> TakeCols(size_t cap, size_t names_cap = 0) {
    ColStore* c = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t k = 0; k < cols_.size(); k++)
        if (cols_[k]->cap >= cap && cols_[k]->names_cap >= names_cap) {
          c = cols_[k];
          cols_.erase(cols_.begin() + (ptrdiff_t)k);
          break;
        }
    }
    if (!c) {
      c = new ColStore();
      const size_t N = (std::max<size_t>(cap, 3u << 20) + 63) & ~(size_t)63;
      // (with names: N + 1 offsets and the bytes behind the columns)
      const size_t NC = names_cap ? (names_cap + 255) & ~(size_t)255 : 0;
      const size_t total = N * 48 + (NC ? (N + 64) * 4 + NC : 0);
      void* b = nullptr;
      if (hipHostMalloc(&b, total, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        b = malloc(total);
        if (!b) { delete c; return nullptr; }
        c->block = nullptr;  // (not pinned: freed with free)
        plain_.push_back(b);
      } else {
        c->block = b;
      }
      uint8_t* q = (uint8_t*)b;
      c->cap = N;
      c->c.cap = (int64_t)N;
      c->c.off = (uint32_t*)q;          q += 4 * N;
      c->c.ref_id = (int32_t*)q;        q += 4 * N;
      c->c.pos = (int32_t*)q;           q += 4 * N;
      c->c.mate_ref_id = (int32_t*)q;   q += 4 * N;
      c->c.mate_pos = (int32_t*)q;      q += 4 * N;
      c->c.n_cigar = (uint32_t*)q;      q += 4 * N;
      c->c.l_seq = (uint32_t*)q;        q += 4 * N;
      c->c.bytes = (uint32_t*)q;        q += 4 * N;
      c->c.span = (int32_t*)q;          q += 4 * N;
      c->c.flags = (uint16_t*)q;        q += 2 * N;
      c->c.mapq = q;                    q += N;
      c->c.l_name = q;                  q += N;
      c->c.facts = q;
      c->c.name_off = nullptr;
      c->c.names = nullptr;
      c->c.names_cap = 0;
      if (NC) {
        q = (uint8_t*)b + N * 48;
        c->c.name_off = (uint32_t*)q;   q += (N + 64) * 4;
        c->c.names = q;
        c->c.names_cap = (int64_t)NC;
        c->names_cap = NC;
      }
    }
    // (column stores are few and of one size: they go back to the pool and stay for the process)
    return std::shared_ptr<ColStore>(c, [this](ColStore* x) {
      std::lock_guard<std::mutex> lk(mu_);
      cols_.push_back(x);
    });
  }

 private:
  std::mutex mu_;
  std::vector<DevStream*> devs_;
  std::vector<ColStore*> cols_;
  std::vector<void*> plain_;
};

// A batch of the resident pipeline's first stage: read and inflated on the device, its bytes at
// dev->p[kHead, kHead + out_total); nothing found in it yet.
struct InfBatch {
  std::shared_ptr<DevStream> dev;
  size_t out_total = 0;
  bool eof = false;
  std::string err;
};

struct ResBatch {
  std::shared_ptr<DevStream> dev;
  size_t at = 0, size = 0;  // the stream's bytes [at, size); at = the start of a record
  std::shared_ptr<ColStore> cols;
  int64_t n = 0;       // whole records found
  size_t exit = 0;     // the offset behind the last of them
  // the device scan failed (a guess was wrong): bytes fetched, walked on the host
  std::shared_ptr<Buf> host;
  Walk host_walk;
  bool fields_ok = true, sorted = true, one_ref = true;
  int32_t r0 = -1, p0 = 0, r1 = -1, p1 = 0;
  bool eof = false;
  std::string err;
};

}  // namespace

struct BamReader::Impl {
  FILE* f = nullptr;
  bool file_eof = false;     // the batch reader met the end of the file
  bool pinned = false;       // inflate into pinned host memory
  std::atomic<int> inflate_device{-1};  // >= 0: BGZF blocks are inflated on that HIP device (may be switched on while reading)
  std::vector<bp_inflater*> inflaters;
  std::vector<bp_bgzf_block> blk_tab;
  double dev_inflate_ms = 0;
  // resident mode
  bool resident = false, res_done = false;
  bool want_names = false;   // the scan also fetches the read names (stitching pairs reads by them)
  int32_t n_ref = 0;
  std::shared_ptr<ResBatch> res_prev;                // (the read-ahead task's) the batch before, for its tail
  std::vector<uint8_t> res_init_tail;                // bytes the host-mode reader had not consumed at the switch
  std::future<std::shared_ptr<ResBatch>> res_next;
  std::future<std::unique_ptr<InfBatch>> inf_next;   // the batch behind that one: being read and inflated while it is scanned
  bp_inflater* scan_inflater = nullptr;              // the scan's own stream and buffers (the inflaters are busy with the next batch)
  double t_scan = 0;                                 // wall seconds of the scan stage (tail, record scan, columns, survey)
  std::atomic<int64_t> res_fallbacks{0};
  double t_readraw = 0, t_inflate = 0, t_take = 0;  // wall seconds inside ReadRaw / Inflate / buffer allocation (BIO_PILEUP_STATS)
  std::shared_ptr<Buf> cur;  // inflated bytes not yet consumed: cur->p[rd, end)
  size_t rd = 0, end = 0;
  uint64_t file_pos = 0;  // file offset of the next byte fread returns
  double wait_s = 0;      // time the consumer waited for inflated batches
  // (offset in cur, file offset) of the blocks cur holds bytes of, ascending
  std::vector<std::pair<size_t, uint64_t>> marks;
  std::vector<Walk> walks;  // NextChunk's, kept for their vectors

  // One batch of BGZF blocks goes through two stages off the main thread, each a batch ahead of
  // the next: its bytes are read from the file and its block headers parsed (one thread: a memcpy
  // out of the page cache, a third of a second per GB), then its blocks are inflated (all threads).
  // File bytes of a batch: pinned memory when the device inflates them (it copies them itself then,
  // at the bus rate, instead of through the runtime's staging), grown like a vector.
  struct RawBytes {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;
    ~RawBytes() { drop(); }
    void drop() {
      if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
      p = nullptr;
      n = cap = 0;
    }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    void resize(size_t m, bool want_pinned) {
      if (m > cap || (want_pinned && !pinned && cap)) {
        const size_t ncap = std::max(m + m / 8 + 4096, cap);
        uint8_t* q = nullptr;
        bool pin = false;
        if (want_pinned) {
          void* hp = nullptr;
          if (hipHostMalloc(&hp, ncap, hipHostMallocPortable) == hipSuccess) { q = (uint8_t*)hp; pin = true; }
          else (void)hipGetLastError();
        }
        if (!q) q = (uint8_t*)malloc(ncap);
        if (!q) throw std::bad_alloc();
        if (n) memcpy(q, p, n);
        const size_t keep = n;
        drop();
        p = q; cap = ncap; pinned = pin; n = keep;
      }
      n = m;
    }
  };
  struct RawBatch {
    RawBytes raw;  // file bytes from `at` on
    struct Blk { size_t off, clen, isize, out; };    // deflate data at raw[off, off + clen), crc32 behind it; inflates to `out`
    std::vector<Blk> blks;
    std::vector<std::pair<size_t, uint64_t>> marks;  // (offset in the inflated buffer, file offset)
    size_t out_total = 0;
    bool eof = false;
    uint64_t end_pos = 0;  // file offset behind the batch
    std::string err;
  };
  struct Batch {
    std::shared_ptr<Buf> buf;  // inflated bytes at buf->p[kHead, size)
    size_t size = kHead;
    std::vector<std::pair<size_t, uint64_t>> marks;
    bool eof = false;
    uint64_t end_pos = 0;
    std::string err;
  };
  std::future<std::unique_ptr<Batch>> next;         // the batch behind the one being consumed: inflating or inflated
  std::future<std::unique_ptr<RawBatch>> next_raw;  // the one behind that: being read (touched by next's task and, once that is done, Seek)
  uint64_t raw_pos = 0;                             // where the read after next_raw starts (likewise)
  size_t est_bytes = 0;                             // compressed size of the last batch read
  std::mutex spare_mu;
  std::vector<std::unique_ptr<RawBatch>> spare;     // RawBatches to use again (their vectors' memory is already faulted in)

  ~Impl() {
    if (next.valid()) next.wait();
    if (res_next.valid()) res_next.wait();
    if (inf_next.valid()) inf_next.wait();
    if (next_raw.valid()) next_raw.wait();
    res_prev.reset();
    for (bp_inflater* x : inflaters) bp_inflate_destroy(x);
    if (scan_inflater) bp_inflate_destroy(scan_inflater);
    if (f) fclose(f);
  }

  std::unique_ptr<RawBatch> ReadRaw(uint64_t at) {
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double* a; std::chrono::steady_clock::time_point t; ~Acc() { *a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{&t_readraw, t_in};
    std::unique_ptr<RawBatch> out;
    {
      std::lock_guard<std::mutex> lk(spare_mu);
      if (!spare.empty()) {
        out = std::move(spare.back());
        spare.pop_back();
      }
    }
    if (!out) out.reset(new RawBatch);
    RawBatch& B = *out;
    B.blks.clear();
    B.marks.clear();
    B.out_total = 0;
    B.eof = false;
    B.err.clear();
    const int fd = fileno(f);
    const int batch_blocks = BatchBlocks(inflate_device >= 0);
    size_t have = 0;
    bool file_end = false;
    auto read_more = [&](size_t want) {  // appends up to `want` file bytes
      if (B.raw.size() < have + want) B.raw.resize(have + want, pinned && inflate_device >= 0);
      // A large read is cut into pieces that are read side by side: one thread copying out of the
      // page cache (3 GB/s) was the longest stage of the pipeline once the blocks were inflated on
      // the device.  A piece that comes back short means the end of the file lies inside it.
      constexpr size_t kPiece = 8u << 20;
      if (want >= 2 * kPiece) {
        const size_t np = std::min<size_t>(8, want / kPiece);
        std::vector<size_t> got(np, 0);
        std::atomic<int> bad(0);
        uint8_t* const dst = B.raw.data() + have;
        const off_t off0 = (off_t)(at + have);
        bp_par::parallel_for(np, (int)np, [&](size_t k) {
          const size_t a = want * k / np, b = want * (k + 1) / np;
          size_t g = 0;
          while (g < b - a) {
            const ssize_t r = pread(fd, dst + a + g, b - a - g, off0 + (off_t)(a + g));
            if (r < 0) { bad = 1; break; }
            if (r == 0) break;
            g += (size_t)r;
          }
          got[k] = g;
        });
        if (bad) { B.err = "bgzf: read failed"; file_end = true; return; }
        for (size_t k = 0; k < np; k++) {
          const size_t a = want * k / np, b = want * (k + 1) / np;
          have += got[k];
          if (got[k] < b - a) { file_end = true; break; }
        }
        return;
      }
      while (want && !file_end) {
        ssize_t got = pread(fd, B.raw.data() + have, want, (off_t)(at + have));
        if (got < 0) { B.err = "bgzf: read failed"; file_end = true; break; }
        if (got == 0) { file_end = true; break; }
        have += (size_t)got;
        want -= (size_t)got;
      }
    };
    read_more(est_bytes ? est_bytes + est_bytes / 16 + 65536 : (size_t)batch_blocks * 20480 + 65536);
    size_t p = 0;
    while ((int)B.blks.size() < batch_blocks && B.err.empty()) {
      // the fixed part of the header and the length of the extra field
      if (have - p < 18 && !file_end) read_more(1 << 20);
      if (have == p) { B.eof = true; break; }
      const uint8_t* hd = B.raw.data() + p;
      if (have - p < 18 || hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4)) {
        B.err = "bgzf: invalid block header";
        break;
      }
      const unsigned xlen = U16(hd + 10);
      if (have - p < 12 + (size_t)xlen && !file_end) read_more(1 << 20);
      if (have - p < 12 + (size_t)xlen) { B.err = "bgzf: truncated block"; break; }
      hd = B.raw.data() + p;
      // The BC subfield is first in every BGZF writer in use; handle the general case too.
      const uint8_t* extra = hd + 12;
      int bsize = -1;
      for (size_t q = 0; q + 4 <= xlen;) {
        unsigned slen = U16(extra + q + 2);
        if (extra[q] == 'B' && extra[q + 1] == 'C' && slen == 2 && q + 6 <= xlen) bsize = U16(extra + q + 4);
        q += 4 + slen;
      }
      if (bsize < 0) { B.err = "bgzf: missing BC subfield"; break; }
      const size_t total = (size_t)bsize + 1;
      const long remaining = (long)total - 12 - (long)xlen;  // deflate data + crc32 + isize
      if (xlen < 6 || remaining < 8) { B.err = "bgzf: invalid block size"; break; }
      if (have - p < total && !file_end) read_more(std::max<size_t>(1 << 20, total));
      if (have - p < total) { B.err = "bgzf: truncated block"; break; }
      const size_t isize = U32(B.raw.data() + p + total - 4);
      if (isize > 65536) { B.err = "bgzf: invalid uncompressed size"; break; }
      B.blks.push_back({p + 12 + xlen, (size_t)remaining - 8, isize, kHead + B.out_total});
      if (isize) B.marks.emplace_back(kHead + B.out_total, at + p);
      p += total;
      B.out_total += isize;
    }
    B.end_pos = at + p;
    if (p) est_bytes = p;
    return out;
  }

  // The blocks of R inflated on the device: the compressed bytes as they lie in the file plus one
  // table entry per block; CRC-32 and ISIZE are checked there.  The batch is cut into parts that
  // travel and inflate side by side on streams of their own (copies either way overlap the
  // kernels).  The output goes to host memory (out_host + offset in the batch) or stays on the
  // device (out_dev + offset).
  void DeviceInflate(RawBatch& R, int idev, uint8_t* out_host, uint8_t* out_dev, std::string* err) {
    blk_tab.clear();
    for (const RawBatch::Blk& b : R.blks)
      if (b.isize)
        blk_tab.push_back({(uint32_t)b.off, (uint32_t)b.clen, (uint32_t)(b.out - kHead), (uint32_t)b.isize,
                           U32(R.raw.data() + b.off + b.clen)});
    const size_t nb = blk_tab.size();
    if (nb == 0) return;
    // Parts pay when the output travels back to the host beside the next part's kernel.  When it stays
    // on the device one launch is best: chr1 30x -bgz end to end 1.06 s with 1 part, 1.10 / 1.13 / 1.15 /
    // 1.47 s with 2 / 3 / 4 / 8 (same box, alternating) - a batch takes one block's decode time however
    // it is cut, and more streams only stand in each other's way and in the scan's.
    const size_t parts = out_dev ? 1 : std::min<size_t>(kDeviceParts, std::max<size_t>(1, nb / 256));
    while (inflaters.size() < std::max<size_t>(parts, 1)) {
      bp_inflater* x = nullptr;
      if (bp_inflate_create(idev, &x) != BP_OK) { *err = bp_inflate_last_error(nullptr); return; }
      inflaters.push_back(x);
    }
    std::vector<std::string> errs(parts);
    std::vector<float> ms(parts, 0.f);
    auto one = [&](size_t k) {
      const size_t a = nb * k / parts, e = nb * (k + 1) / parts;
      if (a == e) return;
      const uint32_t in_lo = blk_tab[a].in_off, in_hi = blk_tab[e - 1].in_off + blk_tab[e - 1].in_len;
      const uint32_t out_lo = blk_tab[a].out_off, out_hi = blk_tab[e - 1].out_off + blk_tab[e - 1].out_len;
      for (size_t i = a; i < e; i++) { blk_tab[i].in_off -= in_lo; blk_tab[i].out_off -= out_lo; }
      int64_t bad = -1;
      const int rc = out_dev ? bp_inflate_blocks_to(inflaters[k], R.raw.data() + in_lo, in_hi - in_lo, blk_tab.data() + a,
                                                    (int64_t)(e - a), out_dev + out_lo, out_hi - out_lo, 1, &bad)
                             : bp_inflate_blocks(inflaters[k], R.raw.data() + in_lo, in_hi - in_lo, blk_tab.data() + a,
                                                 (int64_t)(e - a), out_host + out_lo, out_hi - out_lo, 1, &bad);
      if (rc != BP_OK) {
        const std::string m = bp_inflate_last_error(inflaters[k]);
        errs[k] = m.compare(0, 5, "bgzf:") == 0 ? "bgzf: corrupt block (inflate or CRC failure)" : m;
      }
      float a_ms = 0, c_ms = 0;
      bp_inflate_timing(inflaters[k], &a_ms, &c_ms);
      ms[k] = a_ms + c_ms;
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < parts; k++) th.emplace_back(one, k);
    one(0);
    for (std::thread& t : th) t.join();
    float mx = 0;
    for (size_t k = 0; k < parts; k++) {
      if (!errs[k].empty() && err->empty()) *err = errs[k];
      mx = std::max(mx, ms[k]);
    }
    dev_inflate_ms += mx;
  }

  // Resident mode, first stage: the next batch of blocks read and inflated into a device buffer of
  // its own, kHead bytes in (room for the unconsumed tail of the batch before, which is only known
  // once that batch has been scanned).  Runs beside the second stage's work on the batch before.
  std::unique_ptr<InfBatch> InflateStage() {
    std::unique_ptr<InfBatch> out(new InfBatch);
    InfBatch& I = *out;
    const int idev = inflate_device.load();
    std::unique_ptr<RawBatch> r = next_raw.valid() ? next_raw.get() : ReadRaw(raw_pos);
    raw_pos = r->end_pos;
    if (r->err.empty() && !r->eof) {
      const uint64_t pos = raw_pos;
      next_raw = std::async(std::launch::async, [this, pos] { return ReadRaw(pos); });
    }
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double* a; std::chrono::steady_clock::time_point t; ~Acc() { *a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{&t_inflate, t_in};
    I.err = r->err;
    I.eof = r->eof;
    I.out_total = r->out_total;
    if (!I.err.empty()) return out;
    I.dev = ResidentPool::Get().TakeDev(idev, kHead + r->out_total + 4096);
    if (!I.dev) { I.err = "bam: out of device memory"; return out; }
    if (hipSetDevice(idev) != hipSuccess) { I.err = "hipSetDevice failed"; return out; }
    // (blk_tab offsets are relative to kHead in the batch: out_dev is where offset 0 goes)
    DeviceInflate(*r, idev, nullptr, I.dev->p + kHead, &I.err);
    {
      std::lock_guard<std::mutex> lk(spare_mu);
      if (spare.size() < 3) spare.push_back(std::move(r));
    }
    return out;
  }

  // Second stage: the inflated batch gets the unconsumed tail of the batch before it in front and
  // is scanned on the device; runs a batch ahead of NextChunk, and the first stage a batch ahead of it.
  std::shared_ptr<ResBatch> NextResident() {
    auto out = std::make_shared<ResBatch>();
    ResBatch& B = *out;
    const int idev = inflate_device.load();
    std::unique_ptr<InfBatch> inf = inf_next.valid() ? inf_next.get() : InflateStage();
    if (inf->err.empty() && !inf->eof) inf_next = std::async(std::launch::async, [this] { return InflateStage(); });
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double* a; std::chrono::steady_clock::time_point t; ~Acc() { *a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{&t_scan, t_in};
    B.err = inf->err;
    B.eof = inf->eof;
    if (!B.err.empty()) return out;
    if (hipSetDevice(idev) != hipSuccess) { B.err = "hipSetDevice failed"; return out; }
    const size_t tail = res_prev ? res_prev->size - res_prev->exit : res_init_tail.size();
    size_t head = kHead;
    B.dev = inf->dev;
    if (tail > kHead) {
      // a record longer than the room in front of a batch: the batch moves to a buffer with more
      head = (tail + 255) & ~(size_t)255;
      B.dev = ResidentPool::Get().TakeDev(idev, head + inf->out_total + 4096);
      if (!B.dev) { B.err = "bam: out of device memory"; return out; }
      if (inf->out_total &&
          hipMemcpy(B.dev->p + head, inf->dev->p + kHead, inf->out_total, hipMemcpyDeviceToDevice) != hipSuccess) {
        B.err = "bam: moving a batch on the device failed";
        return out;
      }
    }
    B.at = head - tail;
    B.size = head + inf->out_total;
    if (tail) {
      const hipError_t e = res_prev ? hipMemcpy(B.dev->p + B.at, res_prev->dev->p + res_prev->exit, tail, hipMemcpyDeviceToDevice)
                                    : hipMemcpy(B.dev->p + B.at, res_init_tail.data(), tail, hipMemcpyHostToDevice);
      if (e != hipSuccess) { B.err = std::string("bam: copying a batch's tail: ") + hipGetErrorString(e); return out; }
    }
    // (a device-to-device hipMemcpy may return before the copy has run, and the scan's stream does
    // not wait for the null stream: make sure the bytes are there, and the buffer moved from is idle)
    if (hipStreamSynchronize(nullptr) != hipSuccess) { B.err = "bam: device copy failed"; return out; }
    res_init_tail.clear();
    res_init_tail.shrink_to_fit();
    inf.reset();
    if (!scan_inflater && bp_inflate_create(idev, &scan_inflater) != BP_OK) {
      B.err = bp_inflate_last_error(nullptr);
      return out;
    }
    B.cols = ResidentPool::Get().TakeCols((B.size - B.at) / 96 + 4096, want_names ? (B.size - B.at) / 4 + 4096 : 0);
    if (!B.cols) { B.err = "bam: out of memory"; return out; }
    uint64_t exit = B.at;
    int32_t stopped = 0;
    int rc = B.size > B.at ? bp_bam_scan(scan_inflater, B.dev->p, B.at, B.size, n_ref, &B.cols->c, &B.n, &exit, &stopped) : BP_OK;
    if (rc == 1 || rc == BP_ENOMEM) {
      // a guess of the device walk was wrong (or the records are tiny): this batch is walked on the host
      if (res_fallbacks++ == 0)
        if (const char* e = getenv("BIO_PILEUP_STATS"); e && *e && *e != '0')
          fprintf(stderr, "bam: device scan of a batch gave way to the host's walk (%d): %s\n", rc, bp_inflate_last_error(scan_inflater));
      B.cols.reset();
      B.host = BufPool::Get().Take(B.size - B.at + 64, pinned, B.size - B.at + 64 > kStdCap);
      if (!B.host) { B.err = "bam: out of memory"; return out; }
      if (hipMemcpy(B.host->p, B.dev->p + B.at, B.size - B.at, hipMemcpyDeviceToHost) != hipSuccess) {
        B.err = "bam: copying a batch to the host failed";
        return out;
      }
      WalkFrom(B.host->p - B.at, B.at, B.size, B.size, &B.host_walk);  // (offsets in the stream's coordinates)
      B.n = (int64_t)B.host_walk.off.size();
      exit = B.host_walk.exit;
      B.fields_ok = B.host_walk.fields_ok;
      B.sorted = B.host_walk.sorted;
      B.one_ref = B.host_walk.one_ref;
      B.r0 = B.host_walk.r0; B.p0 = B.host_walk.p0; B.r1 = B.host_walk.r1; B.p1 = B.host_walk.p1;
    } else if (rc != BP_OK) {
      B.err = bp_inflate_last_error(scan_inflater);
      return out;
    } else if (B.n > 0) {
      // what NextChunk's walk surveys, from the columns
      const bp_bam_columns& c = B.cols->c;
      const size_t n = (size_t)B.n, chunk = 65536, nch = (n + chunk - 1) / chunk;
      std::vector<uint8_t> ok(nch, 1), srt(nch, 1), one(nch, 1);
      const int32_t ref0 = c.ref_id[0];
      bp_par::parallel_for(nch, HostThreads(), [&](size_t ci) {
        const size_t a = ci * chunk, e = std::min(n, a + chunk);
        uint64_t last = a ? OrderKey(c.ref_id[a - 1], c.pos[a - 1]) : 0;
        for (size_t i = a; i < e; i++) {
          if (c.facts[i] & 4) ok[ci] = 0;
          const uint64_t key = OrderKey(c.ref_id[i], c.pos[i]);
          if (i && key < last) srt[ci] = 0;
          if (c.ref_id[i] != ref0) one[ci] = 0;
          last = key;
        }
      });
      for (size_t ci = 0; ci < nch; ci++) {
        B.fields_ok &= ok[ci] != 0;
        B.sorted &= srt[ci] != 0;
        B.one_ref &= one[ci] != 0;
      }
      B.r0 = c.ref_id[0]; B.p0 = c.pos[0]; B.r1 = c.ref_id[n - 1]; B.p1 = c.pos[n - 1];
    }
    B.exit = (size_t)exit;
    if (B.eof && B.exit != B.size) B.err = "bam: truncated record";
    res_prev = B.eof ? nullptr : out;
    return out;
  }

  std::unique_ptr<Batch> Inflate(std::unique_ptr<RawBatch> in) {
    std::unique_ptr<Batch> out(new Batch);
    Batch& B = *out;
    RawBatch& R = *in;
    B.err = R.err;
    B.eof = R.eof;
    B.end_pos = R.end_pos;
    B.marks.swap(R.marks);
    if (!B.err.empty()) return out;
    const int idev = inflate_device.load();
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double* a; std::chrono::steady_clock::time_point t; ~Acc() { *a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } acc{&t_inflate, t_in};
    B.buf = BufPool::Get().Take(kHead + R.out_total, pinned, kHead + R.out_total > kStdCap);
    if (!B.buf) { B.err = "bam: out of memory"; return out; }
    t_take += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
    B.size = kHead + R.out_total;
    if (idev >= 0 && !R.blks.empty()) {
      DeviceInflate(R, idev, B.buf->p + kHead, nullptr, &B.err);
    } else if (!R.blks.empty()) {
      uint8_t* const dst = B.buf->p;
      const uint8_t* const raw = R.raw.data();
      const std::vector<RawBatch::Blk>& blks = R.blks;
      const Deflate& ld = GetDeflate();
      std::atomic<int> bad(0);
      // a few blocks per task: the pool's threads take them as they come free
      constexpr size_t kPer = 4;
      bp_par::parallel_for((blks.size() + kPer - 1) / kPer, HostThreads(), [&](size_t c) {
        // (a thread keeps its decompressor as long as it lives)
        thread_local struct Dec { void* p = nullptr; ~Dec() { if (p) GetDeflate().free_(p); } } tl;
        if (ld.ok && !tl.p) tl.p = ld.alloc();
        void* const dec = tl.p;
        for (size_t i = c * kPer; i < std::min(blks.size(), (c + 1) * kPer); i++) {
          const RawBatch::Blk& b = blks[i];
          if (b.isize == 0) continue;
          if (dec) {
            size_t got = 0;
            if (ld.decompress(dec, raw + b.off, b.clen, dst + b.out, b.isize, &got) != 0 || got != b.isize ||
                ld.crc32_(0, dst + b.out, b.isize) != U32(raw + b.off + b.clen))
              bad = 1;
            continue;
          }
          z_stream zs;
          memset(&zs, 0, sizeof(zs));
          if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; continue; }
          zs.next_in = const_cast<uint8_t*>(raw + b.off);
          zs.avail_in = (uInt)b.clen;
          zs.next_out = dst + b.out;
          zs.avail_out = (uInt)b.isize;
          int rc = inflate(&zs, Z_FINISH);
          inflateEnd(&zs);
          if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; continue; }
          uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), dst + b.out, (uInt)b.isize);
          if (crc != U32(raw + b.off + b.clen)) bad = 1;
        }
      });
      if (bad) B.err = "bgzf: corrupt block (inflate or CRC failure)";
    }
    std::lock_guard<std::mutex> lk(spare_mu);
    if (spare.size() < 3) spare.push_back(std::move(in));
    return out;
  }

  // The batch at raw_pos: its bytes are being read already (next_raw) or are read now; the read
  // of the batch behind it is started before this one is inflated.
  std::unique_ptr<Batch> NextBatch() {
    std::unique_ptr<RawBatch> r = next_raw.valid() ? next_raw.get() : ReadRaw(raw_pos);
    raw_pos = r->end_pos;
    if (r->err.empty() && !r->eof) {
      const uint64_t pos = raw_pos;
      next_raw = std::async(std::launch::async, [this, pos] { return ReadRaw(pos); });
    }
    return Inflate(std::move(r));
  }

  // Brings in the next batch behind the unconsumed tail and starts reading the one after it.
  std::string Refill() {
    if (file_eof) return "";
    const auto w0 = std::chrono::steady_clock::now();
    std::unique_ptr<Batch> b = next.valid() ? next.get() : NextBatch();
    wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    if (!b->err.empty()) return b->err;
    const size_t tail = end - rd;
    std::shared_ptr<Buf> nb = b->buf;
    size_t at, nend;
    ptrdiff_t shift;  // of the batch's marks
    if (tail <= kHead) {
      at = kHead - tail;
      if (tail) memcpy(nb->p + at, cur->p + rd, tail);
      nend = b->size;
      shift = 0;
    } else {
      // a record longer than kHead: tail and batch side by side in a buffer of their own
      nb = BufPool::Get().Take(tail + (b->size - kHead), false);
      if (!nb) return "bam: out of memory";
      memcpy(nb->p, cur->p + rd, tail);
      memcpy(nb->p + tail, b->buf->p + kHead, b->size - kHead);
      at = 0;
      nend = tail + (b->size - kHead);
      shift = (ptrdiff_t)tail - (ptrdiff_t)kHead;
    }
    // the mark of the block the tail starts in, and later ones, move with it
    std::vector<std::pair<size_t, uint64_t>> nm;
    if (tail) {
      size_t k = 0;
      while (k + 1 < marks.size() && marks[k + 1].first <= rd) k++;
      for (; k < marks.size(); k++) nm.emplace_back((marks[k].first > rd ? marks[k].first - rd : 0) + at, marks[k].second);
    }
    for (const auto& mk : b->marks) nm.emplace_back((size_t)((ptrdiff_t)mk.first + shift), mk.second);
    marks.swap(nm);
    cur = nb;
    rd = at;
    end = nend;
    file_pos = b->end_pos;
    file_eof = b->eof;
    if (!file_eof) next = std::async(std::launch::async, [this] { return NextBatch(); });
    return "";
  }

  // Makes at least n unconsumed bytes available; returns false at a clean or short EOF.
  bool Need(size_t n, std::string* err) {
    while (end - rd < n) {
      if (file_eof) return false;
      *err = Refill();
      if (!err->empty()) return false;
    }
    return true;
  }
  const uint8_t* at() const { return cur->p + rd; }
};

BamReader::BamReader() : impl_(new Impl) {}
BamReader::~BamReader() {}

void BamReader::SetPinned(bool on) { impl_->pinned = on; }
void BamReader::SetDeviceInflate(int device) { impl_->inflate_device.store(device); }
double BamReader::DeviceInflateSeconds() const { return impl_->dev_inflate_ms * 1e-3; }
int64_t BamReader::ResidentFallbacks() const { return impl_->res_fallbacks.load(); }

std::string BamReader::StartResident(int device, bool want_names) {
  Impl& m = *impl_;
  m.want_names = want_names;
  // what the host-mode reader has read ahead is brought in first: its unconsumed bytes become the
  // tail in front of the first device batch
  if (m.next.valid()) {
    std::unique_ptr<Impl::Batch> b = m.next.get();
    if (!b->err.empty()) return b->err;
    const size_t tail = m.end - m.rd;
    m.res_init_tail.assign(m.cur ? m.cur->p + m.rd : nullptr, m.cur ? m.cur->p + m.rd + tail : nullptr);
    m.res_init_tail.insert(m.res_init_tail.end(), b->buf->p + kHead, b->buf->p + b->size);
    if (b->eof) m.file_eof = true;
  } else if (m.cur) {
    m.res_init_tail.assign(m.cur->p + m.rd, m.cur->p + m.end);
  }
  m.cur.reset();
  m.rd = m.end = 0;
  m.marks.clear();
  m.n_ref = (int32_t)header_.ref_lens.size();
  m.inflate_device.store(device);
  m.resident = true;
  return "";
}

std::string BamReader::FetchHost(Chunk* c) {
  if (c->data || !c->dev) return "";
  Impl& m = *impl_;
  const size_t first = c->off.empty() ? 0 : c->off[0];
  const size_t n = c->bytes - first;
  std::shared_ptr<Buf> hb = BufPool::Get().Take(n + 64, m.pinned, n + 64 > kStdCap);
  if (!hb) return "bam: out of memory";
  if (hipSetDevice(m.inflate_device.load()) != hipSuccess ||
      hipMemcpy(hb->p, c->dev + first, n, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return "bam: copying a chunk to the host failed";
  }
  struct Both { std::shared_ptr<const void> a; std::shared_ptr<Buf> b; };
  auto both = std::make_shared<Both>();
  both->a = c->hold;
  both->b = hb;
  c->hold = both;
  c->data = hb->p - first;  // (record k at data + off[k], as everywhere)
  return "";
}

void BamReader::StageSeconds(double* read_raw, double* inflate, double* take, double* scan) const {
  *read_raw = impl_->t_readraw;
  *inflate = impl_->t_inflate;
  *take = impl_->t_take;
  if (scan) *scan = impl_->t_scan;
}

std::string BamReader::Open(const std::string& path) {
  impl_->f = fopen(path.c_str(), "rb");
  if (!impl_->f) return "open " + path + ": no such file or directory";
  std::string err;
  Impl& m = *impl_;
  if (!m.Need(12, &err)) return err.empty() ? "bam: truncated header" : err;
  if (memcmp(m.at(), "BAM\1", 4) != 0) return "bam: magic number mismatch";
  uint32_t l_text = U32(m.at() + 4);
  if (!m.Need(12 + (size_t)l_text, &err)) return err.empty() ? "bam: truncated header" : err;
  header_.text.assign((const char*)m.at() + 8, l_text);
  while (!header_.text.empty() && header_.text.back() == '\0') header_.text.pop_back();
  uint32_t n_ref = U32(m.at() + 8 + l_text);
  m.rd += 12 + (size_t)l_text;
  for (uint32_t i = 0; i < n_ref; i++) {
    if (!m.Need(4, &err)) return err.empty() ? "bam: truncated header" : err;
    uint32_t l_name = U32(m.at());
    if (!m.Need(8 + (size_t)l_name, &err)) return err.empty() ? "bam: truncated header" : err;
    std::string nm((const char*)m.at() + 4, l_name ? l_name - 1 : 0);
    header_.ref_names.push_back(nm);
    header_.ref_lens.push_back(I32(m.at() + 4 + l_name));
    m.rd += 8 + (size_t)l_name;
  }
  return "";
}

std::string BamReader::Seek(uint64_t voffset) {
  Impl& m = *impl_;
  const uint64_t coff = voffset >> 16;
  // what is being read and inflated ahead is dropped
  if (m.next.valid()) m.next.get();
  if (m.next_raw.valid()) m.next_raw.get();
  m.file_pos = m.raw_pos = coff;
  m.file_eof = false;
  m.rd = m.end = 0;  // (chunks handed out keep their buffer; the reader lets go of it at the refill)
  m.marks.clear();
  std::string err = m.Refill();
  if (!err.empty()) return err;
  if ((voffset & 0xffff) > m.end - m.rd) return "bgzf: virtual offset past the end of its block";
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
  const size_t have = m.end - m.rd;
  if (have < 4) return true;
  return have < 4 + (size_t)U32(m.at());
}

bool BamReader::Parse(const uint8_t* rec, Raw* r) {
  const uint32_t block_size = U32(rec);
  const uint8_t* p = rec + 4;
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
  if (need > block_size) return false;
  r->name = (const char*)p + 32;
  r->l_name = l_read_name ? l_read_name - 1 : 0;
  r->cigar = (const uint32_t*)(p + 32 + l_read_name);  // read with memcpy by callers
  r->seq = p + 32 + l_read_name + 4 * (size_t)r->n_cigar;
  r->qual = r->seq + ((size_t)r->l_seq + 1) / 2;
  r->aux = r->qual + r->l_seq;
  r->l_aux = (uint32_t)(block_size - need);
  return true;
}

// The first record of what is buffered, whole: 1; clean end of the file: 0; else -1.
static int FirstRecord(BamReader::Impl& m, std::string* err_out) {
  std::string err;
  if (!m.Need(4, &err)) {
    if (!err.empty()) { *err_out = err; return -1; }
    if (m.end != m.rd) { *err_out = "bam: truncated record"; return -1; }
    return 0;
  }
  uint32_t block_size = U32(m.at());
  if (block_size < 32) { *err_out = "bam: invalid record block size"; return -1; }
  if (!m.Need(4 + (size_t)block_size, &err)) {
    *err_out = err.empty() ? "bam: truncated record" : err;
    return -1;
  }
  return 1;
}

int BamReader::Next(Raw* r) {
  Impl& m = *impl_;
  int rc = FirstRecord(m, &err_);
  if (rc <= 0) return rc;
  const uint8_t* rec = m.at();
  m.rd += 4 + (size_t)U32(rec);
  // The walk is a pointer chase (a record's length says where the next one starts) through
  // bytes another core has just inflated: pull the lines a few records ahead in now.
  for (int k = 0; k < 5; k++) __builtin_prefetch(rec + 1028 + 64 * k);
  if (!Parse(rec, r)) { err_ = "bam: record fields exceed block size"; return -1; }
  return 1;
}

// The records of a buffer are a chain (a record's block size says where the next one starts), and
// walking it alone costs a cache miss per record on bytes other cores have just inflated: 35 ns x
// 50 M records for a 30x chr1.  So the buffer is cut into one stretch per thread; each thread but
// the first GUESSES where a record starts in its stretch (the first offset behind the cut where a
// record and the two behind it look like records) and walks on from there to the next cut, noting
// offsets and surveying fields as it goes.  The guesses are then checked in order: stretch j+1's
// start is right iff it is where stretch j's walk - right by induction from the known start of
// the buffer - came out.  A wrong or missing guess costs a second walk of that stretch from the
// right place; it never changes the result.
int BamReader::NextChunk(Chunk* c) {
  Impl& m = *impl_;
  c->off.clear();
  if (m.resident) {
    c->dev = nullptr;
    c->cols = nullptr;
    c->data = nullptr;
    for (;;) {
      if (m.res_done) return 0;
      const auto w0 = std::chrono::steady_clock::now();
      std::shared_ptr<ResBatch> b = m.res_next.valid() ? m.res_next.get() : m.NextResident();
      m.wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      if (!b->err.empty()) {
        err_ = b->err;
        m.res_done = true;
        return -1;
      }
      if (b->eof) m.res_done = true;
      else m.res_next = std::async(std::launch::async, [this] { return impl_->NextResident(); });
      if (b->n == 0) continue;  // (no whole record yet: a record longer than a batch, or the file's end)
      c->hold = b;
      c->dev = b->dev->p;
      c->bytes = b->exit;
      c->fields_ok = b->fields_ok;
      c->sorted = b->sorted;
      c->one_ref = b->one_ref;
      c->first_ref = b->r0; c->first_pos = b->p0; c->last_ref = b->r1; c->last_pos = b->p1;
      c->rewalked = 0;
      if (b->cols) {
        c->cols = &b->cols->c;
        c->off.assign(b->cols->c.off, b->cols->c.off + b->n);
      } else {
        c->off = b->host_walk.off;
        c->data = b->host->p - b->at;
      }
      return 1;
    }
  }
  int rc = FirstRecord(m, &err_);
  if (rc <= 0) return rc;
  const uint8_t* base = m.cur->p;
  const size_t lo = m.rd, end = m.end;
  const int32_t n_ref = (int32_t)header_.ref_lens.size();
  constexpr size_t kMinStretch = 256 << 10;
  const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, HostThreads()), (end - lo) / kMinStretch));
  std::vector<Walk>& ws = m.walks;
  if (ws.size() < T + 1) ws.resize(T + 1);
  auto cut = [&](size_t j) { return j >= T ? end : lo + (end - lo) / T * j; };
  bp_par::parallel_for(T, (int)T, [&](size_t j) {
    Walk& w = ws[j];
    size_t p = lo;
    if (j > 0) {
      p = SIZE_MAX;
      // a record start lies within one record length of the cut: look a little further than any sane record is long
      const size_t from = cut(j), upto = std::min(end, from + (64 << 10));
      for (size_t q = from; q < upto; q++) {
        if (!Plausible(base, q, end, n_ref)) continue;
        size_t q2 = q + 4 + U32(base + q);
        if (q2 + 36 <= end) {
          if (!Plausible(base, q2, end, n_ref)) continue;
          const size_t q3 = q2 + 4 + U32(base + q2);
          if (q3 + 36 <= end && !Plausible(base, q3, end, n_ref)) continue;
        }
        p = q;
        break;
      }
      if (p == SIZE_MAX) {
        w.start = SIZE_MAX;
        return;
      }
    }
    WalkFrom(base, p, cut(j + 1), end, &w);
  });
  // check the guesses in order; the chunk ends where a walk stopped
  size_t at = lo;
  bool first = true;
  uint64_t last_key = 0;
  Walk& redo = ws[T];
  c->fields_ok = c->sorted = c->one_ref = true;
  c->rewalked = 0;
  for (size_t j = 0; j < T; j++) {
    if (at >= cut(j + 1)) continue;  // a long record carried the chain past this stretch
    Walk* w = &ws[j];
    if (w->start != at) {
      WalkFrom(base, at, cut(j + 1), end, &redo);
      w = &redo;
      c->rewalked++;
    }
    if (!w->off.empty()) {
      c->off.insert(c->off.end(), w->off.begin(), w->off.end());
      c->fields_ok &= w->fields_ok;
      c->sorted &= w->sorted && (first || OrderKey(w->r0, w->p0) >= last_key);
      if (first) {
        c->first_ref = w->r0;
        c->first_pos = w->p0;
      }
      c->one_ref &= w->one_ref && w->r0 == c->first_ref;
      c->last_ref = w->r1;
      c->last_pos = w->p1;
      last_key = OrderKey(w->r1, w->p1);
      first = false;
    }
    at = w->exit;
    if (w->stopped) break;
  }
  c->hold = m.cur;
  c->data = base;
  c->bytes = at;
  m.rd = at;
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
