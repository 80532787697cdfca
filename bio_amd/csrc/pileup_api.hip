// pileup_api.hip -- host side of libbiopileup.so: the C ABI of include/bio_pileup.h.
//
// One bp_handle = one job of pileupSNPMain (pileup/snp/pileup.go:862-939).  Records are
// packed into the device layout (pileup_device.h) in pinned staging buffers and streamed to
// HBM with hipMemcpyAsync; the whole job stays resident (13 GB of records + 11 GB of rows for
// a 30x chr1, against 288 GB of HBM3E) and bp_finish launches plan + pileup kernels over
// every BED window.
//
// Host-side work that the reference also does per record before counting:
//   * data errors (pileup.go:766-772, :355-359) are raised from bp_submit, in record order;
//   * stitch mode: the mate table (firstread.go:176-204) is resolved here, because it is
//     keyed by read name; the kernel receives one MateEntry per record.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bio_pileup.h"
#include "deflate_api.h"
#include "pack_device.h"
#include "par.h"
#include "pileup_device.h"

namespace bp {
void launch_plan(const KernelParams& P, int32_t* r0, int32_t* r1, hipStream_t s);
void launch_pileup(const KernelParams& P, hipStream_t s);
void launch_stitch_resolve(const KernelParams& P, hipStream_t s);
void launch_column_sums(const uint32_t* rows, int64_t n_rows, unsigned long long* sums, hipStream_t s);
void launch_rows_add(uint32_t* rows, const uint32_t* src, int64_t n_u32, hipStream_t s);
int window_size();
void launch_tile_scan(const uint32_t* rows, uint32_t* off32, int64_t n, unsigned long long* tile_sum,
                      int32_t* overflow, hipStream_t s);
void launch_per_read(const KernelParams& P, PerReadParams Q, hipStream_t s);
void launch_text(const TextParams& T, hipStream_t s);
void launch_text_write(const TextParams& T, hipStream_t s);
void launch_pack(const uint8_t* raw, const PackItem* items, int64_t n, ReadCore* cores, uint8_t* blob, int qpass,
                 int min_base_qual, int clip, hipStream_t s);
}  // namespace bp

using namespace bp;

// ---- scratch of the allocation-free compute (bp_compute_static; no counterpart in the reference).
// What bp_compute keeps in buffers it allocates (the plan kernel's record range per window, the
// work-item queue of deep windows with its counter, the device error flags) lives here in a
// file-scope device array instead, one copy per GPU, held until the process ends:
//   [0, 256)            n_extra, the queue's item counter
//   [256, 512)          DeviceError
//   [512, +4 maxW)      win_r0        [+4 maxW)  win_r1        [+16 maxItems)  extra_items
// In host code the array's NAME is only a stand-in: its device address is looked up once per
// device with hipGetSymbolAddress (bp_init, that device current) and kept in g_scratch_ptr; host
// code uses that pointer, never the name, wherever it needs the memory.
constexpr int64_t kStaticMaxWindows = 1 << 19;  // 536,870,912 window positions
constexpr int64_t kStaticMaxItems = 1 << 18;
constexpr size_t kStaticCounterOff = 0, kStaticErrOff = 256, kStaticR0Off = 512;
constexpr size_t kStaticR1Off = kStaticR0Off + 4 * (size_t)kStaticMaxWindows;
constexpr size_t kStaticItemsOff = kStaticR1Off + 4 * (size_t)kStaticMaxWindows;
constexpr size_t kStaticBytes = kStaticItemsOff + 16 * (size_t)kStaticMaxItems;  // 8 MiB + 512 B
static_assert(kStaticItemsOff % 16 == 0, "work items are uint4");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace {

thread_local std::string g_last_error;

// Device buffers and pinned staging memory of finished jobs are kept for the next job of the
// process instead of going back to the driver: a whole-genome run opens a job per 2^26 positions,
// and hipMalloc / hipFree of its gigabytes (rows, payload, cores) and hipHostMalloc of its pinned
// staging were a third of a job's time on the submitter and closer threads.  At most kCacheBytes
// per process stay cached; an allocation that fails empties the cache and tries once more.
// BP_NO_CACHE=1 turns it off.
constexpr size_t kCacheBytes = (size_t)48 << 30, kCacheMinBytes = (size_t)1 << 20;
struct DevCache {
  struct E { void* p; size_t cap; int dev; };
  std::mutex mu;
  std::vector<E> v;
  size_t bytes = 0;
  bool off = false;
  DevCache() {
    const char* e = getenv("BP_NO_CACHE");
    off = e && *e && *e != '0';
  }
  // the smallest cached buffer of `dev` that holds `need` without being more than twice as large
  void* take(int dev, size_t need, size_t* cap) {
    std::lock_guard<std::mutex> lk(mu);
    size_t best = SIZE_MAX;
    for (size_t k = 0; k < v.size(); k++)
      if (v[k].dev == dev && v[k].cap >= need && v[k].cap / 2 <= need + kCacheMinBytes && (best == SIZE_MAX || v[k].cap < v[best].cap))
        best = k;
    if (best == SIZE_MAX) return nullptr;
    void* p = v[best].p;
    *cap = v[best].cap;
    bytes -= v[best].cap;
    v.erase(v.begin() + (ptrdiff_t)best);
    return p;
  }
  void give(int dev, void* p, size_t cap) {  // with `dev` current
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!off && cap >= kCacheMinBytes && bytes + cap <= kCacheBytes && v.size() < 256) {
        v.push_back({p, cap, dev});
        bytes += cap;
        return;
      }
    }
    (void)hipFree(p);
  }
  void flush(int dev) {  // with `dev` current
    std::vector<E> out;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t k = 0; k < v.size();)
        if (v[k].dev == dev) {
          out.push_back(v[k]);
          bytes -= v[k].cap;
          v.erase(v.begin() + (ptrdiff_t)k);
        } else {
          k++;
        }
    }
    for (const E& e : out) (void)hipFree(e.p);
  }
};
DevCache& dev_cache() {
  static DevCache* c = new DevCache();  // never destroyed: the HIP runtime may be gone by then
  return *c;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void release(int dev) {  // with `dev` current
    dev_cache().give(dev, p, cap);
    p = nullptr;
    cap = 0;
  }
};

struct Staging {  // pinned host memory of one slice in flight; each part allocated when first used
  ReadCore* cores = nullptr;  // bp_submit: packed on the host
  uint8_t* blob = nullptr;
  PackItem* items = nullptr;  // bp_submit_bam: packed on the device
  uint8_t* raw = nullptr;     // ... record bytes gathered here when they cannot go straight from the caller's memory
  size_t raw_cap = 0;
  hipEvent_t done = nullptr;
  int dev = -1;  // the device `done` belongs to
  bool busy = false;
};

constexpr size_t kStageReads = 1u << 20;       // records per staging slice
constexpr size_t kStageBlob = 256u << 20;      // payload bytes per staging slice

// The firstread table (firstreadSNPTable, firstread.go:79-204) on the host: reads waiting for
// their mates.  The reference keeps a 2-D array of slices indexed by (pos mod nCirc, name hash);
// here a ring of chains indexed by the position the read's mate has to start at (the read's
// MatePos: tryRemove looks a read up by exactly that, firstread.go:136-145), entries and names in
// pools, the 64-bit name hash computed beforehand on the packing threads and the name bytes
// compared only when it matches.
struct StoredRead {
  uint64_t hash;
  int32_t key_pos;   // its MatePos
  int32_t pos;
  int64_t idx;       // record index in the job
  uint32_t name_off, name_len;
  MateEntry self;    // what its mate's entry of the mate table will say of it (x: flag bits only)
  int32_t next;      // chain / free list
};

struct alignas(128) FirstReadTable {  // (one per thread in scan_slice: no two on a cache line)
  std::vector<int32_t> head;   // [nCirc] chain heads, -1 = empty
  std::vector<StoredRead> ents;
  std::vector<char> names;
  int32_t free_head = -1;
  int64_t n_live = 0, live_name_bytes = 0;
  int32_t ref = -1;
  int64_t clean_upto = 0;      // chains of keys below it are empty
  uint32_t mask = 0;

  void init(int32_t max_read_span) {
    uint32_t n = 1;
    while (n <= (uint32_t)max_read_span) n <<= 1;  // live keys lie in [pos, pos + maxReadSpan]
    head.assign(n, -1);
    mask = n - 1;
  }
  void drop_chain(uint32_t slot) {
    int32_t e = head[slot];
    while (e >= 0) {
      const int32_t nx = ents[(size_t)e].next;
      live_name_bytes -= ents[(size_t)e].name_len;
      ents[(size_t)e].next = free_head;
      free_head = e;
      n_live--;
      e = nx;
    }
    head[slot] = -1;
  }
  void clear() {
    if (n_live) for (uint32_t k = 0; k <= mask; k++) if (head[k] >= 0) drop_chain(k);
    names.clear();
  }
  // Reads whose mate would have had to start before `pos` can never be matched again
  // (records come in coordinate order): the reference evicts them as orphans (pileup.go:469-500).
  void advance(int32_t r, int32_t pos) {
    if (r != ref) {
      clear();  // finishRef evicts every stored read (pileup.go:626)
      ref = r;
      clean_upto = pos;
      return;
    }
    if (n_live == 0) {
      clean_upto = pos;
      if (!names.empty()) names.clear();
      return;
    }
    int64_t from = clean_upto;
    if ((int64_t)pos - from > (int64_t)mask + 1) from = (int64_t)pos - (int64_t)mask - 1;
    for (int64_t p = from; p < pos; p++)
      if (head[(uint32_t)p & mask] >= 0) drop_chain((uint32_t)p & mask);
    clean_upto = pos;
  }
  void compact_names() {
    std::vector<char> nn;
    nn.reserve((size_t)live_name_bytes + 4096);
    for (uint32_t k = 0; k <= mask; k++)
      for (int32_t e = head[k]; e >= 0; e = ents[(size_t)e].next) {
        StoredRead& s = ents[(size_t)e];
        const uint32_t off = (uint32_t)nn.size();
        nn.insert(nn.end(), names.begin() + s.name_off, names.begin() + s.name_off + s.name_len);
        s.name_off = off;
      }
    names.swap(nn);
  }
  void add(const StoredRead& in, const uint8_t* name) {
    if (names.size() + in.name_len > (64u << 20) && names.size() > 4 * (size_t)live_name_bytes + (1u << 20)) compact_names();
    int32_t e = free_head;
    if (e >= 0) free_head = ents[(size_t)e].next;
    else { e = (int32_t)ents.size(); ents.emplace_back(); }
    StoredRead& s = ents[(size_t)e];
    s = in;
    s.name_off = (uint32_t)names.size();
    names.insert(names.end(), (const char*)name, (const char*)name + in.name_len);
    const uint32_t slot = (uint32_t)in.key_pos & mask;
    s.next = head[slot];
    head[slot] = e;
    n_live++;
    live_name_bytes += in.name_len;
  }
  // tryRemove (firstread.go:135-152): the stored read that starts at `mate_pos`, expects its mate
  // at `pos` and has this name; among several, the one stored first (see DESIGN: known differences).
  bool try_remove(int32_t pos, int32_t mate_pos, uint64_t hash, const uint8_t* name, uint32_t name_len, StoredRead* out) {
    const uint32_t slot = (uint32_t)pos & mask;
    int32_t best = -1, best_prev = -1, prev = -1;
    for (int32_t e = head[slot]; e >= 0; prev = e, e = ents[(size_t)e].next) {
      const StoredRead& s = ents[(size_t)e];
      if (s.hash == hash && s.key_pos == pos && s.pos == mate_pos && s.name_len == name_len &&
          memcmp(names.data() + s.name_off, name, name_len) == 0 && (best < 0 || s.idx < ents[(size_t)best].idx)) {
        best = e;
        best_prev = prev;
      }
    }
    if (best < 0) return false;
    *out = ents[(size_t)best];
    if (best_prev < 0) head[slot] = ents[(size_t)best].next;
    else ents[(size_t)best_prev].next = ents[(size_t)best].next;
    live_name_bytes -= ents[(size_t)best].name_len;
    ents[(size_t)best].next = free_head;
    free_head = best;
    n_live--;
    return true;
  }
};

constexpr int kMatePartitions = 16;

struct PairItem {  // a read that can pair, as its partition's pass over the firstread table needs it
  uint64_t hash;
  const uint8_t* name;
  uint32_t k, name_len;  // index in the slice
  int32_t ref, pos, mate_pos;
  uint32_t rev;          // MATE_PAIR_REV if a pair this read completes counts on the reverse strand
  MateEntry self;
};

struct PairOut {  // a pair found by scan_slice: record indices and what the mate table gets for each
  int64_t second, first;
  MateEntry of_second, of_first;
};

inline uint64_t name_hash(const uint8_t* p, size_t n) {
  // 8 bytes at a time, multiply-mix (the hash only has to spread names; equality is checked on the bytes)
  uint64_t h = 0x9e3779b97f4a7c15ull ^ (n * 0xff51afd7ed558ccdull);
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 29;
    p += 8;
    n -= 8;
  }
  uint64_t w = 0;
  memcpy(&w, p, n);
  h = (h ^ w) * 0xff51afd7ed558ccdull;
  return h ^ (h >> 32);
}

}  // namespace

struct bp_handle {
  bp_opts o{};
  int device = 0;
  std::vector<int32_t> ref_len;
  std::vector<std::string> ref_name;
  std::vector<std::vector<int32_t>> bed;  // per reference endpoint arrays
  bool frozen = false;                    // BED fixed, windows built
  int qpass = 0;                          // dwords of per-base bits per 32 bases in the payloads (pileup_device.h): 0 when every base passes, 1 pass bits, 2 (stitch) pass + drop bits
  bool finished = false;
  std::string err;
  // A bp_submit that failed part-way has already moved the order key, the per-reference record
  // ranges, the coarse index and the mate table past records that were never committed: the
  // job cannot go on.  The error sticks: later submit / compute / finish calls return it again.
  int poison_code = 0;
  std::string poison_msg;

  // flattened intervals + windows
  std::vector<int32_t> iv_ref, iv_start, iv_end;
  std::vector<int64_t> iv_row;  // n_iv + 1
  std::vector<Window> windows;
  int64_t n_rows = 0;

  // record bookkeeping
  int64_t n_reads = 0;
  uint64_t blob_used = 8;  // in 8-byte units; starts after a 64-byte pad (the first unit of a record may read a little before it)
  int64_t algo_in_bytes = 0;
  uint64_t last_key = 0;
  int32_t max_span = 1;
  std::vector<int32_t> coarse_pos; // pos of every kCoarse-th record, for the plan kernel
  size_t coarse_uploaded = 0;
  std::vector<int64_t> ref_first;  // [n_ref+1] first record index of each reference
  std::vector<int64_t> ref_first_closed;  // copy uploaded by bp_compute (kept alive for the async copy)
  int32_t cur_ref = 0;             // references < cur_ref are complete

  // stitch: host mate table
  std::vector<MateEntry> mate;                        // per record
  bool mate_dirty = true;                             // changed since it was last copied to the device
  FirstReadTable stored[kMatePartitions];  // by name hash (scan_slice)
  std::vector<PairOut> pair_out[kMatePartitions];
  std::vector<PairItem> pair_items;   // of the slice, chunk by chunk, each chunk's in partition order
  std::vector<uint32_t> pair_start;   // [chunk][partition] -> first item; kMatePartitions + 1 entries a chunk
  std::vector<uint8_t> scan_cls;
  // bp_submit_bam: fixed fields of the call's records, pulled out of the stream
  std::vector<int32_t> x_ref, x_pos, x_mref, x_mpos;
  std::vector<uint16_t> x_flags;
  std::vector<uint8_t> x_mapq, x_lname;
  std::vector<uint32_t> x_ncig, x_lseq, x_bytes;
  std::vector<int32_t> x_span;
  std::vector<uint8_t> x_cfacts;
  std::vector<uint64_t> scan_hash;
  double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // BP_PROFILE=1: extract, layout, scan-par, scan-serial, items, ensure, copy wait, total
  double scan_s = 0, pack_s = 0;                      // host time in bp_submit so far: serial scan (order, errors, pairing) / parallel passes
  double resolve_ms = 0;                              // stitch: stitch_resolve_kernel, all its runs
  uint8_t qthr[96];

  // device
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  DevBuf d_cores, d_blob, d_mate, d_windows, d_r0, d_r1, d_iv_start, d_iv_end, d_iv_row, d_rows,
      d_err, d_sums, d_qthr, d_dbg, d_ref_first, d_iv_ref, d_pr_off, d_pr_tiles, d_pr_tile0, d_pr_tile_lo, d_pr_feat, d_seq8, d_refname, d_txt_len, d_txt_tile, d_txt_out, d_items, d_coarse, d_raw, d_pack;
  bool per_read_valid = false;
  DeflateScratch* dfl = nullptr;        // bp_render_fetch_bgzf
  const uint8_t* dfl_out = nullptr;     // the members of the text on the device (in dfl's memory), until the next bp_render_text
  size_t dfl_len = 0;
  int32_t seq_ref = -1, seq_start = 0;  // bp_set_ref_seq8
  int64_t seq_len = 0;
  bool iv_ref_uploaded = false;
  size_t txt_len = 0;  // bytes of the text bp_render_basestrand_tsv left on the device
  uint64_t n_feat = 0;
  std::vector<unsigned long long> pr_tile_base;  // per 1024 lists: entries before them
  Staging stage[2];
  int cur_stage = 0;
  bp_timing timing{};
  bool rows_valid = false;
  bool resident = false;            // a bp_finish has run: every table the kernels read is on the device
  DeviceError* host_err = nullptr;  // pinned word the allocation-free compute reads its error flags into
};

namespace {

int fail(bp_handle* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  g_last_error = buf;
  return code;
}

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, e_ == hipErrorOutOfMemory ? BP_ENOMEM : BP_EDEVICE, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                  \
  } while (0)

// Grow a device buffer keeping its first `keep` bytes.
int ensure(bp_handle* h, DevBuf& b, size_t need, size_t keep) {
  if (need <= b.cap) return 0;
  size_t cap = std::max(need, b.cap + b.cap / 2);
  cap = (cap + 255) & ~(size_t)255;
  void* np = dev_cache().take(h->device, cap, &cap);
  if (!np) {
    hipError_t e = hipMalloc(&np, cap);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      dev_cache().flush(h->device);
      e = hipMalloc(&np, cap);
    }
    HIP_TRY(h, e);
  }
  if (b.p && keep) {
    HIP_TRY(h, hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  // (work queued on the stream may still read the old buffer: it is only handed on after a sync)
  if (b.p && !keep) HIP_TRY(h, hipStreamSynchronize(h->stream));
  dev_cache().give(h->device, b.p, b.cap);
  b.p = np;
  b.cap = cap;
  return 0;
}

inline uint64_t sort_key(int32_t ref, int32_t pos) {
  return ((uint64_t)(uint32_t)ref << 32) | (uint32_t)pos;
}

// pileup.GetStrand (pileup/common.go:94-114)
inline int strand_of(int32_t ref, int32_t mate_ref, uint32_t flags) {
  if (ref != mate_ref) return 0;
  uint32_t f = flags & 0xf0u;
  if (f == 0x60u || f == 0x90u) return 1;
  if (f == 0x50u || f == 0xa0u) return 2;
  if (flags & 0x8u) {
    f &= 0x30u;
    if (f == 0) return 1;
    if (f == 0x30u) return 2;
  }
  return 0;
}

// BEDUnion.IntersectsByID (interval/bedunion.go:219-254)
inline bool bed_intersects(const std::vector<int32_t>& e, int32_t start, int32_t limit) {
  if (e.empty()) return false;
  size_t idx = (size_t)(std::lower_bound(e.begin(), e.end(), start + 1) - e.begin());
  if (idx & 1) return true;
  return idx < e.size() && e[idx] < limit;
}

// End of the last BED interval overlapping [start, limit); requires bed_intersects.
inline int32_t bed_last_overlap_end(const std::vector<int32_t>& e, int32_t limit) {
  size_t re = ((size_t)(std::lower_bound(e.begin(), e.end(), limit) - e.begin()) + 1) & ~(size_t)1;
  return e[re - 1];
}

// qualSumTable (pileup/snp/qual.go:35-58) -> per-row pass threshold.  The table is
// nondecreasing along rows and columns (checked), so "qualSum[q0][q1] >= minBaseQual" is
// "q1 >= thr[q0]".
int build_qthr(bp_handle* h) {
  static const double ln10 = 2.30258509299404568401799145468436421;
  static const double log10e = 0.43429448190325182765112891891660508;
  double ep[96];
  uint8_t t[96][96];
  for (int i = 0; i < 96; i++) ep[i] = std::exp((double)i * (-0.1 * ln10));
  for (int i = 0; i < 96; i++)
    for (int j = 0; j <= i; j++) {
      double prod = ep[i] * ep[j];
      double np = prod / ((1.0 - ep[i]) * (1.0 - ep[j]) + prod);
      double r = std::round(std::log(np) * (-10.0 * log10e));
      int q = (int)r;
      if (q < 0) q = 0;
      if (q > 95) q = 95;
      t[i][j] = t[j][i] = (uint8_t)q;
    }
  for (int i = 0; i < 96; i++) {
    int thr = 255;
    for (int j = 95; j >= 0; j--) {
      if (t[i][j] >= h->o.min_base_qual) thr = j;
      else break;
    }
    for (int j = 0; j < 96; j++)
      if ((t[i][j] >= h->o.min_base_qual) != (j >= thr))
        return fail(h, BP_EINVAL, "qualSumTable row %d is not monotone", i);
    h->qthr[i] = (uint8_t)thr;
  }
  return 0;
}

// Build flattened intervals and windows from the per-reference endpoint arrays.
int freeze_bed(bp_handle* h) {
  if (h->frozen) return 0;
  const int W = window_size();
  h->iv_ref.clear(); h->iv_start.clear(); h->iv_end.clear(); h->iv_row.clear();
  h->windows.clear();
  int64_t row = 0;
  for (size_t r = 0; r < h->bed.size(); r++) {
    const std::vector<int32_t>& e = h->bed[r];
    for (size_t k = 0; k + 1 < e.size(); k += 2) {
      if (e[k + 1] <= e[k]) continue;  // empty interval (Subset can emit [x,x))
      h->iv_ref.push_back((int32_t)r);
      h->iv_start.push_back(e[k]);
      h->iv_end.push_back(e[k + 1]);
      h->iv_row.push_back(row);
      row += (int64_t)e[k + 1] - e[k];
    }
  }
  h->iv_row.push_back(row);
  h->n_rows = row;
  const size_t n_iv = h->iv_start.size();
  size_t k = 0;
  while (k < n_iv) {
    // a window starts at the first not-yet-covered BED position
    int32_t ref = h->iv_ref[k];
    int64_t cur = h->iv_start[k];
    while (k < n_iv && h->iv_ref[k] == ref) {
      Window w{};
      w.ref_id = ref;
      w.g0 = (int32_t)cur;
      int64_t lim = cur + W;
      w.iv_lo = (uint32_t)k;
      w.row_base = h->iv_row[k] + (cur - h->iv_start[k]);
      int64_t g1 = cur;
      int64_t rows = 0;
      size_t j = k;
      while (j < n_iv && h->iv_ref[j] == ref && h->iv_start[j] < lim) {
        int64_t s = std::max<int64_t>(h->iv_start[j], cur);
        int64_t e2 = std::min<int64_t>(h->iv_end[j], lim);
        rows += e2 - s;
        g1 = e2;
        if (h->iv_end[j] > lim) break;  // interval continues into the next window
        j++;
      }
      w.g1 = (int32_t)g1;
      w.n_rows = (int32_t)rows;
      bool cont = (j < n_iv && h->iv_ref[j] == ref && h->iv_start[j] < lim && h->iv_end[j] > lim);
      w.iv_hi = (uint32_t)(cont ? j + 1 : j);
      w.full = (w.iv_hi - w.iv_lo == 1) ? 1 : 0;
      h->windows.push_back(w);
      if (cont) {
        k = j;
        cur = lim;
      } else {
        k = j;
        if (k < n_iv && h->iv_ref[k] == ref) cur = h->iv_start[k];
      }
    }
  }
  // upload
  auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
    if (int rc = ensure(h, b, std::max<size_t>(bytes, 256), 0)) return rc;
    if (bytes) HIP_TRY(h, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  if (int rc = up(h->d_windows, h->windows.data(), h->windows.size() * sizeof(Window))) return rc;
  if (int rc = up(h->d_iv_start, h->iv_start.data(), n_iv * 4)) return rc;
  if (int rc = up(h->d_iv_end, h->iv_end.data(), n_iv * 4)) return rc;
  if (int rc = up(h->d_iv_row, h->iv_row.data(), (n_iv + 1) * 8)) return rc;
  if (int rc = ensure(h, h->d_r0, std::max<size_t>(h->windows.size() * 4, 256), 0)) return rc;
  if (int rc = ensure(h, h->d_r1, std::max<size_t>(h->windows.size() * 4, 256), 0)) return rc;
  if (int rc = ensure(h, h->d_rows, std::max<size_t>((size_t)h->n_rows * 44, 256), 0)) return rc;
  if (int rc = up(h->d_qthr, h->qthr, 96)) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->frozen = true;
  return 0;
}

using bp_par::parallel_for;

int pack_threads() {
  unsigned hc = std::thread::hardware_concurrency();
  int t = (int)std::min<unsigned>(hc ? hc : 1, 16);
  return std::max(t, 1);
}

inline uint32_t cigar_op_at(const uint8_t* c, uint32_t k) {  // (a BAM stream gives CIGARs no alignment)
  uint32_t v;
  memcpy(&v, c + 4 * (size_t)k, 4);
  return v;
}

inline int32_t cigar_span(const uint8_t* c, uint32_t n, bool* simple, bool* has_bad) {
  int64_t span = 0;
  bool bad = false;
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t v = cigar_op_at(c, k), op = v & 15u, len = v >> 4;
    // sam.Cigar.Lengths(): M, D, N, =, X consume the reference
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += len;
    if (op > 5) bad = true;
  }
  *simple = (n == 1 && (cigar_op_at(c, 0) & 15u) == 0);
  *has_bad = bad;
  return span > 0x7fffffff ? 0x7fffffff : (int32_t)span;
}

// The records of one submit call: decoded columns (bp_submit), or the records as they lie in an
// inflated BAM stream (bp_submit_bam) with their fixed fields pulled out into columns first.
struct Src {
  int64_t n = 0;
  const int32_t *ref_id = nullptr, *pos = nullptr, *mate_ref_id = nullptr, *mate_pos = nullptr;
  const uint16_t* flags = nullptr;
  const uint8_t* mapq = nullptr;
  const uint32_t *n_cigar = nullptr, *l_seq = nullptr;
  const bp_record_batch* b = nullptr;  // column mode: where the payloads are
  const uint8_t* raw = nullptr;        // stream mode: record i starts (block_size field) at raw + off[i]
  const uint8_t* raw_dev = nullptr;    // ... or at raw_dev + off[i] in DEVICE memory (bp_submit_bam_device; no names, no host bytes)
  const uint32_t* off = nullptr;
  const uint8_t* l_name = nullptr;     // stream mode: l_read_name (with the NUL)
  const uint32_t* rec_bytes = nullptr; // stream mode: 4 + block_size
  const int32_t* span = nullptr;       // stream mode: cigar_span() of every record, worked out when its fields were read
  const uint8_t* cig_facts = nullptr;  // ... bit 0 simple, bit 1 has an op outside MIDNSH
  const uint8_t* names = nullptr;      // device-stream mode: the read names gathered by bp_bam_scan (no NULs) ...
  const uint32_t* name_off = nullptr;  // ... record i's at names + name_off[i]

  const uint8_t* cigar(int64_t i) const {
    return raw ? raw + off[i] + 36 + l_name[i] : (const uint8_t*)(b->cigar + b->cigar_off[i]);
  }
  bool has_names() const { return raw || names || (b && b->name && b->name_off); }
  const uint8_t* name(int64_t i) const {
    return raw ? raw + off[i] + 36 : names ? names + name_off[i] : b->name + b->name_off[i];
  }
  uint32_t name_len(int64_t i) const {
    return raw ? (l_name[i] ? l_name[i] - 1u : 0u) : names ? name_off[i + 1] - name_off[i]
                                                           : (uint32_t)(b->name_off[i + 1] - b->name_off[i]);
  }
};

// a few bytes of a record that lies on the device (error paths only)
std::vector<uint8_t> dev_bytes(const Src& S, int64_t i, size_t at, size_t n) {
  std::vector<uint8_t> v(n, 0);
  if (n && hipMemcpy(v.data(), S.raw_dev + S.off[i] + at, n, hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
  return v;
}

std::string read_name(const Src& S, int64_t i) {
  if (S.raw_dev) {
    const size_t ln = S.l_name[i] ? S.l_name[i] - 1u : 0u;
    const std::vector<uint8_t> v = dev_bytes(S, i, 36, ln);
    return std::string((const char*)v.data(), ln);
  }
  if (!S.has_names()) return std::string();
  return std::string((const char*)S.name(i), (size_t)S.name_len(i));
}

const char* ref_label(bp_handle* h, int32_t ref, char* tmp, size_t n) {
  if (ref >= 0 && (size_t)ref < h->ref_name.size() && !h->ref_name[(size_t)ref].empty())
    return h->ref_name[(size_t)ref].c_str();
  snprintf(tmp, n, "ref%d", ref);
  return tmp;
}

// What the parallel pass over a slice leaves per record for the serial one.
enum : uint8_t {
  SC_KEPT = 1,       // passes the read-level filters and the strand rule (pileup.go:691, :717-729), on a known reference
  SC_TOO_LONG = 2,   // kept, and l_seq > maxReadLen or span > maxReadSpan (pileup.go:766-772)
  SC_HITS = 4,       // kept and [pos, pos + span) meets the job's BED (pileup.go:773-777)
  SC_BAD_OP = 8,     // its CIGAR holds an op outside MIDNSH
  SC_PAIRABLE = 16,  // stitch: goes through the firstread table (firstread.go:180 does not send it on alone)
  SC_MANY_OPS = 32   // more than 65535 CIGAR ops
};

// Host pass over a slice before it is packed.  First in parallel, record by record: span, the
// read-level tests, the BED test, the name hash of reads that can pair.  Then serially, in record
// order, what carries state: the order check, the per-reference record ranges, the coarse index,
// the data errors (the first one in record order is the one reported, as the reference would
// meet it) and the firstread table.  boff/qoff/blob_base: where the packer will put each record's
// payload (a paired read's mate entry names it).
int scan_slice(bp_handle* h, const Src& S, int64_t i0, int64_t i1, std::vector<int32_t>& spans,
               std::vector<uint8_t>& simple, const std::vector<uint64_t>& boff, const std::vector<uint64_t>& qoff,
               uint32_t blob_base, int nthreads) {
  const bp_opts& o = h->o;
  const size_t ns = (size_t)(i1 - i0);
  std::vector<uint8_t>& cls = h->scan_cls;      // (scratch kept with the handle between slices)
  std::vector<uint64_t>& hashes = h->scan_hash;
  cls.assign(ns, 0);
  const size_t chunk = 2048;  // (a 16 MiB piece of a BAM stream is ~60k records: enough tasks for every thread)
  const size_t n_chunks = (ns + chunk - 1) / chunk;
  if (o.stitch) {
    hashes.resize(ns);
    if (h->pair_items.size() < ns) h->pair_items.resize(ns);
    h->pair_start.assign(n_chunks * (kMatePartitions + 1), 0);
  }
  const int32_t nref = (int32_t)h->ref_len.size();
  const auto t_par = std::chrono::steady_clock::now();
  parallel_for(n_chunks, nthreads, [&](size_t c) {
    thread_local std::vector<PairItem> found;  // the chunk's reads that can pair, in record order
    found.clear();
    for (size_t k = c * chunk; k < std::min(ns, (c + 1) * chunk); k++) {
      const int64_t i = i0 + (int64_t)k;
      const uint32_t n_cigar = S.n_cigar[i];
      if (n_cigar > 0xffffu) {
        cls[k] = SC_MANY_OPS;
        continue;
      }
      bool simp = false, has_bad = false;
      int32_t span;
      if (S.span) {
        span = S.span[i];
        simp = (S.cig_facts[i] & 1) != 0;
        has_bad = (S.cig_facts[i] & 2) != 0;
      } else {
        span = cigar_span(S.cigar(i), n_cigar, &simp, &has_bad);
      }
      spans[k] = span;
      simple[k] = simp ? 1 : 0;
      const int32_t ref = S.ref_id[i], pos = S.pos[i];
      const uint32_t flags = S.flags[i];
      if (((uint32_t)o.flag_exclude & flags) != 0 || o.mapq > (int32_t)S.mapq[i] || n_cigar == 0) continue;
      if (ref < 0 || ref >= nref) continue;  // unmapped: never in a shard
      const int strand = strand_of(ref, S.mate_ref_id[i], flags);
      if (o.strand_req != BP_STRAND_NONE) {
        if (strand != o.strand_req) continue;
      } else if (strand == 0 && !o.ignore_strand) {
        continue;
      }
      uint8_t f = SC_KEPT;
      if ((int64_t)S.l_seq[i] > (int64_t)o.max_read_len || span > o.max_read_span) {
        cls[k] = f | SC_TOO_LONG;
        continue;
      }
      const int32_t map_end = pos + span;
      if (bed_intersects(h->bed[(size_t)ref], pos, map_end)) {
        f |= SC_HITS;
        if (has_bad) f |= SC_BAD_OP;
        if (o.stitch) {
          // firstreadSNPTable.addOrRemove (firstread.go:180): processed alone right away?
          const int32_t mate_pos = S.mate_pos[i];
          if (!(strand == 0 || mate_pos >= map_end || (int64_t)mate_pos + o.max_read_span <= (int64_t)pos)) {
            f |= SC_PAIRABLE;
            PairItem it;
            it.name = S.name(i);
            it.name_len = S.name_len(i);
            it.hash = hashes[k] = name_hash(it.name, (size_t)it.name_len);
            it.k = (uint32_t)k;
            it.ref = ref;
            it.pos = pos;
            it.mate_pos = mate_pos;
            it.rev = (!o.ignore_strand && strand == 2) ? MATE_PAIR_REV : 0u;
            it.self.x = (simp ? MATE_SIMPLE : 0u) | MATE_HAS;
            it.self.blob = blob_base + (uint32_t)(boff[k] / 8);
            it.self.span_lseq = (uint32_t)span | ((uint32_t)S.l_seq[i] << 16);
            it.self.ncig_qdelta = n_cigar | ((uint32_t)((qoff[k] - boff[k]) / 8) << 16);
            found.push_back(it);
          }
        }
      }
      cls[k] = f;
    }
    if (o.stitch) {
      // into the chunk's stretch of pair_items, by partition, record order kept inside each
      uint32_t* start = h->pair_start.data() + c * (kMatePartitions + 1);
      uint32_t count[kMatePartitions] = {};
      for (const PairItem& it : found) count[(it.hash >> 58) % kMatePartitions]++;
      uint32_t at = (uint32_t)(c * chunk);
      for (int p = 0; p < kMatePartitions; p++) {
        start[p] = at;
        at += count[p];
      }
      start[kMatePartitions] = at;
      uint32_t fill[kMatePartitions];
      memcpy(fill, start, sizeof fill);
      for (const PairItem& it : found) h->pair_items[fill[(it.hash >> 58) % kMatePartitions]++] = it;
    }
  });
  const auto t_ser = std::chrono::steady_clock::now();
  h->pack_s += std::chrono::duration<double>(t_ser - t_par).count();
  h->prof[2] += std::chrono::duration<double>(t_ser - t_par).count();
  struct Timer {
    bp_handle* h;
    std::chrono::steady_clock::time_point t0;
    ~Timer() {
      const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      h->scan_s += d;
      h->prof[3] += d;
    }
  };
  std::unique_ptr<Timer> timer(new Timer{h, t_ser});

  for (size_t k = 0; k < ns; k++) {
    const int64_t i = i0 + (int64_t)k;
    const int64_t gidx = h->n_reads + (int64_t)k;
    const int32_t ref = S.ref_id[i], pos = S.pos[i];
    const uint64_t key = sort_key(ref, pos);
    if (key < h->last_key)
      return fail(h, BP_EUNSORTED, "bp_submit: record %lld (ref %d pos %d) is out of coordinate order",
                  (long long)gidx, ref, pos);
    h->last_key = key;
    {
      // records of reference r occupy [ref_first[r], ref_first[r+1]); unmapped (-1) sort last
      const int32_t rclamp = (ref < 0 || ref >= nref) ? nref : ref;
      while (h->cur_ref < rclamp) h->ref_first[(size_t)++h->cur_ref] = gidx;
    }
    if ((gidx & (kCoarse - 1)) == 0) h->coarse_pos.push_back(pos);
    const uint8_t f = cls[k];
    if (!f) continue;
    if (f & SC_MANY_OPS)
      return fail(h, BP_EINVAL, "bp_submit: record with %u CIGAR ops (limit 65535)", S.n_cigar[i]);
    const int32_t span = spans[k];
    char tmp[32];
    if (f & SC_TOO_LONG) {
      // ring-buffer sizing assumptions (pileup.go:766-772), checked before the BED test
      if ((int64_t)S.l_seq[i] > (int64_t)o.max_read_len)
        return fail(h, BP_EDATA,
                    "pileupMutable.processShard: maxReadLen is %d, but read %s at %s:%d has length %u",
                    o.max_read_len, read_name(S, i).c_str(), ref_label(h, ref, tmp, sizeof tmp), pos,
                    S.l_seq[i]);
      return fail(h, BP_EDATA,
                  "pileupMutable.processShard: maxReadSpan is %d, but read %s at %s:%d has span %d",
                  o.max_read_span, read_name(S, i).c_str(), ref_label(h, ref, tmp, sizeof tmp), pos,
                  span);
    }
    if (span > h->max_span) h->max_span = span;
    if (f & SC_BAD_OP) {
      // alignRelevantBases (pileup.go:355-359) fails at the first op outside MIDNSH that its
      // walk reaches, i.e. one starting before the end of the last overlapping interval.
      std::vector<uint8_t> cg_dev;
      if (S.raw_dev) cg_dev = dev_bytes(S, i, 36 + (size_t)S.l_name[i], 4 * (size_t)S.n_cigar[i]);
      const uint8_t* cg = S.raw_dev ? cg_dev.data() : S.cigar(i);
      const int32_t last_end = bed_last_overlap_end(h->bed[(size_t)ref], pos + span);
      int32_t rp = pos;
      for (uint32_t c = 0; c < S.n_cigar[i]; c++) {
        const uint32_t v = cigar_op_at(cg, c), op = v & 15u, len = v >> 4;
        if (rp >= last_end) break;
        if (op == 0 || op == 2 || op == 3) rp += (int32_t)len;
        else if (op > 5)
          return fail(h, BP_EDATA, "alignRelevantBases: unexpected CIGAR code %u%c (read %s at %s:%d)",
                      len, "MIDNSHP=XB??????"[op], read_name(S, i).c_str(),
                      ref_label(h, ref, tmp, sizeof tmp), pos);
      }
    }
  }
  timer.reset();
  if (!o.stitch) return 0;
  // firstreadSNPTable.addOrRemove (firstread.go:176-204) for the reads that can pair.  Two reads
  // meet in the table only if they have the same name, so the table is kept as kMatePartitions
  // tables by name hash, each worked by one thread over the slice's records in order: within a
  // partition everything happens in the order one table would see it (so "the one stored first"
  // still wins), and partitions never look at each other's reads.  Evictions are per table and lazy
  // (a stored read whose mate would have had to start before the current position can never be
  // taken again, whenever it is dropped).  The parallel pass above has listed each partition's reads
  // with all that is needed of them (a partition that picked its reads out of the record columns
  // would pull every cache line of every column: 16 threads x the whole slice); the pairs a partition
  // finds are listed too and entered afterwards.
  const auto t_pair = std::chrono::steady_clock::now();
  parallel_for((size_t)kMatePartitions, nthreads, [&](size_t part) {
    FirstReadTable& T = h->stored[part];
    std::vector<PairOut>& out = h->pair_out[part];
    out.clear();
    for (size_t c = 0; c < n_chunks; c++) {
      const uint32_t* start = h->pair_start.data() + c * (kMatePartitions + 1);
      for (uint32_t x = start[part]; x < start[part + 1]; x++) {
        const PairItem& it = h->pair_items[x];
        const int64_t gidx = h->n_reads + (int64_t)it.k;
        T.advance(it.ref, it.pos);
        StoredRead me{};
        me.hash = it.hash;
        me.key_pos = it.mate_pos;
        me.pos = it.pos;
        me.idx = gidx;
        me.name_len = it.name_len;
        me.self = it.self;
        if (it.pos >= it.mate_pos) {
          StoredRead a;
          if (T.try_remove(it.pos, it.mate_pos, it.hash, it.name, it.name_len, &a)) {
            // the pair is counted on the strand of the read that found its mate (pileup.go:799-802)
            MateEntry eb = a.self, ea = me.self;
            eb.x |= ((uint32_t)(a.pos - it.pos) & 0x1ffffu) | MATE_SECOND | it.rev;
            ea.x |= ((uint32_t)(it.pos - a.pos) & 0x1ffffu) | it.rev;
            out.push_back({gidx, a.idx, eb, ea});
            continue;
          }
          if (it.pos != it.mate_pos) continue;  // missing mate: processed alone
        }
        T.add(me, it.name);
      }
    }
  });
  // The entries go into the mate table by record range, not by partition: partitions work through
  // the same records at the same time, and their pairs' entries lie side by side in the table
  // (16 bytes each), so writing them from the partition threads has every cache line of the table
  // bounce between cores.  Each list is in the order of the second read's index.
  constexpr size_t kRanges = 16;
  parallel_for(kRanges, nthreads, [&](size_t r) {
    const int64_t lo = h->n_reads + (int64_t)(ns * r / kRanges), hi = h->n_reads + (int64_t)(ns * (r + 1) / kRanges);
    for (int part = 0; part < kMatePartitions; part++) {
      const std::vector<PairOut>& out = h->pair_out[part];
      auto it = std::lower_bound(out.begin(), out.end(), lo, [](const PairOut& a, int64_t x) { return a.second < x; });
      for (; it != out.end() && it->second < hi; ++it) {
        h->mate[(size_t)it->second] = it->of_second;
        h->mate[(size_t)it->first] = it->of_first;
      }
    }
  });
  h->pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pair).count();
  return 0;
}

// Pinned staging of finished jobs, for the next job (see DevCache; pinned memory is not tied to a device).
struct StagePool {
  std::mutex mu;
  std::vector<Staging> v;
};
StagePool& stage_pool() {
  static StagePool* p = new StagePool();
  return *p;
}
void free_stage(Staging& s) {
  if (s.cores) (void)hipHostFree(s.cores);
  if (s.blob) (void)hipHostFree(s.blob);
  if (s.items) (void)hipHostFree(s.items);
  if (s.raw) (void)hipHostFree(s.raw);
  if (s.done) (void)hipEventDestroy(s.done);
  s = Staging();
}

int acquire_stage(bp_handle* h, Staging** out, bool stream_mode, size_t raw_bytes) {
  Staging& s = h->stage[h->cur_stage];
  h->cur_stage ^= 1;
  if (!s.done) {
    // a finished job's staging that already has what this kind of submit needs, if there is one
    StagePool& sp = stage_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    for (size_t k = 0; k < sp.v.size(); k++)
      if (sp.v[k].dev == h->device && (stream_mode ? sp.v[k].items != nullptr : sp.v[k].cores != nullptr)) {
        s = sp.v[k];
        sp.v.erase(sp.v.begin() + (ptrdiff_t)k);
        break;
      }
  }
  if (!s.done) {
    HIP_TRY(h, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    s.dev = h->device;
  }
  if (s.busy) {
    HIP_TRY(h, hipEventSynchronize(s.done));
    s.busy = false;
  }
  if (!stream_mode && !s.cores) {
    HIP_TRY(h, hipHostMalloc((void**)&s.cores, kStageReads * sizeof(ReadCore), hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc((void**)&s.blob, kStageBlob, hipHostMallocDefault));
  }
  if (stream_mode && !s.items)
    HIP_TRY(h, hipHostMalloc((void**)&s.items, kStageReads * sizeof(PackItem), hipHostMallocDefault));
  if (raw_bytes > s.raw_cap) {
    if (s.raw) (void)hipHostFree(s.raw);
    s.raw = nullptr;
    s.raw_cap = 0;
    const size_t cap = std::max<size_t>(raw_bytes + raw_bytes / 4, 32u << 20);
    HIP_TRY(h, hipHostMalloc((void**)&s.raw, cap, hipHostMallocDefault));
    s.raw_cap = cap;
  }
  *out = &s;
  return 0;
}

// Is p memory the device can copy from by itself (hipHostMalloc / hipHostRegister)?
bool host_pinned(const void* p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

constexpr int kMaxDevices = 64;
uint8_t* g_scratch_ptr[kMaxDevices];  // per device: where that GPU's copy of `scratch` lives
std::mutex g_scratch_mu;

// Called with `device` current.
int resolve_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return BP_OK;  // bp_compute_static refuses such a device
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return BP_OK;
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p)
    return fail(nullptr, BP_EDEVICE, "bp_init: hipGetSymbolAddress(scratch) on device %d: %s", device,
                e != hipSuccess ? hipGetErrorString(e) : "null address");
  g_scratch_ptr[device] = (uint8_t*)p;
  return BP_OK;
}

}  // namespace

extern "C" {

int bp_init(int32_t device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, BP_EDEVICE, "bp_init: no HIP device (%s); libbiopileup has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return fail(nullptr, BP_EINVAL, "bp_init: device %d of %d", device, n);
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BP_EDEVICE, "hipSetDevice: %s", hipGetErrorString(e));
  // (a GPU whose scratch address cannot be found still serves every entry point but
  // bp_compute_static, which then says so)
  (void)resolve_scratch(device);
  return BP_OK;
}

int bp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, BP_EDEVICE, "bp_device_count: no HIP device (%s); libbiopileup has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  return n;
}

int bp_create(const bp_opts* opts, int32_t n_ref, const int32_t* ref_len, const char* const* ref_name,
              bp_handle** out) {
  if (!opts || !out || n_ref < 0 || (n_ref > 0 && !ref_len)) return fail(nullptr, BP_EINVAL, "bp_create: null argument");
  *out = nullptr;
  // Pileup() argument validation (pileup.go:975-988), newQualPassTable (qual.go:62-66)
  if (opts->clip < 0 || opts->clip * 2 >= opts->max_read_len)
    return fail(nullptr, BP_EINVAL, "Pileup: invalid clip= argument");
  if (opts->max_read_len > opts->max_read_span)
    return fail(nullptr, BP_EINVAL, "Pileup: max-read-len= argument cannot be larger than max-read-span= argument");
  if (opts->min_base_qual >= 96 || opts->min_base_qual < 0)
    return fail(nullptr, BP_EINVAL, "newQualPassTable: minBaseQual too large");
  if (opts->max_read_span > 0xffff) return fail(nullptr, BP_EINVAL, "bp_create: max_read_span above 65535");
  if (opts->strand_req < 0 || opts->strand_req > 2) return fail(nullptr, BP_EINVAL, "bp_create: strand_req");
  if (int rc = bp_init(opts->device)) return rc;
  bp_handle* h = new bp_handle();
  h->o = *opts;
  h->qpass = opts->stitch ? 2 : opts->min_base_qual > 0 ? 1 : 0;
  h->device = opts->device;
  h->ref_len.assign(ref_len, ref_len + n_ref);
  h->ref_name.resize((size_t)n_ref);
  if (ref_name)
    for (int i = 0; i < n_ref; i++)
      if (ref_name[i]) h->ref_name[(size_t)i] = ref_name[i];
  h->bed.resize((size_t)n_ref);
  h->ref_first.assign((size_t)n_ref + 1, 0);
  if (int rc = build_qthr(h)) { delete h; return rc; }
  for (FirstReadTable& t : h->stored) t.init(opts->max_read_span);
  auto bail = [&](const char* what, hipError_t e) {
    int rc = fail(nullptr, BP_EDEVICE, "%s: %s", what, hipGetErrorString(e));
    bp_destroy(h);
    return rc;
  };
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return bail("hipStreamCreate", e);
  for (int i = 0; i < 3; i++) {
    e = hipEventCreate(&h->ev[i]);
    if (e != hipSuccess) return bail("hipEventCreate", e);
  }
  e = hipHostMalloc((void**)&h->host_err, 256, hipHostMallocDefault);
  if (e != hipSuccess) return bail("hipHostMalloc", e);
  *out = h;
  return BP_OK;
}

int bp_reserve(bp_handle* h, int64_t n_records, int64_t payload_bytes) {
  if (!h || n_records < 0 || payload_bytes < 0) return fail(h, BP_EINVAL, "bp_reserve: bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  // payload grows by padding: cigar to even count, qual/seq to whole 8-base units
  if (h->o.stitch) h->mate.reserve((size_t)n_records);
  // (and by one pass bit per base where the job has them: l_seq <= 2/3 of the payload, a dword of
  // padding per record)
  size_t blob = (size_t)payload_bytes + (size_t)n_records * 24 + 4096;
  if (h->qpass) blob += (size_t)h->qpass * ((size_t)payload_bytes / 8 + (size_t)n_records * 8);
  if (int rc = ensure(h, h->d_cores, (size_t)n_records * sizeof(ReadCore) + 256, (size_t)h->n_reads * sizeof(ReadCore))) return rc;
  if (int rc = ensure(h, h->d_blob, blob, (size_t)h->blob_used * 8)) return rc;
  return BP_OK;
}

int bp_set_bed(bp_handle* h, int32_t ref_id, const int32_t* endpoints, int32_t n_endpoints) {
  if (!h) return fail(h, BP_EINVAL, "bp_set_bed: null handle");
  if (h->frozen) return fail(h, BP_EINVAL, "bp_set_bed: called after the first bp_submit");
  if (ref_id < 0 || (size_t)ref_id >= h->bed.size() || n_endpoints < 0 || (n_endpoints & 1) ||
      (n_endpoints > 0 && !endpoints))
    return fail(h, BP_EINVAL, "bp_set_bed: bad argument (ref %d, %d endpoints)", ref_id, n_endpoints);
  for (int32_t k = 0; k + 1 < n_endpoints; k++)
    if (endpoints[k] > endpoints[k + 1] || (endpoints[k] == endpoints[k + 1] && (k & 1)))
      return fail(h, BP_EINVAL, "bp_set_bed: endpoints of ref %d are not sorted disjoint intervals", ref_id);
  if (n_endpoints && endpoints[0] < 0) return fail(h, BP_EINVAL, "bp_set_bed: negative coordinate");
  h->bed[(size_t)ref_id].assign(endpoints, endpoints + n_endpoints);
  return BP_OK;
}

// One submit call, either kind.  (Runs between `extern "C" {` and its end: plain C++ linkage is
// all it needs, it is static.)
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct ProfTimer {
  double* acc;
  double t0;
  explicit ProfTimer(double* a) : acc(a), t0(now_s()) {}
  ~ProfTimer() { *acc += now_s() - t0; }
};

static int submit_records(bp_handle* h, const Src& S, const char* who) {
  ProfTimer prof_total(&h->prof[7]);
  if (h->finished) return fail(h, BP_EINVAL, "%s: called after bp_finish", who);
  if (S.n < 0) return fail(h, BP_EINVAL, "%s: negative record count", who);
  if (h->o.stitch && S.n > 0 && !S.has_names())
    return fail(h, BP_EINVAL, "%s: stitch mode needs read names (firstread.go:135-152)", who);
  if (h->poison_code) return fail(h, h->poison_code, "%s", h->poison_msg.c_str());
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = freeze_bed(h)) return rc;
  h->rows_valid = false;
  struct Poison {  // any error return below leaves the handle's bookkeeping half advanced
    bp_handle* h;
    bool armed = true;
    ~Poison() {
      if (armed && !h->poison_code) {
        h->poison_code = BP_EINVAL;
        h->poison_msg = "bp_submit: an earlier bp_submit of this job failed (" + h->err + "); the job cannot continue";
      }
    }
  } poison{h};
  const int nthreads = pack_threads();
  const bool stream_mode = S.raw != nullptr || S.raw_dev != nullptr;
  const bool pinned_src = S.raw != nullptr && S.n > 0 && host_pinned(S.raw + S.off[0]);
  // a copy out of the caller's own memory is in flight: the caller's memory is the caller's again
  // when the call returns, however it returns
  bool wait_for_copy = false;
  struct CopyFence {
    bp_handle* h;
    bool* armed;
    ~CopyFence() {
      if (!*armed) return;
      ProfTimer pt(&h->prof[6]);
      (void)hipStreamSynchronize(h->stream);
    }
  } copy_fence{h, &wait_for_copy};
  // (the calling thread's own and kept between calls: a job's handle lives for one or two submits of
  // a million records, and megabytes of fresh vectors per call are page faults and zero fills)
  // (through references: a lambda run on a worker thread would otherwise name the WORKER's instance)
  thread_local std::vector<int32_t> tl_spans;
  thread_local std::vector<uint8_t> tl_simple;
  thread_local std::vector<uint64_t> tl_boff, tl_qoff;
  thread_local std::vector<uint32_t> tl_gsrc, tl_sz_a, tl_sz_q;
  std::vector<int32_t>& spans = tl_spans;
  std::vector<uint8_t>& simple = tl_simple;
  std::vector<uint64_t>&boff = tl_boff, &qoff = tl_qoff;
  std::vector<uint32_t>& gsrc = tl_gsrc;
  // a record longer than the core can say (0xffff) passes no max-read-len: no payload kept
  auto plen = [](uint32_t l) { return l > 0xffffu ? 0u : l; };
  // payload part sizes of every record of the call, on all threads: the layout loop below is
  // serial (a group closes when it is full) and otherwise spends its time working these out
  std::vector<uint32_t>& sz_a = tl_sz_a;
  std::vector<uint32_t>& sz_q = tl_sz_q;
  {
    const double t_sz = now_s();
    sz_a.resize((size_t)S.n);
    sz_q.resize((size_t)S.n);
    const size_t szchunk = 16384;
    parallel_for(((size_t)S.n + szchunk - 1) / szchunk, nthreads, [&](size_t c) {
      for (size_t i = c * szchunk; i < std::min((size_t)S.n, (c + 1) * szchunk); i++) {
        // (a record's parts stay far below 4 GiB: l_seq <= 0xffff here, n_cigar <= 0xffff is checked by the scan)
        sz_a[i] = (uint32_t)std::min<uint64_t>(base_part_bytes(S.n_cigar[i], plen(S.l_seq[i]), h->qpass), 0xffffffffull);
        sz_q[i] = (uint32_t)qual_part_bytes(plen(S.l_seq[i]));
      }
    });
    h->prof[1] += now_s() - t_sz;
  }
  int64_t i0 = 0;
  while (i0 < S.n) {
    // slice bounded by staging capacity
    int64_t i1 = i0;
    uint64_t bytes = 0;
    const double t_layout = now_s();
    boff.resize(std::min<size_t>((size_t)(S.n - i0), kStageReads));
    qoff.resize(boff.size());
    while (i1 < S.n && (size_t)(i1 - i0) < kStageReads) {
      // one group: base parts back to back, then the qual parts
      int64_t g1 = i1;
      uint64_t ga = 0, gq = 0;
      while (g1 < S.n && g1 - i1 < kGroupRecs && (size_t)(g1 - i0) < kStageReads) {
        const uint64_t a = sz_a[(size_t)g1], q = sz_q[(size_t)g1];
        if (a + q > kGroupBytes) return fail(h, BP_EINVAL, "%s: record payload of %llu bytes", who, (unsigned long long)(a + q));
        if (ga + gq + a + q > kGroupBytes) break;
        ga += a;
        gq += q;
        g1++;
      }
      if (bytes + ga + gq > kStageBlob) break;
      // the stream bytes of a slice travel in one piece too
      if (stream_mode && g1 > i0 && (uint64_t)S.off[g1 - 1] + S.rec_bytes[g1 - 1] - S.off[i0] > kStageBlob && i1 > i0) break;
      uint64_t ao = bytes, qo = bytes + ga;
      for (int64_t i = i1; i < g1; i++) {
        boff[(size_t)(i - i0)] = ao;
        qoff[(size_t)(i - i0)] = qo;
        ao += sz_a[(size_t)i];
        qo += sz_q[(size_t)i];
      }
      bytes += ga + gq;
      i1 = g1;
    }
    const size_t ns = (size_t)(i1 - i0);
    boff.resize(ns);
    qoff.resize(ns);
    h->prof[1] += now_s() - t_layout;
    if ((uint64_t)h->blob_used + bytes / 8 > 0xffffffffull)
      return fail(h, BP_ENOMEM, "%s: more than 32 GiB of record payload in one job", who);
    if (h->n_reads + (int64_t)ns > 0x7fffffff)
      return fail(h, BP_ENOMEM, "%s: more than 2^31-1 records in one job", who);
    spans.assign(ns, 0);
    simple.assign(ns, 0);
    if (h->o.stitch) {
      h->mate.resize((size_t)h->n_reads + ns, MateEntry{0, 0, 0, 0});
      h->mate_dirty = true;
    }
    // The record bytes go to the device as they are and are packed there (pack_kernels.hip):
    // straight out of the caller's memory when that is pinned and most of the run is wanted, else
    // the wanted records are first gathered into pinned staging.  The copy is started before the
    // host scan, which it then runs beside.
    Staging* st = nullptr;
    bool direct = false;
    uint64_t raw_first = 0;
    if (stream_mode) {
      const uint64_t extent = (uint64_t)S.off[i1 - 1] + S.rec_bytes[i1 - 1] - S.off[i0];
      raw_first = S.off[i0];
      if (extent > 0xffffffffull) return fail(h, BP_EINVAL, "%s: records spread over more than 4 GiB", who);
      uint64_t wanted = 0;
      for (int64_t i = i0; i < i1; i++) wanted += S.rec_bytes[i];
      direct = S.raw_dev != nullptr || (pinned_src && wanted * 4 >= extent * 3);
      {
        ProfTimer pt(&h->prof[5]);
        if (int rc = acquire_stage(h, &st, true, direct ? 0 : (size_t)wanted)) return rc;
        if (!S.raw_dev)
          if (int rc = ensure(h, h->d_raw, (size_t)(direct ? extent : wanted) + 64, 0)) return rc;
        // (always a whole slice's worth, 16 MB: a size the cache of finished jobs' buffers always holds)
        if (int rc = ensure(h, h->d_pack, kStageReads * sizeof(PackItem), 0)) return rc;
      }
      if (S.raw_dev) {
        // the records are on the device already: the pack kernel reads them where they lie, and the
        // caller's buffer is the caller's again when the call returns (see CopyFence)
        wait_for_copy = true;
      } else if (direct) {
        HIP_TRY(h, hipMemcpyAsync(h->d_raw.p, S.raw + raw_first, (size_t)extent, hipMemcpyHostToDevice, h->stream));
        wait_for_copy = true;
      } else {
        const auto t_g = std::chrono::steady_clock::now();
        gsrc.resize(ns);
        uint64_t at = 0;
        for (size_t k = 0; k < ns; k++) {
          gsrc[k] = (uint32_t)at;
          at += S.rec_bytes[i0 + (int64_t)k];
        }
        const size_t gchunk = 1024;
        parallel_for((ns + gchunk - 1) / gchunk, nthreads, [&](size_t c) {
          for (size_t k = c * gchunk; k < std::min(ns, (c + 1) * gchunk); k++)
            memcpy(st->raw + gsrc[k], S.raw + S.off[i0 + (int64_t)k], S.rec_bytes[i0 + (int64_t)k]);
        });
        h->pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_g).count();
        HIP_TRY(h, hipMemcpyAsync(h->d_raw.p, st->raw, (size_t)wanted, hipMemcpyHostToDevice, h->stream));
      }
    }
    if (int rc = scan_slice(h, S, i0, i1, spans, simple, boff, qoff, (uint32_t)h->blob_used, nthreads)) return rc;
    const uint32_t blob_base = (uint32_t)h->blob_used;
    // what the record-level tests come to under the job's options (ReadCore.aux)
    auto aux_of = [&](int64_t i, size_t k) {
      const int strand = strand_of(S.ref_id[i], S.mate_ref_id[i], S.flags[i]);
      const bool keep = ((uint32_t)h->o.flag_exclude & S.flags[i]) == 0 && (int32_t)S.mapq[i] >= h->o.mapq &&
                        S.n_cigar[i] != 0 &&
                        (h->o.strand_req != BP_STRAND_NONE ? strand == h->o.strand_req
                                                           : (strand != 0 || h->o.ignore_strand));
      return (uint8_t)((simple[k] ? AUX_SIMPLE : 0) | (S.ref_id[i] == S.mate_ref_id[i] ? AUX_MATE_SAME : 0) |
                       (keep ? AUX_KEEP : 0) | (strand << AUX_STRAND_SHIFT));
    };
    {
      ProfTimer pt(&h->prof[5]);
      if (int rc = ensure(h, h->d_cores, ((size_t)h->n_reads + ns) * sizeof(ReadCore), (size_t)h->n_reads * sizeof(ReadCore))) return rc;
      if (int rc = ensure(h, h->d_blob, ((size_t)h->blob_used) * 8 + bytes + 64, (size_t)h->blob_used * 8)) return rc;
    }
    std::atomic<int64_t> algo(0);
    const size_t chunk = 4096;
    if (stream_mode) {
      const auto t_pack = std::chrono::steady_clock::now();
      parallel_for((ns + chunk - 1) / chunk, nthreads, [&](size_t c) {
        int64_t local = 0;
        for (size_t k = c * chunk; k < std::min(ns, (c + 1) * chunk); k++) {
          const int64_t i = i0 + (int64_t)k;
          PackItem& it = st->items[k];
          it.src = direct ? (uint32_t)(S.off[i] - raw_first) : gsrc[k];
          it.blob = blob_base + (uint32_t)(boff[k] / 8);
          it.qdelta = (uint16_t)((qoff[k] - boff[k]) / 8);
          it.span = (uint16_t)std::min<int32_t>(spans[k], 0xffff);
          it.aux = aux_of(i, k);
          it.l_name = S.l_name[i];
          it.pad = 0;
          local += 32 + 4 * (int64_t)S.n_cigar[i] + ((int64_t)S.l_seq[i] + 1) / 2 + S.l_seq[i];
        }
        algo.fetch_add(local);
      });
      h->pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack).count();
      h->prof[4] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack).count();
      HIP_TRY(h, hipMemcpyAsync(h->d_pack.p, st->items, ns * sizeof(PackItem), hipMemcpyHostToDevice, h->stream));
      launch_pack(S.raw_dev ? S.raw_dev + raw_first : (const uint8_t*)h->d_raw.p, (const PackItem*)h->d_pack.p, (int64_t)ns, (ReadCore*)h->d_cores.p + h->n_reads,
                  (uint8_t*)h->d_blob.p, h->qpass, h->o.min_base_qual, h->o.clip, h->stream);
      HIP_TRY(h, hipGetLastError());
    } else {
      const bp_record_batch* b = S.b;
      if (int rc = acquire_stage(h, &st, false, 0)) return rc;
      const auto t_pack = std::chrono::steady_clock::now();
      parallel_for((ns + chunk - 1) / chunk, nthreads, [&](size_t c) {
        int64_t local = 0;
        for (size_t k = c * chunk; k < std::min(ns, (c + 1) * chunk); k++) {
          const int64_t i = i0 + (int64_t)k;
          ReadCore& rc = st->cores[k];
          const uint32_t n_cigar = b->n_cigar[i], l = plen(b->l_seq[i]);
          rc.pos = b->pos[i];
          rc.flags = b->flags[i];
          rc.mapq = b->mapq[i];
          rc.aux = aux_of(i, k);
          rc.n_cigar = (uint16_t)n_cigar;
          rc.span = (uint16_t)std::min<int32_t>(spans[k], 0xffff);
          rc.l_seq = (uint16_t)std::min<uint32_t>(b->l_seq[i], 0xffffu);
          rc.qdelta = (uint16_t)((qoff[k] - boff[k]) / 8);
          rc.blob = blob_base + (uint32_t)(boff[k] / 8);
          for (uint32_t c = 0; c < 3; c++) rc.cig[c] = c < n_cigar ? b->cigar[b->cigar_off[i] + c] : 0u;
          uint8_t* d = st->blob + boff[k];
          const uint32_t ncp = (n_cigar + 1) & ~1u;
          memcpy(d, b->cigar + b->cigar_off[i], (size_t)n_cigar * 4);
          if (ncp != n_cigar) memset(d + (size_t)n_cigar * 4, 0, 4);
          d += (size_t)ncp * 4;
          const uint32_t U = blob_units(l);
          uint8_t* dq = st->blob + qoff[k];
          memcpy(dq, b->qual + b->qual_off[i], l);
          memset(dq + l, 0, (size_t)U * 8 - l);
          const uint32_t sb = (l + 1) / 2;
          // device layout keeps the first base of a byte in the LOW nibble (BAM: high), so 8
          // bases are one little-endian dword with base k at bits 4k
          {
            const uint8_t* sp = b->seq + b->seq_off[i];
            for (uint32_t x = 0; x < sb; x++) d[x] = (uint8_t)((sp[x] << 4) | (sp[x] >> 4));
          }
          const uint64_t tail = base_part_bytes(n_cigar, l, h->qpass) - (uint64_t)ncp * 4 - sb;
          memset(d + sb, 0, (size_t)tail);
          if (h->qpass) {
            // one bit per base: its quality, after clipQuals (pileup_support.go:51-69), passes addBase's
            // test (pileup.go:227); the counting kernel reads these instead of the quals
            // (8 quals at a time from the zero-padded copy just made: byte flags by SWAR compare,
            // gathered into one byte by a multiply)
            uint8_t* m = d + 4 * (size_t)U;
            const uint32_t clip = (uint32_t)h->o.clip, minq = (uint32_t)h->o.min_base_qual;
            const uint64_t H = 0x8080808080808080ull, m8 = minq * 0x0101010101010101ull;
            const bool all_clipped = clip > 0 && 2 * clip >= l;
            for (uint32_t j = 0; j < U; j++) {
              uint64_t w;
              memcpy(&w, dq + 8 * (size_t)j, 8);
              const uint64_t f = (((w | H) - m8) | w) & H;
              uint32_t bits = (uint32_t)(((f >> 7) * 0x0102040810204080ull) >> 56);
              if (clip > 0) {
                const uint32_t x0 = 8 * j, n = l - x0 < 8 ? l - x0 : 8;  // bases of this byte
                uint32_t cm = 0;  // the clipped ones among them
                if (all_clipped) {
                  cm = (1u << n) - 1u;
                } else {
                  if (x0 < clip) cm |= (1u << (clip - x0 < 8 ? clip - x0 : 8)) - 1u;
                  if (x0 + 8 > l - clip) cm |= ((1u << n) - 1u) & ~((1u << (l - clip > x0 ? l - clip - x0 : 0)) - 1u);
                }
                bits = minq <= 2 ? (bits | cm) : (bits & ~cm);
              }
              if (8 * j + 8 > l) bits &= (1u << (l - 8 * j)) - 1u;  // pad bits 0, as pack_records leaves them
              // (stitch: a dword of drop bits, zero here, follows every dword of pass bits)
              m[h->qpass == 2 ? (size_t)(j >> 2) * 8 + (j & 3u) : (size_t)j] = (uint8_t)bits;
            }
          }
          local += 32 + 4 * (int64_t)n_cigar + ((int64_t)b->l_seq[i] + 1) / 2 + b->l_seq[i];
        }
        algo.fetch_add(local);
      });
      h->pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack).count();
      HIP_TRY(h, hipMemcpyAsync((ReadCore*)h->d_cores.p + h->n_reads, st->cores, ns * sizeof(ReadCore),
                                hipMemcpyHostToDevice, h->stream));
      if (bytes)
        HIP_TRY(h, hipMemcpyAsync((uint8_t*)h->d_blob.p + (size_t)h->blob_used * 8, st->blob, bytes,
                                  hipMemcpyHostToDevice, h->stream));
    }
    h->algo_in_bytes += algo.load();
    HIP_TRY(h, hipEventRecord(st->done, h->stream));
    st->busy = true;
    h->n_reads += (int64_t)ns;
    h->blob_used += bytes / 8;
    i0 = i1;
  }
  if (wait_for_copy) {
    wait_for_copy = false;
    ProfTimer pt(&h->prof[6]);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  poison.armed = false;
  return BP_OK;
}

// Nothing may propagate out of an extern "C" entry point: an exception from the host passes (a
// vector that cannot grow, on this thread or on a pool thread, par.h) ends the call with BP_ENOMEM.
static int guarded_submit(bp_handle* h, const Src& S, const char* who) {
  try {
    return submit_records(h, S, who);
  } catch (const std::bad_alloc&) {
    return fail(h, BP_ENOMEM, "%s: out of host memory", who);
  } catch (const std::exception& ex) {
    return fail(h, BP_ENOMEM, "%s: %s", who, ex.what());
  }
}

int bp_submit(bp_handle* h, const bp_record_batch* b) {
  if (!h || !b) return fail(h, BP_EINVAL, "bp_submit: null argument");
  Src S;
  S.n = b->n;
  S.ref_id = b->ref_id;
  S.pos = b->pos;
  S.mate_ref_id = b->mate_ref_id;
  S.mate_pos = b->mate_pos;
  S.flags = b->flags;
  S.mapq = b->mapq;
  S.n_cigar = b->n_cigar;
  S.l_seq = b->l_seq;
  S.b = b;
  return guarded_submit(h, S, "bp_submit");
}

int bp_submit_bam(bp_handle* h, const uint8_t* bam, size_t n_bytes, const uint32_t* rec_off, int64_t n) {
  if (!h || n < 0 || (n > 0 && (!bam || !rec_off))) return fail(h, BP_EINVAL, "bp_submit_bam: bad argument");
  if (n_bytes > 0xffffffffull) return fail(h, BP_EINVAL, "bp_submit_bam: more than 4 GiB in one call");
  try {
  // The fixed fields of every record (SAM spec 4.2; gbam.Unmarshal, encoding/bam/unmarshal.go:126-224)
  // into columns, on all packing threads; a record that does not fit its block, or the bytes
  // given, is refused before anything of the call is taken.
  const size_t N = (size_t)n;
  h->x_ref.resize(N); h->x_pos.resize(N); h->x_mref.resize(N); h->x_mpos.resize(N);
  h->x_flags.resize(N); h->x_mapq.resize(N); h->x_lname.resize(N);
  h->x_ncig.resize(N); h->x_lseq.resize(N); h->x_bytes.resize(N);
  h->x_span.resize(N); h->x_cfacts.resize(N);
  std::atomic<int64_t> bad(-1);
  const auto t0 = std::chrono::steady_clock::now();
  const size_t chunk = 2048;
  parallel_for((N + chunk - 1) / chunk, pack_threads(), [&](size_t c) {
    for (size_t k = c * chunk; k < std::min(N, (c + 1) * chunk); k++) {
      const uint64_t at = rec_off[k];
      const uint64_t lim = k + 1 < N ? rec_off[k + 1] : n_bytes;  // records in stream order, not overlapping
      uint32_t w[9];
      bool ok = at + 36 <= lim && lim <= n_bytes;
      if (ok) {
        memcpy(w, bam + at, 36);
        const uint32_t block_size = w[0], l_name = w[3] & 0xffu, n_cigar = w[4] & 0xffffu, l_seq = w[5];
        const uint64_t need = 32 + (uint64_t)l_name + 4 * (uint64_t)n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
        ok = block_size >= 32 && need <= block_size && at + 4 + block_size <= lim;
        if (ok) {
          h->x_ref[k] = (int32_t)w[1];
          h->x_pos[k] = (int32_t)w[2];
          h->x_lname[k] = (uint8_t)l_name;
          h->x_mapq[k] = (uint8_t)((w[3] >> 8) & 0xffu);
          h->x_ncig[k] = n_cigar;
          h->x_flags[k] = (uint16_t)(w[4] >> 16);
          h->x_lseq[k] = l_seq;
          h->x_mref[k] = (int32_t)w[6];
          h->x_mpos[k] = (int32_t)w[7];
          h->x_bytes[k] = 4 + block_size;
          // the CIGAR lies right behind the name: its span now, while the record's lines are here
          bool simp = false, has_bad = false;
          h->x_span[k] = cigar_span(bam + at + 36 + l_name, n_cigar, &simp, &has_bad);
          h->x_cfacts[k] = (uint8_t)((simp ? 1 : 0) | (has_bad ? 2 : 0));
        }
      }
      if (!ok) {
        int64_t cur = bad.load();
        while ((cur < 0 || (int64_t)k < cur) && !bad.compare_exchange_weak(cur, (int64_t)k)) {}
        return;
      }
    }
  });
  h->pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  h->prof[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (bad.load() >= 0)
    return fail(h, BP_EINVAL, "bp_submit_bam: record %lld (offset %u) does not fit its block or the bytes given",
                (long long)bad.load(), rec_off[bad.load()]);
  Src S;
  S.n = n;
  S.ref_id = h->x_ref.data();
  S.pos = h->x_pos.data();
  S.mate_ref_id = h->x_mref.data();
  S.mate_pos = h->x_mpos.data();
  S.flags = h->x_flags.data();
  S.mapq = h->x_mapq.data();
  S.n_cigar = h->x_ncig.data();
  S.l_seq = h->x_lseq.data();
  S.raw = bam;
  S.off = rec_off;
  S.l_name = h->x_lname.data();
  S.rec_bytes = h->x_bytes.data();
  S.span = h->x_span.data();
  S.cig_facts = h->x_cfacts.data();
  return guarded_submit(h, S, "bp_submit_bam");
  } catch (const std::bad_alloc&) {
    return fail(h, BP_ENOMEM, "bp_submit_bam: out of host memory");
  } catch (const std::exception& ex) {
    return fail(h, BP_ENOMEM, "bp_submit_bam: %s", ex.what());
  }
}

int bp_submit_bam_device(bp_handle* h, const uint8_t* stream, size_t n_bytes, const bp_bam_columns* c,
                         int64_t first, int64_t n) {
  if (!h || !c || n < 0 || first < 0 || (n > 0 && !stream)) return fail(h, BP_EINVAL, "bp_submit_bam_device: bad argument");
  if (n_bytes > 0xffffffffull) return fail(h, BP_EINVAL, "bp_submit_bam_device: more than 4 GiB in one call");
  if (h->o.stitch && !(c->names && c->name_off))
    return fail(h, BP_EINVAL, "bp_submit_bam_device: stitch mode needs the read names on the host (bp_bam_scan with "
                              "cols->names, or bp_submit_bam)");
  try {
    // what bp_submit_bam checks while it reads the fields: every record inside the stream, its
    // fields inside its block (the scan says so), CIGARs of ops the walk knows
    std::atomic<int64_t> bad(-1);
    const size_t N = (size_t)n, chunk = 16384;
    parallel_for((N + chunk - 1) / chunk, pack_threads(), [&](size_t ci) {
      for (size_t k = ci * chunk; k < std::min(N, (ci + 1) * chunk); k++) {
        const size_t i = (size_t)first + k;
        if ((c->facts[i] & 4) || (uint64_t)c->off[i] + c->bytes[i] > n_bytes) {
          int64_t cur = bad.load();
          while ((cur < 0 || (int64_t)k < cur) && !bad.compare_exchange_weak(cur, (int64_t)k)) {}
          return;
        }
      }
    });
    if (bad.load() >= 0)
      return fail(h, BP_EINVAL, "bp_submit_bam_device: record %lld (offset %u) does not fit its block or the bytes given",
                  (long long)bad.load(), c->off[first + bad.load()]);
    Src S;
    S.n = n;
    S.ref_id = c->ref_id + first;
    S.pos = c->pos + first;
    S.mate_ref_id = c->mate_ref_id + first;
    S.mate_pos = c->mate_pos + first;
    S.flags = c->flags + first;
    S.mapq = c->mapq + first;
    S.n_cigar = c->n_cigar + first;
    S.l_seq = c->l_seq + first;
    S.raw_dev = stream;
    S.off = c->off + first;
    S.l_name = c->l_name + first;
    S.rec_bytes = c->bytes + first;
    S.span = c->span + first;
    S.cig_facts = c->facts + first;
    if (c->names && c->name_off) {
      S.names = c->names;
      S.name_off = c->name_off + first;
    }
    return guarded_submit(h, S, "bp_submit_bam_device");
  } catch (const std::bad_alloc&) {
    return fail(h, BP_ENOMEM, "bp_submit_bam_device: out of host memory");
  } catch (const std::exception& ex) {
    return fail(h, BP_ENOMEM, "bp_submit_bam_device: %s", ex.what());
  }
}

constexpr int32_t kSliceRecs = 2048;  // records per slice of a deep window

// Upper bound on the extra work items of deep windows: a record is counted in every window its
// span can reach.
static int64_t item_capacity(const bp_handle* h) {
  const int64_t per_rec = (int64_t)h->max_span / window_size() + 2;
  return per_rec * (2 * h->n_reads / kSliceRecs + 1) + 16;  // a sliced window has more than kSliceRecs records: <= 2n/kSliceRecs slices
}

// Kernel arguments for the handle's resident job.
static void fill_params(bp_handle* h, KernelParams& P) {
  P.cores = (const ReadCore*)h->d_cores.p;
  P.blob = (const uint8_t*)h->d_blob.p;
  P.mate = h->o.stitch ? (const MateEntry*)h->d_mate.p : nullptr;
  P.windows = (const Window*)h->d_windows.p;
  P.win_r0 = (const int32_t*)h->d_r0.p;
  P.win_r1 = (const int32_t*)h->d_r1.p;
  P.ref_first = (const int64_t*)h->d_ref_first.p;
  P.coarse_pos = (const int32_t*)h->d_coarse.p;
  P.iv_start = (const int32_t*)h->d_iv_start.p;
  P.iv_end = (const int32_t*)h->d_iv_end.p;
  P.iv_row = (const int64_t*)h->d_iv_row.p;
  P.rows = (uint32_t*)h->d_rows.p;
  P.err = (DeviceError*)h->d_err.p;
  P.qthr = (const uint8_t*)h->d_qthr.p;
  P.dbg = nullptr;
  P.slice_recs = kSliceRecs;
  P.extra_cap = (int32_t)std::min<int64_t>(item_capacity(h), 0x7fffffff);
  P.extra_items = (uint4*)((uint8_t*)h->d_items.p + 256);
  P.n_extra = (uint32_t*)h->d_items.p;
  P.n_reads = h->n_reads;
  P.n_windows = (int32_t)h->windows.size();
  P.max_span = h->max_span;
  P.clip = h->o.clip;
  P.flag_exclude = h->o.flag_exclude;
  P.mapq = h->o.mapq;
  P.min_base_qual = h->o.min_base_qual;
  P.stitch = h->o.stitch;
  P.qpass = h->qpass;
  P.ignore_strand = h->o.ignore_strand;
  P.strand_req = h->o.strand_req;
}

// Plan + pileup kernels over the resident job with the kernel arguments P (whose scratch
// pointers the caller chose), timed with the handle's events; the device error flags come back
// through `got` (host memory the copy may target: pinned for the allocation-free path).
static int run_kernels(bp_handle* h, const KernelParams& P, DeviceError* got) {
  HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
  launch_plan(P, (int32_t*)P.win_r0, (int32_t*)P.win_r1, h->stream);
  HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
  launch_pileup(P, h->stream);
  HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(got, P.err, sizeof(*got), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->timing = bp_timing{};
  (void)hipEventElapsedTime(&h->timing.plan_ms, h->ev[0], h->ev[1]);
  (void)hipEventElapsedTime(&h->timing.pileup_ms, h->ev[1], h->ev[2]);
  (void)hipEventElapsedTime(&h->timing.total_ms, h->ev[0], h->ev[2]);
  h->timing.n_windows = (int64_t)h->windows.size();
  h->timing.n_items = (int64_t)h->windows.size();
  h->rows_valid = false;
  if (got->kind == ERR_ITEM_OVERFLOW)
    return fail(h, BP_EDEVICE, "bp_compute: work-item list of deep windows overflowed");
  if (got->read != 0x7fffffff)
    return fail(h, BP_EDATA,
                "qualPassTable.lookup2: base quality >= 96 in a stitched overlap (record %d); the "
                "reference panics with index out of range (qual.go:26,79-81)", got->read);
  h->rows_valid = true;
  h->per_read_valid = false;
  return BP_OK;
}

int bp_compute(bp_handle* h) {
  if (!h) return fail(h, BP_EINVAL, "bp_compute: null handle");
  if (h->poison_code) return fail(h, h->poison_code, "%s", h->poison_msg.c_str());
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = freeze_bed(h)) return rc;
  bool need_resolve = false;
  if (h->o.stitch) {
    const size_t bytes = (size_t)h->n_reads * sizeof(MateEntry);
    if (int rc = ensure(h, h->d_mate, std::max<size_t>(bytes, 256), 0)) return rc;
    // (only after a submit: pairing can reach back to records of earlier submits, so the whole
    // table goes again then; a compute that follows a compute moves nothing)
    if (bytes && h->mate_dirty) {
      HIP_TRY(h, hipMemcpyAsync(h->d_mate.p, h->mate.data(), bytes, hipMemcpyHostToDevice, h->stream));
      need_resolve = true;
    }
    h->mate_dirty = false;
  }
  if (int rc = ensure(h, h->d_err, 256, 0)) return rc;
  if (int rc = ensure(h, h->d_cores, 256, 0)) return rc;
  if (int rc = ensure(h, h->d_blob, 256, 0)) return rc;
  {
    // close the per-reference record ranges: references after the last one seen are empty
    std::vector<int64_t>& rf = h->ref_first_closed;
    rf = h->ref_first;
    const size_t nref = h->ref_len.size();
    // rf[r] = first record with reference >= r, valid for r <= cur_ref (unmapped records
    // count as reference n_ref); references not reached yet are empty and start at the end
    for (size_t r = (size_t)h->cur_ref + 1; r <= nref; r++) rf[r] = h->n_reads;
    if (int rc = ensure(h, h->d_ref_first, (nref + 1) * 8 + 256, 0)) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_ref_first.p, rf.data(), (nref + 1) * 8, hipMemcpyHostToDevice, h->stream));
  }
  if (int rc = ensure(h, h->d_coarse, h->coarse_pos.size() * 4 + 256, h->coarse_uploaded * 4)) return rc;
  if (h->coarse_uploaded < h->coarse_pos.size()) {
    HIP_TRY(h, hipMemcpyAsync((int32_t*)h->d_coarse.p + h->coarse_uploaded, h->coarse_pos.data() + h->coarse_uploaded,
                              (h->coarse_pos.size() - h->coarse_uploaded) * 4, hipMemcpyHostToDevice, h->stream));
    h->coarse_uploaded = h->coarse_pos.size();
  }
  if (int rc = ensure(h, h->d_items, 256 + (size_t)item_capacity(h) * 16, 0)) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_items.p, 0, 256, h->stream));
  DeviceError init{0x7fffffff, ERR_NONE};
  HIP_TRY(h, hipMemcpyAsync(h->d_err.p, &init, sizeof(init), hipMemcpyHostToDevice, h->stream));
  KernelParams P{};
  fill_params(h, P);
#ifdef BP_STAMPS
  if (int rc = ensure(h, h->d_dbg, h->windows.size() * 8 * (8 * 8 + 8 * 4) + 256, 0)) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_dbg.p, 0, h->windows.size() * 8 * (8 * 8 + 8 * 4), h->stream));
  P.dbg = (unsigned long long*)h->d_dbg.p;
#endif
  if (need_resolve) {
    // the pair rule, once for the pairs the table has gained (each pair is worked once: the
    // kernel marks the reads it has done); the counting kernels then find its outcome in the payloads
    {
      // (an empty launch first: the first launch of a kernel has the runtime load its code, and that
      // host time would otherwise sit between the two events)
      KernelParams P0 = P;
      P0.n_reads = 0;
      launch_stitch_resolve(P0, h->stream);
    }
    HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    launch_stitch_resolve(P, h->stream);
    HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventSynchronize(h->ev[1]));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
    h->resolve_ms += ms;
  }
  DeviceError got{};
  if (int rc = run_kernels(h, P, &got)) return rc;
  h->resident = h->finished;
  return BP_OK;
}

// bp_compute for callers that cannot allocate: the same kernels over the same resident records
// and BED, writing the same rows, with their scratch in this GPU's copy of the file-scope
// `scratch` array (through g_scratch_ptr, never the array's name) instead of buffers of the
// handle.  No hipMalloc / hipFree / allocating routine from here on: the tables the kernels read
// were uploaded by the bp_finish that must have run, the error flags come back into the pinned
// word bp_create allocated, and the counters that must start at zero are cleared with memsets on
// the job's stream.  The array keeps whatever the last call on this GPU left in it.
int bp_compute_static(bp_handle* h) {
  if (!h) return fail(h, BP_EINVAL, "bp_compute_static: null handle");
  if (h->poison_code) return fail(h, h->poison_code, "%s", h->poison_msg.c_str());
  if (!h->resident)
    return fail(h, BP_EINVAL, "bp_compute_static: call bp_finish first (it uploads what the kernels read; "
                              "this call allocates and uploads nothing)");
  uint8_t* sp = nullptr;
  if (h->device >= 0 && h->device < kMaxDevices) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    sp = g_scratch_ptr[h->device];
  }
  if (!sp)
    return fail(h, BP_EDEVICE, "bp_compute_static: no scratch address for device %d (hipGetSymbolAddress failed in bp_init, "
                               "or the device ordinal is beyond %d)", h->device, kMaxDevices - 1);
  const int64_t n_win = (int64_t)h->windows.size(), n_items = item_capacity(h);
  if (n_win > kStaticMaxWindows)
    return fail(h, BP_EINVAL, "bp_compute_static: %lld windows of %d positions; the static scratch holds %lld "
                              "(use bp_compute, or cut the job)", (long long)n_win, window_size(), (long long)kStaticMaxWindows);
  if (n_items > kStaticMaxItems)
    return fail(h, BP_EINVAL, "bp_compute_static: up to %lld work items of deep windows (%lld records); the static "
                              "scratch holds %lld (use bp_compute, or cut the job)", (long long)n_items,
                (long long)h->n_reads, (long long)kStaticMaxItems);
  HIP_TRY(h, hipSetDevice(h->device));
  KernelParams P{};
  fill_params(h, P);
  P.win_r0 = (const int32_t*)(sp + kStaticR0Off);
  P.win_r1 = (const int32_t*)(sp + kStaticR1Off);
  P.n_extra = (uint32_t*)(sp + kStaticCounterOff);
  P.extra_items = (uint4*)(sp + kStaticItemsOff);
  P.extra_cap = (int32_t)n_items;
  P.err = (DeviceError*)(sp + kStaticErrOff);
  // what must start at zero: the item counter; and the error flags at "none"
  HIP_TRY(h, hipMemsetAsync(sp + kStaticCounterOff, 0, 256, h->stream));
  HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)(sp + kStaticErrOff), 0x7fffffff, 1, h->stream));
  HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)(sp + kStaticErrOff + 4), (int)ERR_NONE, 1, h->stream));
  return run_kernels(h, P, h->host_err);
}

int bp_finish(bp_handle* h) {
  if (!h) return fail(h, BP_EINVAL, "bp_finish: null handle");
  h->finished = true;
  for (FirstReadTable& t : h->stored) t.clear();
  return bp_compute(h);
}

int64_t bp_num_rows(const bp_handle* h) {
  if (!h) return -1;
  if (h->frozen) return h->n_rows;
  int64_t n = 0;
  for (const auto& e : h->bed)
    for (size_t k = 0; k + 1 < e.size(); k += 2) n += (int64_t)e[k + 1] - e[k];
  return n;
}

int bp_rows(bp_handle* h, int64_t first, int64_t n, int32_t* ref_id, int32_t* pos, uint32_t* vals) {
  if (!h) return fail(h, BP_EINVAL, "bp_rows: null handle");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_rows: no computed rows (call bp_finish)");
  if (first < 0 || n < 0 || first + n > h->n_rows) return fail(h, BP_EINVAL, "bp_rows: range [%lld,%lld) of %lld", (long long)first, (long long)(first + n), (long long)h->n_rows);
  if (n == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  if (vals) {
    // the job's stream does not synchronise with the null stream: wait for what the _async calls left on it
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(vals, (const uint32_t*)h->d_rows.p + first * BP_ROW_U32, (size_t)n * 44, hipMemcpyDeviceToHost));
  }
  if (ref_id || pos) {
    size_t k = (size_t)(std::upper_bound(h->iv_row.begin(), h->iv_row.end(), first) - h->iv_row.begin()) - 1;
    int64_t done = 0;
    while (done < n) {
      int64_t r = first + done;
      while (h->iv_row[k + 1] <= r) k++;
      int64_t off = r - h->iv_row[k];
      int64_t cnt = std::min<int64_t>(n - done, (int64_t)(h->iv_end[k] - h->iv_start[k]) - off);
      for (int64_t t = 0; t < cnt; t++) {
        if (ref_id) ref_id[done + t] = h->iv_ref[k];
        if (pos) pos[done + t] = (int32_t)(h->iv_start[k] + off + t);
      }
      done += cnt;
    }
  }
  return BP_OK;
}

int bp_rows_device(bp_handle* h, const uint32_t** dptr) {
  if (!h || !dptr) return fail(h, BP_EINVAL, "bp_rows_device: null argument");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_rows_device: no computed rows");
  *dptr = (const uint32_t*)h->d_rows.p;
  return BP_OK;
}

int bp_stream(bp_handle* h, void** stream) {
  if (!h || !stream) return fail(h, BP_EINVAL, "bp_stream: null argument");
  *stream = (void*)h->stream;
  return BP_OK;
}

int bp_rows_export_async(bp_handle* h, int64_t first, int64_t n, uint32_t* dst_device) {
  if (!h || (n > 0 && !dst_device)) return fail(h, BP_EINVAL, "bp_rows_export: null argument");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_rows_export: no computed rows");
  if (first < 0 || n < 0 || first + n > h->n_rows)
    return fail(h, BP_EINVAL, "bp_rows_export: rows [%lld, %lld) out of range (%lld rows)", (long long)first,
                (long long)(first + n), (long long)h->n_rows);
  if (n == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(dst_device, (const uint32_t*)h->d_rows.p + first * BP_ROW_U32,
                            (size_t)n * BP_ROW_U32 * 4, hipMemcpyDeviceToDevice, h->stream));
  return BP_OK;
}

int bp_rows_export(bp_handle* h, int64_t first, int64_t n, uint32_t* dst_device) {
  if (int rc = bp_rows_export_async(h, first, n, dst_device)) return rc;
  if (n > 0) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return BP_OK;
}

int bp_rows_accumulate_async(bp_handle* h, int64_t first, int64_t n, const uint32_t* src_device) {
  if (!h || (n > 0 && !src_device)) return fail(h, BP_EINVAL, "bp_rows_accumulate: null argument");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_rows_accumulate: no computed rows");
  if (first < 0 || n < 0 || first + n > h->n_rows)
    return fail(h, BP_EINVAL, "bp_rows_accumulate: rows [%lld, %lld) out of range (%lld rows)", (long long)first,
                (long long)(first + n), (long long)h->n_rows);
  if (n == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  launch_rows_add((uint32_t*)h->d_rows.p + first * BP_ROW_U32, src_device, n * BP_ROW_U32, h->stream);
  HIP_TRY(h, hipGetLastError());
  h->per_read_valid = false;
  return BP_OK;
}

int bp_rows_accumulate(bp_handle* h, int64_t first, int64_t n, const uint32_t* src_device) {
  if (int rc = bp_rows_accumulate_async(h, first, n, src_device)) return rc;
  if (n > 0) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return BP_OK;
}

int bp_column_sums(bp_handle* h, uint64_t sums[BP_ROW_U32]) {
  if (!h || !sums) return fail(h, BP_EINVAL, "bp_column_sums: null argument");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_column_sums: no computed rows");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = ensure(h, h->d_sums, 256, 0)) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_sums.p, 0, 11 * 8, h->stream));
  launch_column_sums((const uint32_t*)h->d_rows.p, h->n_rows, (unsigned long long*)h->d_sums.p, h->stream);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(sums, h->d_sums.p, 11 * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return BP_OK;
}

int bp_per_read_build(bp_handle* h) {
  if (!h) return fail(h, BP_EINVAL, "bp_per_read_build: null handle");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_per_read_build: no computed rows (call bp_finish)");
  if (h->o.stitch)  // the reference panics at the first merged pair (pileup.go:422-443)
    return fail(h, BP_EINVAL, "stitched per-read features not yet supported");
  if (h->per_read_valid) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  const int64_t n = 4 * h->n_rows;
  const int64_t n_tiles = (n + 1023) / 1024;
  if (int rc = ensure(h, h->d_pr_off, (size_t)n * 4 + 256, 0)) return rc;
  if (int rc = ensure(h, h->d_pr_tiles, (size_t)(n_tiles + 1) * 8 + 256, 0)) return rc;
  if (int rc = ensure(h, h->d_err, 256, 0)) return rc;
  uint32_t* off32 = (uint32_t*)h->d_pr_off.p;
  HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
  HIP_TRY(h, hipMemsetAsync(h->d_err.p, 0, 4, h->stream));
  // lengths -> in-tile offsets and tile totals -> tile bases, all on the device; only the total
  // (it sizes the lists) and the overflow flag come back
  launch_tile_scan((const uint32_t*)h->d_rows.p, off32, n, (unsigned long long*)h->d_pr_tiles.p, (int32_t*)h->d_err.p,
                   h->stream);
  HIP_TRY(h, hipGetLastError());
  unsigned long long run = 0;
  int32_t overflow = 0;
  HIP_TRY(h, hipMemcpyAsync(&run, (const unsigned long long*)h->d_pr_tiles.p + n_tiles, 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(&overflow, h->d_err.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (overflow) return fail(h, BP_EINVAL, "bp_per_read_build: more than 2^32 list entries on 256 positions");
  h->n_feat = run;
  h->pr_tile_base.clear();  // fetched when bp_per_read_offsets first needs them
  // + 128 entries of slack behind the last list (see the cursor check in per_read_kernel)
  if (int rc = ensure(h, h->d_pr_feat, ((size_t)run + 128) * sizeof(PerRead) + 256, 0)) return rc;
  const size_t n_iv = h->iv_ref.size();
  if (int rc = ensure(h, h->d_iv_ref, n_iv * 4 + 256, 0)) return rc;
  if (n_iv) HIP_TRY(h, hipMemcpyAsync(h->d_iv_ref.p, h->iv_ref.data(), n_iv * 4, hipMemcpyHostToDevice, h->stream));
  if (int rc = ensure(h, h->d_err, 256, 0)) return rc;
  const int32_t none = 0x7fffffff;
  HIP_TRY(h, hipMemcpyAsync(h->d_err.p, &none, 4, hipMemcpyHostToDevice, h->stream));
  // tiles of kPrT positions, cut per BED interval
  std::vector<int64_t> tile0(n_iv + 1, 0);
  for (size_t k = 0; k < n_iv; k++)
    tile0[k + 1] = tile0[k] + ((int64_t)h->iv_end[k] - h->iv_start[k] + kPrT - 1) / kPrT;
  if (int rc = ensure(h, h->d_pr_tile0, (n_iv + 1) * 8 + 256, 0)) return rc;
  HIP_TRY(h, hipMemcpyAsync(h->d_pr_tile0.p, tile0.data(), (n_iv + 1) * 8, hipMemcpyHostToDevice, h->stream));
  KernelParams P{};
  fill_params(h, P);
  PerReadParams Q{};
  Q.iv_ref = (const int32_t*)h->d_iv_ref.p;
  Q.iv_tile0 = (const int64_t*)h->d_pr_tile0.p;
  Q.off32 = off32;
  Q.tile_base = (const unsigned long long*)h->d_pr_tiles.p;
  Q.total = run;
  Q.n_lists = n;
  Q.out = (PerRead*)h->d_pr_feat.p;
  Q.bad_row = (int32_t*)h->d_err.p;
  Q.n_tiles = tile0[n_iv];
  if (int rc = ensure(h, h->d_pr_tile_lo, (size_t)Q.n_tiles * 4 + 256, 0)) return rc;
  Q.tile_lo = (const int32_t*)h->d_pr_tile_lo.p;
  Q.n_iv = (int32_t)n_iv;
  HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
  launch_per_read(P, Q, h->stream);
  HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
  // (tile0 outlives the copy: the stream is synchronised below)
  HIP_TRY(h, hipGetLastError());
  int32_t bad = 0;
  HIP_TRY(h, hipMemcpyAsync(&bad, h->d_err.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  (void)hipEventElapsedTime(&h->timing.per_read_scan_ms, h->ev[0], h->ev[1]);
  (void)hipEventElapsedTime(&h->timing.per_read_ms, h->ev[1], h->ev[2]);
  if (bad != none)
    return fail(h, BP_EDEVICE, "bp_per_read_build: list lengths of row %d disagree with its counts", bad);
  h->per_read_valid = true;
  return BP_OK;
}

int64_t bp_per_read_total(const bp_handle* h) { return h && h->per_read_valid ? (int64_t)h->n_feat : -1; }

int bp_per_read_offsets(bp_handle* h, int64_t first_row, int64_t n_rows, uint64_t* off) {
  if (!h || !off) return fail(h, BP_EINVAL, "bp_per_read_offsets: null argument");
  if (!h->per_read_valid) return fail(h, BP_EINVAL, "bp_per_read_offsets: call bp_per_read_build first");
  if (first_row < 0 || n_rows < 0 || first_row + n_rows > h->n_rows)
    return fail(h, BP_EINVAL, "bp_per_read_offsets: rows [%lld,%lld) of %lld", (long long)first_row,
                (long long)(first_row + n_rows), (long long)h->n_rows);
  HIP_TRY(h, hipSetDevice(h->device));
  const int64_t i0 = 4 * first_row, m = 4 * n_rows;
  if (h->pr_tile_base.empty()) {
    const size_t n_tiles = (size_t)((4 * h->n_rows + 1023) / 1024);
    h->pr_tile_base.assign(n_tiles + 1, 0);
    HIP_TRY(h, hipMemcpy(h->pr_tile_base.data(), h->d_pr_tiles.p, (n_tiles + 1) * 8, hipMemcpyDeviceToHost));
  }
  std::vector<uint32_t> rel((size_t)m);
  if (m) HIP_TRY(h, hipMemcpy(rel.data(), (const uint32_t*)h->d_pr_off.p + i0, (size_t)m * 4, hipMemcpyDeviceToHost));
  for (int64_t k = 0; k < m; k++) off[k] = h->pr_tile_base[(size_t)((i0 + k) >> 10)] + rel[(size_t)k];
  off[m] = i0 + m >= 4 * h->n_rows ? h->n_feat : h->pr_tile_base[(size_t)((i0 + m) >> 10)] + 0;
  if (i0 + m < 4 * h->n_rows) {  // the first list behind the range
    uint32_t nxt = 0;
    HIP_TRY(h, hipMemcpy(&nxt, (const uint32_t*)h->d_pr_off.p + i0 + m, 4, hipMemcpyDeviceToHost));
    off[m] += nxt;
  }
  return BP_OK;
}

int bp_per_read_features(bp_handle* h, uint64_t first, uint64_t n, bp_per_read* out) {
  if (!h || (!out && n)) return fail(h, BP_EINVAL, "bp_per_read_features: null argument");
  if (!h->per_read_valid) return fail(h, BP_EINVAL, "bp_per_read_features: call bp_per_read_build first");
  if (first + n > h->n_feat)
    return fail(h, BP_EINVAL, "bp_per_read_features: entries [%llu,%llu) of %llu", (unsigned long long)first,
                (unsigned long long)(first + n), (unsigned long long)h->n_feat);
  if (n == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpy(out, (const PerRead*)h->d_pr_feat.p + first, (size_t)n * sizeof(PerRead), hipMemcpyDeviceToHost));
  return BP_OK;
}

int bp_set_ref_seq8(bp_handle* h, int32_t ref_id, int32_t start, const uint8_t* seq8, int64_t n) {
  if (!h || (!seq8 && n)) return fail(h, BP_EINVAL, "bp_set_ref_seq8: null argument");
  if (ref_id < 0 || (size_t)ref_id >= h->ref_len.size() || start < 0 || n < 0)
    return fail(h, BP_EINVAL, "bp_set_ref_seq8: reference %d, start %d, %lld bases", ref_id, start, (long long)n);
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = ensure(h, h->d_seq8, (size_t)n + 256, 0)) return rc;
  if (n) HIP_TRY(h, hipMemcpyAsync(h->d_seq8.p, seq8, (size_t)n, hipMemcpyHostToDevice, h->stream));
  const std::string& nm = (size_t)ref_id < h->ref_name.size() ? h->ref_name[(size_t)ref_id] : std::string();
  if (int rc = ensure(h, h->d_refname, nm.size() + 256, 0)) return rc;
  if (!nm.empty()) HIP_TRY(h, hipMemcpyAsync(h->d_refname.p, nm.data(), nm.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // the caller's buffers are free again
  h->seq_ref = ref_id;
  h->seq_start = start;
  h->seq_len = n;
  return BP_OK;
}

int64_t bp_render_text_bound(const bp_handle* h, int32_t kind, int64_t n_rows) {
  kind &= 0xff;  // (the strand flags do not change the bound)
  if (!h || h->seq_ref < 0 || n_rows < 0 || kind < 0 || kind > 2) return -1;
  const size_t nm = (size_t)h->seq_ref < h->ref_name.size() ? h->ref_name[(size_t)h->seq_ref].size() : 0;
  const size_t head = nm + 1 + 11 + 2;
  if (kind == BP_TEXT_BASESTRAND) return n_rows * (int64_t)(head + 8 * 11);
  if (kind == BP_TEXT_TSV_REF) return n_rows * (int64_t)(head + 11 + 11 + 2);
  return n_rows * (int64_t)(4 * (head + 2 + 11 + 11 + 2));
}

int bp_render_text(bp_handle* h, int32_t kind, int32_t cols, int64_t first, int64_t n, char* out, size_t cap,
                   size_t* len) {
  if (!h || !len) return fail(h, BP_EINVAL, "bp_render_text: null argument");
  const int32_t strand = (kind >> 8) & 3;
  kind &= 0xff;
  if (kind < 0 || kind > 2 || strand == 3) return fail(h, BP_EINVAL, "bp_render_text: kind %d", kind);
  if (strand && (h->o.ignore_strand || h->o.strand_req || h->o.min_base_qual > 0))
    return fail(h, BP_EINVAL, "bp_render_text: one strand's lines come from split-strand rows of a job without a quality rule");
  if (cols & ~(BP_COL_DPREF | BP_COL_DPALT | BP_COL_HIGHQ | BP_COL_LOWQ))
    return fail(h, BP_EINVAL, "bp_render_text: per-read columns are not rendered on the device");
  if (!h->rows_valid) return fail(h, BP_EINVAL, "bp_render_text: no computed rows (call bp_finish)");
  if (h->seq_ref < 0) return fail(h, BP_EINVAL, "bp_render_text: call bp_set_ref_seq8 first");
  if (first < 0 || n < 0 || first + n > h->n_rows)
    return fail(h, BP_EINVAL, "bp_render_text: rows [%lld,%lld) of %lld", (long long)first,
                (long long)(first + n), (long long)h->n_rows);
  *len = 0;
  h->txt_len = 0;
  h->dfl_out = nullptr;
  if (n == 0) return BP_OK;
  if (out && (int64_t)cap < bp_render_text_bound(h, kind, n))
    return fail(h, BP_EINVAL, "bp_render_text: buffer of %llu bytes for %lld rows (see bp_render_text_bound)",
                (unsigned long long)cap, (long long)n);
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t n_iv = h->iv_ref.size();
  if (!h->iv_ref_uploaded) {
    if (int rc = ensure(h, h->d_iv_ref, n_iv * 4 + 256, 0)) return rc;
    if (n_iv) HIP_TRY(h, hipMemcpyAsync(h->d_iv_ref.p, h->iv_ref.data(), n_iv * 4, hipMemcpyHostToDevice, h->stream));
    h->iv_ref_uploaded = true;
  }
  const int64_t n_tiles = (n + 1023) / 1024;
  if (int rc = ensure(h, h->d_txt_len, (size_t)n * 4 + 256, 0)) return rc;
  if (int rc = ensure(h, h->d_txt_tile, (size_t)(n_tiles + 1) * 8 + 256, 0)) return rc;
  if (int rc = ensure(h, h->d_txt_out, (size_t)bp_render_text_bound(h, kind, n) + 256, 0)) return rc;
  if (int rc = ensure(h, h->d_err, 256, 0)) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_err.p, 0, 4, h->stream));
  TextParams T{};
  T.rows = (const uint32_t*)h->d_rows.p;
  T.iv_row = (const int64_t*)h->d_iv_row.p;
  T.iv_start = (const int32_t*)h->d_iv_start.p;
  T.iv_ref = (const int32_t*)h->d_iv_ref.p;
  T.seq8 = (const uint8_t*)h->d_seq8.p;
  T.name = (const char*)h->d_refname.p;
  T.len = (uint32_t*)h->d_txt_len.p;
  T.tile = (unsigned long long*)h->d_txt_tile.p;
  T.out = (char*)h->d_txt_out.p;
  T.bad = (int32_t*)h->d_err.p;
  T.first = first;
  T.n = n;
  T.seq_len = h->seq_len;
  T.n_iv = (int32_t)n_iv;
  T.seq_ref = h->seq_ref;
  T.seq_start = h->seq_start;
  T.name_len = (int32_t)((size_t)h->seq_ref < h->ref_name.size() ? h->ref_name[(size_t)h->seq_ref].size() : 0);
  T.kind = kind;
  T.cols = cols;
  T.strand = strand;
  launch_text(T, h->stream);
  launch_text_write(T, h->stream);
  HIP_TRY(h, hipGetLastError());
  unsigned long long total = 0;
  int32_t bad = 0;
  HIP_TRY(h, hipMemcpyAsync(&total, T.tile + n_tiles, 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(&bad, h->d_err.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (bad) return fail(h, BP_EINVAL, "bp_render_text: a row of [%lld,%lld) has no reference base (bp_set_ref_seq8 covers reference %d, positions [%d,%lld))",
                       (long long)first, (long long)(first + n), h->seq_ref, h->seq_start, (long long)h->seq_start + (long long)h->seq_len);
  if ((int64_t)total > bp_render_text_bound(h, kind, n)) return fail(h, BP_EDEVICE, "bp_render_text: text of %llu bytes", total);
  h->txt_len = (size_t)total;
  *len = (size_t)total;
  if (out) HIP_TRY(h, hipMemcpy(out, h->d_txt_out.p, (size_t)total, hipMemcpyDeviceToHost));
  return BP_OK;
}

int bp_render_fetch(bp_handle* h, char* out, size_t cap) {
  if (!h || (!out && h->txt_len)) return fail(h, BP_EINVAL, "bp_render_fetch: null argument");
  if (cap < h->txt_len) return fail(h, BP_EINVAL, "bp_render_fetch: buffer of %llu bytes for %llu of text",
                                    (unsigned long long)cap, (unsigned long long)h->txt_len);
  if (h->txt_len == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpy(out, h->d_txt_out.p, h->txt_len, hipMemcpyDeviceToHost));
  return BP_OK;
}

int64_t bp_render_bgzf_bound(const bp_handle* h) { return h ? (int64_t)deflate_bgzf_bound(h->txt_len) : -1; }

int bp_render_fetch_bgzf(bp_handle* h, char* out, size_t cap, size_t* len) {
  if (!h || !len) return fail(h, BP_EINVAL, "bp_render_fetch_bgzf: null argument");
  *len = 0;
  if (h->txt_len == 0) return BP_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  if (!h->dfl_out) {  // (a call with out = NULL has left the members of this text on the device otherwise)
    if (out && cap < deflate_bgzf_bound(h->txt_len))
      return fail(h, BP_EINVAL, "bp_render_fetch_bgzf: buffer of %llu bytes for %llu of text (see bp_render_bgzf_bound)",
                  (unsigned long long)cap, (unsigned long long)h->txt_len);
    if (!h->dfl) h->dfl = deflate_scratch_acquire(h->device);
    std::string err;
    // (d_txt_out ends in 256 bytes of slack: the tokeniser's dword loads may look past the text)
    if (int rc = deflate_bgzf_device(h->dfl, (const uint8_t*)h->d_txt_out.p, h->txt_len, h->stream, &h->dfl_out, &h->dfl_len, &err)) {
      h->dfl_out = nullptr;
      return fail(h, rc, "bp_render_fetch_bgzf: %s", err.c_str());
    }
  }
  *len = h->dfl_len;
  if (!out) return BP_OK;
  if (cap < h->dfl_len)
    return fail(h, BP_EINVAL, "bp_render_fetch_bgzf: buffer of %llu bytes for %llu of members", (unsigned long long)cap,
                (unsigned long long)h->dfl_len);
  HIP_TRY(h, hipMemcpy(out, h->dfl_out, h->dfl_len, hipMemcpyDeviceToHost));
  return BP_OK;
}

int bp_get_timing(const bp_handle* h, bp_timing* out) {
  if (!h || !out) return BP_EINVAL;
  *out = h->timing;
  out->submit_scan_ms = (float)(h->scan_s * 1e3);
  out->submit_pack_ms = (float)(h->pack_s * 1e3);
  out->stitch_resolve_ms = (float)h->resolve_ms;
  return BP_OK;
}

int bp_job_size(const bp_handle* h, int64_t* n_records, int64_t* payload_bytes) {
  if (!h || !n_records || !payload_bytes) return BP_EINVAL;
  *n_records = h->n_reads;
  *payload_bytes = (int64_t)h->blob_used * 8;
  return BP_OK;
}

int64_t bp_algorithmic_bytes(const bp_handle* h) {
  if (!h) return -1;
  return h->algo_in_bytes + 44 * bp_num_rows(h);
}

#ifdef BP_STAMPS
// diagnostic build only: copy the phase stamps of the last bp_compute (n_windows*8*8 u64)
int bp_debug_stamps(bp_handle* h, unsigned long long* out, int64_t n) {
  return hipMemcpy(out, h->d_dbg.p, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

const char* bp_last_error(const bp_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

void bp_destroy(bp_handle* h) {
  if (!h) return;
  if (const char* e = getenv("BP_PROFILE"); e && *e && *e != '0')
    fprintf(stderr,
            "bp profile (submit, s): extract %.3f layout %.3f scan-parallel %.3f scan-serial %.3f items %.3f "
            "ensure %.3f copy-wait %.3f total %.3f; %lld records\n",
            h->prof[0], h->prof[1], h->prof[2], h->prof[3], h->prof[4], h->prof[5], h->prof[6], h->prof[7],
            (long long)h->n_reads);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (DevBuf* b : {&h->d_raw, &h->d_pack, &h->d_coarse, &h->d_items, &h->d_iv_ref, &h->d_pr_off, &h->d_pr_tiles, &h->d_pr_tile0, &h->d_pr_tile_lo, &h->d_pr_feat, &h->d_seq8, &h->d_refname, &h->d_txt_len, &h->d_txt_tile, &h->d_txt_out, &h->d_cores, &h->d_blob, &h->d_mate, &h->d_windows, &h->d_r0, &h->d_r1, &h->d_iv_start,
                    &h->d_iv_end, &h->d_iv_row, &h->d_rows, &h->d_err, &h->d_sums, &h->d_qthr, &h->d_dbg, &h->d_ref_first})
    b->release(h->device);
  for (Staging& s : h->stage) {
    if (!s.done) continue;
    s.busy = false;  // (the stream has been synchronised)
    StagePool& sp = stage_pool();
    std::unique_lock<std::mutex> lk(sp.mu);
    if (!dev_cache().off && sp.v.size() < 6) {
      sp.v.push_back(s);
      s = Staging();
    } else {
      lk.unlock();
      free_stage(s);
    }
  }
  deflate_scratch_release(h->device, h->dfl);
  for (hipEvent_t e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->host_err) (void)hipHostFree(h->host_err);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

}  // extern "C"
