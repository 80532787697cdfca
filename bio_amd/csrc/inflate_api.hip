// inflate_api.hip -- host side of the device BGZF inflate (include/bio_pileup.h, bp_inflate_*).
//
// One bp_inflater = one stream with its buffers: the compressed bytes of a batch of BGZF blocks
// go to the device as they lie in the file, the host says where each block's deflate data
// starts and ends and where its ISIZE bytes belong in the output (it has parsed the 18-byte block
// headers and the trailers, bam.cpp), the kernels of inflate_kernels.hip inflate every block and
// check its CRC-32, and the output comes back to the host and/or stays on the device for
// bp_submit_bam's record packing.  What bgzf.Reader does under bam.Reader in the reference
// (encoding/bamprovider/bamprovider.go:417-435).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bio_pileup.h"
#include "bam_device.h"
#include "deflate_api.h"
#include "inflate_device.h"

namespace bp {
void launch_inflate(const uint8_t* comp, const InflateBlock* blks, int n_blocks, uint8_t* out, uint32_t* status,
                    hipStream_t s);
void launch_bgzf_crc(const uint8_t* out, const InflateBlock* blks, int n_blocks, const uint32_t* crc_tables,
                     uint32_t* status, hipStream_t s);
void launch_bam_scan(const uint8_t* buf, uint64_t lo, uint64_t end, int32_t n_ref, int32_t n_stretch, WalkStretch* ws,
                     uint32_t* offs, uint32_t* out4, hipStream_t s);
void launch_bam_names(const uint8_t* buf, int64_t n, const uint32_t* off, const uint32_t* name_off, uint8_t* out,
                      hipStream_t s);
void launch_bam_fields(const uint8_t* buf, int32_t n_stretch, const WalkStretch* ws, const uint32_t* offs,
                       const BamColumns& C, hipStream_t s);
}
using namespace bp;

static_assert(sizeof(bp_bgzf_block) == sizeof(InflateBlock), "bp_bgzf_block is shipped as InflateBlock");

namespace {
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};
std::string g_inflate_error;
}  // namespace

struct bp_inflater {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  Buf comp, blks, out, status;  // device
  Buf walk, offs, cols;         // device: the record scan's stretches, offsets and columns
  Buf names;                    // device: name offsets (n + 1 dwords), then the gathered names
  Buf dfl_in;                   // device: bp_deflate_bgzf's input
  DeflateScratch* dfl = nullptr;
  uint32_t* h_scan = nullptr;   // pinned: the verifier's four words
  uint32_t* tables = nullptr;   // device: CRC tables
  uint32_t* h_status = nullptr; // pinned
  size_t h_status_cap = 0;
  float inflate_ms = 0, crc_ms = 0;
  int64_t rewalked = 0;  // stretches of all scans whose guess was wrong
  std::string err;
};

namespace {

int ifail(bp_inflater* f, int code, const char* fmt, ...) {
  char tmp[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  (f ? f->err : g_inflate_error) = tmp;
  return code;
}

#define INF_TRY(f, call)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return ifail(f, BP_EDEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int grow(bp_inflater* f, Buf& b, size_t need) {
  if (need <= b.cap) return BP_OK;
  if (b.p) INF_TRY(f, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  const size_t cap = need + need / 4 + 4096;
  hipError_t e = hipMalloc(&b.p, cap);
  if (e != hipSuccess) return ifail(f, BP_ENOMEM, "hipMalloc(%zu): %s", cap, hipGetErrorString(e));
  b.cap = cap;
  return BP_OK;
}

const char* status_text(uint32_t s) {
  switch (s) {
    case INF_BAD_BLOCK_TYPE: return "invalid deflate block type";
    case INF_BAD_STORED_LEN: return "stored block length check failed";
    case INF_BAD_CODE_LENGTHS: return "invalid code lengths";
    case INF_BAD_CODE: return "invalid literal/length or distance code";
    case INF_DISTANCE_TOO_FAR: return "match distance reaches in front of the block";
    case INF_OUTPUT_OVERFLOW: return "more data than ISIZE";
    case INF_INPUT_OVERRUN: return "deflate stream runs past the block";
    case INF_OUTPUT_SHORT: return "less data than ISIZE";
    case INF_BAD_CRC: return "CRC-32 mismatch";
    default: return "unknown";
  }
}

}  // namespace

extern "C" {

int bp_inflate_create(int32_t device, bp_inflater** out) {
  if (!out) return ifail(nullptr, BP_EINVAL, "bp_inflate_create: null argument");
  *out = nullptr;
  if (int rc = bp_init(device)) {
    g_inflate_error = bp_last_error(nullptr);
    return rc;
  }
  bp_inflater* f = new bp_inflater();
  f->device = device;
  auto bail = [&](const char* what, hipError_t e) {
    int rc = ifail(nullptr, BP_EDEVICE, "%s: %s", what, hipGetErrorString(e));
    bp_inflate_destroy(f);
    return rc;
  };
  hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return bail("hipStreamCreate", e);
  for (int i = 0; i < 3; i++) {
    e = hipEventCreate(&f->ev[i]);
    if (e != hipSuccess) return bail("hipEventCreate", e);
  }
  std::vector<uint32_t> t(1280);
  crc_tables(t.data());
  e = hipMalloc((void**)&f->tables, 1280 * 4);
  if (e != hipSuccess) return bail("hipMalloc", e);
  e = hipMemcpy(f->tables, t.data(), 1280 * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) return bail("hipMemcpy", e);
  *out = f;
  return BP_OK;
}

void bp_inflate_destroy(bp_inflater* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  if (f->h_scan) (void)hipHostFree(f->h_scan);
  for (Buf* b : {&f->comp, &f->blks, &f->out, &f->status, &f->walk, &f->offs, &f->cols, &f->names, &f->dfl_in})
    if (b->p) (void)hipFree(b->p);
  deflate_scratch_free(f->dfl);
  if (f->tables) (void)hipFree(f->tables);
  if (f->h_status) (void)hipHostFree(f->h_status);
  for (hipEvent_t ev : f->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
}

static int inflate_impl(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                        int64_t n_blocks, uint8_t* out_host, uint8_t* out_device, size_t out_bytes, int32_t check_crc,
                        int64_t* bad_block);

int bp_inflate_blocks(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                      int64_t n_blocks, uint8_t* out_host, size_t out_bytes, int32_t check_crc, int64_t* bad_block) {
  return inflate_impl(f, comp, comp_bytes, blocks, n_blocks, out_host, nullptr, out_bytes, check_crc, bad_block);
}

int bp_inflate_blocks_to(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                         int64_t n_blocks, uint8_t* out_device, size_t out_bytes, int32_t check_crc, int64_t* bad_block) {
  if (!out_device && out_bytes) return ifail(f, BP_EINVAL, "bp_inflate_blocks_to: null output");
  return inflate_impl(f, comp, comp_bytes, blocks, n_blocks, nullptr, out_device, out_bytes, check_crc, bad_block);
}

static int inflate_impl(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                        int64_t n_blocks, uint8_t* out_host, uint8_t* out_device, size_t out_bytes, int32_t check_crc,
                        int64_t* bad_block) {
  if (bad_block) *bad_block = -1;
  if (!f || n_blocks < 0 || (n_blocks > 0 && (!comp || !blocks)))
    return ifail(f, BP_EINVAL, "bp_inflate_blocks: bad argument");
  if (comp_bytes > 0xfffffff0ull || out_bytes > 0xfffffff0ull || n_blocks > 0x7fffffff)
    return ifail(f, BP_EINVAL, "bp_inflate_blocks: more than 4 GiB in one call");
  // every block inside the bytes given and inside the output, at most 64 KiB each (SAM spec 4.1)
  for (int64_t i = 0; i < n_blocks; i++) {
    const bp_bgzf_block& b = blocks[i];
    if ((uint64_t)b.in_off + b.in_len > comp_bytes || (uint64_t)b.out_off + b.out_len > out_bytes ||
        b.out_len > 65536 || b.in_len == 0) {
      if (bad_block) *bad_block = i;
      return ifail(f, BP_EINVAL, "bp_inflate_blocks: block %lld does not fit the bytes given (in %u+%u of %zu, out %u+%u of %zu)",
                   (long long)i, b.in_off, b.in_len, comp_bytes, b.out_off, b.out_len, out_bytes);
    }
  }
  INF_TRY(f, hipSetDevice(f->device));
  f->inflate_ms = f->crc_ms = 0;
  if (n_blocks == 0) return BP_OK;
  // (the bit reader loads whole 256-byte windows up to two ahead: 1 KiB of slack behind the data)
  if (int rc = grow(f, f->comp, comp_bytes + 1024)) return rc;
  if (int rc = grow(f, f->blks, (size_t)n_blocks * sizeof(InflateBlock))) return rc;
  if (!out_device)
    if (int rc = grow(f, f->out, out_bytes + 256)) return rc;
  uint8_t* const dout = out_device ? out_device : (uint8_t*)f->out.p;
  if (int rc = grow(f, f->status, (size_t)n_blocks * 4)) return rc;
  if (f->h_status_cap < (size_t)n_blocks) {
    if (f->h_status) INF_TRY(f, hipHostFree(f->h_status));
    f->h_status = nullptr;
    f->h_status_cap = 0;
    const size_t cap = (size_t)n_blocks + (size_t)n_blocks / 4 + 256;
    INF_TRY(f, hipHostMalloc((void**)&f->h_status, cap * 4, hipHostMallocDefault));
    f->h_status_cap = cap;
  }
  INF_TRY(f, hipMemcpyAsync(f->comp.p, comp, comp_bytes, hipMemcpyHostToDevice, f->stream));
  INF_TRY(f, hipMemcpyAsync(f->blks.p, blocks, (size_t)n_blocks * sizeof(InflateBlock), hipMemcpyHostToDevice, f->stream));
  INF_TRY(f, hipEventRecord(f->ev[0], f->stream));
  launch_inflate((const uint8_t*)f->comp.p, (const InflateBlock*)f->blks.p, (int)n_blocks, dout,
                 (uint32_t*)f->status.p, f->stream);
  INF_TRY(f, hipEventRecord(f->ev[1], f->stream));
  if (check_crc)
    launch_bgzf_crc(dout, (const InflateBlock*)f->blks.p, (int)n_blocks, f->tables,
                    (uint32_t*)f->status.p, f->stream);
  INF_TRY(f, hipEventRecord(f->ev[2], f->stream));
  INF_TRY(f, hipGetLastError());
  INF_TRY(f, hipMemcpyAsync(f->h_status, f->status.p, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, f->stream));
  if (out_host && out_bytes) INF_TRY(f, hipMemcpyAsync(out_host, dout, out_bytes, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipStreamSynchronize(f->stream));
  (void)hipEventElapsedTime(&f->inflate_ms, f->ev[0], f->ev[1]);
  (void)hipEventElapsedTime(&f->crc_ms, f->ev[1], f->ev[2]);
  for (int64_t i = 0; i < n_blocks; i++)
    if (f->h_status[i] != INF_OK) {
      if (bad_block) *bad_block = i;
      return ifail(f, BP_EDATA, "bgzf: corrupt block %lld (%s)", (long long)i, status_text(f->h_status[i]));
    }
  return BP_OK;
}

int bp_bam_scan(bp_inflater* f, const uint8_t* stream, uint64_t first, uint64_t end, int32_t n_ref,
                bp_bam_columns* cols, int64_t* n_out, uint64_t* exit_out, int32_t* stopped_out) {
  if (!f || !stream || !cols || !n_out || !exit_out || !stopped_out || first > end || end > 0xfffffff0ull)
    return ifail(f, BP_EINVAL, "bp_bam_scan: bad argument");
  *n_out = 0;
  *exit_out = first;
  *stopped_out = 0;
  if (first == end) return BP_OK;
  INF_TRY(f, hipSetDevice(f->device));
  const int32_t ns = (int32_t)((end - first + kStretch - 1) / kStretch);
  if (int rc = grow(f, f->walk, (size_t)ns * sizeof(WalkStretch))) return rc;
  if (int rc = grow(f, f->offs, (size_t)ns * kMaxPerStretch * 4)) return rc;
  if (!f->h_scan) INF_TRY(f, hipHostMalloc((void**)&f->h_scan, 64, hipHostMallocDefault));
  // 4 words of verdict live behind the stretches
  if (int rc = grow(f, f->status, 64)) return rc;
  uint32_t* d_out = (uint32_t*)f->status.p;
  launch_bam_scan(stream, first, end, n_ref, ns, (WalkStretch*)f->walk.p, (uint32_t*)f->offs.p, d_out, f->stream);
  INF_TRY(f, hipGetLastError());
  INF_TRY(f, hipMemcpyAsync(f->h_scan, d_out, 20, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipStreamSynchronize(f->stream));
  if (f->h_scan[0] != 0) {  // a guess failed: the host walks this stream
    const uint32_t j = f->h_scan[0] - 1;
    WalkStretch w[2] = {};
    (void)hipMemcpy(w, (const WalkStretch*)f->walk.p + (j ? j - 1 : 0), sizeof(WalkStretch) * (j ? 2 : 1), hipMemcpyDeviceToHost);
    const WalkStretch& cur = w[j ? 1 : 0];
    ifail(f, 1, "bp_bam_scan: stretch %u of %d started at %u (flags %u, %u records) where the chain stands at %u (the stretch before: %u -> %u, %u records, flags %u)",
          j, ns, cur.start, cur.flags, cur.n, f->h_scan[2], w[0].start, w[0].exit, w[0].n, w[0].flags);
    return 1;
  }
  const int64_t n = (int64_t)f->h_scan[1];
  if (n > cols->cap) return ifail(f, BP_ENOMEM, "bp_bam_scan: %lld records, room for %lld", (long long)n, (long long)cols->cap);
  *n_out = n;
  *exit_out = f->h_scan[2];
  *stopped_out = (int32_t)f->h_scan[3];
  f->rewalked += f->h_scan[4];
  if (n == 0) return BP_OK;
  // device columns: one allocation, 48 bytes a record
  const size_t N = ((size_t)n + 63) & ~(size_t)63;
  if (int rc = grow(f, f->cols, N * 48)) return rc;
  uint8_t* b = (uint8_t*)f->cols.p;
  BamColumns C;
  C.off = (uint32_t*)b;           b += 4 * N;
  C.ref = (int32_t*)b;            b += 4 * N;
  C.pos = (int32_t*)b;            b += 4 * N;
  C.mref = (int32_t*)b;           b += 4 * N;
  C.mpos = (int32_t*)b;           b += 4 * N;
  C.ncig = (uint32_t*)b;          b += 4 * N;
  C.lseq = (uint32_t*)b;          b += 4 * N;
  C.bytes = (uint32_t*)b;         b += 4 * N;
  C.span = (int32_t*)b;           b += 4 * N;
  C.flags = (uint16_t*)b;         b += 2 * N;
  C.mapq = b;                     b += N;
  C.lname = b;                    b += N;
  C.facts = b;
  launch_bam_fields(stream, ns, (const WalkStretch*)f->walk.p, (const uint32_t*)f->offs.p, C, f->stream);
  INF_TRY(f, hipGetLastError());
  const size_t m = (size_t)n;
  INF_TRY(f, hipMemcpyAsync(cols->off, C.off, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->ref_id, C.ref, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->pos, C.pos, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->mate_ref_id, C.mref, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->mate_pos, C.mpos, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->n_cigar, C.ncig, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->l_seq, C.lseq, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->bytes, C.bytes, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->span, C.span, 4 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->flags, C.flags, 2 * m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->mapq, C.mapq, m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->l_name, C.lname, m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipMemcpyAsync(cols->facts, C.facts, m, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipStreamSynchronize(f->stream));
  if (cols->names) {
    // the read names, for callers that pair reads (stitching): offsets = running sum of the name
    // lengths (host, from the column just fetched), then one gather on the device
    if (!cols->name_off) return ifail(f, BP_EINVAL, "bp_bam_scan: names wanted without name_off");
    uint64_t total = 0;
    for (size_t i = 0; i < m; i++) {
      cols->name_off[i] = (uint32_t)total;
      total += (cols->facts[i] & 4) ? 0u : cols->l_name[i] ? cols->l_name[i] - 1u : 0u;  // (fields outside the block: no name taken)
    }
    cols->name_off[m] = (uint32_t)total;
    if (total > (uint64_t)cols->names_cap || total > 0xffffffffull)
      return ifail(f, BP_ENOMEM, "bp_bam_scan: %llu bytes of read names, room for %lld", (unsigned long long)total,
                   (long long)cols->names_cap);
    if (total) {
      const size_t ob = ((m + 1) * 4 + 255) & ~(size_t)255;
      if (int rc = grow(f, f->names, ob + (size_t)total)) return rc;
      INF_TRY(f, hipMemcpyAsync(f->names.p, cols->name_off, (m + 1) * 4, hipMemcpyHostToDevice, f->stream));
      launch_bam_names(stream, n, C.off, (const uint32_t*)f->names.p, (uint8_t*)f->names.p + ob, f->stream);
      INF_TRY(f, hipGetLastError());
      INF_TRY(f, hipMemcpyAsync(cols->names, (const uint8_t*)f->names.p + ob, (size_t)total, hipMemcpyDeviceToHost, f->stream));
      INF_TRY(f, hipStreamSynchronize(f->stream));
    }
  }
  return BP_OK;
}

const uint8_t* bp_inflate_device_output(const bp_inflater* f) { return f ? (const uint8_t*)f->out.p : nullptr; }

int bp_inflate_stream(bp_inflater* f, void** stream) {
  if (!f || !stream) return ifail(f, BP_EINVAL, "bp_inflate_stream: null argument");
  *stream = (void*)f->stream;
  return BP_OK;
}

int bp_inflate_timing(const bp_inflater* f, float* inflate_ms, float* crc_ms) {
  if (!f) return BP_EINVAL;
  if (inflate_ms) *inflate_ms = f->inflate_ms;
  if (crc_ms) *crc_ms = f->crc_ms;
  return BP_OK;
}

size_t bp_deflate_bgzf_bound(size_t n) { return deflate_bgzf_bound(n); }

int bp_deflate_bgzf(bp_inflater* f, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* len) {
  if (!f || !len || (n && (!in || !out))) return ifail(f, BP_EINVAL, "bp_deflate_bgzf: null argument");
  *len = 0;
  if (n == 0) return BP_OK;
  if (cap < deflate_bgzf_bound(n))
    return ifail(f, BP_EINVAL, "bp_deflate_bgzf: buffer of %zu bytes for %zu of input (see bp_deflate_bgzf_bound)", cap, n);
  INF_TRY(f, hipSetDevice(f->device));
  if (!f->dfl) f->dfl = deflate_scratch_new();
  if (int rc = grow(f, f->dfl_in, n + 256)) return rc;
  INF_TRY(f, hipMemcpyAsync(f->dfl_in.p, in, n, hipMemcpyHostToDevice, f->stream));
  const uint8_t* d_out = nullptr;
  size_t total = 0;
  std::string err;
  if (int rc = deflate_bgzf_device(f->dfl, (const uint8_t*)f->dfl_in.p, n, f->stream, &d_out, &total, &err))
    return ifail(f, rc, "bp_deflate_bgzf: %s", err.c_str());
  INF_TRY(f, hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, f->stream));
  INF_TRY(f, hipStreamSynchronize(f->stream));
  *len = total;
  return BP_OK;
}

const char* bp_inflate_last_error(const bp_inflater* f) { return f ? f->err.c_str() : g_inflate_error.c_str(); }

}  // extern "C"
