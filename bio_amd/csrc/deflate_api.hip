// deflate_api.hip -- host side of the device's BGZF writer (deflate_kernels.hip): the scratch
// buffers of a run of members, the Huffman codes built from the run's symbol counts (RFC 1951
// 3.2.2: canonical codes from code lengths; lengths limited to 15 bits, 7 for the code-length
// code) and the dynamic block header that announces them (RFC 1951 3.2.7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bio_pileup.h"
#include "deflate_api.h"
#include "deflate_device.h"

namespace bp {

void launch_deflate_tokens(const uint8_t* text, uint64_t n_text, uint32_t* tok, uint32_t* ntok, uint32_t* hist,
                           hipStream_t s);
void launch_deflate_encode(const uint8_t* text, uint64_t n_text, const uint32_t* tok, const uint32_t* ntok,
                           const DeflateCodes* codes, const uint32_t* crc_tables, uint8_t* slots, uint32_t* sizes,
                           unsigned long long* offs, uint8_t* out, hipStream_t s);

// T and the four "advance by 256 zero bytes" tables of wave_crc32 (crc_device.h)
void crc_tables(uint32_t* t /*[1280]*/) {
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    t[n] = c;
  }
  for (int q = 0; q < 4; q++)
    for (uint32_t b = 0; b < 256; b++) {
      uint32_t v = b << (8 * q);
      for (int k = 0; k < 256; k++) v = t[v & 0xffu] ^ (v >> 8);
      t[256 + 256 * q + b] = v;
    }
}

namespace {
bool build_codes(const uint32_t* hist, uint64_t n_members, DeflateCodes* C);
}
namespace {

// Code lengths of a Huffman code for freq[0..n), none longer than maxbits; symbols that do not
// occur get length 0.  With fewer than two symbols in use a second one is given a code, so that
// the code is complete (inflaters refuse incomplete ones).
void huffman_lengths(const uint64_t* freq_in, int n, int maxbits, uint8_t* len) {
  std::vector<uint64_t> freq(freq_in, freq_in + n);
  int used = 0;
  for (int i = 0; i < n; i++) used += freq[i] != 0;
  for (int i = 0; i < n && used < 2; i++)
    if (freq[i] == 0) { freq[i] = 1; used++; }
  memset(len, 0, (size_t)n);
  // Huffman's algorithm on two queues: leaves sorted by count, internal nodes in creation order
  struct Node { uint64_t w; int a, b; };
  std::vector<int> leaves;
  for (int i = 0; i < n; i++)
    if (freq[i]) leaves.push_back(i);
  std::stable_sort(leaves.begin(), leaves.end(), [&](int x, int y) { return freq[x] < freq[y]; });
  std::vector<Node> nodes;  // [0, used) leaves, then internal ones
  for (int s : leaves) nodes.push_back({freq[s], -1, -1});
  size_t li = 0, ni = (size_t)used;
  auto take = [&]() {
    const bool leaf = li < (size_t)used && (ni >= nodes.size() || nodes[li].w <= nodes[ni].w);
    return (int)(leaf ? li++ : ni++);
  };
  while ((size_t)used - li + nodes.size() - ni > 1) {
    const int a = take(), b = take();
    nodes.push_back({nodes[(size_t)a].w + nodes[(size_t)b].w, a, b});
  }
  std::vector<int> depth(nodes.size(), 0);
  for (size_t k = nodes.size(); k-- > (size_t)used;) {
    depth[(size_t)nodes[k].a] = depth[k] + 1;
    depth[(size_t)nodes[k].b] = depth[k] + 1;
  }
  for (int k = 0; k < used; k++) len[leaves[(size_t)k]] = (uint8_t)std::min(depth[(size_t)k], 255);
  // Too deep: cut at maxbits, then lengthen the cheapest codes until the code fits (Kraft sum <= 1)
  // and shorten the dearest ones again until it is complete.  Units of 2^-maxbits.
  bool cut = false;
  for (int k = 0; k < used; k++)
    if (len[leaves[(size_t)k]] > maxbits) { len[leaves[(size_t)k]] = (uint8_t)maxbits; cut = true; }
  if (!cut) return;
  const uint64_t full = 1ull << maxbits;
  auto kraft = [&]() {
    uint64_t s = 0;
    for (int k = 0; k < used; k++) s += full >> len[leaves[(size_t)k]];
    return s;
  };
  uint64_t K = kraft();
  while (K > full) {  // the rarest symbol among those with the longest length below maxbits
    int best = -1;
    for (int k = 0; k < used; k++) {
      const int s = leaves[(size_t)k];
      if (len[s] < maxbits && (best < 0 || len[s] > len[best])) best = s;  // leaves are in rising count order
    }
    K -= full >> (len[best] + 1);
    len[best]++;
  }
  for (int l = maxbits; l >= 2 && K < full; l--)  // shortening a code of length l adds 2^(maxbits - l)
    for (int k = used - 1; k >= 0 && K < full; k--) {
      const int s = leaves[(size_t)k];
      if (len[s] == l && K + (full >> l) <= full) {
        len[s]--;
        K += full >> l;
      }
    }
  if (K != full) {  // (not seen; a flat complete code is always possible)
    int L = 1;
    while ((1 << L) < used) L++;
    const int n_short = (1 << L) - used;  // codes of length L - 1
    for (int k = 0; k < used; k++) len[leaves[(size_t)used - 1 - (size_t)k]] = (uint8_t)(k < n_short ? L - 1 : L);
  }
}

// canonical codes (RFC 1951 3.2.2), bit-reversed, with the length in bits 16-19
void canonical_codes(const uint8_t* len, int n, uint32_t* out) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[len[i]]++;
  count[0] = 0;
  uint32_t next[16] = {0}, code = 0;
  for (int b = 1; b < 16; b++) {
    code = (code + (uint32_t)count[b - 1]) << 1;
    next[b] = code;
  }
  for (int i = 0; i < n; i++) {
    out[i] = 0;
    if (!len[i]) continue;
    const uint32_t c = next[len[i]]++;
    uint32_t r = 0;
    for (int b = 0; b < len[i]; b++) r |= ((c >> b) & 1u) << (len[i] - 1 - b);
    out[i] = r | ((uint32_t)len[i] << 16);
  }
}

struct Bits {
  std::vector<uint32_t> w;
  uint32_t n = 0;
  void put(uint32_t v, int nb) {
    for (int b = 0; b < nb; b++, n++) {
      if ((n >> 5) >= w.size()) w.push_back(0);
      w[n >> 5] |= ((v >> b) & 1u) << (n & 31u);
    }
  }
};

// hist -> codes + header.  False when the header does not fit kDflHdrWords (never: <= 2,300 bits).
bool build_codes(const uint32_t* hist, uint64_t n_members, DeflateCodes* C) {
  uint64_t fl[286], fd[30];
  for (int i = 0; i < 286; i++) fl[i] = hist[i];
  for (int i = 0; i < 30; i++) fd[i] = hist[288 + i];
  fl[256] = n_members;  // one end-of-block per member
  uint8_t ll[286], dl[30];
  huffman_lengths(fl, 286, 15, ll);
  huffman_lengths(fd, 30, 15, dl);
  memset(C, 0, sizeof *C);
  canonical_codes(ll, 286, C->ll);
  canonical_codes(dl, 30, C->dist);
  int hlit = 286, hdist = 30;
  while (hlit > 257 && ll[hlit - 1] == 0) hlit--;
  while (hdist > 1 && dl[hdist - 1] == 0) hdist--;
  // the code lengths as one sequence, run-length coded with the symbols 16, 17, 18
  std::vector<uint8_t> seq(ll, ll + hlit);
  seq.insert(seq.end(), dl, dl + hdist);
  struct Cl { uint8_t sym, ext; };
  std::vector<Cl> cl;
  for (size_t i = 0; i < seq.size();) {
    size_t run = 1;
    while (i + run < seq.size() && seq[i + run] == seq[i]) run++;
    if (seq[i] == 0 && run >= 3) {
      const size_t r = std::min<size_t>(run, 138);
      if (r <= 10) cl.push_back({17, (uint8_t)(r - 3)}); else cl.push_back({18, (uint8_t)(r - 11)});
      i += r;
    } else if (seq[i] != 0 && run >= 4) {  // the length itself, then "repeat previous" 3-6 times
      cl.push_back({seq[i], 0});
      const size_t r = std::min<size_t>(run - 1, 6);
      cl.push_back({16, (uint8_t)(r - 3)});
      i += 1 + r;
    } else {
      cl.push_back({seq[i], 0});
      i++;
    }
  }
  uint64_t fc[19] = {0};
  for (const Cl& c : cl) fc[c.sym]++;
  uint8_t cll[19];
  huffman_lengths(fc, 19, 7, cll);
  uint32_t clc[19];
  canonical_codes(cll, 19, clc);
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cll[order[hclen - 1]] == 0) hclen--;
  Bits b;
  b.put(1, 1);  // BFINAL
  b.put(2, 2);  // BTYPE = dynamic Huffman
  b.put((uint32_t)(hlit - 257), 5);
  b.put((uint32_t)(hdist - 1), 5);
  b.put((uint32_t)(hclen - 4), 4);
  for (int k = 0; k < hclen; k++) b.put(cll[order[k]], 3);
  for (const Cl& c : cl) {
    b.put(clc[c.sym] & 0xffffu, (int)(clc[c.sym] >> 16));
    if (c.sym == 16) b.put(c.ext, 2);
    if (c.sym == 17) b.put(c.ext, 3);
    if (c.sym == 18) b.put(c.ext, 7);
  }
  if (b.w.size() > (size_t)kDflHdrWords) return false;
  C->hdr_bits = b.n;
  std::copy(b.w.begin(), b.w.end(), C->hdr);
  return true;
}

struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct DeflateScratch {
  DBuf tok, meta, slots, out, codes;
  uint32_t* tables = nullptr;
  uint32_t* h_hist = nullptr;          // pinned
  DeflateCodes* h_codes = nullptr;     // pinned
  unsigned long long* h_total = nullptr;
};

DeflateScratch* deflate_scratch_new() { return new DeflateScratch(); }

// A job's handle lives for one job, the buffers are worth keeping (half a gigabyte of device memory
// and three pinned allocations per chunk size): handles borrow a scratch from a small pool per
// device and give it back.  What is still in the pool when the process ends stays there.
namespace {
struct ScratchPool {
  std::mutex mu;
  std::vector<std::pair<int, DeflateScratch*>> v;
};
ScratchPool& scratch_pool() {
  static ScratchPool* p = new ScratchPool();
  return *p;
}
}  // namespace

DeflateScratch* deflate_scratch_acquire(int device) {
  ScratchPool& sp = scratch_pool();
  {
    std::unique_lock<std::mutex> lk(sp.mu);
    for (size_t k = 0; k < sp.v.size(); k++)
      if (sp.v[k].first == device) {
        DeflateScratch* sc = sp.v[k].second;
        sp.v.erase(sp.v.begin() + (long)k);
        return sc;
      }
  }
  return new DeflateScratch();
}

void deflate_scratch_release(int device, DeflateScratch* sc) {
  if (!sc) return;
  ScratchPool& sp = scratch_pool();
  {
    std::unique_lock<std::mutex> lk(sp.mu);
    size_t same = 0;
    for (const auto& e : sp.v) same += e.first == device;
    if (same < 4) {
      sp.v.emplace_back(device, sc);
      return;
    }
  }
  deflate_scratch_free(sc);
}

void deflate_scratch_free(DeflateScratch* sc) {
  if (!sc) return;
  for (DBuf* b : {&sc->tok, &sc->meta, &sc->slots, &sc->out, &sc->codes})
    if (b->p) (void)hipFree(b->p);
  if (sc->tables) (void)hipFree(sc->tables);
  if (sc->h_hist) (void)hipHostFree(sc->h_hist);
  if (sc->h_codes) (void)hipHostFree(sc->h_codes);
  if (sc->h_total) (void)hipHostFree(sc->h_total);
  delete sc;
}

size_t deflate_bgzf_bound(size_t n) { return n + 31 * ((n + kDflMember - 1) / kDflMember); }

int deflate_bgzf_device(DeflateScratch* sc, const uint8_t* d_text, size_t n, hipStream_t stream,
                        const uint8_t** d_out, size_t* total, std::string* err) {
  *d_out = nullptr;
  *total = 0;
  if (n == 0) return BP_OK;
  auto fail = [&](int code, const char* what, hipError_t e) {
    *err = std::string(what) + ": " + hipGetErrorString(e);
    return code;
  };
#define DFL_TRY(call)                                              \
  do {                                                             \
    hipError_t e_ = (call);                                        \
    if (e_ != hipSuccess) return fail(BP_EDEVICE, #call, e_);      \
  } while (0)
  auto grow = [&](DBuf& b, size_t need) -> hipError_t {
    if (need <= b.cap) return hipSuccess;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t cap = need + need / 4 + 4096;
    const hipError_t e = hipMalloc(&b.p, cap);
    if (e == hipSuccess) b.cap = cap;
    return e;
  };
  const size_t n_members = (n + kDflMember - 1) / kDflMember;
  if (!sc->tables) {
    std::vector<uint32_t> t(1280);
    crc_tables(t.data());
    DFL_TRY(hipMalloc((void**)&sc->tables, 1280 * 4));
    DFL_TRY(hipMemcpy(sc->tables, t.data(), 1280 * 4, hipMemcpyHostToDevice));
    DFL_TRY(hipHostMalloc((void**)&sc->h_hist, kDflHist * 4, hipHostMallocDefault));
    DFL_TRY(hipHostMalloc((void**)&sc->h_codes, sizeof(DeflateCodes), hipHostMallocDefault));
    DFL_TRY(hipHostMalloc((void**)&sc->h_total, 8, hipHostMallocDefault));
  }
  // meta: hist [320] | ntok [members x 8] | sizes [members] | (8-byte aligned) offs [members + 1]
  const size_t off_ntok = kDflHist * 4, off_sizes = off_ntok + n_members * kDflSegs * 4;
  const size_t off_offs = (off_sizes + n_members * 4 + 7) & ~(size_t)7;
  if (hipError_t e = grow(sc->tok, n * 4 + 256)) return fail(BP_ENOMEM, "hipMalloc", e);
  if (hipError_t e = grow(sc->meta, off_offs + (n_members + 1) * 8)) return fail(BP_ENOMEM, "hipMalloc", e);
  if (hipError_t e = grow(sc->slots, n_members * (size_t)kDflSlot)) return fail(BP_ENOMEM, "hipMalloc", e);
  if (hipError_t e = grow(sc->out, deflate_bgzf_bound(n) + 256)) return fail(BP_ENOMEM, "hipMalloc", e);
  if (hipError_t e = grow(sc->codes, sizeof(DeflateCodes))) return fail(BP_ENOMEM, "hipMalloc", e);
  uint8_t* const meta = (uint8_t*)sc->meta.p;
  uint32_t* const hist = (uint32_t*)meta;
  uint32_t* const ntok = (uint32_t*)(meta + off_ntok);
  uint32_t* const sizes = (uint32_t*)(meta + off_sizes);
  unsigned long long* const offs = (unsigned long long*)(meta + off_offs);
  DFL_TRY(hipMemsetAsync(hist, 0, kDflHist * 4, stream));
  launch_deflate_tokens(d_text, n, (uint32_t*)sc->tok.p, ntok, hist, stream);
  DFL_TRY(hipGetLastError());
  DFL_TRY(hipMemcpyAsync(sc->h_hist, hist, kDflHist * 4, hipMemcpyDeviceToHost, stream));
  DFL_TRY(hipStreamSynchronize(stream));
  if (!build_codes(sc->h_hist, n_members, sc->h_codes)) {
    *err = "bgzf: the block header of the device's Huffman codes does not fit";
    return BP_EDEVICE;
  }
  DFL_TRY(hipMemcpyAsync(sc->codes.p, sc->h_codes, sizeof(DeflateCodes), hipMemcpyHostToDevice, stream));
  launch_deflate_encode(d_text, n, (const uint32_t*)sc->tok.p, ntok, (const DeflateCodes*)sc->codes.p, sc->tables,
                        (uint8_t*)sc->slots.p, sizes, offs, (uint8_t*)sc->out.p, stream);
  DFL_TRY(hipGetLastError());
  DFL_TRY(hipMemcpyAsync(sc->h_total, offs + n_members, 8, hipMemcpyDeviceToHost, stream));
  DFL_TRY(hipStreamSynchronize(stream));
#undef DFL_TRY
  if (*sc->h_total > deflate_bgzf_bound(n)) {
    *err = "bgzf: the device's members are longer than their bound";
    return BP_EDEVICE;
  }
  *d_out = (const uint8_t*)sc->out.p;
  *total = (size_t)*sc->h_total;
  return BP_OK;
}

}  // namespace bp

// Host half of the writer by itself (no device): the codes and the block header for a histogram.
extern "C" int bp_deflate_codes(const uint32_t* hist, uint64_t n_members, uint32_t* out) {
  if (!hist || !out) return BP_EINVAL;
  bp::DeflateCodes c;
  if (!bp::build_codes(hist, n_members, &c)) return BP_EDEVICE;
  memcpy(out, &c, sizeof c);
  return BP_OK;
}
