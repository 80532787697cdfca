// deflate_kernels.hip -- text that is already on the device leaves it as BGZF members.
//
// The reference's `-bgz` formats hand the rendered lines to a bgzf.Writer, which compresses
// 64 KiB blocks on `parallelism` host threads (pileup/snp/output.go:66-82, :388-392; biogo/hts
// bgzf, absent from the tree).  BGZF (SAM spec 4.1) and DEFLATE (RFC 1951) are public formats and
// a compressor is free in the bytes it produces: what is pinned is that every member inflates
// (zlib in the tests) to exactly the text the plain format writes.  Here the lines are rendered on
// the device (text_write_kernel), so compressing them there means a fifth of the bytes cross the
// bus and reach the file, and no host thread deflates.
//
//   deflate_tokens_kernel   LZ77.  A member (0xff00 bytes of text) is cut into 32 segments, a WAVE
//                           each.  64 consecutive positions per step, lane = position: hash of the
//                           4 bytes there, candidates = the last four earlier positions with that
//                           hash (LDS table of 512 x 4 16-bit positions per wave, primed with the
//                           512 bytes in front of the segment) and the window's own first position
//                           with it, match lengths by comparing dwords, the longest wins; a longer
//                           match one byte on makes this byte a literal (one step of zlib's lazy rule).
//                           Which positions start a token is a serial question (a match skips
//                           the positions it covers): the wave answers it on the scalar unit,
//                           v_readlane of the step from position to position, typically 10-15
//                           hops per 64 bytes.  Tokens go out in order (ballot ranks), their
//                           symbols into a histogram (LDS, then one global one per run).
//   host                    Huffman codes (length-limited, canonical) from the run's histogram and
//                           the dynamic block header that announces them (deflate_api.hip).
//   deflate_encode_kernel   a WAVE per member: lane = token, code + extra bits (up to 48) looked up
//                           in LDS, a prefix sum of the bit counts places every token in the bit
//                           stream, ds_or into a ring of dwords in LDS, complete dwords flushed with
//                           coalesced stores.  BGZF header, CRC-32 (crc_device.h) and ISIZE around it.
//                           A member that would come out larger than stored is written stored.
//   deflate_offsets_kernel  prefix sum of the member sizes;
//   deflate_pack_kernel     members moved from their slots to one contiguous run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "crc_device.h"
#include "deflate_device.h"

namespace bp {

namespace {

// Match search, measured on 268 MB of basestrand-tsv lines (tools/deflate_bench.py; zlib -1 4.56x, -6 6.11x):
// one candidate per hash 4.90x; + one step of lazy evaluation 5.11x; two candidates 5.34x; + the
// window's own first position per hash 5.45x; four candidates 5.59x (tokeniser 3.2 -> 7 ms).
constexpr int kWays = 4;          // candidates kept per hash: the last 1, 2 or 4 positions
constexpr int kHashBits = 9;      // sets of the table: 512 x 4 positions = 4 KB of LDS per wave
// Table size and segment length against a whole run (chr1 30x basestrand-tsv-bgz end to end, same box,
// alternating libraries; ratio on the 268 MB of lines): 1,024 sets and 8 segments a member 1.06 s 5.59x;
// 512 / 16: 0.99 s 5.50x; 256 / 16: 0.98 s 5.46x; 512 / 32: 0.96 s 5.47x; 256 / 32: 0.95 s 5.44x.  The run is
// bound by the device's total kernel time: more and shorter waves (28 a CU instead of 16) fill the gaps
// the reader's and the jobs' kernels leave, and a chunk's tokeniser finishes in a third of the time.
constexpr bool kLazy = true;      // a longer match one byte on turns this byte into a literal
constexpr bool kIntra = true;     // candidates inside the current window of 64 positions
constexpr uint32_t kWarm = 512;
constexpr uint32_t kMinMatch = 4, kMaxMatch = 258, kMaxDist = 32768;

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// 4 bytes at any address: two aligned dwords and v_alignbyte (reads up to 7 bytes behind p)
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)a & 3u);
}

// 16 bytes at any address (reads up to 19 bytes behind them)
__device__ __forceinline__ void ld128(const uint8_t* p, uint32_t (&o)[4]) {
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)a & 3u;
  const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
  o[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
  o[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
  o[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
  o[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// RFC 1951 3.2.5: length 3..258 -> symbol 257..285, extra bits
__device__ __forceinline__ void len_symbol(uint32_t len, uint32_t& sym, uint32_t& nb, uint32_t& ext) {
  const uint32_t l = len - 3;
  nb = 0;
  ext = 0;
  if (len == 258) { sym = 285; return; }
  if (l < 8) { sym = 257 + l; return; }
  nb = (31u - (uint32_t)__builtin_clz(l)) - 2u;
  sym = 261 + 4 * nb + ((l >> nb) & 3u);
  ext = l & ((1u << nb) - 1u);
}
// distance - 1 (0..32767) -> symbol 0..29, extra bits
__device__ __forceinline__ void dist_symbol(uint32_t dm1, uint32_t& sym, uint32_t& nb, uint32_t& ext) {
  nb = 0;
  ext = 0;
  if (dm1 < 4) { sym = dm1; return; }
  nb = (31u - (uint32_t)__builtin_clz(dm1)) - 1u;
  sym = 2 * nb + 2 + ((dm1 >> nb) & 1u);
  ext = dm1 & ((1u << nb) - 1u);
}

}  // namespace

// tok: [n_text] dwords, a segment's tokens from the index of its first byte on; ntok: [members x 8];
// hist: [kDflHist], zeroed by the caller.
__global__ __launch_bounds__(256) void deflate_tokens_kernel(const uint8_t* __restrict__ text, uint64_t n_text,
                                                              uint32_t* __restrict__ tok, uint32_t* __restrict__ ntok,
                                                              uint32_t* __restrict__ hist) {
  __shared__ uint16_t tabs[4][kWays << kHashBits];
  __shared__ uint32_t lh[kDflHist];
  __shared__ uint32_t firsts[4][256];
  const int lane = (int)(threadIdx.x & 63);
  const int wave = (int)rfl(threadIdx.x >> 6);
  uint16_t* const tab = tabs[wave];
  uint32_t* const first = firsts[wave];  // the lowest position of the current window per (short) hash
  for (int k = lane; k < 256; k += 64) first[k] = 0xffffffffu;
  for (int k = (int)threadIdx.x; k < kDflHist; k += 256) lh[k] = 0;
  for (int k = lane; k < (kWays << kHashBits) / 2; k += 64) ((uint32_t*)tab)[k] = 0xffffffffu;
  __syncthreads();
  const uint64_t n_members = (n_text + kDflMember - 1) / kDflMember;
  const uint64_t unit = (uint64_t)blockIdx.x * 4 + (uint64_t)wave;
  if (unit < n_members * kDflSegs) {
    const uint64_t member = unit / kDflSegs;
    const uint32_t seg = (uint32_t)(unit % kDflSegs);
    const uint8_t* const base = text + member * kDflMember;
    const uint64_t left = n_text - member * kDflMember;
    const uint32_t mlen = left < (uint64_t)kDflMember ? (uint32_t)left : (uint32_t)kDflMember;
    const uint32_t s0 = seg * kDflSeg;
    const uint32_t s1 = s0 + kDflSeg < mlen ? s0 + kDflSeg : mlen;  // the segment is [s0, s1) of the member
    uint32_t* const out = tok + member * kDflMember + s0;
    uint32_t count = 0;
    if (s0 < s1) {
      auto hash = [](uint32_t v) { return (v * 2654435761u) >> (32 - kHashBits); };
      // the bytes in front of the segment (the neighbouring wave's) prime the table
      // a bucket is {newest, the one before} when two are kept
      auto lookup = [&](uint32_t hv, uint32_t (&c)[4]) {
        c[0] = c[1] = c[2] = c[3] = 0xffffu;
        if (kWays == 4) {
          const uint2 e = ((const uint2*)tab)[hv];
          c[0] = e.x & 0xffffu; c[1] = e.x >> 16; c[2] = e.y & 0xffffu; c[3] = e.y >> 16;
        } else if (kWays == 2) {
          const uint32_t e = ((const uint32_t*)tab)[hv];
          c[0] = e & 0xffffu;
          c[1] = e >> 16;
        } else {
          c[0] = tab[hv];
        }
      };
      auto insert = [&](uint32_t hv, uint32_t pos, const uint32_t (&c)[4]) {
        if (kWays == 4) ((uint2*)tab)[hv] = make_uint2(pos | (c[0] << 16), c[1] | (c[2] << 16));
        else if (kWays == 2) ((uint32_t*)tab)[hv] = pos | (c[0] << 16);
        else tab[hv] = (uint16_t)pos;
      };
      // equal leading bytes of two 16-byte stretches
      auto head16 = [](const uint32_t (&a)[4], const uint32_t (&b)[4]) {
        const uint32_t x0 = a[0] ^ b[0], x1 = a[1] ^ b[1], x2 = a[2] ^ b[2], x3 = a[3] ^ b[3];
        if (x0) return (uint32_t)__builtin_ctz(x0) >> 3;
        if (x1) return 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
        if (x2) return 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
        if (x3) return 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
        return 16u;
      };
      // a match known to hold for its first 16 bytes, followed further (16 bytes a round trip)
      auto extend = [&](uint32_t cand, uint32_t pos, uint32_t maxlen) {
        uint32_t l = 16;
        while (l < maxlen) {
          uint32_t a[4], b[4];
          ld128(base + cand + l, a);
          ld128(base + pos + l, b);
          const uint32_t h = head16(a, b);
          l += h;
          if (h < 16) break;
        }
        return l;
      };
      for (uint32_t p = (s0 > kWarm ? s0 - kWarm : 0u) + (uint32_t)lane; p < s0; p += 64) {
        const uint32_t hv = hash(ld32(base + p));
        uint32_t c[4];
        lookup(hv, c);
        insert(hv, p, c);
      }
      uint32_t carry = 0;  // positions from the window's start on that an earlier match covers
      for (uint32_t p = s0; p < s1; p += 64) {
        const uint32_t pos = p + (uint32_t)lane;
        const bool in = pos < s1;
        const uint32_t v = in ? ld32(base + pos) : 0u;
        const bool hashable = in && pos + 4 <= mlen;
        uint32_t c[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
        const uint32_t hv = hash(v);
        if (hashable) lookup(hv, c);
        __builtin_amdgcn_wave_barrier();  // every lane has read before any writes (the LDS works a wave's requests in order)
        if (hashable) insert(hv, pos, c);
        // the table knows earlier windows only; inside the window, the first position with the same
        // (8-bit) hash is a candidate for all later ones
        uint32_t cw = 0xffffu;
        if (kIntra) {
          if (hashable) atomicMin(&first[hv & 255u], pos);
          __builtin_amdgcn_wave_barrier();
          if (hashable) {
            cw = first[hv & 255u];
            if (cw >= pos) cw = 0xffffu;
          }
          __builtin_amdgcn_wave_barrier();
          if (hashable) first[hv & 255u] = 0xffffffffu;
        }
        if (carry >= 64) {  // the whole window lies inside a match
          carry -= 64;
          continue;
        }
        const uint32_t room = in ? s1 - pos : 0u;
        const uint32_t maxlen = room < kMaxMatch ? room : kMaxMatch;
        // All candidates at once: 16 bytes of each (one round trip for the lot; most matches of this
        // text end inside them), the longest wins, the nearest on equal lengths (the window's own
        // first, then the table's from the newest on).  Only a match that holds for all 16 is followed.
        uint32_t len = 0, cand = 0xffffu;
        if (maxlen >= kMinMatch) {
          uint32_t cs[kWays + 1], ls[kWays + 1];
          cs[0] = kIntra ? cw : 0xffffu;
#pragma unroll
          for (int k = 0; k < kWays; k++) cs[k + 1] = c[k];
          uint32_t B[4];
          ld128(base + pos, B);
#pragma unroll
          for (int k = 0; k <= kWays; k++) {
            const bool ok = cs[k] != 0xffffu && pos - cs[k] <= kMaxDist;
            uint32_t A[4];
            ld128(base + (ok ? cs[k] : pos), A);
            ls[k] = ok ? head16(A, B) : 0u;
          }
#pragma unroll
          for (int k = 0; k <= kWays; k++) {
            uint32_t l = ls[k];
            if (l == 16 && maxlen > 16 && len < maxlen) l = extend(cs[k], pos, maxlen);
            l = l < maxlen ? l : maxlen;
            if (l > len) { len = l; cand = cs[k]; }
          }
          if (len < kMinMatch) len = 0;
        }
        if (kLazy) {  // a longer match one byte on: this byte goes out as a literal (zlib's lazy evaluation, one step)
          const uint32_t nl = (uint32_t)__shfl_down((int)len, 1, 64);
          if (lane < 63 && len && nl > len) len = 0;
        }
        // which positions start a token: from `carry` on, every token to the next
        const uint32_t step = len ? len : 1u;
        const uint32_t nin = s1 - p < 64u ? s1 - p : 64u;
        unsigned long long on = 0;
        uint32_t cur = carry;
        while (cur < nin) {
          on |= 1ull << cur;
          cur += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)cur);
        }
        carry = cur - nin;
        if ((on >> lane) & 1ull) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(on >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)on, 0u));
          uint32_t t = v & 0xffu, sym = t, nb, ext;
          if (len) {
            const uint32_t dm1 = pos - cand - 1u;
            t = (len << 16) | dm1;
            len_symbol(len, sym, nb, ext);
            uint32_t ds;
            dist_symbol(dm1, ds, nb, ext);
            atomicAdd(&lh[288 + ds], 1u);
          }
          atomicAdd(&lh[sym], 1u);
          out[count + rank] = t;
        }
        count += (uint32_t)__builtin_popcountll(on);
      }
    }
    if (lane == 0) ntok[unit] = count;
  }
  __syncthreads();
  for (int k = (int)threadIdx.x; k < kDflHist; k += 256)
    if (lh[k]) atomicAdd(&hist[k], lh[k]);
}

namespace {

// The bit stream of one member from byte 16 of its slot on (bytes 16-17 are BSIZE, filled in at
// the end; the DEFLATE data starts at bit 16).
struct BitWriter {
  uint32_t* ring;   // 256 dwords of LDS, dword k of the stream at ring[k & 255]; zero where nothing was put
  uint32_t* g;      // the stream in global memory
  uint32_t bitpos;  // bits put so far
  uint32_t flushed; // dwords written to g
  uint32_t first;   // dword 0, kept back until BSIZE is known
};

// every lane puts the low nb (<= 64) bits of val, lane 0's first
__device__ __forceinline__ void put(BitWriter& w, unsigned long long val, uint32_t nb, int lane, bool all) {
  const uint32_t incl = wave_incl_scan(nb, lane);
  if (nb) {
    const uint32_t off = w.bitpos + incl - nb;
    const uint32_t wi = off >> 5, sh = off & 31u;
    const uint32_t v0 = (uint32_t)val, v1 = (uint32_t)(val >> 32);
    const uint32_t d0 = v0 << sh;
    const uint32_t d1 = sh ? (v0 >> (32u - sh)) | (v1 << sh) : v1;
    const uint32_t d2 = sh ? v1 >> (32u - sh) : 0u;
    atomicOr(&w.ring[wi & 255u], d0);
    if (d1) atomicOr(&w.ring[(wi + 1u) & 255u], d1);
    if (d2) atomicOr(&w.ring[(wi + 2u) & 255u], d2);
  }
  w.bitpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  __builtin_amdgcn_wave_barrier();
  // complete dwords (or, at the end of the member, the last one as well) leave the ring
  const uint32_t end = all ? (w.bitpos + 31u) >> 5 : w.bitpos >> 5;
  uint32_t keep = 0;
  for (uint32_t k = w.flushed + (uint32_t)lane; k < end; k += 64) {
    const uint32_t v = w.ring[k & 255u];
    w.ring[k & 255u] = 0;
    if (k) w.g[k] = v; else keep = v;
  }
  if (w.flushed == 0 && end > 0) w.first = (uint32_t)__builtin_amdgcn_readlane((int)keep, 0);
  w.flushed = end;
  __builtin_amdgcn_wave_barrier();
}

}  // namespace

// slots: [members] x kDflSlot bytes (16-byte aligned); sizes: [members] bytes of each finished member
__global__ __launch_bounds__(256) void deflate_encode_kernel(const uint8_t* __restrict__ text, uint64_t n_text,
                                                              const uint32_t* __restrict__ tok,
                                                              const uint32_t* __restrict__ ntok,
                                                              const DeflateCodes* __restrict__ codes,
                                                              const uint32_t* __restrict__ crc_tables,
                                                              uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes) {
  __shared__ uint32_t T[256], W[4][256];
  __shared__ uint32_t ll[288], dc[32];
  __shared__ uint32_t rings[4][256];
  for (int k = (int)threadIdx.x; k < 256; k += 256) {
    T[k] = crc_tables[k];
#pragma unroll
    for (int q = 0; q < 4; q++) W[q][k] = crc_tables[256 + 256 * q + k];
    rings[0][k] = rings[1][k] = rings[2][k] = rings[3][k] = 0;
  }
  for (int k = (int)threadIdx.x; k < 288; k += 256) ll[k] = codes->ll[k];
  if (threadIdx.x < 32) dc[threadIdx.x] = codes->dist[threadIdx.x];
  __syncthreads();
  const int lane = (int)(threadIdx.x & 63);
  const int wave = (int)rfl(threadIdx.x >> 6);
  const uint64_t n_members = (n_text + kDflMember - 1) / kDflMember;
  const uint64_t m = (uint64_t)blockIdx.x * 4 + (uint64_t)wave;
  if (m >= n_members) return;  // (no workgroup barrier below)
  const uint8_t* const base = text + m * kDflMember;
  const uint64_t left = n_text - m * kDflMember;
  const uint32_t mlen = left < (uint64_t)kDflMember ? (uint32_t)left : (uint32_t)kDflMember;
  uint8_t* const slot = slots + m * kDflSlot;
  // gzip member header with the BC extra field (SAM spec 4.1): bytes 0-15, BSIZE behind them
  if (lane < 4) ((uint32_t*)slot)[lane] = lane == 0 ? 0x04088b1fu : lane == 1 ? 0u : lane == 2 ? 0x0006ff00u : 0x00024342u;
  const uint32_t crc = wave_crc32(base, mlen, T, W, lane);
  const uint32_t stored_total = mlen + 31u;  // 18 header + 5 stored-block header + text + 8 trailer
  BitWriter w{rings[wave], (uint32_t*)(slot + 16), 16u, 0u, 0u};
  // the member is given up as DEFLATE as soon as it is longer than the stored form would be
  auto too_long = [&]() { return 16u + ((w.bitpos + 7u) >> 3) + 8u > stored_total; };
  bool stored = false;
  {  // block header: BFINAL, BTYPE = dynamic, the code lengths
    const uint32_t hb = codes->hdr_bits;
    for (uint32_t i0 = 0; i0 * 32u < hb && !stored; i0 += 64) {
      const uint32_t i = i0 + (uint32_t)lane;
      const uint32_t nb = i * 32u < hb ? (hb - i * 32u < 32u ? hb - i * 32u : 32u) : 0u;
      put(w, nb ? codes->hdr[i] : 0u, nb, lane, false);
      stored = too_long();
    }
  }
  for (uint32_t s = 0; s < (uint32_t)kDflSegs && !stored; s++) {
    const uint32_t n = rfl(ntok[m * kDflSegs + s]);
    const uint32_t* const tp = tok + m * kDflMember + (uint64_t)s * kDflSeg;
    for (uint32_t i0 = 0; i0 < n && !stored; i0 += 64) {
      unsigned long long val = 0;
      uint32_t nb = 0;
      if (i0 + (uint32_t)lane < n) {
        const uint32_t t = tp[i0 + (uint32_t)lane];
        const uint32_t len = t >> 16;
        if (len == 0) {
          const uint32_t c = ll[t & 0xffu];
          val = c & 0xffffu;
          nb = c >> 16;
        } else {
          uint32_t ls, lnb, lext, ds, dnb, dext;
          len_symbol(len, ls, lnb, lext);
          dist_symbol(t & 0x7fffu, ds, dnb, dext);
          const uint32_t c1 = ll[ls], c2 = dc[ds];
          const uint32_t n1 = c1 >> 16, n2 = c2 >> 16;
          val = (unsigned long long)(c1 & 0xffffu) | ((unsigned long long)lext << n1) |
                ((unsigned long long)(c2 & 0xffffu) << (n1 + lnb)) | ((unsigned long long)dext << (n1 + lnb + n2));
          nb = n1 + lnb + n2 + dnb;
        }
      }
      put(w, val, nb, lane, false);
      stored = too_long();
    }
  }
  uint32_t total;
  if (!stored) {
    {  // end of block, then zero bits up to a byte boundary
      const uint32_t c = ll[256];
      put(w, lane == 0 ? c & 0xffffu : 0u, lane == 0 ? c >> 16 : 0u, lane, false);
      w.bitpos = (w.bitpos + 7u) & ~7u;
    }
    stored = 16u + (w.bitpos >> 3) + 8u > stored_total;
  }
  if (!stored) {
    // CRC-32 and ISIZE
    put(w, lane == 0 ? crc : mlen, lane < 2 ? 32u : 0u, lane, true);
    total = 16u + (w.bitpos >> 3);
    if (lane == 0) w.g[0] = w.first | (total - 1u);  // BSIZE = member length - 1
  } else {
    // stored block: BFINAL = 1, BTYPE = 00, LEN, ~LEN, the bytes
    total = stored_total;
    if (lane == 0) {
      const uint32_t bs = total - 1u;
      slot[16] = (uint8_t)bs; slot[17] = (uint8_t)(bs >> 8);
      slot[18] = 1;
      slot[19] = (uint8_t)mlen; slot[20] = (uint8_t)(mlen >> 8);
      slot[21] = (uint8_t)~mlen; slot[22] = (uint8_t)(~mlen >> 8);
      uint8_t* const tl = slot + 23 + mlen;
      for (int k = 0; k < 4; k++) tl[k] = (uint8_t)(crc >> (8 * k));
      for (int k = 0; k < 4; k++) tl[4 + k] = (uint8_t)(mlen >> (8 * k));
    }
    for (uint32_t k = (uint32_t)lane; k < mlen; k += 64) slot[23 + k] = base[k];
  }
  if (lane == 0) sizes[m] = total;
}

// offs[0..n] = exclusive prefix sum of sizes[0..n) (one workgroup of 1024)
__global__ __launch_bounds__(1024) void deflate_offsets_kernel(const uint32_t* __restrict__ sizes, int64_t n,
                                                                unsigned long long* __restrict__ offs) {
  __shared__ unsigned long long sh[1024];
  const int64_t per = (n + 1023) / 1024;
  const int64_t a = (int64_t)threadIdx.x * per, b = a + per < n ? a + per : n;
  unsigned long long s = 0;
  for (int64_t k = a; k < b; k++) s += sizes[k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0ull;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = sh[threadIdx.x] - s;
  for (int64_t k = a; k < b; k++) {
    offs[k] = run;
    run += sizes[k];
  }
  if (threadIdx.x == 1023) offs[n] = sh[1023];
}

// member blockIdx.x from its slot to out + offs[m]; dwords of the output that lie inside the member
// are stored whole, the shared ones at its ends byte by byte
__global__ __launch_bounds__(256) void deflate_pack_kernel(const uint8_t* __restrict__ slots,
                                                            const uint32_t* __restrict__ sizes,
                                                            const unsigned long long* __restrict__ offs,
                                                            uint8_t* __restrict__ out) {
  const uint64_t m = blockIdx.x;
  const uint8_t* const src = slots + m * kDflSlot;
  const long long s = (long long)sizes[m];
  const long long d0 = (long long)offs[m];
  const long long w0 = d0 >> 2, w1 = (d0 + s - 1) >> 2;
  for (long long wd = w0 + (long long)threadIdx.x; wd <= w1; wd += 256) {
    const long long b0 = wd * 4 - d0;  // where the dword's first byte lies in the member
    if (b0 >= 0 && b0 + 4 <= s) {
      ((uint32_t*)out)[wd] = ld32(src + b0);
    } else {
      for (int k = 0; k < 4; k++)
        if (b0 + k >= 0 && b0 + k < s) out[wd * 4 + k] = src[b0 + k];
    }
  }
}

void launch_deflate_tokens(const uint8_t* text, uint64_t n_text, uint32_t* tok, uint32_t* ntok, uint32_t* hist,
                           hipStream_t s) {
  const uint64_t n_members = (n_text + kDflMember - 1) / kDflMember;
  if (n_members == 0) return;
  hipLaunchKernelGGL(deflate_tokens_kernel, dim3((unsigned)((n_members * kDflSegs + 3) / 4)), dim3(256), 0, s, text, n_text, tok,
                     ntok, hist);
}
void launch_deflate_encode(const uint8_t* text, uint64_t n_text, const uint32_t* tok, const uint32_t* ntok,
                           const DeflateCodes* codes, const uint32_t* crc_tables, uint8_t* slots, uint32_t* sizes,
                           unsigned long long* offs, uint8_t* out, hipStream_t s) {
  const uint64_t n_members = (n_text + kDflMember - 1) / kDflMember;
  if (n_members == 0) return;
  hipLaunchKernelGGL(deflate_encode_kernel, dim3((unsigned)((n_members + 3) / 4)), dim3(256), 0, s, text, n_text, tok, ntok,
                     codes, crc_tables, slots, sizes);
  hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(1024), 0, s, sizes, (int64_t)n_members, offs);
  hipLaunchKernelGGL(deflate_pack_kernel, dim3((unsigned)n_members), dim3(256), 0, s, slots, sizes, offs, out);
}

}  // namespace bp
