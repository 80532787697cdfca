// pileup_kernels.hip -- gfx950 kernels of the pileup/snp hot path.
//
//   plan_windows_kernel   per window: the range of coordinate-sorted records that can
//                         overlap it (binary search on (ref_id,pos)).
//   pileup_kernel         one workgroup per window: read filters (pileup.go:691, :717-729),
//                         inline 4-bit unpack (replaces biosimd.UnpackSeq, firstread.go:64),
//                         CIGAR walk (pileup.go:275-363), clip (pileup_support.go:51-69),
//                         count rule (pileup.go:223-231), mate-overlap rule (pileup.go:406-446)
//                         into an LDS histogram with ds_add_u32, then one coalesced flush of
//                         Depth + Counts[5][2] per BED position (pileup.go:518-570).
//   column_sums_kernel    checksum of the rows.
//
// Mapping (wave64): a wave takes 64 records at a time; lane l reads record l's 32-byte core
// and applies the read-level filters.  The kept records are then expanded into "units" of 8
// consecutive read bases (one 8-byte qual load + one 4-byte packed-seq load per lane), laid
// out so consecutive lanes hold consecutive units of the same record.  The LDS histogram is
// counter-major (cnt[code][position]) with the position index permuted as
// (p & 7) * W/8 + (p >> 3): lanes of one record that handle the same base-in-unit k hit
// consecutive banks, so a record never conflicts with itself whatever the depth.
//
// Depth is not accumulated separately: every aligned base adds exactly one counter, either
// its Counts[base][strand] slot or slot 10 ("depth only": base failed the quality rule), and
// Depth = sum of the 11 counters at flush.  One LDS atomic per aligned base.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pileup_device.h"

namespace bp {

constexpr int kW = 1024;       // window positions
constexpr int kBlock = 256;    // threads per workgroup
constexpr int kCodes = 11;     // 10 count slots + depth-only

__device__ __forceinline__ int strand_of(int32_t ref, int32_t mate_ref, uint32_t flags) {
  // pileup.GetStrand (pileup/common.go:94-114)
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

// Seq8ToEnumTable (pileup/common.go:66) times 2: 1,2,4,8 -> 0,2,4,6 ; everything else 8 (X).
__device__ __forceinline__ uint32_t code2_of_nibble(uint32_t nib) {
  // two bits per nibble value for the single-bit codes, validity mask 0x0116
  uint32_t idx = (0x00030210u >> (nib * 2)) & 3u;  // nib 1->0, 2->1, 4->2, 8->3
  uint32_t valid = (0x0116u >> nib) & 1u;
  return valid ? idx * 2u : 8u;
}

__device__ __forceinline__ uint64_t sort_key(int32_t ref, int32_t pos) {
  // unmapped (ref -1) sorts last, like biopb.Coord ordering
  return ((uint64_t)(uint32_t)ref << 32) | (uint32_t)pos;
}

__global__ void plan_windows_kernel(KernelParams P, int32_t* win_r0, int32_t* win_r1) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= P.n_windows) return;
  Window win = P.windows[w];
  // first record with (ref,pos) >= (ref, g0 - max_span + 1)
  int64_t lo_pos = (int64_t)win.g0 - P.max_span + 1;
  if (lo_pos < 0) lo_pos = 0;
  uint64_t klo = sort_key(win.ref_id, (int32_t)lo_pos);
  uint64_t khi = sort_key(win.ref_id, win.g1);
  int64_t n = P.n_reads;
  int64_t a = 0, b = n;
  while (a < b) {
    int64_t m = (a + b) >> 1;
    const ReadCore& c = P.cores[m];
    if (sort_key(c.ref_id, c.pos) < klo) a = m + 1; else b = m;
  }
  int64_t r0 = a;
  b = n;
  while (a < b) {
    int64_t m = (a + b) >> 1;
    const ReadCore& c = P.cores[m];
    if (sort_key(c.ref_id, c.pos) < khi) a = m + 1; else b = m;
  }
  win_r0[w] = (int32_t)r0;
  win_r1[w] = (int32_t)a;
}

constexpr int kSub = kW / 8 + 4;     // padded sub-row: 8 residues x kSub
constexpr int kStride = 8 * kSub;    // dwords per counter plane (multiple of 32)
constexpr int kUnroll = 4;           // units in flight per lane

// Counter-major LDS histogram, position index permuted by residue mod 8.
//  * accumulate: the lanes of one record that handle base-in-unit k sit 8 positions apart:
//    same residue, consecutive (p >> 3) -> consecutive banks, whatever their counter plane
//    (kStride % 32 == 0), so a record never conflicts with itself at any depth.
//  * flush: 32 consecutive positions read one plane at banks 4*(p&7) + (p>>3): all distinct.
__device__ __forceinline__ uint32_t lds_index(uint32_t code, uint32_t p) {
  return code * kStride + (p & 7u) * kSub + (p >> 3);
}

// Reference position -> read index of record `c` (mate lookup for the overlap rule).
// Returns -1 when no M op of the record covers ref position `rp`.
__device__ int32_t read_index_at(const ReadCore& c, const uint8_t* blob, int32_t rp) {
  const uint32_t* cg = (const uint32_t*)(blob + (uint64_t)c.blob * 8);
  int32_t r = c.pos, q = 0;
  for (uint32_t k = 0; k < c.n_cigar; k++) {
    uint32_t op = cg[k] & 15u;
    int32_t len = (int32_t)(cg[k] >> 4);
    if (op == 0) {
      if (rp >= r && rp < r + len) return q + (rp - r);
      r += len; q += len;
    } else if (op == 1 || op == 4) {
      q += len;
    } else if (op == 2 || op == 3) {
      r += len;
    } else if (op != 5) {
      break;
    }
    if (r > rp) break;
  }
  return -1;
}

__device__ __forceinline__ uint32_t clipped_qual(uint32_t q, int32_t b, int32_t l_seq, int32_t clip) {
  // clipQuals (pileup_support.go:51-69)
  if (clip > 0 && (2 * clip >= l_seq || b < clip || b >= l_seq - clip)) return 2u;
  return q;
}

typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));

template <bool STITCH>
__global__ __launch_bounds__(kBlock) void pileup_kernel(KernelParams P) {
  __shared__ uint32_t cnt[kCodes * kStride];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr int kWaves = kBlock / 64;

  // XCD-aware item order: workgroups b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of windows so records straddling two windows are re-read from L2.
  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t q8 = nwg >> 3, r8 = nwg & 7u;
  const uint32_t item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;

  const Window win = P.windows[item];
  const int32_t r0 = P.win_r0[item], r1 = P.win_r1[item];
  const int32_t g0 = win.g0, g1 = win.g1;
  const uint32_t wlen = (uint32_t)(g1 - g0);

  for (int i = tid; i < kCodes * kStride / 4; i += kBlock) ((uint4*)cnt)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int32_t clip = P.clip;
  const uint32_t minq = (uint32_t)P.min_base_qual;

  for (int32_t base = r0 + wave * 64; base < r1; base += kWaves * 64) {
    // ---- phase 1: lane = record; read-level filters
    const int32_t r = base + lane;
    uint32_t U = 0, meta = 0, l_seq = 0, blob = 0;
    int32_t pos = 0;
    if (r < r1) {
      const uint4* cp = (const uint4*)(P.cores + r);
      uint4 c0 = cp[0], c1 = cp[1];
      pos = (int32_t)c0.x;
      const int32_t ref_id = (int32_t)c0.y, mate_ref = (int32_t)c0.w;
      const uint32_t flags = c1.x & 0xffffu, mapq = (c1.x >> 16) & 0xffu, aux = c1.x >> 24;
      const uint32_t n_cigar = c1.y & 0xffffu, span = c1.y >> 16;
      l_seq = c1.z;
      blob = c1.w;
      const int strand = strand_of(ref_id, mate_ref, flags);
      // -flag-exclude, -mapq, blank-read filters (pileup.go:691)
      bool keep = ((uint32_t)P.flag_exclude & flags) == 0 && (int32_t)mapq >= P.mapq && n_cigar != 0;
      // strand filters (pileup.go:717-729)
      if (P.strand_req != 0) keep = keep && strand == P.strand_req;
      else keep = keep && (strand != 0 || P.ignore_strand);
      keep = keep && ref_id == win.ref_id && pos < g1 && (int64_t)pos + span > (int64_t)g0;
      const uint32_t sslot = P.ignore_strand ? 0u : (uint32_t)(strand - 1) & 1u;
      // units (8 read bases) that can touch the window
      uint32_t u_lo = 0, u_hi = (l_seq + 7) >> 3;
      if (aux & AUX_SIMPLE) {
        const int64_t b_lo = (int64_t)g0 - pos, b_hi = (int64_t)g1 - pos;
        if (b_lo > 0) u_lo = (uint32_t)(b_lo >> 3);
        if (b_hi < (int64_t)l_seq) u_hi = (uint32_t)((b_hi + 7) >> 3);
      }
      if (keep && u_hi > u_lo) U = u_hi - u_lo;
      // meta: bit0 strand slot, bit1 simple CIGAR, bits 2..15 first unit, bits 16.. n_cigar
      meta = sslot | ((aux & AUX_SIMPLE) ? 2u : 0u) | (u_lo << 2) | (n_cigar << 16);
    }
    uint32_t Umax = U;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      uint32_t v = (uint32_t)__shfl_xor((int)Umax, o);
      Umax = v > Umax ? v : Umax;
    }
    if (Umax == 0) continue;
    const uint32_t magic = (uint32_t)((0x100000000ull + Umax - 1) / Umax);
    const uint32_t total = 64u * Umax;

    // ---- phase 2: lane = unit of 8 bases; kUnroll units in flight per lane
    for (uint32_t t0 = lane; t0 < total; t0 += 64 * kUnroll) {
      uint2 q8v[kUnroll];
      uint32_t s4[kUnroll], meta_g[kUnroll], lseq_g[kUnroll], u_g[kUnroll], gg[kUnroll];
      int32_t pos_g[kUnroll];
      const uint8_t* bptr[kUnroll];
      bool act[kUnroll];
#pragma unroll
      for (int j = 0; j < kUnroll; j++) {
        const uint32_t t = t0 + 64u * j;
        uint32_t g = Umax == 1 ? t : __umulhi(t, magic);
        g = g > 63u ? 63u : g;
        const uint32_t uu = t - g * Umax;
        const uint32_t Ug = (uint32_t)__shfl((int)U, (int)g);
        pos_g[j] = __shfl(pos, (int)g);
        meta_g[j] = (uint32_t)__shfl((int)meta, (int)g);
        lseq_g[j] = (uint32_t)__shfl((int)l_seq, (int)g);
        const uint32_t blob_g = (uint32_t)__shfl((int)blob, (int)g);
        act[j] = t < total && uu < Ug;
        gg[j] = g;
        const uint32_t u = ((meta_g[j] >> 2) & 0x3fffu) + uu;
        u_g[j] = u;
        const uint32_t n_cigar = meta_g[j] >> 16;
        const uint32_t Utot = (lseq_g[j] + 7) >> 3;
        bptr[j] = P.blob + (uint64_t)blob_g * 8;
        q8v[j] = make_uint2(0, 0);
        s4[j] = 0;
        if (act[j]) {
          const uint32_t cig_bytes = ((n_cigar + 1) & ~1u) * 4;
          q8v[j] = *(const uint2*)(bptr[j] + cig_bytes + 8 * u);
          s4[j] = *(const uint32_t*)(bptr[j] + cig_bytes + 8 * Utot + 4 * u);
        }
      }
#pragma unroll
      for (int j = 0; j < kUnroll; j++) {
        if (!act[j]) continue;
        const uint32_t n_cigar = meta_g[j] >> 16;
        const int32_t b0 = (int32_t)(u_g[j] * 8);
        const int32_t lseq = (int32_t)lseq_g[j];
        // reference position of each of the 8 bases, relative to g0 (or a huge value)
        uint32_t prel[8];
        if (meta_g[j] & 2u) {
          const uint32_t p0 = (uint32_t)(pos_g[j] + b0 - g0);
#pragma unroll
          for (int k = 0; k < 8; k++) prel[k] = p0 + k;
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) prel[k] = 0xffffffffu;
          const uint32_t* cg = (const uint32_t*)bptr[j];
          int32_t rr = pos_g[j], qq = 0;
          for (uint32_t c = 0; c < n_cigar; c++) {
            const uint32_t op = cg[c] & 15u;
            const int32_t len = (int32_t)(cg[c] >> 4);
            if (op == 0) {
#pragma unroll
              for (int k = 0; k < 8; k++) {
                const int32_t b = b0 + k;
                if (b >= qq && b < qq + len) prel[k] = (uint32_t)(rr + (b - qq) - g0);
              }
              rr += len; qq += len;
            } else if (op == 1 || op == 4) {
              qq += len;
            } else if (op == 2 || op == 3) {
              rr += len;
            } else if (op != 5) {
              break;  // host rejected reachable bad ops (pileup.go:355-359)
            }
            if (qq >= b0 + 8) break;
          }
        }

        uint32_t sslot = meta_g[j] & 1u;
        int32_t mate_idx = -1;
        bool second = false;
        ReadCore mc;
        if constexpr (STITCH) {
          const int32_t m = P.mate[base + (int32_t)gg[j]];
          if (m >= 0) {
            mate_idx = m >> 1;
            second = (m & 1) != 0;
            mc = P.cores[mate_idx];
            // the pair is counted on the strand of the read that found its mate
            // (pileup.go:799-802); for the stored read that is the mate's strand
            if (!second && !P.ignore_strand)
              sslot = (uint32_t)(strand_of(mc.ref_id, mc.mate_ref_id, mc.flags) - 1) & 1u;
          }
        }
        const int32_t nb = lseq - b0;  // valid bases in this unit (may exceed 8)

#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t p = prel[k];
          if (k >= nb || p >= wlen) continue;
          const uint32_t nsh = ((k & 1) ? (s4[j] << 2 >> (8 * (k >> 1))) : (s4[j] >> (8 * (k >> 1) + 2))) & 0x3cu;
          // Seq8ToEnumTable (pileup/common.go:66) x2: nibble 1,2,4,8 -> 0,2,4,6; others 8 (X)
          const uint32_t c2 = (uint32_t)(0x8888888688848208ull >> nsh) & 15u;
          uint32_t q = ((k < 4 ? q8v[j].x : q8v[j].y) >> (8 * (k & 3))) & 0xffu;
          q = clipped_qual(q, b0 + k, lseq, clip);
          uint32_t code;
          if constexpr (STITCH) {
            int32_t mi = -1;
            const int32_t rp = (int32_t)p + g0;
            if (mate_idx >= 0) mi = read_index_at(mc, P.blob, rp);
            if (mi >= 0) {
              // both reads of the pair cover this position (pileup.go:409-430): Depth++ once,
              // handled by the read that found its mate
              if (!second) continue;
              const uint8_t* mb = P.blob + (uint64_t)mc.blob * 8;
              const uint32_t mcig = ((mc.n_cigar + 1u) & ~1u) * 4;
              const uint32_t mU = (mc.l_seq + 7) >> 3;
              uint32_t mq = mb[mcig + mi];
              const uint32_t mbyte = mb[mcig + 8 * mU + (mi >> 1)];
              const uint32_t mnib = (mi & 1) ? (mbyte & 15u) : (mbyte >> 4);
              mq = clipped_qual(mq, mi, (int32_t)mc.l_seq, clip);
              if (mnib * 4 == nsh) {
                if (q >= 96u || mq >= 96u) {
                  // qualPassTable index out of range: a Go runtime panic in the reference
                  atomicMin(&P.err->read, base + (int32_t)gg[j]);
                  continue;
                }
                const bool pass = mq >= (uint32_t)P.qthr[q] || c2 == 8u;
                code = pass ? c2 + sslot : 10u;
              } else {
                code = 8u + sslot;
              }
            } else {
              const bool pass = q >= minq || c2 == 8u;
              code = pass ? c2 + sslot : 10u;
            }
          } else {
            // addBase (pileup.go:223-231): N/ambiguous always counted
            const bool pass = q >= minq || c2 == 8u;
            code = pass ? c2 + sslot : 10u;
          }
          atomicAdd(&cnt[lds_index(code, p)], 1u);
        }
      }
    }
  }
  __syncthreads();

  // ---- flush: one lane per position; Depth = sum of the 11 counters (pileup.go:518-570)
  for (uint32_t p = tid; p < wlen; p += kBlock) {
    uint32_t v[kCodes];
    uint32_t d = 0;
#pragma unroll
    for (int c = 0; c < kCodes; c++) {
      v[c] = cnt[lds_index(c, p)];
      d += v[c];
    }
    int64_t R;
    if (win.full) {
      R = win.row_base + p;
    } else {
      const int32_t gp = g0 + (int32_t)p;
      uint32_t a = win.iv_lo, b = win.iv_hi;  // last interval with start <= gp
      while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (P.iv_start[m] <= gp) a = m; else b = m;
      }
      const int32_t is = P.iv_start[a];
      if (gp < is || gp >= P.iv_end[a]) continue;  // not a BED position
      R = P.iv_row[a] + (gp - is);
    }
    uint32_t* out = P.rows + R * 11;
    u32x4a o0 = {d, v[0], v[1], v[2]};
    u32x4a o1 = {v[3], v[4], v[5], v[6]};
    u32x3a o2 = {v[7], v[8], v[9]};
    *(u32x4a*)(out) = o0;
    *(u32x4a*)(out + 4) = o1;
    *(u32x3a*)(out + 8) = o2;
  }
}

__global__ void column_sums_kernel(const uint32_t* rows, int64_t n_rows, unsigned long long* sums) {
  __shared__ unsigned long long s[11];
  if (threadIdx.x < 11) s[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long acc[11];
#pragma unroll
  for (int c = 0; c < 11; c++) acc[c] = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows;
       r += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 11; c++) acc[c] += rows[r * 11 + c];
  }
#pragma unroll
  for (int c = 0; c < 11; c++) {
    unsigned long long v = acc[c];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s[c], v);
  }
  __syncthreads();
  if (threadIdx.x < 11) atomicAdd(&sums[threadIdx.x], s[threadIdx.x]);
}

// ---- launch wrappers (called from pileup_api.hip)
void launch_plan(const KernelParams& P, int32_t* r0, int32_t* r1, hipStream_t s) {
  if (P.n_windows == 0) return;
  int threads = 256;
  int blocks = (P.n_windows + threads - 1) / threads;
  hipLaunchKernelGGL(plan_windows_kernel, dim3(blocks), dim3(threads), 0, s, P, r0, r1);
}

void launch_pileup(const KernelParams& P, hipStream_t s) {
  if (P.n_windows == 0) return;
  if (P.stitch) hipLaunchKernelGGL(pileup_kernel<true>, dim3(P.n_windows), dim3(kBlock), 0, s, P);
  else hipLaunchKernelGGL(pileup_kernel<false>, dim3(P.n_windows), dim3(kBlock), 0, s, P);
}

void launch_column_sums(const uint32_t* rows, int64_t n_rows, unsigned long long* sums, hipStream_t s) {
  if (n_rows == 0) return;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(column_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, n_rows, sums);
}

int window_size() { return kW; }

}  // namespace bp
