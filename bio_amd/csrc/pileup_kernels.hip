// pileup_kernels.hip -- gfx950 kernels of the pileup/snp hot path.
//
//   plan_windows_kernel   per window: the range of coordinate-sorted records that can
//                         overlap it (binary search on (ref_id,pos)).
//   pileup_kernel         per window: read filters (pileup.go:691, :717-729),
//                         inline 4-bit unpack (replaces biosimd.UnpackSeq, firstread.go:64),
//                         CIGAR walk (pileup.go:275-363), clip (pileup_support.go:51-69),
//                         count rule (pileup.go:223-231), mate-overlap rule (pileup.go:406-446)
//                         into an LDS histogram with ds_add_u32, then one coalesced flush of
//                         Depth + Counts[5][2] per BED position (pileup.go:518-570).
//   column_sums_kernel    checksum of the rows.
//
// Mapping (wave64, 512-thread workgroups).  A workgroup owns up to kWinPerBlockMax consecutive
// windows of kW reference positions and walks them one after the other:
//   phase 1  thread = record.  Read-level filters, then the CIGAR walk turns every M op into
//            a segment clipped to the window (two per lane per step); a wave hands out
//            segment and unit slots with one DPP prefix sum and ONE packed 64-bit LDS atomic,
//            and leaves a 16-byte record per segment with everything its units need.
//   phase 2  thread = unit = the 32 window positions [32j, 32j+32) of one segment: a 20-byte
//            load of packed bases (plus 36 bytes of quals when a quality rule is active),
//            realigned in registers; the 8 plane codes of 8 bases are computed together
//            (SWAR on nibbles), then 32 ds_add_u32 whose addresses differ only by an
//            immediate.  Units are dealt to lanes so that the lanes of one LDS instruction
//            come from all over the window: reads that overlap on the genome would
//            otherwise serialise on the same counters.
//   flush    thread = position (upper half of the waves, while the lower half already runs
//            phase 1 of the next window): Depth = sum of the 12 planes, row written with 3
//            wide stores, planes zeroed on the way out.
// The LDS histogram is counter-major (cnt[plane][position]) with position p = 32j + r stored
// at r*kSub + j (kSub = kW/32 + 1): consecutive lanes hold consecutive units j of the same
// segment, so the adds of a wave instruction hit consecutive banks, and so do the 32
// positions a flush instruction reads.
//
// Depth is not accumulated separately: every aligned base adds exactly one counter, either
// its Counts[base][strand] slot or a "depth only" slot (planes 10/11: base failed the quality
// rule), and Depth = sum of the 12 planes at flush.  One LDS atomic per aligned base.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "pileup_device.h"

namespace bp {

// Diagnostic build only (-DBP_STAMPS): per-window phase time stamps (s_memtime) written to
// P.dbg; never compiled into the shipped library.
#ifdef BP_STAMPS
#define BP_STAMP(i)                                                                      \
  do {                                                                                   \
    if (!DEEP && P.dbg && (threadIdx.x & 63) == 0)                                       \
      P.dbg[((uint64_t)item * (kBlock / 64) + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define BP_STAMP(i) do {} while (0)
#endif

// Tunables (overridable with -D for experiments; the defaults are what ships).
#ifndef BP_W
#define BP_W 1024
#endif
#ifndef BP_BLOCK
#define BP_BLOCK 512
#endif
#ifndef BP_SEGCAP
#define BP_SEGCAP 512
#endif
#ifndef BP_UNITCAP
#define BP_UNITCAP 2048
#endif
#ifndef BP_DEPTH
#define BP_DEPTH 3
#endif
#ifndef BP_DEPTHS
#define BP_DEPTHS 1   // ... in the stitched variants (twice the loads per unit)
#endif
#ifndef BP_DEPTHQ
#define BP_DEPTHQ 2
#endif
#ifndef BP_WINS
#define BP_WINS 24   // at most; fewer when the job has too few windows to fill the device that way
#endif
#ifndef BP_FLUSH_UPPER
#define BP_FLUSH_UPPER 8   // eighths of a window's positions flushed by the upper half of the waves
#endif
#ifndef BP_CHUNK
#define BP_CHUNK 4      // consecutive unit slots dealt to consecutive lanes
#endif
#ifndef BP_PERMUTE
#define BP_PERMUTE 1
#endif
#ifndef BP_MINWAVES
#define BP_MINWAVES 4
#endif
constexpr int kW = BP_W;          // window positions
constexpr int kBlock = BP_BLOCK;  // threads per workgroup
constexpr int kPlanes = 12;    // 10 count planes + 2 depth-only planes

__device__ __forceinline__ int strand_of(bool mate_same_ref, uint32_t flags) {
  // pileup.GetStrand (pileup/common.go:94-114)
  if (!mate_same_ref) return 0;
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

// First record index in [lo, hi) whose pos is >= x (records of one reference are sorted by
// pos).  The coarse index (every kCoarse-th record, a few hundred KB, cache-resident) narrows
// the range to kCoarse records before the 32-byte cores are touched.
__device__ int64_t first_at_or_after(const KernelParams& P, int64_t lo, int64_t hi, int64_t x) {
  int64_t cl = (lo + kCoarse - 1) / kCoarse, ch = (hi + kCoarse - 1) / kCoarse;
  const int64_t c0 = cl;
  while (cl < ch) {
    const int64_t m = (cl + ch) >> 1;
    if ((int64_t)P.coarse_pos[m] < x) cl = m + 1; else ch = m;
  }
  // record cl*kCoarse is the first sampled one at or after x (or there is none)
  int64_t a = cl > c0 ? (cl - 1) * kCoarse + 1 : lo;
  int64_t b = cl * kCoarse < hi ? cl * kCoarse : hi;
  if (a > b) a = b;
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if ((int64_t)P.cores[m].pos < x) a = m + 1; else b = m;
  }
  return a;
}

__global__ void plan_windows_kernel(KernelParams P, int32_t* win_r0, int32_t* win_r1) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= P.n_windows) return;
  const Window win = P.windows[w];
  // records of the window's reference are the index range [lo, hi), sorted by pos
  const int64_t lo = P.ref_first[win.ref_id], hi = P.ref_first[win.ref_id + 1];
  // first record that can reach g0: pos >= g0 - max_span + 1
  int64_t lo_pos = (int64_t)win.g0 - P.max_span + 1;
  if (lo_pos < 0) lo_pos = 0;
  const int64_t r0 = first_at_or_after(P, lo, hi, lo_pos);
  const int64_t a = first_at_or_after(P, r0, hi, win.g1);
  win_r0[w] = (int32_t)r0;
  int64_t r1 = a;
  // a deep window is cut into even slices of at most slice_recs records, all of them extra
  // work items for the second launch; the first launch only writes its rows (zeros)
  const int64_t n = r1 - r0, sl = P.slice_recs;
  if (P.extra_items && n > sl) {
    const uint32_t cnt = (uint32_t)((n + sl - 1) / sl);
    const int64_t each = (n + cnt - 1) / cnt;
    const uint32_t base = atomicAdd(P.n_extra, cnt);
    for (uint32_t k = 0; k < cnt; k++) {
      if (base + k >= (uint32_t)P.extra_cap) {
        P.err->kind = ERR_ITEM_OVERFLOW;
        break;
      }
      const int64_t s0 = r0 + each * (int64_t)k;
      const int64_t s1 = s0 + each < r1 ? s0 + each : r1;
      P.extra_items[base + k] = make_uint4((uint32_t)w, (uint32_t)s0, (uint32_t)s1, 0u);
    }
    r1 = r0;
  }
  win_r1[w] = (int32_t)r1;
}

constexpr int kUnitPos = 32;               // window positions per unit
constexpr int kSub = kW / kUnitPos + 1;    // padded sub-row: 32 residues x kSub
constexpr int kStride = kUnitPos * kSub;   // dwords per counter plane (multiple of 32)
constexpr int kSegCap = BP_SEGCAP;   // M-segments per round
constexpr int kUnitCap = BP_UNITCAP; // units (32 window positions of one segment) per round
constexpr int kDepth = BP_DEPTH;     // units in flight per lane (software pipeline depth)
constexpr int kDepthQ = BP_DEPTHQ;   // ... in the stitched variants (16 bytes of bits per unit)
constexpr int kWinPerBlockMax = BP_WINS; // consecutive windows per workgroup (KernelParams::wins_per_block)

// LDS histogram: 12 counter planes (A+ A- C+ C- G+ G- T+ T- X+ X- lowq+ lowq-), each plane
// laid out by residue: position p = 32j + r lives at r*kSub + j.
//  * accumulate: a lane handles the 32 positions of unit j of one M-segment, so its 32
//    ds_add_u32 differ only in the instruction's immediate offset (r*kSub*4); consecutive
//    lanes hold consecutive j of the same segment -> consecutive banks, and kStride % 32 == 0
//    keeps that true across planes: a record never conflicts with itself at any depth.
//    Reads that overlap on the genome would hit the same counters in the same instruction;
//    the unit slots are handed to lanes in an order that keeps the 8 groups of 8 lanes of a
//    wave an eighth of the window apart (see issue_unit).
//  * flush: 32 consecutive positions of one plane sit at banks r + j: all distinct.
struct SegRec {
  int32_t rec[kSegCap];  // stitch: record index of the segment (to name the record of a data error)
};
struct NoRec {};

// One M-segment of the current round, with everything a unit needs precomputed by the lane
// that walked the CIGAR:
// (a unit = the kUnitPos = 32 window positions [32j, 32j+32) of one segment)
//   x  low 32 bits of the byte address of the packed bases of the segment's first unit
//      (4-byte aligned); unit u reads 20 bytes at x + 16u
//   y  0..2 address bits 32..34 | 3..5 nibble shift | 6 strand slot | 8..12 first valid
//      position of the first unit | 13..17 last valid position of the last unit |
//      24..29 number of units - 1
//   z  0..14 first unit slot | 16..21 first window unit | 24..28 bit shift of the unit's pass bits
//   w  (address of the dword with the first unit's first pass bit) - (base address above); unit u
//      reads 8 bytes at x + w + 4u (stitch: 16 bytes - pass, drop, pass, drop - at x + w + 8u)
template <bool STITCH>
struct __align__(16) BlockSharedT {
  uint32_t cnt[kPlanes * kStride];
  uint4 seg[kSegCap];
  typename std::conditional<STITCH, SegRec, NoRec>::type ex;
  uint16_t owner[kUnitCap];     // unit slot -> segment
  unsigned long long n_alloc[2];  // segments << 32 | units handed out this round (by round parity)
};

__device__ __forceinline__ uint32_t clipped_qual(uint32_t q, int32_t b, int32_t l_seq, int32_t clip) {
  // clipQuals (pileup_support.go:51-69)
  if (2 * clip >= l_seq || b < clip || b >= l_seq - clip) return 2u;
  return q;
}

typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));

// Inclusive prefix sum over the wave with DPP row shifts and row broadcasts (6 VALU steps, no
// LDS traffic).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

// One M-segment of a record, relative to the window.
struct SegIn {
  int32_t sref;   // first window position (may be < 0)
  int32_t len;
  int32_t q0;     // read index of its first base
};

// Window positions [s1, e1) and unit range of a segment; false when nothing of it is inside.
__device__ __forceinline__ bool seg_extent(const SegIn& g, uint32_t wlen, int32_t& s1, int32_t& e1,
                                           uint32_t& jlo, uint32_t& U) {
  s1 = g.sref > 0 ? g.sref : 0;
  e1 = (int64_t)g.sref + g.len < (int64_t)wlen ? g.sref + g.len : (int32_t)wlen;
  if (e1 <= s1) { jlo = 0; U = 0; return false; }
  jlo = (uint32_t)s1 >> 5;
  U = ((uint32_t)(e1 - 1) >> 5) - jlo + 1;
  return true;
}

template <int QM, bool STITCH>
__device__ __forceinline__ void write_seg(BlockSharedT<STITCH>& S, uint32_t idx, uint32_t us, const SegIn& g,
                                          int32_t s1, int32_t e1, uint32_t jlo, uint32_t U, uint32_t blob,
                                          uint32_t lseqnc, uint32_t sslot, [[maybe_unused]] int32_t rec) {
  // byte offsets from the record's base part [cigar padded to even][seq 4*Ur][bits]
  const uint32_t nc = lseqnc >> 16;
  const int32_t cigb = (int32_t)(((nc + 1) & ~1u) * 4);
  const int32_t d = g.q0 - g.sref;                       // read index of window position 0
  const int32_t b_first = d + (int32_t)(jlo * 32);       // read index of the first unit's base 0 (>= -31)
  const int32_t so = cigb + (b_first >> 1);              // >= -16: the payload buffer starts with a pad
  const uint64_t sa = (uint64_t)((int64_t)((uint64_t)blob * 8) + (int64_t)(so & ~3));  // so may be < 0
  const uint32_t noff = 2u * ((uint32_t)so & 3u) + ((uint32_t)d & 1u);
  uint4 r;
  r.x = (uint32_t)sa;
  r.y = (uint32_t)(sa >> 32) | (noff << 3) | (sslot << 6) | (((uint32_t)s1 & 31u) << 8) |
        (((uint32_t)(e1 - 1) & 31u) << 13) | ((U - 1) << 24);
  // per-base bits: bit b_first of the record's bit string, which lies behind its packed bases (4
  // bytes per 8-base unit of the whole read); one dword per 32 bases, two (pass, drop) when stitching
  r.z = us | (jlo << 16) | (((uint32_t)b_first & 31u) << 24);
  const int32_t bits0 = cigb + 4 * (int32_t)(((lseqnc & 0xffffu) + 7u) >> 3);
  r.w = QM == 0 ? 0u : (uint32_t)(bits0 + (b_first >> 5) * (QM == 3 ? 8 : 4) - (so & ~3));
  S.seg[idx] = r;
  if constexpr (STITCH) S.ex.rec[idx] = rec;
  for (uint32_t i = 0; i < U; i++) S.owner[us + i] = (uint16_t)idx;
}

// Admit up to two M-segments per lane (A, then B) into the round's list; called by every lane
// of the wave.  Slots are handed out with a wave prefix sum and ONE LDS atomic per wave on a
// packed (segments << 32 | units) counter, so unit slots grow with the segment index.
// Returns false for a lane whose segments did not fit this round (it got neither): the
// caller keeps its cursor and retries next round.
template <int QM, bool STITCH>
__device__ __forceinline__ bool emit_wave(BlockSharedT<STITCH>& S, bool wantA, const SegIn& A, bool wantB,
                                          const SegIn& B, uint32_t wlen, uint32_t blob, uint32_t lseqnc,
                                          uint32_t sslot, int32_t rec, uint32_t par) {
  int32_t s1A, e1A, s1B, e1B;
  uint32_t jloA, UA, jloB, UB;
  const bool realA = seg_extent(A, wlen, s1A, e1A, jloA, UA) && wantA;  // clipped-away segments need no slot
  const bool realB = seg_extent(B, wlen, s1B, e1B, jloB, UB) && wantB;
  if (!realA) UA = 0;
  if (!realB) UB = 0;
  const uint32_t nseg = (realA ? 1u : 0u) + (realB ? 1u : 0u), U = UA + UB;
  // inclusive prefix sums of (segments, units) over the wave, packed as segments << 16 | units
  // (a wave adds at most 128 segments and 128 x (kW/32 + 1) units: no carry between the fields)
  const uint32_t mine = (nseg << 16) | U;
  const uint32_t incl = wave_inclusive_scan(mine);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  unsigned long long base = 0;
  if ((threadIdx.x & 63) == 0 && total != 0)
    base = atomicAdd(&S.n_alloc[par], ((unsigned long long)(total >> 16) << 32) | (total & 0xffffu));
  const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  const uint32_t base_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
  const uint32_t excl = incl - mine;
  // (units handed out in a round stay below 2^16... 512 lanes x 2 x 33; the counter's low word
  // cannot carry into the segments)
  const uint32_t idx = base_s + (excl >> 16), us = base_lo + (excl & 0xffffu);
  const bool ok = idx + nseg <= (uint32_t)kSegCap && us + U <= (uint32_t)kUnitCap;
  if (ok) {
    if (realA) write_seg<QM, STITCH>(S, idx, us, A, s1A, e1A, jloA, UA, blob, lseqnc, sslot, rec);
    if (realB) write_seg<QM, STITCH>(S, idx + (realA ? 1u : 0u), us + UA, B, s1B, e1B, jloB, UB, blob, lseqnc, sslot, rec);
  } else {
    for (uint32_t i = us; i < (uint32_t)kUnitCap && i - us < U; i++) S.owner[i] = 0xffffu;
  }
  return ok;
}

// One unit in flight: the loaded bytes and everything processing needs, so that processing does
// not touch the segment list again.
// QM: what the unit brings beside its bases.  0 nothing (min_base_qual 0 without stitching: every
// base passes), 2 the 32 pass bits the packer left behind the bases (KernelParams.qpass; quality
// rule and clip applied there), 3 (stitch) pass and drop bits, the outcome of the pair rule
// worked out once per job by stitch_resolve_kernel.  (1 was the 32 quals themselves, 36 bytes: the
// kernel compared them with min_base_qual and worked out the clip masks itself until the packers
// took that over; 5.73 -> 4.67 ms on chr1 30x -min-base-qual 20.)
template <int QM>
struct UnitLoad {
  uint32_t sd[5];                  // 20 bytes covering the unit's 32 packed bases
  uint32_t qd[QM == 3 ? 4 : QM == 2 ? 2 : 1];  // the dwords covering the unit's pass (and drop) bits
  uint32_t mask;                   // valid positions of the unit
  uint32_t meta;                   // 0..5 window unit | 6..8 nibble shift | 11 strand slot | 16 active |
                                   // 17..21 bit shift of the pass bits
  uint32_t seg;                    // segment (stitch: to name the record of a data error)
};

// Loads are issued unconditionally (inactive lanes read the start of the payload buffer) so
// that every issue is the same number of VMEM instructions and the counted vmcnt waits stay
// exact.  QM = 0 (min_base_qual 0, no stitching: every base passes the quality rule,
// pileup.go:227) skips the qual loads altogether.
// Slot order: neighbouring segments overlap on the genome, and lanes of one instruction
// that add to the same counter serialise.  Chunks of BP_CHUNK consecutive slots (part of
// one segment: distinct units, consecutive banks) are dealt out column-wise over a
// (64 / BP_CHUNK) x R grid, so the chunks a wave holds are R chunks apart in slot order:
// spread evenly over the round's units, i.e. over the window.  t = thread + iteration x block
// -> unit slot (T: none).
__device__ __forceinline__ uint32_t slot_of(uint32_t t, uint32_t T, uint32_t R) {
  // (the LDS works on half-waves: lanes 0-31 take the even columns and lanes 32-63 the odd
  // ones, so that each half spans the whole window)
  constexpr uint32_t kCh = BP_CHUNK, kCols = 64 / kCh;
#if BP_PERMUTE
  const uint32_t a = t >> 6, ci = (t & 63u) / kCh;
  const uint32_t b = ((ci % (kCols / 2)) << 1) | (ci / (kCols / 2));
  t = a < R ? (((b * R + a) * kCh) | (t & (kCh - 1))) : T;  // rows past R hold no slots
#endif
  return t < T ? t : T;
}

template <int QM, bool STITCH>
__device__ __forceinline__ void issue_unit(const KernelParams& P, const BlockSharedT<STITCH>& S, uint32_t t,
                                           uint32_t T, UnitLoad<QM>& L) {
  bool in = t < T;
  uint32_t s = S.owner[in ? t : 0];
  if (s == 0xffffu) { in = false; s = 0; }
  const uint4 r = S.seg[s];
  const uint32_t u = t - (r.z & 0x7fffu);
  const uint32_t jj = ((r.z >> 16) & 0x3fu) + u;
  uint32_t mask = (u == 0 ? 0xffffffffu << ((r.y >> 8) & 31u) : 0xffffffffu) &
                  (u == ((r.y >> 24) & 0x3fu) ? 0xffffffffu >> (31u - ((r.y >> 13) & 31u)) : 0xffffffffu);
  L.mask = in ? mask : 0u;
  L.seg = s;
  L.meta = (in ? jj | 0x10000u : 0u) | (((r.y >> 3) & 7u) << 6) | (((r.z >> 24) & 31u) << 17) | (((r.y >> 6) & 1u) << 11);
  uint64_t off = (((uint64_t)(r.y & 7u) << 32) | r.x) + 16u * u;
  if (!in) off = 0;
  const uint8_t* sp = P.blob + off;
  if constexpr (QM == 2) {
    const uint8_t* mp = sp + (in ? (int64_t)(int32_t)r.w - 12 * (int64_t)u : 0);  // (sp moves 16 bytes a unit, the bits 4)
    const u32x2a m0 = *(const u32x2a*)mp;
    L.qd[0] = m0.x;
    L.qd[1] = m0.y;
  }
  if constexpr (QM == 3) {
    const uint8_t* mp = sp + (in ? (int64_t)(int32_t)r.w - 8 * (int64_t)u : 0);  // (... pass + drop bits 8)
    const u32x4a m0 = *(const u32x4a*)mp;
    L.qd[0] = m0.x; L.qd[1] = m0.y; L.qd[2] = m0.z; L.qd[3] = m0.w;
  }
  const u32x4a s0 = *(const u32x4a*)sp;
  L.sd[4] = *(const uint32_t*)(sp + 16);
  L.sd[0] = s0.x; L.sd[1] = s0.y; L.sd[2] = s0.z; L.sd[3] = s0.w;
}

// The unit's 32 positions are handled as 4 groups of 8; the 8
// plane codes of a group are computed together on the nibble-packed bases (SWAR), so the
// per-base cost is two extractions, one multiply-add and the LDS atomic.
//   code c: A 0, C 1, G 2, T 3, X 4, depth-only 5  ->  plane 2c + strand slot
// Seq8ToEnumTable (pileup/common.go:66): a nibble with exactly one bit set is a base
// (log2 = its code); every other value is X.  A base failing the quality rule goes to the
// depth-only planes, X is counted regardless (addBase, pileup.go:223-231).
//
// Stitched runs take the same path.  What addReadPair (pileup.go:406-446) decides where both
// reads of a pair cover a position was written into the payloads by stitch_resolve_kernel: the
// read stored first has its drop bits set there (the position is counted once, with the other
// read), the read that found its mate carries the outcome in its bases (X where the reads
// disagree) and pass bits (the qualSumTable rule where they agree).  Pass and drop both set marks a
// quality >= 96 at an agreeing position, where the reference's table lookup panics: reported with
// the record's index when a window gets there, as the reference would fail there.
template <int QM, class SH>
__device__ __forceinline__ void process_unit_plain(const KernelParams& P, SH& B, const UnitLoad<QM>& L, uint32_t cpo) {
  constexpr uint32_t M1 = 0x11111111u;
  const uint32_t meta = L.meta;
  const uint32_t jj = meta & 0x3fu;
  uint32_t vmask = L.mask;  // positions still to be counted
  [[maybe_unused]] uint32_t pass32 = 0xffffffffu;
  if constexpr (QM == 2) pass32 = __builtin_amdgcn_alignbit(L.qd[1], L.qd[0], (meta >> 17) & 31u);
  if constexpr (QM == 3) {
    pass32 = __builtin_amdgcn_alignbit(L.qd[2], L.qd[0], (meta >> 17) & 31u);
    const uint32_t drop32 = __builtin_amdgcn_alignbit(L.qd[3], L.qd[1], (meta >> 17) & 31u);
    if (pass32 & drop32 & vmask) atomicMin(&P.err->read, B.ex.rec[L.seg]);
    vmask &= ~drop32;
  }
  const uint32_t* const x = L.sd;  // base k of a dword at bits 4k (the packer stores it that way)
  const uint32_t nsh = (meta >> 4) & 0x1cu;  // 4 * nibble shift
  const uint32_t cbase = (jj + cpo + ((meta >> 11) & 1u) * (uint32_t)kStride) * 4u;
  char* const lds = (char*)B.cnt;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const uint32_t sw = __builtin_amdgcn_alignbit(x[m + 1], x[m], nsh);
    // log2 of a one-hot nibble n is (n >> 1) - (n >> 3): 1,2,4,8 -> 0,1,2,3
    const uint32_t t1 = sw >> 1;
    const uint32_t lg = (t1 & 0x77777777u) - ((sw >> 3) & M1);
    const uint32_t xx = sw ^ t1, a = sw & t1;                         // pair parity / pair full
    const uint32_t oh = (xx ^ (xx >> 2)) & ~(a | (a >> 2)) & M1;      // exactly one bit set
    const uint32_t oh3 = oh | (oh << 1);
    uint32_t cw = (lg & oh3) + ((oh ^ M1) << 2);                      // one-hot ? log2 : 4 (X)
    if constexpr (QM != 0) {
      // the pass bits (quality rule, clip and pair rule already applied): 8 bits -> bit 4k
      uint32_t pn = (pass32 >> (8 * m)) & 0xffu;
      pn = (pn | (pn << 12)) & 0x000f000fu;
      pn = (pn | (pn << 6)) & 0x03030303u;
      pn = (pn | (pn << 3)) & M1;
      const uint32_t pn15 = (pn << 4) - pn;     // 0xf per passing base
      const uint32_t cf = 0x44444444u | oh;     // failing: base -> depth-only (5), X stays X (4)
      cw = (cw & pn15) | (cf & ~pn15);
    }
    // lanes without a unit (or without a valid base in this group) sit the atomics out:
    // they would all add 0 to one address and serialise the whole instruction
    if ((vmask >> (8 * m)) & 0xffu) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t c = (cw >> (4 * k)) & 7u;
        const uint32_t val = (vmask >> (8 * m + k)) & 1u;
#ifdef BP_X_NOATOMIC
        if (val > 1u)
#endif
        atomicAdd((uint32_t*)(lds + c * (uint32_t)(2 * kStride * 4) + cbase + (8 * m + k) * kSub * 4), val);
      }
    }
  }
}

// Advance the CIGAR cursor (c, r, q) = (op index, window position, read index) to the next M
// op that starts inside the window and step past it; false when there is none.
__device__ __forceinline__ bool next_m_op(const uint32_t* cg, uint32_t cg0, uint32_t cg1, uint32_t cg2,
                                          uint32_t n_cigar, uint32_t wlen, uint32_t& c, int32_t& r,
                                          int32_t& q, int32_t& sref, int32_t& slen, int32_t& sq0) {
  while (c < n_cigar) {
    // the first three ops travel in the core; longer CIGARs continue in the payload
    const uint32_t cop = c == 0 ? cg0 : c == 1 ? cg1 : c == 2 ? cg2 : cg[c];
    const uint32_t op = cop & 15u;
    const int32_t len = (int32_t)(cop >> 4);
    c++;
    if (op == 0) {
      sref = r;
      slen = len;
      sq0 = q;
      r += len;
      q += len;
      if (r >= (int32_t)wlen) c = n_cigar;
      return true;
    }
    // I, S consume the read; D, N the reference; H nothing; anything else ends the walk (the
    // host rejected reachable bad ops, pileup.go:355-359).  Written as selects: with
    // branches the compiler spills r and q to a scratch array indexed by the op class.
    q += (op == 1 || op == 4) ? len : 0;
    r += (op == 2 || op == 3) ? len : 0;
    if (op > 5 || r >= (int32_t)wlen) c = n_cigar;
  }
  return false;
}

// Block-uniform values that come from global memory arrive through vector loads (the kernel also
// stores to global memory, so the compiler will not use the scalar cache for them) and would sit
// in a VGPR of every lane for the whole window: move them to SGPRs - when the window starts, not
// where they are fetched (a readfirstlane has to wait for its load).
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Window uni_window(const Window& v) {
  Window w;
  w.ref_id = uni(v.ref_id);
  w.g0 = uni(v.g0);
  w.g1 = uni(v.g1);
  w.iv_lo = (uint32_t)uni((int32_t)v.iv_lo);
  w.iv_hi = (uint32_t)uni((int32_t)v.iv_hi);
  w.full = uni(v.full);
  w.n_rows = uni(v.n_rows);
  w.pad = 0;
  w.row_base = (int64_t)(((uint64_t)(uint32_t)uni((int32_t)((uint64_t)v.row_base >> 32)) << 32) |
                         (uint32_t)uni((int32_t)(uint32_t)v.row_base));
  return w;
}

// Workgroup = KernelParams::wins_per_block consecutive windows, processed one after the other:
//   phase 1  thread = record: read filters, CIGAR -> M-segments clipped to the window
//   phase 2  thread = unit:   counts into the LDS planes
//   flush    thread = position: Depth + Counts row to HBM, planes zeroed on the way out
// Two barriers per window (after phase 1, after phase 2).  The record cores of window i+1
// are loaded while window i is in phase 2, so the only exposed memory latency per window is
// that of the unit loads.
// DEEP = true is the second launch over the extra slices of deep windows: one work item per
// loop iteration, taken from P.extra_items, flushed with atomic adds on top of the rows the
// first launch wrote.
template <bool STITCH, bool MINQ0, bool DEEP>
__global__ __launch_bounds__(kBlock, BP_MINWAVES) void pileup_kernel(KernelParams P) {
  // what a unit brings beside its bases: nothing (every base passes), the packer's pass bits
  // (KernelParams.qpass), or pass and drop bits (stitch: see stitch_resolve_kernel)
  constexpr int QM = STITCH ? 3 : MINQ0 ? 0 : 2;
  __shared__ BlockSharedT<STITCH> S;
  const int tid = threadIdx.x;
  uint32_t n_items = 0;
  if constexpr (DEEP) {
    n_items = *P.n_extra;
    if (n_items > (uint32_t)P.extra_cap) n_items = (uint32_t)P.extra_cap;
    if (blockIdx.x >= n_items) return;
  }

  // XCD-aware order: workgroups b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of windows so records straddling two workgroups are re-read from L2.
  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot8 = blockIdx.x >> 3;
  const uint32_t q8 = nwg >> 3, r8 = nwg & 7u;
  const uint32_t blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
  uint32_t w_first = blk * (uint32_t)P.wins_per_block;
  uint32_t w_last = w_first + (uint32_t)P.wins_per_block;
  if (w_last > (uint32_t)P.n_windows) w_last = (uint32_t)P.n_windows;
  if constexpr (DEEP) {  // here `item` walks the extra items, one block stride apart
    w_first = blockIdx.x;
    w_last = n_items;
  }

  for (int i = tid; i < kPlanes * kStride / 4; i += kBlock) ((uint4*)S.cnt)[i] = make_uint4(0, 0, 0, 0);
  if (tid == 0) S.n_alloc[0] = S.n_alloc[1] = 0;
  uint32_t par = 0;  // round parity (block-uniform)
  __syncthreads();

  // Everything a window needs from global memory before its unit loads is fetched one
  // window ahead (record range two ahead): window descriptor, record range, this thread's
  // first record core.
  Window n_win;
  int32_t n_r0, n_r1, nn_r0 = 0, nn_r1 = 0;
  if constexpr (DEEP) {
    const uint4 it = P.extra_items[w_first];
    n_win = P.windows[it.x];
    n_r0 = (int32_t)it.y;
    n_r1 = (int32_t)it.z;
  } else {
    n_win = P.windows[w_first];
    n_r0 = P.win_r0[w_first];
    n_r1 = P.win_r1[w_first];
    if (w_first + 1 < w_last) {
      nn_r0 = P.win_r0[w_first + 1];
      nn_r1 = P.win_r1[w_first + 1];
    }
  }
  uint4 pc0 = make_uint4(0, 0, 0, 0), pc1 = pc0;
  if (n_r0 + tid < n_r1) {
    const uint4* cp = (const uint4*)(P.cores + n_r0 + tid);
    pc0 = cp[0];
    pc1 = cp[1];
  }

  for (uint32_t item = w_first; item < w_last; item += DEEP ? gridDim.x : 1u) {
    const Window win = uni_window(n_win);
    const int32_t r0 = uni(n_r0), r1 = uni(n_r1);
    const int32_t g0 = win.g0, g1 = win.g1;
    const uint32_t wlen = (uint32_t)(g1 - g0);
    // A short window (a panel target) leaves most columns of the planes unused, and at depth
    // its few positions are what every lane of an instruction adds to.  It gets ncopy copies
    // of its nsub columns, lanes spread over them (neighbouring groups of 4 lanes hold
    // different records over the same positions); the flush adds the copies up.
    // (only in the launch over deep windows: the first launch stays free of it)
    const uint32_t nsub = (wlen + 31u) >> 5;
    const uint32_t ncopy = !DEEP ? 1u : nsub <= 4 ? 8u : nsub <= 8 ? 4u : nsub <= 16 ? 2u : 1u;
    const uint32_t cpo = DEEP ? (((uint32_t)tid >> 2) & (ncopy - 1u)) * nsub : 0u;
    BP_STAMP(0);

    // per-thread record cursor
    int32_t rec = r0 + tid;
    bool have = false, keep = false;
    uint32_t c_cur = 0, n_cigar = 0, blob = 0, lseqnc = 0, sslot = 0, aux = 0, span = 0;
    uint32_t cg0 = 0, cg1 = 0, cg2 = 0;
    int32_t rr = 0, qq = 0;
    auto flush = [&]() {

      // ---- flush: one lane per position; Depth = sum of the 12 planes (pileup.go:518-570);
      // the planes are zeroed as they are read, ready for the next window
      // The upper half of the waves takes three quarters of the positions: the lower half
      // holds the records of the next window and moves on to its phase 1 meanwhile.
      constexpr uint32_t kHalf = kBlock / 2, kCut = (uint32_t)kW / 8 * BP_FLUSH_UPPER;
      const bool upper = (uint32_t)tid >= kHalf;
      const uint32_t p_lo = upper ? (uint32_t)tid - kHalf : kCut + (uint32_t)tid;
      const uint32_t p_hi = upper ? kCut : (uint32_t)kW;
      const uint32_t p_end = ncopy > 1 ? nsub * 32u : (uint32_t)kW;  // columns past it hold the copies
      for (uint32_t p = p_lo; p < p_hi; p += kHalf) {
        if (p >= p_end) break;
        uint32_t v[kPlanes];
        uint32_t d = 0;
        uint32_t* const cp = S.cnt + (p & 31u) * kSub + (p >> 5);
  #pragma unroll
        for (int c = 0; c < kPlanes; c++) {
          v[c] = cp[c * kStride];
          d += v[c];
        }
        if (d != 0) {
  #pragma unroll
          for (int c = 0; c < kPlanes; c++) cp[c * kStride] = 0;
        }
        for (uint32_t cc = 1; cc < ncopy; cc++) {
          uint32_t* const cq = cp + cc * nsub;
          uint32_t dd = 0;
  #pragma unroll
          for (int c = 0; c < kPlanes; c++) {
            const uint32_t x = cq[c * kStride];
            v[c] += x;
            dd += x;
          }
          if (dd != 0) {
  #pragma unroll
            for (int c = 0; c < kPlanes; c++) cq[c * kStride] = 0;
          }
          d += dd;
        }
        if (p >= wlen) continue;
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
        if constexpr (DEEP) {
          if (d != 0) {
            atomicAdd(out, d);
  #pragma unroll
            for (int c = 0; c < 10; c++)
              if (v[c]) atomicAdd(out + 1 + c, v[c]);
          }
        } else {
          u32x4a o0 = {d, v[0], v[1], v[2]};
          u32x4a o1 = {v[3], v[4], v[5], v[6]};
          u32x3a o2 = {v[7], v[8], v[9]};
          *(u32x4a*)(out) = o0;
          *(u32x4a*)(out + 4) = o1;
          *(u32x3a*)(out + 8) = o2;
        }
      }
    };
    // A record's core -> the per-thread cursor.
    auto decode = [&](const uint4 c0, const uint4 c1) {
      const int32_t pos = (int32_t)c0.x;
      const uint32_t flags = c0.y & 0xffffu, mapq = (c0.y >> 16) & 0xffu;
      aux = c0.y >> 24;
      n_cigar = c0.z & 0xffffu;
      span = c0.z >> 16;
      const uint32_t l_seq = c0.w & 0xffffu;
      blob = c1.x;
      cg0 = c1.y; cg1 = c1.z; cg2 = c1.w;
      lseqnc = l_seq | (n_cigar << 16);
      // (The core is fetched as two 16-byte loads.  Without a quality rule nothing reads c0.w, and the
      // compiler narrows the first to a 12-byte load: global_load_dwordx3 at a 32-byte stride ran
      // the whole kernel 11 % slower, 4.30 -> 4.80 ms on chr1 30x.  This keeps the dword wanted.)
      asm volatile("" ::"v"(c0.w));
      const int strand = strand_of((aux & AUX_MATE_SAME) != 0, flags);
      // -flag-exclude, -mapq, blank-read filters (pileup.go:691)
      keep = ((uint32_t)P.flag_exclude & flags) == 0 && (int32_t)mapq >= P.mapq && n_cigar != 0;
      // strand filters (pileup.go:717-729)
      if (P.strand_req != 0) keep = keep && strand == P.strand_req;
      else keep = keep && (strand != 0 || P.ignore_strand);
      keep = keep && pos < g1 && (int64_t)pos + span > (int64_t)g0;
      sslot = P.ignore_strand ? 0u : (uint32_t)(strand - 1) & 1u;
      if constexpr (STITCH) {
        // a pair is counted on the strand of the read that found its mate (pileup.go:799-802);
        // stitch_resolve_kernel left that strand in the core of the read stored first
        if ((aux & AUX_PAIR_FIRST) && !P.ignore_strand) sslot = (aux >> 7) & 1u;
      }
      c_cur = 0;
      rr = pos - g0;
      qq = 0;
      have = true;
    };
    // round 0 works on the records fetched a window ahead; the next window's are fetched in
    // round 0 too, behind its first barrier, and the end-of-round barrier is the block-wide vote on
    // "records left"
    bool first_round = true;
    uint4 nc0 = make_uint4(0, 0, 0, 0), nc1 = nc0;
    bool next_issued = false;
    for (;;) {
      // (no barrier here: the slot counter of this round was reset behind the previous
      // round's first barrier, and phase 1 touches nothing the flush still reads)
      BP_STAMP(1);
      // ---- phase 1: thread = record; read-level filters, CIGAR -> M-segments in the window
      // (Wave priority: the two workgroups of a CU are in different phases most of the time; the
      // unit pass goes first, 4.32 -> 4.30 ms on chr1 30x.)
      __builtin_amdgcn_s_setprio(0);
      const bool has_rec = rec < r1;
      if (has_rec && !have) {
        uint4 c0 = pc0, c1 = pc1;
        if (!first_round) {
          const uint4* cp = (const uint4*)(P.cores + rec);
          c0 = cp[0];
          c1 = cp[1];
        }
        decode(c0, c1);
      }
      first_round = false;
      // CIGAR walk (pileup.go:275-363): every M op is one segment; the wave admits up to two
      // segments per lane per step, all lanes together
      bool more = has_rec && keep, failed = false;
      for (;;) {
        bool wantA = false, wantB = false;
        SegIn A = {0, 0, 0}, B = {0, 0, 0};
        uint32_t cB = 0;              // cursor after the last segment taken
        int32_t rrB = 0, qqB = 0;
        if (more) {
          const uint32_t* cg = (const uint32_t*)(P.blob + (uint64_t)blob * 8);
          if (aux & AUX_SIMPLE) {
            wantA = true;
            A.sref = rr;
            A.len = (int32_t)span;
          } else {
            uint32_t c = c_cur;
            int32_t r_ = rr, q_ = qq;
            int32_t sA = 0, lA = 0, qA = 0, sB = 0, lB = 0, qB = 0;
            wantA = next_m_op(cg, cg0, cg1, cg2, n_cigar, wlen, c, r_, q_, sA, lA, qA);
            if (wantA) wantB = next_m_op(cg, cg0, cg1, cg2, n_cigar, wlen, c, r_, q_, sB, lB, qB);
            A.sref = sA; A.len = lA; A.q0 = qA;
            B.sref = sB; B.len = lB; B.q0 = qB;
            cB = c; rrB = r_; qqB = q_;
            if (!wantA) more = false;
          }
        }
        if (!__any(wantA)) break;
        const bool ok = emit_wave<QM, STITCH>(S, wantA, A, wantB, B, wlen, blob, lseqnc, sslot, rec, par);
        if (wantA) {
          if (!ok) {
            failed = true;
            more = false;
          } else if (aux & AUX_SIMPLE) {
            more = false;
          } else {
            c_cur = cB; rr = rrB; qq = qqB;
            more = c_cur < n_cigar;
          }
        }
      }
      if (has_rec && !failed) {
        rec += kBlock;
        have = false;
      }
      const int pending = rec < r1;
      BP_STAMP(2);
      __syncthreads();
      BP_STAMP(3);

      // ---- prefetch for the next window: everything it needs from global memory before its
      // unit loads (window descriptor, record range two ahead, this thread's first record core)
      // is fetched while this window is in phase 2 and flushed.
      if (!next_issued) {
        next_issued = true;
        if constexpr (DEEP) {
          if (item + gridDim.x < w_last) {
            const uint4 it = P.extra_items[item + gridDim.x];
            n_win = P.windows[it.x];
            n_r0 = (int32_t)it.y;
            n_r1 = (int32_t)it.z;
            if (n_r0 + tid < n_r1) {
              const uint4* cp = (const uint4*)(P.cores + n_r0 + tid);
              nc0 = cp[0];
              nc1 = cp[1];
            }
          }
        } else if (item + 1 < w_last) {
          n_r0 = nn_r0;
          n_r1 = nn_r1;
          n_win = P.windows[item + 1];
          if (n_r0 + tid < n_r1) {
            const uint4* cp = (const uint4*)(P.cores + n_r0 + tid);
            nc0 = cp[0];
            nc1 = cp[1];
          }
          if (item + 2 < w_last) {
            nn_r0 = P.win_r0[item + 2];
            nn_r1 = P.win_r1[item + 2];
          }
        }
      }

      // ---- phase 2: thread = unit (32 window positions of one segment); kDepth units in
      // flight per lane
      __builtin_amdgcn_s_setprio(3);
      if (tid == 0) S.n_alloc[par ^ 1u] = 0;  // next round's counter; next touched after the barrier below
      const uint32_t n_alloc_u = (uint32_t)S.n_alloc[par];
      const uint32_t T = n_alloc_u < (uint32_t)kUnitCap ? n_alloc_u : (uint32_t)kUnitCap;
#ifdef BP_STAMPS_P2
      const uint32_t wbase = (uint32_t)tid & ~63u;
      // diagnostic: serialise one unit at a time and time its three parts
      {
        unsigned long long a_issue = 0, a_wait = 0, a_proc = 0;
        for (uint32_t tb = wbase; tb < T; tb += kBlock) {
          UnitLoad<QM> L1;
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long t0 = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
          issue_unit<QM, STITCH>(P, S, slot_of(tb + (tid & 63), T, (T + 63) / 64), T, L1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long t1 = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long t2 = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
          process_unit_plain<QM>(P, S, L1, cpo);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long t3 = __builtin_amdgcn_s_memtime();
          a_issue += t1 - t0; a_wait += t2 - t1; a_proc += t3 - t2;
        }
        if (P.dbg && (tid & 63) == 0) {
          unsigned long long* o = P.dbg + (unsigned long long)P.n_windows * 64 + ((uint64_t)item * (kBlock / 64) + (tid >> 6)) * 4;
          o[0] = a_issue; o[1] = a_wait; o[2] = a_proc; o[3] = (T + kBlock - 1) / kBlock;
        }
      }
#else
      {
        // Every thread runs n_it unit slots (block-uniform).  The steady-state loop has no
        // branch in its body, so the compiler's counted vmcnt waits keep kD units' loads in
        // flight; at 30x a window is 2-3 units per thread and the loop body rarely runs.
        constexpr int kD = STITCH ? kDepthQ : kDepth;
        const uint32_t R = (T + 63) / 64;  // rows of the 8 x R slot grid
        const uint32_t n_it = (R * 64 + kBlock - 1) / kBlock;
        UnitLoad<QM> L[kD];
        uint32_t t = (uint32_t)tid;
#pragma unroll
        for (int d = 0; d < kD; d++) {
          issue_unit<QM, STITCH>(P, S, slot_of(t, T, R), T, L[d]);
          t += kBlock;
        }
        uint32_t it = 0;
        for (; it + kD < n_it; it += kD) {
#pragma unroll
          for (int d = 0; d < kD; d++) {
            process_unit_plain<QM>(P, S, L[d], cpo);
            issue_unit<QM, STITCH>(P, S, slot_of(t, T, R), T, L[d]);
            t += kBlock;
          }
        }
        const uint32_t rem = n_it - it;
#pragma unroll
        for (int d = 0; d < kD; d++)
          if ((uint32_t)d < rem) process_unit_plain<QM>(P, S, L[d], cpo);
      }
#endif
      BP_STAMP(4);
      par ^= 1u;
      if (!__syncthreads_or(pending)) break;
    }
    BP_STAMP(5);
    flush();
    BP_STAMP(6);
    pc0 = nc0;
    pc1 = nc1;
  }
}

// ---- stitching: the pair rule, worked out once per job (addReadPair, pileup.go:406-446)
// A record without a mate in the table does nothing.  The read of a pair that was stored first
// gets the pair's strand into its core (the pair is counted on the strand of the read that found
// its mate, pileup.go:799-802).  For the read that found its mate both CIGARs are walked side by
// side and, for every reference position that an M op of each read covers:
//   * the mate's drop bit is set there and its pass bit cleared (the position is counted once, here);
//   * the reads disagree: the read's own base becomes N (code 15 -> X, counted whatever the quality);
//   * they agree: its own pass bit becomes qualPassTable.lookup2(q0, q1) on the clipped quals
//     (qthr[q] = smallest mate quality that passes beside q, qual.go:35-77); the counting kernel
//     then counts the base, or depth only, and X regardless, as it does for any base;
//   * they agree and a quality is >= 96: pass and drop bit both set - the reference panics on
//     that table lookup; the counting kernel reports the record when a window gets there.
// Each pair is worked once (AUX_PAIR_REV without AUX_PAIR_FIRST marks a read that has done it):
// a second pass would see the Ns it wrote.
// Mapping: thread = record for the table look-up; the reads of a wave's 64 records that have a pair
// to work are then dealt to its four groups of 16 lanes, lane = 8 consecutive bases of the read (one
// dword of packed bases, two of quals, one byte of pass bits; the mate's bytes for the same
// positions realigned in registers) or one dword of the mate's bits.  Pairs of single-M-op reads
// (nearly all) touch every byte once and use plain stores; with indels two stretches can meet in
// one dword, and the updates go through L2 atomics.
__device__ __forceinline__ uint32_t bits_lo_hi(int32_t lo, int32_t hi, int32_t width) {  // bits [lo, hi) of a width-bit word
  if (lo < 0) lo = 0;
  if (hi > width) hi = width;
  if (hi <= lo) return 0u;
  return (hi - lo >= 32 ? 0xffffffffu : ((1u << (hi - lo)) - 1u)) << lo;
}

// byte-wise a >= b (flag in bit 7 of each byte) for bytes a < 128, any b
__device__ __forceinline__ uint32_t ge_bytes(uint32_t a, uint32_t b) {
  constexpr uint32_t H = 0x80808080u;
  // b >= 128 can never be reached; else (a | 0x80) - b keeps bit 7 iff a >= b
  return ((a | H) - (b & 0x7f7f7f7fu)) & ~b & H;
}

// bytes of q at clipped read positions -> 2 (clipQuals, pileup_support.go:51-69); the group's 8
// bases are read indices b8 .. b8+7 of a read of l bases
__device__ __forceinline__ void clip8(uint32_t& qlo, uint32_t& qhi, int32_t b8, int32_t l, int32_t clip) {
  uint32_t cm;  // bit k: base b8 + k is clipped
  if (2 * clip >= l) cm = 0xffu;
  else cm = (bits_lo_hi(0, clip - b8, 8) | bits_lo_hi(l - clip - b8, 8, 8)) & 0xffu;
  if (!cm) return;
  const uint32_t lo = ((cm & 15u) * 0x00204081u & 0x01010101u) * 0xffu, hi = ((cm >> 4) * 0x00204081u & 0x01010101u) * 0xffu;
  qlo = (qlo & ~lo) | (0x02020202u & lo);
  qhi = (qhi & ~hi) | (0x02020202u & hi);
}

// One stretch of reference positions that an M op of each read covers: `len` positions from read
// index bi0 of the deciding read and mi0 of its mate.  Lane sl of the pair's 16 takes the 8-base
// groups g0 + sl, g0 + sl + 16, ... of the read and the bit dwords w0 + sl, ... of the mate; every
// load of a pass is issued before anything is stored.
__device__ __forceinline__ void resolve_stretch(const KernelParams& P, const uint8_t* qthr, uint32_t sl, bool plain,
                                                int32_t bi0, int32_t mi0, int32_t len, uint32_t ol, uint32_t ml,
                                                uint32_t* oseq, uint32_t* obits, const uint32_t* oq,
                                                const uint32_t* mseq, uint32_t* mbits, const uint32_t* mq) {
  // (a CIGAR longer than its read: nothing to look at past the read's end)
  if (len > (int32_t)ol - bi0) len = (int32_t)ol - bi0;
  if (len > (int32_t)ml - mi0) len = (int32_t)ml - mi0;
  if (len <= 0) return;
  const int32_t clip = P.clip;
  constexpr uint32_t H = 0x80808080u;
  const uint32_t g1 = (uint32_t)(bi0 + len - 1) >> 3, w1 = (uint32_t)(mi0 + len - 1) >> 5;
  uint32_t g = ((uint32_t)bi0 >> 3) + sl, w = ((uint32_t)mi0 >> 5) + sl;
  for (; g <= g1 || w <= w1; g += 16, w += 16) {
    const bool has_g = g <= g1, has_w = w <= w1;
    const uint32_t gg = has_g ? g : (uint32_t)bi0 >> 3;  // (an idle lane loads what its neighbour loads)
    const int32_t b8 = (int32_t)(8u * gg);  // read index of the group's first base
    const int32_t m0 = mi0 + (b8 - bi0);    // the mate's read index there (>= -7: the loads below may start
                                            // a few bytes before the mate's bases / quals, inside the buffer)
    const uint32_t* const mp = mseq + (m0 >> 3);
    const uint32_t* const mqp = mq + (m0 >> 2);
    uint32_t* const pw = obits + 2 * (size_t)(gg >> 2);
    uint8_t* const pb = (uint8_t*)pw + (gg & 3u);
    uint32_t* const mw = mbits + 2 * (size_t)(has_w ? w : (uint32_t)mi0 >> 5);
    const uint32_t sw = oseq[gg];
    const uint32_t mp0 = mp[0], mp1 = mp[1];
    const uint32_t qlo_ = oq[2 * gg], qhi_ = oq[2 * gg + 1];
    const uint32_t n0 = mqp[0], n1 = mqp[1], n2 = mqp[2];
    uint32_t pold = 0, mpass = 0, mdrop = 0;
    if (plain) {
      pold = *pb;
      mpass = mw[0];
      mdrop = mw[1];
    }
    if (has_g) {
      const uint32_t v8 = bits_lo_hi(bi0 - b8, bi0 + len - b8, 8);
      const uint32_t msw = __builtin_amdgcn_alignbit(mp1, mp0, 4u * ((uint32_t)m0 & 7u));
      uint32_t qlo = qlo_, qhi = qhi_;
      uint32_t mqlo = __builtin_amdgcn_alignbyte(n1, n0, (uint32_t)m0 & 3u);
      uint32_t mqhi = __builtin_amdgcn_alignbyte(n2, n1, (uint32_t)m0 & 3u);
      if (clip > 0) {
        clip8(qlo, qhi, b8, (int32_t)ol, clip);
        clip8(mqlo, mqhi, m0, (int32_t)ml, clip);
      }
      // bit k: the reads agree at base k of the group
      const uint32_t dn = msw ^ sw;
      uint32_t eq8 = ~(dn | (dn >> 1) | (dn >> 2) | (dn >> 3)) & 0x11111111u;
      eq8 = (eq8 | (eq8 >> 3)) & 0x03030303u;
      eq8 = (eq8 | (eq8 >> 6)) & 0x000f000fu;
      eq8 = (eq8 | (eq8 >> 12)) & 0xffu;
      const uint32_t x8 = v8 & ~eq8, e8 = v8 & eq8;
      // bytes >= 96 (in either read): the reference's table lookup is out of range there
      const uint32_t blo = ((((qlo & 0x7f7f7f7fu) + 0x20202020u) | qlo) | (((mqlo & 0x7f7f7f7fu) + 0x20202020u) | mqlo)) & H;
      const uint32_t bhi = ((((qhi & 0x7f7f7f7fu) + 0x20202020u) | qhi) | (((mqhi & 0x7f7f7f7fu) + 0x20202020u) | mqhi)) & H;
      const uint32_t bad8 = (((((blo >> 7) * 0x00204081u) >> 21) & 15u) | (((((bhi >> 7) * 0x00204081u) >> 21) & 15u) << 4)) & e8;
      // the pair passes iff mate qual >= qthr[own qual] (qualSumTable row, qual.go:35-77): 8 byte
      // lookups (indices clamped below 96: those positions are `bad` anyway), then a SWAR byte compare
      const uint32_t ilo = qlo & ~((blo >> 7) * 0xffu) & 0x7f7f7f7fu, ihi = qhi & ~((bhi >> 7) * 0xffu) & 0x7f7f7f7fu;
      const uint32_t tlo = (uint32_t)qthr[ilo & 0xffu] | ((uint32_t)qthr[(ilo >> 8) & 0xffu] << 8) |
                           ((uint32_t)qthr[(ilo >> 16) & 0xffu] << 16) | ((uint32_t)qthr[ilo >> 24] << 24);
      const uint32_t thi = (uint32_t)qthr[ihi & 0xffu] | ((uint32_t)qthr[(ihi >> 8) & 0xffu] << 8) |
                           ((uint32_t)qthr[(ihi >> 16) & 0xffu] << 16) | ((uint32_t)qthr[ihi >> 24] << 24);
      // byte-wise mq >= t for mq < 128 (a bad byte's result is overridden), t <= 255: 9-bit compare in two halves
      const uint32_t plo = ge_bytes(mqlo, tlo), phi = ge_bytes(mqhi, thi);
      const uint32_t pass8 = ((((plo >> 7) * 0x00204081u) >> 21) & 15u) | (((((phi >> 7) * 0x00204081u) >> 21) & 15u) << 4);
      const uint32_t set8 = e8 & (pass8 | bad8), clr8 = e8 & ~set8;
      if (x8) {
        uint32_t m = (x8 | (x8 << 12)) & 0x000f000fu;
        m = (m | (m << 6)) & 0x03030303u;
        m = (m | (m << 3)) & 0x11111111u;
        m = (m << 4) - m;
        if (plain) oseq[g] = sw | m; else atomicOr(&oseq[g], m);
      }
      if (plain) {
        if (e8) *pb = (uint8_t)((pold & ~clr8) | set8);
        if (bad8) pb[4] = (uint8_t)(pb[4] | bad8);
      } else {
        const uint32_t sh = 8u * (g & 3u);
        if (clr8) atomicAnd(&pw[0], ~(clr8 << sh));
        if (set8) atomicOr(&pw[0], set8 << sh);
        if (bad8) atomicOr(&pw[1], bad8 << sh);
      }
    }
    if (has_w) {
      const uint32_t m = bits_lo_hi(mi0 - (int32_t)(32u * w), mi0 + len - (int32_t)(32u * w), 32);
      if (plain) {
        mw[0] = mpass & ~m;
        mw[1] = mdrop | m;
      } else {
        atomicAnd(&mw[0], ~m);
        atomicOr(&mw[1], m);
      }
    }
  }
}

// next M op of a CIGAR: [s, e) on the reference (relative to the deciding read's start), first read index q0
struct MOpCursor {
  const uint32_t* cg;
  uint32_t n, k;
  int32_t r, q, s, e, q0;
  __device__ __forceinline__ bool next() {
    while (k < n) {
      const uint32_t cop = cg[k++];
      const uint32_t op = cop & 15u;
      const int32_t len = (int32_t)(cop >> 4);
      if (op == 0) {
        s = r; e = r + len; q0 = q;
        r += len; q += len;
        return true;
      }
      q += (op == 1 || op == 4) ? len : 0;
      r += (op == 2 || op == 3) ? len : 0;
      if (op > 5) k = n;  // (the host rejected reachable bad ops, pileup.go:355-359)
    }
    return false;
  }
};

__global__ __launch_bounds__(256) void stitch_resolve_kernel(KernelParams P, int64_t n) {
  __shared__ uint8_t s_qthr[96];
  // what the 16 lanes working a pair need of it, left by the lane that looked the record up:
  // mate entry x, y, z, w | core flags/aux, n_cigar/span, l_seq/qdelta, blob
  __shared__ uint4 s_pair[4][64][2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid < 96) s_qthr[tid] = P.qthr[tid];
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  bool work = false;
  uint4 me = make_uint4(0, 0, 0, 0), c0 = me;
  uint32_t oblob = 0;
  if (i < n) {
    me = ((const uint4*)P.mate)[i];
    c0 = ((const uint4*)(P.cores + i))[0];
    oblob = ((const uint32_t*)(P.cores + i))[4];
    if (me.x & MATE_HAS) {
      uint8_t* const aux_p = (uint8_t*)(P.cores + i) + 7;  // ReadCore::aux
      const uint32_t aux = c0.y >> 24;
      if (!(me.x & MATE_SECOND)) {
        // (proper pairs have both reads on one strand by pileup.GetStrand: the core is written only
        // where the pair's strand is not the read's own)
        const uint32_t own_rev = (((aux >> AUX_STRAND_SHIFT) & 3u) - 1u) & 1u, pair_rev = (me.x & MATE_PAIR_REV) ? 1u : 0u;
        if (own_rev != pair_rev && !(aux & AUX_PAIR_FIRST))
          *aux_p = (uint8_t)(aux | AUX_PAIR_FIRST | (pair_rev ? AUX_PAIR_REV : 0));
      } else if ((aux & (AUX_PAIR_FIRST | AUX_PAIR_REV)) != AUX_PAIR_REV) {  // not done in an earlier pass
        // the mate starts `delta` after this read and spans me.z & 0xffff: can they share a position at all?
        const int32_t delta = ((int32_t)(me.x << 15)) >> 15, ospan = (int32_t)(c0.z >> 16), mspan = (int32_t)(me.z & 0xffffu);
        // (spans are clamped to 0xffff in the core and the table: such a read may reach further)
        if ((delta < ospan || ospan == 0xffff) && (delta + mspan > 0 || mspan == 0xffff)) {
          *aux_p = (uint8_t)(aux | AUX_PAIR_REV);
          work = true;
        }
      }
    }
  }
  const unsigned long long todo = __ballot(work);
  if (work) {
    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
    s_pair[wv][slot][0] = me;
    s_pair[wv][slot][1] = make_uint4(c0.y, c0.z, c0.w, oblob);
  }
  __syncthreads();
  const uint32_t n_todo = (uint32_t)__popcll(todo);
  const uint32_t sub = lane >> 4, sl = lane & 15u;
  uint8_t* const blob = (uint8_t*)P.blob;
  for (uint32_t k = sub; k < n_todo; k += 4) {
    const uint4 m4 = s_pair[wv][k][0], c4 = s_pair[wv][k][1];
    const uint32_t on = c4.y & 0xffffu, ol = c4.z & 0xffffu, oqd = c4.z >> 16;
    const uint32_t mn = m4.w & 0xffffu, mqd = m4.w >> 16, ml = m4.z >> 16;
    if (ol == 0xffffu || ml == 0xffffu) continue;  // (no payload kept: such a read passes no max-read-len)
    uint8_t* const ob = blob + (uint64_t)c4.w * 8;
    uint8_t* const mb = blob + (uint64_t)m4.y * 8;
    uint32_t* const oseq = (uint32_t*)(ob + ((on + 1) & ~1u) * 4);
    uint32_t* const mseq = (uint32_t*)(mb + ((mn + 1) & ~1u) * 4);
    uint32_t* const obits = oseq + ((ol + 7) >> 3);
    uint32_t* const mbits = mseq + ((ml + 7) >> 3);
    const uint32_t* const oq = (const uint32_t*)(ob + 8 * (size_t)oqd);
    const uint32_t* const mq = (const uint32_t*)(mb + 8 * (size_t)mqd);
    const int32_t delta = ((int32_t)(m4.x << 15)) >> 15;  // mate start - own start, 17 bits signed
    if (((c4.x >> 24) & AUX_SIMPLE) && (m4.x & MATE_SIMPLE)) {
      // one M op each: one stretch, no CIGAR to fetch, every byte touched once
      const int32_t ospan = (int32_t)(c4.y >> 16), mspan = (int32_t)(m4.z & 0xffffu);
      const int32_t lo = delta > 0 ? delta : 0, hi = ospan < delta + mspan ? ospan : delta + mspan;
      if (hi > lo) resolve_stretch(P, s_qthr, sl, true, lo, lo - delta, hi - lo, ol, ml, oseq, obits, oq, mseq, mbits, mq);
      continue;
    }
    MOpCursor A{(const uint32_t*)ob, on, 0, 0, 0, 0, 0, 0};
    MOpCursor B{(const uint32_t*)mb, mn, 0, delta, 0, 0, 0, 0};
    bool ha = A.next(), hb = B.next();
    while (ha && hb) {
      const int32_t lo = A.s > B.s ? A.s : B.s, hi = A.e < B.e ? A.e : B.e;
      if (hi > lo)
        resolve_stretch(P, s_qthr, sl, false, A.q0 + (lo - A.s), B.q0 + (lo - B.s), hi - lo, ol, ml, oseq, obits, oq, mseq,
                        mbits, mq);
      if (A.e <= B.e) ha = A.next(); else hb = B.next();
    }
  }
}

void launch_stitch_resolve(const KernelParams& P, hipStream_t s) {
  if (!P.mate) return;
  // (n_reads 0: one workgroup that finds nothing to do)
  hipLaunchKernelGGL(stitch_resolve_kernel, dim3((unsigned)std::max<int64_t>(1, (P.n_reads + 255) / 256)), dim3(256), 0, s, P, P.n_reads);
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

// ---------------------------------------------------------------- per-read feature lists
// appendBase (pileup.go:235-250): with a per-read column set every passing non-X base also
// appends {index in read, read length, qual, GetStrand} to PerRead[base] of its position, in
// read-arrival order.  The list lengths are the counts the pileup kernel already produced
// (Counts[b][0] + Counts[b][1]), so the lists are laid out by an exclusive scan of the counts
// and filled by a gather that visits the records in index order (= arrival order): the
// reference's order with no sort and no contended atomics.
//
// The gather is a banded transpose (records x positions -> positions x records) and goes
// through LDS like one.  A wave owns a tile of kPrT = 32 consecutive BED positions and walks
// the records that can reach it, 64 at a time:
//   phase 1  lane = record.  The CIGAR walk leaves one 16-byte piece per M op that overlaps the
//            tile, in (record, op) order, in a small LDS list (wave prefix sum for the slots).
//   phase 2  lane = piece, two chunks of 64 at a time, all loads before the first store.
//            The 32 bases and quals under the tile are loaded with wide loads and
//            realigned in registers (tile position k <-> nibble/byte k).  Then, position by
//            position: a ballot of the lanes whose base is the position's most frequent one
//            gives every such lane its rank (mbcnt) behind the position's cursor - lanes are
//            in record order, so ranks are arrival order; the few lanes with another listed base
//            take a second, per-base ballot.  The lanes of one ballot are consecutive entries
//            of one list, so every store instruction writes one contiguous run of the output.

__global__ void text_tile_scan_kernel(unsigned long long* tile, int64_t n_tiles);  // exclusive scan by one workgroup (below)

// List lengths (lens[4r + b] = Counts[b][0] + Counts[b][1] of row r, read straight from the
// rows) scanned in tiles of 1024 lists: off32 gets the exclusive sum inside the tile, tile_sum
// the tile's total.  The host scans the totals; a list's offset is tile_base[i >> 10] + off32[i]
// (pr_off), so there is no second pass over the offsets.
__global__ __launch_bounds__(256) void tile_scan_kernel(const uint32_t* rows, uint32_t* off32, int64_t n,
                                                        unsigned long long* tile_sum, int32_t* overflow) {
  __shared__ unsigned long long sh[256];
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 4 lists per row
  const int64_t base = row * 4;
  unsigned long long v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = base + k < n ? (unsigned long long)rows[row * 11 + 1 + 2 * k] + rows[row * 11 + 2 + 2 * k] : 0;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned long long t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = sh[threadIdx.x] - s;
  if (base < n) {  // n is a multiple of 4; a tile's total is checked against 2^32 on the host
    uint4 o;
    o.x = (uint32_t)run; run += v[0];
    o.y = (uint32_t)run; run += v[1];
    o.z = (uint32_t)run; run += v[2];
    o.w = (uint32_t)run;
    *(uint4*)(off32 + base) = o;
  }
  if (threadIdx.x == 255) {
    tile_sum[blockIdx.x] = sh[255];
    if (sh[255] > 0xffffffffull) *overflow = 1;  // the in-tile offsets are 32 bits
  }
}

// offset of list i (i = 4 x row + base; i = n_lists gives the total)
__device__ __forceinline__ unsigned long long pr_off(const PerReadParams& Q, int64_t i) {
  return i >= Q.n_lists ? Q.total : Q.tile_base[i >> 10] + Q.off32[i];
}

// a[e] without a jump table: e is a lane's own
__device__ __forceinline__ uint32_t sel4(const uint32_t (&a)[4], uint32_t e) {
  return (a[0] & (0u - (uint32_t)(e == 0))) | (a[1] & (0u - (uint32_t)(e == 1))) | (a[2] & (0u - (uint32_t)(e == 2))) |
         (a[3] & (0u - (uint32_t)(e == 3)));
}

// bits [lo, hi) of a 32-bit word, lo/hi clamped to [0, 32]
__device__ __forceinline__ uint32_t bit_range(int32_t lo, int32_t hi) {
  lo = lo < 0 ? 0 : lo;
  hi = hi > 32 ? 32 : hi;
  if (hi <= lo) return 0u;
  const uint32_t upto_hi = hi >= 32 ? 0xffffffffu : (1u << hi) - 1u;
  return upto_hi & ~((1u << lo) - 1u);  // lo < hi <= 32: lo <= 31
}

// One entry of a list.  The lanes of a store are consecutive entries of one list (ranks of
// one ballot), so a wave instruction writes one contiguous run of the output.  6 x slot is a
// 32-bit byte offset from a scalar base (a tile's lists stay far below 4 GiB).  Three 2-byte
// stores: one aligned dword + one short per entry (the dword at +0 or +2 by the slot's parity)
// was measured slower, 36.5 -> 39.7 ms on chr1 at 60x.
__device__ __forceinline__ void pr_put(uint8_t* gout, uint32_t slot, uint32_t w01, uint32_t w2) {
  uint16_t* const d = (uint16_t*)(gout + ((slot << 2) + (slot << 1)));
  d[0] = (uint16_t)w01;
  d[1] = (uint16_t)(w01 >> 16);
  d[2] = (uint16_t)w2;
}

// tile -> its BED interval and its index inside it
__device__ __forceinline__ int32_t pr_tile_interval(const PerReadParams& Q, int64_t t, int64_t& chunk) {
  int32_t a = 0, b = Q.n_iv;  // last interval with iv_tile0 <= t
  while (b - a > 1) {
    const int32_t m = (a + b) >> 1;
    if (Q.iv_tile0[m] <= t) a = m; else b = m;
  }
  chunk = t - Q.iv_tile0[a];
  return a;
}

// Per tile: the first record that can reach it (pos > first position - max_span).  A kernel of
// its own, thread = tile: the searches are chains of dependent loads, which thousands of
// independent threads hide and one wave per tile does not.
__global__ void per_read_plan_kernel(KernelParams P, PerReadParams Q, int32_t* tile_lo) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Q.n_tiles) return;
  int64_t chunk;
  const int32_t a = pr_tile_interval(Q, t, chunk);
  const int32_t ref = Q.iv_ref[a];
  int64_t want = (int64_t)P.iv_start[a] + chunk * kPrT - P.max_span + 1;
  if (want < 0) want = 0;
  tile_lo[t] = (int32_t)first_at_or_after(P, P.ref_first[ref], P.ref_first[ref + 1], want);
}

// What a lane holds of its piece for one chunk: the 32 bases (nibble k = tile position k,
// 0 outside the piece) and 32 quals under the tile, and the constant parts of its entries.
struct PrChunk {
  uint32_t B[4], Qv[8];
  uint32_t w01base, w2hi, clipped;
};

template <bool CLIP>
__global__ __launch_bounds__(kPrBlock) void per_read_kernel(KernelParams P, PerReadParams Q) {
  extern __shared__ __align__(16) uint8_t pr_lds[];
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction, and told so: the tile's interval, rows, offsets and record range
  // are then worked out on the scalar unit and held in SGPRs (95 -> 82 VGPRs, 36.9 -> 35.2 ms)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // consecutive workgroups go to different XCDs, each with its own L2: give every XCD a
  // contiguous run of tiles, so that the records neighbouring tiles share come from one L2
  const uint32_t per_xcd = gridDim.x >> 3;  // the launcher rounds the grid up to a multiple of 8
  const int64_t t = ((int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * (kPrBlock / 64) + wave;
  if (t >= Q.n_tiles) return;  // wave-uniform; the kernel has no workgroup barrier
  uint8_t* const wl = pr_lds + (size_t)wave * kPrWaveBytes;
  uint4* const pieces = (uint4*)wl;                               // [kPrPieces]
  uint32_t* const cur = (uint32_t*)(wl + kPrPieces * 16);         // [kPrT][4] next free entry of every list
  uint32_t* const curmaj = cur + kPrT * 4;                        // [kPrT] ... of the major list
  // tile -> interval, positions, rows
  int64_t chunk;
  const int32_t a = pr_tile_interval(Q, t, chunk);
  const int32_t ref = Q.iv_ref[a];
  const int32_t p0 = P.iv_start[a] + (int32_t)(chunk * kPrT);
  const int32_t iv_end = P.iv_end[a];
  const int32_t pend = iv_end - p0 < kPrT ? iv_end : p0 + kPrT;  // the tile is [p0, pend)
  const int32_t n_pos = pend - p0;
  const int64_t R0 = P.iv_row[a] + chunk * kPrT;
  const unsigned long long off0 = pr_off(Q, 4 * R0), off1 = pr_off(Q, 4 * (R0 + n_pos));
  if (off1 - off0 > 600000000ull) {  // 6 x entry index must fit 32 bits (a tile this deep: 18 M reads on 32 positions)
    atomicMin(Q.bad_row, (int32_t)(R0 > 0x7ffffffe ? 0x7ffffffe : R0));
    return;
  }
  const uint32_t n_tile = (uint32_t)(off1 - off0);
  // lane k < 32 sets up position k: its four lists inside the tile's stretch of the output and
  // which of them is the longest ("major": it gets the cheap path).  The cursors (next free
  // entry of every list) live in LDS: curmaj[k] for the major list, cur[k][e] for the others.
  const int pl = lane & (kPrT - 1);
  const bool active = lane < kPrT && pl < n_pos;
  uint32_t lbase[4] = {0, 0, 0, 0}, llen[4] = {0, 0, 0, 0};
  if (active) {
    unsigned long long o[5];
#pragma unroll
    for (int e = 0; e < 5; e++) o[e] = pr_off(Q, 4 * (R0 + pl) + e);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      lbase[e] = (uint32_t)(o[e] - off0);
      llen[e] = (uint32_t)(o[e + 1] - o[e]);
    }
  }
  uint32_t maj = 0;
#pragma unroll
  for (uint32_t e = 1; e < 4; e++)
    if (llen[e] > sel4(llen, maj)) maj = e;
  const uint32_t v_maj = 1u << maj;  // Seq8 code of the major base (fasta.go:50-62)
  if (lane < kPrT) {
    curmaj[pl] = sel4(lbase, maj);
#pragma unroll
    for (int e = 0; e < 4; e++) cur[pl * 4 + e] = lbase[e];
  }
  // the tile's lists are one stretch of the output, addressed from a scalar base
  const unsigned long long off0s = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(off0 >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off0);
  uint8_t* const gout = (uint8_t*)Q.out + off0s * 6;

  // records of the reference that can cover the tile: pos in (p0 - max_span, pend), the first
  // of them found by per_read_plan_kernel
  const int64_t hi = P.ref_first[ref + 1];
  const int64_t lo = Q.tile_lo[t];
  const uint4* cores4 = (const uint4*)P.cores;
  const uint32_t minq = (uint32_t)P.min_base_qual;

  uint32_t n_q = 0;  // pieces waiting in the list
  uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = nx0;  // ReadCore of record i + lane, loaded a round ahead
  if (lo + lane < hi) {
    nx0 = cores4[2 * (lo + lane)];
    nx1 = cores4[2 * (lo + lane) + 1];
  }
  for (int64_t i = lo;;) {
    // ---- phase 1: lane = record i + lane; its pieces join the list
    bool last = true;
    int n_take = 64;
    if (i < hi) {
      const bool have = i + lane < hi;
      const uint4 c0 = nx0, c1 = nx1;
      const int32_t pos = (int32_t)c0.x;
      const uint32_t aux = c0.y >> 24, n_cigar = c0.z & 0xffffu, span = c0.z >> 16, l_seq = c0.w & 0xffffu,
                     qdelta = c0.w >> 16;
      // records are sorted: once one starts behind the tile, so does every later one
      const bool last_batch = __builtin_amdgcn_ballot_w64(!have || pos >= pend) != 0;
      // record-level tests (pileup.go:691, :717-729) were folded into AUX_KEEP by the packer
      const bool rel = have && (aux & AUX_KEEP) && pos < pend && (int64_t)pos + span > (int64_t)p0;
      const uint32_t* cg = (const uint32_t*)(P.blob + (uint64_t)c1.x * 8);
      uint32_t n_mine = 0;
      if (rel) {
        if (aux & AUX_SIMPLE) {
          n_mine = 1;
        } else {
          int32_t r = pos;
          for (uint32_t k = 0; k < n_cigar && r < pend; k++) {
            const uint32_t w = k == 0 ? c1.y : k == 1 ? c1.z : k == 2 ? c1.w : cg[k];
            const uint32_t op = w & 15u;
            const int32_t len = (int32_t)(w >> 4);
            if (op == 0) {
              if (len > 0 && r + len > p0) n_mine++;   // r < pend holds
              r += len;
            } else if (op == 2 || op == 3) {
              r += len;
            } else if (op != 1 && op != 4 && op != 5) {
              break;
            }
          }
        }
      }
      const uint32_t incl = wave_inclusive_scan(n_mine);
      // records whose pieces still fit the list (fewer than 128 wait in it, a record has at
      // most kPrT: at least one record always fits)
      const unsigned long long fit = __builtin_amdgcn_ballot_w64(n_q + incl <= (uint32_t)kPrPieces);
      n_take = fit == ~0ull ? 64 : __builtin_ctzll(~fit);
      if (rel && lane < n_take && n_mine) {
        uint32_t slot = n_q + incl - n_mine;
        const uint32_t dz = qdelta | (((n_cigar + 1u) >> 1) << 16);
        const uint32_t dw = l_seq | (((aux >> AUX_STRAND_SHIFT) & 3u) << 16);
        if (aux & AUX_SIMPLE) {
          const int32_t pa = pos > p0 ? pos - p0 : 0;
          const int32_t pb = ((int64_t)pos + span < (int64_t)pend ? pos + (int32_t)span : pend) - p0;
          pieces[slot] = make_uint4(c1.x, ((uint32_t)(p0 - pos) & 0xfffffu) | ((uint32_t)pa << 20) | ((uint32_t)pb << 26), dz, dw);
        } else {
          int32_t r = pos, qi = 0;
          for (uint32_t k = 0; k < n_cigar && r < pend; k++) {
            const uint32_t w = k == 0 ? c1.y : k == 1 ? c1.z : k == 2 ? c1.w : cg[k];
            const uint32_t op = w & 15u;
            const int32_t len = (int32_t)(w >> 4);
            if (op == 0) {
              if (len > 0 && r + len > p0) {
                const int32_t pa = r > p0 ? r - p0 : 0;
                const int32_t pb = (r + len < pend ? r + len : pend) - p0;
                pieces[slot++] = make_uint4(c1.x, ((uint32_t)(qi + p0 - r) & 0xfffffu) | ((uint32_t)pa << 20) | ((uint32_t)pb << 26), dz, dw);
              }
              r += len; qi += len;
            } else if (op == 1 || op == 4) {
              qi += len;
            } else if (op == 2 || op == 3) {
              r += len;
            } else if (op != 5) {
              break;
            }
          }
        }
      }
      n_q += (uint32_t)__builtin_amdgcn_readlane((int)incl, n_take - 1);
      last = last_batch && n_take == 64;
      __builtin_amdgcn_wave_barrier();
      if (!last) {  // the next round's records, in flight while this round's pieces are worked on
        nx0 = nx1 = make_uint4(0, 0, 0, 0);
        if (i + n_take + lane < hi) {
          nx0 = cores4[2 * (i + n_take + lane)];
          nx1 = cores4[2 * (i + n_take + lane) + 1];
        }
      }
    }

    // more records first, while fewer than two chunks of pieces wait
    if (!last && n_q < 128) {
      i += n_take;
      continue;
    }

    // ---- phase 2: lane = piece, two chunks of 64 at a time; the loads of both chunks are
    // issued before the first store (loads and stores share the in-order vmcnt counter).  A
    // tile of 32 positions at 60x has ~80 pieces: all its loads come before all its stores.
    // (Measured: no faster than one chunk at a time; kept because it is no slower either and
    // the wait for a chunk's loads no longer sits behind the previous chunk's stores.)
    uint32_t pc = 0;
    for (; last ? pc < n_q : pc + 128 <= n_q; pc += 128) {
      // No cursor may stand behind the tile's stretch (it would if a list outgrew the count that
      // sized it): two chunks move a cursor by at most 128 entries and the output buffer ends in
      // that much slack, so checking here keeps every store inside the buffer.
      {
        bool over = false;
        if (lane < kPrT) {
          const uint4 c4 = *(const uint4*)(cur + 4 * pl);
          const uint32_t cmx = curmaj[pl];
          over = c4.x > n_tile || c4.y > n_tile || c4.z > n_tile || c4.w > n_tile || cmx > n_tile;
        }
        if (__builtin_amdgcn_ballot_w64(over)) {
          atomicMin(Q.bad_row, (int32_t)(R0 > 0x7ffffffe ? 0x7ffffffe : R0));
          return;
        }
      }
      PrChunk ch[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        PrChunk& C = ch[c];
        const uint32_t at = pc + 64u * c;
        uint4 d = make_uint4(0, 0, 0, 0);
        if (at + lane < n_q) d = pieces[at + lane];
        const int32_t x0 = ((int32_t)(d.y << 12)) >> 12;   // read index under tile position 0 (may be < 0)
        const int32_t pa = (int32_t)((d.y >> 20) & 63u), pb = (int32_t)(d.y >> 26);  // valid tile positions [pa, pb)
        C.w01base = (uint32_t)x0 + ((d.w & 0xffffu) << 16);  // + k: {index in read, read length}
        C.w2hi = ((d.w >> 16) & 3u) << 8;
        C.clipped = 0;  // tile positions whose qual clipQuals sets to 2 (pileup_support.go:47-69)
        if (CLIP) {
          const int32_t pl_seq = (int32_t)(d.w & 0xffffu);
          if (2 * P.clip >= pl_seq) C.clipped = 0xffffffffu;
          else C.clipped = bit_range(0, P.clip - x0) | bit_range(pl_seq - P.clip - x0, 32);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) C.B[j] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) C.Qv[j] = 0;
        if (at < n_q) {  // wave-uniform: the second chunk of the last pair may not exist
          const uint8_t* rb = P.blob + (uint64_t)d.x * 8;
          // 32 bases under the tile: nibbles x0 .. x0+31 of the packed bases
          const uint32_t* bp = (const uint32_t*)(rb + (size_t)(d.z >> 16) * 8) + (x0 >> 3);
          const u32x4a bv = *(const u32x4a*)bp;  // one 16-byte and one 4-byte load (4-byte aligned)
          const uint32_t b4 = bp[4];
          // and the 32 quals: bytes x0 .. x0+31
          const uint32_t* qp = (const uint32_t*)(rb + 8ull * (d.z & 0xffffu)) + (x0 >> 2);
          const u32x4a qv = *(const u32x4a*)qp, qw = *(const u32x4a*)(qp + 4);
          const uint32_t q8 = qp[8];
          const uint32_t br[5] = {bv.x, bv.y, bv.z, bv.w, b4};
          const uint32_t qr[9] = {qv.x, qv.y, qv.z, qv.w, qw.x, qw.y, qw.z, qw.w, q8};
          const uint32_t shb = ((uint32_t)x0 & 7u) * 4u, shq = ((uint32_t)x0 & 3u) * 8u;
#pragma unroll
          for (int j = 0; j < 4; j++)  // positions outside the piece read as 0
            C.B[j] = __builtin_amdgcn_alignbit(br[j + 1], br[j], shb) & bit_range(4 * (pa - 8 * j), 4 * (pb - 8 * j));
#pragma unroll
          for (int j = 0; j < 8; j++) C.Qv[j] = __builtin_amdgcn_alignbit(qr[j + 1], qr[j], shq);
        }
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        if (pc + 64u * c >= n_q) break;
        PrChunk C = ch[c];
#pragma unroll 1
        for (int j = 0; j < kPrT / 8; j++) {
          const uint32_t Bj = C.B[0], Q0 = C.Qv[0], Q1 = C.Qv[1];
          // the 8 major cursors of the group (the same value in every lane)
          uint32_t cm[8];
          {
            const uint4 c03 = *(const uint4*)(curmaj + 8 * j), c47 = *(const uint4*)(curmaj + 8 * j + 4);
            cm[0] = c03.x; cm[1] = c03.y; cm[2] = c03.z; cm[3] = c03.w;
            cm[4] = c47.x; cm[5] = c47.y; cm[6] = c47.z; cm[7] = c47.w;
          }
#pragma unroll
          for (int kk = 0; kk < 8; kk++) {
            const int k = 8 * j + kk;
            const uint32_t nib = (Bj >> (4 * kk)) & 15u;
            uint32_t q = ((kk < 4 ? Q0 : Q1) >> (8 * (kk & 3))) & 255u;
            if (CLIP) q = (C.clipped >> k) & 1u ? 2u : q;
            const uint32_t s_maj = (uint32_t)__builtin_amdgcn_readlane((int)v_maj, k);
            const unsigned long long Qm = __builtin_amdgcn_ballot_w64(q >= minq);
            const unsigned long long M = __builtin_amdgcn_ballot_w64(nib == s_maj) & Qm;
            const unsigned long long Pm = __builtin_amdgcn_ballot_w64(nib != 0) & Qm;
            const uint32_t w01 = C.w01base + (uint32_t)k;
            const uint32_t w2 = q | C.w2hi;
            {
              const uint32_t slot = cm[kk] + __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
              if (nib == s_maj && q >= minq) pr_put(gout, slot, w01, w2);
              cm[kk] += (uint32_t)__builtin_popcountll(M);
            }
            if (Pm != M) {
              // other listed bases (Seq8ToEnumTable, pileup/common.go:66: 1,2,4,8 -> A,C,G,T; anything
              // else is X: counted, never listed, pileup.go:239-240).  One base code per round,
              // the lowest waiting lane's: nearly always a single round.
              const bool minor = nib != 0 && nib != s_maj && q >= minq && (nib & (nib - 1u)) == 0;
              unsigned long long Mm = __builtin_amdgcn_ballot_w64(minor);
              while (Mm) {
                const uint32_t nib1 = (uint32_t)__builtin_amdgcn_readlane((int)nib, (int)__builtin_ctzll(Mm));
                const unsigned long long Me = __builtin_amdgcn_ballot_w64(nib == nib1) & Mm;
                uint32_t* const ce = cur + k * 4 + ((nib1 >> 1) - (nib1 >> 3));
                const uint32_t c0 = *ce;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(Me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Me, 0u));
                if (minor && nib == nib1) {
                  pr_put(gout, c0 + rank, w01, w2);
                  if (rank == 0) *ce = c0 + (uint32_t)__builtin_popcountll(Me);
                }
                Mm &= ~Me;
              }
            }
          }
          if (lane == 0) {
            *(uint4*)(curmaj + 8 * j) = make_uint4(cm[0], cm[1], cm[2], cm[3]);
            *(uint4*)(curmaj + 8 * j + 4) = make_uint4(cm[4], cm[5], cm[6], cm[7]);
          }
          // next 8 positions
          C.B[0] = C.B[1]; C.B[1] = C.B[2]; C.B[2] = C.B[3];
#pragma unroll
          for (int x = 0; x < 6; x++) C.Qv[x] = C.Qv[x + 2];
        }
      }
    }
    if (last) break;
    // what is left (fewer than 128 pieces) moves to the front of the list
    if (pc > 0 && pc < n_q) {
      uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
      if (pc + lane < n_q) d0 = pieces[pc + lane];
      if (pc + 64 + lane < n_q) d1 = pieces[pc + 64 + lane];
      __builtin_amdgcn_wave_barrier();
      if (pc + lane < n_q) pieces[lane] = d0;
      if (pc + 64 + lane < n_q) pieces[64 + lane] = d1;
    }
    n_q -= pc;
    __builtin_amdgcn_wave_barrier();
    i += n_take;
  }
  // the lists must be exactly as long as the counts said
  if (active) {
    bool complete = true;
#pragma unroll
    for (uint32_t e = 0; e < 4; e++) {
      const uint32_t endc = e == maj ? curmaj[pl] : cur[pl * 4 + e];
      complete = complete && endc == lbase[e] + llen[e];
    }
    if (!complete) {
      const int64_t R = R0 + pl;
      atomicMin(Q.bad_row, (int32_t)(R > 0x7ffffffe ? 0x7ffffffe : R));
    }
  }
}

// tile_sum: [n_tiles + 1]; on return (stream order) tile_sum[t] = entries before tile t and
// tile_sum[n_tiles] = entries in all lists.  *overflow is set when a tile holds 2^32 or more.
void launch_tile_scan(const uint32_t* rows, uint32_t* off32, int64_t n, unsigned long long* tile_sum,
                      int32_t* overflow, hipStream_t s) {
  const int64_t n_tiles = (n + 1023) / 1024;
  if (n) hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, rows, off32, n, tile_sum, overflow);
  hipLaunchKernelGGL(text_tile_scan_kernel, dim3(1), dim3(1024), 0, s, tile_sum, n_tiles);
}
int per_read_tile_positions() { return kPrT; }
void launch_per_read(const KernelParams& P, PerReadParams Q, hipStream_t s) {
  if (Q.n_tiles == 0) return;
  hipLaunchKernelGGL(per_read_plan_kernel, dim3((unsigned)((Q.n_tiles + 255) / 256)), dim3(256), 0, s, P, Q,
                     (int32_t*)Q.tile_lo);
  const size_t lds = (size_t)kPrWaveBytes * (kPrBlock / 64);
  const int64_t n_blocks = (Q.n_tiles + kPrBlock / 64 - 1) / (kPrBlock / 64);
  const dim3 grid((unsigned)((n_blocks + 7) / 8 * 8)), block(kPrBlock);
  if (P.clip > 0) hipLaunchKernelGGL((per_read_kernel<true>), grid, block, lds, s, P, Q);
  else hipLaunchKernelGGL((per_read_kernel<false>), grid, block, lds, s, P, Q);
}

// ---------------------------------------------------------------- rows -> text
// ConvertPileupRowsToBasestrandTSV (output.go:376-453) for the column set without per-read
// lists: "<chrom>\t<pos+1>\t<REF>\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\n" per BED position.  The
// reference does this serially on the host at the end of the run ("up to ~35%" of its wall
// clock, output.go:143-148); here the text is laid out by a scan of the line lengths and
// written by one thread per row, and only text crosses PCIe (34 bytes per row instead of 44).
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
  return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
       : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ char* put_dec(char* p, uint32_t v, char term) {
  const uint32_t d = dec_digits(v);
  for (uint32_t k = d; k-- > 0;) { p[k] = (char)('0' + v % 10u); v /= 10u; }
  p[d] = term;
  return p + d + 1;
}
// row -> (reference, position)
__device__ __forceinline__ void text_row_pos(const TextParams& T, int64_t R, int32_t& ref, int32_t& pos) {
  int32_t a = 0, b = T.n_iv;  // last interval with iv_row <= R
  while (b - a > 1) {
    const int32_t m = (a + b) >> 1;
    if (T.iv_row[m] <= R) a = m; else b = m;
  }
  ref = T.iv_ref[a];
  pos = T.iv_start[a] + (int32_t)(R - T.iv_row[a]);
}

// The lines of one row.  kind 0: basestrand-tsv (output.go:437-453).  kind 1 / 2: the .ref.tsv /
// .alt.tsv files of the tsv format (output.go:149-276) with the columns of T.cols that are not
// per-read lists: ref line = chrom, pos, REF, [DP], [ref_depth_tier1 = Counts[ref][+]+Counts[ref][-]],
// [ref_depth_tier2 = 0]; one alt line per base A,C,G,T,N other than REF with a nonzero count =
// chrom, pos, REF, ALT, [DP], [alt_depth_tier1], [alt_depth_tier2 = 0].
enum { TEXT_COL_DPREF = 1, TEXT_COL_DPALT = 2, TEXT_COL_HIGHQ = 64, TEXT_COL_LOWQ = 128 };  // pileup.ParseCols bits (common.go:118-166)
__device__ __forceinline__ uint32_t seq8_enum(uint32_t b8) {  // Seq8ToEnumTable (common.go:66)
  return b8 == 1 ? 0u : b8 == 2 ? 1u : b8 == 4 ? 2u : b8 == 8 ? 3u : 4u;
}
__device__ __forceinline__ uint32_t text_row_len(const TextParams& T, int32_t pos, const uint32_t* r, uint32_t b8) {
  const uint32_t head = (uint32_t)T.name_len + 1u + dec_digits((uint32_t)pos + 1u) + 1u + 2u;
  if (T.kind == 0) {
    uint32_t L = head;
#pragma unroll
    for (int c = 1; c <= 8; c++) L += dec_digits(r[c]) + 1u;
    return L;
  }
  const uint32_t rb = seq8_enum(b8);
  const uint32_t lowq = (T.cols & TEXT_COL_LOWQ) ? 2u : 0u;
  if (T.kind == 1) {
    uint32_t L = head + lowq;
    if (T.cols & TEXT_COL_DPREF) L += dec_digits(r[0]) + 1u;
    if (T.cols & TEXT_COL_HIGHQ) L += dec_digits(r[1 + 2 * rb] + r[2 + 2 * rb]) + 1u;
    return L;
  }
  uint32_t L = 0;
  for (uint32_t alt = 0; alt < 5; alt++) {
    const uint32_t c = r[1 + 2 * alt] + r[2 + 2 * alt];
    if (alt == rb || c == 0) continue;
    L += head + 2u + lowq;
    if (T.cols & TEXT_COL_DPALT) L += dec_digits(r[0]) + 1u;
    if (T.cols & TEXT_COL_HIGHQ) L += dec_digits(c) + 1u;
  }
  return L;
}
__device__ __forceinline__ char* text_head(const TextParams& T, char* p, int32_t pos, uint32_t b8) {
  for (int32_t k = 0; k < T.name_len; k++) p[k] = T.name[k];
  p += T.name_len;
  *p++ = '\t';
  p = put_dec(p, (uint32_t)pos + 1u, '\t');
  const char kSeq8[17] = "=ACMGRSVTWYHKDBN";  // Seq8ToASCIITable (pileup/common.go:72)
  *p++ = kSeq8[b8];
  *p++ = '\t';
  return p;
}
__device__ __forceinline__ void text_row_write(const TextParams& T, char* p, int32_t pos, const uint32_t* r, uint32_t b8) {
  if (T.kind == 0) {
    p = text_head(T, p, pos, b8);
#pragma unroll
    for (int c = 1; c <= 8; c++) p = put_dec(p, r[c], c == 8 ? '\n' : '\t');
    return;
  }
  const uint32_t rb = seq8_enum(b8);
  if (T.kind == 1) {
    p = text_head(T, p, pos, b8);
    if (T.cols & TEXT_COL_DPREF) p = put_dec(p, r[0], '\t');
    if (T.cols & TEXT_COL_HIGHQ) p = put_dec(p, r[1 + 2 * rb] + r[2 + 2 * rb], '\t');
    if (T.cols & TEXT_COL_LOWQ) { *p++ = '0'; *p++ = '\t'; }
    p[-1] = '\n';  // EndLine: the last tab becomes the newline
    return;
  }
  const char kEnum[6] = "ACGTN";  // EnumToASCIITable (common.go:69)
  for (uint32_t alt = 0; alt < 5; alt++) {
    const uint32_t c = r[1 + 2 * alt] + r[2 + 2 * alt];
    if (alt == rb || c == 0) continue;
    p = text_head(T, p, pos, b8);
    *p++ = kEnum[alt];
    *p++ = '\t';
    if (T.cols & TEXT_COL_DPALT) p = put_dec(p, r[0], '\t');
    if (T.cols & TEXT_COL_HIGHQ) p = put_dec(p, c, '\t');
    if (T.cols & TEXT_COL_LOWQ) { *p++ = '0'; *p++ = '\t'; }
    p[-1] = '\n';
  }
}

// A row as the text is made from: with T.strand set, one strand's share of a split-strand row -
// what a -per-strand pass of the reference holds for that strand (pileup.go:718-723, :1118-1125:
// only that strand's reads are piled up, format tsv puts every count in slot 0): the strand's
// counts in slot 0, Depth = their sum (exact while no base fails the quality rule: such bases
// count in Depth only and are not in the row; the host asks for this only with min_base_qual 0).
__device__ __forceinline__ void text_load_row(const TextParams& T, int64_t i, uint32_t (&r)[11]) {
  const uint32_t* src = T.rows + i * 11;
#pragma unroll
  for (int k = 0; k < 11; k++) r[k] = src[k];
  if (T.strand) {
    const int s = T.strand - 1;
    uint32_t d = 0;
#pragma unroll
    for (int b = 0; b < 5; b++) {
      const uint32_t c = r[1 + 2 * b + s];
      r[1 + 2 * b] = c;
      r[2 + 2 * b] = 0;
      d += c;
    }
    r[0] = d;
  }
}

// line lengths, scanned inside tiles of 1024 rows
__global__ __launch_bounds__(256) void text_len_kernel(TextParams T) {
  __shared__ uint32_t sh[256];
  const int64_t base = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = 0;
    const int64_t i = base + k;
    if (i < T.n) {
      int32_t ref, pos;
      text_row_pos(T, T.first + i, ref, pos);
      if (ref != T.seq_ref || pos < T.seq_start || (int64_t)pos - T.seq_start >= T.seq_len) *T.bad = 1;
      else {
        uint32_t r[11];
        text_load_row(T, T.first + i, r);
        v[k] = text_row_len(T, pos, r, T.seq8[pos - T.seq_start] & 15u);
      }
    }
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < T.n) T.len[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) T.tile[blockIdx.x] = sh[255];
}

// exclusive scan of the tile totals by one workgroup (8 per thread and step); tile[n_tiles] =
// the grand total
__global__ __launch_bounds__(1024) void text_tile_scan_kernel(unsigned long long* tile, int64_t n_tiles) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t b = 0; b < n_tiles; b += 8192) {
    const int64_t i0 = b + (int64_t)threadIdx.x * 8;
    unsigned long long v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      v[k] = i0 + k < n_tiles ? tile[i0 + k] : 0;
      s += v[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned long long t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    unsigned long long run = carry + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (i0 + k < n_tiles) tile[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile[n_tiles] = carry;
}

__global__ void text_write_kernel(TextParams T) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T.n) return;
  int32_t ref, pos;
  text_row_pos(T, T.first + i, ref, pos);
  if (ref != T.seq_ref || pos < T.seq_start || (int64_t)pos - T.seq_start >= T.seq_len) return;
  uint32_t r[11];
  text_load_row(T, T.first + i, r);
  text_row_write(T, T.out + T.tile[i >> 10] + T.len[i], pos, r, T.seq8[pos - T.seq_start] & 15u);
}

void launch_text(const TextParams& T, hipStream_t s) {
  if (T.n == 0) return;
  const int64_t n_tiles = (T.n + 1023) / 1024;
  hipLaunchKernelGGL(text_len_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, T);
  hipLaunchKernelGGL(text_tile_scan_kernel, dim3(1), dim3(1024), 0, s, T.tile, n_tiles);
}
void launch_text_write(const TextParams& T, hipStream_t s) {
  if (T.n == 0) return;
  hipLaunchKernelGGL(text_write_kernel, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, s, T);
}

// ---- launch wrappers (called from pileup_api.hip)
void launch_plan(const KernelParams& P, int32_t* r0, int32_t* r1, hipStream_t s) {
  if (P.n_windows == 0) return;
  int threads = 256;
  int blocks = (P.n_windows + threads - 1) / threads;
  hipLaunchKernelGGL(plan_windows_kernel, dim3(blocks), dim3(threads), 0, s, P, r0, r1);
}

template <bool DEEP>
static void launch_pileup_variant(const KernelParams& P, dim3 g, hipStream_t s) {
  const dim3 b(kBlock);
  if (P.stitch) {
    // pair rule, quality rule and clip are in the payloads' pass and drop bits (stitch_resolve_kernel)
    hipLaunchKernelGGL((pileup_kernel<true, false, DEEP>), g, b, 0, s, P);
  } else if (P.min_base_qual == 0) {
    // every base passes the quality rule, clipped or not: the quals are never read
    hipLaunchKernelGGL((pileup_kernel<false, true, DEEP>), g, b, 0, s, P);
  } else {
    // quality rule and clip are in the packer's pass bits: one variant
    hipLaunchKernelGGL((pileup_kernel<false, false, DEEP>), g, b, 0, s, P);
  }
}

void launch_pileup(const KernelParams& P, hipStream_t s) {
  if (P.n_windows == 0) return;
  // A workgroup's set-up (zeroing its LDS, priming the window prefetch) is shared by the windows it
  // takes: 24 of them on a long contig (chr1 30x: 4.36 -> 4.31 ms, stitched 6.05 -> 6.00 ms against
  // 8), fewer when that would leave less than ~4 workgroups per resident slot (512 on 256 CUs).
  KernelParams Q = P;
  Q.wins_per_block = std::max(1, std::min(kWinPerBlockMax, P.n_windows / 2048));
  launch_pileup_variant<false>(Q, dim3((unsigned)((P.n_windows + Q.wins_per_block - 1) / Q.wins_per_block)), s);
  // extra slices of deep windows; blocks beyond the item count (read on the device) exit at once
  if (P.extra_items) launch_pileup_variant<true>(Q, dim3((unsigned)std::min(P.extra_cap, 2048)), s);
}

__global__ void rows_add_kernel(uint32_t* rows, const uint32_t* src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    rows[i] += src[i];
}

void launch_rows_add(uint32_t* rows, const uint32_t* src, int64_t n_u32, hipStream_t s) {
  const int64_t blocks = (n_u32 + 255) / 256;
  rows_add_kernel<<<dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s>>>(rows, src, n_u32);
}

void launch_column_sums(const uint32_t* rows, int64_t n_rows, unsigned long long* sums, hipStream_t s) {
  if (n_rows == 0) return;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(column_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, n_rows, sums);
}

int window_size() { return kW; }

}  // namespace bp
