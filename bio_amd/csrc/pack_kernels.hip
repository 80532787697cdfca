// pack_kernels.hip -- records of an inflated BAM stream -> ReadCore + payload, on the device.
//
// What gbam.Unmarshal (encoding/bam/unmarshal.go:126-224) does for the fields the pileup reads,
// fused with the packing bp_submit does on host threads: the host only finds the records and
// runs its scan over their fixed fields; the bytes go to the GPU as they lie in the stream and
// one wave per record rewrites them into the layout the pileup kernel loads (pileup_device.h):
// CIGAR padded to an even op count, bases with the first base of a byte in the LOW nibble,
// quals, every part padded with zeros to whole 8-byte units.
//
// HBM-bound byte shuffling: 2 x record bytes per record.  Source bytes have no alignment (a BAM
// record starts wherever the last one ended): a lane fetches the two aligned dwords around the four
// bytes it wants and realigns them in registers (v_alignbyte); the stores are aligned dwords.  A
// lane never reads past the aligned dword that holds the last byte it needs, so nothing behind the
// end of the stream's last record is touched beyond that dword.
#include <hip/hip_runtime.h>

#include "pack_device.h"
#include "pileup_device.h"

namespace bp {

namespace {

constexpr int kPackBlock = 256;  // 16 lanes to a record (a 150-base record is 25 + 38 dwords): 16 records per workgroup; measured
                                 // on 3.9 M records of a 30x BAM: 64 lanes 0.63 ms per 2^20 records, 32 lanes 0.40, 16 lanes 0.28, 8 lanes 0.27
constexpr int kPackLanes = 16;

// bytes p[0..n) (n <= 4) as a little-endian word, the rest zero; p has any alignment
__device__ __forceinline__ uint32_t ld_bytes(const uint8_t* p, uint32_t n) {
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* const q = (const uint32_t*)(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)a & 3u;
  const uint32_t lo = q[0];
  const uint32_t hi = sh + n > 4 ? q[1] : 0u;
  const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sh);
  return n >= 4 ? v : v & ((1u << (8u * n)) - 1u);
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return ld_bytes(p, 4); }

__global__ __launch_bounds__(kPackBlock) void pack_records(const uint8_t* __restrict__ raw,
                                                           const PackItem* __restrict__ items, int64_t n,
                                                           ReadCore* __restrict__ cores, uint8_t* __restrict__ blob,
                                                           int qpass, int min_base_qual, int clip) {
  const int64_t k = (int64_t)blockIdx.x * (kPackBlock / kPackLanes) + (threadIdx.x / kPackLanes);
  if (k >= n) return;  // (uniform over the record's lanes, which are all the exchanges below involve)
  const uint32_t lane = threadIdx.x & (kPackLanes - 1);
  const PackItem it = items[k];
  const uint8_t* const p = raw + it.src + 4;  // the fixed block behind block_size
  const uint32_t w12 = ld_u32(p + 12);        // n_cigar | flag << 16
  const uint32_t n_cigar = w12 & 0xffffu;
  const uint32_t l_seq = ld_u32(p + 16);
  const uint32_t l = l_seq > 0xffffu ? 0u : l_seq;  // such a record passes no max-read-len: no payload kept
  const uint8_t* const cig = p + 32 + it.l_name;
  const uint8_t* const seq = cig + 4 * (size_t)n_cigar;
  const uint8_t* const qual = seq + ((size_t)l_seq + 1) / 2;
  if (lane < 8) {
    // ReadCore, one dword a lane
    uint32_t v;
    switch (lane) {
      case 0: v = ld_u32(p + 4); break;                                                            // pos
      case 1: v = (w12 >> 16) | (((ld_u32(p + 8) >> 8) & 0xffu) << 16) | ((uint32_t)it.aux << 24); break;  // flags, mapq, aux
      case 2: v = n_cigar | ((uint32_t)it.span << 16); break;
      case 3: v = (l_seq > 0xffffu ? 0xffffu : l_seq) | ((uint32_t)it.qdelta << 16); break;
      case 4: v = it.blob; break;
      default: v = lane - 5 < n_cigar ? ld_u32(cig + 4 * (lane - 5)) : 0u; break;                   // cig[0..2]
    }
    ((uint32_t*)(cores + k))[lane] = v;
  }
  const uint32_t ncp = (n_cigar + 1) & ~1u;
  const uint32_t U = (l + 7) >> 3;  // blob_units()
  const uint32_t sb = (l + 1) >> 1;
  // pass bits, one per base, to a whole dword; qpass == 2 (stitch): each followed by a dword of
  // drop bits, zero here
  const uint32_t qp_dw = qpass ? (uint32_t)qpass * ((U + 3) >> 2) : 0u;
  const uint32_t base_dw = (ncp + U + qp_dw + 1) & ~1u;   // base_part_bytes() / 4, to whole 8-byte units
  uint32_t* const d = (uint32_t*)(blob + (size_t)it.blob * 8);
  for (uint32_t w = lane; w < base_dw; w += kPackLanes) {
    uint32_t v = 0;
    if (w < n_cigar) {
      v = ld_u32(cig + 4 * (size_t)w);
    } else if (w >= ncp && w - ncp < U) {
      const uint32_t b0 = 4 * (w - ncp);
      v = b0 < sb ? ld_bytes(seq + b0, sb - b0 < 4 ? sb - b0 : 4) : 0u;
      v = ((v & 0x0f0f0f0fu) << 4) | ((v >> 4) & 0x0f0f0f0fu);  // first base of a byte to the low nibble
    } else if (w >= ncp + U && w - (ncp + U) < qp_dw && (qpass != 2 || ((w - (ncp + U)) & 1u) == 0)) {
      continue;  // a dword of pass bits: written below, from the quals
    }
    d[w] = v;
  }
  // quals, a dword (4 bases) a lane; and their pass bits: bit b of pass dword t = base 32t + b passes
  // the quality rule (clipQuals, pileup_support.go:51-69, then addBase's test, pileup.go:227).  The 8
  // lanes that hold the 32 quals of one pass dword put their 4 bits together with three exchanges.
  uint32_t* const dq = d + 2 * (size_t)it.qdelta;
  uint32_t* const dp = d + ncp + U;
  const bool all_clipped = 2 * (uint32_t)clip >= l;
  for (uint32_t w0 = 0; w0 < 2 * U; w0 += kPackLanes) {
    const uint32_t w = w0 + lane, b0 = 4 * w;
    const bool have = w < 2 * U;
    const uint32_t v = have && b0 < l ? ld_bytes(qual + b0, l - b0 < 4 ? l - b0 : 4) : 0u;
    if (have) dq[w] = v;
    if (qpass) {
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t x = 0; x < 4; x++) {
        const uint32_t i = b0 + x;
        uint32_t q = (v >> (8 * x)) & 0xffu;
        if (all_clipped || i < (uint32_t)clip || i >= l - (uint32_t)clip) q = 2;
        bits |= (i < l && q >= (uint32_t)min_base_qual ? 1u : 0u) << x;
      }
      bits <<= 4 * (lane & 7u);
      bits |= (uint32_t)__shfl_xor((int)bits, 1);
      bits |= (uint32_t)__shfl_xor((int)bits, 2);
      bits |= (uint32_t)__shfl_xor((int)bits, 4);
      if (have && (lane & 7u) == 0) dp[(w >> 3) * (qpass == 2 ? 2u : 1u)] = bits;
    }
  }
}

}  // namespace

void launch_pack(const uint8_t* raw, const PackItem* items, int64_t n, ReadCore* cores, uint8_t* blob, int qpass,
                 int min_base_qual, int clip, hipStream_t s) {
  if (n <= 0) return;
  const int64_t per = kPackBlock / kPackLanes;
  hipLaunchKernelGGL(pack_records, dim3((unsigned)((n + per - 1) / per)), dim3(kPackBlock), 0, s, raw, items, n,
                     cores, blob, qpass, min_base_qual, clip);
}

}  // namespace bp
