// pack_kernels.hip -- records of an inflated BAM stream -> ReadCore + payload, on the device.
//
// What gbam.Unmarshal (encoding/bam/unmarshal.go:126-224) does for the fields the pileup reads,
// fused with the packing bp_submit does on host threads: the host only finds the records and
// runs its scan over their fixed fields; the bytes go to the GPU as they lie in the stream and
// one wave per record rewrites them into the layout the pileup kernel loads (pileup_device.h):
// CIGAR padded to an even op count, bases with the first base of a byte in the LOW nibble,
// quals, every part padded with zeros to whole 8-byte units.
//
// HBM-bound byte shuffling: 2 x record bytes per record, a few tens of microseconds per 16 MiB.
// Source bytes have no alignment (a BAM record starts wherever the last one ended), so they are
// loaded as bytes; the stores are aligned dwords.
#include <hip/hip_runtime.h>

#include "pack_device.h"
#include "pileup_device.h"

namespace bp {

namespace {

constexpr int kPackBlock = 256;  // 4 waves = 4 records per workgroup

__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// up to four bytes p[0..n), the rest zero
__device__ __forceinline__ uint32_t ld_upto4(const uint8_t* p, uint32_t n) {
  uint32_t v = 0;
  if (n > 0) v |= (uint32_t)p[0];
  if (n > 1) v |= (uint32_t)p[1] << 8;
  if (n > 2) v |= (uint32_t)p[2] << 16;
  if (n > 3) v |= (uint32_t)p[3] << 24;
  return v;
}

__global__ __launch_bounds__(kPackBlock) void pack_records(const uint8_t* __restrict__ raw,
                                                           const PackItem* __restrict__ items, int64_t n,
                                                           ReadCore* __restrict__ cores, uint8_t* __restrict__ blob,
                                                           int qpass, int min_base_qual, int clip) {
  const int64_t k = (int64_t)blockIdx.x * (kPackBlock / 64) + (threadIdx.x >> 6);
  if (k >= n) return;
  const uint32_t lane = threadIdx.x & 63u;
  const PackItem it = items[k];
  const uint8_t* const p = raw + it.src + 4;  // the fixed block behind block_size
  const uint32_t n_cigar = ld_u16(p + 12);
  const uint32_t l_seq = ld_u32(p + 16);
  const uint32_t l = l_seq > 0xffffu ? 0u : l_seq;  // such a record passes no max-read-len: no payload kept
  const uint8_t* const cig = p + 32 + it.l_name;
  const uint8_t* const seq = cig + 4 * (size_t)n_cigar;
  const uint8_t* const qual = seq + ((size_t)l_seq + 1) / 2;
  if (lane < 8) {
    // ReadCore, one dword a lane
    uint32_t v;
    switch (lane) {
      case 0: v = ld_u32(p + 4); break;                                                      // pos
      case 1: v = ld_u16(p + 14) | ((uint32_t)p[9] << 16) | ((uint32_t)it.aux << 24); break;  // flags, mapq, aux
      case 2: v = n_cigar | ((uint32_t)it.span << 16); break;
      case 3: v = (l_seq > 0xffffu ? 0xffffu : l_seq) | ((uint32_t)it.qdelta << 16); break;
      case 4: v = it.blob; break;
      default: v = lane - 5 < n_cigar ? ld_u32(cig + 4 * (lane - 5)) : 0u; break;             // cig[0..2]
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
  for (uint32_t w = lane; w < base_dw; w += 64) {
    uint32_t v = 0;
    if (w < n_cigar) {
      v = ld_u32(cig + 4 * (size_t)w);
    } else if (w >= ncp && w - ncp < U) {
      const uint32_t b0 = 4 * (w - ncp);
      v = b0 < sb ? ld_upto4(seq + b0, sb - b0) : 0u;
      v = ((v & 0x0f0f0f0fu) << 4) | ((v >> 4) & 0x0f0f0f0fu);  // first base of a byte to the low nibble
    } else if (w >= ncp + U && w - (ncp + U) < qp_dw && (qpass != 2 || ((w - (ncp + U)) & 1u) == 0)) {
      // bit b of the dword: base 32t + b passes the quality rule (clipQuals, then addBase's test)
      const uint32_t b0 = 32 * ((w - (ncp + U)) >> (qpass == 2 ? 1 : 0));
      for (uint32_t b = 0; b < 32 && b0 + b < l; b++) {
        const uint32_t i = b0 + b;
        uint32_t q = qual[i];
        if (2 * (uint32_t)clip >= l || i < (uint32_t)clip || i >= l - (uint32_t)clip) q = 2;
        v |= (q >= (uint32_t)min_base_qual ? 1u : 0u) << b;
      }
    }
    d[w] = v;
  }
  uint32_t* const dq = d + 2 * (size_t)it.qdelta;
  for (uint32_t w = lane; w < 2 * U; w += 64) {
    const uint32_t b0 = 4 * w;
    dq[w] = b0 < l ? ld_upto4(qual + b0, l - b0) : 0u;
  }
}

}  // namespace

void launch_pack(const uint8_t* raw, const PackItem* items, int64_t n, ReadCore* cores, uint8_t* blob, int qpass,
                 int min_base_qual, int clip, hipStream_t s) {
  if (n <= 0) return;
  const int64_t per = kPackBlock / 64;
  hipLaunchKernelGGL(pack_records, dim3((unsigned)((n + per - 1) / per)), dim3(kPackBlock), 0, s, raw, items, n,
                     cores, blob, qpass, min_base_qual, clip);
}

}  // namespace bp
