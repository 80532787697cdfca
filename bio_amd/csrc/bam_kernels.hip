// bam_kernels.hip -- the BAM record scan on the device (second half).
//
// What bam.Reader.Read does behind bamprovider's iterator (encoding/bamprovider/bamprovider.go:
// 417-435; record layout as gbam.Unmarshal decodes it, encoding/bam/unmarshal.go:126-224): find
// the records of the inflated stream and read their fixed fields.  The stream was inflated on the
// device (inflate_kernels.hip) and stays there; only 45 bytes per record - the columns the
// host's pairing, filtering and layout passes read - travel to the host, instead of the ~270
// bytes of the record itself (which went to the host and back again before).
//
// The records are a chain (a record's block_size says where the next one starts).  As on the host
// (BamReader::NextChunk) the stream is cut into stretches of 64 KiB; a wave GUESSES where a record
// starts behind its stretch's cut - the first offset at which a record and the two behind it look
// like records, 64 candidate offsets at a time - and walks the chain from there to the next cut.
// The guesses are then verified in order by one wave: stretch j+1's start is right iff it is
// where stretch j's walk - right by induction from the stream's known first record - came out.
// A wrong or missing guess fails the whole call (status), and the caller falls back to the host's
// walk for that batch: never a wrong result.  A third kernel reads the fixed fields of every
// record found (lane = record) and its CIGAR's reference span into columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "bam_device.h"

namespace bp {

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t* base, uint64_t p) {  // unaligned, little-endian
  const uint32_t* a = (const uint32_t*)(base + (p & ~(uint64_t)3));
  return __builtin_amdgcn_alignbyte(a[1], a[0], (uint32_t)(p & 3));
}

// BamReader's Plausible(): could a record start at p?
__device__ __forceinline__ bool plausible(const uint8_t* base, uint64_t p, uint64_t end, int32_t n_ref) {
  if (end - p < 36) return false;
  const uint32_t bs = ld32(base, p), w3 = ld32(base, p + 12), w4 = ld32(base, p + 16), l_seq = ld32(base, p + 20);
  const int32_t ref = (int32_t)ld32(base, p + 4), pos = (int32_t)ld32(base, p + 8);
  const int32_t mref = (int32_t)ld32(base, p + 24), mpos = (int32_t)ld32(base, p + 28);
  const uint32_t l_name = w3 & 0xffu, n_cigar = w4 & 0xffffu;
  if (bs < 32 || bs > (1u << 26)) return false;
  // (the mate's reference only has to be an id at all: files with mates on references the header
  // does not list exist, and a guess that misses a true record costs a serial re-walk)
  if (ref < -1 || ref >= n_ref || mref < -1 || pos < -1 || mpos < -1 || l_name == 0) return false;
  const uint64_t need = 32 + (uint64_t)l_name + 4 * (uint64_t)n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (need > bs) return false;
  const uint64_t nul = p + 36 + l_name - 1;
  return nul >= end || base[nul] == 0;
}

}  // namespace

// One wave per stretch.  lo: the first record of the stream (known); stretch j covers
// [lo + j * kStretch, lo + (j + 1) * kStretch) clipped to end.
__global__ __launch_bounds__(256) void bam_walk_kernel(const uint8_t* __restrict__ buf, uint64_t lo, uint64_t end,
                                                        int32_t n_ref, int32_t n_stretch, WalkStretch* __restrict__ ws,
                                                        uint32_t* __restrict__ offs) {
  const int lane = (int)(threadIdx.x & 63);
  const int j = (int)(blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  if (j >= n_stretch) return;
  const uint64_t cut0 = lo + (uint64_t)j * kStretch, cut1 = min(end, cut0 + kStretch);
  uint64_t p = cut0;
  bool found = true;
  if (j > 0) {
    // A record start lies within one record length of the cut: look as far as the host does.  A
    // candidate whose successors can be checked (they lie inside the stream) and look like records
    // is taken at once; one that merely looks like a record itself but reaches past the end of the
    // stream - as the bytes just in front of a real record often do, read as a record with a huge
    // block size - is only taken when no checked candidate follows.
    found = false;
    bool weak_found = false;
    uint64_t weak = 0;
    const uint64_t upto = min(end, cut0 + (64u << 10));
    for (uint64_t q0 = cut0; q0 < upto && !found; q0 += 64) {
      const uint64_t q = q0 + (uint64_t)lane;
      bool ok = q < upto && plausible(buf, q, end, n_ref);
      bool strong = false;
      if (ok) {
        const uint64_t q2 = q + 4 + ld32(buf, q);
        if (q2 + 36 <= end) {
          ok = plausible(buf, q2, end, n_ref);
          strong = ok;
          if (ok) {
            const uint64_t q3 = q2 + 4 + ld32(buf, q2);
            if (q3 + 36 <= end) ok = strong = plausible(buf, q3, end, n_ref);
          }
        }
      }
      const unsigned long long ms = __ballot(ok && strong), mw = __ballot(ok && !strong);
      if (ms) {
        found = true;
        p = q0 + (uint64_t)__builtin_ctzll(ms);
      } else if (mw && !weak_found) {
        weak_found = true;
        weak = q0 + (uint64_t)__builtin_ctzll(mw);
      }
    }
    if (!found && weak_found) {
      found = true;
      p = weak;
    }
  }
  uint32_t n = 0, flags = 0;
  uint32_t* const mine = offs + (size_t)j * kMaxPerStretch;
  const uint64_t start = p;
  if (found) {
    // (wave-uniform walk; lane 0 writes)
    while (p < cut1) {
      if (end - p < 4) { flags |= WALK_STOPPED; break; }
      const uint32_t bs = ld32(buf, p);
      if (bs < 32 || end - p - 4 < bs) { flags |= WALK_STOPPED; break; }
      if (n >= (uint32_t)kMaxPerStretch) { flags |= WALK_OVERFLOW; break; }
      if (lane == 0) mine[n] = (uint32_t)p;
      n++;
      p += 4 + (uint64_t)bs;
    }
  } else {
    flags |= WALK_NO_START;
  }
  if (lane == 0) {
    WalkStretch w;
    w.start = found ? (uint32_t)start : 0xffffffffu;
    w.exit = (uint32_t)min(p, (uint64_t)0xffffffffu);
    w.n = n;
    w.flags = flags;
    ws[j] = w;
  }
}

// One wave: the guesses checked in order, the stretches' record counts scanned.  The stretches
// come through LDS a tile at a time (all lanes load and store, lane 0 walks the tile).
// out[0] = status (0 ok, else 1 + the stretch whose guess was wrong or missing: the caller walks
// this batch on the host), out[1] = number of records, out[2] = where the walk came out (the first
// record start not taken), out[3] = 1 when it stopped at a record cut off by the end of the stream.
__global__ __launch_bounds__(64) void bam_verify_kernel(const uint8_t* __restrict__ buf, uint64_t lo, uint64_t end,
                                                         int32_t n_stretch, WalkStretch* __restrict__ ws,
                                                         uint32_t* __restrict__ offs, uint32_t* __restrict__ out) {
  constexpr int kTile = 512;
  __shared__ uint32_t t[kTile * 5];
  __shared__ uint32_t st[5];  // status, total, stopped, done, stretches walked again
  __shared__ uint64_t at_sh;
  const int lane = (int)threadIdx.x;
  if (lane == 0) { st[0] = 0; st[1] = 0; st[2] = 0; st[3] = 0; st[4] = 0; at_sh = lo; }
  __syncthreads();
  uint32_t* const g = (uint32_t*)ws;
  for (int j0 = 0; j0 < n_stretch; j0 += kTile) {
    const int m = min(kTile, n_stretch - j0);
    for (int k = lane; k < m * 5; k += 64) t[k] = g[(size_t)j0 * 5 + k];
    __syncthreads();
    if (lane == 0) {
      uint64_t at = at_sh;
      uint32_t total = st[1];
      for (int k = 0; k < m; k++) {
        uint32_t* w = t + 5 * k;  // start, exit, n, flags, base
        w[4] = total;
        if (st[3]) { w[3] |= WALK_SKIPPED; continue; }
        const uint64_t cut1 = min(end, lo + (uint64_t)(j0 + k + 1) * kStretch);
        if (at >= cut1) { w[3] |= WALK_SKIPPED; continue; }  // a long record carried the chain past this stretch
        if (w[3] & WALK_OVERFLOW) {
          st[0] = 1u + (uint32_t)(j0 + k);
          st[3] = 1;
          w[3] |= WALK_SKIPPED;
          continue;
        }
        if ((w[3] & WALK_NO_START) || (uint64_t)w[0] != at) {
          // a guess that was wrong or missing: this stretch is walked again from where the chain
          // stands, by this one lane (as the host's reader does; rare)
          uint32_t* const mine = offs + (size_t)(j0 + k) * kMaxPerStretch;
          uint64_t p = at;
          uint32_t n = 0, fl = 0;
          while (p < cut1) {
            if (end - p < 4) { fl |= WALK_STOPPED; break; }
            const uint32_t bs = ld32(buf, p);
            if (bs < 32 || end - p - 4 < bs) { fl |= WALK_STOPPED; break; }
            if (n >= (uint32_t)kMaxPerStretch) { fl |= WALK_OVERFLOW; break; }
            mine[n++] = (uint32_t)p;
            p += 4 + (uint64_t)bs;
          }
          if (fl & WALK_OVERFLOW) {
            st[0] = 1u + (uint32_t)(j0 + k);
            st[3] = 1;
            w[3] |= WALK_SKIPPED;
            continue;
          }
          w[0] = (uint32_t)at;
          w[1] = (uint32_t)p;
          w[2] = n;
          w[3] = fl | WALK_REWALKED;
          st[4]++;
        }
        total += w[2];
        at = w[1];
        if (w[3] & WALK_STOPPED) { st[2] = 1; st[3] = 1; }
      }
      at_sh = at;
      st[1] = total;
    }
    __syncthreads();
    for (int k = lane; k < m * 5; k += 64) g[(size_t)j0 * 5 + k] = t[k];
    __syncthreads();
  }
  if (lane == 0) {
    out[0] = st[0];
    out[1] = st[1];
    out[2] = (uint32_t)at_sh;
    out[3] = st[2];
    out[4] = st[4];
  }
}

// Lane = record: the fixed fields (gbam.Unmarshal's, encoding/bam/unmarshal.go:126-224) and the
// CIGAR's reference span (sam.Cigar.Lengths: M, D, N, =, X) into columns.
__global__ __launch_bounds__(256) void bam_fields_kernel(const uint8_t* __restrict__ buf, int32_t n_stretch,
                                                          const WalkStretch* __restrict__ ws,
                                                          const uint32_t* __restrict__ offs, BamColumns C) {
  const int lane = (int)(threadIdx.x & 63);
  const int j = (int)(blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  if (j >= n_stretch) return;
  const WalkStretch w = ws[j];
  if (w.flags & WALK_SKIPPED) return;
  const uint32_t* const mine = offs + (size_t)j * kMaxPerStretch;
  for (uint32_t i0 = 0; i0 < w.n; i0 += 64) {
    const uint32_t i = i0 + (uint32_t)lane;
    if (i >= w.n) continue;
    const uint64_t p = mine[i];
    const size_t o = (size_t)w.base + i;
    const uint32_t bs = ld32(buf, p), w3 = ld32(buf, p + 12), w4 = ld32(buf, p + 16), l_seq = ld32(buf, p + 20);
    const uint32_t l_name = w3 & 0xffu, n_cigar = w4 & 0xffffu;
    const uint64_t need = 32 + (uint64_t)l_name + 4 * (uint64_t)n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
    uint8_t facts = need <= bs ? 0 : 4;  // bit 2: the fields do not fit the block
    int64_t span = 0;
    if (!facts) {
      bool bad = false;
      uint32_t first_op = 1;
      for (uint32_t k = 0; k < n_cigar; k++) {
        const uint32_t v = ld32(buf, p + 36 + l_name + 4 * (uint64_t)k), op = v & 15u, len = v >> 4;
        if (k == 0) first_op = op;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += len;
        if (op > 5) bad = true;
      }
      if (n_cigar == 1 && first_op == 0) facts |= 1;
      if (bad) facts |= 2;
    }
    C.off[o] = (uint32_t)p;
    C.ref[o] = (int32_t)ld32(buf, p + 4);
    C.pos[o] = (int32_t)ld32(buf, p + 8);
    C.mref[o] = (int32_t)ld32(buf, p + 24);
    C.mpos[o] = (int32_t)ld32(buf, p + 28);
    C.flags[o] = (uint16_t)(w4 >> 16);
    C.mapq[o] = (uint8_t)((w3 >> 8) & 0xffu);
    C.lname[o] = (uint8_t)l_name;
    C.ncig[o] = n_cigar;
    C.lseq[o] = l_seq;
    C.bytes[o] = 4 + bs;
    C.span[o] = span > 0x7fffffff ? 0x7fffffff : (int32_t)span;
    C.facts[o] = facts;
  }
}

// Read names, gathered: record i's l_read_name - 1 bytes (the NUL left out) to out + name_off[i].
// Thread = record; a name is a few dozen bytes.
__global__ __launch_bounds__(256) void bam_names_kernel(const uint8_t* __restrict__ buf, int64_t n,
                                                         const uint32_t* __restrict__ off,
                                                         const uint32_t* __restrict__ name_off,
                                                         uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* const s = buf + (size_t)off[i] + 36;
  const uint32_t a = name_off[i], len = name_off[i + 1] - a;
  for (uint32_t k = 0; k < len; k++) out[a + k] = s[k];
}

void launch_bam_names(const uint8_t* buf, int64_t n, const uint32_t* off, const uint32_t* name_off, uint8_t* out,
                      hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(bam_names_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, n, off, name_off, out);
}

void launch_bam_scan(const uint8_t* buf, uint64_t lo, uint64_t end, int32_t n_ref, int32_t n_stretch, WalkStretch* ws,
                     uint32_t* offs, uint32_t* out4, hipStream_t s) {
  if (n_stretch <= 0) return;
  hipLaunchKernelGGL(bam_walk_kernel, dim3((unsigned)((n_stretch + 3) / 4)), dim3(256), 0, s, buf, lo, end, n_ref, n_stretch,
                     ws, offs);
  hipLaunchKernelGGL(bam_verify_kernel, dim3(1), dim3(64), 0, s, buf, lo, end, n_stretch, ws, offs, out4);
}

void launch_bam_fields(const uint8_t* buf, int32_t n_stretch, const WalkStretch* ws, const uint32_t* offs,
                       const BamColumns& C, hipStream_t s) {
  if (n_stretch <= 0) return;
  hipLaunchKernelGGL(bam_fields_kernel, dim3((unsigned)((n_stretch + 3) / 4)), dim3(256), 0, s, buf, n_stretch, ws, offs, C);
}

}  // namespace bp
