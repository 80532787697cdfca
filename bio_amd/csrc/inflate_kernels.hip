// inflate_kernels.hip -- BGZF blocks inflated on the device.
//
// What runs in front of processShard in the reference: bamprovider's iterator reads records from
// hts' bam.Reader over its bgzf.Reader (encoding/bamprovider/bamprovider.go:417-435; biogo/hts
// bgzf, a go.mod dependency absent from the tree), which inflates one BGZF block after the
// other with compress/flate and checks each block's CRC-32 and ISIZE.  BGZF (SAM spec 4.1) and
// DEFLATE (RFC 1951) are public formats; zlib is the byte-exact checker in the tests.
//
// A BGZF block is at most 64 KiB of output and independent of every other block, so blocks are the
// parallel axis: one WAVE inflates one block.  Huffman decoding is a serial walk of a bit stream;
// the wave does it as wave-uniform code (bit buffer, counters and table entries live in scalar
// registers; the compressed bytes come in 256 at a time, one dword per lane, and are picked out
// with v_readlane) and leaves up to 64 decoded tokens - a literal, or a (length, distance) match -
// one per lane.  The LZ77 side is then lane-parallel: a prefix sum of the token lengths places
// every token in the output, and each lane produces output BYTES: it finds the token that covers
// its byte, and follows match sources back (a source inside the same batch of tokens is itself
// the output of an earlier token) until it arrives at a literal or at a byte the wave wrote in an
// earlier batch, which it loads from global memory.  So a batch costs one global load and one
// store per 64 output bytes, whatever the matches overlap.  Decode tables (10-bit root for
// literals/lengths, 8-bit for distances, canonical bit-by-bit decode behind them for longer
// codes) live in LDS, 6.7 KB per wave; they are built lane-parallel from the code lengths.
//
// A second kernel checks CRC-32 (one wave per block, lanes striding over the block's dwords with a
// table for "advance by 256 zero bytes", combined at the end) and marks blocks whose checksum
// differs: the reference's reader refuses such a block too.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "crc_device.h"
#include "inflate_device.h"

namespace bp {

namespace {

constexpr int kLitRoot = 10, kDistRoot = 8, kClRoot = 7;
constexpr int kWaves = 4;  // per workgroup; each wave works alone

struct WaveLds {
  uint32_t lit_tab[1 << kLitRoot];
  uint32_t dist_tab[1 << kDistRoot];  // also the code-length code's table while a header is read
  uint32_t tok[64];                   // the batch's tokens ...
  int32_t off[64];                    // ... and where each starts in the batch's output
  uint16_t lit_sorted[288];           // symbols in canonical order, for codes longer than the root
  uint16_t dist_sorted[32];
  uint16_t lit_cnt[16];               // codes per length
  uint16_t dist_cnt[16];
  uint8_t lens[320];                  // code lengths of the block being set up
  uint32_t dump[2];                   // where lanes that have nothing to store store (see below)
};

// Lane-dependent BRANCHES are kept out of the serial phases: where the compiler merges the join of
// such a branch with a block that also carries the decoder's (wave-uniform) state, its uniformity
// analysis has to call that state divergent, and from there on the bit buffer lives in vector
// registers and every loop runs under an exec mask.  So lanes with nothing to store store to a
// dump slot (an address select, no branch), and the state is passed through v_readfirstlane
// behind the few lane-parallel regions that do branch (table build, batch write).

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_wave_barrier(); }

// Table entry: bits 0-3 code length (0: not in the root table), 4-5 kind (0 literal, 1 length or
// distance, 2 end of block, 3 invalid symbol), 8-11 extra bits, 16-31 literal / base value.
__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t len) {
  if (sym < 256) return len | (sym << 16);
  if (sym == 256) return len | (2u << 4);
  if (sym > 285) return len | (3u << 4);
  const uint32_t g = sym - 257;
  uint32_t base, xb;
  if (g < 8) { base = 3 + g; xb = 0; }
  else if (g == 28) { base = 258; xb = 0; }
  else { xb = (g >> 2) - 1; base = 3 + ((4 + (g & 3)) << xb); }
  return len | (1u << 4) | (xb << 8) | (base << 16);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
  if (sym > 29) return len | (3u << 4);
  uint32_t base, xb;
  if (sym < 4) { base = 1 + sym; xb = 0; }
  else { xb = (sym >> 1) - 1; base = 1 + ((2 + (sym & 1)) << xb); }
  return len | (1u << 4) | (xb << 8) | (base << 16);
}

// Canonical Huffman tables from code lengths, by the whole wave.  MODE 0 literal/length, 1
// distance, 2 the code-length code.  Returns false for an over-subscribed set of lengths
// (incomplete sets are let through: a code nobody assigned decodes to an error when it comes).
template <int MODE>
__device__ __forceinline__ bool build_table(const uint8_t* lens, int n_sym, uint32_t* tab, uint16_t* sorted, uint16_t* cnt_out,
                                            uint32_t* dump, int lane) {
  constexpr int R = MODE == 0 ? 5 : 1;  // rounds of 64 symbols
  constexpr int ROOT = MODE == 0 ? kLitRoot : MODE == 1 ? kDistRoot : kClRoot;
  uint32_t len[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int s = r * 64 + lane;
    len[r] = s < n_sym ? lens[s] : 0;
  }
  uint32_t mycnt = 0;  // lane L: codes of length L
  for (int L = 1; L <= 15; L++) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < R; r++) c += (uint32_t)__popcll(__ballot(len[r] == (uint32_t)L));
    if (lane == L) mycnt = c;
  }
  uint32_t myfirst = 0, myoffs = 0;  // lane L: first code of length L, and its place in `sorted`
  {
    int32_t left = 1;
    uint32_t code = 0, idx = 0;
    for (int L = 1; L <= 15; L++) {
      const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, L);
      left = (left << 1) - (int32_t)c;
      if (left < 0) return false;
      if (lane == L) { myfirst = code; myoffs = idx; }
      code = (code + c) << 1;
      idx += c;
    }
  }
  for (int k0 = 0; k0 < (1 << ROOT); k0 += 64) tab[k0 + lane] = 0;  // (every loop's trip count is wave-uniform,
  // so that the compiler keeps the decoder's state in scalar registers: a lane-dependent exit makes all of it 'divergent')
  *(lane < 16 ? &cnt_out[lane] : (uint16_t*)dump) = (uint16_t)mycnt;
  lds_fence();
  for (int L = 1; L <= 15; L++) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, L);
    if (c == 0) continue;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)myfirst, L);
    const uint32_t offs = (uint32_t)__builtin_amdgcn_readlane((int)myoffs, L);
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const bool mine = len[r] == (uint32_t)L;
      const unsigned long long m = __ballot(mine);
      if (m) {
        const uint32_t rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t sym = (uint32_t)(r * 64 + lane);
        *(mine ? &sorted[offs + rank] : (uint16_t*)dump) = (uint16_t)sym;
        if (L <= ROOT) {
          const uint32_t rev = __brev(first + rank) >> (32 - L);
          const uint32_t e = MODE == 0 ? lit_entry(sym, (uint32_t)L) : MODE == 1 ? dist_entry(sym, (uint32_t)L)
                                                                                 : ((uint32_t)L | (sym << 16));
          for (uint32_t j = 0; j < (1u << (ROOT - L)); j++) *(mine ? &tab[rev + (j << L)] : dump) = e;
        }
        base += (uint32_t)__popcll(m);
      }
    }
  }
  lds_fence();
  return true;
}

// The compressed bytes of one block as a bit stream; everything but win/nxt is wave-uniform.
struct Bits {
  uint64_t buf;
  int cnt;               // valid bits in buf
  uint32_t win, nxt;     // lane k: dword k of the 256-byte window being read, and of the one behind it
  int idx;               // next dword of win
  const uint32_t* fetch; // the window behind nxt
  int64_t loaded_bits;   // bits moved into buf so far, minus those skipped for alignment
  int64_t base_bits;     // bit offset (from the block's first byte) this reader was seated at

  __device__ __forceinline__ void seat(const uint8_t* data, uint32_t byte_off, int lane) {
    const uint8_t* p = data + byte_off;
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3;
    const uint32_t* w = (const uint32_t*)a;
    win = w[lane];
    nxt = w[64 + lane];
    fetch = w + 128;
    idx = 0;
    buf = 0;
    cnt = 0;
    base_bits = 8 * (int64_t)byte_off;
    loaded_bits = 0;
    refill();
    const int skip = 8 * (int)((uintptr_t)p & 3);
    buf >>= skip;
    cnt -= skip;
    loaded_bits -= skip;
  }
  // at least 32 valid bits afterwards
  __device__ __forceinline__ void refill() {
    if (cnt <= 32) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)win, idx);
      buf |= (uint64_t)d << cnt;
      cnt += 32;
      loaded_bits += 32;
      idx++;
      if (idx == 64) {
        win = nxt;
        nxt = fetch[threadIdx.x & 63];
        fetch += 64;
        idx = 0;
      }
    }
  }
  // back to scalar registers behind a lane-parallel region
  __device__ __forceinline__ void uniform() {
    buf = ((uint64_t)rfl((uint32_t)(buf >> 32)) << 32) | rfl((uint32_t)buf);
    cnt = (int)rfl((uint32_t)cnt);
    idx = (int)rfl((uint32_t)idx);
    const uintptr_t f = (uintptr_t)fetch;
    fetch = (const uint32_t*)(((uintptr_t)rfl((uint32_t)(f >> 32)) << 32) | rfl((uint32_t)f));
    loaded_bits = (int64_t)(((uint64_t)rfl((uint32_t)((uint64_t)loaded_bits >> 32)) << 32) | rfl((uint32_t)loaded_bits));
    base_bits = (int64_t)(((uint64_t)rfl((uint32_t)((uint64_t)base_bits >> 32)) << 32) | rfl((uint32_t)base_bits));
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
  __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
  __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
  __device__ __forceinline__ int64_t consumed() const { return base_bits + loaded_bits - cnt; }
};

// A code that is not in the root table: canonical decode one bit at a time.  Returns the symbol
// and its length, or -1.
__device__ __forceinline__ int slow_decode(uint64_t buf, const uint16_t* cnt, const uint16_t* sorted, int* len_out) {
  uint32_t code = 0, first = 0, index = 0;
  for (int L = 1; L <= 15; L++) {
    code |= (uint32_t)(buf & 1);
    buf >>= 1;
    const uint32_t c = rfl(cnt[L]);
    if ((int32_t)(code - c) < (int32_t)first) {
      *len_out = L;
      return (int)rfl(sorted[index + (code - first)]);
    }
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// The batch's tokens (tok/off in LDS, nt of them, `total` output bytes) written to o[0, total):
// lane = output byte.  o[-k] are bytes of earlier batches of this block.
__device__ __forceinline__ void write_batch(const WaveLds& S, uint8_t* o, int total, int lane) {
  for (int p0 = 0; p0 < total; p0 += 64) {
    const int p = p0 + lane;
    int s = p;
    uint32_t val = 0;
    bool busy = p < total;
    while (__ballot(busy)) {  // (a wave-uniform exit)
      if (busy) {
        // the token that covers s: the last one starting at or before it
        int t = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
          if (S.off[t + step] <= s) t += step;  // (off of unused slots is INT_MAX; t + step <= 63)
        const uint32_t tk = S.tok[t];
        if (tk >> 31) {
          val = tk & 0xffu;
          busy = false;
        } else {
          const int to = S.off[t];
          const int L = (int)((tk >> 16) & 0x1ffu), D = (int)(tk & 0xffffu), i = s - to;
          int j = i;
          if (D < L) {
            if (D == 1) j = 0;
            else {
              const int q = (int)((float)i * __builtin_amdgcn_rcpf((float)D));
              j = i - q * D;
              if (j < 0) j += D;
              else if (j >= D) j -= D;
            }
          }
          s = to - D + j;
          if (s < 0) {
            val = o[s];
            busy = false;
          }
        }
      }
    }
    if (p < total) o[p] = (uint8_t)val;
  }
}

__device__ uint32_t inflate_one(const uint8_t* __restrict__ comp, const InflateBlock blk, uint8_t* __restrict__ out,
                                WaveLds& S, int lane) {
  const uint8_t* data = comp + blk.in_off;
  uint8_t* const obase = out + blk.out_off;
  const int out_len = (int)blk.out_len;
  const int64_t in_bits = 8 * (int64_t)blk.in_len;
  int out_pos = 0;
  Bits B;
  B.seat(data, 0, lane);
  for (;;) {  // deflate blocks
    B.refill();
    const uint32_t bfinal = B.take(1), btype = B.take(2);
    if (btype == 3) return INF_BAD_BLOCK_TYPE;
    if (btype == 0) {
      // stored: to the next byte boundary, LEN, ~LEN, bytes
      B.drop((int)((-B.consumed()) & 7));
      B.refill();
      const uint32_t len = B.take(16);
      B.refill();
      const uint32_t nlen = B.take(16);
      if ((len ^ nlen) != 0xffffu) return INF_BAD_STORED_LEN;
      const int64_t at = B.consumed() >> 3;
      if (at + len > (int64_t)blk.in_len) return INF_INPUT_OVERRUN;
      if (out_pos + (int)len > out_len) return INF_OUTPUT_OVERFLOW;
      // (lanes behind the end copy the last byte once more)
      for (uint32_t i0 = 0; i0 < len; i0 += 64) {
        const uint32_t i = min(i0 + (uint32_t)lane, len - 1);
        obase[out_pos + i] = data[at + i];
      }
      out_pos += (int)len;
      B.seat(data, (uint32_t)(at + len), lane);
    } else {
      int hlit = 288, hdist = 32;
      if (btype == 1) {
        // fixed code (RFC 1951 3.2.6): 288 literal/length codes of 8, 9, 7, 8 bits, 32 distance codes of 5
        for (int s0 = 0; s0 < 320; s0 += 64) {
          const int s = s0 + lane;
          S.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
        }
        lds_fence();
      } else {
        B.refill();
        hlit = (int)B.take(5) + 257;
        hdist = (int)B.take(5) + 1;
        const int hclen = (int)B.take(4) + 4;
        if (hlit > 286 || hdist > 30) return INF_BAD_CODE_LENGTHS;
        // the code-length code: 3 bits each, in the order of RFC 1951 3.2.7
        *(lane < 19 ? &S.lens[lane] : (uint8_t*)&S.dump[0]) = 0;
        lds_fence();
        for (int k = 0; k < hclen; k++) {
          B.refill();
          const uint32_t v = B.take(3);
          // order: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
          const int ord = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 7 - ((k - 5) >> 1) : 8 + ((k - 4) >> 1);
          S.lens[ord] = (uint8_t)v;
        }
        lds_fence();
        if (!build_table<2>(S.lens, 19, S.dist_tab, S.lit_sorted, S.lit_cnt, S.dump, lane)) return INF_BAD_CODE_LENGTHS;
        // the literal/length and distance code lengths, run-length coded
        // (S.lens is reused for the result: the code-length code's own lengths are in its table now)
        const int n_len = hlit + hdist;
        int i = 0;
        uint32_t prev = 0;
        while (i < n_len) {
          B.refill();
          const uint32_t e = rfl(S.dist_tab[B.peek(kClRoot)]);
          if ((e & 15u) == 0) return INF_BAD_CODE_LENGTHS;
          B.drop((int)(e & 15u));
          const uint32_t sym = e >> 16;
          if (sym < 16) {
            S.lens[i++] = (uint8_t)sym;
            prev = sym;
          } else {
            uint32_t rep, v;
            if (sym == 16) {
              if (i == 0) return INF_BAD_CODE_LENGTHS;
              rep = 3 + B.take(2);
              v = prev;
            } else if (sym == 17) {
              rep = 3 + B.take(3);
              v = 0;
              prev = 0;
            } else {
              rep = 11 + B.take(7);
              v = 0;
              prev = 0;
            }
            if (i + (int)rep > n_len) return INF_BAD_CODE_LENGTHS;
            // (at most 138: three stores, each lane its own byte)
#pragma unroll
            for (uint32_t k0 = 0; k0 < 192; k0 += 64) {
              uint8_t* dst = k0 + lane < rep ? &S.lens[i + k0 + lane] : (uint8_t*)&S.dump[0];
              *dst = (uint8_t)v;
            }
            i += (int)rep;
          }
        }
        lds_fence();
        if (rfl(S.lens[256]) == 0) return INF_BAD_CODE_LENGTHS;  // no end-of-block code
      }
      {
        // the distance lengths are taken out before the literal table's build reuses S.lens
        const uint32_t dl = lane < hdist ? S.lens[hlit + lane] : 0;
        if (!build_table<0>(S.lens, hlit, S.lit_tab, S.lit_sorted, S.lit_cnt, S.dump, lane)) return INF_BAD_CODE_LENGTHS;
        *(lane < 32 ? &S.lens[lane] : (uint8_t*)&S.dump[0]) = (uint8_t)dl;
        lds_fence();
        if (!build_table<1>(S.lens, 32, S.dist_tab, S.dist_sorted, S.dist_cnt, S.dump, lane)) return INF_BAD_CODE_LENGTHS;
        B.uniform();
        out_pos = (int)rfl((uint32_t)out_pos);
      }
      // the block's symbols, 64 tokens at a time
      for (;;) {
        uint32_t tok = 0;
        int nt = 0;
        bool eob = false;
        while (nt < 64) {
          B.refill();
          uint32_t e = rfl(S.lit_tab[B.peek(kLitRoot)]);
          if ((e & 15u) == 0) {
            int len = 0;
            const int sym = slow_decode(B.buf, S.lit_cnt, S.lit_sorted, &len);
            if (sym < 0) return INF_BAD_CODE;
            e = lit_entry((uint32_t)sym, (uint32_t)len);
          }
          B.drop((int)(e & 15u));
          const uint32_t kind = (e >> 4) & 3u;
          if (kind == 0) {
            if (lane == nt) tok = 0x80000000u | (e >> 16);
            nt++;
          } else if (kind == 1) {
            const int xb = (int)((e >> 8) & 15u);
            const uint32_t length = (e >> 16) + B.take(xb);
            B.refill();
            uint32_t d = rfl(S.dist_tab[B.peek(kDistRoot)]);
            if ((d & 15u) == 0) {
              int len = 0;
              const int sym = slow_decode(B.buf, S.dist_cnt, S.dist_sorted, &len);
              if (sym < 0) return INF_BAD_CODE;
              d = dist_entry((uint32_t)sym, (uint32_t)len);
            }
            if (((d >> 4) & 3u) != 1) return INF_BAD_CODE;
            B.drop((int)(d & 15u));
            const int dxb = (int)((d >> 8) & 15u);
            const uint32_t dist = (d >> 16) + B.take(dxb);
            if (lane == nt) tok = (length << 16) | dist;
            nt++;
          } else if (kind == 2) {
            eob = true;
            break;
          } else {
            return INF_BAD_CODE;
          }
        }
        if (B.consumed() > in_bits) return INF_INPUT_OVERRUN;
        // place the tokens
        const bool valid = lane < nt;
        const bool is_match = valid && !(tok >> 31);
        const int mlen = valid ? (is_match ? (int)((tok >> 16) & 0x1ffu) : 1) : 0;
        int incl = mlen;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_up(incl, d, 64);
          if (lane >= d) incl += v;
        }
        const int off = incl - mlen;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (out_pos + total > out_len) return INF_OUTPUT_OVERFLOW;
        if (__ballot(is_match && (int)(tok & 0xffffu) > out_pos + off)) return INF_DISTANCE_TOO_FAR;
        S.tok[lane] = tok;
        S.off[lane] = valid ? off : 0x7fffffff;
        lds_fence();
        // bytes of earlier batches must have arrived before they are read back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        write_batch(S, obase + out_pos, total, lane);
        lds_fence();
        B.uniform();
        out_pos = (int)rfl((uint32_t)(out_pos + total));
        if (rfl((uint32_t)eob)) break;
      }
    }
    if (B.consumed() > in_bits) return INF_INPUT_OVERRUN;
    if (bfinal) break;
  }
  if (out_pos != out_len) return INF_OUTPUT_SHORT;
  return INF_OK;
}

}  // namespace

__global__ __launch_bounds__(64 * kWaves) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp,
                                                                    const InflateBlock* __restrict__ blks, int n_blocks,
                                                                    uint8_t* __restrict__ out,
                                                                    uint32_t* __restrict__ status) {
  __shared__ WaveLds lds[kWaves];
  const int lane = (int)(threadIdx.x & 63);
  const int wave = (int)rfl(threadIdx.x >> 6);
  WaveLds& S = lds[wave];
  const int nw = (int)gridDim.x * kWaves;
  for (int b = (int)blockIdx.x * kWaves + wave; b < n_blocks; b += nw) {
    const InflateBlock blk = blks[b];
    const uint32_t st = inflate_one(comp, blk, out, S, lane);
    if (lane == 0) status[b] = st;
  }
}

// ---- CRC-32 (IEEE, reflected, as in gzip) of every block's output -----------------------------
// tables[0..255] = the byte table T; tables[256 + 256 q + b] = "advance (b << 8q) by 256 zero bytes".
__global__ __launch_bounds__(256) void bgzf_crc_kernel(const uint8_t* __restrict__ out,
                                                        const InflateBlock* __restrict__ blks, int n_blocks,
                                                        const uint32_t* __restrict__ tables,
                                                        uint32_t* __restrict__ status) {
  __shared__ uint32_t T[256], W[4][256];
  for (int k = (int)threadIdx.x; k < 256; k += (int)blockDim.x) {
    T[k] = tables[k];
#pragma unroll
    for (int q = 0; q < 4; q++) W[q][k] = tables[256 + 256 * q + k];
  }
  __syncthreads();
  const int lane = (int)(threadIdx.x & 63);
  const int wave = (int)rfl(threadIdx.x >> 6);
  const int nw = (int)gridDim.x * 4;
  for (int b = (int)blockIdx.x * 4 + wave; b < n_blocks; b += nw) {
    const InflateBlock blk = blks[b];
    const uint32_t crc = wave_crc32(out + blk.out_off, blk.out_len, T, W, lane);
    if (lane == 0 && crc != blk.crc32 && status[b] == INF_OK) status[b] = INF_BAD_CRC;
  }
}

void launch_inflate(const uint8_t* comp, const InflateBlock* blks, int n_blocks, uint8_t* out, uint32_t* status,
                    hipStream_t s) {
  if (n_blocks <= 0) return;
  const int grid = (n_blocks + kWaves - 1) / kWaves;
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)std::min(grid, 1 << 16)), dim3(64 * kWaves), 0, s, comp, blks,
                     n_blocks, out, status);
}

void launch_bgzf_crc(const uint8_t* out, const InflateBlock* blks, int n_blocks, const uint32_t* crc_tables,
                     uint32_t* status, hipStream_t s) {
  if (n_blocks <= 0) return;
  hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)std::min((n_blocks + 3) / 4, 1 << 16)), dim3(256), 0, s, out, blks,
                     n_blocks, crc_tables, status);
}

}  // namespace bp
