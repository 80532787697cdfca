// crc_device.h -- CRC-32 (IEEE, reflected, as in gzip) of a byte range by one wave.
// T = the byte table; W[q][b] = "advance (b << 8q) by 256 zero bytes" (inflate_api.hip, crc_tables:
// tables[0..255] = T, tables[256 + 256 q + b] = W[q][b]); both in LDS.  Every lane returns the CRC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bp {

__device__ __forceinline__ uint32_t wave_crc32(const uint8_t* base, uint32_t n, const uint32_t* T,
                                               const uint32_t (*W)[256], int lane) {
  auto adv1 = [&](uint32_t v) { return T[v & 0xffu] ^ (v >> 8); };
  if (n < 4) {
    uint32_t crc = 0xffffffffu;
    for (uint32_t k = 0; k < n; k++) crc = adv1(crc ^ base[k]);
    return ~crc;
  }
  // the range as a message with h zero bytes in front, so that its dwords are aligned ones
  const uint32_t h = (uint32_t)((uintptr_t)base & 3);
  const uint32_t* a = (const uint32_t*)(base - h);
  const uint32_t np = h + n, ndw = np >> 2, tail = np & 3;
  uint32_t s = 0;
  int last = -1;
  for (uint32_t k = (uint32_t)lane; k < ndw; k += 64) {
    uint32_t d = a[k];
    // the initial state 0xffffffff is the first four bytes of the message inverted
    if (k == 0) d = (d & (0xffffffffu << (8 * h))) ^ (0xffffffffu << (8 * h));
    if (k == 1 && h) d ^= 0xffffffffu >> (32 - 8 * h);
    const uint32_t v = s ^ d;
    if (k + 64 < ndw) {
      s = W[0][v & 0xffu] ^ W[1][(v >> 8) & 0xffu] ^ W[2][(v >> 16) & 0xffu] ^ W[3][v >> 24];
    } else {
      s = adv1(adv1(adv1(adv1(v))));
      last = (int)k;
    }
  }
  if (last >= 0) {
    const uint32_t m = (ndw - 1 - (uint32_t)last) * 4 + tail;  // bytes behind this lane's last dword
    for (uint32_t k = 0; k < m; k++) s = adv1(s);
  }
  if (lane == 63 && tail) {
    uint32_t t = 0;
    const uint8_t* tp = (const uint8_t*)(a + ndw);
    // (when the message has fewer than two dwords the inverted prefix reaches into the tail)
    for (uint32_t k = 0; k < tail; k++) {
      uint32_t byte = tp[k];
      if (ndw == 1 && h) byte ^= (0xffffffffu >> (32 - 8 * h) >> (8 * k)) & 0xffu;
      t = adv1(t ^ byte);
    }
    s ^= t;
  }
#pragma unroll
  for (int d = 32; d; d >>= 1) s ^= (uint32_t)__shfl_xor((int)s, d, 64);
  return ~s;
}

}  // namespace bp
