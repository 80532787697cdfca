// Micro-benchmark: how fast can runs of 6-byte entries be written?  (per-read gather)
// Every wave owns a stretch of 2048 consecutive entries (a tile's lists) and writes it in 32
// rounds of 64 consecutive entries, one per lane.
//   A: three 2-byte stores per entry (what per_read_kernel does)
//   B: one aligned dword + one short per entry, by the slot's parity
//   C: entries staged in LDS, the stretch written by 16-byte stores
//   D: the same bytes as plain 16-byte streaming stores (ceiling)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store6_bench.hip -o gpurun_out/store6_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kRun = 2048;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t* out, int64_t n_runs) {
  __shared__ __align__(16) uint8_t lds[4][kRun * 6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t run = (int64_t)blockIdx.x * 4 + wave;
  if (run >= n_runs) return;
  uint8_t* const g = out + run * kRun * 6;
  if (MODE == 3) {
    for (int i = lane; i < kRun * 6 / 16; i += 64) {
      const uint32_t v = (uint32_t)run + i;
      ((uint4*)g)[i] = make_uint4(v, v + 1, v + 2, v + 3);
    }
    return;
  }
  for (int r = 0; r < kRun / 64; r++) {
    const uint32_t slot = r * 64 + lane;
    const uint32_t w01 = slot + ((uint32_t)run << 16), w2 = slot * 7u;
    if (MODE == 0) {
      uint16_t* d = (uint16_t*)(g + slot * 6);
      d[0] = (uint16_t)w01; d[1] = (uint16_t)(w01 >> 16); d[2] = (uint16_t)w2;
    } else if (MODE == 1) {
      uint8_t* d = g + slot * 6;
      if (slot & 1) { *(uint16_t*)d = (uint16_t)w01; *(uint32_t*)(d + 2) = (w01 >> 16) | (w2 << 16); }
      else { *(uint32_t*)d = w01; *(uint16_t*)(d + 4) = (uint16_t)w2; }
    } else {
      uint16_t* d = (uint16_t*)(lds[wave] + slot * 6);
      d[0] = (uint16_t)w01; d[1] = (uint16_t)(w01 >> 16); d[2] = (uint16_t)w2;
    }
  }
  if (MODE == 2) {
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kRun * 6 / 16; i += 64) ((uint4*)g)[i] = ((const uint4*)lds[wave])[i];
  }
}

int main() {
  const int64_t n_runs = 1 << 18;  // 512 Mi entries, 3 GiB
  const size_t bytes = (size_t)n_runs * kRun * 6;
  uint8_t* out;
  CHECK(hipMalloc(&out, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<uint8_t> ref(1 << 20), got(1 << 20);
  const char* names[4] = {"A 3 x short", "B dword+short", "C LDS-staged 16B", "D plain 16B"};
  for (int m = 0; m < 4; m++) {
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
      CHECK(hipEventRecord(e0));
      const dim3 grid((unsigned)(n_runs / 4)), block(256);
      if (m == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, n_runs);
      if (m == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, n_runs);
      if (m == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, n_runs);
      if (m == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, n_runs);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(m == 0 ? ref.data() : got.data(), out + (bytes / 2), 1 << 20, hipMemcpyDeviceToHost));
    const bool same = m == 0 || m == 3 || ref == got;
    printf("%-18s %8.3f ms  %7.1f GB/s  %s\n", names[m], best, bytes / best / 1e6, same ? "" : "MISMATCH");
  }
  return 0;
}
