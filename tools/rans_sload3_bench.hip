// rans_sload3_bench - k_rans_s' records in a RING that the wave's lanes overwrite every few trips (its lines stay in the L2: a
// store to a line the L2 does not hold does not bring it in - rans_sload2_bench), read back through the scalar data cache, which
// may still hold a lap-old copy: with and without s_dcache_inv per trip; the final state says whether stale records were read.
// Build: hipcc --offload-arch=gfx950 -O2 -I../screenpressor_amd/csrc -o rans_sload3_bench rans_sload3_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "scpr_rans_s.hpp"
using namespace scpr;
// RING = 0: a fresh region per trip.  INV: s_dcache_inv before the trip's first scalar load.
template <int RING, bool INV>
__global__ void k(uint64_t* out, u32* sink, uint4* rec, int trips) {
  const int lane = threadIdx.x;
  u32 x = 1u << 23, acc = 0;
  uint4 mine = rec[lane];
  auto slot = [&](int t) { return rec + (size_t)(RING ? t % RING : t) * 64; };
  auto rec_of = [&](int t) { uint4 r = mine; r.w += (u32)((t * 37) & 1023) << 5; return r; };
  slot(0)[lane] = rec_of(0);
  slot(1)[lane] = rec_of(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint64_t t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; t++) {
    slot(t + 2)[lane] = rec_of(t + 2);
    u32 vout = 0;
    const uint4* p = slot(t);
    if (INV) asm volatile("s_dcache_inv" ::: "memory");
    asm volatile(SCPR_RS_LOAD_A(0x0, 0x40)
                 SCPR_RS_LOAD_B(0x80, 0xc0) SCPR_RS_SET_A(0, 1, 2, 3, 4, 5, 6, 7)
                 SCPR_RS_LOAD_A(0x100, 0x140) SCPR_RS_SET_B(8, 9, 10, 11, 12, 13, 14, 15)
                 SCPR_RS_LOAD_B(0x180, 0x1c0) SCPR_RS_SET_A(16, 17, 18, 19, 20, 21, 22, 23)
                 SCPR_RS_LOAD_A(0x200, 0x240) SCPR_RS_SET_B(24, 25, 26, 27, 28, 29, 30, 31)
                 SCPR_RS_LOAD_B(0x280, 0x2c0) SCPR_RS_SET_A(32, 33, 34, 35, 36, 37, 38, 39)
                 SCPR_RS_LOAD_A(0x300, 0x340) SCPR_RS_SET_B(40, 41, 42, 43, 44, 45, 46, 47)
                 SCPR_RS_LOAD_B(0x380, 0x3c0) SCPR_RS_SET_A(48, 49, 50, 51, 52, 53, 54, 55)
                 "s_waitcnt lgkmcnt(0)\n\t" SCPR_RS_SET_B(56, 57, 58, 59, 60, 61, 62, 63)
                 : [x] "+s"(x), [vo] "+v"(vout) : [p] "s"(p) : SCPR_RS_CLOBBERS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= vout;
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0, out[1] = x;
  sink[lane] = x + acc;
}
int main() {
  const int trips = 2048;
  std::vector<uint4> h((size_t)(trips + 4) * 64);
  for (size_t e = 0; e < h.size(); e++) {
    uint32_t freq = 300 + (e * 7) % 900, shift = 0;
    while (freq > (1u << shift)) shift++;
    h[e] = make_uint4(freq << 19, (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq), 4096 - freq, (((e * 13) & 1023) << 5) | (shift - 1));
  }
  uint4* rec; uint64_t* d; u32* s;
  hipMalloc(&rec, h.size() * 16 * 8); hipMalloc(&d, 64); hipMalloc(&s, 256);
  for (int r = 0; r < 8; r++) hipMemcpy(rec + r * h.size(), h.data(), h.size() * 16, hipMemcpyHostToDevice);
  uint64_t c[2];
  auto show = [&](const char* what) { hipDeviceSynchronize(); hipMemcpy(c, d, 16, hipMemcpyDeviceToHost); printf("%-64s %.1f cycles per entry, final state %08x\n", what, (double)c[0] / (trips * 64.0), (unsigned)c[1]); };
  hipLaunchKernelGGL((k<0, false>), dim3(1), dim3(64), 0, 0, d, s, rec, trips); show("a fresh region per trip (the reference result)");
  hipLaunchKernelGGL((k<4, false>), dim3(1), dim3(64), 0, 0, d, s, rec + h.size(), trips); show("ring of 4 trips (4 KB), no invalidate");
  hipLaunchKernelGGL((k<4, true>), dim3(1), dim3(64), 0, 0, d, s, rec + 2 * h.size(), trips); show("ring of 4 trips, s_dcache_inv per trip");
  hipLaunchKernelGGL((k<16, false>), dim3(1), dim3(64), 0, 0, d, s, rec + 3 * h.size(), trips); show("ring of 16 trips (16 KB), no invalidate");
  hipLaunchKernelGGL((k<16, true>), dim3(1), dim3(64), 0, 0, d, s, rec + 4 * h.size(), trips); show("ring of 16 trips, s_dcache_inv per trip");
  hipLaunchKernelGGL((k<64, false>), dim3(1), dim3(64), 0, 0, d, s, rec + 5 * h.size(), trips); show("ring of 64 trips (64 KB), no invalidate");
  hipLaunchKernelGGL((k<64, true>), dim3(1), dim3(64), 0, 0, d, s, rec + 6 * h.size(), trips); show("ring of 64 trips, s_dcache_inv per trip");
  return 0;
}
