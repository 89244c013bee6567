// rans_sload2_bench - where k_rans_s' time goes: the product's own 64-step block (scpr_rans_s.hpp's macros) over 16-byte records
// that (a) lie in memory from before the launch, read for the first time / a second time, (b) were stored by the wave's own lanes
// two trips earlier, as in the kernel.  One wave.  Design tool.
// Build: hipcc --offload-arch=gfx950 -O2 -I../screenpressor_amd/csrc -o rans_sload2_bench rans_sload2_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "scpr_rans_s.hpp"
using namespace scpr;
template <int MODE>
__global__ void k(uint64_t* out, u32* sink, uint4* rec, int trips) {
  const int lane = threadIdx.x;
  u32 x = 1u << 23, acc = 0, tp1 = 0, tp2 = 0;
  const uint4 mine = rec[lane];
  uint64_t t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; t++) {
    u32 touch = 0;
    if (MODE == 2) touch = ((const volatile u32*)(rec + (size_t)(t + 3) * 64 + lane))[0];
    if (MODE >= 1) rec[(size_t)(t + 2) * 64 + lane] = mine;
    if (MODE == 3) touch = ((const volatile u32*)(rec + (size_t)(t + 4) * 64 + lane))[0];
    u32 vout = 0;
    const uint4* p = rec + (size_t)t * 64;
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
    if (MODE == 3) {
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(tp2) : : "memory");
      acc ^= vout ^ (tp2 & 1);
      tp2 = tp1, tp1 = touch;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) : : "memory");
      acc ^= vout ^ (touch & 1);
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  sink[lane] = x + acc;
}
int main() {
  const int trips = 2048;
  std::vector<uint4> h((size_t)(trips + 4) * 64);
  for (size_t e = 0; e < h.size(); e++) {
    uint32_t freq = 300 + (e * 7) % 900, shift = 0;
    while (freq > (1u << shift)) shift++;
    h[e] = make_uint4(freq << 19, (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq), 4096 - freq, (((e * 13) & 4095) << 5) | (shift - 1));
  }
  uint4* rec; uint64_t* d; u32* s;
  hipMalloc(&rec, h.size() * 16 * 4); hipMalloc(&d, 64); hipMalloc(&s, 256);
  for (int r = 0; r < 4; r++) hipMemcpy(rec + r * h.size(), h.data(), h.size() * 16, hipMemcpyHostToDevice);
  uint64_t c;
  auto show = [&](const char* what) { hipDeviceSynchronize(); hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost); printf("%-70s %.1f cycles per entry\n", what, (double)c / (trips * 64.0)); };
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, rec, trips); show("records from before the launch, first pass over them");
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, rec, trips); show("the same records again (2 MB: in the L2 if the same XCD runs it)");
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, s, rec, trips); show("and again");
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, rec + h.size(), trips); show("records stored by the wave's own lanes two trips before they are read");
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, s, rec + h.size(), trips); show("the same again");
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, s, rec + 2 * h.size(), trips); show("stored two trips before, their lines READ by the lanes a trip before that");
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, s, rec + 2 * h.size(), trips); show("the same again");
  hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, s, rec + 3 * h.size(), trips - 4); show("lines read FOUR trips ahead, the wait leaves that one load out (trips - 4 of them)");
  hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, s, rec + 3 * h.size(), trips - 4); show("the same again");
  return 0;
}
