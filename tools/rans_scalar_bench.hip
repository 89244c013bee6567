// rans_scalar_bench - what the rANS state step costs per entry with the state on the SCALAR unit (one wave per block, the
// entry's constants taken out of vector registers with lane reads) against the vector form of k_rans.  Design tool.
// Build: hipcc --offload-arch=gfx950 -O2 -o rans_scalar_bench rans_scalar_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
// The entry's five constants are read into scalar registers one entry AHEAD (two register sets, A and B): a lane read takes ~16
// cycles to reach a scalar consumer, and read where they are used they would add that to every entry's chain.
#define RD(K, P) "v_readlane_b32 s" #P "0, %1, " #K "\n v_readlane_b32 s" #P "1, %2, " #K "\n v_readlane_b32 s" #P "2, %3, " #K "\n v_readlane_b32 s" #P "3, %4, " #K "\n v_readlane_b32 s" #P "4, %5, " #K "\n"
#define STEP(K, KN, P, Q)                                                                                              \
  asm volatile("s_cmp_ge_u32 %0, s" #P "0\n s_cselect_b32 s20, 8, 0\n s_lshr_b32 s21, %0, s20\n"                          \
               "v_readlane_b32 s" #Q "0, %1, " #KN "\n v_readlane_b32 s" #Q "1, %2, " #KN "\n"                              \
               "s_cmp_ge_u32 s21, s" #P "0\n s_cselect_b32 s20, 8, 0\n s_lshr_b32 s21, s21, s20\n"                        \
               "v_writelane_b32 %6, %0, " #K "\n v_readlane_b32 s" #Q "2, %3, " #KN "\n"                                  \
               "s_mul_hi_u32 s22, s21, s" #P "1\n v_readlane_b32 s" #Q "3, %4, " #KN "\n s_lshr_b32 s22, s22, s" #P "3\n"    \
               "v_readlane_b32 s" #Q "4, %5, " #KN "\n s_mul_i32 s22, s22, s" #P "2\n s_add_u32 s21, s21, s" #P "4\n s_add_u32 %0, s22, s21" \
               : "+s"(x), "+v"(vxmax), "+v"(vrcp), "+v"(vmul), "+v"(vsh), "+v"(vbias), "+v"(vout)::"s20", "s21", "s22", "s23", "s30", "s31", "s32", "s33", "s34", "s40", "s41", "s42", "s43", "s44", "scc");
#define STEP2(K) STEP(K, K - 1, 3, 4)
__global__ void k(uint64_t* out, uint32_t* sink, int trips) {
  const uint32_t lane = threadIdx.x;
  // freq 300 of 4096 (a middling symbol): x_max = freq << 19, exact reciprocal of ryg's RansEncSymbolInit
  uint32_t freq = 300 + (lane & 7), shift = 0;
  while (freq > (1u << shift)) shift++;
  uint32_t vxmax = freq << 19, vrcp = (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq), vmul = 4096 - freq, vsh = shift - 1, vbias = 17 * lane, vout = 0;
  uint32_t x = 1u << 23;
  uint64_t t0 = now();
  for (int t = 0; t < trips; t++) {
    asm volatile(RD(63, 3) : "+s"(x), "+v"(vxmax), "+v"(vrcp), "+v"(vmul), "+v"(vsh), "+v"(vbias)::"s30", "s31", "s32", "s33", "s34");
    STEP(63, 62, 3, 4) STEP(62, 61, 4, 3) STEP(61, 60, 3, 4) STEP(60, 59, 4, 3) STEP(59, 58, 3, 4) STEP(58, 57, 4, 3) STEP(57, 56, 3, 4) STEP(56, 55, 4, 3)
    STEP(55, 54, 3, 4) STEP(54, 53, 4, 3) STEP(53, 52, 3, 4) STEP(52, 51, 4, 3) STEP(51, 50, 3, 4) STEP(50, 49, 4, 3) STEP(49, 48, 3, 4) STEP(48, 47, 4, 3)
    STEP(47, 46, 3, 4) STEP(46, 45, 4, 3) STEP(45, 44, 3, 4) STEP(44, 43, 4, 3) STEP(43, 42, 3, 4) STEP(42, 41, 4, 3) STEP(41, 40, 3, 4) STEP(40, 39, 4, 3)
    STEP(39, 38, 3, 4) STEP(38, 37, 4, 3) STEP(37, 36, 3, 4) STEP(36, 35, 4, 3) STEP(35, 34, 3, 4) STEP(34, 33, 4, 3) STEP(33, 32, 3, 4) STEP(32, 31, 4, 3)
    STEP(31, 30, 3, 4) STEP(30, 29, 4, 3) STEP(29, 28, 3, 4) STEP(28, 27, 4, 3) STEP(27, 26, 3, 4) STEP(26, 25, 4, 3) STEP(25, 24, 3, 4) STEP(24, 23, 4, 3)
    STEP(23, 22, 3, 4) STEP(22, 21, 4, 3) STEP(21, 20, 3, 4) STEP(20, 19, 4, 3) STEP(19, 18, 3, 4) STEP(18, 17, 4, 3) STEP(17, 16, 3, 4) STEP(16, 15, 4, 3)
    STEP(15, 14, 3, 4) STEP(14, 13, 4, 3) STEP(13, 12, 3, 4) STEP(12, 11, 4, 3) STEP(11, 10, 3, 4) STEP(10, 9, 4, 3) STEP(9, 8, 3, 4) STEP(8, 7, 4, 3)
    STEP(7, 6, 3, 4) STEP(6, 5, 4, 3) STEP(5, 4, 3, 4) STEP(4, 3, 4, 3) STEP(3, 2, 3, 4) STEP(2, 1, 4, 3) STEP(1, 0, 3, 4) STEP(0, 63, 4, 3)
    vbias ^= vout & 1;  // (the trip's output is used)
  }
  uint64_t t1 = now();
  if (lane == 0) out[0] = t1 - t0, sink[0] = x + vout;
}
int main() {
  uint64_t* d; uint32_t* s;
  hipMalloc(&d, 64); hipMalloc(&s, 64);
  const int trips = 256;
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, trips);
  hipDeviceSynchronize();
  uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("scalar rANS step, one wave: %.1f cycles per entry (17 instructions: 5 lane reads - one entry ahead -, 11 scalar, 1 lane write)\n", (double)h / (trips * 64.0));
  // several waves on one CU (the scalar unit is shared by the CU's four SIMDs)
  for (int w : {4, 8}) {
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64 * w), 0, 0, d, s, trips);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("  %d waves on the CU: %.1f cycles per entry (wave 0's clock)\n", w, (double)h / (trips * 64.0));
  }
  return 0;
}
