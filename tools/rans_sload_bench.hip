// rans_sload_bench - the rANS state step with the state on the SCALAR unit and the entry's constants brought in by SCALAR LOADS
// (32-byte records { x_max, reciprocal, 4096 - freq, shift, bias, -, -, - } laid out in memory, sixteen dwords = two entries per
// s_load_dwordx16, a group of four entries ahead), one wave per block; against rans_scalar_bench (lane reads: 137 cycles per
// entry) and k_rans' vector form (~137).  Also: several such waves per CU / per SIMD (the scalar unit is shared).  Design tool.
// Build: hipcc --offload-arch=gfx950 -O2 -o rans_sload_bench rans_sload_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
// one entry from registers s[B .. B+4]; the state before it goes to lane K of vout (what the byte writer needs)
#define ENT(B0, B1, B2, B3, B4, K)                                                                                       \
  "s_cmp_ge_u32 %0, s" #B0 "\n s_cselect_b32 s20, 8, 0\n v_writelane_b32 %1, %0, " #K "\n s_lshr_b32 s21, %0, s20\n"            \
  "s_cmp_ge_u32 s21, s" #B0 "\n s_cselect_b32 s20, 8, 0\n s_lshr_b32 s21, s21, s20\n"                                     \
  "s_mul_hi_u32 s22, s21, s" #B1 "\n s_lshr_b32 s22, s22, s" #B3 "\n s_mul_i32 s22, s22, s" #B2 "\n s_add_u32 s21, s21, s" #B4 "\n s_add_u32 %0, s22, s21\n"
#define CLOB "s20", "s21", "s22", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
__global__ void k(uint64_t* out, uint32_t* sink, const uint32_t* rec, int groups) {
  const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t* r = rec + (size_t)wave * groups * 64;  // four entries of eight dwords per group... eight entries per 64 dwords
  uint32_t x = 1u << 23, vout = 0;
  uint64_t t0 = now();
  // set A = s36..s67 (four entries), set B = s68..s99
  asm volatile("s_load_dwordx16 s[36:51], %2, 0x0\n s_load_dwordx16 s[52:67], %2, 0x40\n" : "+s"(x), "+v"(vout) : "s"(r) : CLOB);
  for (int g = 0; g < groups; g += 2) {
    const uint32_t* p = r + (size_t)g * 32;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[68:83], %2, 0x80\n s_load_dwordx16 s[84:99], %2, 0xc0\n"
                 ENT(36, 37, 38, 39, 40, 0) ENT(44, 45, 46, 47, 48, 1) ENT(52, 53, 54, 55, 56, 2) ENT(60, 61, 62, 63, 64, 3)
                 "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[36:51], %2, 0x100\n s_load_dwordx16 s[52:67], %2, 0x140\n"
                 ENT(68, 69, 70, 71, 72, 4) ENT(76, 77, 78, 79, 80, 5) ENT(84, 85, 86, 87, 88, 6) ENT(92, 93, 94, 95, 96, 7)
                 : "+s"(x), "+v"(vout) : "s"(p) : CLOB);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint64_t t1 = now();
  if (lane == 0) out[wave] = t1 - t0, sink[wave] = x + vout;
}
int main() {
  const int groups = 4096, maxw = 256 * 16;
  std::vector<uint32_t> h((size_t)(groups + 4) * 32);
  for (int e = 0; e < (groups + 4) * 4; e++) {
    uint32_t freq = 300 + (e * 7) % 900, shift = 0;
    while (freq > (1u << shift)) shift++;
    uint32_t* q = &h[(size_t)e * 8];
    q[0] = freq << 19, q[1] = (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq), q[2] = 4096 - freq, q[3] = shift - 1, q[4] = (e * 13) & 4095;
  }
  uint32_t* rec; uint64_t* d; uint32_t* s;
  hipMalloc(&rec, (size_t)maxw * groups * 64 * 4 + 4096); hipMalloc(&d, maxw * 8); hipMalloc(&s, maxw * 4);
  for (int w = 0; w < maxw; w++) hipMemcpy(rec + (size_t)w * groups * 64, h.data(), (size_t)groups * 32 * 4 + 512, hipMemcpyHostToDevice);
  // NB every wave reads (groups * 32) dwords of ITS region: distinct lines per wave, as the real kernel's records would be
  struct { int blocks, threads; const char* what; } cfg[] = {{1, 64, "one wave"}, {1, 256, "4 waves on one CU (one per SIMD)"}, {1, 512, "8 waves on one CU"}, {1, 1024, "16 waves on one CU"},
                                                            {256, 256, "1024 waves (4 per CU)"}, {256, 512, "2048 waves (8 per CU)"}, {512, 256, "2048 waves as 512 workgroups"}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), 0, 0, d, s, rec, groups);
    hipDeviceSynchronize();
    std::vector<uint64_t> t(c.blocks * c.threads / 64);
    hipMemcpy(t.data(), d, t.size() * 8, hipMemcpyDeviceToHost);
    uint64_t mx = 0, mn = ~0ull;
    for (auto v : t) mx = v > mx ? v : mx, mn = v < mn ? v : mn;
    printf("%-36s %.1f .. %.1f cycles per entry (13 instructions: 11 scalar, a lane write, a quarter of two loads)\n", c.what, (double)mn / (groups * 4.0), (double)mx / (groups * 4.0));
  }
  return 0;
}
