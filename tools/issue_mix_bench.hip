// issue_mix_bench — how the order of scalar and vector instructions changes what a lone wave pays per instruction
// (companion of lonewave_bench.hip; design tool).  Build: hipcc --offload-arch=gfx950 -O2 -o issue_mix issue_mix_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 128
__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
#define TIMED(name, body, per)                                   \
  t0 = now();                                                    \
  _Pragma("unroll") for (int i = 0; i < REP; i++) { body; }      \
  t1 = now();                                                    \
  if (threadIdx.x == 0) out[slot] = (t1 - t0), cnt[slot] = REP * (per); \
  slot++;
__global__ void k(uint64_t* out, int* cnt) {
  uint64_t t0, t1;
  int slot = 0;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, v0 = threadIdx.x, v1 = 5, v2 = 6, v3 = 7;
  // 0: 4 independent scalar chains, round robin
  TIMED("salu4", asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)), 4)
  // 1: 4 independent vector chains
  TIMED("valu4", asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)), 4)
  // 2: alternating scalar / vector, all independent of their neighbours (2 chains each)
  TIMED("alt", asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+v"(v0), "+v"(v1)), 4)
  // 3: one scalar chain and one vector chain interleaved (each depends on the instruction two back)
  TIMED("alt1", asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1" : "+s"(s0), "+v"(v0)), 4)
  // 4: two scalar chains interleaved (depends on the instruction two back)
  TIMED("salu2", asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+s"(s0), "+s"(s1)), 4)
  // 5: compare + select adjacent
  TIMED("cmpsel", asm volatile("s_cmp_lt_u32 %0, %1\n s_cselect_b32 %2, %0, %1\n s_cmp_lt_u32 %1, %3\n s_cselect_b32 %0, %1, %3" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc"), 4)
  // 6: compare, an unrelated add, select
  TIMED("cmp_x_sel", asm volatile("s_cmp_lt_u32 %0, %1\n v_add_u32 %3, %3, 1\n s_cselect_b32 %2, %0, %1\n v_add_u32 %4, %4, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+v"(v0), "+v"(v1)::"scc"), 4)
  // 7: compare + not-taken branch adjacent
  TIMED("cmpbr", asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %1, %1, 1\n1:" : "+s"(s0), "+s"(s1)::"scc"), 3)
  // 8: compare, unrelated vector op, not-taken branch
  TIMED("cmp_x_br", asm volatile("s_cmp_eq_u32 %0, 0\n v_add_u32 %2, %2, 1\n s_cbranch_scc1 1f\n s_add_u32 %1, %1, 1\n1:" : "+s"(s0), "+s"(s1), "+v"(v0)::"scc"), 4)
  // 9: v_cmp -> s_and vcc -> s_flbit -> v_readlane chain (the table search)
  TIMED("search", asm volatile("v_cmp_ge_u32 vcc, %1, %0\n s_and_b32 %2, vcc_lo, 0xffff\n s_flbit_i32_b32 %2, %2\n s_sub_i32 %2, 31, %2\n v_readlane_b32 %1, %0, %2" : "+v"(v0), "+s"(s0), "+s"(s1)::"vcc", "scc"), 5)
  // 10: scalar run of 8 then vector run of 8, all independent within the run (grouped by type)
  TIMED("grp", asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)), 8)
  if (s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3 == 0x12345) out[31] = 1;
}
int main() {
  uint64_t* d; int* c;
  hipMalloc(&d, 32 * 8); hipMalloc(&c, 32 * 4);
  hipMemset(d, 0, 32 * 8); hipMemset(c, 0, 32 * 4);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
  hipDeviceSynchronize();
  uint64_t h[32]; int hc[32];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
  const char* nm[] = {"4 independent scalar", "4 independent vector", "scalar/vector alternating, independent", "1 scalar + 1 vector chain alternating", "2 scalar chains alternating",
                      "compare+select adjacent", "compare, other, select", "compare+branch(not taken) adjacent", "compare, other, branch", "search chain (5 instr)", "4 scalar then 4 vector"};
  for (int i = 0; i < 11; i++) printf("%-44s %6.2f cycles per instruction\n", nm[i], (double)h[i] / hc[i]);
  return 0;
}
