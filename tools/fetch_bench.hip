// fetch_bench - is a lone wave bound by instruction FETCH (bytes of code) rather than by issue (instructions)?
// Straight-line runs of independent instructions in 4-byte and 8-byte encodings, and the same work in a loop.
// Design tool.  Build: hipcc --offload-arch=gfx950 -O2 -o fetch_bench fetch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 128
__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
#define TIMED(body, per)                                         \
  t0 = now();                                                    \
  _Pragma("unroll") for (int i = 0; i < REP; i++) { body; }      \
  t1 = now();                                                    \
  if (threadIdx.x == 0) out[slot] = (t1 - t0), cnt[slot] = REP * (per); \
  slot++;
__global__ void k(uint64_t* out, int* cnt, int loops) {
  uint64_t t0, t1;
  int slot = 0;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, v0 = threadIdx.x, v1 = 5, v2 = 6, v3 = 7;
  // 0: 4 independent scalar adds, 4-byte encodings
  TIMED(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)), 4)
  // 1: the same with 32-bit literals: 8-byte encodings
  TIMED(asm volatile("s_add_u32 %0, %0, 0x12345\n s_add_u32 %1, %1, 0x12345\n s_add_u32 %2, %2, 0x12345\n s_add_u32 %3, %3, 0x12345" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)), 4)
  // 2: 4 independent vector adds, VOP2 (4 bytes)
  TIMED(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)), 4)
  // 3: the same as VOP3 (8 bytes)
  TIMED(asm volatile("v_add_u32_e64 %0, %0, 1\n v_add_u32_e64 %1, %1, 1\n v_add_u32_e64 %2, %2, 1\n v_add_u32_e64 %3, %3, 1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)), 4)
  // 4: s_nop 0 (4 bytes, nothing to wait for)
  TIMED(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"), 4)
  // 5: one dependent scalar chain, 4-byte
  TIMED(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s0)), 4)
  // 6: one dependent scalar chain, 8-byte
  TIMED(asm volatile("s_add_u32 %0, %0, 0x12345\n s_add_u32 %0, %0, 0x12345\n s_add_u32 %0, %0, 0x12345\n s_add_u32 %0, %0, 0x12345" : "+s"(s0)), 4)
  // 7: a LOOP of 16 independent 4-byte scalar adds (64 bytes of code + the loop's own three instructions)
  {
    uint32_t n = (uint32_t)loops;
    t0 = now();
    asm volatile("1:\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                 " s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                 " s_sub_u32 %4, %4, 1\n s_cmp_lg_u32 %4, 0\n s_cbranch_scc1 1b" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(n)::"scc");
    t1 = now();
    if (threadIdx.x == 0) out[slot] = (t1 - t0), cnt[slot] = loops * 19;
    slot++;
  }
  // 8: a LOOP of 64 independent 4-byte scalar adds
  {
    uint32_t n = (uint32_t)loops;
    t0 = now();
    asm volatile("1:\n .rept 16\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n .endr\n"
                 " s_sub_u32 %4, %4, 1\n s_cmp_lg_u32 %4, 0\n s_cbranch_scc1 1b" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(n)::"scc");
    t1 = now();
    if (threadIdx.x == 0) out[slot] = (t1 - t0), cnt[slot] = loops * 67;
    slot++;
  }
  // 9: a LOOP of 64 independent 8-byte scalar adds
  {
    uint32_t n = (uint32_t)loops;
    t0 = now();
    asm volatile("1:\n .rept 16\n s_add_u32 %0, %0, 0x12345\n s_add_u32 %1, %1, 0x12345\n s_add_u32 %2, %2, 0x12345\n s_add_u32 %3, %3, 0x12345\n .endr\n"
                 " s_sub_u32 %4, %4, 1\n s_cmp_lg_u32 %4, 0\n s_cbranch_scc1 1b" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(n)::"scc");
    t1 = now();
    if (threadIdx.x == 0) out[slot] = (t1 - t0), cnt[slot] = loops * 67;
    slot++;
  }
  // 10: straight line, two interleaved dependent chains (4-byte)
  TIMED(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+s"(s0), "+s"(s1)), 4)
  // 11: straight line, two interleaved dependent chains (8-byte)
  TIMED(asm volatile("s_add_u32 %0, %0, 0x12345\n s_add_u32 %1, %1, 0x12345\n s_add_u32 %0, %0, 0x12345\n s_add_u32 %1, %1, 0x12345" : "+s"(s0), "+s"(s1)), 4)
  if (s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3 == 0x12345) out[31] = 1;
}
int main() {
  uint64_t* d; int* c;
  hipMalloc(&d, 32 * 8); hipMalloc(&c, 32 * 4);
  hipMemset(d, 0, 32 * 8); hipMemset(c, 0, 32 * 4);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, 200);
  hipDeviceSynchronize();
  uint64_t h[32]; int hc[32];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
  const char* nm[] = {"4 independent scalar, 4-byte", "4 independent scalar, 8-byte (literal)", "4 independent vector, VOP2 4-byte", "4 independent vector, VOP3 8-byte", "s_nop 0",
                      "dependent scalar chain, 4-byte", "dependent scalar chain, 8-byte", "LOOP of 16 independent scalar 4-byte", "LOOP of 64 independent scalar 4-byte", "LOOP of 64 independent scalar 8-byte",
                      "2 scalar chains interleaved, 4-byte", "2 scalar chains interleaved, 8-byte"};
  for (int i = 0; i < 12; i++) printf("%-44s %6.2f cycles per instruction\n", nm[i], (double)h[i] / hc[i]);
  return 0;
}
