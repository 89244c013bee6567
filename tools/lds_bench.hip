// lds_bench - what LDS operations cost a lone wave (design tool): read round trips, writes and atomic adds in front of a read,
// aligned / unaligned / same-address forms, exec switching, lane reads on the way to the scalar unit.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_bench lds_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }
#define TIMED(body)                                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
  t0 = now();                                                    \
  _Pragma("unroll") for (int i = 0; i < REP; i++) { body; }      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
  t1 = now();                                                    \
  if (threadIdx.x == 0) out[slot] = (t1 - t0);                   \
  slot++;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint64_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0;  // (every value read is added to an address: zero keeps the addresses where they are)
  __syncthreads();
  uint64_t t0, t1;
  int slot = 0;
  const uint32_t lane = threadIdx.x;
  uint32_t base = (uint32_t)(size_t)lds;
  uint32_t a_lane = base + 4 * lane;            // one word per lane, conflict free
  uint32_t a_l15 = base + 4 * (lane & 15);      // four lanes per word
  uint32_t a_same = base;                       // all lanes one word
  uint32_t a_16 = base + 16 * lane;             // 16 bytes per lane, aligned
  uint32_t v = 0, one = 1, k0 = lane == 0 ? 1u : 0u, s = 0;
  u32x4 q;
  // 0: read b32 -> use (address depends on the value read: a pure round trip)
  TIMED(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane)::"memory"))
  // 1: read b128 all lanes the same aligned address
  TIMED(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a_same) : "memory"); asm volatile("v_and_b32 %0, 0, %1\n v_add_u32 %2, %2, %0" : "=v"(v), "+v"(q.x), "+v"(a_same)))
  // 2: read b128 per lane 16-byte aligned
  TIMED(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a_16) : "memory"); asm volatile("v_and_b32 %0, 0, %1\n v_add_u32 %2, %2, %0" : "=v"(v), "+v"(q.x), "+v"(a_16)))
  // 3: read b128 per lane at 4-byte steps (unaligned for three lanes of four)
  TIMED(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a_l15) : "memory"); asm volatile("v_and_b32 %0, 0, %1\n v_add_u32 %2, %2, %0" : "=v"(v), "+v"(q.x), "+v"(a_l15)))
  // 4: write b32 (64 lanes, own words) then read -> use
  TIMED(asm volatile("ds_write_b32 %1, %2 offset:1024\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(one) : "memory"))
  // 5: add (64 lanes, own words) then read -> use
  TIMED(asm volatile("ds_add_u32 %1, %2 offset:1024\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(k0) : "memory"))
  // 6: add (four lanes per word) then read -> use
  TIMED(asm volatile("ds_add_u32 %2, %3 offset:1024\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(a_l15), "v"(k0) : "memory"))
  // 7: add (all lanes one word) then read -> use
  TIMED(asm volatile("ds_add_u32 %2, %3 offset:1024\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(a_same), "v"(k0) : "memory"))
  // 8: add by lane 0 alone (exec switched) then read -> use
  TIMED(asm volatile("s_mov_b64 exec, 1\n ds_add_u32 %2, %3 offset:1024\n s_mov_b64 exec, -1\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(a_same), "v"(one) : "memory"))
  // 9: write by all lanes to one word then read -> use
  TIMED(asm volatile("ds_write_b32 %2, %3 offset:1024\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(a_same), "v"(one) : "memory"))
  // 10: read -> readfirstlane -> scalar add -> back to the address (the decoder's record fetch: value to the scalar unit and on)
  TIMED(asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %2, %0\n s_and_b32 %2, %2, 0\n v_add_u32 %1, %1, %2" : "=v"(v), "+v"(a_lane), "+s"(s)::"memory", "scc"))
  // 11: readfirstlane -> s_add -> v_mov -> readfirstlane ... (no LDS): the lane-read / scalar / vector loop
  TIMED(asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1" : "+v"(v), "+s"(s)::"scc"))
  // 12: exec switched around a vector add (what switching exec costs by itself)
  TIMED(asm volatile("s_mov_b64 exec, 1\n v_add_u32 %0, %0, 1\n s_mov_b64 exec, -1" : "+v"(v)))
  // 13: two reads in flight, use of the first only (lgkmcnt(1))
  TIMED(asm volatile("ds_read_b32 %0, %1\n ds_read_b32 %2, %1 offset:2048\n s_waitcnt lgkmcnt(1)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane), "=v"(one)::"memory"))
  // 14: read; 64-lane add; wait for the read only
  TIMED(asm volatile("ds_read_b32 %0, %1\n ds_add_u32 %2, %3 offset:1024\n s_waitcnt lgkmcnt(1)\n v_add_u32 %1, %1, %0" : "=v"(v), "+v"(a_lane) : "v"(a_l15), "v"(k0) : "memory"))
  if (v + s + q.x + one == 0x12345) out[31] = 1;
}
int main() {
  uint64_t* d;
  hipMalloc(&d, 32 * 8);
  hipMemset(d, 0, 32 * 8);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  uint64_t h[32];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[] = {"read b32 -> use", "read b128, all lanes one aligned address", "read b128, 16 bytes per lane aligned", "read b128 at 4-byte steps (unaligned)",
                      "write (own words) + read -> use", "add (own words) + read -> use", "add (4 lanes per word) + read -> use", "add (all lanes one word) + read -> use",
                      "add by lane 0 (exec switched) + read -> use", "write (all lanes one word) + read -> use", "read -> readfirstlane -> s_and -> v_add", "readfirstlane -> s_add -> v_mov loop",
                      "exec switched around a v_add", "two reads, wait for the first", "read, 64-lane add, wait for the read"};
  for (int i = 0; i < 15; i++) printf("%-48s %7.1f cycles per turn\n", nm[i], (double)h[i] / REP);
  return 0;
}
