// lonewave_bench — what one wave alone on a SIMD pays per dependent operation on gfx950.
// The GOP decoder is a single serial chain per wave, so its speed is set by these latencies,
// not by bandwidth.  Build: hipcc --offload-arch=gfx950 -O2 -o lonewave_bench lonewave_bench.hip
// Output: cycles (s_memtime) per step of each dependent chain, and the s_memtime rate in MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 512

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

__global__ void k_bench(uint64_t* out, uint32_t* gbuf, int n_g) {
  __shared__ uint32_t lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (uint32_t)((i * 7 + 13) & 4095);
  __syncthreads();
  uint64_t t0, t1;
  int slot = 0;
  uint32_t v = lane, s = 0;

  // 0: dependent VALU adds
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;

  // 1: dependent SALU adds
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;

  // 2: VALU -> readfirstlane -> SALU -> VALU round trip
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1" : "+v"(v), "+s"(s));
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;

  // 3: LDS pointer chase (ds_read_b32 dependent chain)
  {
    uint32_t a = (uint32_t)lane * 4;
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; i++) {
      uint32_t r;
      asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
      a = (r & 4095u) * 4;
    }
    t1 = now();
    v += a;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }

  // 4: LDS b128 read chase
  {
    uint32_t a = (uint32_t)lane * 16;
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; i++) {
      uint4 r;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
      a = (r.x & 1023u) * 16;
    }
    t1 = now();
    v += a;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }

  // 5: taken uniform branch (a loop whose body is one s_add)
  {
    uint32_t c = REP;
    t0 = now();
    asm volatile(
        "1:\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
        : "+s"(c)::"scc");
    t1 = now();
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }

  // 6: DPP row scan step (v_add_dpp row_shr:1 + required nop), dependent
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;

  // 7: ballot -> scalar ctz -> readlane -> VALU
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) {
    uint64_t m = __ballot(v & 1u);
    int p = m ? __builtin_ctzll(m) : 0;
    uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)v, p);
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "s"(q));
  }
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;

  // 8: global pointer chase, wave-uniform address, vector load (L2 / HBM depending on n_g)
  {
    uint32_t a = 0;
    t0 = now();
    for (int i = 0; i < REP; i++) {
      a = __hip_atomic_load(&gbuf[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    }
    t1 = now();
    v += a;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 9: global pointer chase through the vector L1 (plain loads)
  {
    uint32_t a = 0;
    t0 = now();
    for (int i = 0; i < REP; i++) {
      a = gbuf[a];
      asm volatile("" : "+v"(a));
    }
    t1 = now();
    v += a;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 10: scalar-cache pointer chase (s_load_dword)
  {
    uint32_t a = 0;
    const uint32_t* base = gbuf;
    t0 = now();
    for (int i = 0; i < REP; i++) {
      uint32_t r;
      asm volatile("s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(a * 4) : "memory");
      a = r;
    }
    t1 = now();
    v += a;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 11: LDS write then read of another lane's word (store -> load forwarding across lanes)
  {
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; i++) {
      uint32_t r;
      asm volatile("ds_write_b32 %1, %2\n ds_read_b32 %0, %3\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)lane * 4), "v"(v), "v"((uint32_t)((lane + 1) & 63) * 4) : "memory");
      v = r + 1;
    }
    t1 = now();
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 12: 32-bit multiply chain (v_mul_lo_u32)
  t0 = now();
#pragma unroll
  for (int i = 0; i < REP; i++) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v));
  t1 = now();
  if (lane == 0) out[slot] = t1 - t0;
  slot++;
  // 13: independent VALU adds (issue rate, 4 chains)
  {
    uint32_t a = v, b = v + 1, c = v + 2, d = v + 3;
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP / 4; i++) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    t1 = now();
    v += a + b + c + d;
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 14: not-taken conditional branch + VALU (cost of a skipped s_cbranch)
  {
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; i++) asm volatile("s_cmp_eq_u32 %1, -1\n s_cbranch_scc1 2f\n v_add_u32 %0, %0, 1\n2:" : "+v"(v) : "s"(s) : "scc");
    t1 = now();
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  // 15: taken forward branch over one instruction
  {
    t0 = now();
#pragma unroll
    for (int i = 0; i < REP; i++) asm volatile("s_cmp_lg_u32 %1, -1\n s_cbranch_scc1 3f\n v_add_u32 %0, %0, 1\n3:\n v_add_u32 %0, %0, 2" : "+v"(v) : "s"(s) : "scc");
    t1 = now();
    if (lane == 0) out[slot] = t1 - t0;
    slot++;
  }
  if (lane == 0) {
    out[30] = now();
    out[31] = wall_clock64();
  }
  gbuf[n_g + lane] = v + s;
}

int main() {
  const char* names[] = {"valu add (dependent)", "salu add (dependent)", "valu->readfirstlane->salu->valu", "lds b32 pointer chase", "lds b128 pointer chase",
                         "taken loop branch (3 salu)", "dpp row_shr + nop", "ballot->ctz->readlane->valu", "global chase, L2 (sc1)", "global chase, vector L1",
                         "scalar-cache chase", "lds write + cross-lane read", "v_mul_lo_u32 (dependent)", "valu add (4 independent)", "cmp + not-taken branch + valu",
                         "cmp + taken fwd branch + valu"};
  const int n_g = 1 << 14;
  std::vector<uint32_t> h(n_g + 64);
  for (int i = 0; i < n_g; i++) h[i] = (uint32_t)((i * 2654435761u + 12345u) % n_g);
  uint32_t* g;
  uint64_t* out;
  hipMalloc(&g, h.size() * 4);
  hipMalloc(&out, 64 * 8);
  hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  uint64_t r[64], r2[64];
  for (int pass = 0; pass < 3; pass++) {
    hipMemset(out, 0, 64 * 8);
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, out, g, n_g);
    hipDeviceSynchronize();
    hipMemcpy(pass == 1 ? r : r2, out, 64 * 8, hipMemcpyDeviceToHost);
  }
  printf("s_memtime ticks per wall-clock (100 MHz) tick: %.3f\n", (double)(r2[30] - r[30]) / (double)(r2[31] - r[31]));
  for (int i = 0; i < 16; i++) printf("%-36s %8.2f ticks/step\n", names[i], (double)r2[i] / REP);
  return 0;
}
