// scalar_cache_bench - what a lone wave pays for a dependent scalar load that hits the scalar data cache (a pointer chase inside
// 4 KB), for one that misses it and hits the L2 (a chase through 1 MB), for a 16-byte load, and for a scalar store + load of the
// same word.  Design tool (is the scalar cache a faster home for the decoder's hot records than LDS + lane reads?).
// Build: hipcc --offload-arch=gfx950 -O2 -o scalar_cache_bench scalar_cache_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void chase(uint64_t* out, uint32_t* sink, const uint32_t* tab, int steps) {
  uint32_t off = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; i += 8) {
    asm volatile(
        "s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
        "s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
        "s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
        "s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
        : "+s"(off) : "s"(tab) : "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0, sink[0] = off;
}
__global__ void chase4(uint64_t* out, uint32_t* sink, const uint32_t* tab, int steps) {  // 16-byte loads, the next offset in the first word
  uint32_t off = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; i += 4) {
    asm volatile(
        "s_load_dwordx4 s[40:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s40\n s_load_dwordx4 s[40:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s40\n"
        "s_load_dwordx4 s[40:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s40\n s_load_dwordx4 s[40:43], %1, %0\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s40\n"
        : "+s"(off) : "s"(tab) : "memory", "s40", "s41", "s42", "s43");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0, sink[0] = off;
}
__global__ void store_load(uint64_t* out, uint32_t* sink, uint32_t* tab, int steps) {  // a word is stored and read back: the read sees the store?
  uint32_t v = 1, bad = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; i++) {
    uint32_t r;
    const uint32_t off = (uint32_t)(i & 63) * 64;
    asm volatile("s_store_dword %1, %2, %3\n s_load_dword %0, %2, %3\n s_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(v), "s"(tab), "s"(off) : "memory");
    bad += r != v;
    v = v * 3 + 1;
  }
  asm volatile("s_dcache_wb" ::: "memory");
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0, sink[0] = bad;
}
int main() {
  const int steps = 1 << 14;
  uint32_t* tab; uint64_t* d; uint32_t* s;
  hipMalloc(&tab, 4 << 20); hipMalloc(&d, 64); hipMalloc(&s, 64);
  uint64_t c; uint32_t sv;
  auto fill = [&](uint32_t bytes, uint32_t stride) {  // a cycle of offsets over `bytes`, `stride` apart (a multiplicative walk)
    std::vector<uint32_t> h((4 << 20) / 4, 0);
    const uint32_t n = bytes / stride;
    for (uint32_t i = 0; i < n; i++) h[(size_t)i * stride / 4] = ((i * 167u + 13u) % n) * stride;  // 167 is coprime with the powers of two used
    hipMemcpy(tab, h.data(), 4 << 20, hipMemcpyHostToDevice);
  };
  auto show = [&](const char* what, int st) { hipDeviceSynchronize(); hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&sv, s, 4, hipMemcpyDeviceToHost); printf("%-72s %.1f cycles per step (check %u)\n", what, (double)c / st, sv); };
  fill(4096, 64);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, s, tab, steps);
  show("dependent s_load_dword, 4 KB (64 lines): scalar cache hits", steps);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chase4, dim3(1), dim3(64), 0, 0, d, s, tab, steps);
  show("dependent s_load_dwordx4 (+ s_mov), 4 KB", steps);
  fill(16384, 64);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, s, tab, steps);
  show("dependent s_load_dword, 16 KB (256 lines)", steps);
  fill(65536, 64);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, s, tab, steps);
  show("dependent s_load_dword, 64 KB", steps);
  fill(1 << 20, 64);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, s, tab, steps);
  show("dependent s_load_dword, 1 MB: L2 hits", steps);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(store_load, dim3(1), dim3(64), 0, 0, d, s, tab, 4096);
  show("s_store_dword + s_load_dword of the same word (check = stale reads)", 4096);
  return 0;
}
