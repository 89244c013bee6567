// Design aid: does rocprim::radix_sort_pairs sort a bit range that ends at bit 32 for small inputs?
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
  for (size_t n : {1000ul, 100000ul, 199344ul, 400000ul, 2000000ul, 20000000ul})
    for (int endb : {29, 31, 32}) {
      std::vector<unsigned> hk(n), hv(n);
      std::mt19937 rng(7);
      for (size_t i = 0; i < n; i++) hk[i] = (unsigned)(rng() & (endb == 32 ? 0xFFFFFFFFu : ((1u << endb) - 1u))), hv[i] = (unsigned)i;
      unsigned *k0, *k1, *v0, *v1;
      hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
      hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
      hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
      size_t tmp = 0;
      rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 8, endb, 0);
      void* t; hipMalloc(&t, tmp);
      hipError_t e = rocprim::radix_sort_pairs(t, tmp, k0, k1, v0, v1, n, 8, endb, 0);
      hipDeviceSynchronize();
      std::vector<unsigned> ok(n), ov(n);
      hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(ov.data(), v1, n * 4, hipMemcpyDeviceToHost);
      size_t bad = 0, unstable = 0;
      for (size_t i = 1; i < n; i++) {
        if ((ok[i] >> 8) < (ok[i - 1] >> 8)) bad++;
        if ((ok[i] >> 8) == (ok[i - 1] >> 8) && ov[i] < ov[i - 1]) unstable++;
      }
      printf("n=%zu bits 8..%d: err=%d tmp=%zu out-of-order=%zu unstable=%zu\n", n, endb, (int)e, tmp, bad, unstable);
      hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(t);
    }
  return 0;
}
