// scpr_ctxsort.hpp - the colour symbols of a chunk, grouped by context in stream order (what the colour chains read).
//
// Every literal pixel yields three (context, byte) pairs, one per colour plane (EncodeRGB, screencap.cpp:631-643; contexts per
// screencap.h:35-36: 12 bits from the two bytes before, 4096 per plane).  A colour chain is one (generation, plane, context)
// and needs its own bytes in stream order.  Rounds 1-4 sorted (generation, context) keys with rocPRIM's radix sort: 23 key bits,
// three passes, a library kernel with a wrong-output defect that had to be fenced.  This is the repo's own, smaller problem:
//  * the producers (k_symbols, k_pemit) already write a generation's pairs plane by plane (three arrays of L pairs, L = the
//    generation's literals), in stream order - generation and plane need no sorting at all;
//  * what is left is a STABLE partition of each (generation, plane) SEGMENT by its 12 context bits: two counting passes of six
//    bits (64 digits = the 64 lanes of a wave), least significant first.
// Per pass: k_cs_count (digit counts per block of 4096 pairs), k_cs_scan (one wave per segment: lane d runs digit d over the
// segment's blocks), k_cs_scatter (every pair to its place).  Nothing waits on another workgroup; ranks inside a wave come from
// six ballots (the lanes whose digit equals mine, below me), never from the order in which atomics happen to be served: the
// result is deterministic and stable by construction.  k_chain_starts (scpr_kernels.hpp) still proves the order afterwards.
#pragma once
#include "scpr_wave.hpp"

namespace scpr {

constexpr int CS_B = 4096;      // pairs per block: 4 waves x 16 rounds x 64 lanes; wave w takes the block's w-th quarter in order
constexpr int CS_ROUNDS = 16;
struct CsBlock {
  u32 begin, len, seg, pad;  // pairs [begin, begin + len) of the arrays, all of one segment
};
struct CsSeg {
  u32 blk_begin, blk_end, base, pad;  // the segment's blocks; base: where its pairs begin (in and out)
};

// the lanes (of the valid ones, `m0`) whose six-bit digit equals x, from the six ballots of the digit's bits
__device__ __forceinline__ u64 cs_match(const u64 (&b)[6], u64 m0, u32 x) {
  u64 m = m0;
#pragma unroll
  for (int k = 0; k < 6; k++) m &= ((x >> k) & 1u) ? b[k] : ~b[k];
  return m;
}
__device__ __forceinline__ void cs_ballots(u32 d, bool valid, u64 (&b)[6], u64& m0) {
  m0 = __ballot(valid);
#pragma unroll
  for (int k = 0; k < 6; k++) b[k] = __ballot(valid && ((d >> k) & 1u));
}

// digit counts of a block: blkcnt[block * 64 + d]
template <int SHIFT>
__global__ __launch_bounds__(256) void k_cs_count(const u32* __restrict__ keys, const CsBlock* __restrict__ blocks, u32* __restrict__ blkcnt) {
  __shared__ u32 wcnt[4][64];
  const CsBlock blk = blocks[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32* k = keys + blk.begin;
  u32 cnt = 0;  // of digit `lane` in this wave's quarter
#pragma unroll 4
  for (int r = 0; r < CS_ROUNDS; r++) {
    const u32 i = (u32)(w * CS_ROUNDS + r) * 64u + (u32)lane;
    const bool valid = i < blk.len;
    const u32 d = valid ? (k[i] >> SHIFT) & 63u : 0u;
    u64 b[6], m0;
    cs_ballots(d, valid, b, m0);
    cnt += (u32)__popcll(cs_match(b, m0, (u32)lane));
  }
  wcnt[w][lane] = cnt;
  __syncthreads();
  if (w == 0) blkcnt[(size_t)blockIdx.x * 64 + lane] = wcnt[0][lane] + wcnt[1][lane] + wcnt[2][lane] + wcnt[3][lane];
}

// one wave per segment, lane d = digit d: blkoff[block * 64 + d] = where the block's pairs of digit d go
__global__ __launch_bounds__(64) void k_cs_scan(const CsSeg* __restrict__ segs, const u32* __restrict__ blkcnt, u32* __restrict__ blkoff) {
  const CsSeg s = segs[blockIdx.x];
  const int lane = threadIdx.x;
  u32 run = 0;
  for (u32 b = s.blk_begin; b < s.blk_end; b++) {
    const u32 c = blkcnt[(size_t)b * 64 + lane];
    blkoff[(size_t)b * 64 + lane] = run;
    run += c;
  }
  const u32 add = s.base + (u32)(wave_incl_scan((int)run) - (int)run);  // digits below mine, in the whole segment
  for (u32 b = s.blk_begin; b < s.blk_end; b++) blkoff[(size_t)b * 64 + lane] += add;
}

// (registers: the second pass, whose writes go to 64 far-apart places a round, is faster with four waves to a SIMD and a few
// registers fewer - 0.79 -> 0.61 ms -, the first, whose neighbours mostly share a digit, with the three it gets by itself: 0.33 against 0.40)
template <int SHIFT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SHIFT == 14 ? 4 : 3, SHIFT == 14 ? 4 : 3))) void k_cs_scatter(const u32* __restrict__ keys, const u32* __restrict__ vals, const CsBlock* __restrict__ blocks, const u32* __restrict__ blkoff,
                                                   u32* __restrict__ keys_out, u32* __restrict__ vals_out) {
  __shared__ u32 wcnt[4][64];
  const CsBlock blk = blocks[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32* k = keys + blk.begin;
  const u32* v = vals + blk.begin;
  u32 kk[CS_ROUNDS], vv[CS_ROUNDS];
#pragma unroll
  for (int r = 0; r < CS_ROUNDS; r++) {
    const u32 i = (u32)(w * CS_ROUNDS + r) * 64u + (u32)lane;
    const bool valid = i < blk.len;
    kk[r] = valid ? k[i] : 0u;
    vv[r] = valid ? v[i] : 0u;
  }
  u32 cnt = 0;
#pragma unroll
  for (int r = 0; r < CS_ROUNDS; r++) {
    const u32 i = (u32)(w * CS_ROUNDS + r) * 64u + (u32)lane;
    u64 b[6], m0;
    cs_ballots((kk[r] >> SHIFT) & 63u, i < blk.len, b, m0);
    cnt += (u32)__popcll(cs_match(b, m0, (u32)lane));
  }
  wcnt[w][lane] = cnt;
  __syncthreads();
  // lane d: where the next pair of digit d of THIS wave's quarter goes
  u32 cur = blkoff[(size_t)blockIdx.x * 64 + lane];
#pragma unroll
  for (int q = 0; q < 3; q++) cur += q < w ? wcnt[q][lane] : 0u;
  const u64 below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < CS_ROUNDS; r++) {
    const u32 i = (u32)(w * CS_ROUNDS + r) * 64u + (u32)lane;
    const bool valid = i < blk.len;
    const u32 d = (kk[r] >> SHIFT) & 63u;
    u64 b[6], m0;
    cs_ballots(d, valid, b, m0);
    const u32 rank = (u32)__popcll(cs_match(b, m0, d) & below);
    const u32 base = (u32)__builtin_amdgcn_ds_bpermute((int)(d << 2), (int)cur);  // cur of lane d
    if (valid) {
      keys_out[base + rank] = kk[r];
      vals_out[base + rank] = vv[r];
    }
    cur += (u32)__popcll(cs_match(b, m0, (u32)lane));
  }
}

}  // namespace scpr
