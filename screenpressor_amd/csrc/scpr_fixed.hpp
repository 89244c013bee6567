// Fixed-alphabet chains of the encoder (FixedSizeRansCtx, ans_contexts.h:1054-1132): the six pixel-type models (keyed by the
// previous run's type) and the six run-length models (keyed by the run's type) of a generation, over the unified run list.
//
// A model's table only changes when its running total crosses the scale (incrCnt, :1070-1091), after a number of symbols that
// is known when the epoch starts: the symbols up to there are independent lookups, the rebuild is a wave prefix scan.  What
// the chain of ONE model needs is therefore its own symbols, densely, in stream order: round 2's k_fixed_chain walked the whole
// run list with every one of its twelve waves, 64 runs at a time, and kept the few its model was concerned with - on a one-GOP
// stream (one generation, 3 M runs) that was one workgroup for 79 ms.  Here the run list is first PARTITIONED by model, stably
// (count per block of 2048 runs, scan, scatter: two streaming passes), and a chain then takes 64 of its own symbols per trip,
// with the next trip's loads in flight.
//
//   k_fix_count     per block of FIX_B runs: how many belong to each of the twelve models
//   k_fix_scan      per model: exclusive scan of the block counts (+ the model's total)
//   k_fix_scatter   every run to its place in the list of its pixel-type model and of its run-length model: symbol, position
//   k_fix_genstart  where each generation's share of each model's list begins and ends
//   k_fixed_chain2  one workgroup per generation, one wave per model: the epoch-parallel chain over the model's own list
#pragma once
#include "scpr_wave.hpp"

namespace scpr {

constexpr int FIX_B = 2048;       // runs per partition block (256 threads x 8 rounds)
constexpr int FIX_CLASSES = 12;   // 0..5 pixel-type model keyed by the previous type, 6..11 run-length model keyed by the type

// the two models a run belongs to (-1: a header run codes no pixel type), its two symbols and their stream positions
struct FixRun {
  int ct, cn;
  u32 st, sn, pt, pn;
};
__device__ __forceinline__ FixRun fix_run(u32 r, u32 pos) {
  const int type = (int)(r & 7u), lastt = (int)((r >> 3) & 7u);
  const bool hdr = (r >> 31) != 0;
  FixRun f;
  f.ct = hdr ? -1 : lastt;
  f.cn = 6 + type;
  f.st = (u32)type;
  f.sn = (r >> 8) & 255u;
  f.pt = pos;
  f.pn = pos + (hdr ? 3u : (type == 0 ? 4u : 1u));  // a literal's three colour bytes sit between the type and the length
  return f;
}

__global__ __launch_bounds__(256) void k_fix_count(const u32* __restrict__ runs, u32 R, u32* __restrict__ blkcnt, u32 nblk) {
  __shared__ u32 cnt[FIX_CLASSES];
  if (threadIdx.x < FIX_CLASSES) cnt[threadIdx.x] = 0;
  __syncthreads();
  u32 mine[FIX_CLASSES];
#pragma unroll
  for (int c = 0; c < FIX_CLASSES; c++) mine[c] = 0;
  for (int it = 0; it < FIX_B / 256; it++) {
    const u32 i = blockIdx.x * FIX_B + it * 256 + threadIdx.x;
    const bool ok = i < R;
    const FixRun f = fix_run(ok ? runs[i] : 0x80000000u, 0);
#pragma unroll
    for (int c = 0; c < FIX_CLASSES; c++) {
      const u64 b = __ballot(ok && (c < 6 ? f.ct == c : f.cn == c));
      mine[c] += (u32)__builtin_popcountll(b);  // (the same in every lane of the wave)
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < FIX_CLASSES; c++)
      if (mine[c]) atomicAdd(&cnt[c], mine[c]);
  }
  __syncthreads();
  if (threadIdx.x < FIX_CLASSES) blkcnt[(size_t)threadIdx.x * (nblk + 1) + blockIdx.x] = cnt[threadIdx.x];
}

// blkoff[c][b] = runs of model c before block b (b = nblk: the model's total, also written to ctotal[c])
__global__ __launch_bounds__(256) void k_fix_scan(const u32* __restrict__ blkcnt, u32* __restrict__ blkoff, u32 nblk, u32* __restrict__ ctotal) {
  __shared__ u32 part[256];
  const int c = blockIdx.x, t = threadIdx.x;
  const u32* in = blkcnt + (size_t)c * (nblk + 1);
  u32* out = blkoff + (size_t)c * (nblk + 1);
  const u32 per = (nblk + 255) / 256, a = min(nblk, t * per), b = min(nblk, a + per);
  u32 s = 0;
  for (u32 i = a; i < b; i++) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const u32 v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  u32 run = part[t] - s;
  for (u32 i = a; i < b; i++) {
    const u32 v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 255) {
    out[nblk] = part[255];
    ctotal[c] = part[255];
  }
}

// Stable: a model's list holds its runs in run-list (= stream) order.  fl_sym / fl_pos: the twelve lists back to back.
__global__ __launch_bounds__(256) void k_fix_scatter(const u32* __restrict__ runs, const u32* __restrict__ runpos, u32 R, const u32* __restrict__ blkoff, u32 nblk,
                                                     const u32* __restrict__ ctotal, u8* __restrict__ fl_sym, u32* __restrict__ fl_pos) {
  __shared__ u32 base[FIX_CLASSES], wcnt[4][FIX_CLASSES];
  const int wv = threadIdx.x >> 6;
  if (threadIdx.x < FIX_CLASSES) {
    u32 cb = 0;
    for (int c = 0; c < (int)threadIdx.x; c++) cb += ctotal[c];
    base[threadIdx.x] = cb + blkoff[(size_t)threadIdx.x * (nblk + 1) + blockIdx.x];
  }
  __syncthreads();
  for (int it = 0; it < FIX_B / 256; it++) {
    const u32 i = blockIdx.x * FIX_B + it * 256 + threadIdx.x;
    const bool ok = i < R;
    const FixRun f = fix_run(ok ? runs[i] : 0x80000000u, ok ? runpos[i] : 0u);
    u32 rt = 0, rn = 0;
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int c = 0; c < FIX_CLASSES; c++) {
      const bool in = ok && (c < 6 ? f.ct == c : f.cn == c);
      const u64 b = __ballot(in);
      if (in) (c < 6 ? rt : rn) = (u32)__builtin_popcountll(b & lt);
      if ((threadIdx.x & 63) == 0) wcnt[wv][c] = (u32)__builtin_popcountll(b);
    }
    __syncthreads();
    if (ok) {
      if (f.ct >= 0) {
        u32 d = base[f.ct] + rt;
        for (int w = 0; w < wv; w++) d += wcnt[w][f.ct];
        fl_sym[d] = (u8)f.st;
        fl_pos[d] = f.pt;
      }
      u32 d = base[f.cn] + rn;
      for (int w = 0; w < wv; w++) d += wcnt[w][f.cn];
      fl_sym[d] = (u8)f.sn;
      fl_pos[d] = f.pn;
    }
    __syncthreads();
    if (threadIdx.x < FIX_CLASSES) base[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    __syncthreads();
  }
}

// gstart[(c * ngens + g) * 2 + {0, 1}]: first and one-past-last index of generation g's symbols in model c's part of the lists
__global__ __launch_bounds__(64) void k_fix_genstart(const u32* __restrict__ runs, u32 R, const GenRange* __restrict__ ranges, int ngens, const u32* __restrict__ blkoff, u32 nblk,
                                                     const u32* __restrict__ ctotal, u32* __restrict__ gstart) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const GenRange rg = ranges[g];
  for (int e = 0; e < 2; e++) {
    const u32 idx = min(e ? rg.run_end : rg.run_begin, R);
    const u32 blk = idx / FIX_B;
    u32 part[FIX_CLASSES];
#pragma unroll
    for (int c = 0; c < FIX_CLASSES; c++) part[c] = 0;
    for (u32 i0 = blk * FIX_B; i0 < idx; i0 += 64) {
      const u32 i = i0 + lane;
      const bool ok = i < idx;
      const FixRun f = fix_run(ok ? runs[i] : 0x80000000u, 0);
#pragma unroll
      for (int c = 0; c < FIX_CLASSES; c++) part[c] += (u32)__builtin_popcountll(__ballot(ok && (c < 6 ? f.ct == c : f.cn == c)));
    }
    if (lane < FIX_CLASSES) {
      u32 cb = 0, mine = 0;
      for (int c = 0; c < lane; c++) cb += ctotal[c];
#pragma unroll
      for (int c = 0; c < FIX_CLASSES; c++) mine = c == lane ? part[c] : mine;
      gstart[((size_t)lane * ngens + g) * 2 + e] = cb + blkoff[(size_t)lane * (nblk + 1) + blk] + mine;
    }
  }
}

// persist_in / persist_out: the tables kept between calls (P-frames continue the models, screencap.cpp:1118).  They are the same
// array when the call has one generation; with several, the first generation's workgroup reads while the last one's writes, so
// the host hands out two arrays and swaps them.
__global__ __launch_bounds__(768) void k_fixed_chain2(const u8* __restrict__ fl_sym, const u32* __restrict__ fl_pos, const u32* __restrict__ gstart, int ngens, int load_first,
                                                      const FixedPersist* persist /* [12] */, FixedPersist* persist_out, u32* __restrict__ entries) {
  // one wave per model; a wave owns its table: fc = freq | cum << 16 (what goes to the coder as it is), cnt; lanes of a wave
  // talk through LDS in program order (wavefront fences only, no barriers)
  __shared__ u32 tab[FIX_CLASSES][2][256];
  const int cls = threadIdx.x >> 6, gen = blockIdx.x, lane = threadIdx.x & 63;
  u32* fc = tab[cls][0];
  u32* cnt = tab[cls][1];
  const int nsym = cls >= 6 ? 256 : 6;
  int total;
  if (gen == 0 && load_first && persist[cls].valid) {
    for (int j = lane; j < nsym; j += 64) {
      fc[j] = persist[cls].freq[j] | (persist[cls].cum[j] << 16);
      cnt[j] = persist[cls].cnt[j];
    }
    total = persist[cls].total;
  } else {
    const int fr = kProbScale / nsym, c0 = fr - (fr >> 1);  // renew, ans_contexts.h:1114-1131
    for (int j = lane; j < nsym; j += 64) {
      fc[j] = (u32)fr | ((u32)(fr * j) << 16);
      cnt[j] = c0;
    }
    total = c0 * nsym;
  }
  total = (int)rfl((u32)total);
  wave_fence();
  const u32 s = rfl(gstart[((size_t)cls * ngens + gen) * 2]), e = rfl(gstart[((size_t)cls * ngens + gen) * 2 + 1]);
  // A trip is 256 symbols, four consecutive ones per lane (symbol i of the trip: lane i / 4, slot i % 4), and the loads of the
  // next TWO trips are in flight while one is coded: the list streams from HBM (a microsecond away), a trip is a few hundred
  // cycles of work, and with one trip of 64 in flight the chain spent its time waiting for its own input.
  auto ld_sym = [&](u32 at) __attribute__((always_inline)) -> u32 {  // four symbols (bytes) from list index `at` (any alignment)
    u32 v = 0;
    if (at + 4 <= e) {
      __builtin_memcpy(&v, fl_sym + at, 4);
    } else {
      for (u32 q = 0; q < 4 && at + q < e; q++) v |= (u32)fl_sym[at + q] << (8 * q);
    }
    return v;
  };
  auto ld_pos = [&](u32 at) __attribute__((always_inline)) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at + 4 <= e) {
      __builtin_memcpy(&v, fl_pos + at, 16);
    } else {
      if (at < e) v.x = fl_pos[at];
      if (at + 1 < e) v.y = fl_pos[at + 1];
      if (at + 2 < e) v.z = fl_pos[at + 2];
    }
    return v;
  };
  u32 sy1 = 0, sy2 = 0;
  uint4 po1 = make_uint4(0, 0, 0, 0), po2 = po1;
  if (s + 4u * lane < e) sy1 = ld_sym(s + 4u * lane), po1 = ld_pos(s + 4u * lane);
  if (s + 256u + 4u * lane < e) sy2 = ld_sym(s + 256u + 4u * lane), po2 = ld_pos(s + 256u + 4u * lane);
  for (u32 base = s; base < e; base += 256) {
    const u32 sy = sy1;
    const uint4 po = po1;
    sy1 = sy2, po1 = po2;
    if (base + 512u + 4u * lane < e) sy2 = ld_sym(base + 512u + 4u * lane), po2 = ld_pos(base + 512u + 4u * lane);
    const int cntm = (int)min(256u, e - base);
    const u32 pq[4] = {po.x, po.y, po.z, po.w};
    int done = 0;
    while (done < cntm) {
      const int room = (kProbScale - kStepDense - total) / kStepDense + 1;  // the room-th symbol from here brings the rebuild on
      const int take = min(room, cntm - done);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = 4 * lane + q;
        if (i >= done && i < done + take) {
          const u32 sym = (sy >> (8 * q)) & 255u;
          entries[pq[q]] = fc[sym];
          atomicAdd(&cnt[sym], (u32)kStepDense);
        }
      }
      total += kStepDense * take;
      done += take;
      wave_fence();
      if (take == room) {  // counts become the frequencies (incrCnt, ans_contexts.h:1075-1090)
        if (nsym == 256) {
          const uint4 cq = ((const uint4*)cnt)[lane];
          const int c0 = (int)cq.x, c1 = (int)cq.y, c2 = (int)cq.z, c3 = (int)cq.w, sum = c0 + c1 + c2 + c3;
          const int cf = wave_incl_scan(sum) - sum;
          ((uint4*)fc)[lane] = make_uint4((u32)c0 | ((u32)cf << 16), (u32)c1 | ((u32)(cf + c0) << 16), (u32)c2 | ((u32)(cf + c0 + c1) << 16), (u32)c3 | ((u32)(cf + c0 + c1 + c2) << 16));
          const int h0 = c0 - (c0 >> 1), h1 = c1 - (c1 >> 1), h2 = c2 - (c2 >> 1), h3 = c3 - (c3 >> 1);
          ((uint4*)cnt)[lane] = make_uint4((u32)h0, (u32)h1, (u32)h2, (u32)h3);
          total = wave_sum(h0 + h1 + h2 + h3);
        } else {
          const int c = lane < 6 ? (int)cnt[lane] : 0;
          const int inc = row_incl_scan(c);  // (six entries: inside the first row of sixteen lanes)
          const int h = c - (c >> 1);
          if (lane < 6) {
            fc[lane] = (u32)c | ((u32)(inc - c) << 16);
            cnt[lane] = (u32)h;
          }
          total = row16_sum(h);
        }
        wave_fence();
      }
    }
  }
  if (gen == ngens - 1) {  // the last generation of the call is the live one
    wave_fence();
    for (int j = lane; j < nsym; j += 64) {
      persist_out[cls].freq[j] = fc[j] & 0xFFFFu;
      persist_out[cls].cum[j] = fc[j] >> 16;
      persist_out[cls].cnt[j] = cnt[j];
    }
    if (lane == 0) {
      persist_out[cls].total = total;
      persist_out[cls].valid = 1;
    }
  }
}

}  // namespace scpr
