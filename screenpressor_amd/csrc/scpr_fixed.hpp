// Fixed-alphabet chains of the encoder (FixedSizeRansCtx, ans_contexts.h:1054-1132): the six pixel-type models (keyed by the
// previous run's type) and the six run-length models (keyed by the run's type) of a generation over the unified run list, and
// the nine P-frame models (block index bytes, block-type runs, block types, four rect coordinates, two vector components) over
// the list of P-frame symbols.
//
// A model's table only changes when its running total crosses the scale (incrCnt, :1070-1091), after a number of symbols that
// is known when the epoch starts: the symbols up to there are independent lookups, the rebuild is a wave prefix scan.  What
// the chain of ONE model needs is therefore its own symbols, densely, in stream order: round 2's chains walked the whole list
// with every one of their waves, 64 elements at a time, and kept the few their model was concerned with - on a one-GOP stream
// (one generation, 3 M runs, 7 M P-frame symbols) that was one workgroup for 79 ms.  Here the lists are first PARTITIONED by
// model, stably (count per block of 2048 elements, scan, scatter: two streaming passes, every thread over eight consecutive
// elements, two barriers per block), and a chain then takes 256 of its own symbols per trip with the next trips' loads in flight.
//
//   k_part_count     per block of PART_B elements: how many items belong to each model
//   k_part_scan      per model: exclusive scan of the block counts (+ the model's total)
//   k_part_scatter   every item to its place in its model's list: symbol, stream position
//   k_part_genstart  where each generation's share of each model's list begins and ends
//   k_fixed_chain2   one workgroup per generation, one wave per model: the epoch-parallel chain over the model's own list
#pragma once
#include "scpr_wave.hpp"
#include "scpr_inter.hpp"

namespace scpr {

constexpr int PART_B = 2048;  // elements per partition block (256 threads x 8 consecutive elements)

// What a list element contributes: up to two items (model, symbol, stream position); model -1: none.
struct RunItems {  // runs[i], runpos[i]: the run's pixel type goes to the model of the previous type (a header run codes none),
                   // its length to the model of its own type
  static constexpr int NCLS = 12;
  static __device__ __forceinline__ void get(u32 r, u32 pos, int& c0, u32& s0, u32& p0, int& c1, u32& s1, u32& p1) {
    const int type = (int)(r & 7u), lastt = (int)((r >> 3) & 7u);
    const bool hdr = (r >> 31) != 0;
    c0 = hdr ? -1 : lastt;
    s0 = (u32)type;
    p0 = pos;
    c1 = 6 + type;
    s1 = (r >> 8) & 255u;
    p1 = pos + (hdr ? 3u : (type == 0 ? 4u : 1u));  // a literal's three colour bytes sit between the type and the length
  }
  static __device__ __forceinline__ int nsym(int cls) { return cls >= 6 ? 256 : 6; }
};
struct MiscItems {  // misc[i] = model << 16 | symbol, miscpos[i]
  static constexpr int NCLS = MC_COUNT;
  static __device__ __forceinline__ void get(u32 v, u32 pos, int& c0, u32& s0, u32& p0, int& c1, u32& s1, u32& p1) {
    c0 = (int)(v >> 16);
    s0 = v & 0xFFFFu;
    p0 = pos;
    c1 = -1;
    s1 = p1 = 0;
  }
  static __device__ __forceinline__ int nsym(int cls) { return cls == MC_BT ? 5 : (cls >= MC_SXY && cls < MC_SXY + 4) ? 16 : (cls >= MC_MX) ? 512 : 256; }
};

// per-model counters of a thread (or, summed, of a wave: at most 512): 10-bit fields, three to a word
template <int NW>
__device__ __forceinline__ void fld_add(u32 (&w)[NW], int c) {
  const int k = (c * 11) >> 5, sh = 10 * (c - 3 * k);  // c / 3 for c < 12
#pragma unroll
  for (int q = 0; q < NW; q++) w[q] += q == k ? 1u << sh : 0u;
}
template <int NW>
__device__ __forceinline__ u32 fld_get(const u32 (&w)[NW], int c) {
  const int k = (c * 11) >> 5, sh = 10 * (c - 3 * k);
  u32 v = 0;
#pragma unroll
  for (int q = 0; q < NW; q++) v = q == k ? w[q] : v;
  return (v >> sh) & 1023u;
}

template <class SRC>
__global__ __launch_bounds__(256) void k_part_count(const u32* __restrict__ el, u32 n, u32* __restrict__ blkcnt, u32 nblk) {
  constexpr int NC = SRC::NCLS, NW = (NC + 2) / 3;
  __shared__ u32 cnt[NC];
  if (threadIdx.x < NC) cnt[threadIdx.x] = 0;
  __syncthreads();
  u32 w[NW];
#pragma unroll
  for (int q = 0; q < NW; q++) w[q] = 0;
  const u32 i0 = blockIdx.x * PART_B + threadIdx.x * 8;
  if (i0 < n) {
    u32 v[8];
    if (i0 + 8 <= n) {
      *(uint4*)&v[0] = *(const uint4*)(el + i0);
      *(uint4*)&v[4] = *(const uint4*)(el + i0 + 4);
    } else {
      for (int q = 0; q < 8; q++) v[q] = i0 + q < n ? el[i0 + q] : 0;
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (i0 + q < n) {
        int c0, c1;
        u32 s0, p0, s1, p1;
        SRC::get(v[q], 0, c0, s0, p0, c1, s1, p1);
        if (c0 >= 0) fld_add<NW>(w, c0);
        if (c1 >= 0) fld_add<NW>(w, c1);
      }
  }
#pragma unroll
  for (int q = 0; q < NW; q++) w[q] = (u32)wave_sum((int)w[q]);  // (fields stay below 1024: 64 lanes x 8)
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const u32 k = fld_get<NW>(w, c);
      if (k) atomicAdd(&cnt[c], k);
    }
  }
  __syncthreads();
  if (threadIdx.x < NC) blkcnt[(size_t)threadIdx.x * (nblk + 1) + blockIdx.x] = cnt[threadIdx.x];
}

// blkoff[c][b] = items of model c before block b (b = nblk: the model's total, also written to ctotal[c])
constexpr int SCAN_T = 1024;
__global__ __launch_bounds__(SCAN_T) void k_part_scan(const u32* __restrict__ blkcnt, u32* __restrict__ blkoff, u32 nblk, u32* __restrict__ ctotal) {
  // tiles of SCAN_T counts, read and written side by side (round 5: a thread used to walk its own run of ~32 counts, 128 bytes
  // apart from its neighbour's - 0.46 ms for the headline's 33 000 blocks per model, on the longer of the encoder's two branches)
  // (eight counts per thread and round: the headline's 33 000 blocks per model are five rounds - each round is a trip to memory
  // and two barriers, and the kernel runs beside the colour partition, which leaves it little of the machine)
  constexpr int V = 8;
  __shared__ u32 wtot[SCAN_T / 64];
  const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const u32* in = blkcnt + (size_t)c * (nblk + 1);
  u32* out = blkoff + (size_t)c * (nblk + 1);
  u32 carry = 0;
  for (u32 base = 0; base < nblk; base += SCAN_T * V) {
    const u32 i0 = base + (u32)t * V;
    u32 v[V], sum = 0;
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = i0 + k < nblk ? in[i0 + k] : 0u;
#pragma unroll
    for (int k = 0; k < V; k++) sum += v[k];
    const u32 incl = (u32)wave_incl_scan((int)sum);
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < SCAN_T / 64; q++) {
      const u32 x = wtot[q];
      before += q < w ? x : 0u;
      all += x;
    }
    u32 run = carry + before + incl - sum;
#pragma unroll
    for (int k = 0; k < V; k++) {
      if (i0 + k < nblk) out[i0 + k] = run;
      run += v[k];
    }
    carry += all;
    __syncthreads();
  }
  if (t == 0) {
    out[nblk] = carry;
    ctotal[c] = carry;
  }
}

// Stable: a model's list holds its items in list (= stream) order.  fl_sym / fl_pos: the models' lists back to back.
// A block's items are first laid out in LDS, model after model (a model's share of the block is one contiguous piece of its
// list), and then written out side by side: a thread's own items go to a dozen places two and four bytes at a time, which left
// the L2 as one 32-byte write per store (PMC: 6.0 GB for 1.4 GB of lists on the headline).
template <class SRC>
__global__ __launch_bounds__(256) void k_part_scatter(const u32* __restrict__ el, const u32* __restrict__ elpos, u32 n, const u32* __restrict__ blkoff, u32 nblk,
                                                      const u32* __restrict__ ctotal, u16* __restrict__ fl_sym, u32* __restrict__ fl_pos) {
  constexpr int NC = SRC::NCLS, NW = (NC + 2) / 3;
  constexpr int CAP = 2 * PART_B;  // items of a block at most (two per element)
  __shared__ u32 wtot[4][NW], wb[4][NC], lstart[NC + 1], gbase[NC];
  __shared__ u32 spos[CAP];
  __shared__ u16 ssym[CAP];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 i0 = blockIdx.x * PART_B + threadIdx.x * 8;
  u32 v[8], ps[8];
#pragma unroll
  for (int q = 0; q < 8; q++) v[q] = ps[q] = 0;
  if (i0 + 8 <= n) {
    *(uint4*)&v[0] = *(const uint4*)(el + i0);
    *(uint4*)&v[4] = *(const uint4*)(el + i0 + 4);
    *(uint4*)&ps[0] = *(const uint4*)(elpos + i0);
    *(uint4*)&ps[4] = *(const uint4*)(elpos + i0 + 4);
  } else {
    for (int q = 0; q < 8; q++)
      if (i0 + q < n) v[q] = el[i0 + q], ps[q] = elpos[i0 + q];
  }
  u32 w[NW];
#pragma unroll
  for (int q = 0; q < NW; q++) w[q] = 0;
#pragma unroll
  for (int q = 0; q < 8; q++)
    if (i0 + q < n) {
      int c0, c1;
      u32 s0, p0, s1, p1;
      SRC::get(v[q], ps[q], c0, s0, p0, c1, s1, p1);
      if (c0 >= 0) fld_add<NW>(w, c0);
      if (c1 >= 0) fld_add<NW>(w, c1);
    }
  u32 pre[NW];  // items of each model in the lanes before this one (of the wave)
#pragma unroll
  for (int q = 0; q < NW; q++) {
    const u32 inc = (u32)wave_incl_scan((int)w[q]);
    pre[q] = inc - w[q];
    if (lane == 63) wtot[wv][q] = inc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // where each model's share of the block begins in LDS, and each wave's share of that
    u32 at = 0;
    for (int c = 0; c < NC; c++) {
      lstart[c] = at;
      for (int q = 0; q < 4; q++) {
        wb[q][c] = at;
        u32 t[NW];
#pragma unroll
        for (int z = 0; z < NW; z++) t[z] = wtot[q][z];
        at += fld_get<NW>(t, c);
      }
    }
    lstart[NC] = at;
  }
  if (threadIdx.x < NC) {  // ... and in the lists: the model's place + the blocks before
    const int c = threadIdx.x;
    u32 g = blkoff[(size_t)c * (nblk + 1) + blockIdx.x];
    for (int k = 0; k < c; k++) g += ctotal[k];
    gbase[c] = g;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; q++)
    if (i0 + q < n) {
      int c0, c1;
      u32 s0, p0, s1, p1;
      SRC::get(v[q], ps[q], c0, s0, p0, c1, s1, p1);
      if (c0 >= 0) {
        const u32 d = wb[wv][c0] + fld_get<NW>(pre, c0);
        ssym[d] = (u16)s0;
        spos[d] = p0;
        fld_add<NW>(pre, c0);
      }
      if (c1 >= 0) {
        const u32 d = wb[wv][c1] + fld_get<NW>(pre, c1);
        ssym[d] = (u16)s1;
        spos[d] = p1;
        fld_add<NW>(pre, c1);
      }
    }
  __syncthreads();
  const u32 total = lstart[NC];
  for (u32 i = threadIdx.x; i < total; i += 256) {  // neighbouring threads, neighbouring items of (mostly) the same list
    int c = 0;
#pragma unroll
    for (int k = 1; k < NC; k++) c += i >= lstart[k] ? 1 : 0;
    const u32 d = gbase[c] + (i - lstart[c]);
    fl_sym[d] = ssym[i];
    fl_pos[d] = spos[i];
  }
}

// gstart[(c * ngens + g) * 2 + {0, 1}]: first and one-past-last index of generation g's items in model c's part of the lists.
// ranges: the generation's [begin, end) in the element list (GenRange / MiscRange: two words).
template <class SRC>
__global__ __launch_bounds__(64) void k_part_genstart(const u32* __restrict__ el, u32 n, const uint2* __restrict__ ranges, int ngens, const u32* __restrict__ blkoff, u32 nblk,
                                                      const u32* __restrict__ ctotal, u32* __restrict__ gstart) {
  constexpr int NC = SRC::NCLS, NW = (NC + 2) / 3;
  const int g = blockIdx.x, lane = threadIdx.x;
  const uint2 rg = ranges[g];
  for (int e = 0; e < 2; e++) {
    const u32 idx = min(e ? rg.y : rg.x, n);
    const u32 blk = idx / PART_B;
    u32 w[NW];  // this lane's 32 consecutive elements of the block, as far as they lie before idx (a field stays below 64)
#pragma unroll
    for (int q = 0; q < NW; q++) w[q] = 0;
    const u32 i0 = blk * PART_B + (u32)lane * 32u;
    for (u32 i = i0; i < min(idx, i0 + 32u); i++) {
      int c0, c1;
      u32 s0, p0, s1, p1;
      SRC::get(el[i], 0, c0, s0, p0, c1, s1, p1);
      if (c0 >= 0) fld_add<NW>(w, c0);
      if (c1 >= 0) fld_add<NW>(w, c1);
    }
    u32 mine = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const u32 t = (u32)wave_sum((int)fld_get<NW>(w, c));
      mine = c == lane ? t : mine;
    }
    if (lane < NC) {
      u32 cb = 0;
      for (int c = 0; c < lane; c++) cb += ctotal[c];
      gstart[((size_t)lane * ngens + g) * 2 + e] = cb + blkoff[(size_t)lane * (nblk + 1) + blk] + mine;
    }
  }
}

// persist_in / persist_out: the tables kept between calls (P-frames continue the models, screencap.cpp:1118).  They are the same
// array when the call has one generation; with several, the first generation's workgroup reads while the last one's writes, so
// the host hands out two arrays and swaps them.  Dynamic LDS: 2 * MAXSYM words per wave.
template <class SRC, int MAXSYM>
__global__ __launch_bounds__(64 * SRC::NCLS) void k_fixed_chain2(const u16* __restrict__ fl_sym, const u32* __restrict__ fl_pos, const u32* __restrict__ gstart, int ngens, int load_first,
                                                                 const FixedPersist* persist /* [NCLS] */, FixedPersist* persist_out, u32* __restrict__ entries, u32 window) {
  // one wave per model; a wave owns its table: fc = freq | cum << 16 (what goes to the coder as it is), cnt; lanes of a wave
  // talk through LDS in program order (wavefront fences only, no barriers)
  extern __shared__ __align__(16) u32 chain_lds[];
  // The waves of a generation write into the same stretch of the coder's entry array, each its own model's entries: a 4-byte
  // store here, another model's next to it much later - left to themselves the twelve fronts drift apart by megabytes and the L2
  // hands every store on as a 32-byte write by itself (4.5 GB for 134 M entries, profiles/).  They go in step instead: a wave
  // publishes the stream position of the first symbol of the trip it is about to code, and starts that trip only when it lies
  // within a window of the hindmost front - so that what the L2 holds of the entry array is a few hundred kilobytes per
  // generation and a line has collected its models' words before it leaves.  The hindmost wave never waits (no deadlock); a
  // wave that is done publishes the end of positions.
  __shared__ u32 front[SRC::NCLS];
  const int cls = threadIdx.x >> 6, gen = blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x < SRC::NCLS) front[threadIdx.x] = 0u;
  __syncthreads();
  u32* fc = chain_lds + (size_t)cls * 2 * MAXSYM;
  u32* cnt = fc + MAXSYM;
  const int nsym = SRC::nsym(cls);
  const int per = nsym >= 64 ? nsym >> 6 : 1;  // table entries per lane in a rebuild (lanes past the alphabet hold none)
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
  // next two trips are in flight while one is coded (the list streams from HBM).
  auto ld_sym = [&](u32 at) __attribute__((always_inline)) -> uint2 {  // four 16-bit symbols from list index `at` (any alignment)
    uint2 v = make_uint2(0, 0);
    if (at + 4 <= e) {
      __builtin_memcpy(&v, fl_sym + at, 8);
    } else {
      u32 t[4] = {0, 0, 0, 0};
      for (u32 q = 0; q < 4 && at + q < e; q++) t[q] = fl_sym[at + q];
      v.x = t[0] | (t[1] << 16);
      v.y = t[2] | (t[3] << 16);
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
  uint2 sy1 = make_uint2(0, 0), sy2 = sy1;
  uint4 po1 = make_uint4(0, 0, 0, 0), po2 = po1;
  if (s + 4u * lane < e) sy1 = ld_sym(s + 4u * lane), po1 = ld_pos(s + 4u * lane);
  if (s + 256u + 4u * lane < e) sy2 = ld_sym(s + 256u + 4u * lane), po2 = ld_pos(s + 256u + 4u * lane);
  for (u32 base = s; base < e; base += 256) {
    const uint2 sy = sy1;
    const uint4 po = po1;
    if (window) {  // (0: every wave at its own pace)
      const u32 first = rfl(po.x);  // lane 0, slot 0: the trip's first symbol
      if (lane == 0) __hip_atomic_store(&front[cls], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (;;) {
        u32 f = lane < SRC::NCLS ? lds_peek(&front[lane]) : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) f = min(f, (u32)__shfl_xor((int)f, d));  // (NCLS <= 16: the first row of lanes)
        const u32 hind = rfl(f);
        if (first - hind <= window) break;  // (first >= hind: this wave's own front is among them)
        __builtin_amdgcn_s_sleep(4);
      }
    }
    sy1 = sy2, po1 = po2;
    if (base + 512u + 4u * lane < e) sy2 = ld_sym(base + 512u + 4u * lane), po2 = ld_pos(base + 512u + 4u * lane);
    const int cntm = (int)min(256u, e - base);
    const u32 pq[4] = {po.x, po.y, po.z, po.w};
    const u32 sq[4] = {sy.x & 0xFFFFu, sy.x >> 16, sy.y & 0xFFFFu, sy.y >> 16};
    int done = 0;
    while (done < cntm) {
      const int room = (kProbScale - kStepDense - total) / kStepDense + 1;  // the room-th symbol from here brings the rebuild on
      const int take = min(room, cntm - done);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = 4 * lane + q;
        if (i >= done && i < done + take) {
          entries[pq[q]] = fc[sq[q]];
          atomicAdd(&cnt[sq[q]], (u32)kStepDense);
        }
      }
      total += kStepDense * take;
      done += take;
      wave_fence();
      if (take == room) {  // counts become the frequencies (incrCnt, ans_contexts.h:1075-1090)
        int c[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int j = lane * per + q;
          c[q] = (q < per && j < nsym) ? (int)cnt[j] : 0;
          sum += c[q];
        }
        int cf = wave_incl_scan(sum) - sum, ns = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int j = lane * per + q;
          if (q < per && j < nsym) {
            fc[j] = (u32)c[q] | ((u32)cf << 16);
            cf += c[q];
            const int h = c[q] - (c[q] >> 1);
            cnt[j] = (u32)h;
            ns += h;
          }
        }
        total = wave_sum(ns);
        wave_fence();
      }
    }
  }
  if (lane == 0) __hip_atomic_store(&front[cls], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
