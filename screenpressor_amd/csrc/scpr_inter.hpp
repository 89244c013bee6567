// P-frame analysis and symbol emission (CompressP, screencap.cpp:1091-1271).
//
//   k_pblocks    per 16x16 block: compare with the previous plane, bounding box of the
//                changed pixels (DecideBlockTypes :985-1039); 4 blocks per wave, one row per lane
//   k_mvsearch   per changed block: the fixed-order exact-match search of FindMV (:737-811),
//                64 candidates per step, first hit by ballot
//   k_mvresolve  the serial part of FindMV (:715-735): "last found vector" and "vector of the
//                block above" are tried first and chain through the blocks in raster order
//                and, through mvs[], through the frames; blocks are resolved 64 at a time
//                under the current last vector and re-tested only after it changes
//   k_pcount     per pixel-coded block: greedy predictor runs over the changed rect (:1043-1065)
//   k_pscan      per frame: stream offsets of every block, block-type run lengths (:1155-1169)
//   k_pemit      per block: rect / motion / run symbols at their stream positions (:1179-1248)
#pragma once
#include "scpr_kernels.hpp"

namespace scpr {

// binfo: bit0 changed, bits 4-7 sx1, 8-11 sy1, 12-15 sx2 (inclusive), 16-19 sy2 (inclusive), 20-22 type (1 whole, 2 partial)
__device__ __forceinline__ u32 binfo_pack(int sx1, int sy1, int sx2, int sy2, int type) {
  return 1u | ((u32)sx1 << 4) | ((u32)sy1 << 8) | ((u32)sx2 << 12) | ((u32)sy2 << 16) | ((u32)type << 20);
}
struct Rect {
  int x1, y1, x2, y2;  // absolute, exclusive end
};
__device__ __forceinline__ Rect binfo_rect(u32 v, int bx, int by) {
  Rect r;
  r.x1 = bx * 16 + (int)((v >> 4) & 15);
  r.y1 = by * 16 + (int)((v >> 8) & 15);
  r.x2 = bx * 16 + (int)((v >> 12) & 15) + 1;
  r.y2 = by * 16 + (int)((v >> 16) & 15) + 1;
  return r;
}
// "no vector" in the per-frame dictionary: not a packed vector (components are within +-256; (-1,-1) packs to 0xFFFFFFFF)
constexpr u32 MV_NONE = 0x7FFF7FFFu;
__device__ __forceinline__ u32 mv_pack(int dx, int dy) { return ((u32)(dx & 0xFFFF)) | ((u32)(dy & 0xFFFF) << 16); }
__device__ __forceinline__ int mv_x(u32 v) { return (int)(int16_t)(v & 0xFFFF); }
__device__ __forceinline__ int mv_y(u32 v) { return (int)(int16_t)(v >> 16); }

struct PFrame {
  int slot, prev_slot;
};

// gmask[pi][group]: which blocks of each group of G = min(64, blocks per row) consecutive blocks changed (the serial
// resolution pass below only visits groups with a bit set)
__global__ __launch_bounds__(64) void k_pblocks(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, u32* __restrict__ binfo, u32* __restrict__ pflag,
                                                unsigned long long* __restrict__ gmask) {
  const int pi = blockIdx.y, lane = threadIdx.x;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const int b = blockIdx.x * 4 + (lane >> 4), row = lane & 15;
  const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
  const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
  const bool vb = b < nblocks;
  const int by = vb ? b / nbx : 0, bx = vb ? b - by * nbx : 0;
  const int y = by * 16 + row, bw = min(16, g.W - bx * 16), bh = min(16, g.H - by * 16);
  int first = 64, last = -1;
  if (vb && y < g.H) {
    const u32* c = (const u32*)(cur + (size_t)y * g.S + bx * 48);
    const u32* p = (const u32*)(prv + (size_t)y * g.S + bx * 48);
    const int nbytes = bw * 3;
    for (int k = 0; k * 4 < nbytes; k++) {
      u32 d = c[k] ^ p[k];
      const int rem = nbytes - k * 4;
      if (rem < 4) d &= (1u << (8 * rem)) - 1u;
      if (d) {
        first = min(first, k * 4 + (__builtin_ctz(d) >> 3));
        last = max(last, k * 4 + ((31 - __builtin_clz(d)) >> 3));
      }
    }
  }
  const bool rc = last >= 0;
  const u64 m = __ballot(rc);
  const u32 rows = (u32)(m >> (16 * (lane >> 4))) & 0xFFFFu;
  int fx = rc ? first / 3 : 64, lx = rc ? last / 3 : -1;
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    fx = min(fx, __shfl_xor(fx, d, 16));
    lx = max(lx, __shfl_xor(lx, d, 16));
  }
  if (vb && row == 0) {
    u32 v = 0;
    if (rows) {
      const int sy1 = __builtin_ctz(rows), sy2 = 31 - __builtin_clz(rows);
      const int type = (fx > 0 || sy1 > 0 || lx < bw - 1 || sy2 < bh - 1) ? 2 : 1;
      v = binfo_pack(fx, sy1, lx, sy2, type);
      if (pflag[pi] == 0) atomicOr(&pflag[pi], 1u);
      const int G = min(64, nbx), NG = (nblocks + G - 1) / G;
      atomicOr(&gmask[(size_t)pi * NG + b / G], 1ull << (b % G));
    }
    binfo[(size_t)pi * nblocks + b] = v;
  }
}

// SameBlocks (screencap.cpp:817-825): rect of the current plane at (x1,y1) == rect of prev at (x,y)
__device__ __forceinline__ bool same_rect(const u8* cur, const u8* prv, int S, const Rect& r, int x, int y) {
  const int wb = (r.x2 - r.x1) * 3, h = r.y2 - r.y1;
  const u8* a = cur + (size_t)r.y1 * S + r.x1 * 3;
  const u8* b = prv + (size_t)y * S + x * 3;
  for (int q = 0; q < h; q++) {
    int k = 0;
    for (; k + 4 <= wb; k += 4) {
      u32 va, vb;
      __builtin_memcpy(&va, a + k, 4);
      __builtin_memcpy(&vb, b + k, 4);
      if (va != vb) return false;
    }
    for (; k < wb; k++)
      if (a[k] != b[k]) return false;
    a += S;
    b += S;
  }
  return true;
}

// search windows of FindMV (:691-709)
struct Windows {
  int rx1, rx2, ry1, ry2, fx1, fx2, fy1, fy2;
};
__device__ __forceinline__ Windows mv_windows(const Rect& r, const Geom& g, int far_x, int far_y, int near_x, int near_y) {
  Windows w;
  const int dw = r.x2 - r.x1, dh = r.y2 - r.y1;
  w.rx1 = max(r.x1 - near_x, 0);
  w.ry1 = max(r.y1 - near_y, 0);
  w.rx2 = r.x1 + near_x;
  w.ry2 = r.y1 + near_y;
  if (w.rx2 + dw > g.W) w.rx2 = g.W - dw + 1;
  if (w.ry2 + dh > g.H) w.ry2 = g.H - dh + 1;
  w.fx1 = max(r.x1 - far_x, 0);
  w.fy1 = max(r.y1 - far_y, 0);
  w.fx2 = r.x1 + far_x;
  w.fy2 = r.y1 + far_y;
  if (w.fx2 + dw > g.W) w.fx2 = g.W - dw + 1;
  if (w.fy2 + dh > g.H) w.fy2 = g.H - dh + 1;
  return w;
}

struct MvParams {
  int far_x, far_y, near_x, near_y;
};

// One wave per changed block: candidates in the order of FindMV's search loops (:737-811).
// smv: bit31 found, low bits mv_pack.
__global__ __launch_bounds__(64) void k_mvsearch(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, const u32* __restrict__ binfo, MvParams mp,
                                                 u32* __restrict__ smv) {
  const int pi = blockIdx.y, b = blockIdx.x, lane = threadIdx.x;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const u32 info = binfo[(size_t)pi * nblocks + b];
  if (!(info & 1u)) return;
  const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
  const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
  const int by = b / nbx, bx = b - by * nbx;
  const Rect r = binfo_rect(info, bx, by);
  const Windows w = mv_windows(r, g, mp.far_x, mp.far_y, mp.near_x, mp.near_y);
  const int x1 = r.x1, y1 = r.y1;
  // segment lengths
  const int common = max(0, min(y1 - w.fy1, w.fy2 - y1 - 1));
  const int n1 = 2 * common;
  const int n2 = max(0, (y1 - 1 - common) - w.fy1 + 1);  // remaining rows upwards
  const int n3 = max(0, w.fy2 - (y1 + 1 + common));      // remaining rows downwards
  const int n4 = max(0, x1 - w.fx1 + 1), n5 = max(0, w.fx2 - x1);
  const int ny_up = max(0, y1 - w.ry1 + 1), ny_dn = max(0, w.ry2 - y1 - 1), ny = ny_up + ny_dn;
  const int nxl = max(0, x1 - w.rx1 + 1), nxr = max(0, w.rx2 - x1 - 1);
  const int n6 = nxl * ny, n7 = nxr * ny;
  const int total = n1 + n2 + n3 + n4 + n5 + n6 + n7;
  auto cand = [&](int i, int& x, int& y) {
    x = x1;
    y = y1;
    if (i < n1) {
      const int k = i >> 1;
      y = (i & 1) ? y1 + 1 + k : y1 - 1 - k;
      return;
    }
    i -= n1;
    if (i < n2) {
      y = y1 - 1 - common - i;
      return;
    }
    i -= n2;
    if (i < n3) {
      y = y1 + 1 + common + i;
      return;
    }
    i -= n3;
    if (i < n4) {
      x = x1 - i;
      return;
    }
    i -= n4;
    if (i < n5) {
      x = x1 + i;
      return;
    }
    i -= n5;
    int xi, yi;
    if (i < n6) {
      xi = i / ny;
      yi = i - xi * ny;
      x = x1 - xi;
    } else {
      i -= n6;
      xi = i / ny;
      yi = i - xi * ny;
      x = x1 + 1 + xi;
    }
    y = yi < ny_up ? y1 - yi : y1 + 1 + (yi - ny_up);
  };
  for (int base = 0; base < total; base += 64) {
    const int i = base + lane;
    int x = 0, y = 0;
    bool ok = false;
    if (i < total) {
      cand(i, x, y);
      ok = same_rect(cur, prv, g.S, r, x, y);
    }
    const u64 m = __ballot(ok);
    if (m) {
      const int f = __builtin_ctzll(m);
      const int fx = __shfl(x, f), fy = __shfl(y, f);
      if (lane == 0) {  // |d| <= 256: two 10-bit fields, biased by 512
        const int dx = fx - x1, dy = fy - y1;
        smv[(size_t)pi * nblocks + b] = 0x80000000u | ((u32)(dx + 512) & 0x3FF) | (((u32)(dy + 512) & 0x3FF) << 10);
      }
      return;
    }
  }
  if (lane == 0) smv[(size_t)pi * nblocks + b] = 0;
}
__device__ __forceinline__ u32 smv_to_mv(u32 s) { return mv_pack((int)(s & 0x3FF) - 512, (int)((s >> 10) & 0x3FF) - 512); }

// The vectors most often found by the search in a frame (and in the frame before it, whose
// vectors reach this frame through mvs[]): the candidates FindMV's two predicted tries
// (:715-735) will almost always ask about.  dict[pi][0..7] = mv_pack or 0xFFFFFFFF.
constexpr int MVDICT = 8;
__global__ __launch_bounds__(256) void k_mvdict(Geom g, const u32* __restrict__ binfo, const u32* __restrict__ smv, u32* __restrict__ dict) {
  __shared__ u32 hkey[1024];
  __shared__ u32 hcnt[1024];
  __shared__ u32 best[256];
  __shared__ u32 bidx[256];
  const int pi = blockIdx.x, tid = threadIdx.x;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  for (int i = tid; i < 1024; i += 256) {
    hkey[i] = MV_NONE;
    hcnt[i] = 0;
  }
  __syncthreads();
  for (int f = max(0, pi - 1); f <= pi; f++)
    for (int b = tid; b < nblocks; b += 256) {
      const u32 sv = smv[(size_t)f * nblocks + b];
      if (!(binfo[(size_t)f * nblocks + b] & 1u) || !(sv >> 31)) continue;
      const u32 key = smv_to_mv(sv);
      u32 h = (key * 2654435761u) >> 22;
      for (int probe = 0; probe < 1024; probe++, h = (h + 1) & 1023) {
        const u32 old = atomicCAS(&hkey[h], MV_NONE, key);
        if (old == MV_NONE || old == key) {
          atomicAdd(&hcnt[h], 1u);
          break;
        }
      }
    }
  __syncthreads();
  for (int k = 0; k < MVDICT; k++) {  // repeated arg-max
    u32 bc = 0, bi = 0;
    for (int i = tid; i < 1024; i += 256)
      if (hcnt[i] > bc) {
        bc = hcnt[i];
        bi = (u32)i;
      }
    best[tid] = bc;
    bidx[tid] = bi;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
      if (tid < d && best[tid + d] > best[tid]) {
        best[tid] = best[tid + d];
        bidx[tid] = bidx[tid + d];
      }
      __syncthreads();
    }
    if (tid == 0) {
      dict[pi * MVDICT + k] = best[0] ? hkey[bidx[0]] : MV_NONE;
      if (best[0]) hcnt[bidx[0]] = 0;
    }
    __syncthreads();
  }
}

// row r of the rect at (x1,y1) in the current plane == row r of the rect at (x,y) in the previous plane
__device__ __forceinline__ bool same_row(const u8* cur, const u8* prv, int S, const Rect& r, int x, int y, int row) {
  const int wb = (r.x2 - r.x1) * 3;
  const u8* a = cur + (size_t)(r.y1 + row) * S + r.x1 * 3;
  const u8* b = prv + (size_t)(y + row) * S + x * 3;
  u32 diff = 0;
#pragma unroll
  for (int k = 0; k < 12; k++)
    if (k * 4 < wb) {
      u32 va, vb;
      __builtin_memcpy(&va, a + k * 4, 4);
      __builtin_memcpy(&vb, b + k * 4, 4);
      u32 d = va ^ vb;
      const int rem = wb - k * 4;
      if (rem < 4) d &= (1u << (8 * rem)) - 1u;
      diff |= d;
    }
  return diff == 0;
}

// pre[pi][b]: bit k set when dictionary vector k lies in the far window and matches exactly.
// Four blocks per wave, one rect row per lane.
__global__ __launch_bounds__(64) void k_mvpretest(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, const u32* __restrict__ binfo, MvParams mp,
                                                  const u32* __restrict__ dict, u32* __restrict__ pre) {
  const int pi = blockIdx.y, lane = threadIdx.x;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const int b = blockIdx.x * 4 + (lane >> 4), row = lane & 15;
  const bool vb = b < nblocks;
  const u32 info = vb ? binfo[(size_t)pi * nblocks + b] : 0;
  const bool changed = info & 1u;
  if (!__ballot(changed)) return;
  const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
  const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
  const int by = vb ? b / nbx : 0, bx = vb ? b - by * nbx : 0;
  const Rect r = binfo_rect(info, bx, by);
  const Windows w = mv_windows(r, g, mp.far_x, mp.far_y, mp.near_x, mp.near_y);
  u32 bits = 0;
  for (int k = 0; k < MVDICT; k++) {
    const u32 mv = dict[pi * MVDICT + k];
    bool bad = false;  // this row differs (or the candidate is unusable)
    if (changed) {
      const int x = r.x1 + mv_x(mv), y = r.y1 + mv_y(mv);
      const bool inwin = mv != MV_NONE && x >= w.fx1 && x < w.fx2 && y >= w.fy1 && y < w.fy2;
      if (!inwin) bad = true;
      else if (row < r.y2 - r.y1) bad = !same_row(cur, prv, g.S, r, x, y, row);
    }
    const u64 m = __ballot(bad);
    if (!((m >> (16 * (lane >> 4))) & 0xFFFFull)) bits |= 1u << k;
  }
  if (vb && row == 0) pre[(size_t)pi * nblocks + b] = changed ? bits : 0;
}

__device__ __forceinline__ void wave_fence_inter() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// Serial resolution over the P-frames of the chunk, in order (mvs[] carries over).
// btype: 0 unchanged, 1/2 pixel-coded (whole/partial), 3/4 motion (whole/partial).
// pinfo[pi] = {xx1, xx2}: bounding box corners of the changed blocks as block indices (:1145-1150)
// The walk is one wave, but its per-group inputs (block info, search result, pre-test bits, the vector
// memory) are a chain of dependent reads: with `use_lds` the whole workgroup copies a frame's arrays into
// LDS first (coalesced, 16 bytes per block: 128 KiB at 1080p) and the vector memory lives there for the
// whole launch; bigger frames fall back to global reads.
__global__ __launch_bounds__(256) void k_mvresolve(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, int npf, const u32* __restrict__ binfo,
                                                   const u32* __restrict__ smv, const u32* __restrict__ dict, const u32* __restrict__ pre, MvParams mp, u32* mvs,
                                                   u8* __restrict__ btype, u32* __restrict__ bmv, int* __restrict__ pinfo, const unsigned long long* __restrict__ gmask,
                                                   int use_lds) {
  extern __shared__ __align__(16) u32 stage[];  // [4][nblocks] when use_lds: vector memory, block info, search result, pre-test bits
  const int lane = threadIdx.x & 63;
  const bool walker = threadIdx.x < 64;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const int G = min(64, nbx);
  u32* const l_mv = stage;
  u32* const l_bi = stage + nblocks;
  u32* const l_sm = stage + 2 * nblocks;
  u32* const l_pr = stage + 3 * nblocks;
  if (use_lds) {
    for (int i = threadIdx.x; i < nblocks; i += 256) l_mv[i] = mvs[i];
    __syncthreads();
  }
  for (int pi = 0; pi < npf; pi++) {
    if (use_lds) {
      for (int i = threadIdx.x; i < nblocks; i += 256) {
        l_bi[i] = binfo[(size_t)pi * nblocks + i];
        l_sm[i] = smv[(size_t)pi * nblocks + i];
        l_pr[i] = pre[(size_t)pi * nblocks + i];
      }
      __syncthreads();
    }
   if (walker) {
    const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
    const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
    u32 dk[MVDICT];
#pragma unroll
    for (int k = 0; k < MVDICT; k++) dk[k] = dict[pi * MVDICT + k];
    auto dict_index = [&](u32 mv) __attribute__((always_inline)) {
      int idx = -1;
#pragma unroll
      for (int k = 0; k < MVDICT; k++)
        if (dk[k] == mv && idx < 0) idx = k;
      return idx;
    };
    u32 last = 0;  // last vector found by search (0,0), wave-uniform
    int bx1 = nbx, bx2 = -1, by1 = nby, by2 = -1;
    // groups without a changed block are skipped (their types and vectors were zeroed by the host): the loop
    // is a chain of dependent global reads per group, and most groups of a screen capture are untouched
    const int NG = (nblocks + G - 1) / G;
    for (int g0 = 0; g0 < NG; g0 += 64) {
     u64 active = __ballot(g0 + lane < NG && gmask[(size_t)pi * NG + g0 + lane] != 0ull);
     while (active) {
      const int base = (g0 + __builtin_ctzll(active)) * G;
      active &= active - 1;
      const int b = base + lane;
      const bool vb = lane < G && b < nblocks;
      const u32 info = vb ? (use_lds ? l_bi[b] : binfo[(size_t)pi * nblocks + b]) : 0;
      const bool changed = info & 1u;
      const int by = vb ? b / nbx : 0, bx = vb ? b - by * nbx : 0;
      int type = changed ? (int)((info >> 20) & 7) : 0;
      u32 my_mv = 0;
      bool has_mv = false;
      if (changed) {
        bx1 = min(bx1, bx);
        bx2 = max(bx2, bx);
        by1 = min(by1, by);
        by2 = max(by2, by);
      }
      u64 unresolved = __ballot(changed);
      if (unresolved) {
        const Rect r = binfo_rect(info, bx, by);
        const Windows w = mv_windows(r, g, mp.far_x, mp.far_y, mp.near_x, mp.near_y);
        const u32 s = changed ? (use_lds ? l_sm[b] : smv[(size_t)pi * nblocks + b]) : 0;
        const u32 pbits = changed ? (use_lds ? l_pr[b] : pre[(size_t)pi * nblocks + b]) : 0;
        wave_fence_inter();  // vectors stored by other lanes of this wave in earlier groups
        const u32 umv = (changed && by > 0) ? (use_lds ? l_mv[b - nbx] : __hip_atomic_load(&mvs[b - nbx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
        const int ui = dict_index(umv);
        auto exact = [&](u32 mv) __attribute__((always_inline)) {  // window test + SameBlocks for a vector outside the dictionary
          const int x = r.x1 + mv_x(mv), y = r.y1 + mv_y(mv);
          return x >= w.fx1 && x < w.fx2 && y >= w.fy1 && y < w.fy2 && same_rect(cur, prv, g.S, r, x, y);
        };
        while (unresolved) {
          const bool mine = changed && ((unresolved >> lane) & 1ull);
          const int li = dict_index(last);
          bool ta = false, tb = false;
          if (mine) {
            ta = li >= 0 ? ((pbits >> li) & 1u) : exact(last);
            if (!ta && by > 0 && umv != last) tb = ui >= 0 ? ((pbits >> ui) & 1u) : exact(umv);
          }
          const u64 Sm = __ballot(mine && !ta && !tb && (s >> 31));
          const int f = Sm ? __builtin_ctzll(Sm) : 64;
          const u64 upto = f >= 63 ? ~0ull : ((2ull << f) - 1ull);
          if (mine && ((upto >> lane) & 1ull)) {
            if (ta) {
              my_mv = last;
              has_mv = true;
            } else if (tb) {
              my_mv = umv;
              has_mv = true;
            } else if (lane == f) {
              my_mv = smv_to_mv(s);
              has_mv = true;
            }
          }
          if (f < 64) last = __shfl(my_mv, f);
          unresolved &= ~upto;
        }
      }
      if (vb) {
        if (has_mv) {
          type += 2;
          if (use_lds) l_mv[b] = my_mv;
          else __hip_atomic_store(&mvs[b], my_mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        btype[(size_t)pi * nblocks + b] = (u8)type;
        bmv[(size_t)pi * nblocks + b] = my_mv;
      }
     }
    }
    // bounding box of the changed blocks
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      bx1 = min(bx1, __shfl_xor(bx1, d));
      by1 = min(by1, __shfl_xor(by1, d));
      bx2 = max(bx2, __shfl_xor(bx2, d));
      by2 = max(by2, __shfl_xor(by2, d));
    }
    if (lane == 0) {
      pinfo[pi * 2] = bx2 < 0 ? 0 : by1 * nbx + bx1;
      pinfo[pi * 2 + 1] = bx2 < 0 ? -1 : by2 * nbx + bx2;
    }
   }
    if (use_lds) __syncthreads();  // the walk is done with this frame's arrays
  }
  if (use_lds)
    for (int i = threadIdx.x; i < nblocks; i += 256) mvs[i] = l_mv[i];
}

// The same resolution as a PIPELINE OF WAVES.  The only thing one frame hands to the next is the vector memory mvs[] (the
// vector of the block above, :726-735); a frame reads mvs[b - nbx] when it is at block b and writes mvs[b] there, so frame
// i + 1 may work on a group of blocks as soon as frame i is one block row past it (then frame i has read what frame i + 1 is about
// to overwrite, and has written what frame i + 1 is about to read).  One workgroup of sixteen waves: wave w takes frames w,
// w + 16, ... in order; the vector memory lives in LDS for the whole launch; every wave publishes (frame << 16 | blocks done) in
// an LDS word after each group, and waits on the word of the wave that has the frame before its own.  A chunk of 300 P-frames
// was 65 ms of one wave; with the frames a block row apart up to ~40 of them are in flight, sixteen here.
constexpr int MVP_WAVES = 16;
__global__ __launch_bounds__(64 * MVP_WAVES) void k_mvresolve_pipe(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, int npf, const u32* __restrict__ binfo,
                                                                   const u32* __restrict__ smv, const u32* __restrict__ dict, const u32* __restrict__ pre, MvParams mp, u32* mvs,
                                                                   u8* __restrict__ btype, u32* __restrict__ bmv, int* __restrict__ pinfo,
                                                                   const unsigned long long* __restrict__ gmask, u32* __restrict__ err) {
  extern __shared__ __align__(16) u32 stage[];  // the vector memory: nblocks words
  __shared__ u32 prog[MVP_WAVES];               // per wave: frame << 16 | blocks of it that are done (0xFFFF: all)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const int G = min(64, nbx), NG = (nblocks + G - 1) / G;
  u32* const l_mv = stage;
  for (int i = threadIdx.x; i < nblocks; i += 64 * MVP_WAVES) l_mv[i] = mvs[i];
  if (threadIdx.x < MVP_WAVES) prog[threadIdx.x] = 0;
  __syncthreads();
  const int prev_w = (wv + MVP_WAVES - 1) % MVP_WAVES;
  u32 seen = 0;  // the last value read from the previous frame's wave
  for (int pi = wv; pi < npf; pi += MVP_WAVES) {
    const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
    const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
    u32 dk[MVDICT];
#pragma unroll
    for (int k = 0; k < MVDICT; k++) dk[k] = dict[pi * MVDICT + k];
    auto dict_index = [&](u32 mv) __attribute__((always_inline)) {
      int idx = -1;
#pragma unroll
      for (int k = 0; k < MVDICT; k++)
        if (dk[k] == mv && idx < 0) idx = k;
      return idx;
    };
    // frame pi may touch blocks below `upto` once frame pi - 1 has finished every block below upto + nbx
    auto wait_for = [&](int upto) __attribute__((always_inline)) {
      if (pi == 0) return;
      const u32 need = ((u32)(pi - 1) << 16) | (u32)min(upto + nbx, 0xFFFF);
      int spins = 0;
      while (seen < need) {
        seen = __hip_atomic_load(&prog[prev_w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (seen < need) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 22)) {  // (cannot happen: the frame before is always ahead; never spin for ever on a GPU)
            if (lane == 0) atomicOr(err, 8u);
            break;
          }
        }
      }
      asm volatile("" ::: "memory");  // (LDS operations of a wave execute in order: what follows reads after the word above)
    };
    auto publish = [&](int done) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the vectors stored above are in LDS before the word that says so (global stores are not waited for)
      if (lane == 0) __hip_atomic_store(&prog[wv], ((u32)pi << 16) | (u32)min(done, 0xFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    u32 last = 0;  // last vector found by search (0,0), wave-uniform
    int bx1 = nbx, bx2 = -1, by1 = nby, by2 = -1;
    for (int g0 = 0; g0 < NG; g0 += 64) {
      u64 active = __ballot(g0 + lane < NG && gmask[(size_t)pi * NG + g0 + lane] != 0ull);
      while (active) {
        const int gi = g0 + __builtin_ctzll(active);
        const int base = gi * G;
        active &= active - 1;
        wait_for(base + G);
        const int b = base + lane;
        const bool vb = lane < G && b < nblocks;
        const u32 info = vb ? binfo[(size_t)pi * nblocks + b] : 0;
        const bool changed = info & 1u;
        const int by = vb ? b / nbx : 0, bx = vb ? b - by * nbx : 0;
        int type = changed ? (int)((info >> 20) & 7) : 0;
        u32 my_mv = 0;
        bool has_mv = false;
        if (changed) {
          bx1 = min(bx1, bx);
          bx2 = max(bx2, bx);
          by1 = min(by1, by);
          by2 = max(by2, by);
        }
        u64 unresolved = __ballot(changed);
        if (unresolved) {
          const Rect r = binfo_rect(info, bx, by);
          const Windows w = mv_windows(r, g, mp.far_x, mp.far_y, mp.near_x, mp.near_y);
          const u32 s = changed ? smv[(size_t)pi * nblocks + b] : 0;
          const u32 pbits = changed ? pre[(size_t)pi * nblocks + b] : 0;
          const u32 umv = (changed && by > 0) ? __hip_atomic_load(&l_mv[b - nbx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
          const int ui = dict_index(umv);
          auto exact = [&](u32 mv) __attribute__((always_inline)) {  // window test + SameBlocks for a vector outside the dictionary
            const int x = r.x1 + mv_x(mv), y = r.y1 + mv_y(mv);
            return x >= w.fx1 && x < w.fx2 && y >= w.fy1 && y < w.fy2 && same_rect(cur, prv, g.S, r, x, y);
          };
          while (unresolved) {
            const bool mine = changed && ((unresolved >> lane) & 1ull);
            const int li = dict_index(last);
            bool ta = false, tb = false;
            if (mine) {
              ta = li >= 0 ? ((pbits >> li) & 1u) : exact(last);
              if (!ta && by > 0 && umv != last) tb = ui >= 0 ? ((pbits >> ui) & 1u) : exact(umv);
            }
            const u64 Sm = __ballot(mine && !ta && !tb && (s >> 31));
            const int f = Sm ? __builtin_ctzll(Sm) : 64;
            const u64 upto = f >= 63 ? ~0ull : ((2ull << f) - 1ull);
            if (mine && ((upto >> lane) & 1ull)) {
              if (ta) {
                my_mv = last;
                has_mv = true;
              } else if (tb) {
                my_mv = umv;
                has_mv = true;
              } else if (lane == f) {
                my_mv = smv_to_mv(s);
                has_mv = true;
              }
            }
            if (f < 64) last = __shfl(my_mv, f);
            unresolved &= ~upto;
          }
        }
        if (vb) {
          if (has_mv) {
            type += 2;
            __hip_atomic_store(&l_mv[b], my_mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          btype[(size_t)pi * nblocks + b] = (u8)type;
          bmv[(size_t)pi * nblocks + b] = my_mv;
        }
        publish(base + G);
      }
      // the groups skipped in this stretch count as done - but a frame never claims more than the frame before it allows (the
      // frame after reads through this claim what ALL earlier frames have written)
      wait_for(min((g0 + 64) * G, nblocks));
      publish(min((g0 + 64) * G, nblocks));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      bx1 = min(bx1, __shfl_xor(bx1, d));
      by1 = min(by1, __shfl_xor(by1, d));
      bx2 = max(bx2, __shfl_xor(bx2, d));
      by2 = max(by2, __shfl_xor(by2, d));
    }
    if (lane == 0) {
      pinfo[pi * 2] = bx2 < 0 ? 0 : by1 * nbx + bx1;
      pinfo[pi * 2 + 1] = bx2 < 0 ? -1 : by2 * nbx + bx2;
    }
    wait_for(nblocks);
    publish(0xFFFF);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nblocks; i += 64 * MVP_WAVES) mvs[i] = l_mv[i];
}

// ---- inter predictors (GetPixelTypeP / PixelTypeFitsP and the edge forms, :525-604) ----
__device__ __forceinline__ bool eq3p(const u8* a, const u8* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }
__device__ __forceinline__ bool grad3p(const u8* p, int off) {
  return p[0] == (int)p[-3] + (int)p[off + 3] - (int)p[off] && p[1] == (int)p[-2] + (int)p[off + 4] - (int)p[off + 1] && p[2] == (int)p[-1] + (int)p[off + 5] - (int)p[off + 2];
}
__device__ __forceinline__ int inter_type(const u8* p, const u8* pr, int off, bool inner) {
  if (!inner) return eq3p(p, pr) ? 3 : 0;
  if (eq3p(p, p - 3)) return 1;
  if (eq3p(p, pr)) return 3;
  if (eq3p(p, p + off)) return 5;
  if (eq3p(p, p + off + 3)) return 2;
  if (grad3p(p, off)) return 4;
  return 0;
}
__device__ __forceinline__ bool inter_fits(int t, const u8* p, const u8* pr, const u8* lastp, int off, bool inner) {
  if (!inner) return t == 0 ? eq3p(p, lastp) : t == 3 ? eq3p(p, pr) : false;
  switch (t) {
    case 0: return eq3p(p, lastp);
    case 1: return eq3p(p, p - 3);
    case 2: return eq3p(p, p + off + 3);
    case 3: return eq3p(p, pr);
    case 4: return grad3p(p, off);
    default: return eq3p(p, p + off);
  }
}

// Walks the greedy runs of one changed rect (:1043-1065) and calls emit(type, n, first_index, last_index)
// for each run (indices are byte offsets into the plane).
template <class Emit>
__device__ __forceinline__ void walk_rect_runs(const u8* cur, const u8* prv, int S, const Rect& r, Emit&& emit) {
  const int off = -S - 3;
  int n = 0, t = 0, lasti = 0, first = 0;
  bool open = false;
  for (int y = r.y1; y < r.y2; y++) {
    int i = y * S + r.x1 * 3;
    for (int x = r.x1; x < r.x2; x++, i += 3) {
      const bool inner = x > 0 && y > 0;
      if (open && n < 255 && inter_fits(t, cur + i, prv + i, cur + lasti, off, inner)) {
        n++;
      } else {
        if (open) emit(t, n, first, lasti);
        t = inter_type(cur + i, prv + i, off, inner);
        n = 1;
        first = i;
        open = true;
      }
      lasti = i;
    }
  }
  if (open) emit(t, n, first, lasti);
}

// bcnt[b] = runs | literal runs << 16 for pixel-coded blocks
// Two kernels since round 5: a P-frame has ~330 pixel-coded blocks among its 8160, and with a lane per block a wave of 64
// neighbours walked two or three rects - one lane each, pixel by pixel - while the rest of it waited (2.6 ms for 294 frames).
// k_pactive lists the pixel-coded blocks of the chunk (list[0] = their number, zeroed by the host; the others' counts are 0;
// one atomic per workgroup of 1024 blocks), k_pcount walks the list with every lane at work: 0.1 + 1.0 ms.
// (k_pemit's pixel runs were moved to the same list as well - and cost more than they saved, 1.1 + 1.3 ms against 2.1: what is
// left of k_pemit is one thread per frame walking the block types, and the list's 1500 waves do not fill the card.)
__global__ __launch_bounds__(1024) void k_pactive(int nblocks, const u8* __restrict__ btype, u32* __restrict__ list, u32* __restrict__ bcnt) {
  __shared__ u32 wcnt[16], wbase;
  const int pi = blockIdx.y, b = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t at = (size_t)pi * nblocks + b;
  const int type = b < nblocks ? (int)btype[at] : 0;
  const bool act = type == 1 || type == 2;
  if (b < nblocks && !act) bcnt[at] = 0;
  const u64 m = __ballot(act);
  if (lane == 0) wcnt[w] = (u32)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tot = 0;
    for (int k = 0; k < 16; k++) tot += wcnt[k];
    wbase = tot ? atomicAdd(&list[0], tot) : 0u;
  }
  __syncthreads();
  if (act) {
    u32 before = 0;
    for (int k = 0; k < w; k++) before += wcnt[k];
    list[1 + wbase + before + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)at;
  }
}
__global__ __launch_bounds__(64) void k_pcount(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, const u32* __restrict__ binfo, const u32* __restrict__ list,
                                               u32* __restrict__ bcnt) {
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const u32 n = list[0];
  for (u32 i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) {
    const u32 at = list[1 + i];
    const int pi = (int)(at / (u32)nblocks), b = (int)(at - (u32)pi * (u32)nblocks);
    const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
    const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
    const int by = b / nbx, bx = b - by * nbx;
    const Rect r = binfo_rect(binfo[at], bx, by);
    int nr = 0, nl = 0;
    walk_rect_runs(cur, prv, g.S, r, [&](int t, int, int, int) {
      nr++;
      nl += t == 0;
    });
    bcnt[at] = (u32)nr | ((u32)nl << 16);
  }
}

// Per frame, one WAVE walks the blocks in raster order, 64 at a time (prefix sums over the lanes; a block's flags need the
// nearest motion block / pixel-coded block before it: the highest set bit below the lane in a ballot):
//   boff[b] = {symbol offset, run offset, colour-symbol offset, misc offset} relative to the frame's bases
//   bflag[b] bit0: motion vector equals the last coded one (:1202); bits 8..: index+1 of the previous pixel-coded block
//   ptot[pi] = {runs, symbols, colour symbols, misc symbols, block-type symbols}
// (One THREAD per frame did this with three dependent loads and two stores per block: 7 ms for 294 frames of 8160 blocks.)
struct BOff {
  u32 sym, run, col, misc;
};
__device__ __forceinline__ u32 pscan_excl(u32 v, int lane, u32& total) {  // exclusive prefix sum over the wave
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = (u32)__shfl_up((int)inc, d);
    if (lane >= d) inc += o;
  }
  total = (u32)__shfl((int)inc, 63);
  return inc - v;
}
__global__ __launch_bounds__(64) void k_pscan(Geom g, int npf, const u8* __restrict__ btype, const u32* __restrict__ bmv, const u32* __restrict__ bcnt, const int* __restrict__ pinfo,
                                              BOff* __restrict__ boff, u32* __restrict__ bflag, u32* __restrict__ ptot) {
  const int pi = blockIdx.x, lane = threadIdx.x;
  if (pi >= npf) return;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const u8* bt = btype + (size_t)pi * nblocks;
  const int xx1 = pinfo[pi * 2], xx2 = pinfo[pi * 2 + 1];
  const u64 below = (1ull << lane) - 1ull;
  // block-type run-length symbols: one (type, length) pair per run of equal types, length <= 255 - a run of L equal types is
  // ceil(L / 255) pairs, and every pair is two symbols
  u32 nbt = 0;
  {
    int carry = xx1;  // start of the run that reaches into the group at hand
    for (int x0 = xx1; x0 <= xx2; x0 += 64) {
      const int x = x0 + lane;
      const bool in = x <= xx2;
      const int t = in ? (int)bt[x] : -1, tp = (in && x > xx1) ? (int)bt[x - 1] : -2;
      int rs = (in && (x == xx1 || t != tp)) ? x : (lane == 0 ? carry : -1);  // start of the run the block lies in: a running maximum
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(rs, d);
        if (lane >= d) rs = max(rs, o);
      }
      nbt += 2u * (u32)__builtin_popcountll(__ballot(in && (x - rs) % 255 == 0));
      carry = __shfl(rs, 63);
    }
  }
  u32 sym = 4 + nbt, run = 0, col = 0, misc = 4 + nbt;
  u32 lastmv = 0;   // vector of the last motion block so far (the reference's `lastmv` changes exactly there)
  int prevpix = 0;  // index + 1 of the last pixel-coded block so far
  for (int b0 = 0; b0 < nblocks; b0 += 64) {
    const int b = b0 + lane;
    const bool in = b < nblocks;
    const int t = in ? (int)bt[b] : 0;
    const bool is_mv = t >= 3, is_pix = t == 1 || t == 2;
    const u32 mv = is_mv ? bmv[(size_t)pi * nblocks + b] : 0u;
    const u32 c = is_pix ? bcnt[(size_t)pi * nblocks + b] : 0u;
    const u64 mvmask = __ballot(is_mv), pixmask = __ballot(is_pix);
    const u64 mvb = mvmask & below, pxb = pixmask & below;
    const int pl = mvb ? 63 - __builtin_clzll(mvb) : 0;
    const u32 pmv = (u32)__shfl((int)mv, pl);
    const bool same = is_mv && b > 0 && mv == (mvb ? pmv : lastmv);
    const u32 nr = c & 0xFFFFu, nl = c >> 16;
    const u32 rect = (t == 2 || t == 4) ? 4u : 0u;
    const u32 isym = rect + (is_mv ? (same ? 1u : 3u) : is_pix ? 2 * nr + 3 * nl : 0u);
    const u32 imisc = rect + ((is_mv && !same) ? 2u : 0u);
    u32 tsym, trun, tcol, tmisc;
    const u32 esym = pscan_excl(isym, lane, tsym), erun = pscan_excl(is_pix ? nr : 0u, lane, trun), ecol = pscan_excl(is_pix ? 3 * nl : 0u, lane, tcol),
              emisc = pscan_excl(imisc, lane, tmisc);
    if (in) {
      boff[(size_t)pi * nblocks + b] = BOff{sym + esym, run + erun, col + ecol, misc + emisc};
      const int pp = pxb ? b0 + (63 - __builtin_clzll(pxb)) + 1 : prevpix;
      bflag[(size_t)pi * nblocks + b] = ((u32)pp << 8) | (same ? 1u : 0u);
    }
    sym += tsym, run += trun, col += tcol, misc += tmisc;
    if (mvmask) lastmv = (u32)__shfl((int)mv, 63 - __builtin_clzll(mvmask));
    if (pixmask) prevpix = b0 + (63 - __builtin_clzll(pixmask)) + 1;
  }
  if (xx2 < xx1) {  // nothing changed: no stream at all
    sym = run = col = misc = 0;
    nbt = 0;
  }
  if (lane == 0) {
    u32* o = ptot + (size_t)pi * 8;
    o[0] = run;
    o[1] = sym;
    o[2] = col;
    o[3] = misc;
    o[4] = nbt;
  }
}

// misc fixed-alphabet symbols of a P-frame: ctx << 16 | value, with the stream position
enum { MC_X = 0, MC_BN = 1, MC_BT = 2, MC_SXY = 3, MC_MX = 7, MC_MY = 8, MC_COUNT = 9 };

struct PBase {  // per P-frame (device copy of the relevant FrameBase fields)
  u32 run_base, sym_base, col_base, misc_base, nbt, gen, pad0, pad1;
};

// grid.x = ceil(nblocks/64) + 1; the extra block's thread 0 writes the frame header symbols
__global__ __launch_bounds__(64) void k_pemit(const u8* __restrict__ planes, Geom g, const PFrame* __restrict__ pf, const PBase* __restrict__ pb, const u32* __restrict__ binfo,
                                              const u8* __restrict__ btype, const u32* __restrict__ bmv, const BOff* __restrict__ boff, const u32* __restrict__ bflag,
                                              const int* __restrict__ pinfo, MvParams mp, u32* __restrict__ runs, u32* __restrict__ runpos, u32* __restrict__ keys,
                                              u32* __restrict__ vals, u32* __restrict__ misc, u32* __restrict__ miscpos, u32* __restrict__ entries) {
  const int pi = blockIdx.y;
  const int nbx = (g.W + 15) >> 4, nby = (g.H + 15) >> 4, nblocks = nbx * nby;
  const PBase fb = pb[pi];
  const int xx1 = pinfo[pi * 2], xx2 = pinfo[pi * 2 + 1];
  if (xx2 < xx1) return;
  const u8* bt = btype + (size_t)pi * nblocks;
  if ((int)blockIdx.x == (nblocks + 63) / 64) {
    // the frame's head: the corner blocks' indices (:1145-1150) and the block types between them as (type, length) pairs, a run
    // of equal types cut at 255 (:1155-1169) - symbol 4 + 2 k is the type of piece k, 4 + 2 k + 1 its length.  One wave, 64
    // blocks a round, the pieces found as in k_pscan (one thread walking the 8160 blocks was 1.1 of this kernel's 2.1 ms).
    const int lane = threadIdx.x;
    const u64 below = (1ull << lane) - 1ull;
    auto put_at = [&](u32 idx, int ctx, int v) {
      misc[fb.misc_base + idx] = ((u32)ctx << 16) | (u32)v;
      miscpos[fb.misc_base + idx] = fb.sym_base + idx;
    };
    if (lane < 4) put_at((u32)lane, MC_X, ((lane < 2 ? xx1 : xx2) >> (8 * (lane & 1))) & 255);
    int carry = xx1, lastb = xx1;  // start of the run that reaches into the round at hand; the last piece's first block
    u32 pieces = 0;
    for (int x0 = xx1; x0 <= xx2; x0 += 64) {
      const int x = x0 + lane;
      const bool in = x <= xx2;
      const int t = in ? (int)bt[x] : -1, tp = (in && x > xx1) ? (int)bt[x - 1] : -2;
      int rs = (in && (x == xx1 || t != tp)) ? x : (lane == 0 ? carry : -1);  // start of the run the block lies in: a running maximum
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(rs, d);
        if (lane >= d) rs = max(rs, o);
      }
      const bool first = in && (x - rs) % 255 == 0;  // the block starts a piece
      const u64 fm = __ballot(first);
      if (first) {
        const u64 fb4 = fm & below;
        const u32 k = pieces + (u32)__builtin_popcountll(fb4);
        const int prevb = fb4 ? x0 + (63 - __builtin_clzll(fb4)) : lastb;
        if (k > 0) put_at(4u + 2u * (k - 1u) + 1u, MC_BN, x - prevb);  // the piece before ends here
        put_at(4u + 2u * k, MC_BT, t);
      }
      if (fm) lastb = x0 + (63 - __builtin_clzll(fm));
      pieces += (u32)__builtin_popcountll(fm);
      carry = __shfl(rs, 63);
    }
    if (lane == 0) put_at(4u + 2u * (pieces - 1u) + 1u, MC_BN, xx2 + 1 - lastb);
    return;
  }
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nblocks) return;
  const int t = bt[b];
  if (!t) return;
  const u8* cur = planes + (size_t)pf[pi].slot * g.plane_stride;
  const u8* prv = planes + (size_t)pf[pi].prev_slot * g.plane_stride;
  const int by = b / nbx, bx = b - by * nbx;
  const u32 info = binfo[(size_t)pi * nblocks + b];
  const Rect r = binfo_rect(info, bx, by);
  const BOff o = boff[(size_t)pi * nblocks + b];
  const u32 fl = bflag[(size_t)pi * nblocks + b];
  u32 pos = fb.sym_base + o.sym, mk = fb.misc_base + o.misc;
  auto put = [&](int ctx, int v) {
    misc[mk] = ((u32)ctx << 16) | (u32)v;
    miscpos[mk] = pos;
    mk++;
    pos++;
  };
  if (t == 2 || t == 4) {  // :1190-1197
    put(MC_SXY + 0, r.x1 - bx * 16);
    put(MC_SXY + 1, r.y1 - by * 16);
    put(MC_SXY + 2, r.x2 - 1 - bx * 16);
    put(MC_SXY + 3, r.y2 - 1 - by * 16);
  }
  if (t >= 3) {  // :1199-1214
    const u32 mv = bmv[(size_t)pi * nblocks + b];
    const bool same = fl & 1u;
    entries[pos++] = (u32)(kProbScale / 2) | ((same ? (u32)(kProbScale / 2) : 0u) << 16);  // encodeBool, screencap.h:407-410
    if (!same) {
      put(MC_MX, mv_x(mv) + mp.far_x);
      put(MC_MY, mv_y(mv) + mp.far_y);
    }
    return;
  }
  // pixel runs (:1215-1245).  The colour context of the first literal comes from the last pixel
  // coded before this block: the bottom-right pixel of the previous pixel-coded block's rect
  // (cx = cx1 = 0 at the start of the frame, :1176).
  u32 pg = 0, pbv = 0;
  const int pp = (int)(fl >> 8);
  if (pp) {
    const int pbi = pp - 1, pby = pbi / nbx, pbx = pbi - pby * nbx;
    const Rect q = binfo_rect(binfo[(size_t)pi * nblocks + pbi], pbx, pby);
    const u8* px = cur + (size_t)(q.y2 - 1) * g.S + (q.x2 - 1) * 3;
    pg = px[1];
    pbv = px[2];
  }
  u32 ri = fb.run_base + o.run, ci = fb.col_base + o.col / 3u;  // (col_base: the frame's first literal in its generation's plane arrays; pad0: their stride)
  int lastt = 0;
  walk_rect_runs(cur, prv, g.S, r, [&](int type, int n, int first, int lasti) {
    runs[ri] = make_run(type, lastt, n, false);
    runpos[ri] = pos;
    ri++;
    if (type == 0) {
      emit_colour(fb.gen, ld3(cur + first), pg, pbv, pos + 1, ci, fb.pad0, keys, vals);
      ci += 1;
    }
    pos += 2 + (type == 0 ? 3 : 0);
    lastt = type;
    pg = cur[lasti + 1];
    pbv = cur[lasti + 2];
  });
}

// ---- the P-frame models' chains are in scpr_fixed.hpp (MiscItems) ----
struct MiscRange {
  u32 begin, end;
};

}  // namespace scpr
