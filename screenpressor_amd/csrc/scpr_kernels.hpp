// HIP kernels of the MI355X-native ScreenPressor path (gfx950, wave64).
//
// Encoder pipeline for a batch of frames resident in HBM:
//   k_pack*           RGB32/24/16 -> packed RGB24 planes, flat-frame detection     (screencap.cpp:1652-1678, :1436-1444)
//   k_tiles           per 1024-pixel tile: predictor type + "fits" bitmaps via
//                     wave ballots, greedy-run successor of every pixel (kept: type | length of the run that would start
//                     there), pointer doubling in LDS -> exit map for every possible entry       (ClassifyPixelsI, :876-919)
//   k_entries         chase the tile entries through the exit maps (one wave/frame)
//   k_runs            one lane per tile walks the real path from its entry and writes the run records
//   k_header          runs of the first row + pixel (0,1)                           (CompressI, :344-362)
//   k_scan_tiles / k_bases   prefix sums -> run / symbol / colour-symbol offsets
//   k_symbols         unified run list + (context,value,position) of every colour symbol   (WritePixel/EncodeRGB, :609-643)
//   (rocPRIM)         stable radix sort of colour symbols by (generation, plane, context)
//   k_part_* / k_fixed_chain2 (scpr_fixed.hpp)  the run list partitioned by fixed-alphabet model, then one wave per model:
//                     epoch-parallel lookups, LDS-resident table, wave prefix-scan rebuilds  (FixedSizeRansCtx, ans_contexts.h:1054-1132)
//   k_colour_chain_w  one wave per colour context: the 7-kind state machine (scpr_wave.hpp) (Context, ans_contexts.cpp:34-50)
//   k_rans            one lane per 131072-entry block: byte-wise rANS, reverse order (ransmt.h:116-134, rans_byte.h:59-102)
//   k_offsets/k_gather  packet assembly
// Decoder: k_decode_gop_w (scpr_wave.hpp: one wave per GOP, serial symbol chain) + k_unpack*.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scpr_model.hpp"

namespace scpr {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int TILE = 1024;      // pixels per classification tile
constexpr int HALO = 255;       // a run is at most 255 pixels
constexpr int EXITED = 2047;    // successor beyond the frame end
constexpr int NCOLCTX = 3 * 4096;
constexpr int NFIXED_I = 12;    // pixel-type[6] + run-length[6]
constexpr int RANS_SCRATCH = 2 * kBlockEntries + 8;

struct Geom {
  int W, H, S, NP;   // width, height, RGB24 stride, pixels
  int p0;            // first classified raster pixel: (1,1)
  int ntiles;
  int workers;
  u32 plane_stride;  // bytes between planes (H*S rounded up + slack)
};

// run record of the unified run list
//   bits 0-2 type, 3-5 previous type, 8-15 n, 31 header run (no pixel-type symbol)
__device__ __forceinline__ u32 make_run(int type, int lastt, int n, bool hdr) {
  return (u32)type | ((u32)lastt << 3) | ((u32)n << 8) | (hdr ? 0x80000000u : 0u);
}

__device__ __forceinline__ u32 ld3(const u8* p) {  // three bytes, little endian
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v & 0xFFFFFFu;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// ------------------------------------------------------------------ pack ---
// RGB32 -> RGB24 plane (alpha dropped), 4 pixels (16 B in, 12 B out) per lane.
// flat[f] is set when any pixel differs from pixel 0; first[f] = pixel 0.
__device__ __forceinline__ bool pack32_item(const u8* __restrict__ src, u8* __restrict__ planes, const Geom& g, int f, int idx, int G, u32* first) {
  const int y = idx / G, gx = idx - y * G;
  const u32* s = (const u32*)(src + (size_t)f * g.W * g.H * 4 + (size_t)y * g.W * 4) + gx * 4;
  const u32 px0 = *(const u32*)(src + (size_t)f * g.W * g.H * 4) & 0xFFFFFFu;
  const int nv = min(4, g.W - gx * 4);
  u32 a, b, c, d;
  if ((g.W & 3) == 0) {  // rows are 16-byte aligned: one 16-byte load per lane, 1 KiB per wave instruction
    const uint4 v = *(const uint4*)s;
    a = v.x & 0xFFFFFFu;
    b = v.y & 0xFFFFFFu;
    c = v.z & 0xFFFFFFu;
    d = v.w & 0xFFFFFFu;
  } else {
    a = s[0] & 0xFFFFFFu;
    b = nv > 1 ? s[1] & 0xFFFFFFu : 0;
    c = nv > 2 ? s[2] & 0xFFFFFFu : 0;
    d = nv > 3 ? s[3] & 0xFFFFFFu : 0;
  }
  const bool diff = a != px0 || (nv > 1 && b != px0) || (nv > 2 && c != px0) || (nv > 3 && d != px0);
  if (idx == 0) first[f] = px0;
  u32* o = (u32*)(planes + (size_t)f * g.plane_stride + (size_t)y * g.S) + gx * 3;
  const int room = g.S - gx * 12;  // bytes left in the row: 4, 8 or >= 12
  o[0] = a | (b << 24);
  if (room > 4) o[1] = (b >> 8) | (c << 16);
  if (room > 8) o[2] = (c >> 16) | (d << 8);
  return diff;
}
constexpr int PACK_ITEMS = 8;  // quads per thread: a workgroup per 256 quads is bound by the dispatcher, not by HBM
__global__ __launch_bounds__(256) void k_pack32(const u8* __restrict__ src, u8* __restrict__ planes, Geom g, u32* flat, u32* first) {
  __shared__ __attribute__((aligned(16))) u32 stage[4][192];  // per wave: the 768 output bytes of its 64 pixel quads
  const int f = blockIdx.y, G = (g.W + 3) >> 2;
  const int total = g.H * G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u8* fsrc = src + (size_t)f * g.W * g.H * 4;
  bool diff = false;
  for (int it = 0; it < PACK_ITEMS; it++) {
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    const int wave0 = idx & ~63;
    if ((g.W & 3) == 0 && wave0 + 64 <= total) {
      // Rows are whole quads and carry no padding: the plane is one contiguous run of 12-byte quads, so the
      // wave's output is 768 contiguous bytes: the quads go through LDS and leave as 16-byte stores from 48 lanes.
      const uint4 v = ((const uint4*)fsrc)[idx];
      const u32 px0 = *(const u32*)fsrc & 0xFFFFFFu;
      const u32 a = v.x & 0xFFFFFFu, b = v.y & 0xFFFFFFu, c = v.z & 0xFFFFFFu, d = v.w & 0xFFFFFFu;
      diff |= a != px0 || b != px0 || c != px0 || d != px0;
      if (idx == 0) first[f] = px0;
      u32* st = stage[wv];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous trip's reads are done (same wave: LDS is in order)
      st[3 * lane] = a | (b << 24);
      st[3 * lane + 1] = (b >> 8) | (c << 16);
      st[3 * lane + 2] = (c >> 16) | (d << 8);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // other lanes' words are read below
      if (lane < 48) {
        const uint4 o = ((const uint4*)st)[lane];
        ((uint4*)(planes + (size_t)f * g.plane_stride + (size_t)wave0 * 12))[lane] = o;
      }
    } else if (idx < total) {
      diff |= pack32_item(src, planes, g, f, idx, G, first);
    }
  }
  // flat detection: one relaxed L2 read and at most one atomic per wave
  if (__ballot(diff) && lane_id() == 0 && __hip_atomic_load(&flat[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flat[f], 1u);
}

// RGB24 rows (pitch = S) -> plane copy with zeroed row padding
__global__ __launch_bounds__(256) void k_pack24(const u8* __restrict__ src, u8* __restrict__ planes, Geom g, u32* flat, u32* first) {
  const int f = blockIdx.y, G = g.S >> 2, total = g.H * G;
  const u8* fsrc = src + (size_t)f * g.S * g.H;
  const u32 px0 = ld3(fsrc), b0 = px0 & 255u, b1 = (px0 >> 8) & 255u, b2 = px0 >> 16;
  // a flat frame repeats its first pixel: dword gx of a row holds the pattern at phase gx % 3 (4 = 1 mod 3)
  const u32 e0 = b0 | (b1 << 8) | (b2 << 16) | (b0 << 24), e1 = b1 | (b2 << 8) | (b0 << 16) | (b1 << 24), e2 = b2 | (b0 << 8) | (b1 << 16) | (b2 << 24);
  bool diff = false;
  for (int it = 0; it < PACK_ITEMS; it++) {  // several dwords per thread (see k_pack32)
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (idx >= total) break;
    const int y = idx / G, gx = idx - y * G;
    u32 v = ((const u32*)(fsrc + (size_t)y * g.S))[gx];
    const int valid = g.W * 3 - gx * 4;  // bytes of this dword that are pixel data
    const u32 mask = valid >= 4 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
    v &= mask;
    const int q = gx % 3;
    diff |= v != ((q == 0 ? e0 : q == 1 ? e1 : e2) & mask);
    ((u32*)(planes + (size_t)f * g.plane_stride + (size_t)y * g.S))[gx] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) first[f] = px0;
  // flat detection: one relaxed L2 read and at most one atomic per wave
  if (__ballot(diff) && lane_id() == 0 && __hip_atomic_load(&flat[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flat[f], 1u);
}

// RGB16 -> plane through the caller's channel masks (screencap.cpp:1665-1678)
__global__ __launch_bounds__(256) void k_pack16(const u8* __restrict__ src, u8* __restrict__ planes, Geom g, u32* flat, u32* first,
                                                u32 rm, u32 gm, u32 bm, int rs, int gs, int bs) {
  const int f = blockIdx.y, total = g.H * g.W;
  const int pitch = g.W * 2;  // rows back to back, as the reference reads them (screencap.cpp:1668: i = y*X*2)
  const u8* fr = src + (size_t)f * pitch * g.H;
  auto conv = [&](u32 w) { return ((w & rm) >> rs) | (((w & gm) >> gs) << 8) | (((w & bm) >> bs) << 16); };
  const u32 v0 = conv(*(const u16*)fr) & 0xFFFFFFu;
  bool diff = false;
  for (int it = 0; it < PACK_ITEMS; it++) {
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (idx >= total) break;
    const int y = idx / g.W, x = idx - y * g.W;
    const u32 v = conv(*(const u16*)(fr + (size_t)y * pitch + x * 2)) & 0xFFFFFFu;
    diff |= v != v0;
    u8* o = planes + (size_t)f * g.plane_stride + (size_t)y * g.S + x * 3;
    o[0] = (u8)v;
    o[1] = (u8)(v >> 8);
    o[2] = (u8)(v >> 16);
    if (x == g.W - 1)
      for (int k = g.W * 3; k < g.S; k++) (planes + (size_t)f * g.plane_stride + (size_t)y * g.S)[k] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) first[f] = v0;
  if (__ballot(diff) && lane_id() == 0 && __hip_atomic_load(&flat[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flat[f], 1u);
}

// lossy pre-quantisation on the plane's dwords, then padding back to zero
// (DoLoss, screencap.cpp:201-220, :852-861)
__global__ __launch_bounds__(256) void k_loss(u8* planes, Geom g, const int* slots, u32 loss_mask, u32 corr_mask) {
  const int slot = slots[blockIdx.y], G = g.S >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= g.H * G) return;
  const int y = idx / G, gx = idx - y * G;
  u32* row = (u32*)(planes + (size_t)slot * g.plane_stride + (size_t)y * g.S);
  u32 v = (row[gx] & loss_mask) | corr_mask;
  const int valid = g.W * 3 - gx * 4;
  if (valid < 4) v &= (valid <= 0) ? 0u : ((1u << (8 * valid)) - 1u);
  row[gx] = v;
}

// plane -> RGB32 with alpha 255 (screencap.cpp:1711-1725), 4 pixels per lane
__global__ __launch_bounds__(256) void k_unpack32(const u8* __restrict__ planes, u8* __restrict__ dst, Geom g, int pitch) {
  const int f = blockIdx.y, G = (g.W + 3) >> 2, total = g.H * G;
  for (int it = 0; it < PACK_ITEMS; it++) {  // several quads per thread (see k_pack32)
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int y = idx / G, gx = idx - y * G;
    const u32* s = (const u32*)(planes + (size_t)f * g.plane_stride + (size_t)y * g.S) + gx * 3;
    const int room = g.S - gx * 12;
    u32 w0 = s[0], w1 = room > 4 ? s[1] : 0, w2 = room > 8 ? s[2] : 0;
    u32* o = (u32*)(dst + (size_t)f * pitch * g.H + (size_t)y * pitch) + gx * 4;
    const int nv = min(4, g.W - gx * 4);
    if (nv == 4 && ((pitch | (int)(size_t)dst) & 15) == 0) {  // rows and base 16-byte aligned: one store
      *(uint4*)o = make_uint4((w0 & 0xFFFFFFu) | 0xFF000000u, ((w0 >> 24) | ((w1 & 0xFFFFu) << 8)) | 0xFF000000u, ((w1 >> 16) | ((w2 & 0xFFu) << 16)) | 0xFF000000u,
                              (w2 >> 8) | 0xFF000000u);
    } else {
      o[0] = (w0 & 0xFFFFFFu) | 0xFF000000u;
      if (nv > 1) o[1] = ((w0 >> 24) | ((w1 & 0xFFFFu) << 8)) | 0xFF000000u;
      if (nv > 2) o[2] = ((w1 >> 16) | ((w2 & 0xFFu) << 16)) | 0xFF000000u;
      if (nv > 3) o[3] = (w2 >> 8) | 0xFF000000u;
    }
  }
}
// the same for the frames of a list (slots[blockIdx.y]): the host-pointer decode call unpacks what the decoder's chains have not
// already sent to the host row by row (P-frames, flat frames) - straight into the host's mapped buffer
__global__ __launch_bounds__(256) void k_unpack32_list(const u8* __restrict__ planes, u8* __restrict__ dst, Geom g, int pitch, const int* __restrict__ slots) {
  const int f = slots[blockIdx.y], G = (g.W + 3) >> 2, total = g.H * G;
  for (int it = 0; it < PACK_ITEMS; it++) {
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int y = idx / G, gx = idx - y * G;
    const u32* s = (const u32*)(planes + (size_t)f * g.plane_stride + (size_t)y * g.S) + gx * 3;
    const int room = g.S - gx * 12;
    u32 w0 = s[0], w1 = room > 4 ? s[1] : 0, w2 = room > 8 ? s[2] : 0;
    u32* o = (u32*)(dst + (size_t)f * pitch * g.H + (size_t)y * pitch) + gx * 4;
    const int nv = min(4, g.W - gx * 4);
    if (nv == 4 && ((pitch | (int)(size_t)dst) & 15) == 0) {
      *(uint4*)o = make_uint4((w0 & 0xFFFFFFu) | 0xFF000000u, ((w0 >> 24) | ((w1 & 0xFFFFu) << 8)) | 0xFF000000u, ((w1 >> 16) | ((w2 & 0xFFu) << 16)) | 0xFF000000u,
                              (w2 >> 8) | 0xFF000000u);
    } else {
      o[0] = (w0 & 0xFFFFFFu) | 0xFF000000u;
      if (nv > 1) o[1] = ((w0 >> 24) | ((w1 & 0xFFFFu) << 8)) | 0xFF000000u;
      if (nv > 2) o[2] = ((w1 >> 16) | ((w2 & 0xFFu) << 16)) | 0xFF000000u;
      if (nv > 3) o[3] = (w2 >> 8) | 0xFF000000u;
    }
  }
}
// plane -> RGB24 rows with the caller's pitch / RGB16 (screencap.cpp:1726-1737)
__global__ __launch_bounds__(256) void k_unpack_rows(const u8* __restrict__ planes, u8* __restrict__ dst, Geom g, int pitch, int bpp, int rs, int gs, int bs) {
  const int f = blockIdx.y, total = g.H * g.W;
  for (int it = 0; it < PACK_ITEMS; it++) {
    const int idx = (blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int y = idx / g.W, x = idx - y * g.W;
    const u8* s = planes + (size_t)f * g.plane_stride + (size_t)y * g.S + x * 3;
    u8* o = dst + (size_t)f * pitch * g.H + (size_t)y * pitch;
    if (bpp == 3) {
      o[x * 3] = s[0];
      o[x * 3 + 1] = s[1];
      o[x * 3 + 2] = s[2];
    } else {
      *(u16*)(o + x * 2) = (u16)((s[0] << rs) + (s[1] << gs) + (s[2] << bs));
    }
  }
}
// RGB24 output whose pitch is the plane's stride: the plane (rows with zeroed padding) is the output
__global__ __launch_bounds__(256) void k_copy_planes(const u8* __restrict__ planes, u8* __restrict__ dst, Geom g) {
  const int f = blockIdx.y;
  const size_t bytes = (size_t)g.H * g.S;  // a multiple of 4
  const u32* s = (const u32*)(planes + (size_t)f * g.plane_stride);
  u32* o = (u32*)(dst + (size_t)f * bytes);
  for (int it = 0; it < PACK_ITEMS; it++) {
    const size_t i = ((size_t)blockIdx.x * PACK_ITEMS + it) * 256 + threadIdx.x;
    if (i * 4 < bytes) o[i] = s[i];
  }
}

// ------------------------------------------------------- classification ---
// Predictor type of raster pixel p (priority 1,5,2,4,0 - GetPixelType,
// screencap.cpp:502-521) and the set of predictors that fit it
// (PixelTypeFits, :560-574): bit0 previous pixel (types 0/1), bit1 top (2),
// bit2 gradient (4), bit3 top-left (5).
__device__ __forceinline__ void classify_pixel(const u8* plane, const Geom& g, int y, int x, int& type, int& fits) {
  const u8* c = plane + (size_t)y * g.S + x * 3;
  // the pixel and the one before it are six adjacent bytes, and so are the two above them: two 8-byte windows instead of four
  // 4-byte ones (in column 0 the previous pixel is the last one of the row above and is fetched by itself)
  u64 wc, wt;
  __builtin_memcpy(&wc, c - 3, 8);
  __builtin_memcpy(&wt, c - g.S - 3, 8);
  const u32 vc = (u32)(wc >> 24) & 0xFFFFFFu, vt = (u32)(wt >> 24) & 0xFFFFFFu, vtl = (u32)wt & 0xFFFFFFu;
  u32 vl = (u32)wc & 0xFFFFFFu;
  if (x == 0) vl = ld3(plane + (size_t)(y - 1) * g.S + (g.W - 1) * 3);
  const bool e_l = vc == vl, e_t = vc == vt, e_tl = vc == vtl;
  bool gr = true;
#pragma unroll
  for (int k = 0; k < 24; k += 8) gr &= (int)((vc >> k) & 255) == (int)((vl >> k) & 255) + (int)((vt >> k) & 255) - (int)((vtl >> k) & 255);
  fits = (e_l ? 1 : 0) | (e_t ? 2 : 0) | (gr ? 4 : 0) | (e_tl ? 8 : 0);
  type = e_l ? 1 : e_tl ? 5 : e_t ? 2 : gr ? 4 : 0;
  // a row band of the reference's worker pool starts a new run (screencap.cpp:365-388)
  if (x == 0 && g.workers > 1) {
    int k = (int)(((long long)y * g.workers + g.H - 1) / g.H);
    if (k > 0 && k < g.workers && (int)((long long)g.H * k / g.workers) == y) fits = 0;
  }
}
__device__ __forceinline__ int fit_bit_of_type(int t) { return t == 2 ? 1 : t == 4 ? 2 : t == 5 ? 3 : 0; }

// number of consecutive set bits starting at bit `start`, at most cap
__device__ __forceinline__ int ones_from(const u64* row, int start, int cap) {
  int w = start >> 6, o = start & 63;
  u64 inv = ~(row[w] >> o);
  int z = inv ? __builtin_ctzll(inv) : 64;
  if (z < 64 - o) return min(z, cap);
  int cnt = 64 - o;
  w++;
  while (cnt < cap) {
    u64 m = ~row[w];
    if (m) {
      cnt += __builtin_ctzll(m);
      break;
    }
    cnt += 64;
    w++;
  }
  return min(cnt, cap);
}

// block-wide exclusive scan of one int per thread (256 threads); total via *tot
__device__ __forceinline__ int block_excl_scan(int v, int* tot, int* wsum /*shared[5]*/) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < wave; k++) base += wsum[k];
  if (tot) *tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc - v;
}

// exit map row: for every entry offset e < 255: [2e] = offset at which the path
// leaves into the next tile, [2e+1] = type of the run that crosses the border.
// tilecnt: {runs, literal runs} per tile.   runrec: rel_start | type<<10 | n<<16
__global__ __launch_bounds__(256) void k_tiles(const u8* __restrict__ planes, Geom g, const int* __restrict__ slots, u8* __restrict__ exitmap, u16* __restrict__ tnmap) {
  __shared__ u64 fm[4][24];
  __shared__ __attribute__((aligned(4))) u8 ty[TILE];
  __shared__ u16 nzw[4][26];
  __shared__ __attribute__((aligned(16))) u32 lp[TILE];  // a pixel's way out of its quarter of the tile: successor << 16 | start of the run it sits in
  // Workgroups go round the eight XCDs in launch order, and every XCD has its own L2: tile t reads the row above its pixels,
  // which tile t - W / 1024 (the one before the last, at 1080p) has just read.  The launch order is therefore remapped so
  // that each XCD gets one contiguous eighth of the (frame, tile) space - neighbouring tiles meet in the same L2 instead
  // of being fetched from memory once per XCD.
  const u32 nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const u32 xq = nwg >> 3, xr = nwg & 7u, xcd = lin & 7u, xk = lin >> 3;
  const u32 ord = xcd < xr ? xcd * (xq + 1u) + xk : xr * (xq + 1u) + (xcd - xr) * xq + xk;
  const int fidx = (int)(ord / gridDim.x);
  const int slot = slots[fidx], tile = (int)(ord - (u32)fidx * gridDim.x), tid = threadIdx.x;
  const u8* plane = planes + (size_t)slot * g.plane_stride;
  const int tstart = g.p0 + tile * TILE;
  // row and column of the tile's first pixel, once per workgroup; a pixel's own come from there by stepping over
  // row ends (one division per pixel was a quarter of this kernel's instructions)
  const int ty0 = __builtin_amdgcn_readfirstlane(tstart / g.W), tx0 = __builtin_amdgcn_readfirstlane(tstart - ty0 * g.W);
  if (tid < 4 * 24) ((u64*)fm)[tid] = 0;
  __syncthreads();
  if (g.S == 3 * g.W && g.workers == 1) {
    // Rows without padding (every width that is a multiple of four): the plane IS the raster - pixel p at byte 3 p, its left
    // neighbour the pixel before it also in column 0 (the last one of the row above, :881), the row above W pixels back - so
    // a lane takes FOUR pixels from two 16-byte loads (they and the one before them; the same above) with no row or column in
    // sight, and adds its four bits per predictor to the maps with one LDS `or` each.  36 instructions per pixel; the
    // pixel-at-a-time form below (any stride, worker bands) is 95, and this loop was 60 % of the kernel.
    // The tile's 1024 pixels are one such round for every lane; the 255 pixels of the halo are ONE more pixel per lane (bits by
    // ballot, as below) - as a second round of four they were the first wave's alone, and a workgroup's first waves share a SIMD.
    {
      const int q = tid, r0 = 4 * q, p = tstart + r0;  // pixels r0 .. r0 + 3 of the tile
      if (p < g.NP) {  // (a lane's last pixels may lie past the frame: their loads stay inside the plane's slack, their bits are dropped)
        u32 e[4], a[4];
        __builtin_memcpy(e, plane + (size_t)p * 3 - 3, 16);
        __builtin_memcpy(a, plane + (size_t)(p - g.W) * 3 - 3, 16);
        const u32 M = 0xFFFFFFu;
        const u32 c[5] = {e[0] & M, __builtin_amdgcn_alignbit(e[1], e[0], 24) & M, __builtin_amdgcn_alignbit(e[2], e[1], 16) & M, e[2] >> 8, e[3] & M};
        const u32 t[5] = {a[0] & M, __builtin_amdgcn_alignbit(a[1], a[0], 24) & M, __builtin_amdgcn_alignbit(a[2], a[1], 16) & M, a[2] >> 8, a[3] & M};
        u32 nl = 0, nt = 0, ng = 0, ntl = 0, types = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const u32 vc = c[i + 1], vl = c[i], vt = t[i + 1], vtl = t[i];
          const bool in = p + i < g.NP;
          const bool e_l = in && vc == vl, e_t = in && vc == vt, e_tl = in && vc == vtl;
          bool gr = in;
#pragma unroll
          for (int b = 0; b < 24; b += 8) gr &= (int)((vc >> b) & 255) == (int)((vl >> b) & 255) + (int)((vt >> b) & 255) - (int)((vtl >> b) & 255);
          nl |= e_l ? 1u << i : 0u;
          nt |= e_t ? 1u << i : 0u;
          ng |= gr ? 1u << i : 0u;
          ntl |= e_tl ? 1u << i : 0u;
          types |= (u32)(e_l ? 1 : e_tl ? 5 : e_t ? 2 : gr ? 4 : 0) << (8 * i);
        }
        ((u32*)ty)[q] = types;
        const int sh = (q & 7) * 4;  // bits r0 .. r0 + 3 of the map: word r0 >> 5
        u32* f32 = (u32*)fm;
        atomicOr(&f32[0 * 48 + (q >> 3)], nl << sh);
        atomicOr(&f32[1 * 48 + (q >> 3)], nt << sh);
        atomicOr(&f32[2 * 48 + (q >> 3)], ng << sh);
        atomicOr(&f32[3 * 48 + (q >> 3)], ntl << sh);
      }
    }
    {
      const int p = tstart + TILE + tid;  // halo pixel TILE + tid
      bool e_l = false, e_t = false, e_tl = false, gr = false;
      if (tid < HALO && p < g.NP) {
        u64 wc, wt;
        __builtin_memcpy(&wc, plane + (size_t)p * 3 - 3, 8);
        __builtin_memcpy(&wt, plane + (size_t)(p - g.W) * 3 - 3, 8);
        const u32 vc = (u32)(wc >> 24) & 0xFFFFFFu, vl = (u32)wc & 0xFFFFFFu, vt = (u32)(wt >> 24) & 0xFFFFFFu, vtl = (u32)wt & 0xFFFFFFu;
        e_l = vc == vl, e_t = vc == vt, e_tl = vc == vtl;
        gr = true;
#pragma unroll
        for (int b = 0; b < 24; b += 8) gr &= (int)((vc >> b) & 255) == (int)((vl >> b) & 255) + (int)((vt >> b) & 255) - (int)((vtl >> b) & 255);
      }
      const u64 ml = __ballot(e_l), mt = __ballot(e_t), mg = __ballot(gr), mtl = __ballot(e_tl);
      if (lane_id() == 0) {
        const int w = (TILE >> 6) + (tid >> 6);  // map word of pixels TILE + 64 wave ..
        fm[0][w] = ml, fm[1][w] = mt, fm[2][w] = mg, fm[3][w] = mtl;
      }
    }
  } else
  for (int k = 0; k < 5; k++) {
    const int r = k * 256 + tid, p = tstart + r;
    int type = 0, fits = 0;
    if (r < TILE + HALO && p < g.NP) {
      int y = ty0, x = tx0 + r;
      if (g.W >= 512) {  // at most TILE + HALO + W pixels past the row start: a few row ends
        while (x >= g.W) x -= g.W, y++;
      } else {
        y = p / g.W, x = p - y * g.W;
      }
      classify_pixel(plane, g, y, x, type, fits);
    }
    if (r < TILE) ty[r] = (u8)type;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      u64 m = __ballot((fits >> b) & 1);
      if (lane_id() == 0) fm[b][r >> 6] = m;
    }
  }
  __syncthreads();
  // The first pixel at or after the start of map word w that does NOT fit predictor b (the next five words are enough: a run is at
  // most 255 pixels): with it the length of the run that would start at a pixel is its own word's trailing ones or, if those reach
  // the word's end, a difference - no loop over words, whose trip count differed from lane to lane (a wave went round it as often
  // as its longest run needed, most waves at least once).
  if (tid < 4 * 25) {
    const int b = tid / 25, w = tid - b * 25;
    int pos = 64 * 24;
#pragma unroll
    for (int k = 4; k >= 0; k--) {
      const u64 m = w + k < 24 ? ~fm[b][w + k] : 0ull;
      pos = m ? 64 * (w + k) + __builtin_ctzll(m) : pos;
    }
    nzw[b][w] = (u16)pos;
  }
  __syncthreads();
  // Where does the path that ENTERS the tile at offset e < HALO leave it, and in a run that began where?  Every pixel r has a
  // successor j(r) in (r, r + 255] - the start of the next run if a run started at r.  Rounds 1-4 squared the successor function
  // over the whole tile (pointer doubling in LDS: two arrays of 1024 words, up to ten rounds of four gathers and a barrier; it
  // was most of this kernel).  Successors only point FORWARD, so the tile is resolved from its end instead: wave w owns the
  // quarter [256 w, 256 w + 256), in four blocks of 64 pixels held in registers, last block first -
  //  * inside a block the paths are squared with lane permutes (no memory, no barrier; a path has left its block of 64 after
  //    three or four rounds as a rule, six at most),
  //  * what lands in a later block of the quarter takes that block's finished answer with one more permute,
  // and the quarters are chained through LDS afterwards: an entry point is in quarter 0, a run is at most 255 long, so its
  // path visits each later quarter once - three gathers.  One barrier, a third of the LDS operations.
  const int lane = lane_id(), wv = tid >> 6;
  u32 vq[4];  // block b of my quarter, pixel 256 wv + 64 b + lane: successor << 16 | start of the run it sits in
  {
    // the successors: a lane takes four pixels in a row (their types one word, their byte for k_runs one store) and hands them
    // to the block order below through the wave's own quarter of `lp`
    const int r0 = wv * 256 + 4 * lane, p0 = tstart + r0;
    u32 jv[4] = {((u32)EXITED << 16) | (u32)r0, ((u32)EXITED << 16) | (u32)(r0 + 1), ((u32)EXITED << 16) | (u32)(r0 + 2), ((u32)EXITED << 16) | (u32)(r0 + 3)};
    if (p0 < g.NP) {
      const u32 types = ((const u32*)ty)[r0 >> 2];
      const size_t at = ((size_t)slot * g.ntiles + tile) * TILE + r0;
      u32 tn[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32 t = (types >> (8 * i)) & 255u;
        // for k_runs: type and length of the run that would start here (it never looks at a pixel again) - type | n << 3, 16 bits
        // (rounds 2-4: one byte with the length cut at 31 and a second, sparse array for the longer ones - whose scattered byte
        // writes cost as much traffic as the second byte does, and whose reads were a trip to memory inside k_runs' walk)
        // (the map of the predictor the type goes by, fit_bit_of_type as a table in a constant: types differ from lane to lane)
        const u32 pb = (0x320100u >> (4u * t)) & 3u;
        const int st = r0 + i + 1, w = st >> 6, o = st & 63;
        const u64 inv = ~(fm[pb][w] >> o);
        const int z = inv ? __builtin_ctzll(inv) : 64;
        const int n = min(z < 64 - o ? z : (int)nzw[pb][w + 1] - st, HALO - 1);
        if (p0 + i < g.NP) jv[i] = ((u32)((p0 + i + 1 + n >= g.NP) ? EXITED : r0 + i + 1 + n) << 16) | (u32)(r0 + i);
        tn[i] = t | ((u32)n << 3);
      }
      *(uint2*)(tnmap + at) = make_uint2(tn[0] | (tn[1] << 16), tn[2] | (tn[3] << 16));  // (entries of pixels past the frame's end are never read)
    }
    *(uint4*)&lp[r0] = make_uint4(jv[0], jv[1], jv[2], jv[3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS is served in order within a wave: what other lanes wrote is there)
#pragma unroll
    for (int k = 0; k < 4; k++) vq[k] = lp[wv * 256 + k * 64 + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the quarter is rewritten below, after the reads)
  }
  // (the word is successor << 16 | run start: "has not left the block" is one unsigned compare of the whole word, and the lane to
  // ask is bits 16..21 - `ds_bpermute` takes a byte address and uses bits 2..7 of it, so word >> 14 is the address as it stands)
#pragma unroll
  for (int b = 3; b >= 0; b--) {
    const u32 bend = (u32)(wv * 256 + b * 64 + 64) << 16;
    u32 v = vq[b];
    for (int it = 0; it < 7; it++) {  // (six squarings cover 64 pixels)
      const bool in = v < bend;  // (successor > pixel >= the block's first)
      if (__ballot(in) == 0) break;
      const u32 t = (u32)__builtin_amdgcn_ds_bpermute((int)(v >> 14), (int)v);
      v = in ? t : v;
    }
    const u32 blk = v >> 22, at = v >> 14;  // successor / 64: the block it lies in
#pragma unroll
    for (int b2 = b + 1; b2 < 4; b2++) {  // a later block of my quarter: its answers are final (they point past the quarter)
      const u32 t = (u32)__builtin_amdgcn_ds_bpermute((int)at, (int)vq[b2]);
      v = blk == (u32)(wv * 4 + b2) ? t : v;
    }
    vq[b] = v;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) lp[wv * 256 + k * 64 + lane] = vq[k];
  __syncthreads();
  if (tid < HALO) {
    u32 v = lp[tid];
#pragma unroll
    for (int q = 0; q < 3; q++) {  // quarters 1, 2, 3
      const u32 j = v >> 16;
      if (j < (u32)TILE) v = lp[j];
    }
    const int j = (int)(v >> 16);
    u8* row = exitmap + ((size_t)slot * g.ntiles + tile) * 512;
    // 255: the path ended inside this tile (frame end); the type of the run that crosses the border
    *(u16*)(row + 2 * tid) = (u16)((u32)(u8)(j == EXITED ? 255 : j - TILE) | ((u32)ty[v & 0xFFFFu] << 8));
  }
}

// one wave per frame walks its tiles: entry[tile] = {offset of the first run
// start inside the tile, type of the run active when the tile begins}
__global__ __launch_bounds__(256) void k_entries(const u8* __restrict__ exitmap, u8* __restrict__ entry, Geom g, const int* __restrict__ slots) {
  // (the exit maps of 64 tiles, 32 KB, are brought in by the whole workgroup with 16-byte loads - one wave with 4-byte loads took
  // 0.2 of this kernel's 0.3 ms for them, between k_tiles and k_runs on the encoder's critical path - and walked by one lane,
  // while the other waves bring in the next 64: two buffers)
  __shared__ __attribute__((aligned(16))) u32 buf[2][64 * 128];
  const int slot = slots[blockIdx.x], tid = threadIdx.x;
  const uint4* const src0 = (const uint4*)(exitmap + (size_t)slot * g.ntiles * 512);
  auto stage = [&](int base, int which, int first, int step) {
    const int n = min(64, g.ntiles - base);
    const uint4* srcw = src0 + (size_t)base * 32;
    for (int i = first; i < n * 32; i += step) ((uint4*)buf[which])[i] = srcw[i];
  };
  stage(0, 0, tid, 256);
  __syncthreads();
  int e = 0, tin = 0, which = 0;
  for (int base = 0; base < g.ntiles; base += 64, which ^= 1) {
    const int n = min(64, g.ntiles - base);
    if (tid >= 64) {
      if (base + 64 < g.ntiles) stage(base + 64, which ^ 1, tid - 64, 192);  // the next batch, by the waves that do not walk
    } else if (tid == 0) {
      const u8* b8 = (const u8*)buf[which];
      for (int k = 0; k < n; k++) {
        *(u16*)(entry + ((size_t)slot * g.ntiles + base + k) * 2) = (u16)((u32)e | ((u32)tin << 8));
        if (e < HALO) {
          const u32 nx = ((const u16*)b8)[k * 256 + e];  // exit | type of the run that crosses << 8
          e = (int)(nx & 255u);
          tin = (int)(nx >> 8);
        }
      }
    }
    __syncthreads();
  }
}

// The runs of every tile, from where k_entries says the frame's path enters it: one LANE per tile walks the path (the type and
// the length of the run that would start at any pixel were written by the first pass: one byte per step), ~110 steps on average,
// and writes the tile's run records in order.  The tiles of a frame are independent once their entry points are known, and a
// frame has 2000 of them: the walk needs no pointer doubling (the second k_tiles pass it replaces re-classified every pixel and
// squared the successor function nine times to mark the same path).
// tilecnt: {runs, literal runs} per tile.   runrec: rel_start | type<<10 | n<<16
// A fixed number of workgroups (two per CU: the host's choice) take the (frame, group of 256 tiles) pairs in launch order,
// grid-stride: the lines of the type/length map being walked at any moment are those of ~512 pairs and stay in the L2.  (The
// first form bought the same limit with 65 000 bytes of LDS it never touched - and starved beside another kernel that holds LDS.)
// Round 5: the map is 16 bits per pixel (type | length << 3); rounds 2-4 had a byte with the length cut at 31 and a second, sparse
// array for the longer ones, i.e. a second dependent load in the walk for every run of 32 pixels or more.  (Also tried this
// round: the maps of 32 tiles copied into LDS and walked there by 32 lanes - 4.6 ms instead of 1.8: the walk is ~45 instructions
// per step on lanes that finish at very different times, and LDS holds a sixth of the tiles that registers and the L2 keep in flight.)
// (Round 5, second form) Tiles are handed out ONE AT A TIME: a frame's tiles hold anything from two runs to five hundred (flat
// areas and text: median 20, mean 112, the busiest of 64 neighbours 308), and with a tile fixed to a lane a wave lasted as long as
// its busiest tile, 2.8 times the average.  The loop below is flat - every round a lane either takes a step of its walk or, if its
// tile is finished, writes the tile's tail and moves to its next one.  A workgroup owns the groups of 256 tiles b, b + nWG, ...
// of the call (as before) and hands their tiles to its lanes in order from a counter in LDS; a lane's next tile was drawn two
// moves ago and its entry point asked for one move ago, so that a move waits for LDS only (the frame list sits there too).
// (First attempt: a global counter and the frame list in memory - a move then waits for two loads and an atomic, and with 64
// lanes some lane moves in almost every round: 5.1 ms.)
constexpr int KRUNS_MAXSLOTS = 4096;
__global__ __launch_bounds__(256) void k_runs(Geom g, const int* __restrict__ slots, int nslots, const u8* __restrict__ entry, const u16* __restrict__ tnmap, u32* __restrict__ runrec,
                                              u32* __restrict__ tilecnt, u32 rcp_ntiles) {
  __shared__ int s_slots[KRUNS_MAXSLOTS];
  __shared__ u32 s_ticket;
  for (int i = threadIdx.x; i < nslots; i += 256) s_slots[i] = slots[i];
  if (threadIdx.x == 0) s_ticket = 768u;
  __syncthreads();
  const u32 total = (u32)nslots * (u32)g.ntiles, nwg = gridDim.x, wg = blockIdx.x;
  auto tile_of = [&](u32 t) -> u32 { return ((t >> 8) * nwg + wg) * 256u + (t & 255u); };  // the workgroup's t-th tile (>= total: none left)
  auto index_of = [&](u32 T, int& tile) -> size_t {  // tile T of the call (frame T / ntiles of the list) -> its index in the maps
    u32 fi = __umulhi(T, rcp_ntiles);
    fi -= (fi * (u32)g.ntiles > T) ? 1u : 0u;        // (the reciprocal is rounded up ...
    fi += ((fi + 1u) * (u32)g.ntiles <= T) ? 1u : 0u;  // ... or, for one tile per frame, cut to 32 bits)
    tile = (int)(T - fi * (u32)g.ntiles);
    return (size_t)s_slots[fi] * g.ntiles + (size_t)tile;
  };
  u32 T = tile_of(threadIdx.x), Tn = tile_of(256u + threadIdx.x), Tnn = tile_of(512u + threadIdx.x);
  u32 en = 255u;  // entry point of tile Tn (255: a run from an earlier tile already reached the frame end - or no such tile)
  size_t tin = 0;
  int tilen = 0;
  if (Tn < total) {
    tin = index_of(Tn, tilen);
    en = entry[tin * 2];
  }
  size_t ti = 0;
  const u16* tn = nullptr;
  u32* rec = nullptr;
  int tstart = 0, r = TILE, cnt = 0, lit = 0;
  u32 q0 = 0, q1 = 0, q2 = 0;  // records leave four at a time (16-byte stores: a quarter of the partial-line writes)
  auto begin = [&](size_t idx, int tile, u32 e) __attribute__((always_inline)) {
    ti = idx;
    tn = tnmap + ti * TILE;
    rec = runrec + ti * TILE;
    tstart = g.p0 + tile * TILE;
    r = e < (u32)HALO ? (int)e : TILE;
    cnt = 0, lit = 0;
  };
  if (T < total) {
    int tile0;
    const size_t idx = index_of(T, tile0);
    begin(idx, tile0, entry[idx * 2]);
  }
  while (T < total) {
    if (r < TILE && tstart + r < g.NP) {
      const u32 v = tn[r];
      const int t = (int)(v & 7u);
      const int n = (int)(v >> 3);
      int j = r + 1 + n;
      int len = j - r;
      if (tstart + j >= g.NP) {  // the run reaches the end of the frame
        len = g.NP - (tstart + r);
        j = TILE;
      }
      const u32 rv = (u32)r | ((u32)t << 10) | ((u32)len << 16);
      const int k = cnt & 3;
      if (k == 3) ((uint4*)rec)[cnt >> 2] = make_uint4(q0, q1, q2, rv);
      q0 = k == 0 ? rv : q0;
      q1 = k == 1 ? rv : q1;
      q2 = k == 2 ? rv : q2;
      cnt++;
      lit += t == 0;
      r = j;
    } else {
      {  // the last, partial group of records
        const int k = cnt & 3, base = cnt & ~3;
        if (k > 0) rec[base] = q0;
        if (k > 1) rec[base + 1] = q1;
        if (k > 2) rec[base + 2] = q2;
      }
      tilecnt[ti * 2] = (u32)cnt;
      tilecnt[ti * 2 + 1] = (u32)lit;
      T = Tn;
      if (T < total) begin(tin, tilen, en);
      Tn = Tnn;
      en = 255u;
      if (Tn < total) {
        tin = index_of(Tn, tilen);
        en = entry[tin * 2];
      }
      Tnn = tile_of(atomicAdd(&s_ticket, 1u));
    }
  }
}

// runs of identical pixels over raster pixels 0..W (first row and pixel (0,1)):
// hdrrec[j] = start | n<<16  (CompressI, screencap.cpp:346-362)
__global__ __launch_bounds__(64) void k_header(const u8* __restrict__ planes, Geom g, const int* __restrict__ slots, u32* __restrict__ hdrrec, u32* __restrict__ hdrcnt) {
  __shared__ u64 nq[80];
  const int slot = slots[blockIdx.x], lane = threadIdx.x;
  const u8* plane = planes + (size_t)slot * g.plane_stride;
  const int n = g.W + 1, words = (n + 63) >> 6;
  for (int w = 0; w < words; w++) {
    int k = w * 64 + lane;
    bool ne = false;
    if (k >= 1 && k < n) {
      const u8* a = k < g.W ? plane + k * 3 : plane + g.S;
      const u8* b = plane + (k - 1) * 3;
      ne = ld3(a) != ld3(b);
    }
    u64 m = __ballot(ne);
    if (lane == 0) nq[w] = m;
  }
  __syncthreads();
  if (lane == 0) {
    u32* rec = hdrrec + (size_t)slot * (g.W + 2);
    int start = 0, cnt = 0;
    while (start < n) {
      // next position > start with a differing pixel, or start+255
      int lim = min(n, start + 255), k = start + 1, nxt = lim;
      while (k < lim) {
        u64 m = nq[k >> 6] >> (k & 63);
        if (m) {
          int z = k + __builtin_ctzll(m);
          if (z < lim) nxt = z;
          break;
        }
        k = (k | 63) + 1;
      }
      rec[cnt++] = (u32)start | ((u32)(nxt - start) << 16);
      start = nxt;
    }
    hdrcnt[slot] = (u32)cnt;
  }
}

// per frame: exclusive scans of {runs, literals} over its tiles
__global__ __launch_bounds__(256) void k_scan_tiles(const u32* __restrict__ tilecnt, u32* __restrict__ tileoff, u32* __restrict__ frametot, Geom g, const int* __restrict__ slots) {
  __shared__ int wsum[5];
  const int slot = slots[blockIdx.x], tid = threadIdx.x;
  int run_base = 0, lit_base = 0;
  for (int base = 0; base < g.ntiles; base += 256) {
    int t = base + tid;
    int r = t < g.ntiles ? (int)tilecnt[((size_t)slot * g.ntiles + t) * 2] : 0;
    int l = t < g.ntiles ? (int)tilecnt[((size_t)slot * g.ntiles + t) * 2 + 1] : 0;
    int tr = 0, tl = 0;
    int ro = block_excl_scan(r, &tr, wsum);
    __syncthreads();
    int lo = block_excl_scan(l, &tl, wsum);
    __syncthreads();
    if (t < g.ntiles) {
      tileoff[((size_t)slot * g.ntiles + t) * 2] = (u32)(run_base + ro);
      tileoff[((size_t)slot * g.ntiles + t) * 2 + 1] = (u32)(lit_base + lo);
    }
    run_base += tr;
    lit_base += tl;
  }
  if (tid == 0) {
    frametot[slot * 2] = (u32)run_base;
    frametot[slot * 2 + 1] = (u32)lit_base;
  }
}

// per batch: frame bases (one thread).  kinds: 0 key frame, 1 flat (no symbols), 2 P-frame
constexpr u64 kChunkTotalLimit = 0xFFFF0000ull;  // runs / coder entries / colour symbols / P-frame symbols of one encode chunk (32-bit positions)
struct FrameBase {
  u32 run_base, sym_base, col_base, misc_base, nruns, nsyms, ncol, nmisc, hdr_runs, nbt, pad0, pad1;  // pad0 / pad1: where the frame's literals go in its generation's plane arrays, and their stride (the host fills them in: encode_chunk)
};
__global__ __launch_bounds__(64) void k_bases(const int* __restrict__ kinds, const int* __restrict__ pidx, int nfr, const u32* __restrict__ frametot,
                                              const u32* __restrict__ hdrcnt, const u32* __restrict__ ptot, FrameBase* __restrict__ bases, u32* __restrict__ totals, u64 limit) {
  if (blockIdx.x != 0) return;
  // Bases are 32-bit (stream positions travel through the sort as 32-bit values), the sums are kept in 64 bits: totals[4] =
  // the number of leading frames whose bases and totals all stay below 2^32 - the host cuts the chunk there and codes the
  // rest in the next one (an incompressible 1080p frame has ~10 M symbols: 413 of them, or 103 at 4K, pass 2^32).
  // One wave, 64 frames a round: the counts are loaded side by side and summed by wave scans (one thread walking the frames
  // was a dependent load per frame, 83 us for 300 frames between the classification and k_symbols).
  const int lane = threadIdx.x;
  auto scan64 = [&](u64 v) {  // inclusive, over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u64 t = (u64)__shfl_up((unsigned long long)v, d);
      if (lane >= d) v += t;
    }
    return v;
  };
  u64 rb = 0, sb = 0, cb = 0, mb = 0;  // totals of the rounds before
  u32 nfit = (u32)nfr;
  for (int base = 0; base < nfr; base += 64) {
    const int i = base + lane;
    FrameBase b;
    b.run_base = b.sym_base = b.col_base = b.misc_base = 0;
    b.nruns = b.nsyms = b.ncol = b.nmisc = b.hdr_runs = b.nbt = b.pad0 = b.pad1 = 0;
    if (i < nfr) {
      const int kind = kinds[i];
      if (kind == 0) {
        const int slot = i;  // planes of a chunk sit in slots 0..n-1
        const u32 R = frametot[slot * 2], L = frametot[slot * 2 + 1], Hr = hdrcnt[slot];
        b.nruns = Hr + R;
        b.nsyms = 4 * Hr + 2 * R + 3 * L;
        b.ncol = 3 * (Hr + L);
        b.hdr_runs = Hr;
      } else if (kind == 2) {
        const u32* t = ptot + (size_t)pidx[i] * 8;
        b.nruns = t[0];
        b.nsyms = t[1];
        b.ncol = t[2];
        b.nmisc = t[3];
        b.nbt = t[4];
      }
    }
    const u64 ri = rb + scan64(b.nruns), si = sb + scan64(b.nsyms), ci = cb + scan64(b.ncol), mi = mb + scan64(b.nmisc);  // sums up to and with frame i
    b.run_base = (u32)(ri - b.nruns);
    b.sym_base = (u32)(si - b.nsyms);
    b.col_base = (u32)(ci - b.ncol);
    b.misc_base = (u32)(mi - b.nmisc);
    if (i < nfr) bases[i] = b;
    const u64 most = max(max(ri, si), max(ci, mi));
    const u64 over = __ballot(i < nfr && most >= limit);
    if (over && nfit == (u32)nfr) nfit = (u32)(base + __builtin_ctzll(over));
    rb = (u64)__shfl((unsigned long long)ri, 63), sb = (u64)__shfl((unsigned long long)si, 63), cb = (u64)__shfl((unsigned long long)ci, 63), mb = (u64)__shfl((unsigned long long)mi, 63);
  }
  if (lane == 0) {
    totals[0] = (u32)rb;
    totals[1] = (u32)sb;
    totals[2] = (u32)cb;
    totals[3] = (u32)mb;
    totals[4] = nfit;
  }
}

// colour context ids from the two previous bytes (SC_CXSHIFT = 2, MAKECX1,
// screencap.h:35-36; WritePixel/EncodeRGB, screencap.cpp:609-643).  The three pairs of a literal go to the three PLANE arrays of
// its generation (idx, idx + stride, idx + 2 * stride; stride = the generation's literals): scpr_ctxsort.hpp sorts each plane's
// share by context and needs generation and plane from nobody.
__device__ __forceinline__ void emit_colour(u32 gen, u32 pix, u32 prev_g, u32 prev_b, u32 pos, u32 idx, u32 stride, u32* __restrict__ keys, u32* __restrict__ vals) {
  const u32 c0 = pix & 255, c1 = (pix >> 8) & 255, c2 = (pix >> 16) & 255;
  const u32 cx0 = (prev_b >> 2) | ((prev_g >> 2) << 6);
  const u32 cx1 = (c0 >> 2) | ((prev_b >> 2) << 6);
  const u32 cx2 = (c1 >> 2) | ((c0 >> 2) << 6);
  const u32 k0 = cx0, k1 = 4096 + cx1, k2 = 8192 + cx2;
  keys[idx] = (gen << 22) | (k0 << 8) | c0;
  keys[idx + stride] = (gen << 22) | (k1 << 8) | c1;
  keys[idx + 2 * stride] = (gen << 22) | (k2 << 8) | c2;
  vals[idx] = pos;
  vals[idx + stride] = pos + 1;
  vals[idx + 2 * stride] = pos + 2;
}

// Chain offsets from the sorted keys: cstart[q] = first sorted position whose chain id (generation * NCOLCTX +
// plane/context) is >= q, for q = 0 .. nchains (so cstart[nchains] = n).  One thread per sorted position plus a
// sentinel; a thread fills the (usually empty) gap of chains between its predecessor and itself.  This replaces a
// histogram of one global atomic per colour symbol, most of them on a handful of hot contexts.
// The same pass proves the order it relies on (free here: both keys are in registers anyway; rounds 1-4 sorted with rocPRIM,
// which has returned unsorted output, and the proof stayed when scpr_ctxsort.hpp took over): a key below its predecessor - or
// one that names no chain of this call - sets bit 5 of *err and writes nothing.
__global__ __launch_bounds__(256) void k_chain_starts(const u32* __restrict__ skeys, u32 n, u32 nchains, u32* __restrict__ cstart, u32* __restrict__ err) {
  // four positions per thread (one 16-byte load and the key before them); position n is the sentinel
  const u32 i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  auto chain = [](u32 key) { return (key >> 22) * (u32)NCOLCTX + ((key >> 8) & 0x3FFFu); };
  u32 k[5] = {0u, 0u, 0u, 0u, 0u};  // keys i0 - 1 .. i0 + 3
  if (i0 + 4 <= n) {
    const uint4 v = *(const uint4*)(skeys + i0);
    k[1] = v.x, k[2] = v.y, k[3] = v.z, k[4] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) k[1 + j] = i0 + j < n ? skeys[i0 + j] : 0u;
  }
  if (i0 > 0 && i0 <= n) k[0] = skeys[i0 - 1];
  const int lane = threadIdx.x & 63;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u32 i = i0 + (u32)j;
    const bool live = i <= n;
    const u32 qi = i < n ? chain(k[1 + j]) : nchains;
    const u32 q0 = i > 0 ? chain(k[j]) + 1 : 0u;
    const bool b = (i > 0 && i < n && (k[1 + j] >> 8) < (k[j] >> 8)) || qi > nchains;
    bad |= b && live;
    // the gaps are filled by the whole wave, one boundary after the other (a lane by itself went round a loop of thousands of
    // empty chains wherever two used contexts lie far apart, with 63 lanes waiting for it)
    u64 todo = __ballot(live && !b && q0 <= qi);
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1;
      const u32 a = (u32)__builtin_amdgcn_readlane((int)q0, l), e = (u32)__builtin_amdgcn_readlane((int)qi, l), at = (u32)__builtin_amdgcn_readlane((int)i, l);
      for (u32 q = a + (u32)lane; q <= e; q += 64) cstart[q] = at;
    }
  }
  if (bad) atomicOr(err, 32u);
}
__global__ void k_debug_swap(u32* a, u32 i, u32 j) {  // scpr_debug_inject(2)
  const u32 t = a[i];
  a[i] = a[j];
  a[j] = t;
}

// unified run list + colour symbols.  grid = (ceil(ntiles / 4) + 1, frames): four tiles to a workgroup, the extra
// block handles the header runs.
__global__ __launch_bounds__(256) void k_symbols(const u8* __restrict__ planes, Geom g, const int* __restrict__ slots, const int* __restrict__ gens, const int* __restrict__ fidx,
                                                 const FrameBase* __restrict__ bases, const u32* __restrict__ runrec, const u32* __restrict__ tilecnt,
                                                 const u32* __restrict__ tileoff, const u8* __restrict__ entry, const u32* __restrict__ hdrrec,
                                                 u32* __restrict__ runs, u32* __restrict__ runpos, u32* __restrict__ keys, u32* __restrict__ vals) {
  const int fi = blockIdx.y, slot = slots[fi], tid = threadIdx.x;
  const u32 gen = (u32)gens[fi];
  const FrameBase fb = bases[fidx[fi]];
  const u8* plane = planes + (size_t)slot * g.plane_stride;
  const u32 Hr = fb.hdr_runs;
  const int tgroups = (g.ntiles + 3) >> 2;
  if ((int)blockIdx.x == tgroups) {  // header runs: C C C N each, no pixel-type symbol
    const u32* rec = hdrrec + (size_t)slot * (g.W + 2);
    for (u32 j = tid; j < Hr; j += 256) {
      u32 r = rec[j], start = r & 0xFFFF, n = r >> 16;
      u32 pos = fb.sym_base + 4 * j;
      runs[fb.run_base + j] = make_run(0, 0, (int)n, true);
      runpos[fb.run_base + j] = pos;
      const u8* px = start < (u32)g.W ? plane + start * 3 : plane + g.S;
      u32 pg = 0, pb = 0;
      if (start > 0) {  // previous pixel in raster order (start <= W, so it is in row 0)
        const u8* pp = plane + (start - 1) * 3;
        pg = pp[1];
        pb = pp[2];
      }
      emit_colour(gen, ld3(px), pg, pb, pos, fb.pad0 + j, fb.pad1, keys, vals);
    }
    return;
  }
  // One WAVE per tile, four tiles to a workgroup: half of a frame's tiles hold 20 runs or fewer (flat areas), and a workgroup of
  // 256 threads with two barriers per round spent most of its lanes and its time on nothing (round 5: 1.0 -> 0.61 ms).  A
  // literal's rank among the literals of its round is a ballot.
  const int lane = lane_id(), tile = (int)blockIdx.x * 4 + (tid >> 6);
  if (tile >= g.ntiles) return;
  const size_t ti = (size_t)slot * g.ntiles + tile;
  const int cnt = (int)tilecnt[ti * 2];
  const u32 run_off = tileoff[ti * 2], lit_off = tileoff[ti * 2 + 1];
  const u32* rec = runrec + ti * TILE;
  const int tstart = g.p0 + tile * TILE;
  int lit_run = 0;  // literals in earlier rounds of this tile
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    u32 r = 0;
    int type = -1, lastt = 0;
    if (i < cnt) {
      r = rec[i];
      type = (r >> 10) & 7;
      lastt = i > 0 ? (int)((rec[i - 1] >> 10) & 7) : (int)entry[ti * 2 + 1];
    }
    const u64 lm = __ballot(type == 0);
    const int lrank = __popcll(lm & lanemask_lt()), tl = __popcll(lm);
    if (i < cnt) {
      const u32 n = r >> 16, rel = r & 1023;
      const u32 litidx = lit_off + lit_run + lrank;
      const u32 pos = fb.sym_base + 4 * Hr + 2 * (run_off + i) + 3 * litidx;
      const u32 ri = fb.run_base + Hr + run_off + i;
      runs[ri] = make_run(type, lastt, (int)n, false);
      runpos[ri] = pos;
      if (type == 0) {
        const int p = tstart + (int)rel;
        const int y = p / g.W, x = p - y * g.W;
        const u8* px = plane + (size_t)y * g.S + x * 3;
        const u8* pp = x > 0 ? px - 3 : plane + (size_t)(y - 1) * g.S + (g.W - 1) * 3;
        emit_colour(gen, ld3(px), pp[1], pp[2], pos + 1, fb.pad0 + Hr + litidx, fb.pad1, keys, vals);
      }
    }
    lit_run += tl;
  }
}

// --------------------------------------------------------- fixed chains ---
// (the chains themselves are in scpr_fixed.hpp)
struct GenRange {
  u32 run_begin, run_end;
};
struct FixedPersist {  // one fixed-alphabet table as kept between calls (P-frames continue the models, screencap.cpp:1118)
  u32 freq[512], cum[512], cnt[512];
  int total, valid, pad0, pad1;
};
// -------------------------------------------------------- colour chains ---
struct Arena {
  DenseTab* tabs;  // cap + 1 tables: 0 .. cap - 1 for contexts, table `cap` is the sink of an overflow (WaveModel::alloc_dense)
  u32* top;
  u32 cap;
  u32* err;
};
// (the chains themselves are in scpr_wave.hpp: one wave per chain, WaveModel)
// After a call that ran several generations only the last one's tables are needed again, and they sit anywhere below the
// arena's top among the tables of the generations that ended: the live records' tables move to the bottom of a second
// arena, in record order, and the records get the new indices.  One wave per record (word 0: kind in the low byte - 6 and 7
// are the kinds with a table -, word 2: table index; stamp_word >= 0: only records that carry `stamp` there are live).
__global__ __launch_bounds__(64) void k_compact_tables(u32* __restrict__ recs, int words, int nrec, int stamp_word, u32 stamp, const DenseTab* __restrict__ src, DenseTab* __restrict__ dst,
                                                       u32* __restrict__ counter) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r >= nrec) return;
  u32* rec = recs + (size_t)r * words;
  const u32 kind = rec[0] & 255u;
  if (kind != 6u && kind != 7u) return;
  if (stamp_word >= 0 && rec[stamp_word] != stamp) return;
  u32 idx = 0;
  if (lane == 0) idx = atomicAdd(counter, 1u);
  idx = (u32)__builtin_amdgcn_readfirstlane((int)idx);
  const u32* s = (const u32*)&src[rec[2]];
  u32* d = (u32*)&dst[idx];
  static_assert(sizeof(DenseTab) == 64 * 6 * 4, "one table is six words per lane");
#pragma unroll
  for (int k = 0; k < 6; k++) d[k * 64 + lane] = s[k * 64 + lane];
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0) rec[2] = idx;
}

// ------------------------------------------------------------------ rANS ---
struct RansBlock {
  u32 begin, len;
};
// One lane per block of <= 131072 entries, processed last to first; bytes are
// written backwards into the block's scratch (ransmt.h:116-134).  x / freq uses
// the exact 32-bit reciprocal (rans_byte.h:171-240).
// One lane per block of <= 131072 entries, processed last to first; bytes are written backwards into the
// block's scratch (ransmt.h:116-134).  The chain over the coder state is the whole cost (a lone wave issues a
// dependent instruction every ~8 cycles), so the workgroup is a three-stage pipeline of waves (two of them feed), one trip
// (16 entries per lane) apart, handing over through LDS:
//   feeder  loads the entries, looks up the reciprocals, lays a 16-byte record per entry:
//           { x_max, reciprocal, (4096 - freq) | shift << 24, bias | raw byte << 16 | raw << 31 }
//   coder   only advances the state (13 instructions per entry) and passes { state before, bytes out } on
//   writer  turns that into bytes: the renormalisation emits at most two (x < 2^31, x_max = freq << 19 >= 2^19);
//           both are stored every time, as one 16-bit store just below the write position, which then moves by
//           the number that count; what is left below is overwritten by the next entry (the block's scratch
//           has room for two spare bytes).
// The state update is ryg's x + bias + (x / freq) * (4096 - freq) (rans_byte.h:199-240) with the exact 32-bit
// reciprocal, which also covers freq = 1 (bias + 4095, quotient x - 1).  A raw byte (freq 0) is the same code
// with a limit no state reaches and a zero multiplier; so is the padding of a short block.
__device__ __forceinline__ uint4 rans_record(u32 v, const RansRcp r) {
  const u32 fr = v & 0xFFFF, cf = v >> 16;
  const bool raw = fr == 0;
  uint4 o;
  o.x = raw ? 0xFFFFFFFFu : fr << 19;                                   // x_max = ((L >> 12) << 8) * freq
  o.y = raw ? 0u : r.rcp;                                               // raw: quotient 0
  o.z = raw ? 0u : ((u32)kProbScale - fr) | ((u32)r.shift << 24);       // multiplier in the low 24 bits (what a 24-bit multiply reads), shift above
  o.w = raw ? (0x80000000u | (cf << 16)) : cf + r.pad;                  // bias in the low 16 bits; raw: flag in the sign bit, the byte in bits 16-23
  return o;
}
// coder: returns what the writer needs, { state before the step, bytes out (0..2) | raw flag and byte }
__device__ __forceinline__ uint2 rans_step(u32& x, const uint4 r) {
  const u32 n = (u32)(x >= r.x) + (u32)((x >> 8) >= r.x);
  const uint2 pass = make_uint2(x, n | (r.w & 0xFFFF0000u));
  const u32 xs = x >> (8 * n);
  const u32 q = __umulhi(xs, r.y) >> (r.z >> 24);
  x = __umul24(q, r.z) + (xs + (r.w & 0xFFFFu));
  return pass;
}
// writer: straight-line vector code (selects, no exec-masked branches)
__device__ __forceinline__ void rans_emit(u8* const base, u32& off, const uint2 pass) {
  const u32 n = pass.y & 3u;
  const u32 rawmask = (u32)((int)pass.y >> 31);                         // all ones for a raw byte
  const u32 swapped = __builtin_amdgcn_perm(0u, pass.x, 0x0c0c0001u);   // bytes 0,1 of the state exchanged: the first byte out sits at the higher address
  const u32 rawval = __builtin_amdgcn_perm(0u, pass.y, 0x0c0c020cu);    // the raw byte in byte 1
  const u16 o16 = (u16)((rawmask & rawval) | (~rawmask & swapped));
  __builtin_memcpy(base + (size_t)off - 2, &o16, 2);                    // just below the write position
  off -= n - rawmask;                                                   // n bytes, or one raw byte
}
constexpr int RANS_TRIP = 32;  // entries per hand-over between the four waves (a barrier each: 16 -> 32 took the stage from 8.65 to 7.5 ms; 40 is slower again, the unrolled trip outgrows its registers)
__global__ __launch_bounds__(256) void k_rans(const u32* __restrict__ entries, const RansBlock* __restrict__ blocks, int nblocks, const RansRcp* __restrict__ rcp_g,
                                              u8* __restrict__ scratch, u32* __restrict__ blksize, const u32* __restrict__ err) {
  if (*err & 32u) {  // the colour symbols were not sorted (k_chain_starts): their entries were never written - nothing to code
    const int b0 = blockIdx.x * 64 + (int)threadIdx.x;
    if (threadIdx.x < 64 && b0 < nblocks) blksize[b0] = 0;
    return;
  }
  __shared__ RansRcp lrcp[kProbScale + 1];          // reciprocals in LDS: the lookup is off the HBM path
  __shared__ uint4 rec[2][RANS_TRIP][64];           // feeders -> coder, two trips
  __shared__ uint2 hand[2][RANS_TRIP][64];          // coder -> writer, two trips
  __shared__ u32 xfinal[64];
  for (int i = threadIdx.x; i <= kProbScale; i += 256) lrcp[i] = rcp_g[i];
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;  // 0 coder, 1 and 2 feeders (half a trip each), 3 writer
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < nblocks;
  RansBlock blk{0, 0};
  if (live) blk = blocks[b];
  int trips = (int)((blk.len + RANS_TRIP - 1) / RANS_TRIP);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) trips = max(trips, __shfl_xor(trips, d));
  __syncthreads();
  const u32* e = entries + blk.begin;
  constexpr int HALF = RANS_TRIP / 2;
  const int k0 = role == 2 ? HALF : 0;  // the feeder's half of a trip
  // entries of trip t: [len - 16 (t + 1), len - 16 t); indices below 0 do not exist (the short trip of a block comes last)
  auto fetch = [&](int t, u32 v[HALF]) {
    const int lo = (int)blk.len - RANS_TRIP * (t + 1) + k0;
#pragma unroll
    for (int g4 = 0; g4 < HALF / 4; g4++) {
      const int i = lo + 4 * g4;
      if (i >= 0) {
        uint4 q;
        __builtin_memcpy(&q, e + i, 16);
        v[4 * g4] = q.x;
        v[4 * g4 + 1] = q.y;
        v[4 * g4 + 2] = q.z;
        v[4 * g4 + 3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[4 * g4 + k] = i + k >= 0 ? e[i + k] : 0xFFFFFFFFu;  // 0xFFFFFFFF: no entry (a real one has freq <= 4096)
      }
    }
  };
  auto lay = [&](int t, const u32 v[HALF]) {
#pragma unroll
    for (int k = 0; k < HALF; k++) {
      uint4 r = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);  // padding: nothing out, state unchanged
      if (v[k] != 0xFFFFFFFFu) r = rans_record(v[k], lrcp[v[k] & 0xFFFF]);  // entry 0 of the table is that of freq 1
      rec[t & 1][k0 + k][lane] = r;
    }
  };
  const bool feeder = role == 1 || role == 2;
  u8* const base = scratch + (size_t)(live ? b : 0) * RANS_SCRATCH;  // the block's scratch; the write position is an offset that runs down from its top
  u32 off = RANS_SCRATCH;
  u32 x = kRansL;
  u32 nxt[HALF];  // feeder: the entries of the trip after the one being laid out (their loads are in flight meanwhile)
  if (feeder && trips > 0) {
    u32 cur[HALF];
    fetch(0, cur);
    if (trips > 1) fetch(1, nxt);
    lay(0, cur);
  }
  __syncthreads();
  for (int t = 0; t <= trips; t++) {  // one extra turn: the writer runs a trip behind the coder
    if (feeder) {
      if (t + 1 < trips) {
        u32 cur[HALF];
#pragma unroll
        for (int k = 0; k < HALF; k++) cur[k] = nxt[k];
        if (t + 2 < trips) fetch(t + 2, nxt);
        lay(t + 1, cur);
      }
    } else if (role == 0) {
      if (t < trips) {
#pragma unroll
        for (int k = RANS_TRIP - 1; k >= 0; k--) hand[t & 1][k][lane] = rans_step(x, rec[t & 1][k][lane]);
        if (t == trips - 1) xfinal[lane] = x;
      }
    } else if (t > 0 && live) {
#pragma unroll
      for (int k = RANS_TRIP - 1; k >= 0; k--) rans_emit(base, off, hand[(t - 1) & 1][k][lane]);
    }
    __syncthreads();
  }
  if (role == 3 && live) {
    const u32 xf = trips > 0 ? xfinal[lane] : (u32)kRansL;
    u8* p = base + off - 4;  // RansEncFlush, rans_byte.h:90-102
    p[0] = (u8)xf;
    p[1] = (u8)(xf >> 8);
    p[2] = (u8)(xf >> 16);
    p[3] = (u8)(xf >> 24);
    blksize[b] = (u32)RANS_SCRATCH - (off - 4);
  }
}

// ---------------------------------------------------------------- gather ---
struct Packet {
  u32 hdr_len;     // bytes in front of the coded blocks (1..4)
  u32 hdr;         // those bytes, little endian
  u32 blk_begin, blk_count;
};
// sizes[f], pktoff[f], blkdst[b]; totals[3] = total bytes
__global__ __launch_bounds__(256) void k_offsets(const Packet* __restrict__ pk, int nfr, const u32* __restrict__ blksize, u32* __restrict__ sizes, u64* __restrict__ pktoff,
                                                 u64* __restrict__ blkdst, u64* __restrict__ total) {
  __shared__ int wsum[5];
  __shared__ u64 carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nfr; base += 256) {
    const int f = base + tid;
    int sz = 0;
    if (f < nfr) {
      sz = (int)pk[f].hdr_len;
      for (u32 b = 0; b < pk[f].blk_count; b++) sz += (int)blksize[pk[f].blk_begin + b];
    }
    int tot = 0;
    int off = block_excl_scan(sz, &tot, wsum);
    __syncthreads();
    if (f < nfr) {
      u64 o = carry + (u64)off;
      sizes[f] = (u32)sz;
      pktoff[f] = o;
      o += pk[f].hdr_len;
      for (u32 b = 0; b < pk[f].blk_count; b++) {
        blkdst[pk[f].blk_begin + b] = o;
        o += blksize[pk[f].blk_begin + b];
      }
    }
    __syncthreads();
    if (tid == 0) carry += (u64)tot;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}
// grid.x = nblocks + nframes: copies a coded block, or writes a packet header
__global__ __launch_bounds__(256) void k_gather(const Packet* __restrict__ pk, int nfr, int nblocks, const u8* __restrict__ scratch, const u32* __restrict__ blksize,
                                                const u64* __restrict__ pktoff, const u64* __restrict__ blkdst, u8* __restrict__ out, u64 out_cap, u32* __restrict__ err) {
  const int b = blockIdx.x;
  if (b < nblocks) {
    const u32 sz = blksize[b];
    const u8* s = scratch + (size_t)(b + 1) * RANS_SCRATCH - sz;
    const u64 d = blkdst[b];
    if (d + sz > out_cap) {
      if (threadIdx.x == 0) atomicOr(err, 2u);
      return;
    }
    for (u32 i = threadIdx.x; i < sz; i += 256) out[d + i] = s[i];
  } else {
    const int f = b - nblocks;
    if (f < nfr && threadIdx.x < pk[f].hdr_len) {
      const u64 d = pktoff[f] + threadIdx.x;
      if (d < out_cap) out[d] = (u8)(pk[f].hdr >> (8 * threadIdx.x));
      else atomicOr(err, 2u);
    }
  }
}

// ---------------------------------------------------------------- decode ---
// first four bytes of every packet (header byte + flat colour) in one launch: one small D2H copy instead of one per frame
__global__ __launch_bounds__(256) void k_heads(const u8* __restrict__ packets, const u64* __restrict__ offs, int n, u32* __restrict__ heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 a = offs[i], len = offs[i + 1] - a;
  u32 v = 0;
  for (u64 k = 0; k < 4 && k < len; k++) v |= (u32)packets[a + k] << (8 * k);
  heads[i] = v;
}

// (the decoder proper is in scpr_wave.hpp)
// flat key frame: every pixel = the 3 bytes after the header (screencap.cpp:1537-1553)
__global__ __launch_bounds__(256) void k_fill_flat(u8* planes, Geom g, int slot, u32 rgb) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= g.H * g.S) return;
  const int y = idx / g.S, bx = idx - y * g.S;
  u8 v = bx < g.W * 3 ? (u8)(rgb >> (8 * (bx % 3))) : 0;
  planes[(size_t)slot * g.plane_stride + idx] = v;
}

}  // namespace scpr
