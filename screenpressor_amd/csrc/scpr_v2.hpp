// Version 2 streams (decode only): range coder + plain adaptive count tables.
//
// The frame layer of version 2 is that of versions 3/4 (DecompressI / DecompressP are templates
// over the entropy policy, screencap.cpp:414-498, :1275-1432); what differs is the policy, UseRC
// (screencap.h:105-265): a carry-less 32-bit range decoder (RangeCoderSub, sub.h:20-57,
// sub.cpp:43-60) and one count table per model (DecodeVal / DecodeValUni, sub.cpp:86-110,
// :144-177), no "same vector" flag for motion blocks (:263-264), motion tables sized by the
// caller's range (:235-260).  The reference only ever ENCODES version 4 (CreateCodec(4) on the
// compress side), so this is a decoder: one wave per GOP like scpr_wave.hpp, the frame walkers
// (decode_intra_frame / decode_inter_frame) are shared.
//
// A table's total and the 16 group sums of a colour table are always the sums of the counts they
// cover (they start so, every update adds the same step to both, a halving recomputes them), so
// only the counts are stored: the total falls out of the prefix scan that finds the symbol.
#pragma once
#include "scpr_wave.hpp"

namespace scpr {

struct V2Fixed {       // count tables of UseRC; rows are zero past the alphabet
  u32 n[6][256];       // ntab: run lengths by pixel type, step 400
  u32 bn[256];         // ntab2: block-type run lengths, step 20
  u32 x[256];          // xxtab: changed-block index bytes, step 1
  u32 m[2][512];       // mvtab: motion vector components, alphabet 2 * range, step 100
  u32 sxy[4][16];      // sxytab, step 100
  u32 p[6][8];         // ptypetab, step 1000
  u32 bt[8];           // bttab, step 10
};
struct __attribute__((aligned(16))) V2Lds {
  V2Fixed fx;
  u32 tile[17 * 17 + 1];  // (+ the spare cell of decode_inter_frame: WaveLds::tile)
  u32 ptile[256];
  uint2 jobs[256];
};
constexpr int V2_COLTAB = 256;          // words per colour table in HBM
constexpr u32 V2_BOT = 1u << 16;        // BOT_C, sub.h:17
constexpr u32 V2_TOP = 1u << 24;        // TOP, sub.h:14

struct WaveDecV2 {
  static constexpr bool kHelpers = false;  // one wave per GOP, no helper waves
  static constexpr bool kFastRuns = false;  // decode_intra_frame: no second instance of the run body for this coder
  struct NAsk {};                           // (the fast runs' early table read, scpr_wave.hpp: not for this coder)
  int ndec = 0;                             // (unused: the range coder has no blocks)
  const int lane;
  V2Lds& L;
  // input stream (same reader as WaveDec: wave-uniform word loads into a 64-bit shift buffer)
  const u8* src_end;
  const u32* wbase = nullptr;
  u32 wpos = 0, wmax = 0, nextw = 0;
  u64 buf = 0;
  int nb = 0;
  int left = 0;              // bytes of the current packet not yet taken (the reference throws when it runs out, sub.cpp:52-53)
  const u8* pkt_end = nullptr;
  // coder
  u32 code = 0, range = 0;
  // models
  u32* gtabs;                // [NCOLCTX][V2_COLTAB] colour counts of this GOP
  int mx2, my2;              // alphabets of the two motion tables
  bool bad = false;
  static constexpr bool oom = false;  // (no dense-table arena in the version 2 models)

  __device__ __forceinline__ WaveDecV2(V2Lds& l, const u8* e, u32* tabs, int mx2_, int my2_) : lane(lane_id()), L(l), src_end(e), gtabs(tabs), mx2(mx2_), my2(my2_) {}
  __device__ __forceinline__ void tick() {}
  template <int SEC>
  __device__ __forceinline__ void stamp() {}
  template <int EV>
  __device__ __forceinline__ void event() {}

  __device__ __forceinline__ u32 fetch_word(u32 i) {
    const u32 m = i < wmax ? i : wmax;
    return rfl(wbase[m]);
  }
  __device__ __forceinline__ u32 take_byte() {
    if (nb < 1) {
      buf |= (u64)nextw << (8 * nb);
      nb += 4;
      wpos++;
      nextw = fetch_word(wpos);
    }
    const u32 b = (u32)buf & 255u;
    buf >>= 8;
    nb--;
    if (--left < 0) bad = true;
    return b;
  }
  // UseRC::decodeBegin -> RangeCoderSub::DecodeBegin (sub.h:30-42): five bytes, the first one falls off the 32-bit code
  __device__ __forceinline__ void stream_init(const u8* s) {
    wave_fence();
    const size_t a = (size_t)rfl64((u64)(size_t)s);
    wbase = (const u32*)(a & ~(size_t)3);
    const u32 skip = (u32)(a & 3);
    const size_t e = (size_t)rfl64((u64)(size_t)src_end), ab = a & ~(size_t)3;
    wmax = e > ab ? (u32)((e - 1 - ab) >> 2) : 0u;  // the aligned word that holds the last byte of the packet buffer: nothing past it is touched (`left` guards the bytes)
    buf = (u64)(fetch_word(0) >> (8 * skip));
    nb = 4 - (int)skip;
    wpos = 1;
    nextw = fetch_word(1);
    left = (int)((size_t)rfl64((u64)(size_t)pkt_end) - a) + 1;  // the reference passes the whole packet length as what follows the header byte
    code = 0;
    range = 0xFFFFFFFFu;
    for (int i = 0; i < 5; i++) code = (code << 8) | take_byte();
  }
  // GetFreq + the symbol's interval + Decode (sub.cpp:43-60)
  __device__ __forceinline__ u32 get_freq(u32 tot) {
    range = range / tot;
    return code / range;
  }
  __device__ __forceinline__ void advance(u32, u32, u32) {}  // (WaveDec's form: named by decode_intra_frame in a branch this coder never takes)
  __device__ __forceinline__ void advance(u32 cum, u32 freq) {
    code -= cum * range;
    range *= freq;
    while (range < V2_TOP && !bad) {
      code = (code << 8) | take_byte();
      range <<= 8;
    }
  }

  // DecodeVal (sub.cpp:86-110) on a table in LDS.  PER consecutive symbols per lane; the prefix
  // scan over the lanes gives the total, the cumulative count below the symbol and the symbol.
  template <int PER>
  __device__ __forceinline__ int dec_lds(u32* cnt, int maxc, u32 step) {
    wave_fence();
    u32 c[PER], s = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int j = lane * PER + q;
      c[q] = j < maxc ? cnt[j] : 0u;
      s += c[q];
    }
    const u32 incl = (u32)wave_incl_scan((int)s);
    const u32 tot = rdl(incl, 63);
    const u32 value = get_freq(tot);
    const u64 m = __ballot(incl > value);
    int own = 63;
    if (m) own = __builtin_ctzll(m);
    else bad = true;  // value >= total: only a damaged stream
    u32 cum = incl - s;
    int k = 0;
#pragma unroll
    for (int q = 0; q < PER - 1; q++)
      if (k == q && value >= cum + c[q]) {
        cum += c[q];
        k = q + 1;
      }
    u32 fr = c[0];
#pragma unroll
    for (int q = 1; q < PER; q++) fr = k == q ? c[q] : fr;
    const int kk = (int)rdl((u32)k, own);
    int sym = own * PER + kk;
    u32 scum = rdl(cum, own), sfr = rdl(fr, own);
    if (sym >= maxc || sfr == 0) {  // damaged stream
      bad = true;
      sym = 0;
      scum = 0;
      sfr = 1;
    }
    advance(scum, sfr);
    if (tot + step > V2_BOT) {  // halve every count (the bumped one included), sub.cpp:99-107
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int j = lane * PER + q;
        if (j < maxc) cnt[j] = ((c[q] + (j == sym ? step : 0u)) >> 1) + 1u;
      }
    } else if (lane == own) {
      cnt[sym] = sfr + step;
    }
    wave_fence();
    return sym;
  }
  // DecodeValUni (sub.cpp:144-177) on a colour table in HBM (256 counts, four per lane)
  __device__ __forceinline__ int dec_global(u32* cnt, u32 step) {
    const uint4 v = ((const uint4*)cnt)[lane];
    const u32 c[4] = {v.x, v.y, v.z, v.w};
    const u32 s = v.x + v.y + v.z + v.w;
    const u32 incl = (u32)wave_incl_scan((int)s);
    const u32 tot = rdl(incl, 63);
    const u32 value = get_freq(tot);
    const u64 m = __ballot(incl > value);
    int own = 63;
    if (m) own = __builtin_ctzll(m);
    else bad = true;
    u32 cum = incl - s;
    int k = 0;
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (k == q && value >= cum + c[q]) {
        cum += c[q];
        k = q + 1;
      }
    const u32 fr = k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3];
    const int kk = (int)rdl((u32)k, own);
    const int sym = own * 4 + kk;
    u32 scum = rdl(cum, own), sfr = rdl(fr, own);
    if (sfr == 0) {
      bad = true;
      scum = 0;
      sfr = 1;
    }
    advance(scum, sfr);
    if (tot + step > V2_BOT) {
      uint4 o;
      o.x = ((c[0] + (lane * 4 + 0 == sym ? step : 0u)) >> 1) + 1u;
      o.y = ((c[1] + (lane * 4 + 1 == sym ? step : 0u)) >> 1) + 1u;
      o.z = ((c[2] + (lane * 4 + 2 == sym ? step : 0u)) >> 1) + 1u;
      o.w = ((c[3] + (lane * 4 + 3 == sym ? step : 0u)) >> 1) + 1u;
      ((uint4*)cnt)[lane] = o;
    } else if (lane == own) {
      cnt[sym] = sfr + step;
    }
    // the next symbol may use the same table: the store has to be in L2 before its load is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return sym;
  }

  // (the template parameters and the second argument only exist to match WaveDec - its symbols hand their coder step on to
  // the next one, decode_intra_frame - and are never used with this coder)
  template <bool CHK = true, bool PIPE = false, bool DOUT = false, bool PRE = false>
  __device__ __forceinline__ int fixed_n(int t, u32* = nullptr, const NAsk* = nullptr) { return dec_lds<4>(L.fx.n[t], 256, 400); }   // SC_NSTEP
  __device__ __forceinline__ void fixed_n_ask(int, NAsk&) {}
  template <bool CHK = true, bool DEFER = false>
  __device__ __forceinline__ int fixed_p(int t, u32* = nullptr) { return dec_lds<1>(L.fx.p[t], 6, 1000); }    // SC_UNSTEP
  __device__ __forceinline__ int fixed_x(int k) { return k == 0 ? dec_lds<4>(L.fx.x, 256, 1) : dec_lds<4>(L.fx.bn, 256, 20); }  // SC_XXSTEP / SC_BTNSTEP
  __device__ __forceinline__ int fixed_bt() { return dec_lds<1>(L.fx.bt, 5, 10); }            // SC_BTSTEP
  __device__ __forceinline__ int fixed_sxy(int k) { return dec_lds<1>(L.fx.sxy[k], 16, 100); }  // SC_SXYSTEP
  __device__ __forceinline__ int fixed_mv(int k) { return dec_lds<8>(L.fx.m[k], k ? my2 : mx2, 100); }  // SC_MSTEP
  __device__ __forceinline__ bool get_bool() { return false; }                                // canEncodeBool = false
  template <bool CHK = true, int MODE = 0, bool PF = false>
  __device__ __forceinline__ int colour(int ctxid, u32* = nullptr) { return dec_global(gtabs + (size_t)ctxid * V2_COLTAB, 400); }  // SC_STEP

  // RenewI with the renew* of UseRC (screencap.h:146-260): every count 1
  __device__ __forceinline__ void fixed_init() {
    wave_fence();
    u32* f = (u32*)&L.fx;
    for (int i = lane; i < (int)(sizeof(V2Fixed) / 4); i += 64) f[i] = 0;
    wave_fence();
    for (int t = 0; t < 6; t++) {
      for (int j = lane; j < 256; j += 64) L.fx.n[t][j] = 1;
      if (lane < 6) L.fx.p[t][lane] = 1;
    }
    for (int j = lane; j < 256; j += 64) {
      L.fx.bn[j] = 1;
      L.fx.x[j] = 1;
    }
    for (int j = lane; j < mx2; j += 64) L.fx.m[0][j] = 1;
    for (int j = lane; j < my2; j += 64) L.fx.m[1][j] = 1;
    if (lane < 16)
      for (int k = 0; k < 4; k++) L.fx.sxy[k][lane] = 1;
    if (lane < 5) L.fx.bt[lane] = 1;
    uint4* g = (uint4*)gtabs;
    const uint4 ones = make_uint4(1, 1, 1, 1);
    for (int i = lane; i < NCOLCTX * V2_COLTAB / 4; i += 64) g[i] = ones;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_fence();
  }
  __device__ __forceinline__ void fixed_load(const V2Fixed* __restrict__ B) {
    wave_fence();
    const u32* s = (const u32*)B;
    u32* d = (u32*)&L.fx;
    for (int i = lane; i < (int)(sizeof(V2Fixed) / 4); i += 64) d[i] = s[i];
    wave_fence();
  }
  __device__ __forceinline__ void fixed_store(V2Fixed* __restrict__ B) {
    wave_fence();
    u32* d = (u32*)B;
    const u32* s = (const u32*)&L.fx;
    for (int i = lane; i < (int)(sizeof(V2Fixed) / 4); i += 64) d[i] = s[i];
  }
};

template <bool HAS_P>
__global__ __launch_bounds__(64) void k_decode_gop_v2(const u8* __restrict__ packets, const u8* packets_end, const DecFrame* __restrict__ frames, const DecGop* __restrict__ gops,
                                                      u8* __restrict__ planes, Geom g, u32* __restrict__ tabs, u32* __restrict__ status, int ring_bytes,
                                                      V2Fixed* __restrict__ fixedstore, int far_x, int far_y) {
  __shared__ V2Lds L;
  extern __shared__ __align__(16) u8 pix[];
  const DecGop gop = gops[blockIdx.x];
  const int lane = lane_id();
  WaveDecV2 D(L, packets_end, tabs + (size_t)blockIdx.x * NCOLCTX * V2_COLTAB, 2 * far_x, 2 * far_y);
  if (gop.load) D.fixed_load(&fixedstore[blockIdx.x]);
  else D.fixed_init();
  for (int fi = gop.first; fi < gop.first + gop.count && !D.bad; fi++) {
    const DecFrame fr = frames[fi];
    u8* dst = planes + (size_t)fr.slot * g.plane_stride;
    D.pkt_end = packets + fr.src_off + fr.src_len;
    if (fr.kind == 0) {
      D.fixed_init();  // RenewI (:418)
      D.stream_init(packets + fr.src_off + 1);
      decode_intra_frame(D, g, dst, (u32*)pix, ring_bytes >> 2);
    } else if (HAS_P && fr.kind == 2) {
      decode_inter_frame(D, g, dst, planes + (size_t)fr.prev_slot * g.plane_stride, packets + fr.src_off, pix + ring_bytes, far_x, far_y);
    }
  }
  D.fixed_store(&fixedstore[blockIdx.x]);
  if (D.bad && lane == 0) atomicOr(status, 4u);
}

}  // namespace scpr
