// Wave-cooperative decoder (gfx950, one 64-lane wave per GOP) and the wave-per-chain colour
// model of the encoder.
//
// The decode of a GOP is a single dependent chain (every symbol's model depends on the bytes
// decoded before it; DecompressI, screencap.cpp:414-498; DecompressP, :1275-1432), so the wave
// does not split the symbols; it splits the WORK PER SYMBOL:
//   * the rANS state, the byte reader and all control flow are wave-uniform (scalar unit);
//   * run-length tables (FixedSizeRansCtx<256> x 6, ans_contexts.h:1054-1132) sit in LDS one
//     symbol per lane: a symbol below 64 is one compare + population count, and the table word,
//     the "below 64?" word and the running total arrive behind ONE wait; counts are bumped with
//     ds_add, rebuilds are wave prefix scans;
//   * the six pixel-type tables live in one vector register pair and are searched by one compare;
//   * colour contexts are 80-byte records cached in LDS (direct mapped, write-back to HBM);
//     a small table (SmallContext, :155-290) is one packed word per lane with incrementally
//     maintained prefix sums: lookup = one wait, one ballot, one readlane; dense tables
//     (Cx6/Cx7, :377-998) are four symbols per lane in HBM;
//   * runs are written by all lanes into an LDS ring of pixels; the gradient predictor is a wave
//     prefix sum of (top - topleft) deltas; finished rows leave as 12-byte-per-lane stores.
// What a lone wave pays per dependent instruction, and the rules that follow, are in DESIGN.md §3
// (tools/lonewave_bench.hip).
#pragma once
#include <type_traits>
#include "scpr_kernels.hpp"

namespace scpr {

// Block placement hints: the decoder is one serial chain per wave, a taken branch costs it about
// two extra instruction slots, so the common case of every test should fall through.
#define SCPR_LIKELY(c) __builtin_expect(!!(c), 1)
#define SCPR_UNLIKELY(c) __builtin_expect(!!(c), 0)

// Lanes of one wave exchange data through LDS in program order (DS operations of a
// wave execute in order), but to the compiler each lane is a thread of its own: without
// a fence it may forward a lane's own earlier store to its later load and miss what
// another lane wrote in between.  A wavefront-scope fence emits no instruction and
// forbids exactly that.
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 rfl64(u64 v) { return (u64)rfl((u32)v) | ((u64)rfl((u32)(v >> 32)) << 32); }
__device__ __forceinline__ u32 rdl(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
// ---- memory operations by SOME lanes, without a branch ------------------------------------------------------------------
// The compiler structurises a loop's control flow as soon as one lane-divergent branch sits anywhere inside it - the loop's own
// wave-uniform branches included, which then come back as flag registers and exec tests on the common path.
// "if (lane == 0) store" is such a branch.  These do the same with exec narrowed around ONE instruction and put back: the
// compiler sees a call, not control flow.  (exec may be empty for the instruction: it then does nothing.)  The global forms
// wait for their own completion - they are for the rare ways - and the LDS forms need no wait: a wave's DS operations
// execute in order.  `p` of the LDS forms is any pointer into LDS (its low 32 bits are the LDS offset).
__device__ __forceinline__ void lds_st_if(const void* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(__ballot(on)), "v"((u32)(size_t)p), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void lds_st16_if(const void* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(__ballot(on)), "v"((u32)(size_t)p), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void lds_or_if(const void* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tds_or_b32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(__ballot(on)), "v"((u32)(size_t)p), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void lds_add_if(const void* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tds_add_u32 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(__ballot(on)), "v"((u32)(size_t)p), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void glb_st_if(u32* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dword %2, %3, off\n\ts_mov_b64 exec, %0\n\ts_waitcnt vmcnt(0)" : "=&s"(keep) : "s"(__ballot(on)), "v"(p), "v"(v) : "memory", "scc");
}
// (the same without the wait: for a store whose completion nothing here depends on - a later load of the same address by this wave
// is served after it, the memory pipeline keeps a wave's accesses to one address in order)
__device__ __forceinline__ void glb_st_if_nowait(u32* p, u32 v, bool on) {
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dword %2, %3, off\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(__ballot(on)), "v"(p), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void glb_or_lane0(u32* p, u32 v) {  // agent scope, nothing returned (error flags)
  u64 keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_or %1, %2, off\n\ts_mov_b64 exec, %0\n\ts_waitcnt vmcnt(0)" : "=&s"(keep) : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32 glb_add_lane0(u32* p, u32 v) {  // returns what was there before, in every lane
  u64 keep;
  u32 old = 0;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, %3, off sc0\n\ts_mov_b64 exec, %0\n\ts_waitcnt vmcnt(0)" : "=&s"(keep), "+v"(old) : "v"(p), "v"(v) : "memory");
  return rfl(old);
}
// inclusive scan inside each row of 16 lanes: four DPP row_shr adds (no LDS traffic)
__device__ __forceinline__ int row_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  return v;
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  v = row_incl_scan(v);
  const int t0 = (int)rdl((u32)v, 15), t1 = (int)rdl((u32)v, 31), t2 = (int)rdl((u32)v, 47);
  const int row = lane_id() >> 4;
  return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}
__device__ __forceinline__ int row16_sum(int v) { return (int)rdl((u32)row_incl_scan(v), 15); }  // lanes 0..15
__device__ __forceinline__ int wave_sum(int v) {
  v = row_incl_scan(v);
  return (int)(rdl((u32)v, 15) + rdl((u32)v, 31) + rdl((u32)v, 47) + rdl((u32)v, 63));
}
__device__ __forceinline__ int shift_for(int tot) {  // number of doublings until tot > 2048 (ans_contexts.h:196-199), tot >= 2
  return __builtin_clz(min((u32)(tot - 1), 0x800u)) - 20;  // (the clamp on the argument, not on the result: stays on the scalar unit)
}

constexpr int CACHE_N = 256;  // colour records cached in LDS
constexpr u32 kNoCtx = 0xFFFFFFFFu;  // tag of an empty cache slot

struct FixedBlob {      // every fixed-alphabet model of the decoder (its image in HBM): freq | cum << 16, counts, running totals
  u32 nfc[6][256];     // run lengths, by pixel type (ntab)
  u32 ncnt[6][256];
  u32 mfc[2][512];     // motion vector components (mvtab)
  u32 mcnt[2][512];
  u32 xfc[2][256];     // [0] changed-block index bytes (xxtab), [1] block-type run lengths (ntab2)
  u32 xcnt[2][256];
  u32 sfc[4][16];      // changed-rect coordinates (sxytab)
  u32 scnt[4][16];
  u32 pfc[6][8];       // pixel types, by previous type (ptypetab)
  u32 pcnt[6][8];
  u32 bfc[8];          // block types (bttab)
  u32 bcnt[8];
  int ftot[24];        // totals: 0-5 run lengths, 6-11 pixel types, 12/13 mv, 14/15 index/length, 16-19 rect, 20 block type
};
// A colour context as the decoder keeps it (LDS cache line and HBM backing store, 80 bytes):
//   w[0] kind | maxpos << 8 | fshift << 16 | d << 20 | (kind is 4 or 5) << 31 (rewritten when it changes), w[1] total | fmax << 16 (rewritten after
//   every symbol), w[2] dense table index (written when the table is allocated), w[3] cache tag,
//   w[4..11] the 256-bit symbol set (kinds 1-3 and 6), or
//   w[4 + i] small-table entry i (kinds 4/5), see SmallTab.
struct DecRec {
  u32 w[20];
};
constexpr int DECREC_WORDS = 20;
// What of the fixed models sits in LDS while a wave decodes (the pixel-type tables live in registers).
// words per run-length table: 256 entries (freq | cum << 16), the running total at 256, padding, and the 256 counts from
// NTAB_CNT on (the lane that holds an entry reaches its count and lane 0 the total with a constant offset from one address)
constexpr int NTAB_STRIDE = 640, NTAB_CNT = 320;
struct FixedLds {       // what a key frame needs
  u32 ntab[6][NTAB_STRIDE];
  int ftot[24];        // same numbering as FixedBlob::ftot; 0-11 are unused here
};
struct FixedLdsP {      // the tables only P-frames use
  u32 mfc[2][512];
  u32 mcnt[2][512];
  u32 xfc[2][256];
  u32 xcnt[2][256];
  u32 bfc[8];
  u32 bcnt[8];
};                      // (the four 16-symbol tables of the rect coordinates live in registers: WaveDec::sxfc / sxcnt)
// (Everything from `fp` on is for P-frames: a batch of key frames - k_decode_gop_w<false> - allocates the struct up to there,
// which lets three of its workgroups share a CU at 1080p: 768 key frames per round instead of 512.)
struct __attribute__((aligned(16))) WaveLds {
  FixedLds fx;
  u32 crec[CACHE_N][DECREC_WORDS];
  u16 tmp[256];
  u32 dtag[64];        // tags of the dense-table cache (WaveModel::tab_of)
  // key frames whose picture goes to the HOST (scpr_decompress_batch_host): the chain says how many rows of the plane are complete
  // in HBM (bit 31: and nothing more will come), a second wave of the workgroup converts them to RGB32 and sends them over PCIe
  // (row_streamer) - stores to the host's memory issued by the chain itself cost it 4 % (120 ms per key frame instead of 115.6)
  // Measured (tools/exp_host_dec.py, ablation builds): the chain's own announcements cost nothing (114.8 ms per key frame with a
  // streamer that leaves at once, 114.5 without them); the streamer at work costs the chain 3 % (118.3 ms) - its stores to the
  // host's memory and the chain's loads (packet blocks, record misses: 3700 a frame) go through the CU's one vector-memory
  // pipeline, and a store that is waiting for PCIe holds a load up.  A streamer on ANOTHER compute unit would share one with
  // another chain (every CU has one) and the launch is as slow as its slowest chain: not built.
  struct {
    u32 rows;
    u32 pad[3];
  } hs;
  FixedLdsP fp;
  u32 tile[17 * 17 + 1];  // P-frame block under reconstruction, with one row above and one column to the left (+ a cell that takes the stores of idle lanes)
  u32 ptile[256];      // the same rect in the previous frame (row-major, w * h pixels): what "previous frame" runs copy (decode_inter_frame reads it through the tile's pointer)
  uint2 jobs[256];     // motion-block copies on their way to the helper waves (a ring, see hc)
  // Helper waves (P-frame GOPs: the workgroup is the chain's wave + helpers, see helper_loop): commands go out by bumping
  // `seq` after the arguments are in place, every helper adds 1 to `done` when it has finished the command it saw.
  // Motion-block copies do not go through commands: the chain appends a job to the ring `jobs` and publishes the new count
  // in `jtail`; helper h takes the jobs whose number is h modulo the number of helpers, in order, as soon as they appear, and
  // keeps the number of its next job in hprog[h] - every job below the smallest hprog is finished.
  struct {
    u32 seq, op, done, nhelp;
    u64 src, dst;
    u32 bytes, jtail;
    u32 hprog[16];
  } hc;
};
enum : u32 { HOP_COPY = 1, HOP_EXIT = 3 };

// The colour model of one context, operated by a whole wave.  The header is
// wave-uniform (scalar registers), a small table is one entry per lane (lanes
// 0..15), the 256-bit symbol set lives in words 4..11 of the context's record
// and dense tables live in the HBM arena.  Shared by the decoder (symbol from
// coder value) and the encoder chains (interval from symbol).
struct ColHdr {
  int kind, maxpos, fshift, d, total;  // total: cached sum of kinds 4-7 (kind 4: kept exact, the reference recomputes it per symbol)
  int fmax;                             // kinds 4/5: the count of entry maxpos (derived: saves reading it back for every symbol)
  int dirty;                            // decoder: kind, fshift or d changed (the record word that holds them and maxpos is rewritten)
  u32 dense;
  u32 top;                              // kinds 4/5: symbol | P << 8 of entry maxpos (small_top): what a hit on the top entry needs
};
// Small table (kinds 4/5): entry i in lane i, sorted by symbol, one packed word per lane:
//   symbol | count << 8 | P << 20,   P = sum of the counts of the entries before it.
// The entry's interval starts at symbol - i (the unmet symbols below it, one slot each) + P
// (+ the bonus when it sits above the top entry).  Keeping P up to date costs one masked add
// per symbol and saves the prefix scan.  Counts and P stay below 4096 (the total does).
// Unused lanes hold kSmallNone: count 0 and a start above every coder value.
constexpr u32 kSmallNone = 0xFFF000FFu;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int dpp_row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ u32 sm_sym(u32 w) { return w & 255u; }
__device__ __forceinline__ u32 sm_fq(u32 w) { return (w >> 8) & 4095u; }
__device__ __forceinline__ u32 sm_p(u32 w) { return w >> 20; }

// Dense tables are read and written 8 bytes per lane.  A table may sit in the arena (global memory) or in the decoder's LDS
// cache; the address space is part of the instruction (a FLAT access to LDS goes through the vector memory pipeline first and
// costs several hundred cycles), so the table code exists once per space.
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
#define SCPR_LDS_AS __attribute__((address_space(3)))
template <bool LDS>
__device__ __forceinline__ u32x2 tab_ld(const u16* arr, int lane) {
  if constexpr (LDS) return ((const SCPR_LDS_AS u32x2*)(size_t)(u32)(size_t)arr)[lane];  // LDS offset = low 32 bits of the flat address
  else return ((const u32x2*)arr)[lane];
}
template <bool LDS>
__device__ __forceinline__ void tab_st(u16* arr, int lane, u32x2 v) {
  if constexpr (LDS) ((SCPR_LDS_AS u32x2*)(size_t)(u32)(size_t)arr)[lane] = v;
  else ((u32x2*)arr)[lane] = v;
}

struct WaveModel {
  bool oom = false;  // the dense-table arena has overflowed (alloc_dense)
  const int lane;
  const int l15;  // lane & 15: a small table is kept in every row of 16 lanes alike (so that it can be stored without a lane mask)
  u16* tmp;  // 256 x u16 LDS scratch
  Arena arena;
  int f0;
  // Decoder only: an LDS cache of dense tables (direct mapped by table index, write-back).  A symbol of a dense context reads
  // 24 bytes per lane of its table and writes 8 back: from the arena that is an L2 round trip of ~2500 cycles on the one
  // chain of the GOP, and in a GOP that has been going for a while more than half of the colour symbols are such symbols.
  u8* dcache = nullptr;   // ndc slots of sizeof(DenseTab) bytes in LDS (null: tables are used where they are, in the arena)
  u32* dtag = nullptr;    // per slot: table index + 1 (0: empty)
  u32 dmask = 0;          // ndc - 1 (ndc is a power of two)
  u32 dc_lds = 0;         // the cache's LDS offset (0: none)
#ifdef SCPR_PROFILE
  u32 dmiss = 0;
  u64 dmiss_ticks = 0;
#endif
  __device__ __forceinline__ WaveModel(u16* tmp_, Arena a, int f0_) : lane(lane_id()), l15(lane_id() & 15), tmp(tmp_), arena(a), f0(f0_) {}
  template <bool DST_LDS>
  static __device__ __forceinline__ void copy_tab(DenseTab* dst, const DenseTab* src, int lane) {
    const u32x2 a = tab_ld<!DST_LDS>(src->freq, lane), b = tab_ld<!DST_LDS>(src->cum, lane), c = tab_ld<!DST_LDS>(src->cnt, lane);
    tab_st<DST_LDS>(dst->freq, lane, a);
    tab_st<DST_LDS>(dst->cum, lane, b);
    tab_st<DST_LDS>(dst->cnt, lane, c);
  }
  // where table idx is to be read and written (fresh: it is about to be written whole, its old contents do not matter)
  __device__ __forceinline__ DenseTab* tab_of(u32 idx, bool fresh = false) {
    if (dc_lds == 0u) return arena.tabs + idx;  // (the number, not the pointer: a generic pointer's null test reads the aperture register)
    wave_fence();
    const u32 slot = idx & dmask;
    const u32 tag = rfl(dtag[slot]);
    // (the slot as a bare LDS offset in a pointer's clothes: everything that takes a cached table goes through tab_ld / tab_st<true>,
    // which use the low 32 bits only - a real generic pointer into LDS drags its aperture and null tests along)
    DenseTab* c = (DenseTab*)(size_t)(dc_lds + slot * (u32)sizeof(DenseTab));
    if (SCPR_UNLIKELY(tag != idx + 1u)) {
#ifdef SCPR_PROFILE
      dmiss++;
      const u64 tm0 = __builtin_readcyclecounter();
#endif
      if (SCPR_UNLIKELY(idx > arena.cap)) {  // beyond the sink: never a table (a record that is not what it says): reported, not followed
        glb_or_lane0(arena.err, 16u);
        oom = true;
        return c;
      }
      if (tag) copy_tab<false>(arena.tabs + (tag - 1u), c, lane);
      if (!fresh) copy_tab<true>(c, arena.tabs + idx, lane);
      dtag[slot] = idx + 1u;  // (every lane the same word: no lane mask, no branch)
      wave_fence();
#ifdef SCPR_PROFILE
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      dmiss_ticks += __builtin_readcyclecounter() - tm0;
#endif
    }
    return c;
  }
  __device__ __forceinline__ void flush_tabs() {
    if (dc_lds == 0u) return;
    wave_fence();
    for (u32 slot = 0; slot <= dmask; slot++) {
      const u32 tag = rfl(dtag[slot]);
      if (tag) copy_tab<false>(arena.tabs + (tag - 1u), (const DenseTab*)(size_t)(dc_lds + slot * (u32)sizeof(DenseTab)), lane);
    }
  }
  __device__ __forceinline__ int small_fmax(const ColHdr& h, u32 w) { return (int)sm_fq(rdl(w, h.maxpos)); }
  // The header fields are wave-uniform wherever they were computed; after the rare paths (which work on the vector
  // unit) this says so to the compiler: without it the common path keeps the header in vector registers too.
  static __device__ __forceinline__ void scalar_hdr(ColHdr& h) {
    h.kind = (int)rfl((u32)h.kind), h.maxpos = (int)rfl((u32)h.maxpos), h.fshift = (int)rfl((u32)h.fshift), h.d = (int)rfl((u32)h.d);
    h.total = (int)rfl((u32)h.total), h.fmax = (int)rfl((u32)h.fmax);
    h.dirty = 1;
  }

  static __device__ __forceinline__ ColHdr unpack(u32 h0, u32 h1, u32 h2) {
    ColHdr h;
    h.kind = h0 & 255;
    h.maxpos = (h0 >> 8) & 255;
    h.fshift = (h0 >> 16) & 255;
    h.d = h1 & 0xFFFF;
    h.total = h1 >> 16;
    h.fmax = 0;  // see small_fmax()
    h.dirty = 0;
    h.dense = h2;
    h.top = 0;
    return h;
  }
  // The top entry (the one the spare code space goes to) takes most of the symbols of a small table - 85 % on desktop
  // content, and more than half of all colour symbols belong to contexts that have met ONE symbol.  Its symbol and the sum of
  // the counts before it, kept beside the header, turn such a symbol into scalar arithmetic (top_hit): no search, no lane read.
  // They change only when a symbol takes another path, and every such path ends with this refresh.
  __device__ __forceinline__ u32 small_top(const ColHdr& h, u32 w) {
    const u32 wt = rdl(w, h.maxpos);
    return sm_sym(wt) | ((sm_p(wt) + sm_sym(wt) - (u32)h.maxpos) << 8);  // symbol | start of its interval in counts (below the total: 12 bits)
  }
  // the table brought up to date with the top-entry hits that top_hit<DEC, true> only counted in the header
  __device__ __forceinline__ void small_settle(const ColHdr& h, u32& w) {
    const u32 delta = (u32)h.fmax - (h.top >> 20);
    // (masks, not a nested choice: the optimiser turns that one back into a branch on the lane number; the two conditions
    // exclude each other)
    const u32 is_top = 0u - (u32)(l15 == h.maxpos), is_above = 0u - (u32)((u32)(l15 - h.maxpos - 1) < (u32)(h.d - h.maxpos - 1));
    w += (is_top & (delta << 8)) | (is_above & (delta << 20));
  }
  // A symbol that is the top entry of a small table, no rescale due after it.  DEC: `in` is the coder value, else the symbol.
  // Returns a negative number and has applied the symbol (w, h.total, h.fmax updated; c, ofr, ocf set) - or a non-negative one
  // with nothing touched.  Same arithmetic as small_hit() for p == maxpos.
  // LAZY (the decoder): the table itself is left alone - the count of the top entry is h.fmax, and what the entries above it
  // have not been told yet is h.fmax minus the count the table holds for the top entry (kept in bits 20.. of h.top); the next
  // symbol that takes another path brings the table up to date first (small_settle).
  template <bool DEC, bool LAZY = false>
  __device__ __forceinline__ int top_hit(ColHdr& h, u32& w, int in, int& c, u32& ofr, u32& ocf) {
    const int tot = h.total, mp = h.maxpos;
    const int sh = __builtin_clz((u32)(tot - 1)) - 20, bonus = (kProbScale >> sh) - tot;  // == (kProbScale - (tot << sh)) >> sh
    const int ts = (int)(h.top & 255u);
    const int ap = (int)((h.top >> 8) & 0xFFFu), width = h.fmax + bonus;
    const int norescale = tot + 2 * kStepSmall - kProbScale - 1;  // negative: no rescale after this symbol
    int t;
    if (DEC) {
      const int vv = in >> sh;
      t = (vv - ap - width) & (ap - vv - 1) & norescale;  // each negative when fine: below the entry's end, not below its start
    } else {
      t = (ts == in ? -1 : 0) & norescale;
    }
    if (SCPR_LIKELY(t < 0)) {
      ofr = (u32)width << sh;
      ocf = (u32)ap << sh;
      if (!LAZY) w += l15 == mp ? (u32)kStepSmall << 8 : ((u32)(l15 - mp - 1) < (u32)(h.d - mp - 1) ? (u32)kStepSmall << 20 : 0u);
      h.total = tot + kStepSmall;
      h.fmax += kStepSmall;
      c = ts;
    }
    return t;
  }
  // top_hit<true, true> for a caller that branches on the test itself (the decoder's colour()): from the record's words as they
  // are read (h1 = total | count of the top entry << 16, top = small_top() | ...), nothing touched.  The interval comes first
  // (ocf = start << sh, ofr = width << sh: what the coder step wants anyway) and the test is on d = v - ocf, which is the
  // coder step's own `v - cf`: inside iff 0 <= d < ofr (the same as start <= v >> sh < start + width: both ends are
  // multiples of 1 << sh).  Negative: a hit, and no rescale is due after it.
  static __device__ __forceinline__ int top_test(u32 h1, u32 top, int v, u32& ofr, u32& ocf, u32& od) {
    const int tot = (int)(h1 & 0xFFFFu);
    const int sh = __builtin_clz((u32)(tot - 1)) - 20;
    const int width = (int)(h1 >> 16) + (kProbScale >> sh) - tot;
    ocf = ((top >> 8) & 0xFFFu) << sh;
    ofr = (u32)width << sh;
    const int d = v - (int)ocf;
    od = (u32)d;
    return (d - (int)ofr) & ~d & (tot + 2 * kStepSmall - kProbScale - 1);  // each negative when fine
  }
  static __device__ __forceinline__ u32 pack0(const ColHdr& h) { return (u32)h.kind | ((u32)h.maxpos << 8) | ((u32)h.fshift << 16); }
  static __device__ __forceinline__ u32 pack1(const ColHdr& h) { return (u32)h.d | ((u32)h.total << 16); }
  // packed table from per-lane symbols and counts (after a rebuild of the counts, or a load); returns the exact total
  __device__ __forceinline__ int small_pack(u32& w, int sym, int fq, int d) {
    fq = l15 < d ? fq : 0;
    const int incl = row_incl_scan(fq);
    w = l15 < d ? ((u32)sym | ((u32)fq << 8) | ((u32)(incl - fq) << 20)) : kSmallNone;
    return 256 - d + (int)rdl((u32)incl, 15);
  }
  // ColState image (encoder persistence): entry i <-> record bytes 16+i (symbol) and 32+2i (count)
  __device__ __forceinline__ void load_small(const u32* r, int d, u32& w) {
    const bool act = l15 < d;
    small_pack(w, act ? ((const u8*)r)[16 + l15] : 0, act ? ((const u16*)r)[16 + l15] : 0, d);
  }
  __device__ __forceinline__ void store_small(u32* r, int d, u32 w) {
    if (lane < d) {
      ((u8*)r)[16 + lane] = (u8)sm_sym(w);
      ((u16*)r)[16 + lane] = (u16)sm_fq(w);
    }
  }
  __device__ __forceinline__ void store_header(u32* r, const ColHdr& h) {
    if (lane == 0) {
      r[0] = pack0(h);
      r[1] = pack1(h);
      r[2] = h.dense;
    }
  }
  __device__ __forceinline__ u32 alloc_dense() {
    u32 idx = glb_add_lane0(arena.top, 1u);
    if (SCPR_UNLIKELY(idx >= arena.cap)) {  // (wave-uniform)
      glb_or_lane0(arena.err, 1u);
      idx = 0x80000000u;
    }
    // The arena is full: the context gets the SINK, table `cap` (allocated behind the usable ones, never a live context's
    // table).  Contexts that share it overwrite each other's contents, so whatever is coded from here on is thrown away (the
    // host runs the chunk again with the true bound, or fails the call), but it stays harmless: every record still names a
    // table that exists, a symbol looked up in the sink is kept inside 0..255 (dense_hit / dense_impl), and the chain stops at
    // its next check of `oom` (a row / a rect later at most).
    if (SCPR_UNLIKELY(idx >> 31)) {
      oom = true;
      idx = arena.cap;
    }
    return idx;
  }
  // this lane's 4 bits of the 256-bit set stored in r[4..11]
  __device__ __forceinline__ u32 set_bits4(const u32* r) { return (r[4 + (lane >> 3)] >> ((lane & 7) * 4)) & 15u; }

  // writes a dense table from per-lane (freq, count) of symbols 4*lane..+3; returns the count total
  template <bool LDS = false>
  __device__ __forceinline__ int write_dense(DenseTab* t, const int fr[4], const int cn[4]) {
    const int s = fr[0] + fr[1] + fr[2] + fr[3];
    int cf = wave_incl_scan(s) - s;
    u32 c0 = (u32)cf, c1 = c0 + fr[0], c2 = c1 + fr[1], c3 = c2 + fr[2];
    tab_st<LDS>(t->freq, lane, u32x2{(u32)(fr[0] & 0xFFFF) | ((u32)fr[1] << 16), (u32)(fr[2] & 0xFFFF) | ((u32)fr[3] << 16)});
    tab_st<LDS>(t->cum, lane, u32x2{(c0 & 0xFFFF) | (c1 << 16), (c2 & 0xFFFF) | (c3 << 16)});
    tab_st<LDS>(t->cnt, lane, u32x2{(u32)(cn[0] & 0xFFFF) | ((u32)cn[1] << 16), (u32)(cn[2] & 0xFFFF) | ((u32)cn[3] << 16)});
    return wave_sum(cn[0] + cn[1] + cn[2] + cn[3]);
  }

  // write_dense() on what tab_of() returned: a slot of the LDS cache where there is one (NOT a pointer that can be followed,
  // see tab_of), the table in the arena otherwise
  __device__ __forceinline__ int write_table(DenseTab* t, const int fr[4], const int cn[4]) {
    return dc_lds != 0u ? write_dense<true>(t, fr, cn) : write_dense<false>(t, fr, cn);
  }

  // Context::update for kinds 0-3 (ans_contexts.cpp:3-31, :52-59): c arrived raw.
  // On promotion to a small table (kind 4/5) the packed entries are returned in w.
  __device__ __forceinline__ void note_raw(u32* r, ColHdr& h, int c, u32& w) {
    wave_fence();
    if (h.kind == 0) {
      lds_st_if(&r[4 + lane], (lane == (c >> 5)) ? (1u << (c & 31)) : 0u, lane < 8);
      h.kind = 1;
      h.d = 1;
      return;
    }
    const u32 sw = rfl(r[4 + (c >> 5)]);
    if (!((sw >> (c & 31)) & 1u)) {
      r[4 + (c >> 5)] = sw | (1u << (c & 31));  // (every lane the same word)
      h.d++;
      if (h.kind == 1 && h.d == 15) h.kind = 2;
      else if (h.kind == 2 && h.d == 65) h.kind = 3;
      return;
    }
    const int d = h.d;
    const u32 bits = set_bits4(r);  // the whole set, 4 symbols per lane
    const int pc = __builtin_popcount(bits);
    if (h.kind == 1) {  // -> sorted small table, 50 each, 100 for the repeated symbol (ans_contexts.h:161-172)
      const int base = wave_incl_scan(pc) - pc;
      const int lc = c >> 2;
      h.maxpos = (int)rdl((u32)base, lc) + __builtin_popcount(rdl(bits, lc) & ((1u << (c & 3)) - 1u));
      // rank -> lane: scatter through the LDS scratch (at most 14 symbols)
#pragma unroll
      for (int q = 0; q < 4; q++) lds_st16_if(&tmp[base + __builtin_popcount(bits & ((1u << q) - 1u))], (u32)(lane * 4 + q), (bits >> q) & 1u);
      wave_fence();
      const int sym0 = (int)tmp[l15 < d ? l15 : 0];  // (read by every lane, chosen afterwards: no branch on the lane)
      const int sym = l15 < d ? sym0 : 0;
      wave_fence();
      h.kind = d <= 4 ? 4 : 5;
      h.total = small_pack(w, sym, sym == c ? 2 * kStepSmall : kStepSmall, d);
      h.fmax = 2 * kStepSmall;  // maxpos is the repeated symbol
      return;
    }
    h.dense = alloc_dense();
    r[2] = h.dense;  // (every lane the same word)  the table index only changes here: the per-symbol header store leaves word 2 alone
    DenseTab* t = tab_of(h.dense, true);
    int fr[4], cn[4];
    if (h.kind == 2) {  // Cx6::create23, ans_contexts.h:491-531
      const int tot = 256 - d + d * f0 + f0, sh = shift_for(tot), wdt = 1 << sh;
#pragma unroll
      for (int q = 0; q < 4; q++) {  // (choices, not branches, on what differs from lane to lane)
        const int j = lane * 4 + q;
        const bool met = (bits >> q) & 1u;
        const int fm = (((j == c) ? 2 * f0 : f0) << sh) & 0xFFFF;
        fr[q] = met ? fm : wdt;
        cn[q] = met ? fm - (fm >> 1) : 0;
      }
      const int sum = write_table(t, fr, cn);
      h.kind = 6;
      h.fshift = sh;
      h.total = ((256 - d) << (sh > 0 ? sh - 1 : 0)) + sum;
    } else {  // Cx7::create(Cx3&), :917-951
      const int g0 = (kProbScale - (256 - d)) / (d + 1), c0 = g0 - (g0 >> 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = lane * 4 + q;
        fr[q] = (((bits >> q) & 1u) ? g0 : 1) + (j == c ? g0 : 0);
        cn[q] = (((bits >> q) & 1u) ? c0 : 1) + (j == c ? kStepDense : 0);
      }
      h.kind = 7;
      h.total = write_table(t, fr, cn);
    }
  }

  // The decoder's common case of small_op() below with every test behind one branch: the coder value falls on an entry
  // of the table, and the table is not due for a rescale.  Returns a negative number and has applied the symbol (w,
  // h.total, h.fmax, h.maxpos updated; c, ofr, ocf set) - or a non-negative one with nothing touched: small_op() then
  // does the symbol from scratch.  (A branch costs a lone wave ~12 cycles even when it is not taken: three sign bits
  // and-ed together are cheaper than three tests.)
  __device__ __forceinline__ int small_hit(ColHdr& h, u32& w, int in, int& c, u32& ofr, u32& ocf) {
    const int d = h.d, tot = h.total;
    const int sh = __builtin_clz((u32)(tot - 1)) - 20, bonus = (kProbScale - (tot << sh)) >> sh;
    const int vv = in >> sh;
    const int st = (int)sm_sym(w) + (int)sm_p(w) + (l15 > h.maxpos ? bonus : 0) - l15;
    const u32 m = (u32)__ballot(st <= vv) & 0xFFFFu;
    const int p = 31 - (m ? __builtin_clz(m) : 32);  // -1: below the first entry
    const u32 wp = rdl(w, p);
    const int sp = (int)sm_sym(wp), fpr = (int)sm_fq(wp);
    const int ap = sp + (int)sm_p(wp) - p + (p > h.maxpos ? bonus : 0);
    const int endp = ap + fpr + (p == h.maxpos ? bonus : 0);
    // each negative when fine: the value is inside entry p, no rescale after this symbol, there is an entry p
    const int t = (vv - endp) & (tot + 2 * kStepSmall - kProbScale - 1) & ~p;
    if (SCPR_LIKELY(t < 0)) {
      ofr = (u32)(endp - ap) << sh;
      ocf = (u32)ap << sh;
      w += l15 == p ? (u32)kStepSmall << 8 : ((u32)(l15 - p - 1) < (u32)(d - p - 1) ? (u32)kStepSmall << 20 : 0u);
      h.total = tot + kStepSmall;
      if (fpr + kStepSmall > h.fmax) {
        h.maxpos = p;
        h.fmax = fpr + kStepSmall;
      }
      c = sp;
    }
    return t;
  }
  // kinds 4/5: SmallContext::decode / ::encode (ans_contexts.h:195-283), on the packed table w.
  // DEC: `in` is the coder value (state & 4095), the symbol is returned; else `in` is the symbol.
  // Both directions come down to the lane p of the last entry at or below the input (by interval
  // start when decoding, by symbol when encoding): the input is that entry, or the unmet symbol
  // (in - its end) slots above it.
  // When a full 16-table meets a 17th symbol the context becomes kind 6 (Cx6::create(Cx5&, c),
  // :454-489): the set goes to r[4..11], the table to the arena.
  template <bool DEC>
  __device__ __forceinline__ int small_op(u32* r, ColHdr& h, u32& w, int in, u32& ofr, u32& ocf) {
    int d = h.d;
    int tot = h.total;
    // shift_for(tot) without the clamp: a small table's total stays in [2, 4096], so the leading-zero count is
    // at least 20 (and the whole expression stays on the scalar unit)
    const int sh = __builtin_clz((u32)(tot - 1)) - 20, bonus = (kProbScale - (tot << sh)) >> sh;  // spare code space goes to the top entry
    const int vv = DEC ? in >> sh : 0;
    const int above = (l15 > h.maxpos ? bonus : 0) - l15;
    const int st = (int)sm_sym(w) + (int)sm_p(w) + above;  // where this lane's interval starts
    const u32 m = DEC ? (u32)__ballot(st <= vv) & 0xFFFFu : (u32)__ballot((int)sm_sym(w) <= in) & ((1u << d) - 1u);
    // (no entry at or below the input: p = -1 and what is read through it is garbage until the rare path below
    // puts it right; the common path is not asked to branch round that case)
    const int p = 31 - (m ? __builtin_clz(m) : 32);
    const u32 wp = rdl(w, p);
    int sp = (int)sm_sym(wp);
    int fpr = (int)sm_fq(wp);
    const int ap = sp + (int)sm_p(wp) - p + (p > h.maxpos ? bonus : 0);
    int endp = ap + fpr + (p == h.maxpos ? bonus : 0);
    int over;  // < 0: the input is entry p
    if (DEC) {
      over = vv - endp;
      over = p >= 0 ? over : 0;
    } else {
      over = (int)(((u32)(p >> 31) | (u32)(sp ^ in)) == 0u) * -1;  // no entry at or below the symbol, or another symbol: not a hit
    }
    if (SCPR_LIKELY(over < 0)) {
      ofr = (u32)(endp - ap) << sh;
      ocf = (u32)ap << sh;
      // the count of p, the P of p+1 .. d-1 (masks: see small_settle)
      w += ((0u - (u32)(l15 == p)) & ((u32)kStepSmall << 8)) | ((0u - (u32)((u32)(l15 - p - 1) < (u32)(d - p - 1))) & ((u32)kStepSmall << 20));
      tot += kStepSmall;
      if (fpr + kStepSmall > h.fmax) {  // p is the top entry already, or becomes it (:181)
        h.maxpos = p;
        h.fmax = fpr + kStepSmall;
      }
      if (SCPR_UNLIKELY(tot + kStepSmall > kProbScale)) {  // rescale, :186-193
        const int fq = (int)sm_fq(w);
        tot = small_pack(w, (int)sm_sym(w), fq - (fq >> 1), d);
        h.fmax -= h.fmax >> 1;
      }
      h.total = tot;
    }
    // keeps the two tests apart: plain ifs, the common case first (merged, they come back as if/else)
    if (DEC) asm volatile("" : "+s"(over));
    else over = (int)rfl((u32)over);
    if (SCPR_LIKELY(over < 0)) return sp;
    int pp = (int)sm_p(wp);
    if (p < 0) sp = -1, endp = 0, fpr = 0, pp = 0;
    const int c = DEC ? sp + 1 + vv - endp : in;
    ofr = 1u << sh;
    ocf = (u32)(DEC ? vv : c - sp - 1 + endp) << sh;
    const int pos = p + 1;
    const int cap = h.kind == 4 ? 4 : 16;
    if (d < cap || h.kind == 4) {  // addSymb (:174-184), or Cx5::create(Cx4&, c) (:350-369) when the 4-table is full
      const u32 up = (u32)dpp_row_shr1((int)w);
      {  // (selects by masks: no branch on the lane number)
        const u32 mv = 0u - (u32)(l15 > pos && l15 <= d), me = 0u - (u32)(l15 == pos);
        w = (mv & (up + ((u32)kStepSmall << 20))) | (me & ((u32)c | ((u32)kStepSmall << 8) | ((u32)(pp + fpr) << 20))) | (~(mv | me) & w);
      }
      const bool grow = d == cap;  // kind 4 -> 5: maxpos restarts at 0 (value-initialised in the reference)
      d++;
      if (grow) {
        h.maxpos = 0;
        h.fmax = (int)sm_fq(rdl(w, 0));
        h.kind = 5;
        tot += kStepSmall - 1;  // the exact total of the grown table
      } else {
        if (h.maxpos >= pos) h.maxpos++;
        if (tot + 2 * kStepSmall > kProbScale) {
          const int fq = (int)sm_fq(w);
          tot = small_pack(w, (int)sm_sym(w), fq - (fq >> 1), d);
          h.fmax -= h.fmax >> 1;
        } else {
          tot += h.kind == 4 ? kStepSmall - 1 : kStepSmall;  // kind 5 counts the step only (its total drifts, :174-184)
        }
      }
      h.d = d;
      h.total = tot;
      scalar_hdr(h);
      return c;
    }
    // kind 5 full
    const bool act = l15 < d;
    const int fqv = act ? (int)sm_fq(w) : 0;
    wave_fence();
#pragma unroll
    for (int q = 0; q < 4; q++) tmp[lane * 4 + q] = 0;
    wave_fence();
    lds_st16_if(&tmp[sm_sym(w) & 255u], (u32)fqv, act);
    wave_fence();
    const int tex = 256 - d + row16_sum(fqv), s2 = shift_for(tex), wdt = 1 << s2, base = wdt - (wdt >> 1);
    int fr[4], cn[4];
    u32 bits = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {  // (choices, not branches, on what differs from lane to lane)
      const int j = lane * 4 + q, pf = tmp[j];
      const int fm = (pf << s2) & 0xFFFF;
      fr[q] = pf ? fm : wdt;
      cn[q] = j == c ? base + (kStepHash << s2) : pf ? fm - (fm >> 1) : 0;
      bits |= (pf || j == c) ? 1u << q : 0u;
    }
    wave_fence();
    lds_st_if(&r[4 + lane], 0u, lane < 12);  // the set's eight words, and words 12 .. 15: the decoder keeps a dense table's widest symbol in 12 / 13 (WaveDec::colour)
    wave_fence();
    lds_or_if(&r[4 + (lane >> 3)], bits << ((lane & 7) * 4), true);
    wave_fence();
    h.dense = alloc_dense();
    r[2] = h.dense;  // (every lane the same word)  the table index only changes here: the per-symbol header store leaves word 2 alone
    const int sum = write_table(tab_of(h.dense, true), fr, cn);
    h.kind = 6;
    h.fshift = s2;
    h.fmax = 0;  // (a dense context's header has no top-entry count there: the decoder counts the widest symbol's unsettled hits in it)
    h.d = d + 1;
    h.total = ((256 - (d + 1)) << (s2 > 0 ? s2 - 1 : 0)) + sum;
    wave_fence();
    w = r[4 + l15];  // the decoder stores w over the entries after every symbol: make that harmless for the new set
    scalar_hdr(h);
    return c;
  }

  // Encoder chains: one context per chain, so a dense context's table (24 bytes per lane) stays in registers for the chain - a
  // symbol that has been met, with no rescale due after it, is two lane reads and a masked add; anything else goes through
  // dense_op() on the table in the arena (the caller stores the registers first and loads them again afterwards).  A chain of
  // 13000 symbols through a dense context was 13000 L2 round trips (the longest chain of a frame: it set the stage's time).
  __device__ __forceinline__ int dense_enc_hit(ColHdr& h, int c, u32x2 fq, u32x2 cu, u32x2& cq, u32 bits, u32& ofr, u32& ocf) {
    const int own = c >> 2, kk = c & 3;
    const u32 fw = rdl(kk < 2 ? fq.x : fq.y, own), cw = rdl(kk < 2 ? cu.x : cu.y, own);
    const u32 sh16 = (u32)(kk & 1) * 16u;
    const int step = h.kind == 7 ? kStepDense : kStepHash << h.fshift;
    const u32 met = (rdl(bits, own) >> kk) & 1u;  // (kind 7: every symbol counts as met)
    const int tst = (met ? -1 : 0) & (h.total + 2 * step - kProbScale - 1);  // each negative when fine
    if (SCPR_LIKELY(tst < 0)) {
      ofr = (fw >> sh16) & 0xFFFFu;
      ocf = (cw >> sh16) & 0xFFFFu;
      const u32 inc = (u32)step << sh16;
      const bool mine = lane == own;
      cq.x += (mine && kk < 2) ? inc : 0u;
      cq.y += (mine && kk >= 2) ? inc : 0u;
      h.total += step;
    }
    return tst;
  }
  // kinds 6/7: Cx6/Cx7 decode and encode (ans_contexts.h:640-740, :953-997), 4 symbols per lane
  template <bool DEC>
  __device__ __forceinline__ int dense_op(u32* r, ColHdr& h, int in, u32& ofr, u32& ocf) {
    wave_fence();
    DenseTab* t = tab_of(h.dense);
    if (dc_lds != 0u) return dense_impl<DEC, true>(t, r, h, in, ofr, ocf);
    return dense_impl<DEC, false>(t, r, h, in, ofr, ocf);
  }
  template <bool DEC, bool LDS>
  __device__ __forceinline__ int dense_impl(DenseTab* t, u32* r, ColHdr& h, int in, u32& ofr, u32& ocf) {
    const u32x2 cu = tab_ld<LDS>(t->cum, lane), fq = tab_ld<LDS>(t->freq, lane), cq = tab_ld<LDS>(t->cnt, lane);
    const u32 v = (u32)in;
    const u32 c0 = cu.x & 0xFFFF, c1 = cu.x >> 16, c2 = cu.y & 0xFFFF, c3 = cu.y >> 16;
    const u64 m = DEC ? __ballot(c0 <= v) : 0;
    const int own = DEC ? 63 - __builtin_clzll(m) : in >> 2;
    const int k = DEC ? (c1 <= v) + (c2 <= v) + (c3 <= v) : (in & 3);
    // (a word by bit 1 of k, a half by bit 0: a chain of choices comes back as branches on the lane)
    const u32 selc = (((k & 2) ? cu.y : cu.x) >> (16 * (k & 1))) & 0xFFFFu;
    const u32 self = (((k & 2) ? fq.y : fq.x) >> (16 * (k & 1))) & 0xFFFFu;
    const int kk = (int)rdl((u32)k, own);
    const int j = own * 4 + kk;
    ofr = rdl(self, own);
    ocf = rdl(selc, own);
    int cn[4] = {(int)(cq.x & 0xFFFF), (int)(cq.x >> 16), (int)(cq.y & 0xFFFF), (int)(cq.y >> 16)};
    int fr[4] = {(int)(fq.x & 0xFFFF), (int)(fq.x >> 16), (int)(fq.y & 0xFFFF), (int)(fq.y >> 16)};
    if constexpr (DEC) {
      // The decoder's shortcut for the table's widest symbol (WaveDec::colour) counts its hits in the header (h.fmax) and leaves
      // the table alone: they are owed to that symbol's count before anything here looks at the counts, and the symbol is
      // forgotten - this is the way on which intervals change (and a symbol that is not met yet changes nothing, but comes by
      // rarely: the next wide symbol that is looked up the long way is remembered again).
      const u32 t13 = rfl(r[13]);
      if (t13 >> 31) {
        const int tj = (int)(t13 & 255u), owed = h.fmax * (h.kind == 6 ? kStepHash << h.fshift : kStepDense);
#pragma unroll
        for (int q = 0; q < 4; q++) cn[q] += (lane == (tj >> 2) && q == (tj & 3)) ? owed : 0;
        h.fmax = 0;
        r[13] = 0u;  // (every lane the same word)
      }
    }
    const u32 bits = h.kind == 6 ? set_bits4(r) : 15u;
    u32 nbits = bits;
    int step = kStepDense;
    if (h.kind == 6) {
      step = kStepHash << h.fshift;
      const bool present = (rdl(bits, own) >> kk) & 1u;
      if (SCPR_UNLIKELY(!present)) {
        if (h.d >= kHashMaxSyms) {  // 41st symbol: becomes kind 7, uncounted (:631, Cx7::create(const Cx6&) :868-915)
          const int wdt = 1 << h.fshift, base = wdt - (wdt >> 1);
#pragma unroll
          for (int q = 0; q < 4; q++) cn[q] = ((bits >> q) & 1u) ? cn[q] : base;
          tab_st<LDS>(t->cnt, lane, u32x2{(u32)cn[0] | ((u32)cn[1] << 16), (u32)cn[2] | ((u32)cn[3] << 16)});
          h.kind = 7;
          return DEC ? (j & 255) : j;
        }
        {  // placeSymbol, :621-638 (the owning lane's doing, by selects and a one-lane LDS operation: no branch on the lane)
          const bool mine = lane == own;
          nbits |= mine ? 1u << kk : 0u;
#pragma unroll
          for (int q = 0; q < 4; q++) cn[q] = (mine && q == kk) ? (int)ofr - ((int)ofr >> 1) : cn[q];
          lds_or_if(&r[4 + (lane >> 3)], (1u << kk) << ((lane & 7) * 4), mine);
        }
        h.d++;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) cn[q] += (lane == own && q == kk) ? step : 0;
    h.total += step;
    if (SCPR_UNLIKELY(h.total + step > kProbScale)) {
      if (h.kind == 7) {  // Cx7::incrCnt rebuild, :963-980
#pragma unroll
        for (int q = 0; q < 4; q++) {
          fr[q] = cn[q];
          cn[q] -= cn[q] >> 1;
        }
        h.total = write_dense<LDS>(t, fr, cn);
      } else {  // Cx6::rescale, :742-796
        const int wdt = 1 << (h.fshift > 0 ? h.fshift - 1 : 0);
        if (h.fshift > 0) h.fshift--;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool met = (nbits >> q) & 1u;
          fr[q] = met ? cn[q] : wdt;
          cn[q] = met ? cn[q] - (cn[q] >> 1) : 0;
        }
        const int sum = write_dense<LDS>(t, fr, cn);
        h.total = ((256 - h.d) << (h.fshift > 0 ? h.fshift - 1 : 0)) + sum;
      }
    } else {  // (the counts go back from all lanes alike - only the owner's have changed: no branch on the lane)
      tab_st<LDS>(t->cnt, lane, u32x2{(u32)cn[0] | ((u32)cn[1] << 16), (u32)cn[2] | ((u32)cn[3] << 16)});
    }
    return DEC ? (j & 255) : j;  // (a table that other contexts have scribbled on after an arena overflow may match no lane: own = -1)
  }
  // The rescale of dense_impl() by itself, on a table whose counts are up to date (encoder chains: k_colour_chain_w counts a
  // whole epoch's symbols with LDS atomics and then calls this): Cx7::incrCnt's rebuild (:963-980) / Cx6::rescale (:742-796).
  template <bool LDS>
  __device__ __forceinline__ void dense_rescale(DenseTab* t, u32* r, ColHdr& h) {
    wave_fence();
    const u32x2 cq = tab_ld<LDS>(t->cnt, lane);
    int cn[4] = {(int)(cq.x & 0xFFFF), (int)(cq.x >> 16), (int)(cq.y & 0xFFFF), (int)(cq.y >> 16)};
    int fr[4];
    if (h.kind == 7) {
      for (int q = 0; q < 4; q++) {
        fr[q] = cn[q];
        cn[q] -= cn[q] >> 1;
      }
      h.total = write_dense<LDS>(t, fr, cn);
    } else {
      const u32 bits = set_bits4(r);
      const int wdt = 1 << (h.fshift > 0 ? h.fshift - 1 : 0);
      if (h.fshift > 0) h.fshift--;
      for (int q = 0; q < 4; q++) {
        if ((bits >> q) & 1u) {
          fr[q] = cn[q];
          cn[q] -= cn[q] >> 1;
        } else {
          fr[q] = wdt;
          cn[q] = 0;
        }
      }
      const int sum = write_dense<LDS>(t, fr, cn);
      h.total = ((256 - h.d) << (h.fshift > 0 ? h.fshift - 1 : 0)) + sum;
    }
    wave_fence();
  }
};

#ifdef SCPR_PROFILE
__device__ u64 g_prof[24];
__device__ u64 g_cprof[32];  // colour symbols by class: [2k] s_memtime ticks, [2k + 1] count - see WaveDec::cls_end
// design aid (tools/profile_chains.py): one record per colour chain of 2048 symbols or more - length, cycles, final kind | d << 8,
// symbols that took the general small-table path, symbols coded by the epoch-parallel dense path, raw symbols
__device__ u32 g_chainrec[8192][8];
__device__ u32 g_chainrec_n;
#endif
#ifdef SCPR_NO_DTOP
constexpr bool kDenseTop = false;  // (A/B builds)
#else
constexpr bool kDenseTop = true;
#endif
struct WaveDec : WaveModel {
  WaveLds& L;
  // input stream
  const u8* src;
  const u8* src_end;
  // 4-byte aligned base of the current packet - as a GLOBAL pointer: through a generic one the block load is a FLAT load, which
  // counts on the LDS counter as well, and wherever the compiler then copies the block's register (every join of a loop that may
  // refill it: the rolled colour loop of a P-frame's literal) it waits for "vmcnt(0) lgkmcnt(0)" - i.e. for the ds_add of the
  // symbol before, ~70 cycles, twice per colour symbol of a P-frame (round 4; the ISA listing showed it, no counter did)
  typedef const __attribute__((address_space(1))) u32* GlobalWords;
  GlobalWords wbase = nullptr;
  u32 wpos = 0, wmax = 0;       // next word to take, the word that holds the last byte of the packet buffer
  u32 tailmask = 0;             // the bytes of word wmax that lie at or past the end of the buffer
  u32 blk = 0;                  // lane i: word i of the current 64-word block
  u64 buf = 0;
  int nb = 0;
  // coder
  u32 x = 0;
  int ndec = 0;
  // pixel-type tables (ptypetab) in registers: table t in lanes 8t..8t+5 (entries) and 8t+7 (the running total, in pcnt)
  u32 pfc = 0xFFFFFFFFu, pcnt = 0;
  // the changed-rect coordinate tables (sxytab) likewise: table t in lanes 16t..16t+15 (their totals are qtot lanes 4..7)
  u32 sxfc = 0xFFFFFFFFu, sxcnt = 0;
  // the P-frame tables (fixed_any; lane = table - 12: motion x / y, block index bytes, block-type run lengths, rect x1 y1 x2 y2,
  // block type): the widest symbol of each as start | (width - 1) << 12 and its number, and the table's running total.  Motion
  // vectors repeat from block to block: the widest symbol takes almost every lookup, with one compare and nothing read.
  u32 qtopA = 0, qtopB = 0, qtot = 0;
  u32 crec_ea = 0;  // LDS offset of the record cache + 4 * (lane & 15): a slot's `ea` is this + 80 * slot
  // what a hit on a small table's top entry adds to the record's second word (total | top count << 16), in lane 0; nothing elsewhere
  u32 ktop = 0;
  u32 kl0 = 0;   // all ones in lane 0, nothing in the others (a value for lane 0's address alone, by one `and`)
  // models
  DecRec* gstates;
  bool bad = false;
  bool has_p = true;  // false: the workgroup's LDS ends before the P-frame tables (WaveLds::fp on)

  __device__ __forceinline__ WaveDec(WaveLds& l, const u8* s, const u8* e, DecRec* gs, Arena a, int f0_) : WaveModel(l.tmp, a, f0_), L(l), src(s), src_end(e), gstates(gs) {
    ktop = lane_id() == 0 ? (u32)kStepSmall * 0x10001u : 0u;
    kl0 = lane_id() == 0 ? 0xFFFFFFFFu : 0u;
    crec_ea = (u32)(size_t)&l.crec[0][0] + 4u * (u32)(lane_id() & 15);
    asm volatile("" : "+v"(ktop), "+v"(crec_ea));  // (kept in registers: not made afresh for every symbol)
  }

  // ---------------------------------------------------------------- input ---
  // The packet is read 256 bytes at a time: one load gives every lane one 4-byte word of the block and the
  // coder takes them out with a lane read: one memory wait per 256 bytes instead of one per word.  (Asking
  // for the next block ahead of use was tried: the register in flight gets copied at every join of the
  // control flow, and each copy waits for the load.)  Word indices are clamped to the aligned word that holds
  // the last byte of the packet buffer: no address at or past the end of that word is ever touched (an aligned
  // dword never crosses a page), and the bytes of it that lie past the buffer's end are replaced by 0xFF.
  __device__ __forceinline__ u32 load_block(u32 b) {
    const u32 i = b * 64u + (u32)lane;
    return __builtin_nontemporal_load(&wbase[i < wmax ? i : wmax]);
  }
  __device__ __forceinline__ void stream_init(const u8* s) {  // decodeBegin, screencap.h:295-301
    wave_fence();
    const size_t a = (size_t)rfl64((u64)(size_t)s);  // wave-uniform: keeps the whole stream state in scalar registers
    wbase = (GlobalWords)(a & ~(size_t)3);
    const u32 skip = (u32)(a & 3);
    const size_t e = (size_t)rfl64((u64)(size_t)src_end), ab = a & ~(size_t)3;
    wmax = e > ab ? (u32)((e - 1 - ab) >> 2) : 0u;
    const u32 valid = (u32)((e - 1) & 3) + 1u;  // bytes of word wmax inside the buffer
    tailmask = valid == 4u ? 0u : 0xFFFFFFFFu << (8u * valid);
    blk = load_block(0);
    u32 w0 = rdl(blk, 0);
    if (SCPR_UNLIKELY(wmax == 0)) w0 |= tailmask;
    buf = (u64)(w0 >> (8 * skip));
    nb = 4 - (int)skip;
    wpos = 1;
    ndec = 0;
    x = take_u32();
  }
  __device__ __forceinline__ void tick() {}
  // section timing for design work (only with -DSCPR_PROFILE): time since the previous stamp goes to section `sec`
#ifdef SCPR_PROFILE
  u64 prof[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0;
  template <int EV>
  __device__ __forceinline__ void event() { prof[EV]++; }  // 8 colour symbols, 9 record cache misses, 10 dense, 11 raw, 12 small-table slow path, 13 runs, 14 literal runs, 15 run lengths above 63
  template <int SEC>
  __device__ __forceinline__ void stamp() {
    const u64 t = __builtin_readcyclecounter();
    if (prof_last) prof[SEC] += t - prof_last;
    prof_last = t;
  }
  // colour symbols by class (0 top entry of a small table, 1 another entry, 2 small table's general path, 3 dense hit, 4 dense
  // general path, 5 raw, 6 record-cache miss [the miss alone, also inside its symbol's class], 7 dense-table cache miss [likewise])
  u64 cprof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cls_t0 = 0;
  u64 xprof[4] = {0, 0, 0, 0};  // P-frames, dense tables: symbols, answered from the record (the widest symbol), widest symbols learnt, forgotten
  u64 rprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // P-frame runs through the general fills: all, literal, left, above, previous frame, above-left / gradient, longer than 64, their pixels
  __device__ __forceinline__ void cls_begin() { cls_t0 = __builtin_readcyclecounter(); }
  template <int K>
  __device__ __forceinline__ void cls_end() {
    cprof[2 * K] += __builtin_readcyclecounter() - cls_t0;
    cprof[2 * K + 1]++;
  }
  // (a class that is only known at run time: a wave-uniform flag, both ways written out - no indexed register array)
  template <int K0, int K1>
  __device__ __forceinline__ void cls_end2(int second) {
    const u64 dt = __builtin_readcyclecounter() - cls_t0;
    if (rfl((u32)second)) cprof[2 * K1] += dt, cprof[2 * K1 + 1]++;
    else cprof[2 * K0] += dt, cprof[2 * K0 + 1]++;
  }
  template <int K>
  __device__ __forceinline__ void cls_add(u64 t) {
    cprof[2 * K] += t;
    cprof[2 * K + 1]++;
  }
#else
  template <int SEC>
  __device__ __forceinline__ void stamp() {}
  template <int EV>
  __device__ __forceinline__ void event() {}
  __device__ __forceinline__ void cls_begin() {}
  template <int K>
  __device__ __forceinline__ void cls_end() {}
  template <int K0, int K1>
  __device__ __forceinline__ void cls_end2(int) {}
#endif
  __device__ __forceinline__ void need(int k) {
    while (SCPR_UNLIKELY(nb < k)) {
      u32 w = rdl(blk, (int)(wpos & 63u));
      // Past the end of the packet buffer the reader supplies 0xFF bytes: a damaged stream can run off the end,
      // and the refill loop of the coder (advance) must still terminate (on zero bytes it would not).
      if (SCPR_UNLIKELY(wpos >= wmax)) w = wpos > wmax ? 0xFFFFFFFFu : (w | tailmask);
      buf |= (u64)w << (8 * nb);
      nb += 4;
      wpos++;
      if (SCPR_UNLIKELY((wpos & 63u) == 0)) blk = load_block(wpos >> 6);
    }
  }
  __device__ __forceinline__ u32 take_byte() {
    need(1);
    u32 b = (u32)buf & 255u;
    buf >>= 8;
    nb--;
    return b;
  }
  __device__ __forceinline__ u32 take_u32() {  // RansDecInit, rans_byte.h:106-119
    need(4);
    u32 v = (u32)buf;
    buf >>= 32;
    nb -= 4;
    return v;
  }
  __device__ __forceinline__ void advance(u32 cf, u32 fr, u32 v) {  // RansDecAdvance, rans_byte.h:130-146
    u32 hi = x >> kProbBits;
    asm("s_mul_i32 %0, %1, %2" : "=s"(hi) : "s"(hi), "s"(fr));  // the state is wave-uniform: keep it on the scalar unit
    x = hi + (v - cf);
    while (SCPR_UNLIKELY(x < kRansL)) x = (x << 8) | take_byte();  // ends on any input: past the packets the reader supplies 0xFF bytes
  }
  // CHK = false: the caller has made sure the block does not end within this run and adds the run's symbols to ndec itself
  // (decode_intra_frame's fast runs): a test-and-branch per symbol is ~20 cycles of a lone wave's time.
  static constexpr bool kFastRuns = true;
  static constexpr bool kHelpers = true;  // decode_inter_frame: the workgroup may have helper waves (WaveLds::hc)
  template <bool CHK = true>
  __device__ __forceinline__ void count() {  // screencap.h:327-331
    if (!CHK) return;
    if (SCPR_UNLIKELY(++ndec == kBlockEntries)) {
      x = take_u32();
      ndec = 0;
    }
  }

  // ---------------------------------------------------------------- fixed ---
  // renew() of every table (RenewI, screencap.cpp:178-198)
  __device__ __forceinline__ void fixed_init() {
    FixedLds& F = L.fx;
    FixedLdsP& FP = L.fp;
    auto fill = [&](u32* fc, u32* cnt, int nsym, int cap, int ti) __attribute__((always_inline)) {
      const int fr = kProbScale / nsym, c0 = fr - (fr >> 1);
      for (int j = lane; j < cap; j += 64) {
        fc[j] = j < nsym ? ((u32)fr | ((u32)(fr * j) << 16)) : 0xFFFFFFFFu;
        cnt[j] = j < nsym ? (u32)c0 : 0u;
      }
      if (lane == 0) F.ftot[ti] = c0 * nsym;
    };
    wave_fence();
    for (int t = 0; t < 6; t++) {
      for (int j = lane; j < NTAB_CNT; j += 64) F.ntab[t][j] = j < 256 ? (16u | ((u32)(16 * j) << 16)) : j == 256 ? 8u * 256u : 0u;
      for (int j = lane; j < 256; j += 64) F.ntab[t][NTAB_CNT + j] = 8u;
    }
    {
      const int j = lane & 7, fr = kProbScale / 6, c0 = fr - (fr >> 1);
      pfc = (lane < 48 && j < 6) ? ((u32)fr | ((u32)(fr * j) << 16)) : 0xFFFFFFFFu;
      pcnt = lane < 48 ? (j < 6 ? (u32)c0 : j == 7 ? (u32)(c0 * 6) : 0u) : 0u;
    }
    if (has_p) {
      for (int t = 0; t < 2; t++) {
        fill(FP.mfc[t], FP.mcnt[t], 512, 512, 12 + t);
        fill(FP.xfc[t], FP.xcnt[t], 256, 256, 14 + t);
      }
      sxfc = 256u | ((u32)(256 * (lane & 15)) << 16);
      sxcnt = 128u;
      if (lane >= 16 && lane < 20) F.ftot[lane] = 128 * 16;
      fill(FP.bfc, FP.bcnt, 5, 8, 20);
      // every symbol of a renewed table is equally wide: its widest symbol is symbol 0
      const int nsyms = lane < 2 ? 512 : lane < 4 ? 256 : lane < 8 ? 16 : 5, fr = kProbScale / nsyms;
      qtopA = lane < 9 ? (u32)(fr - 1) << 12 : 0u;
      qtopB = 0;
      qtot = lane < 9 ? (u32)((fr - (fr >> 1)) * nsyms) : 0u;
    }
    wave_fence();
  }
  // qtopA / qtopB / qtot from the tables in LDS and their totals (after fixed_load)
  __device__ __forceinline__ void fixed_tops(const int* ftot) {
    FixedLdsP& FP = L.fp;
    u32 a, b;
    qtopA = qtopB = 0;
    for (int t = 0; t < 2; t++) {
      top_of_table<8>(FP.mfc[t], 512, a, b);
      if (lane == t) qtopA = a, qtopB = b;
      top_of_table<4>(FP.xfc[t], 256, a, b);
      if (lane == 2 + t) qtopA = a, qtopB = b;
    }
    top_of_table<1>(FP.bfc, 5, a, b);
    if (lane == 8) qtopA = a, qtopB = b;
    qtot = lane < 9 ? (u32)ftot[12 + lane] : 0u;
  }
  // the renewed P-frame tables written straight into a models' image in HBM: a batch of key frames has no room for them in
  // LDS (WaveLds), but the P-frames of a later call continue from this image
  __device__ __forceinline__ void fixed_store_renewed_p(FixedBlob* __restrict__ B) {
    auto fill = [&](u32* fc, u32* cnt, int nsym, int cap, int ti) __attribute__((always_inline)) {
      const int fr = kProbScale / nsym, c0 = fr - (fr >> 1);
      for (int j = lane; j < cap; j += 64) {
        fc[j] = j < nsym ? ((u32)fr | ((u32)(fr * j) << 16)) : 0xFFFFFFFFu;
        cnt[j] = j < nsym ? (u32)c0 : 0u;
      }
      if (lane == 0) B->ftot[ti] = c0 * nsym;
    };
    for (int t = 0; t < 2; t++) {
      fill(B->mfc[t], B->mcnt[t], 512, 512, 12 + t);
      fill(B->xfc[t], B->xcnt[t], 256, 256, 14 + t);
    }
    for (int t = 0; t < 4; t++) fill(B->sfc[t], B->scnt[t], 16, 16, 16 + t);
    fill(B->bfc, B->bcnt, 5, 8, 20);
  }
  // the models of an earlier call, from their image in HBM
  __device__ __forceinline__ void fixed_load(const FixedBlob* __restrict__ B) {
    FixedLds& F = L.fx;
    FixedLdsP& FP = L.fp;
    wave_fence();
    for (int t = 0; t < 6; t++) {
      for (int j = lane; j < NTAB_CNT; j += 64) F.ntab[t][j] = j < 256 ? B->nfc[t][j] : j == 256 ? (u32)B->ftot[t] : 0u;
      for (int j = lane; j < 256; j += 64) F.ntab[t][NTAB_CNT + j] = B->ncnt[t][j];
    }
    {
      const int t = lane >> 3, j = lane & 7;
      pfc = lane < 48 ? B->pfc[t][j] : 0xFFFFFFFFu;
      pcnt = lane < 48 ? (j == 7 ? (u32)B->ftot[6 + t] : B->pcnt[t][j]) : 0u;
    }
    if (has_p) {
      for (int i = lane; i < 1024; i += 64) {
        (&FP.mfc[0][0])[i] = (&B->mfc[0][0])[i];
        (&FP.mcnt[0][0])[i] = (&B->mcnt[0][0])[i];
      }
      for (int i = lane; i < 512; i += 64) {
        (&FP.xfc[0][0])[i] = (&B->xfc[0][0])[i];
        (&FP.xcnt[0][0])[i] = (&B->xcnt[0][0])[i];
      }
      sxfc = (&B->sfc[0][0])[lane];
      sxcnt = (&B->scnt[0][0])[lane];
      if (lane < 8) {
        FP.bfc[lane] = B->bfc[lane];
        FP.bcnt[lane] = B->bcnt[lane];
      }
    }
    if (lane < 24) F.ftot[lane] = B->ftot[lane];
    wave_fence();
    if (has_p) fixed_tops(B->ftot);
    wave_fence();
  }
  __device__ __forceinline__ void fixed_store(FixedBlob* __restrict__ B) {
    FixedLds& F = L.fx;
    FixedLdsP& FP = L.fp;
    wave_fence();
    for (int t = 0; t < 6; t++) {
      for (int j = lane; j < 256; j += 64) {
        B->nfc[t][j] = F.ntab[t][j];
        B->ncnt[t][j] = F.ntab[t][NTAB_CNT + j];
      }
      if (lane == 0) B->ftot[t] = (int)F.ntab[t][256];
    }
    if (has_p) {
      for (int i = lane; i < 1024; i += 64) {
        (&B->mfc[0][0])[i] = (&FP.mfc[0][0])[i];
        (&B->mcnt[0][0])[i] = (&FP.mcnt[0][0])[i];
      }
      for (int i = lane; i < 512; i += 64) {
        (&B->xfc[0][0])[i] = (&FP.xfc[0][0])[i];
        (&B->xcnt[0][0])[i] = (&FP.xcnt[0][0])[i];
      }
      (&B->sfc[0][0])[lane] = sxfc;
      (&B->scnt[0][0])[lane] = sxcnt;
      if (lane < 8) {
        B->bfc[lane] = FP.bfc[lane];
        B->bcnt[lane] = FP.bcnt[lane];
      }
    }
    if (lane < 48) {
      const int t = lane >> 3, j = lane & 7;
      B->pfc[t][j] = pfc;
      B->pcnt[t][j] = j == 7 ? 0u : pcnt;
      if (j == 7) B->ftot[6 + t] = (int)pcnt;
    }
    if (has_p) {
      if (lane < 9) B->ftot[12 + lane] = (int)qtot;
    } else {
      fixed_store_renewed_p(B);  // (a key frame renews them, screencap.cpp:178-198, and a batch of key frames never uses them)
    }
  }

  // Run length after a pixel of type t (decode + incrCnt, ans_contexts.h:1093-1112, :1070-1091).
  // Symbol j of a table is LDS word j: the first 64 symbols (almost every run) are one per lane and
  // are found with one compare and a population count; word 64 tells whether that is enough, word
  // 256 is the running total.  The three words come back from one wait.
  // PIPE: the coder step of the symbol before (pend: advance + count) is taken while the table's words are on their way
  // from LDS (see record<CHK, PIPE>).
  // DOUT: its own coder step is left in `pend` for the caller (the key-frame loop takes it under the run's ring read).
  // PRE: the table's three words were asked for by fixed_n_ask() as soon as the type was known - in front of a literal's three
  // colour symbols, under which their LDS round trip then goes by (a run without a literal comes here at once: as before).
  struct NAsk {
    u32 e0, e1, et;
  };
  __device__ __forceinline__ void fixed_n_ask(int t, NAsk& q) {
    wave_fence();
    const u32 addr = (u32)(size_t)L.fx.ntab[t] + 4u * (u32)lane;
    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:1024" : "=v"(q.e0), "=v"(q.e1), "=v"(q.et) : "v"(addr) : "memory");
  }
  template <bool CHK = true, bool PIPE = false, bool DOUT = false, bool PRE = false>
  __device__ __forceinline__ int fixed_n(int t, u32* pend = nullptr, const NAsk* pre = nullptr) {
    wave_fence();
    u32* tab = L.fx.ntab[t];
    const u32 addr = (u32)(size_t)tab + 4u * (u32)lane;  // LDS offset = low 32 bits of the flat address
    u32 e0, e1, et;
    if constexpr (PRE) {
      e0 = pre->e0, e1 = pre->e1, et = pre->et;
      advance(pend[0], pend[1], pend[2]);
      count<CHK>();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(et) : : "memory");
    } else if constexpr (PIPE) {
      asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:1024" : "=v"(e0), "=v"(e1), "=v"(et) : "v"(addr) : "memory");
      advance(pend[0], pend[1], pend[2]);
      count<CHK>();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1), "+v"(et) : : "memory");
    } else {
      asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(e0), "=v"(e1), "=v"(et) : "v"(addr) : "memory");
    }
    const u32 v = x & (kProbScale - 1), lim = (v + 1) << 16;  // cum <= v  <=>  (freq | cum << 16) < lim
    const int tot0 = (int)rfl(et);
    // (plain ifs only: an else on this path costs the common case a taken branch)
    int sym = __builtin_popcountll(__ballot(e0 < lim)) - 1;
    u32 s = rdl(e0, sym);
    if (SCPR_UNLIKELY(rfl(e1) < lim)) {  // cum of symbol 64 is not above v: the symbol is further up
      event<15>();
      const u32 e2 = tab[128 + lane], e3 = tab[192 + lane];
      sym = 63 + __builtin_popcountll(__ballot(e1 < lim)) + __builtin_popcountll(__ballot(e2 < lim)) + __builtin_popcountll(__ballot(e3 < lim));
      const int q = sym >> 6, l = sym & 63;
      s = q == 1 ? rdl(e1, l) : q == 2 ? rdl(e2, l) : rdl(e3, l);
      lds_add_if(&tab[NTAB_CNT + sym], (u32)kStepDense, lane == 0);
    }
    {
      // the count of the symbol (from the lane that holds it: none for a symbol above 63, counted above) and the
      // total (lane 0); the other lanes add 0 to padding
      const u32 dc = lane == sym ? (u32)kStepDense : 0u, dt = lane == 0 ? (u32)kStepDense : 0u;
      asm volatile("ds_add_u32 %0, %1 offset:%3\n\tds_add_u32 %0, %2 offset:1024" ::"v"(addr), "v"(dc), "v"(dt), "n"(4 * NTAB_CNT) : "memory");
    }
    if constexpr (DOUT) pend[0] = 0u, pend[1] = s & 0xFFFF, pend[2] = v - (s >> 16);  // (handed on as the difference: every producer the same shape)
    else advance(s >> 16, s & 0xFFFF, v);
    if (SCPR_UNLIKELY(tot0 + 2 * kStepDense > kProbScale)) {  // rebuild from the counts, ans_contexts.h:1075-1090
      wave_fence();
      u32* cnt = tab + NTAB_CNT;
      int base = 0, ns = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = (int)cnt[lane + 64 * q];
        const int inc = wave_incl_scan(c);
        tab[lane + 64 * q] = (u32)c | ((u32)(base + inc - c) << 16);
        base += (int)rdl((u32)inc, 63);
        const int h = c - (c >> 1);
        cnt[lane + 64 * q] = (u32)h;
        ns += h;
      }
      ns = wave_sum(ns);
      tab[256] = (u32)ns;  // (every lane the same word)
      wave_fence();
    }
    if constexpr (!DOUT) count<CHK>();
    return sym;
  }
  // Pixel type after a pixel of type t: all six tables are searched by the same compare
  // DEFER: the coder step (advance + count) is left to the next symbol, which takes it while its own table is on its way
  template <bool CHK = true, bool DEFER = false>
  __device__ __forceinline__ int fixed_p(int t, u32* pend = nullptr) {
    const u32 v = x & (kProbScale - 1), lim = (v + 1) << 16;
    const u64 m = __ballot(pfc < lim);
    const u32 mt = (u32)(m >> (8 * t)) & 0xFFu;  // never 0: the first cum is 0
    const int j = 31 - __builtin_clz(mt);
    const int own = 8 * t + j, tl = 8 * t + 7;
    const u32 s = rdl(pfc, own);
    const int tot = (int)rdl(pcnt, tl) + kStepDense;
    pcnt += (lane == own || lane == tl) ? (u32)kStepDense : 0u;
    if constexpr (DEFER) pend[0] = 0u, pend[1] = s & 0xFFFF, pend[2] = v - (s >> 16);
    else advance(s >> 16, s & 0xFFFF, v);
    if (SCPR_UNLIKELY(tot + kStepDense > kProbScale)) {
      const bool in = (lane >> 3) == t && (lane & 7) < 6;
      const int c = in ? (int)pcnt : 0;
      const int inc = wave_incl_scan(c);
      const int h = c - (c >> 1);
      const int nt = wave_sum(h);
      pfc = in ? (u32)c | ((u32)(inc - c) << 16) : pfc;
      pcnt = in ? (u32)h : pcnt;
      pcnt = lane == tl ? (u32)nt : pcnt;
    }
    if constexpr (!DEFER) count<CHK>();
    return j;
  }

  // the widest symbol of a table in LDS (PER entries per lane): { start | (width - 1) << 12, symbol }
  template <int PER>
  __device__ __forceinline__ void top_of_table(const u32* fc, int nsym, u32& ta, u32& tb) {
    wave_fence();
    u32 best = 0;  // width << 16 | 0xFFFF - symbol: the widest, the lowest of equals
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int j = lane * PER + q;
      const u32 e = j < nsym ? fc[j] : 0u;
      const u32 key = j < nsym ? ((e & 0xFFFFu) << 16) | (0xFFFFu - (u32)j) : 0u;
      best = max(best, key);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) best = max(best, (u32)__shfl_xor((int)best, d));
    best = rfl(best);
    const int sym = (int)(0xFFFFu - (best & 0xFFFFu));
    const u32 e = rfl(fc[sym]);
    ta = (e >> 16) | (((e & 0xFFFFu) - 1u) << 12);
    tb = (u32)sym;
  }
  template <int PER>
  __device__ __forceinline__ int fixed_rebuild(u32* fc, u32* cnt, int nsym) {  // incrCnt rebuild, ans_contexts.h:1075-1090
    wave_fence();
    int c[PER], s = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      int j = lane * PER + q;
      c[q] = j < nsym ? (int)cnt[j] : 0;
      s += c[q];
    }
    int cf = wave_incl_scan(s) - s, ns = 0;
    wave_fence();
#pragma unroll
    for (int q = 0; q < PER; q++) {
      int j = lane * PER + q;
      if (j < nsym) {
        fc[j] = (u32)c[q] | ((u32)cf << 16);
        cf += c[q];
        int h = c[q] - (c[q] >> 1);
        cnt[j] = (u32)h;
        ns += h;
      }
    }
    wave_fence();
    return wave_sum(ns);
  }
  // The P-frame tables (searched in LDS): symbol whose interval holds the coder value, then the
  // table update.  PER entries per lane; entries past the alphabet hold ~0.  First the table's widest symbol (qtopA / qtopB).
  template <int PER>
  __device__ __forceinline__ int fixed_any(u32* fc, u32* cnt, int nsym, int ti) {
    wave_fence();
    const u32 v = x & (kProbScale - 1);
    const int tl = ti - 12;
    const u32 ta = rdl(qtopA, tl);
    const int tot0 = (int)rdl(qtot, tl);
    u32 cf = ta & 0xFFFu, fr = (ta >> 12) + 1u;
    int sym = (int)rdl(qtopB, tl);
    int miss = (v - cf) < fr ? 0 : -1;  // (unsigned: below the start wraps)
    if (SCPR_UNLIKELY(miss < 0)) {
      const u32 lim = (v + 1) << 16;
      u32 e[PER];
#pragma unroll
      for (int q = 0; q < PER; q++) e[q] = (PER > 1 || lane < (nsym <= 8 ? 8 : 16)) ? fc[lane * PER + q] : 0xFFFFFFFFu;
      const u64 m = __ballot(e[0] < lim);
      const int own = 63 - __builtin_clzll(m);
      int k = 0;
#pragma unroll
      for (int q = 1; q < PER; q++) k += e[q] < lim;
      u32 sel = e[0];
#pragma unroll
      for (int q = 1; q < PER; q++) sel = (k == q) ? e[q] : sel;
      const int kk = PER > 1 ? (int)rdl((u32)k, own) : 0;
      const u32 sv = rdl(sel, own);
      sym = own * PER + kk;
      cf = sv >> 16;
      fr = sv & 0xFFFFu;
    }
    {
      // no lane mask: lane 0 adds the step to the symbol's count, the others add 0 to words of the scratch area
      const u32 addr = lane == 0 ? (u32)(size_t)&cnt[sym] : (u32)(size_t)tmp + 4u * (u32)lane;
      const u32 dv = lane == 0 ? (u32)kStepDense : 0u;
      asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(dv) : "memory");
    }
    int tot = tot0 + kStepDense;
    advance(cf, fr, v);
    if (SCPR_UNLIKELY(tot + kStepDense > kProbScale)) {
      tot = fixed_rebuild<PER>(fc, cnt, nsym);
      u32 na, nb;
      top_of_table<PER>(fc, nsym, na, nb);
      if (lane == tl) qtopA = na, qtopB = nb;
    }
    if (lane == tl) qtot = (u32)tot;
    count();
    return sym;
  }
  __device__ __forceinline__ int fixed_mv(int t) { return fixed_any<8>(L.fp.mfc[t], L.fp.mcnt[t], 512, 12 + t); }
  __device__ __forceinline__ int fixed_x(int t) { return fixed_any<4>(L.fp.xfc[t], L.fp.xcnt[t], 256, 14 + t); }
  // a coordinate of a changed rect: the tables are in registers, searched like the pixel types (fixed_p)
  __device__ __forceinline__ int fixed_sxy(int t) {
    const u32 v = x & (kProbScale - 1), lim = (v + 1) << 16;
    const u64 m = __ballot(sxfc < lim);
    const u32 mt = (u32)(m >> (16 * t)) & 0xFFFFu;  // never 0: the first cum is 0
    const int j = 31 - __builtin_clz(mt);
    const int own = 16 * t + j, tl = 4 + t;
    const u32 s = rdl(sxfc, own);
    int tot = (int)rdl(qtot, tl) + kStepDense;
    sxcnt += lane == own ? (u32)kStepDense : 0u;
    advance(s >> 16, s & 0xFFFF, v);
    if (SCPR_UNLIKELY(tot + kStepDense > kProbScale)) {  // incrCnt rebuild, ans_contexts.h:1075-1090
      const bool in = (lane >> 4) == t;
      const int c = in ? (int)sxcnt : 0;
      const int inc = wave_incl_scan(c);
      const int h = c - (c >> 1);
      tot = wave_sum(h);
      if (in) {
        sxfc = (u32)c | ((u32)(inc - c) << 16);
        sxcnt = (u32)h;
      }
    }
    if (lane == tl) qtot = (u32)tot;
    count();
    return j;
  }
  __device__ __forceinline__ int fixed_bt() { return fixed_any<1>(L.fp.bfc, L.fp.bcnt, 5, 20); }
  __device__ __forceinline__ bool get_bool() {  // decodeBool, screencap.h:411-421
    const u32 v = x & (kProbScale - 1);
    const bool flag = v >= kProbScale / 2;
    advance(flag ? kProbScale / 2 : 0, kProbScale / 2, v);
    count();
    return flag;
  }

  // --------------------------------------------------------------- colour ---
  // The record of a context in the LDS cache: header + tag (one broadcast read) and the small
  // table (one word per lane, lanes 16.. mirror lanes 0..15) come back from one wait.
  // PIPE: the coder step of the symbol before (pcf, pfr, pv: advance + count) runs between asking for the record and using
  // it - the step needs nothing of the record and the record nothing of the step, and the LDS round trip (68+ cycles in which a
  // lone wave issues nothing) is as long as the step (colour<CHK, MODE>).
  static __device__ __forceinline__ int slot_of(int ctxid) { return (ctxid ^ (ctxid >> 7)) & (CACHE_N - 1); }
  // Two address registers: `ea` = the slot + 4 * (lane & 15) - this lane's table entry is at ea + 16, and lane 0's ea is the
  // record itself, which is where colour()'s ds_add goes - and the slot's own address in every lane for the header's four
  // words (ONE address for all lanes: a 16-byte read per lane at 4-byte steps is 132 cycles instead of 73, tools/lds_bench.hip).
  template <bool CHK, bool PIPE>
  __device__ __forceinline__ void record(int ctxid, u32& w, u32& ea, u32& h0, u32& h1, u32& hz, u32 pcf = 0, u32 pfr = 0, u32 pv = 0) {
    wave_fence();
    u32x4 hw;
    const u32 sa = (u32)(4 * DECREC_WORDS) * (u32)slot_of(ctxid);
    const u32 ha = (u32)(size_t)&L.crec[0][0] + sa;
    ea = crec_ea + sa;
    if constexpr (PIPE) {
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %2 offset:16" : "=v"(hw), "=v"(w) : "v"(ea), "v"(ha) : "memory");
      advance(pcf, pfr, pv);
      count<CHK>();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hw), "+v"(w) : : "memory");  // (ties what was read to the wait: nothing that uses it moves above)
    } else {
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(hw), "=v"(w) : "v"(ea), "v"(ha) : "memory");
    }
    // all four words go to the scalar unit BEFORE the tag is tested (the empty asm wants them): a lane read takes ~30 cycles to
    // reach a scalar consumer, and behind the test the header's reads would start that wait a second time
    const u32 tag = rfl(hw.w);  // the context the slot holds (kNoCtx: none)
    h0 = rfl(hw.x);
    h1 = rfl(hw.y);
    u32 h2 = rfl(hw.z);
    asm volatile("" : "+s"(h0), "+s"(h1), "+s"(h2));
    if (SCPR_UNLIKELY(tag != (u32)ctxid)) {
      event<9>();
#ifdef SCPR_PROFILE
      const u64 tm0 = __builtin_readcyclecounter();
#endif
      u32* r = L.crec[slot_of(ctxid)];
      {  // (the record's words by its first lanes, without a branch on the lane: masked stores, a load every lane can make)
        // The new record is ASKED FOR first and the old one sent off behind it, unwaited: one trip to L2 on the chain instead of
        // two (the store used to be waited for before the load was issued: 961 ticks per miss, 609 misses per P-frame).
        const int wl = min(lane, DECREC_WORDS - 1);
        const bool in = lane < DECREC_WORDS;
        const u32 old = r[wl];
        const u32 nw = gstates[ctxid].w[wl];
        glb_st_if_nowait(&gstates[tag != kNoCtx ? tag : (u32)ctxid].w[wl], old, in && tag != kNoCtx);
        lds_st_if(&r[wl], lane == 3 ? (u32)ctxid : nw, in);
      }
      wave_fence();
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(hw), "=v"(w) : "v"(ea), "v"(ha) : "memory");
      h0 = rfl(hw.x);
      h1 = rfl(hw.y);
      h2 = rfl(hw.z);
#ifdef SCPR_PROFILE
      cls_add<6>(__builtin_readcyclecounter() - tm0);
#endif
    }
    hz = h2;  // (wave-uniform already)
  }
  // the header's fields from the record's first two words (fshift and the dense index: colour() fills them in for the kinds that have them)
  static __device__ __forceinline__ void hdr_of(ColHdr& h, u32 h0, u32 h1) {
    h.kind = h0 & 255;
    h.maxpos = (h0 >> 8) & 255;
    h.fshift = 0;
    h.d = (h0 >> 20) & 0x7FF;
    h.total = h1 & 0xFFFF;
    h.fmax = h1 >> 16;
    h.dirty = 0;
    h.dense = 0;
  }
  static __device__ __forceinline__ u32 dec_pack0(const ColHdr& h) {
    return (u32)h.kind | ((u32)h.maxpos << 8) | ((u32)h.fshift << 16) | ((u32)h.d << 20) | ((h.kind | 1) == 5 ? 0x80000000u : 0u);
  }
  __device__ __forceinline__ void flush_records() {
    wave_fence();
    for (int slot = 0; slot < CACHE_N; slot++) {
      const u32 tag = rfl(L.crec[slot][3]);
      if (tag != kNoCtx && lane < DECREC_WORDS) gstates[tag].w[lane] = L.crec[slot][lane];
    }
  }
  // The common case of dense_op<true>() in straight-line form, like small_hit(): the table is in the LDS cache (or comes into
  // it), the coder value falls on a symbol the context has met (kind 7: every symbol), and no rescale is due after it.
  // Returns a negative number and has applied the symbol (c, ofr, ocf set; the symbol's count and h.total bumped) - or a
  // non-negative one with nothing touched: dense_op() then does the symbol from scratch.  The lane that owns the interval is
  // found with one ballot over each lane's first start; the other three starts of that lane, its four widths and the
  // met-symbol set (lanes 0..7 of w, read with the record) are taken out with lane reads, so everything after the table read
  // is scalar; the counts go back from all lanes alike (no lane mask).
  __device__ __forceinline__ int dense_hit(ColHdr& h, u32 w, int v, int& c, u32& ofr, u32& ocf) {
    if (dc_lds == 0u) return 0;
    DenseTab* t = tab_of(h.dense);
    const u32 ta = (u32)(size_t)t + 8u * (u32)lane;  // LDS offset of this lane's four symbols (freq at +0, cum at +512, cnt at +1024)
    u32x2 fq, cu, cq;
    asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:512\n\tds_read_b64 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(fq), "=v"(cu), "=v"(cq) : "v"(ta) : "memory");
    const u64 m = __ballot((cu.x & 0xFFFFu) <= (u32)v);  // never empty: symbol 0 starts at 0
    const int own = 63 - __builtin_clzll(m);
    const u32 sc0 = rdl(cu.x, own), sc1 = rdl(cu.y, own), sf0 = rdl(fq.x, own), sf1 = rdl(fq.y, own);
    const u32 c1 = sc0 >> 16, c2 = sc1 & 0xFFFFu, c3 = sc1 >> 16;
    const int kk = (int)(c1 <= (u32)v) + (int)(c2 <= (u32)v) + (int)(c3 <= (u32)v);  // the starts inside a lane go up strictly (no symbol is empty)
    const u32 cw = kk < 2 ? sc0 : sc1, fw = kk < 2 ? sf0 : sf1;
    const u32 sh16 = (u32)(kk & 1) * 16u;
    const int j = own * 4 + kk;
    const int is7 = h.kind == 7;
    const int step = is7 ? kStepDense : kStepHash << h.fshift;
    const u32 met = is7 ? 1u : (rdl(w, j >> 5) >> (j & 31)) & 1u;
    // each negative when fine: the symbol has been met, no rescale after it
    const int tst = ((int)met - 1 >= 0 ? -1 : 0) & (h.total + 2 * step - kProbScale - 1);
    if (SCPR_LIKELY(tst < 0)) {
      ocf = (cw >> sh16) & 0xFFFFu;
      ofr = (fw >> sh16) & 0xFFFFu;
      c = j & 255;  // (see dense_impl)
      const u32 inc = (u32)step << sh16;
      const bool mine = lane == own;
      cq.x += (mine && kk < 2) ? inc : 0u;
      cq.y += (mine && kk >= 2) ? inc : 0u;
      asm volatile("ds_write_b64 %0, %1 offset:1024" ::"v"(ta), "v"(cq) : "memory");
      h.total += step;
      event<16>();
    }
    return tst;
  }
  // The same for the symbols of P-frames, where two in five colour symbols come this way (key frames: one in forty): scalar
  // tests throughout - the slot as an LDS offset (a generic pointer is tested for null when it is narrowed), the lane's other
  // three starts compared by sign bits, the met-symbol bit read for both kinds (no branch around a lane read).  Eight
  // instructions fewer of ~88, 2 % of a P-frame.  The key-frame loop keeps the form above: with this one in it the same
  // kernel is 0.3-0.6 % slower on key frames (code layout; measured twice), and the headline is key frames.
  __device__ __forceinline__ int dense_hit_p(ColHdr& h, u32 w, int v, int& c, u32& ofr, u32& ocf) {
    const u32 dc = dc_lds;
    if (SCPR_UNLIKELY(dc == 0u)) return 0;
    wave_fence();
    const u32 slot = h.dense & dmask;
    u32 tb = dc + slot * (u32)sizeof(DenseTab);
    u32 ta = tb + 8u * (u32)lane;  // this lane's four symbols (freq at +0, cum at +512, cnt at +1024)
    u32x2 fq, cu, cq;
    // the slot's tag and the table's words behind ONE wait (95 % of the lookups find their table in its slot: the words are then
    // the right ones; otherwise the table is brought in and read again)
    u32 tgv;
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %5 offset:512\n\tds_read_b64 %3, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(tgv), "=v"(fq), "=v"(cu), "=v"(cq)
                 : "v"((u32)(size_t)&dtag[slot]), "v"(ta)
                 : "memory");
    if (SCPR_UNLIKELY(rfl(tgv) != h.dense + 1u)) {  // not there: brought in (or refused: oom), same slot
      tb = (u32)(size_t)tab_of(h.dense);
      ta = tb + 8u * (u32)lane;
      asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:512\n\tds_read_b64 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(fq), "=v"(cu), "=v"(cq) : "v"(ta) : "memory");
    }
    const u64 m = __ballot((cu.x & 0xFFFFu) <= (u32)v);  // never empty: symbol 0 starts at 0
    const int own = 63 - __builtin_clzll(m);
    const u32 sc0 = rdl(cu.x, own), sc1 = rdl(cu.y, own), sf0 = rdl(fq.x, own), sf1 = rdl(fq.y, own);
    const u32 c1 = sc0 >> 16, c2 = sc1 & 0xFFFFu, c3 = sc1 >> 16;
    // how many of the lane's other three starts are not above v: sign bits (the starts are 16-bit numbers and go up strictly)
    const u32 uv = (u32)v;
    const int kk = (int)(((c1 - uv - 1u) >> 31) + ((c2 - uv - 1u) >> 31) + ((c3 - uv - 1u) >> 31));
    const int hi = kk >> 1;  // the symbol sits in the lane's second word
    const u32 cw = hi ? sc1 : sc0, fw = hi ? sf1 : sf0;
    const u32 sh16 = (u32)(kk & 1) * 16u;
    const int j = own * 4 + kk;
    const int is7 = h.kind & 1;  // (the kind is 6 or 7 here)
    const int step = is7 ? kStepDense : kStepHash << h.fshift;
    // (kind 7 has met every symbol; the bit of kind 6 is read for both - lanes 0..7 of w hold words in either kind's record)
    const u32 met = (u32)is7 | ((rdl(w, j >> 5) >> (j & 31)) & 1u);
    // each negative when fine: the symbol has been met, no rescale after it
    const int tst = (0 - (int)met) & (h.total + 2 * step - kProbScale - 1);
    if (SCPR_LIKELY(tst < 0)) {
      ocf = (cw >> sh16) & 0xFFFFu;
      ofr = (fw >> sh16) & 0xFFFFu;
      c = j & 255;  // (see dense_impl)
      const u32 inc = (u32)step << sh16;
      const bool mine = lane == own;
      cq.x += (mine && !hi) ? inc : 0u;
      cq.y += (mine && hi) ? inc : 0u;
      asm volatile("ds_write_b64 %0, %1 offset:1024" ::"v"(ta), "v"(cq) : "memory");
      h.total += step;
      event<16>();
    }
    return tst;
  }
  // decodeC (screencap.h:318-333).  MODE 0: the whole symbol.  MODE 1 / 2 (the three colour symbols of a key frame's literal):
  // the symbol's coder step is NOT taken here but handed to the caller in (pcf, pfr, pv), who passes it to the next symbol -
  // MODE 2 takes the step it is handed while its own record is on its way from LDS (record<CHK, true>) - and takes the last
  // one itself (advance + count).
  // PF: called from a P-frame's runs (dense_hit_p)
  template <bool CHK = true, int MODE = 0, bool PF = false>
  __device__ __forceinline__ int colour(int ctxid, u32* pend = nullptr) {
    ColHdr h;
    u32 w, ra = 0, ea, h0, h1, hz;
    u32* r = nullptr;
    cls_begin();
    if constexpr (MODE == 2) record<CHK, true>(ctxid, w, ea, h0, h1, hz, pend[0], pend[1], pend[2]);
    else record<CHK, false>(ctxid, w, ea, h0, h1, hz);
    event<8>();
    // The three ways of a symbol - the top entry of a small table, anything else of a small table, the other kinds - each
    // END with the tail (coder step, header words) written out for it.  All conditions are wave-uniform scalars, nothing below
    // holds a branch on the lane number, and the build leaves uniform regions unstructurised (build.py), so these are plain
    // scalar branches with the unlikely ways out of line.
    int c = 0;
    u32 fr, cf;
    const u32 v = x & (kProbScale - 1);
    int maxpos0 = 0;
    // the header's fields, for the two ways that want them: from copies of the words the optimiser cannot see through, or it
    // computes every field in front of the first branch, on the top-entry hit's path too
    auto header = [&]() __attribute__((always_inline)) {
      u32 q0 = h0, q1 = h1;
      int qc = ctxid;
      asm volatile("" : "+s"(q0), "+s"(q1), "+s"(qc));
      hdr_of(h, q0, q1);
      maxpos0 = h.maxpos;
      r = L.crec[slot_of(qc)];  // (the record's address as a pointer and as an LDS offset: only these ways want them)
      ra = (u32)(size_t)r;
    };
    auto tail = [&]() __attribute__((always_inline)) {
      if constexpr (MODE == 0) advance(cf, fr, v);
      else pend[0] = 0u, pend[1] = fr, pend[2] = v - cf;
      wave_fence();
      // the header: the word that changes with every symbol from all lanes alike (same address, same value), the other one when it changes
      const u32 n1 = (u32)h.total | ((u32)h.fmax << 16);
      asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(ra), "v"(n1) : "memory");
      if (SCPR_UNLIKELY(h.dirty | (h.maxpos ^ maxpos0))) {
        const u32 n0 = dec_pack0(h);
        asm volatile("ds_write_b32 %0, %1" ::"v"(ra), "v"(n0) : "memory");
      }
      wave_fence();
      if constexpr (MODE == 0) count<CHK>();
    };
    // The hit that needs no table - the top entry of a small table, and in P-frames the widest symbol of a dense one - is ONE
    // block for both (two copies of it were two more joins at which the compiler copies the packet block's register, behind a
    // wait for every load and store in flight: the dense copy measured no gain at all that way, 10.42 against 10.40 ms per P-frame).
    const bool small = (int)h0 < 0;  // sign bit: a small table (kind 4 or 5)
    int tt = 0;
    u32 dd = 0, hitc = 0, hitadd = 0;
    if (SCPR_LIKELY(small)) {
      tt = top_test(h1, hz, (int)v, fr, cf, dd);
      hitc = hz;
      hitadd = ktop;
    } else if constexpr (PF && kDenseTop) {
      // A dense table's widest symbol (round 5).  The intervals of a dense table are FROZEN between its rebuilds - a symbol only
      // adds to its count (Cx7::incrCnt / Cx6, ans_contexts.h:954-981, :742-796) - and the contexts that go dense in a P-frame are
      // not noise: 68-72 % of their symbols are the table's widest one, 2750 of 4096 wide on average (tools/r5/prof_dense_top.py).
      // The record remembers that symbol (words 12 / 13: start | width << 16, number | 1 << 31; learnt below from the first
      // lookup that meets a symbol at least half the scale wide - then the widest -, forgotten on the way that rebuilds:
      // dense_impl), and a value inside its interval needs no table: no second LDS round trip, no search.  Its hits are
      // counted in the header's spare half word (where a small table keeps its top entry's count) and owed to the table's
      // count until the long way comes by, with one LDS add like the top entry of a small table.
      const u32 t13 = rdl(w, 9), t12 = rdl(w, 8);
      const u32 tcf = t12 & 0xFFFFu;
      const u32 stp = (h0 & 1u) ? (u32)kStepDense : (u32)kStepHash << ((h0 >> 16) & 15u);
      fr = t12 >> 16;
      dd = v - tcf;
      // (sign bits and-ed together: kind 6 or 7 - bits 1 and 2 of the kind, no other kind on this side has both -, the symbol is
      // known, the value is not below its interval and not at or above its end, no rebuild is due after it)
      tt = (int)(((h0 << 29) & (h0 << 30)) & t13 & ~dd & (dd - fr) & ((h1 & 0xFFFFu) + 2u * stp - (u32)kProbScale - 1u));
      hitc = t13;
      hitadd = kl0 & (stp + 0x10000u);  // lane 0: the total and the owed hits, one add; the other lanes' words get nothing
#ifdef SCPR_PROFILE
      xprof[0] += (h0 & 0xFEu) == 6u;
      xprof[1] += tt < 0;
#endif
    }
    if (SCPR_LIKELY(tt < 0)) {
      // A hit on the top entry changes two numbers, both in the header's second word: the total and the top entry's count
      // go up by the same step (the table itself hears of it later: small_settle).  One LDS add, nothing unpacked, nothing
      // packed: lane 0's address is the record (its ea), the other lanes add nothing to words further on.
      c = (int)(hitc & 255u);
#ifdef SCPR_PROFILE
      if (small) event<17>();
      else event<10>();
#endif
      if constexpr (MODE == 0) advance(0u, fr, dd);  // (dd = v - cf: the test has it already)
      else pend[0] = 0u, pend[1] = fr, pend[2] = dd;
      wave_fence();
      // (an add by all lanes, four to a word, in front of the next read costs 5 cycles more than a plain write - and
      // 9 less than switching exec for lane 0 alone, tools/lds_bench.hip)
      asm volatile("ds_add_u32 %0, %1 offset:4" ::"v"(ea), "v"(hitadd) : "memory");
      wave_fence();
      if constexpr (MODE == 0) count<CHK>();
#ifdef SCPR_PROFILE
      if (small) cls_end<0>();
      else cls_end<3>();
#else
      cls_end<0>();
#endif
    } else if (small) {
      {  // another entry, an unmet symbol, or a rescale is due
        header();
        h.top = hz;
        small_settle(h, w);
        const int t = small_hit(h, w, (int)v, c, fr, cf);
        if (SCPR_UNLIKELY(t >= 0)) {
          event<12>();
#ifdef SCPR_PROFILE
          const u64 tm0 = __builtin_readcyclecounter();
#endif
          c = small_op<true>(r, h, w, (int)v, fr, cf);
#ifdef SCPR_PROFILE
          cls_add<2>(__builtin_readcyclecounter() - tm0);
#endif
        }
        if ((h.kind | 1) == 5) {  // still a small table: its top entry may have moved, or the counts before it have changed
          const u32 nt = small_top(h, w) | ((u32)h.fmax << 20);
          asm volatile("ds_write_b32 %0, %1 offset:8" ::"v"(ra), "v"(nt) : "memory");
        }
        wave_fence();
        // the entries go back from every row of 16 lanes alike: no lane mask (a full table has just become a dense
        // one: then w holds what is there already)
        asm volatile("ds_write_b32 %0, %1 offset:16" ::"v"(ea), "v"(w) : "memory");
        wave_fence();
        tail();
        cls_end<1>();
      }
    } else {
      header();

      h.fshift = (int)((h0 >> 16) & 15u);
      h.dense = hz;
#ifdef SCPR_PROFILE
      if (h.kind >= 6 && h.dense >= arena.cap && !prof[18]) {  // design aid: the first record that names a table outside the arena
        prof[18] = 0x100000000ull | (u32)ctxid;
        prof[21] = h0;
        prof[22] = h.dense;
        prof[23] = ((u64)(oom ? 1u : 0u) << 32) | (u32)ndec;
      }
#endif
      if (h.kind < 4) {  // a raw symbol leaves the coder alone, which is an advance over the whole range
        event<11>();
        fr = kProbScale, cf = 0;
        c = (int)take_byte();
        note_raw(r, h, c, w);
        wave_fence();
        if (h.kind == 4 || h.kind == 5) {  // promoted to a small table
          r[4 + l15] = w;  // (every row of 16 lanes holds the table alike: no lane mask)
          const u32 nt = small_top(h, w) | ((u32)h.fmax << 20);
          r[2] = nt;  // (every lane the same word)
        }
        wave_fence();
        scalar_hdr(h);
        cls_end<5>();
      } else {
        event<10>();
        const int t = PF ? dense_hit_p(h, w, (int)v, c, fr, cf) : dense_hit(h, w, (int)v, c, fr, cf);
        if constexpr (PF && kDenseTop) {
          if (SCPR_UNLIKELY((t < 0) & (fr >= (u32)kProbScale / 2u))) {  // at least half the scale: this table's widest symbol until its next rebuild
#ifdef SCPR_PROFILE
            xprof[2]++;
#endif
            u32x2 nt;
            nt.x = cf | (fr << 16);
            nt.y = (u32)c | 0x80000000u;
            asm volatile("ds_write_b64 %0, %1 offset:48" ::"v"(ra), "v"(nt) : "memory");  // words 12 / 13, from all lanes alike
          }
        }
        if (SCPR_UNLIKELY(t >= 0)) {
#ifdef SCPR_PROFILE
          const u64 tm0 = __builtin_readcyclecounter();
#endif
          c = dense_op<true>(r, h, (int)v, fr, cf);
          scalar_hdr(h);
#ifdef SCPR_PROFILE
          cls_add<4>(__builtin_readcyclecounter() - tm0);
#endif
        }
        cls_end<3>();
      }
      tail();
      
    }
    return c;
  }
};

// ---------------------------------------------------------------------------------------
// One wave per GOP: a key frame (or a flat frame that renews the models) and the P-frames
// that depend on it, decoded in order by the same wave so that the models, the coder and
// the previous plane never leave the CU between frames.
struct DecFrame {
  u64 src_off;   // packet offset in the packet buffer
  u32 src_len;
  int slot;      // destination plane
  int kind;      // 0 coded key frame, 1 flat key frame (plane filled by k_fill_flat), 2 P-frame
  int prev_slot; // P-frames: plane of the previous frame
};
struct DecGop {
  int first, count;  // frames[first .. first+count)
  int load;          // 1: the models continue from fixedstore[gop] / states (state of an earlier call)
  int pad;
};

// three bytes at p, through L2 (bypasses this CU's vector L1, which may hold lines from before
// the wave's own stores to the same plane)
__device__ __forceinline__ u32 lds_peek_w(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ u32 ld3_l2(const u8* p) {
  const size_t a = (size_t)p;
  const u32* w = (const u32*)(a & ~(size_t)3);
  const u32 sh = (u32)(a & 3) * 8;
  const u32 lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32 v = lo >> sh;
  if (sh > 8) v |= __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << (32 - sh);
  return v & 0xFFFFFFu;
}

// Key frame (DecompressI, screencap.cpp:414-498).  Decoded pixels go to an LDS ring of 32-bit
// pixels indexed by raster position (it always holds the last two rows: the predictors read
// "previous", "top" and "top-left" from it); every finished row is packed to RGB24 and flushed
// to HBM four pixels per lane.  The plane in HBM is never read back.
// rows_word / taken_word (may be null): two LDS words between the chain and the workgroup's row streamer (below).  The chain
// announces in rows_word how many rows are complete IN THE RING; the streamer takes them from there, writes them to the caller's
// picture as RGB32 while the chain decodes (a key frame is 110 ms of chain and 8 MB of pixels) and says in taken_word how many rows
// it has taken.  The ring holds the last two rows and a little more, so the chain may not run further ahead than that: once per row,
// `guard` pixels before the row's end, it makes sure the row before has been taken (as a rule it was long ago; the test is one LDS
// read per row).  write_plane: the RGB24 plane in HBM is still wanted (the frame a later P-frame or the next call starts from) -
// a batch of key frames for a streamer writes it for its last frame only (round 5: the decode launch's HBM traffic went from
// plane out + plane back in + picture out, 6.4 GB for the headline's 300 frames, to the pictures alone).
template <class DEC>
__device__ __forceinline__ void decode_intra_frame(DEC& D, const Geom& g, u8* __restrict__ dst, u32* ring, int ring_pixels, u32* rows_word = nullptr, u32* taken_word = nullptr,
                                                   bool write_plane = true) {
  const int lane = D.lane;
  const u32 pm = (u32)ring_pixels - 1u;
  const int W = g.W, H = g.H, S = g.S, NP = g.NP;
  const int pad = S - 3 * W;  // 0..3 zero bytes after each row
  const int chunk = W < 64 ? W : 64;  // pixels of a predicted run rebuilt together: never more than a row, they read the row above
  // rows [flushed, to) are complete in the ring: pack them to RGB24 and store them (row padding = 0)
  int flushed = 0, rowbase = 0;  // rowbase = flushed * W: first pixel of the row being decoded
  auto flush_rows = [&](int to) __attribute__((always_inline)) {
    wave_fence();
    if (rows_word != nullptr) __hip_atomic_store(rows_word, (u32)to, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // rows [flushed, to) are whole in the ring
    if (!write_plane) {
      rowbase += (to - flushed) * W;
      flushed = to;
      return;
    }
    for (; flushed < to; flushed++, rowbase += W) {
      const u32 p0 = (u32)rowbase;
      u8* row = dst + (size_t)flushed * S;
      for (int gq = lane; gq * 4 < W; gq += 64) {
        const u32 q = p0 + 4u * gq;
        const int nv = min(4, W - 4 * gq);
        const u32 a = ring[q & pm], b = nv > 1 ? ring[(q + 1) & pm] : 0u, c = nv > 2 ? ring[(q + 2) & pm] : 0u, d = nv > 3 ? ring[(q + 3) & pm] : 0u;
        u32* o = (u32*)row + gq * 3;
        const int room = S - gq * 12;
        o[0] = a | (b << 24);
        if (room > 4) o[1] = (b >> 8) | (c << 16);
        if (room > 8) o[2] = (c >> 16) | (d << 8);
      }
    }
  };

  u32 lastpix = 0;  // last decoded pixel; 0 gives context 0 for the first pixel (cx = cx1 = 0, :419)
  // One loop for both phases so that every model routine is instantiated once:
  //   header phase (p <= W): literal + run length over the first row and pixel (0,1)  (:421-438)
  //   body: pixel type, literal if type 0, run length                                  (:443-494)
  int p = 0, t = 0;  // next pixel (raster index), type of the previous run
  const u32 slow_types = pad ? 0x38u : 0x18u;  // bit t: not one of the plain fills below (3 does not exist, 4 gradient, 5 when rows are padded)
  int lim = W + 1;     // where the runs of the current phase must end: the header phase covers pixels 0..W
  int rowend = W + 1;  // the row being decoded ends here (first row: with the header phase, one pixel later)
  int fastend = 0;     // (set with rowend when the header phase is over)
  // The streamer's share of the ring (see above).  Writing goes on, unchecked, from one guard point to the next: from `guard`
  // pixels before the end of row r to `guard` pixels before the end of row r + 1, plus the 255 pixels a run may overshoot and
  // the 64 a pass of the wave scribbles ahead.  Those positions alias rows up to r - 1 as long as 2 W + 319 - guard <= ring, so
  // at the guard point of row r the rows below r must have been taken (row r - 1 was announced a row ago).
  const int guard_px = max(0, 2 * W + 319 - ring_pixels);  // (< W: the ring is W + 512 pixels or more)
  int gpos = 0x7FFFFFFF, gneed = 0;
  auto guard = [&]() __attribute__((always_inline)) {
    if (taken_word != nullptr && SCPR_UNLIKELY(p >= gpos)) {
      while ((int)lds_peek_w(taken_word) < gneed) __builtin_amdgcn_s_sleep(1);
      gpos = 0x7FFFFFFF;
      fastend = min(rowend, NP);
    }
  };
  // One run: its type (after the header phase), the pixel of a literal, its length, its pixels.  Two instances:
  // the careful one does everything (header phase, coder block ends), the fast one is entered only where neither
  // can occur and leaves the per-symbol block-end test out (a test and a branch per symbol is ~20 cycles).
  auto run = [&](auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    D.template stamp<4>();
    u32 pend[3];  // the fast instance (wave decoder): every symbol's coder step is taken by the next symbol, under its table fetch
    constexpr bool CHAIN = FAST && DEC::kFastRuns;
    [[maybe_unused]] typename DEC::NAsk nask;
    if constexpr (CHAIN) {
      t = D.template fixed_p<false, true>(t, pend);
      D.fixed_n_ask(t, nask);
    } else if constexpr (FAST) t = D.template fixed_p<false>(t);
    else if (lim == NP) t = D.fixed_p(t);
    D.template stamp<0>();
    D.template event<13>();
    if (t == 0) D.template event<14>();
    u32 px = lastpix;
    if (t == 0) {  // DecodeRGB, screencap.cpp:662-679: contexts are the two previous bytes >> 2 (MAKECX1, screencap.h:35-36)
      u32 a = (lastpix >> 18) & 63, b = (lastpix >> 10) & 63;
      px = 0;
      if constexpr (DEC::kFastRuns) {  // (the wave decoder of versions 3 / 4: the coder step of a symbol under the next one's record fetch)
        u32 c = (u32)D.template colour<!FAST, CHAIN ? 2 : 1>((int)(a | (b << 6)), pend);
        px = c;
        b = a;
        a = c >> 2;
        c = (u32)D.template colour<!FAST, 2>(4096 + (int)(a | (b << 6)), pend);
        px |= c << 8;
        b = a;
        a = c >> 2;
        c = (u32)D.template colour<!FAST, 2>(8192 + (int)(a | (b << 6)), pend);
        px |= c << 16;
        if constexpr (!CHAIN) {  // (the fast instance hands the last step on to the run length)
          D.advance(pend[0], pend[1], pend[2]);
          D.template count<!FAST>();
        }
      } else {
#pragma unroll
        for (int plane = 0; plane < 3; plane++) {
          const u32 c = (u32)D.colour(plane * 4096 + (int)(a | (b << 6)));
          px |= c << (8 * plane);
          b = a;
          a = c >> 2;
        }
      }
      D.template stamp<1>();
    }
    int n;
    if constexpr (CHAIN) {
      n = D.template fixed_n<false, true, true, true>(t, pend, &nask);  // (its own step: under the run's ring read, below)
      D.ndec += t == 0 ? 5 : 2;  // the symbols of this run (type, three colour bytes of a literal, length)
    } else if constexpr (FAST) {
      n = D.template fixed_n<false>(t);
      D.ndec += t == 0 ? 5 : 2;
    } else {
      n = D.fixed_n(t);
      // (the careful instance runs once per row: the place to notice that the arena is full - alloc_dense -, at most a row late:
      // this run then ends the frame and nothing more is decoded; a test in the fast instance costs 2.5 % of the decoder)
      if (SCPR_UNLIKELY(D.oom)) lim = p;
    }
    D.template stamp<2>();
#ifdef SCPR_PROFILE
    if constexpr (std::is_same<DEC, WaveDec>::value) {  // key-frame runs by pixel type (slots 1.. of rprof: 0 1 2 [3] 4 5), and those longer than a pass
      D.rprof[1] += t == 0, D.rprof[2] += t == 1, D.rprof[3] += t == 2, D.rprof[4] += t == 4, D.rprof[5] += t == 5, D.rprof[6] += n > 64, D.rprof[7] += (u64)n;
    }
#endif
    // an empty run, or one longer than what is left (of the header row), ends the frame like the type that does not exist
    const int tt = SCPR_UNLIKELY((u32)(n - 1) >= (u32)(lim - p)) ? 3 : t;
    const int nf = ((slow_types >> tt) & 1u) ? 0 : n;  // pixels of the run for the common path below
    // literal / copy of the previous pixel (0, 1): every pixel of the run has the same value;
    // copy of the pixel above (2) or above-left (5): read one row back in the ring.
    // Almost every run fits one pass of the wave.  All 64 lanes store: the lanes past the end of the run scribble
    // on pixels that are not decoded yet (the ring is W + 512 pixels or more: 64 pixels ahead of the run alias
    // pixels more than a row and 448 pixels back, older than anything still read), which spares the lane mask.
    // (The other types take this path with an empty run and are handled below.)
    const u32 back = (u32)W + (u32)(t >> 2);
    u32 v = px;
    int m = min(chunk, nf);
    u32 pq = (u32)(p + lane);
    wave_fence();  // pixels written by other lanes are read here
    if (t >= 2) v = ring[(pq - back) & pm];
    if constexpr (CHAIN) D.advance(pend[0], pend[1], pend[2]);  // the run length's coder step, while the row above is on its way
    ring[pq & pm] = v;
    wave_fence();
    lastpix = rdl(v, m - 1);
    if (SCPR_UNLIKELY((u32)(nf - 1) >= (u32)chunk)) {  // a long run, or one of the rare types
     if (nf > 0) {
#pragma nounroll
      for (int q0 = chunk; q0 < nf; q0 += chunk) {
        wave_fence();  // a run may be longer than a row
        m = min(chunk, nf - q0);
        pq = (u32)(p + q0 + lane);
        if (t >= 2) v = ring[(pq - back) & pm];
        ring[pq & pm] = v;
      }
      wave_fence();
      lastpix = rdl(v, m - 1);
     } else
     if (tt == 3) {  // the frame is refused, the loop ends here
      D.bad = true;
      p = NP - n;
     } else {  // above-left with row padding in the way, or the gradient predictor
      lastpix = px;
      u32 v = 0;
      int m = 0;
      for (int q0 = 0; q0 < n; q0 += chunk) {
        wave_fence();
        m = min(chunk, n - q0);
        const u32 pq = (u32)(p + q0 + lane);
        u32 tl = ring[(pq - W - 1) & pm];
        if (pad) {  // in column 0 "above-left" is the bytes just before the row above in memory: the tail of the
                    // last pixel two rows up followed by that row's padding (screencap.cpp:881)
          const int xq = ((int)pq - rowbase) % W;  // (not a loop: its trip count would differ from lane to lane)
          tl >>= xq == 0 ? 8 * pad : 0;
        }
        if (t == 5) {
          v = tl;
        } else {  // gradient: previous + top - topleft, a running sum along the run (mod 256 per channel)
          const u32 tp = ring[(pq - W) & pm];
          int d0 = (int)(tp & 255) - (int)(tl & 255), d1 = (int)((tp >> 8) & 255) - (int)((tl >> 8) & 255), d2 = (int)((tp >> 16) & 255) - (int)((tl >> 16) & 255);
          d0 = lane >= m ? 0 : d0, d1 = lane >= m ? 0 : d1, d2 = lane >= m ? 0 : d2;
          d0 = wave_incl_scan(d0);
          d1 = wave_incl_scan(d1);
          d2 = wave_incl_scan(d2);
          v = (u32)(((int)(lastpix & 255) + d0) & 255) | ((u32)(((int)((lastpix >> 8) & 255) + d1) & 255) << 8) | ((u32)(((int)((lastpix >> 16) & 255) + d2) & 255) << 16);
          lastpix = rdl(v, m - 1);
        }
        lds_st_if(&ring[pq & pm], v, lane < m);
      }
      wave_fence();
      lastpix = rdl(v, m - 1);
     }
    }
    D.template stamp<3>();
    p += n;
  };
  auto row_end = [&]() __attribute__((always_inline)) {
    if (SCPR_UNLIKELY(p >= rowend)) {  // a row is complete (the first time: the header phase is over as well)
      lim = NP;
      int done = flushed + 1;
      while ((done + 1) * W <= p) done++;
      flush_rows(done);
      rowend = rowbase + W;
      fastend = min(rowend, NP);  // the fast runs below go on while p < fastend: to the end of the row, never past the frame
      if (taken_word != nullptr) {  // ... or to this row's guard point
        gneed = flushed;
        gpos = rowend - guard_px;
        fastend = min(gpos, NP);
      }
    }
  };
  while (SCPR_LIKELY(p < NP)) {
    run(std::false_type{});
    guard();
    row_end();
    if constexpr (DEC::kFastRuns) {
      if (SCPR_LIKELY(lim == NP)) {
        // as long as the row goes on and the coder block does not end within the next run (both differences negative)
        while (SCPR_LIKELY((int)((u32)(p - fastend) & (u32)(D.ndec - (kBlockEntries - 5))) < 0)) run(std::true_type{});
        guard();
        row_end();
      }
    }
  }
  if (!D.bad) flush_rows(H);
}

// ---- helper waves -------------------------------------------------------------------------------------------------
// Whatever a P-frame needs that is not the symbol chain is bulk copying: the new plane starts as the previous one (6 MB at
// 1080p: 2 ms for one wave, an eighth of the frame) and motion blocks are copied from it.  A GOP with P-frames therefore runs as
// a workgroup: wave 0 is the chain, the other waves sleep on an LDS word and do those copies when told - at the start of a frame
// the plane copy runs while the chain decodes the frame header and the block types (which touch no pixel), motion blocks are
// queued and copied in batches while the chain goes on with the next symbols.
__device__ __forceinline__ u32 lds_peek(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
struct HelpJobs;  // (the motion-block queue is WaveLds::jobs)
__device__ __forceinline__ void copy_motion_jobs(const uint2* jobs, int first, int njobs, int stride_jobs, u8* __restrict__ cur, const u8* __restrict__ prv, int S, int lane) {
  for (int base = first; base < njobs; base += stride_jobs) {
    const int jb = base + (lane >> 4), r = lane & 15;
    if (jb < njobs) {
      const uint2 j = jobs[jb];
      const int jx1 = (int)(j.x & 0x1FFF), jy1 = (int)((j.x >> 13) & 0x1FFF), jw = (int)((j.x >> 26) & 15) + 1;
      const int jh = (int)(j.y & 15) + 1, jmx = (int)((j.y >> 4) & 0x3FF) - 512, jmy = (int)((j.y >> 14) & 0x3FF) - 512;
      if (r < jh) {
        const u8* sp = prv + (size_t)(jy1 + r + jmy) * S + (jx1 + jmx) * 3;
        u8* dp = cur + (size_t)(jy1 + r) * S + jx1 * 3;
        const int wb = jw * 3;
        u32 v[12];
#pragma unroll
        for (int k = 0; k < 12; k++)
          if (k * 4 < wb) __builtin_memcpy(&v[k], sp + k * 4, 4);
#pragma unroll
        for (int k = 0; k < 12; k++) {
          if (k * 4 + 4 <= wb) __builtin_memcpy(dp + k * 4, &v[k], 4);
          else if (k * 4 < wb)
            for (int q = 0; q < wb - k * 4; q++) dp[k * 4 + q] = (u8)(v[k] >> (8 * q));
        }
      }
    }
  }
}
// slice `part` of `parts` of a plane copy (16 KiB pieces, sixteen 16-byte loads in flight per lane)
__device__ __forceinline__ void copy_plane_part(u8* __restrict__ cur, const u8* __restrict__ prv, size_t bytes, int part, int parts, int lane) {
  size_t o = (size_t)part * 16384 + (size_t)lane * 16;
  const size_t step = (size_t)parts * 16384;
  for (; o + 15 * 1024 + 16 <= bytes; o += step) {
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = *(const uint4*)(prv + o + k * 1024);
#pragma unroll
    for (int k = 0; k < 16; k++) *(uint4*)(cur + o + k * 1024) = v[k];
  }
  if (part == 0) {  // what is left of the last, partial piece
    const size_t whole = (bytes / 16384) * 16384;
    // (a piece that did not fit the test above although it starts inside the plane)
    for (size_t q = whole + (size_t)lane * 16; q + 16 <= bytes; q += 1024) *(uint4*)(cur + q) = *(const uint4*)(prv + q);
    for (size_t q = (bytes & ~(size_t)15) + lane; q < bytes; q += 64) cur[q] = prv[q];
  }
}
// one motion block, by a whole wave: lane = row * 4 + quarter of the row (four pixels, three words)
__device__ __forceinline__ void copy_motion_block(uint2 j, u8* __restrict__ cur, const u8* __restrict__ prv, int S, int lane) {
  const int jx1 = (int)(j.x & 0x1FFF), jy1 = (int)((j.x >> 13) & 0x1FFF), jw = (int)((j.x >> 26) & 15) + 1;
  const int jh = (int)(j.y & 15) + 1, jmx = (int)((j.y >> 4) & 0x3FF) - 512, jmy = (int)((j.y >> 14) & 0x3FF) - 512;
  const int r = lane >> 2, o = (lane & 3) * 12, wb = jw * 3;
  if (r < jh && o < wb) {
    const u8* sp = prv + (size_t)(jy1 + r + jmy) * S + (jx1 + jmx) * 3 + o;
    u8* dp = cur + (size_t)(jy1 + r) * S + jx1 * 3 + o;
    if (wb - o >= 12) {
      u32 v[3];
#pragma unroll
      for (int k = 0; k < 3; k++) __builtin_memcpy(&v[k], sp + k * 4, 4);
#pragma unroll
      for (int k = 0; k < 3; k++) __builtin_memcpy(dp + k * 4, &v[k], 4);
    } else {
      u8 v[9];
#pragma unroll
      for (int k = 0; k < 9; k++)
        if (k < wb - o) v[k] = sp[k];
#pragma unroll
      for (int k = 0; k < 9; k++)
        if (k < wb - o) dp[k] = v[k];
    }
  }
}
__device__ __forceinline__ void helper_loop(WaveLds& L, int S, int hw, int nh) {
  const int lane = lane_id();
  u32 seen = 0, next = (u32)hw;  // the command last seen; the number of this helper's next motion job
  u8* cur = nullptr;
  const u8* prv = nullptr;
  for (;;) {
    const u32 sq = lds_peek(&L.hc.seq);
    if (sq != seen) {
      seen = sq;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const u32 op = lds_peek(&L.hc.op);
      if (op == HOP_EXIT) break;
      cur = (u8*)(size_t)L.hc.dst;  // (the motion jobs that follow are for these planes too)
      prv = (const u8*)(size_t)L.hc.src;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the planes were written by other waves: nothing stale from this CU's L1
      copy_plane_part(cur, prv, (size_t)L.hc.bytes, hw, nh, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached L2
      if (lane == 0) __hip_atomic_fetch_add(&L.hc.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    const u32 tail = __hip_atomic_load(&L.hc.jtail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((int)(tail - next) > 0) {
      const uint2 j = L.jobs[next & 255u];
      copy_motion_block(j, cur, prv, S, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      next += (u32)nh;
      if (lane == 0) __hip_atomic_store(&L.hc.hprog[hw], next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __builtin_amdgcn_s_sleep(2);
    }
  }
}
// the chain's side
__device__ __forceinline__ void help_post(WaveLds& L, u32 op, const u8* src, u8* dst, u32 bytes) {
  if (lane_id() == 0) {
    L.hc.op = op;
    L.hc.src = (u64)(size_t)src;
    L.hc.dst = (u64)(size_t)dst;
    L.hc.bytes = bytes;
    __hip_atomic_store(&L.hc.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&L.hc.seq, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
// every motion job below number `upto` has been copied (and its stores have reached L2)
__device__ __forceinline__ void help_wait_jobs(WaveLds& L, u32 upto, int nhelp) {
  const int lane = lane_id();
  for (;;) {
    const u32 v = lane < nhelp ? lds_peek(&L.hc.hprog[lane]) : upto;
    if (!__ballot((int)(v - upto) < 0)) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void help_wait(WaveLds& L) {
  const u32 nh = L.hc.nhelp;
  while (lds_peek(&L.hc.done) < nh) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__constant__ const u32 kRcp16[17] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096};
// P-frame (DecompressP, screencap.cpp:1275-1432).  The new plane starts as a copy of the previous
// one; motion blocks are copied from the previous plane, pixel-coded rects are rebuilt in an LDS
// tile (with the row above and the column to the left as predictor context) and written back.
template <class DEC>
__device__ __forceinline__ void decode_inter_frame(DEC& D, const Geom& g, u8* __restrict__ cur, const u8* __restrict__ prv, const u8* head, u8* bts, int far_x, int far_y) {
  const int lane = D.lane;
  const int W = g.W, H = g.H, S = g.S;
  const int nbx = (W + 15) >> 4, nby = (H + 15) >> 4, nblocks = nbx * nby;
  D.template stamp<4>();
  __threadfence();  // the previous plane was written by this wave (or by another kernel): make it readable
  int nhelp = 0;
  if constexpr (DEC::kHelpers) nhelp = (int)D.L.hc.nhelp;
  auto hpost = [&](u32 op, u32 bytes) __attribute__((always_inline)) {
    if constexpr (DEC::kHelpers) help_post(D.L, op, prv, cur, bytes);
  };
  auto hwait = [&]() __attribute__((always_inline)) {
    if constexpr (DEC::kHelpers) help_wait(D.L);
  };
  bool copy_pending = false;
  if (nhelp) {  // the helpers copy the plane while this wave decodes the header and the block types
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's stores to the previous plane have reached L2)
    hpost(HOP_COPY, (u32)((size_t)H * S));
    copy_pending = true;
  } else {
    copy_plane_part(cur, prv, (size_t)H * S, 0, 1, lane);
  }
  const u32 first = *(const volatile u8*)head;
  if (SCPR_UNLIKELY(!(first & 1u))) {  // nothing changed (:1286-1291)
    if (copy_pending) hwait();
    return;
  }
  D.stream_init(head + 1);
  auto get_x = [&]() __attribute__((always_inline)) {
    D.tick();
    return D.fixed_x(0);
  };
  int lo = get_x(), hi = get_x();
  const int xx1 = (hi << 8) + lo;
  lo = get_x();
  hi = get_x();
  const int xx2 = (hi << 8) + lo;
  if (SCPR_UNLIKELY(xx2 >= nblocks || xx1 > xx2)) {
    D.bad = true;
    if (copy_pending) hwait();
    return;
  }
  for (int i = lane; i < nblocks; i += 64) bts[i] = 0;
  wave_fence();
  for (int b = xx1; b <= xx2 && !D.bad;) {  // block types, run-length coded (:1306-1313)
    D.tick();
    const int c = D.fixed_bt();
    D.tick();
    const int n = D.fixed_x(1);
    if (SCPR_UNLIKELY(n < 1 || b + n > nblocks)) {
      D.bad = true;
      break;
    }
    for (int q = lane; q < n; q += 64) bts[b + q] = (u8)c;
    b += n;
  }
  wave_fence();
  if (copy_pending) hwait();  // from here on the plane is touched
  D.template stamp<5>();
  u32 lastpix = 0;  // cx = cx1 = 0 (:1317)
  int lastmx = 0, lastmy = 0;
  const u32 rcpn = (u32)((0x100000000ull + (u32)nbx - 1u) / (u32)nbx);
  u32* tile = D.L.tile;
  u32* ptile = D.L.ptile;
  uint2* jobs = D.L.jobs;
  // Motion-block copies (the source is the previous plane, which this frame never modifies): with helper waves every block
  // is handed over as soon as it is decoded and copied while the chain goes on (see WaveLds::hc); a rect waits for them
  // only when one of the blocks its context border comes from is a motion block.  Without helpers the blocks are
  // queued and copied by this wave, four at a time, before the next rect and at the end of the frame.
  int njobs = 0;                            // without helpers: blocks waiting in `jobs`
  u32 jt = 0;                               // with helpers: blocks handed over so far (the ring goes on from frame to frame)
  if constexpr (DEC::kHelpers) jt = D.L.hc.jtail;
  auto flush_jobs = [&]() __attribute__((always_inline)) {
    wave_fence();
    if constexpr (DEC::kHelpers) {
      if (nhelp) {
        help_wait_jobs(D.L, jt, nhelp);
        return;
      }
    }
    copy_motion_jobs(jobs, 0, njobs, 4, cur, prv, S, lane);
    njobs = 0;
    wave_fence();
  };
  for (int gbase = 0; gbase < nblocks && !D.bad; gbase += 64) {
   const u32 tb = gbase + lane < nblocks ? bts[gbase + lane] : 0u;
   u64 bm = __ballot(tb != 0);
   while (bm && !D.bad) {
    const int bj = __builtin_ctzll(bm);
    bm &= bm - 1;
    const int b = gbase + bj;
    const int t = (int)rdl(tb, bj);
    const int by = nbx == 1 ? b : (int)__umulhi((u32)b, rcpn), bx = b - by * nbx;  // == b / nbx (b < 2^18, 2 <= nbx <= 512: the rounding error stays below one; the reciprocal of 1 does not fit 32 bits)
    int x1 = bx * 16, y1 = by * 16, x2 = min(x1 + 16, W), y2 = min(y1 + 16, H);
    if ((t - 1) & 1) {  // changed rect inside the block (:1333-1346)
      D.tick();
      const int a0 = D.fixed_sxy(0);
      D.tick();
      const int a1 = D.fixed_sxy(1);
      D.tick();
      const int a2 = D.fixed_sxy(2);
      D.tick();
      const int a3 = D.fixed_sxy(3);
      x2 = x1 + a2 + 1;
      y2 = y1 + a3 + 1;
      x1 += a0;
      y1 += a1;
      if (SCPR_UNLIKELY(x2 > W || y2 > H || x1 >= x2 || y1 >= y2)) {
        D.bad = true;
        break;
      }
    }
    const int w = x2 - x1, h = y2 - y1;
    if ((t - 1) & 2) {  // motion block (:1348-1368)
      int mx = lastmx, my = lastmy;
      D.tick();
      if (!D.get_bool()) {
        D.tick();
        mx = D.fixed_mv(0) - far_x;
        D.tick();
        my = D.fixed_mv(1) - far_y;
      }
      lastmx = mx;
      lastmy = my;
      if (SCPR_UNLIKELY(x1 + mx < 0 || y1 + my < 0 || x2 + mx > W || y2 + my > H)) {
        D.bad = true;
        break;
      }
      const uint2 job = make_uint2((u32)x1 | ((u32)y1 << 13) | ((u32)(w - 1) << 26), (u32)(h - 1) | ((u32)(mx + 512) << 4) | ((u32)(my + 512) << 14));
      if (nhelp) {
        if constexpr (DEC::kHelpers) {
          // (the ring slots written from here to the next test held the jobs 193..256 back)
          if (SCPR_UNLIKELY((jt & 63u) == 0)) help_wait_jobs(D.L, jt - 192u, nhelp);
          jobs[jt & 255u] = job;
          jt++;
          __hip_atomic_store(&D.L.hc.jtail, jt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        if (lane == 0) jobs[njobs] = job;
        if (SCPR_UNLIKELY(++njobs == 256)) flush_jobs();
      }
      D.template stamp<20>();
      continue;
    }
    D.template stamp<6>();
    if (nhelp) {  // a copied block may be this rect's left/top context: is one of the three blocks the border lies in a motion block?
      const int nq = lane == 0 ? (bx > 0 ? b - 1 : -1) : lane == 1 ? (by > 0 ? b - nbx : -1) : lane == 2 ? (bx > 0 && by > 0 ? b - nbx - 1 : -1) : -1;
      const u32 nt = nq >= 0 ? (u32)bts[nq] : 0u;
      if (__ballot(nt >= 3u)) flush_jobs();
    } else if (njobs) {
      flush_jobs();
    }
    // pixel-coded rect (:1370-1421): context border from the plane, then runs inside the tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores to the plane have reached L2
    D.template stamp<3>();
    u32 border = 0;
    int border_at = 0;
    {
      // the 33 border pixels (row above: lanes 0..16, column to the left: lanes 17..32) in ONE round trip to L2 - a
      // loop over the 289 tile words with the loads inside waits for L2 five times per rect
      // (The tile is not cleared: every cell is written - the 33 border cells below, the rect's cells by its runs - before
      // anything that is kept reads it.)  The border is asked for here and put into the tile when the first run of the rect is
      // about to be filled in: the symbols of that run (type, literal, length) are decoded while the L2 round trip is under way.
      const int ty = lane < 17 ? 0 : lane - 16, tx = lane < 17 ? lane : 0;
      if (lane < 33 && ty <= h && tx <= w) {
        const int xq = x1 - 1 + tx, yq = y1 - 1 + ty;
        if (xq >= 0 && yq >= 0) border = ld3_l2(cur + (size_t)yq * S + xq * 3);
      }
      border_at = ty * 17 + tx;
    }
    // The rect's pixels in the previous frame come with the same round trip: a third of the runs of a P-frame copy from
    // there (type 3), and a load inside each of those runs was an L1/L2 round trip on the chain, a dozen per rect.
    u32 pv[4];  // (lane = row * 4 + quarter: pixels 4 * quarter .. + 3 of the row)
    const int prow = lane >> 2, pcol = (lane & 3) * 4;
    {
      const u8* pp = prv + (size_t)(y1 + prow) * S + (x1 + pcol) * 3;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        pv[k] = 0;
        if (prow < h && pcol + k < w) pv[k] = ld3(pp + 3 * k);
      }
    }
    bool border_due = true;
    D.template stamp<6>();
    int li = 0, pt = 0;  // position in the rect, row-major (the rect is w * h <= 256 pixels)
    int rrow = 0, rcol = 0;  // ... and as row / column (li == rrow * w + rcol)
    // The cells of a run go to the tile when the NEXT run is placed (or the rect ends): their value is on its way from LDS
    // until then, and nothing on the chain waits for it - a run that copies (71 % of a P-frame's runs) used to stall on that
    // round trip twice, before its store and before handing its last pixel to the scalar unit, which only a literal that
    // follows needs (for its contexts).  pend_at: per lane, the cell (the spare cell 17 * 17: none); pend_last: the lane
    // whose value is the run's last pixel.
    u32 pend_v = lastpix;
    int pend_at = 17 * 17, pend_last = 0;
    const int lend = w * h;
    const int lc = min(lane, 15);
    const int rcpw = (int)kRcp16[w];  // ceil(65536 / w): (li * rcpw) >> 16 == li / w for li <= 256, w <= 16
    // One run of the rect, in two instances like the key-frame loop: the careful one tests for the end of the coder
    // block after every symbol, the fast one is entered while the block cannot end within a run and counts the
    // run's symbols in one go.  A refused stream ends the rect (li = lend) instead of leaving the loops from inside.
    auto prun = [&](auto fast_tag) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_tag)::value;
      const int last_t = pt;
      D.template stamp<7>();
      u32 pend[3];  // the fast instance (wave decoder): every symbol's coder step is taken by the next symbol, under its table fetch
      constexpr bool CHAIN = FAST && DEC::kFastRuns;
      [[maybe_unused]] typename DEC::NAsk nask;
      if constexpr (CHAIN) {
        pt = D.template fixed_p<false, true>(last_t, pend);
        D.fixed_n_ask(pt, nask);  // (the run-length table: on its way while a literal's colour symbols are decoded, see decode_intra_frame)
      } else pt = D.template fixed_p<!FAST>(last_t);
      D.template stamp<0>();
      D.template event<13>();
      if (pt == 0) D.template event<14>();
      u32 px = 0;
      if (pt == 0) {
        lastpix = rdl(pend_v, pend_last) & 0xFFFFFFu;  // the last pixel of the run before (long arrived)
        u32 a = (lastpix >> 18) & 63, bb = (lastpix >> 10) & 63;
        // (unrolled: a rolled loop carries the packet block's register round its back edge, and every copy of it waits for
        // whatever load may still be writing it - 11.03 -> 10.58 ms per P-frame, round 4; in round 3, with the block load a FLAT
        // load whose waits also covered the LDS queue, the unrolled form had measured slower and the blame went to the instruction cache)
#pragma unroll
        for (int plane = 0; plane < 3; plane++) {
          u32 c;
          if constexpr (CHAIN) c = (u32)D.template colour<false, 2, true>(plane * 4096 + (int)(a | (bb << 6)), pend);
          else c = (u32)D.template colour<!FAST, 0, true>(plane * 4096 + (int)(a | (bb << 6)));
          px |= c << (8 * plane);
          bb = a;
          a = c >> 2;
        }
      }
      D.template stamp<1>();
      int rem;
      if constexpr (CHAIN) rem = D.template fixed_n<false, true, false, true>(pt, pend, &nask);
      else rem = D.template fixed_n<!FAST>(pt);
      D.template stamp<2>();
      if constexpr (!FAST) {  // (a rect always starts in this instance)
        if (SCPR_UNLIKELY(D.oom)) D.bad = true;  // the arena is full (alloc_dense): nothing more is decoded, at most a rect late
        if (border_due) {
          wave_fence();
          lds_st_if(&tile[border_at], border, lane < 33);
#pragma unroll
          for (int k = 0; k < 4; k++)
            lds_st_if(&ptile[prow * w + pcol + k], pv[k], prow < h && pcol + k < w);
          wave_fence();
          border_due = false;
        }
      }
      if (FAST) D.ndec += pt == 0 ? 5 : 2;
      if (SCPR_UNLIKELY(rem < 1)) {
        D.bad = true;
        rem = 0;
        li = lend;
      }
      // Most runs of a rect are a few pixels that stay on their rect row (2.8 pixels on average on desktop content, 38 runs to a
      // rect): for those everything is an offset from the run's first cell in the tile - "left" is the cell before it (the
      // tile's column 0 is the column left of the rect), "above" and "above-left" the cells one tile row up - and the row / column
      // of the next run is an addition.  One LDS read (none for a literal), one write, no multiplication: the general forms
      // below, which place every lane's pixel by a division by the rect's width, cost such a run ~650 cycles.
      // negative: the single-row form (at least one pixel, the row holds them all, not the gradient; three sign bits, no select)
      int slow_run = ~((rem - 1) | (w - rcol - rem) | ((pt ^ 4) - 1));
      if (SCPR_LIKELY(slow_run < 0)) {
        const int base = (rrow + 1) * 17 + rcol + 1;
        wave_fence();
        tile[pend_at] = pend_v;  // the run before (this run may copy from it)
        wave_fence();
        u32 v = px;
        if (pt != 0) {
          // where lane 0 reads (a cell of the tile, or - cells past the tile's end - of the previous frame's rect), and whether
          // the lanes read side by side (above, above-left, previous frame) or all the same cell (left).  ARITHMETIC on the type,
          // not choices: the optimiser turned the nested choices into a tree of branches (flag registers, exec tests, four taken
          // branches with padding between their blocks: ~25 instructions where these are 14; round 4, ISA listing of the run loop)
          u32 upt = (u32)pt;
          asm volatile("" : "+s"(upt));
          const int back = 1 + (int)(min(upt >> 1, 1u) << 4) + (int)(upt >> 2);  // 1 -> 1 (left), 2 -> 17 (above), 5 -> 18 (above-left)
          const u32 is3 = 0u - (((upt ^ 3u) - 1u) >> 31);                          // all ones for 3 (previous frame)
          const int s0 = (int)((is3 & (u32)((int)(ptile - tile) + li)) | (~is3 & (u32)(base - back)));
          const int each = 0 - (int)((0u - (upt ^ 1u)) >> 31);                     // 0 for 1 (every lane the same cell), -1 otherwise
          v = tile[s0 + (lc & each)];
        }
        // (no lane mask: the lanes past the run will write to the spare cell)
        pend_v = v;
        pend_at = lane < rem ? base + lane : 17 * 17;
        pend_last = rem - 1;
        li += rem;
        rcol += rem;
        if (rcol == w) rcol = 0, rrow++;
        rem = 0;
      }
      D.template stamp<21>();
      // (two plain ifs, the common one first: everything the general forms need - their tests included - stays off the path of
      // the single-row runs)
      if constexpr (DEC::kFastRuns && FAST) asm volatile("" : "+s"(slow_run));
      else slow_run = (int)rfl((u32)slow_run);  // (the careful instance, and the version 2 decoder, whose symbols come off the vector unit)
      if (SCPR_UNLIKELY(slow_run >= 0)) {
#ifdef SCPR_PROFILE
        if constexpr (std::is_same<DEC, WaveDec>::value) {
          D.rprof[0]++;
          D.rprof[1] += pt == 0, D.rprof[2] += pt == 1, D.rprof[3] += pt == 2, D.rprof[4] += pt == 3, D.rprof[5] += pt > 3;
          D.rprof[6] += rem > 64;
          D.rprof[7] += (u64)(rem > 0 ? rem : 0);
        }
#endif
        wave_fence();
        tile[pend_at] = pend_v;  // (the general forms work on the tile as it is and hand the last pixel over at once)
        wave_fence();
        lastpix = rdl(pend_v, pend_last) & 0xFFFFFFu;
        // Literal (0), left (1), above (2) and previous frame (3): no pixel of the run depends on another pixel of the run that
        // is not a plain copy of it, so the whole run is one pass - lane i takes the run's i-th pixel wherever the rect's rows
        // wrap it to.  "Left" is the pixel left of the run's start on its first row and the column left of the rect on the rows
        // below; "above" of a pixel further than a rect row into the run is another pixel of the run, and through it the pixel
        // above run pixel i mod w.  (Above-left and the gradient go row by row in the loop below.)
        if (SCPR_LIKELY(!((0x30 >> pt) & 1)) && rem > 0) {
          const int li0 = li;
          if (SCPR_UNLIKELY(li0 + rem > lend)) {  // the run goes on below the rect
            D.bad = true;
            li = lend;
          } else {
            const int row0 = (li0 * rcpw) >> 16, col0 = li0 - row0 * w;
            u32 v = px;
            for (int b = 0; b < rem; b += 64) {
              const int i = min(b + lane, rem - 1);
              const bool act = b + lane < rem;
              const int li = li0 + i;
              const int row = (li * rcpw) >> 16, col = li - row * w;
              wave_fence();
              if (pt == 1) {
                v = tile[(row + 1) * 17 + (row == row0 ? col0 : 0)];
              } else if (pt == 2) {
                const int ls = li0 + (i - ((i * rcpw) >> 16) * w);
                const int rs = (ls * rcpw) >> 16;
                v = tile[rs * 17 + (ls - rs * w) + 1];
              } else if (pt == 3) {
                v = ptile[li];
              }
              lds_st_if(&tile[(row + 1) * 17 + col + 1], v, act);
            }
            wave_fence();
            lastpix = rdl(v, (rem - 1) & 63) & 0xFFFFFFu;
            li = li0 + rem;
          }
          rem = 0;
        }
        while (rem > 0) {
          if (SCPR_UNLIKELY(li >= lend)) {  // the run goes on below the rect
            D.bad = true;
            break;
          }
          const int rrow = (li * rcpw) >> 16, rcol = li - rrow * w;
          const int seg = min(rem, w - rcol);  // pixels of this run on the current rect row (<= 16)
          const int ty = rrow + 1, tx0 = rcol + 1;
          const bool act = lane < seg;
          const int tx = tx0 + lane;
          wave_fence();
          // literal (0): px.  Copies inside the tile - of the previous pixel (1: one word for every lane), of the pixel
          // above (2) or above-left (5) - are one read at an index built from the type (every lane reads: the lanes past
          // the segment stay inside the tile's rows through lc).  The copy from the previous frame (3) and the gradient
          // (4) are rare and replace the result.
          u32 v = px;
          if (pt != 0) {
            const int srow = pt == 1 ? ty : ty - 1, scol = tx0 - (pt == 2 ? 0 : 1);
            v = tile[srow * 17 + scol + (pt == 1 ? 0 : lc)];
          }
          if (SCPR_UNLIKELY((0x18u >> pt) & 1u)) {
            if (pt == 3) {
              {  // (read by every lane - the lanes past the segment read the segment's first pixel -, chosen afterwards)
                const u32 pvv = ptile[li + (act ? lane : 0)];
                v = act ? pvv : px;
              }
            } else {
              // (read by every lane, inside the tile's row through lc; the lanes past the segment count as 0 below)
              const u32 tp = tile[(ty - 1) * 17 + tx0 + lc], tl = tile[(ty - 1) * 17 + tx0 + lc - 1];
              const u32 base = tile[ty * 17 + tx0 - 1];
              int d0 = (int)(tp & 255) - (int)(tl & 255), d1 = (int)((tp >> 8) & 255) - (int)((tl >> 8) & 255), d2 = (int)((tp >> 16) & 255) - (int)((tl >> 16) & 255);
              d0 = act ? d0 : 0, d1 = act ? d1 : 0, d2 = act ? d2 : 0;
              d0 = row_incl_scan(d0);
              d1 = row_incl_scan(d1);
              d2 = row_incl_scan(d2);
              v = (u32)(((int)(base & 255) + d0) & 255) | ((u32)(((int)((base >> 8) & 255) + d1) & 255) << 8) | ((u32)(((int)((base >> 16) & 255) + d2) & 255) << 16);
            }
          }
          lds_st_if(&tile[ty * 17 + tx], v, act);
          wave_fence();
          lastpix = rdl(v, seg - 1) & 0xFFFFFFu;
          rem -= seg;
          li += seg;
        }
        if (SCPR_UNLIKELY(D.bad)) li = lend;
        rrow = (li * rcpw) >> 16;  // (the general forms move li only)
        rcol = li - rrow * w;
        pend_v = lastpix;
        pend_at = 17 * 17;
        pend_last = 0;
      }
      D.template stamp<22>();
    };
    while (li < lend) {
      D.tick();
      prun(std::false_type{});
      while (SCPR_LIKELY((int)((u32)(li - lend) & (u32)(D.ndec - (kBlockEntries - 5))) < 0)) prun(std::true_type{});
    }
    D.template stamp<7>();
    wave_fence();
    tile[pend_at] = pend_v;  // the last run's cells
    lastpix = rdl(pend_v, pend_last) & 0xFFFFFFu;
    wave_fence();
    for (int i = lane; i < w * h; i += 64) {  // the finished rect goes to the plane
      const int yy = (i * rcpw) >> 16, xq = i - yy * w;
      const u32 v = tile[(yy + 1) * 17 + xq + 1];
      u8* dp = cur + (size_t)(y1 + yy) * S + (x1 + xq) * 3;
      dp[0] = (u8)v;
      dp[1] = (u8)(v >> 8);
      dp[2] = (u8)(v >> 16);
    }
    D.template stamp<19>();
   }
  }
  flush_jobs();  // the frame is complete (and the next one reads it) when every block is in place
}

// The second wave of a key frame's workgroup when the picture goes to the host: rows [done, rows) of the plane - RGB24, complete
// in HBM, announced by the chain in `*rows_word` - go out as RGB32 with alpha 255 (ScreenCodec::DecompressFrame's conversion,
// screencap.cpp:1711-1725) to the host's buffer (mapped into the device's address space: the stores cross PCIe), four pixels per
// lane.  Ends when the chain has set bit 31 and everything announced is sent: the chain sets it on every way out.
constexpr u32 kRowsEnd = 0x80000000u;
__device__ __forceinline__ void row_streamer(const u32* rows_word, u32* taken_word, const u32* ring, int ring_pixels, u8* __restrict__ hdst, int hpitch, int hbpp, const Geom& g) {
  // the rows the chain has finished, out of its LDS ring (32-bit pixels by raster position) into the picture: RGB32, or (hbpp 3,
  // hpitch = the plane's own stride) RGB24 rows exactly as the chain would have packed them into the plane
  const int lane = lane_id();
  const int W = g.W;
  const u32 pm = (u32)ring_pixels - 1u;
  const bool h16 = (((size_t)hdst | (size_t)hpitch) & 15) == 0;
  int done = 0;
  for (;;) {
    const u32 r = __hip_atomic_load(rows_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int upto = min((int)(r & ~kRowsEnd), g.H);
    if (done < upto) {
      for (; done < upto; done++) {
        const u32 q0 = (u32)done * (u32)W;
        u32* ho = (u32*)(hdst + (size_t)done * hpitch);
        if (hbpp == 3) {
          for (int gq = lane; gq * 4 < W; gq += 64) {
            const int nv = min(4, W - gq * 4), room = hpitch - gq * 12;
            const u32 q = q0 + 4u * (u32)gq;
            const u32 a = ring[q & pm], b = nv > 1 ? ring[(q + 1) & pm] : 0u, c = nv > 2 ? ring[(q + 2) & pm] : 0u, d = nv > 3 ? ring[(q + 3) & pm] : 0u;
            u32* o = ho + gq * 3;
            o[0] = a | (b << 24);
            if (room > 4) o[1] = (b >> 8) | (c << 16);
            if (room > 8) o[2] = (c >> 16) | (d << 8);
          }
        } else
        for (int gq = lane; gq * 4 < W; gq += 64) {
          const int nv = min(4, W - gq * 4);
          const u32 q = q0 + 4u * (u32)gq;
          const u32 A = 0xFF000000u;
          const u32 p0 = ring[q & pm] | A, p1 = nv > 1 ? ring[(q + 1) & pm] | A : 0u, p2 = nv > 2 ? ring[(q + 2) & pm] | A : 0u, p3 = nv > 3 ? ring[(q + 3) & pm] | A : 0u;
          u32* o = ho + gq * 4;
          if (h16 && nv == 4) {
            *(uint4*)o = make_uint4(p0, p1, p2, p3);
          } else {
            o[0] = p0;
            if (nv > 1) o[1] = p1;
            if (nv > 2) o[2] = p2;
            if (nv > 3) o[3] = p3;
          }
        }
        // (the row's pixels are in registers or gone: the chain may have its part of the ring back)
        wave_fence();
        __hip_atomic_store(taken_word, (u32)(done + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      continue;
    }
    if (r & kRowsEnd) break;
    __builtin_amdgcn_s_sleep(32);
  }
}

template <bool HAS_P>
__global__ __launch_bounds__(HAS_P ? 512 : 128) void k_decode_gop_w(const u8* __restrict__ packets, const u8* packets_end, const DecFrame* __restrict__ frames, const DecGop* __restrict__ gops,
                                                     u8* __restrict__ planes, Geom g, DecRec* __restrict__ states, Arena arena, int f0, u32* __restrict__ status, int ring_bytes,
                                                     FixedBlob* __restrict__ fixedstore, int far_x, int far_y, int ndc, int dcache_off, u8* __restrict__ hout, int hpitch, int keep_slot, int hbpp) {
  __shared__ __attribute__((aligned(16))) u8 Lraw[HAS_P ? sizeof(WaveLds) : offsetof(WaveLds, fp)];
  WaveLds& L = *(WaveLds*)Lraw;
  // ring_bytes = 4 * (power of two >= W + 512) pixels, then (P-frames) one byte per 16x16 block, then (at dcache_off) ndc dense tables
  extern __shared__ __align__(16) u8 pix[];
  const DecGop gop = gops[blockIdx.x];
  const int lane = lane_id();
  // A batch of key frames for the host (hout): the workgroup has a second wave, the row streamer of its coded key frame (a GOP
  // of this kernel holds one, its first frame; the frames after it can only be flat ones).
  const bool stream_rows = !HAS_P && hout != nullptr && blockDim.x > 64;
  if (!HAS_P && blockDim.x > 64) {
    if (threadIdx.x == 0) L.hs.rows = (stream_rows && gop.count > 0 && frames[gop.first].kind == 0) ? 0u : kRowsEnd, L.hs.pad[0] = 0u;
    __syncthreads();
    if (threadIdx.x >= 64) {
      if (threadIdx.x < 128 && stream_rows) {
        const DecFrame f0 = frames[gop.first];
        row_streamer(&L.hs.rows, &L.hs.pad[0], (const u32*)pix, ring_bytes >> 2, hout + (size_t)f0.slot * (size_t)hpitch * g.H, hpitch, hbpp, g);
      }
      return;
    }
  }
  for (int i = lane; i < CACHE_N; i += 64) L.crec[i][3] = kNoCtx;  // empty cache
  if (threadIdx.x < 64) L.dtag[lane] = 0;
  if (HAS_P) {
    // Helper waves: bulk copies for the chain (wave 0), until it says stop.  A workgroup's waves go round the CU's four SIMDs:
    // wave 4 would sit on the chain's SIMD and take issue slots from it with its polling, so it leaves at once.
    const int nwaves = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6);
    const int nh = nwaves > 4 ? nwaves - 2 : nwaves - 1;
    if (threadIdx.x == 0) L.hc.seq = 0, L.hc.op = 0, L.hc.done = 0, L.hc.nhelp = (u32)nh, L.hc.jtail = 0;
    if (threadIdx.x < 16) L.hc.hprog[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (wv > 0) {
      if (wv != 4) helper_loop(L, g.S, wv < 4 ? wv - 1 : wv - 2, nh);
      return;
    }
  }
  WaveDec D(L, packets, packets_end, states + (size_t)blockIdx.x * NCOLCTX, arena, f0);
  D.has_p = HAS_P;
  if (ndc) {
    D.dcache = pix + dcache_off;
    D.dtag = L.dtag;
    D.dmask = (u32)ndc - 1u;
    D.dc_lds = (u32)(size_t)D.dcache;
  }
  if (gop.load) D.fixed_load(&fixedstore[blockIdx.x]);
  else D.fixed_init();
  for (int fi = gop.first; fi < gop.first + gop.count && !D.bad && !D.oom; fi++) {
    const DecFrame fr = frames[fi];
    u8* dst = planes + (size_t)fr.slot * g.plane_stride;
    if (fr.kind == 0) {
      D.fixed_init();  // RenewI (:418); the colour records of the GOP start cleared
      D.stream_init(packets + fr.src_off + 1);
      // (stream_rows: the picture goes to the host's buffer row by row, by the workgroup's second wave; whatever way the frame
      // ends, the end is announced - with the rows that are complete: all of them, or what a refused stream left)
      // (the plane itself is written only where somebody will read it: keep_slot, the chunk's last frame - what a P-frame of the
      // next call starts from - and every frame that is not streamed)
      const bool streamed = stream_rows && fi == gop.first;
      decode_intra_frame(D, g, dst, (u32*)pix, ring_bytes >> 2, streamed ? &L.hs.rows : nullptr, streamed ? &L.hs.pad[0] : nullptr, !streamed || fr.slot == keep_slot);
      if (streamed) __hip_atomic_store(&L.hs.rows, kRowsEnd | (D.bad ? 0u : (u32)g.H), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (HAS_P && fr.kind == 2) {
      decode_inter_frame(D, g, dst, planes + (size_t)fr.prev_slot * g.plane_stride, packets + fr.src_off, pix + ring_bytes, far_x, far_y);
    }
  }
  if (HAS_P && L.hc.nhelp) help_post(L, HOP_EXIT, nullptr, nullptr, 0);
  D.flush_records();
  D.flush_tabs();
  D.fixed_store(&fixedstore[blockIdx.x]);
  if (D.bad && lane == 0) atomicOr(status, 4u);
#ifdef SCPR_PROFILE
  if (lane == 0)
    for (int i = 0; i < 24; i++) atomicAdd((unsigned long long*)&g_prof[i], (unsigned long long)(i == 15 ? D.dmiss : D.prof[i]));
  if (lane == 0) {
    D.cprof[14] += D.dmiss_ticks, D.cprof[15] += D.dmiss;
    for (int i = 0; i < 16; i++) atomicAdd((unsigned long long*)&g_cprof[i], (unsigned long long)D.cprof[i]);
    for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)&g_cprof[16 + i], (unsigned long long)D.rprof[i]);
    for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)&g_cprof[24 + i], (unsigned long long)D.xprof[i]);
  }
#endif
}

// ------------------------------------------------------------- encoder chains ---
// Non-empty colour chains in four length classes (longest first): the chain kernel walks the classes in
// order with a grid stride, so the long chains - the critical path of the stage - are the first thing the
// waves pick up, one each, and the short ones fill in behind.
constexpr int CHAIN_CLASSES = 4;
__device__ __forceinline__ int chain_class(u32 len) { return len >= 4096 ? 0 : len >= 1024 ? 1 : len >= 192 ? 2 : 3; }
__global__ __launch_bounds__(1024) void k_chain_lists(const u32* __restrict__ cstart, int nchains, u32* __restrict__ lists, u32 cap, u32* __restrict__ counts) {
  __shared__ u32 cnt[CHAIN_CLASSES], base[CHAIN_CLASSES];
  const int q = blockIdx.x * 1024 + threadIdx.x;
  if (threadIdx.x < CHAIN_CLASSES) cnt[threadIdx.x] = 0;
  __syncthreads();
  u32 len = 0;
  if (q < nchains) len = cstart[q + 1] - cstart[q];
  const int which = chain_class(len);
  u32 local = 0;
  if (len) local = atomicAdd(&cnt[which], 1u);  // LDS: position inside this block's share
  __syncthreads();
  if (threadIdx.x < CHAIN_CLASSES) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]) : 0u;  // one global atomic per class per block
  __syncthreads();
  if (len) {
    const u32 idx = base[which] + local;
    if (idx < cap) lists[(size_t)which * cap + idx] = (u32)q;
  }
}

// One wave per colour chain (Context::encode over the symbols of one context in stream
// order, ans_contexts.cpp:34-50).  The whole state of the chain stays in registers:
// header in scalar registers, small table one entry per lane; only the 256-bit symbol
// set (LDS) and dense tables (arena) live in memory.  Entries are produced 64 at a
// time and scattered to their stream positions.
struct ChainPersist {
  const ColState* states_in;  // [NCOLCTX] records of the live generation when the call started (pad1 = generation stamp)
  ColState* states_out;       // where the last generation of this call leaves its records: the same array when the call
                              // has one generation (a chain reads its record, then writes it), another one otherwise:
                              // chains of the first and of the last generation of a context run concurrently
  u32 stamp_in;       // stamp of the live generation when the call started
  u32 stamp_out;      // stamp given to the last generation of this call
  int load_first;     // generation 0 of the call continues the live generation
  int ngens;
};
__global__ __launch_bounds__(64) void k_colour_chain_w(const u32* __restrict__ skeys, const u32* __restrict__ svals, const u32* __restrict__ cstart, const u32* __restrict__ lists,
                                                       const u32* __restrict__ counts, u32 cap, int f0, Arena arena, ChainPersist cp, u32* __restrict__ entries) {
  __shared__ u32 rec[16];
  __shared__ u16 tmp[256];
  __shared__ __attribute__((aligned(16))) DenseTab ltab;  // the chain's dense table while the context is of kind 6 or 7
  WaveModel M(tmp, arena, f0);
  const int lane = M.lane;
  if (rfl(*arena.err) & 32u) return;  // the sorted keys are not sorted (k_chain_starts): the chain starts mean nothing
  u32 cn[CHAIN_CLASSES], n = 0;
#pragma unroll
  for (int k = 0; k < CHAIN_CLASSES; k++) {
    cn[k] = min(rfl(counts[k]), cap);  // (loaded values are wave-uniform: said so, or the loops below are compiled as divergent)
    n += cn[k];
  }
  for (u32 li = blockIdx.x; li < n; li += gridDim.x) {
    u32 i = li, seg = 0;  // position li of the classes laid end to end
#pragma unroll
    for (int k = 0; k < CHAIN_CLASSES - 1; k++)
      if (seg == (u32)k && i >= cn[k]) {
        i -= cn[k];
        seg = k + 1;
      }
    const u32 q = rfl(lists[(size_t)seg * cap + i]);
    const u32 start = rfl(cstart[q]), len = rfl(cstart[q + 1]) - start;
    // the long chains are the critical path of the stage: their waves win the issue arbitration of their SIMD
    if (len >= 1024) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    const int gen = (int)(q / NCOLCTX), ctx = (int)(q - (u32)gen * NCOLCTX);
    ColHdr h = WaveModel::unpack(0, 0, 0);
    u32 T = kSmallNone;
#ifdef SCPR_PROFILE
    const u64 pr_t0 = __builtin_readcyclecounter();
    u32 pr_slow = 0, pr_par = 0, pr_raw = 0, pr_dser = 0, pr_batch = 0;
#endif
    // A dense context (kinds 6/7) keeps its table in LDS for the chain (`ltab`; tlive: it is there).  Its intervals are frozen
    // between rescales and a rescale comes after a known number of symbols (`room`: every symbol adds the same step to the
    // total, ans_contexts.h:686-691, :954-981), so the symbols up to there are INDEPENDENT lookups: one lane each, the counts by
    // LDS atomics, the rebuild by a wave scan (dense_rescale) - the epoch parallelism of the fixed models (k_fixed_chain).
    // Only a symbol the context has not met (kind 6: at most 24 in its life) goes through the general path by itself.
    // (One symbol at a time, a chain of 10^5 symbols through a dense context was the critical path of the stage.)
    bool tlive = false;
    wave_fence();
    if (gen == 0 && cp.load_first) {  // continue the model of this context from the previous call
      const u32* src = (const u32*)&cp.states_in[ctx];
      const u32 w = lane < 16 ? src[lane] : 0;
      if (rdl(w, 3) == cp.stamp_in) {
        if (lane < 16) rec[lane] = w;
        wave_fence();
        h = WaveModel::unpack(rdl(w, 0), rdl(w, 1), rdl(w, 2));
        if (h.kind == 4 || h.kind == 5) {
          M.load_small(rec, h.d, T);
          h.fmax = M.small_fmax(h, T);
          h.top = M.small_top(h, T);
        }
      }
    }
    // (the sorted lists stream from HBM, a microsecond away: the loads of the next two trips are in flight while one is coded)
    u32 k1 = 0, p1 = 0, k2 = 0, p2 = 0;
    if ((u32)lane < len) k1 = skeys[start + lane], p1 = svals[start + lane];
    if (64u + (u32)lane < len) k2 = skeys[start + 64u + lane], p2 = svals[start + 64u + lane];
    for (u32 base = 0; base < len; base += 64) {
      const int m = (int)min(64u, len - base);
      const u32 key = k1, pos = p1;
      u32 mine = 0;
      k1 = k2, p1 = p2;
      if (base + 128u + (u32)lane < len) k2 = skeys[start + base + 128u + lane], p2 = svals[start + base + 128u + lane];
      const int cl = (int)(key & 255u);  // this lane's symbol
      int j = 0;
      while (j < m) {
        if (h.kind >= 6) {
          if (!tlive) {
            wave_fence();
            WaveModel::copy_tab<true>(&ltab, arena.tabs + h.dense, lane);
            tlive = true;
          }
          wave_fence();
          const int step = h.kind == 7 ? kStepDense : kStepHash << h.fshift;
          const bool act = lane >= j && lane < m;
          const bool met = h.kind == 7 || ((rec[4 + (cl >> 5)] >> (cl & 31)) & 1u);
          const u64 um = __ballot(act && !met);
          const int first_unmet = um ? (int)__builtin_ctzll(um) : m;
          const int room = (kProbScale - step - h.total) / step + 1;  // symbols until the rescale (the room-th one brings it on)
          const int take = min(min(first_unmet, m) - j, room);
          if (SCPR_LIKELY(take > 0)) {
            if (lane >= j && lane < j + take) {
              mine = (u32)ltab.freq[cl] | ((u32)ltab.cum[cl] << 16);
              atomicAdd((u32*)&ltab.cnt[cl & ~1], (u32)step << (16 * (cl & 1)));  // (two 16-bit counts per word; a count stays below 4096 + step)
            }
            h.total += take * step;
            j += take;
#ifdef SCPR_PROFILE
            pr_par += (u32)take;
#endif
            if (take == room) {
              M.dense_rescale<true>(&ltab, rec, h);
              WaveModel::scalar_hdr(h);
            }
            continue;
          }
          // symbol j has not been met by this context (or the total leaves no room: never, the rescale keeps it below): by itself
          const int c = (int)rdl((u32)cl, j);
          u32 fr = 0, cf = (u32)c;
          wave_fence();
          M.dense_impl<false, true>(&ltab, rec, h, c, fr, cf);
          WaveModel::scalar_hdr(h);
#ifdef SCPR_PROFILE
          pr_dser++;
#endif
          if (lane == j) mine = (fr & 0xFFFFu) | (cf << 16);
          j++;
          continue;
        }
        // A small table (kinds 4/5) changes with every symbol, but what a HIT changes is plain arithmetic: 50 on the entry's count,
        // on the sums before the entries above it and on the total (SmallContext::encode, ans_contexts.h:195-236).  So the
        // interval of the i-th symbol of a batch of hits is a closed form of the table at the batch's start and of how many
        // earlier symbols of the batch hit the same entry / a lower entry - counts that ballots give every lane at once.  The
        // batch ends before the first symbol that is not such a hit: a symbol the table does not hold, the symbol after which a
        // rescale is due (total + 100 > 4096: after 41 hits at most), or a hit that makes another entry the top one (the spare
        // code space moves with it); that symbol goes through the general path below by itself.  (One symbol at a time a hit
        // costs a lone wave ~470 cycles: chains of 10^5 symbols through contexts with a handful of colours - text on a
        // background - were the critical path of the stage, 40 ms of a one-GOP encode.)
        if ((h.kind | 1) == 5 && m - j >= 3) {
          const int d = h.d, mp = h.maxpos;
          const int i = lane - j;
          const int tot_i = h.total + kStepSmall * i;
          // the lanes of the batch: from j up to the symbol after which the rescale is due (the totals are known beforehand)
          const bool act = lane >= j && lane < m && tot_i + 2 * kStepSmall <= kProbScale;
          int pe = -1;            // this lane's symbol is entry pe of the table (-1: not in it, or not looked at)
          u32 wp = 0, ceq = 0, clt = 0, hmp = 0;  // its packed entry; earlier symbols of the batch on the same entry / on lower entries / on the top entry
          // One turn per DIFFERENT symbol of the batch, in the order they first occur (a batch has two to four as a rule, whatever
          // the size of the table): the lanes that carry it, their rank among themselves, its entry.  A symbol the table does not
          // hold ends the batch at its first lane - every lane before that has been through a turn of its own.
          u64 rem = __ballot(act);
          for (int turn = 0; rem && turn < 8; turn++) {
            const int f = (int)__builtin_ctzll(rem);
            const u32 sy = rdl((u32)cl, f);
            const u32 tm = (u32)__ballot(sm_sym(T) == sy) & ((1u << d) - 1u);  // (the table sits in lanes 0 .. d - 1)
            const bool eq = act && (u32)cl == sy;
            const u64 me = __ballot(eq);
            rem &= ~me;
            if (SCPR_UNLIKELY(!tm)) break;
            const int e = (int)__builtin_ctz(tm);
            const u32 we = rdl(T, e);
            const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(me >> 32), __builtin_amdgcn_mbcnt_lo((u32)me, 0u));
            pe = eq ? e : pe;
            wp = eq ? we : wp;
            ceq = eq ? below : ceq;
            clt += (u32)cl > sy ? below : 0u;  // (entries are sorted by symbol: a lower entry is a smaller symbol)
            hmp = e == mp ? below : hmp;
          }
          const int fq = (int)sm_fq(wp) + kStepSmall * (int)ceq, pp = (int)sm_p(wp) + kStepSmall * (int)clt;
          const int sh = __builtin_clz((u32)(tot_i - 1)) - 20, bonus = (kProbScale >> sh) - tot_i;
          const int ap = (int)sm_sym(wp) + pp - pe + (pe > mp ? bonus : 0), width = fq + (pe == mp ? bonus : 0);
          // the batch ends before the first lane that is not a plain hit: not looked up (not in the table, past the rescale, a ninth
          // symbol), or a hit that makes another entry the top one
          const bool bad = lane >= j && lane < m && (pe < 0 || (pe != mp && fq + kStepSmall > h.fmax + kStepSmall * (int)hmp));
          const u64 bm = __ballot(bad);
          const int cut = bm ? (int)__builtin_ctzll(bm) : m;
          const int take = cut - j;
          if (SCPR_LIKELY(take > 0)) {
            const bool taken = lane >= j && lane < cut;
            if (taken) mine = ((u32)width << sh) | (((u32)ap << sh) << 16);
            // the table after the batch: every entry's count + 50 per symbol that hit it (one turn per different symbol again), the
            // sums before the entries rebuilt
            int addv = 0, nmp = 0;
            for (u64 r2 = __ballot(taken); r2;) {
              const int f = (int)__builtin_ctzll(r2);
              const u64 me = __ballot(taken && cl == (int)rdl((u32)cl, f));
              const int ne = __builtin_popcountll(me), e = (int)rdl((u32)pe, f);
              r2 &= ~me;
              addv = M.l15 == e ? ne : addv;
              nmp = e == mp ? ne : nmp;
            }
            M.small_pack(T, (int)sm_sym(T), (int)sm_fq(T) + kStepSmall * addv, d);
            h.total += kStepSmall * take;
            h.fmax += kStepSmall * nmp;
            h.top = M.small_top(h, T);
            WaveModel::scalar_hdr(h);
            j += take;
#ifdef SCPR_PROFILE
            pr_batch += (u32)take;
#endif
            continue;
          }
        }
        const int c = (int)rdl((u32)cl, j);
        u32 fr = 0, cf = (u32)c;
        wave_fence();
        // plain ifs, the common case first (see the decoder's colour())
        int small = ((h.kind | 1) == 5) ? -1 : 0;
        if (SCPR_LIKELY(small < 0)) {
          int cc, tt = M.top_hit<false>(h, T, c, cc, fr, cf);  // the symbol of the top entry: scalar arithmetic only
          tt = (int)rfl((u32)tt);
          if (SCPR_UNLIKELY(tt >= 0)) {
            M.small_op<false>(rec, h, T, c, fr, cf);
            if ((h.kind | 1) == 5) h.top = M.small_top(h, T);
#ifdef SCPR_PROFILE
            pr_slow++;
#endif
          }
        }
        small = (int)rfl((u32)small);  // (keeps the two tests apart)
        if (SCPR_UNLIKELY(small >= 0)) {  // kinds 0-3: the symbol goes out raw (and may make the context a table)
#ifdef SCPR_PROFILE
          pr_raw++;
#endif
          M.note_raw(rec, h, c, T);
          WaveModel::scalar_hdr(h);
          if ((h.kind | 1) == 5) h.top = M.small_top(h, T);
        }
        if (lane == j) mine = (fr & 0xFFFFu) | (cf << 16);
        j++;
      }
      if (lane < m) entries[pos] = mine;
    }
#ifdef SCPR_PROFILE
    if (len >= 2048u && lane == 0) {
      const u32 k = atomicAdd(&g_chainrec_n, 1u);
      if (k < 8192u) {
        u32* o = g_chainrec[k];
        o[0] = len, o[1] = (u32)(__builtin_readcyclecounter() - pr_t0), o[2] = (u32)h.kind | ((u32)h.d << 8), o[3] = pr_slow, o[4] = pr_par, o[5] = pr_raw, o[6] = pr_dser | (pr_batch << 8), o[7] = q;
      }
    }
#endif
    if (gen == cp.ngens - 1) {  // live generation: keep the state for the next call
      wave_fence();
      if (tlive) WaveModel::copy_tab<false>(arena.tabs + h.dense, &ltab, lane);
      if (h.kind == 4 || h.kind == 5) M.store_small(rec, h.d, T);
      M.store_header(rec, h);
      if (lane == 0) rec[3] = cp.stamp_out;
      wave_fence();
      if (lane < 16) ((u32*)&cp.states_out[ctx])[lane] = rec[lane];
    }
  }
}

}  // namespace scpr
