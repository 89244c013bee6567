// Host side of the MI355X-native ScreenPressor path: the ScreenCodec-shaped
// state machine (screencap.cpp:1456-1743 in the reference tree), workspace
// management, kernel sequencing on one HIP stream, and the C ABI declared in
// include/scpr_amd.h.  GPU only: there is no CPU code path for the codec.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/scpr_amd.h"
#include "scpr_kernels.hpp"
#include "scpr_wave.hpp"
#include "scpr_rans_s.hpp"
#include "scpr_fixed.hpp"
#include "scpr_ctxsort.hpp"
#include "scpr_v2.hpp"
#include "scpr_inter.hpp"

using namespace scpr;

#define HIPCHK(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "[scpr] HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
      return SCPR_E_DEVICE;                                                                    \
    }                                                                                          \
  } while (0)

namespace {

static bool debug_alloc() {  // SCPR_DEBUG_ALLOC=1: every device allocation is reported (design aid: a steady state must have none)
  static const bool on = getenv("SCPR_DEBUG_ALLOC") != nullptr;
  return on;
}
struct DevBuf {  // grow-only device allocation
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (debug_alloc()) fprintf(stderr, "[scpr alloc] reserve %zu (had %zu)\n", bytes, cap);
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  hipError_t reserve_keep(size_t bytes, size_t keep, hipStream_t st) {  // grow, preserving the first `keep` bytes
    if (bytes <= cap) return hipSuccess;
    if (debug_alloc()) fprintf(stderr, "[scpr alloc] reserve_keep %zu (had %zu, keeps %zu)\n", bytes, cap, keep);
    void* np = nullptr;
    size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(np, p, std::min(keep, cap), hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

enum Stage { ST_PACK, ST_CLASSIFY, ST_INTER, ST_SCAN, ST_SYMBOLS, ST_SORT, ST_FIXED, ST_COLOUR, ST_RANS, ST_GATHER, ST_DECODE, ST_UNPACK, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"pack", "classify", "inter", "scan", "symbols", "sort", "fixed_chain", "colour_chain", "rans", "gather", "decode", "unpack"};

}  // namespace

struct FixBufs {  // scratch of fixed_chains(): block counts and offsets, model totals, the partitioned lists, generation starts
  DevBuf cnt, off, tot, sym, pos, gen;
};
struct scpr_codec {
  int device = 0;
  hipStream_t stream = nullptr;
  bool inited = false, have_codec = false, crashed = false;
  scpr_params prm{};
  Geom g{};
  int bpp = 0, pitch_in = 0, version = 4, f0 = 32;
  int rs = 0, gs = 0, bs = 0, last_loss = 0;
  u32 loss_mask = 0xFFFFFFFFu, corr_mask = 0;
  // CScreenCapt state carried between frames
  u32 frames_done = 0;  // fn
  bool last_flat = false;
  u32 last_flat_rgb = 0;
  int slots = 0;   // frames per encode chunk (bounded by the per-frame scratch the encoder needs)
  int dslots = 0;  // frames per decode chunk (the decoder needs planes only: a long stream keeps all its GOPs in flight)
  size_t plane_slots = 0;  // planes allocated for frames; the previous frame of the stream lives in slot `pslot` == plane_slots
  u32 planes_stride = 0;   // geometry the planes were last cleared for (plane stride, row stride, picture size)
  int planes_S = 0, planes_W = 0, planes_H = 0;
  int pslot = 0;
  // per-slot worst-case buffers
  DevBuf planes, exitmap, entry, runrec, tilecnt, tileoff, hdrrec, hdrcnt, frametot, tnmap;
  size_t tn_half = 0;
  // per-batch buffers
  DevBuf flags, slotlist, genlist, bases, totals, runs, runpos, keys[2], vals[2], cstart, csblocks, cssegs, cscnt, csoff, entries, ranges;
  DevBuf rblocks, rscratch, rrec, rsize, packets, pktoff, blkdst, outsizes, total64, arena, arena2, arena_top, err, rcp;  // (arena2: where compact_tables moves the live tables)
  DevBuf decframes, decstates, hoststage_in, hoststage_out, chainlists, chaincounts;
  FixBufs fixr, fixm;  // run list / P-frame symbol list
  // P-frame buffers
  DevBuf mvdict, mvpre, gmask;
  DevBuf kinds, pidx, fidx, pframes, pflag, binfo, smv, btype, bmv, bcnt, boff, bflag, pinfo, ptot, pbase, misc, miscpos, miscranges, plist;
  // state of the live generation, carried between calls (models are reset only by key frames, screencap.cpp:1118)
  DevBuf mvs, mvs_keep, fixed_persist, misc_persist, colour_persist;
  bool live_valid = false;   // a generation is live (a key frame or flat frame has been coded)
  bool live_has_state = false;  // ... and it has coded symbols (a flat frame renews the models without coding any)
  u32 live_stamp = 0, next_stamp = 1;
  int live_buf = 0;  // which half of fixed_persist / misc_persist / colour_persist holds the live generation
  size_t arena_used_bound = 0;  // dense tables held by the live generation (the arena top read back after every call)
  size_t live_symbols = 0;      // colour symbols coded so far in the live generation (a bound on the tables it can own)
  // decoder side of the same
  DevBuf decgops, decfixed, dec_fixed_persist, dec_colour_persist, dec_arena, dec_arena2, dec_arena_top;  // (its own dense-table arena: one codec may compress and decompress)
  bool dec_live = false;
  u32 h_dec_top0 = 1;
  size_t dec_arena_used = 1;  // tables held by the live GOP of the decoder (table 0 is the sink of an overflowing run, never a real table)
  u64 dec_live_bytes = 0;     // packet bytes of the live GOP so far (a bound on the tables it can own)
  // second stream: the fixed-model chains run beside the colour chains (they write disjoint entries)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<GenRange> h_ranges;      // host images of the per-generation ranges (uploaded asynchronously: they must outlive the call)
  std::vector<MiscRange> h_miscranges;
  // Pinned staging for the small host <-> device copies of the batch calls (frame lists, totals, sizes, error words).  A copy
  // between the device and PAGEABLE host memory is carried out inside the runtime, under its own locks and waits: measured with
  // two codecs in two host threads, one codec's 8-byte read-back did not return until the OTHER codec's 118 ms kernel had finished
  // (rocprofv3 --hip-trace).  Through pinned memory a copy is a DMA on the codec's own stream and a wait is a
  // wait for that stream.  h2d(): the source is copied here first (it may die right after the call).  d2h(): lands here, and
  // sync_out() hands it to where it was meant to go after the stream has been waited for.  A request the pool cannot hold
  // falls back to the direct copy.
  int ncu = 256;  // compute units the codec's kernels may run on (the device's, or the bits of its CU mask)
  u8* pin = nullptr;
  size_t pin_cap = 0, pin_used = 0;
  struct PendingOut {
    void* dst;
    const void* src;
    size_t bytes;
  };
  std::vector<PendingOut> pin_out;
  // timing
  hipEvent_t ev[ST_COUNT + 1][2];
  bool ev_used[ST_COUNT];
  float stage_ms[ST_COUNT];
  float total_ms = 0;
  int64_t dbg_entries = 0;
  // What a compress call changes, kept so that a call whose packets turn out not to fit the caller's buffer leaves the codec as it
  // found it (SCPR_E_CAPACITY; the reference's own guard, CheckDstLength, screencap.cpp:300-314, is commented out and its caller
  // provides W*H*6 bytes).  The vector memory is copied at every call's start (a few KB); the rest - models, the live generation's
  // dense tables, the previous frame - only when a chunk's packets MAY not fit (2 bytes per coder entry + 4 per block + headers
  // against the room left), or at the start of a call of several frames whose buffer is below the closed-form worst case.
  DevBuf snap_mvs, snap_state;
  bool snap_taken = false;
  size_t snap_tables = 0;
  int dbg_inject = 0;  // scpr_debug_inject: tests make the next compress call fail at a chosen place
  bool dbg_armed = false;  // ... only in a codec created with SCPR_ENABLE_DEBUG_INJECT=1 in the environment
  int rans_scalar_max = 0;  // blocks per call up to which the coder is k_rans_s (SCPR_RANS_SCALAR_MAX at creation; 0 once its self-check has failed)
  int rans_recoded = 0;     // calls whose blocks were coded a second time (k_rans) because k_rans_s' self-check spoke
  // the host-pointer batch calls (scpr_compress_batch_host / scpr_decompress_batch_host): a stream of its own for the copies that
  // run beside the kernels, staging on the device for two sub-batches of frames, and the host ranges this codec has registered
  // with the runtime (hipHostRegister: pinned and mapped into the device's address space; released by scpr_destroy)
  hipStream_t stream3 = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr};
  DevBuf hb_frames, hb_packets, hb_list;
  struct HostRange {
    void* base;
    size_t bytes;
    bool owned;  // registered by this codec (another codec, or the caller, may have pinned the same memory first)
  };
  std::vector<HostRange> host_ranges;
};
struct EncHostState {  // the host half of that
  u32 frames_done, last_flat_rgb, live_stamp, next_stamp;
  bool last_flat, live_valid, live_has_state;
  int live_buf;
  size_t arena_used_bound, live_symbols;
};
static EncHostState host_state(const scpr_codec* c) {
  return {c->frames_done, c->last_flat_rgb, c->live_stamp, c->next_stamp, c->last_flat, c->live_valid, c->live_has_state, c->live_buf, c->arena_used_bound, c->live_symbols};
}
static void set_host_state(scpr_codec* c, const EncHostState& h) {
  c->frames_done = h.frames_done, c->last_flat_rgb = h.last_flat_rgb, c->live_stamp = h.live_stamp, c->next_stamp = h.next_stamp, c->last_flat = h.last_flat;
  c->live_valid = h.live_valid, c->live_has_state = h.live_has_state, c->live_buf = h.live_buf, c->arena_used_bound = h.arena_used_bound, c->live_symbols = h.live_symbols;
}

static void stage_begin(scpr_codec* c, int s, hipStream_t on = nullptr) {
  if (!c->ev_used[s]) (void)hipEventRecord(c->ev[s][0], on ? on : c->stream);
}
static void stage_end(scpr_codec* c, int s, hipStream_t on = nullptr) {
  (void)hipEventRecord(c->ev[s][1], on ? on : c->stream);
  c->ev_used[s] = true;
}
static void timing_reset(scpr_codec* c) {
  for (int s = 0; s < ST_COUNT; s++) {
    c->ev_used[s] = false;
    c->stage_ms[s] = 0;
  }
  c->total_ms = 0;
}
// events bracket the first..last launch of a stage; called after a stream sync
static void timing_collect(scpr_codec* c) {
  for (int s = 0; s < ST_COUNT; s++)
    if (c->ev_used[s]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev[s][0], c->ev[s][1]) == hipSuccess) {
        c->stage_ms[s] += ms;
        c->total_ms += ms;
      }
      c->ev_used[s] = false;
    }
}

static void* pin_take(scpr_codec* c, size_t bytes) {
  if (!c->pin) {
    static const size_t cap = getenv("SCPR_PIN_BYTES") ? (size_t)strtoull(getenv("SCPR_PIN_BYTES"), nullptr, 0) : (size_t)8 << 20;
    if (cap == 0 || hipHostMalloc((void**)&c->pin, cap, hipHostMallocDefault) != hipSuccess) {
      c->pin = nullptr;
      return nullptr;
    }
    c->pin_cap = cap;
  }
  const size_t at = (c->pin_used + 63) & ~(size_t)63;
  if (at + bytes > c->pin_cap) return nullptr;
  c->pin_used = at + bytes;
  return c->pin + at;
}
static hipError_t h2d(scpr_codec* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  void* q = st == c->stream ? pin_take(c, bytes) : nullptr;
  if (!q) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  memcpy(q, src, bytes);
  return hipMemcpyAsync(dst, q, bytes, hipMemcpyHostToDevice, st);
}
static hipError_t d2h(scpr_codec* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  void* q = st == c->stream ? pin_take(c, bytes) : nullptr;
  if (!q) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  c->pin_out.push_back({dst, q, bytes});
  return hipMemcpyAsync(q, src, bytes, hipMemcpyDeviceToHost, st);
}
// waits for the stream; what d2h() has fetched goes to its destinations, and the pool is free again (every copy queued on the
// codec's stream has been carried out)
static hipError_t sync_out(scpr_codec* c, hipStream_t st) {
  hipError_t e = hipStreamSynchronize(st);
  if (st == c->stream) {
    if (e == hipSuccess)
      for (const auto& o : c->pin_out) memcpy(o.dst, o.src, o.bytes);
    c->pin_out.clear();
    c->pin_used = 0;
  }
  return e;
}

// at the start of an entry point: whatever a call that ended in an error left pending is dropped (its destinations are gone)
static void pin_reset(scpr_codec* c) {
  if (!c->pin_out.empty() || c->pin_used) (void)hipStreamSynchronize(c->stream);
  c->pin_out.clear();
  c->pin_used = 0;
}

static void setup_loss(scpr_codec* c, int loss) {  // SetupLossMask, screencap.cpp:127-139
  u32 mask = 0;
  for (int i = 0; i < loss; i++) mask = (mask << 1) | 1;
  mask = (mask << 8) + mask;
  mask = (mask << 16) + mask;
  c->loss_mask = ~mask;
  u32 cm = (1u << loss) >> 1;
  cm = (cm << 8) + cm;
  c->corr_mask = (cm << 16) + cm;
  c->last_loss = loss;
}

// Planes for the frames of a chunk (slots 0..n-1) and, behind them, the previous frame of the stream (slot pslot): grown when a
// call needs more, the previous frame carried over - a codec used one frame at a time holds two planes, not a batch's worth.
static int ensure_planes(scpr_codec* c, size_t n) {
  if (n <= c->plane_slots && c->planes.p) return SCPR_OK;
  const Geom& g = c->g;
  const size_t most = (size_t)std::max(c->slots, c->dslots);
  const size_t want = std::max(n, std::min(most, 2 * c->plane_slots));
  void* np = nullptr;
  HIPCHK(hipMalloc(&np, (want + 1) * (size_t)g.plane_stride));
  HIPCHK(hipMemsetAsync(np, 0, (want + 1) * (size_t)g.plane_stride, c->stream));  // (row padding stays zero: RGB24 output copies whole rows)
  if (c->planes.p)
    HIPCHK(hipMemcpyAsync((u8*)np + want * (size_t)g.plane_stride, c->planes.as<u8>() + c->plane_slots * (size_t)g.plane_stride, g.plane_stride, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(sync_out(c, c->stream));
  c->planes.release();
  c->planes.p = np;
  c->planes.cap = (want + 1) * (size_t)g.plane_stride;
  c->plane_slots = want;
  c->pslot = (int)want;
  return SCPR_OK;
}
// the encoder's per-frame scratch for a chunk of n frames (grow-only; a codec that only decodes never allocates it)
static int ensure_enc_scratch(scpr_codec* c, size_t n) {
  const Geom& g = c->g;
  const size_t ns = n + 1;
  HIPCHK(c->exitmap.reserve(ns * g.ntiles * 512));
  c->tn_half = ns * g.ntiles * TILE;  // type | length of the run that would start at a pixel: 16 bits per pixel
  HIPCHK(c->tnmap.reserve(c->tn_half * 2));
  HIPCHK(c->entry.reserve(ns * g.ntiles * 2));
  HIPCHK(c->runrec.reserve(ns * g.ntiles * TILE * 4));
  HIPCHK(c->tilecnt.reserve(ns * g.ntiles * 8));
  HIPCHK(c->tileoff.reserve(ns * g.ntiles * 8));
  HIPCHK(c->hdrrec.reserve(ns * (g.W + 2) * 4));
  HIPCHK(c->hdrcnt.reserve(ns * 4));
  HIPCHK(c->frametot.reserve(ns * 8));
  HIPCHK(c->flags.reserve(ns * 8));
  HIPCHK(c->slotlist.reserve(ns * 4));
  HIPCHK(c->genlist.reserve(ns * 4));
  HIPCHK(c->bases.reserve(ns * sizeof(FrameBase)));
  HIPCHK(c->ranges.reserve(ns * sizeof(GenRange)));
  HIPCHK(c->kinds.reserve(ns * 4));
  HIPCHK(c->pidx.reserve(ns * 4));
  HIPCHK(c->fidx.reserve(ns * 4));
  HIPCHK(c->pframes.reserve(ns * sizeof(PFrame)));
  HIPCHK(c->pflag.reserve(ns * 4));
  HIPCHK(c->pinfo.reserve(ns * 8));
  HIPCHK(c->ptot.reserve(ns * 32));
  HIPCHK(c->pbase.reserve(ns * sizeof(PBase)));
  HIPCHK(c->miscranges.reserve(ns * sizeof(MiscRange)));
  return SCPR_OK;
}

static int ensure_codec(scpr_codec* c, int version) {  // CreateCodec + CScreenCapt::Init, screencap.cpp:1587-1617, :69-124
  if (c->have_codec) return SCPR_OK;
  if (version < 2 || version > 4) return SCPR_E_BAD_VERSION;  // BadVersionException (:1589-1590); version 2 is decode-only here
  const scpr_params& p = c->prm;
  if (p.bits_per_pixel != 16 && p.bits_per_pixel != 24 && p.bits_per_pixel != 32) return SCPR_E_BAD_VERSION;
  // (width: the LDS row ring of the decoder; height: motion-block jobs carry x and y in 13 bits each, scpr_wave.hpp)
  if (p.width < 3 || p.height < 2 || p.width > 8000 || p.height > 8191 || p.workers < 1 || p.height < 2 * p.workers) return SCPR_E_PARAM;
  // a near window wider than the far one makes the reference code motion symbols below zero (mv + msr_x with |mv| up to msrlow_x,
  // screencap.cpp:691-704, :1206): outside the format
  if (p.low_range_x > std::min<u32>(p.high_range_x, 256) || p.low_range_y > std::min<u32>(p.high_range_y, 256)) return SCPR_E_PARAM;
  if (version == 2 && (p.high_range_x > 256 || p.high_range_y > 256 || !p.high_range_x || !p.high_range_y)) return SCPR_E_PARAM;  // its motion tables hold 2 * range symbols (LDS: 512)
  c->version = version;
  c->f0 = version == 3 ? 64 : 32;
  Geom& g = c->g;
  g.W = (int)p.width;
  g.H = (int)p.height;
  g.S = (g.W * 3 + 3) & ~3;
  g.NP = g.W * g.H;
  g.p0 = g.W + 1;
  g.ntiles = (g.NP - g.p0 + TILE - 1) / TILE;
  if (g.ntiles < 1) g.ntiles = 1;
  g.workers = (int)p.workers;
  g.plane_stride = (u32)(((size_t)g.H * g.S + 16 + 255) & ~(size_t)255);
  // encode chunk size: keep the per-slot worst-case buffers within ~24 GiB (they are allocated for the chunks that come: ensure_enc_scratch)
  size_t per_slot = (size_t)g.plane_stride + (size_t)g.ntiles * (512 + 2 + TILE * 6 + 16) + (size_t)(g.W + 2) * 4;
  size_t s = (24ull << 30) / per_slot;
  c->slots = (int)std::min<size_t>(std::max<size_t>(s, 1), 768);  // 768 = three decoder workgroups on each of the 256 CUs
  // decode chunk size: planes only, up to ~32 GiB of them - what a chunk holds is what runs side by side, and a GOP is one chain
  c->dslots = (int)std::min<size_t>(std::max<size_t>((32ull << 30) / g.plane_stride, 1), 4096);
  if (c->planes.p && c->planes.cap >= 2 * (size_t)g.plane_stride) {  // the planes of an earlier Init are kept (Deinit / Init per stream is cheap)
    c->plane_slots = std::min<size_t>(c->planes.cap / g.plane_stride - 1, (size_t)std::max(c->slots, c->dslots));
    c->pslot = (int)c->plane_slots;
    // Only the previous-frame slot is cleared: a stream starts with a key frame, which reads no previous frame, and every frame
    // written to a slot (pack kernels, decoder) writes its rows whole, padding included - clearing all planes was 1.9 GB per Init
    // for a codec that had held 300 frames of 1080p.  A new geometry clears them all: the slack between rows and planes sits
    // elsewhere.
    // (the SAME picture size: equal strides are shared by other sizes - W = 3 and 4 both have S = 12 - whose row padding, trailing
    // rows and the slack behind a plane then hold the old picture's pixels)
    const bool same_geom = c->planes_stride == g.plane_stride && c->planes_S == g.S && c->planes_W == g.W && c->planes_H == g.H;
    HIPCHK(hipMemsetAsync(c->planes.as<u8>() + (same_geom ? c->plane_slots * (size_t)g.plane_stride : 0), 0,
                          (same_geom ? 1 : c->plane_slots + 1) * (size_t)g.plane_stride, c->stream));
  } else {
    c->planes.release();
    c->plane_slots = 0;
    c->pslot = 0;
    int rc = ensure_planes(c, 1);
    if (rc != SCPR_OK) return rc;
  }
  HIPCHK(c->totals.reserve(64));
  {
    const size_t nblk = (size_t)((g.W + 15) / 16) * ((g.H + 15) / 16);
    HIPCHK(c->mvs.reserve(nblk * 4));
    HIPCHK(hipMemsetAsync(c->mvs.p, 0, nblk * 4, c->stream));  // calloc'd in the reference (screencap.cpp:96-97), never reset
    // two copies of everything the chains keep between calls (see k_fixed_chain: a call with several generations reads one, writes the other)
    HIPCHK(c->fixed_persist.reserve(2 * 12 * sizeof(FixedPersist)));
    HIPCHK(c->misc_persist.reserve(2 * MC_COUNT * sizeof(FixedPersist)));
    HIPCHK(hipMemsetAsync(c->fixed_persist.p, 0, 2 * 12 * sizeof(FixedPersist), c->stream));
    HIPCHK(hipMemsetAsync(c->misc_persist.p, 0, 2 * MC_COUNT * sizeof(FixedPersist), c->stream));
    HIPCHK(c->colour_persist.reserve(2 * (size_t)NCOLCTX * sizeof(ColState)));
    HIPCHK(hipMemsetAsync(c->colour_persist.p, 0, 2 * (size_t)NCOLCTX * sizeof(ColState), c->stream));
  }
  c->live_valid = false;
  c->dec_live = false;
  c->live_has_state = false;
  c->live_stamp = 0;
  c->next_stamp = 1;
  c->live_buf = 0;
  c->arena_used_bound = 0;
  c->live_symbols = 0;
  HIPCHK(c->arena_top.reserve(16));
  HIPCHK(c->dec_arena_top.reserve(16));
  c->dec_arena_used = 1;
  c->dec_live_bytes = 0;
  HIPCHK(c->err.reserve(64));
  HIPCHK(c->total64.reserve(16));
  // reciprocal table for every frequency on the 12-bit scale
  std::vector<RansRcp> tab(kProbScale + 1);
  for (u32 f = 0; f <= (u32)kProbScale; f++) tab[f] = rans_rcp(f ? f : 1);
  HIPCHK(c->rcp.reserve(tab.size() * sizeof(RansRcp)));
  HIPCHK(h2d(c, c->rcp.p, tab.data(), tab.size() * sizeof(RansRcp), c->stream));
  HIPCHK(sync_out(c, c->stream));
  setup_loss(c, (int)p.loss);
  c->planes_stride = g.plane_stride;
  c->planes_S = g.S;
  c->planes_W = g.W;
  c->planes_H = g.H;
  c->have_codec = true;
  return SCPR_OK;
}

// ---------------------------------------------------------------------------
// One chunk of frames (all resident as RGB24 planes in slots 0..n-1).
// ---------------------------------------------------------------------------
struct ChunkFrame {
  int kind;      // 0 coded key frame, 1 flat key frame, 2 P-frame
  int gen;       // model generation inside the chunk
  u32 hdr, hdr_len;
};

// The tables of the records in `recs` to the bottom of the arena, by way of a second buffer (k_compact_tables); `first` = the
// index the first one gets (the decoder keeps table 0 out of use).  Returns the new top through *top.
static int compact_tables(scpr_codec* c, DevBuf& arena, DevBuf& other, DevBuf& topbuf, void* recs, int words, int stamp_word, u32 stamp, u32 first, size_t old_top, u32* top) {
  hipStream_t st = c->stream;
  const size_t most = std::min<size_t>(old_top, (size_t)NCOLCTX + first) + 64;
  HIPCHK(other.reserve(most * sizeof(DenseTab)));
  HIPCHK(h2d(c, topbuf.p, &first, 4, st));
  hipLaunchKernelGGL(k_compact_tables, dim3(NCOLCTX), dim3(64), 0, st, (u32*)recs, words, (int)NCOLCTX, stamp_word, stamp, arena.as<DenseTab>(), other.as<DenseTab>(), topbuf.as<u32>());
  HIPCHK(d2h(c, top, topbuf.p, 4, st));
  HIPCHK(sync_out(c, st));  // (`first` and `top` are the caller's)
  // ... and back to the bottom of the arena itself (at most 19 MB; swapping the buffers instead would leave the small one as
  // the arena and make the next call allocate gigabytes again)
  if (*top > first) HIPCHK(hipMemcpyAsync(arena.as<DenseTab>() + first, other.as<DenseTab>() + first, (size_t)(*top - first) * sizeof(DenseTab), hipMemcpyDeviceToDevice, st));
  return SCPR_OK;
}
// ---- a compress call that can be taken back (see scpr_codec::snap_state) --------------------------------------------------
static size_t snap_fixed_bytes() { return 2 * 12 * sizeof(FixedPersist); }
static size_t snap_misc_bytes() { return 2 * MC_COUNT * sizeof(FixedPersist); }
static size_t snap_colour_bytes() { return 2 * (size_t)NCOLCTX * sizeof(ColState); }
// the device half: models (both copies), the live generation's dense tables and the arena's top, the previous frame of the stream.
// `hs` = the host state to go with it (the tables it counts are the ones copied).  Nothing of it has been touched by the call so
// far when this runs: the chains, the arena and the previous-frame slot are only written after a chunk's totals are known.
static int snap_take(scpr_codec* c, const EncHostState& hs) {
  if (c->snap_taken) return SCPR_OK;
  hipStream_t st = c->stream;
  const size_t fb = snap_fixed_bytes(), mb = snap_misc_bytes(), cb = snap_colour_bytes(), pb = c->g.plane_stride;
  const size_t tabs = c->arena.p ? std::min<size_t>(hs.arena_used_bound, c->arena.cap / sizeof(DenseTab)) : 0;
  HIPCHK(c->snap_state.reserve(fb + mb + cb + 16 + pb + tabs * sizeof(DenseTab)));
  u8* q = c->snap_state.as<u8>();
  HIPCHK(hipMemcpyAsync(q, c->fixed_persist.p, fb, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(q + fb, c->misc_persist.p, mb, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(q + fb + mb, c->colour_persist.p, cb, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(q + fb + mb + cb, c->arena_top.p, 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(q + fb + mb + cb + 16, c->planes.as<u8>() + (size_t)c->pslot * pb, pb, hipMemcpyDeviceToDevice, st));
  if (tabs) HIPCHK(hipMemcpyAsync(q + fb + mb + cb + 16 + pb, c->arena.p, tabs * sizeof(DenseTab), hipMemcpyDeviceToDevice, st));
  c->snap_tables = tabs;
  c->snap_taken = true;
  return SCPR_OK;
}
static int snap_restore(scpr_codec* c, const EncHostState& hs) {
  hipStream_t st = c->stream;
  const int nblk = ((c->g.W + 15) / 16) * ((c->g.H + 15) / 16);
  HIPCHK(hipStreamSynchronize(c->stream2));
  HIPCHK(hipMemcpyAsync(c->mvs.p, c->snap_mvs.p, (size_t)nblk * 4, hipMemcpyDeviceToDevice, st));
  if (c->snap_taken) {
    const size_t fb = snap_fixed_bytes(), mb = snap_misc_bytes(), cb = snap_colour_bytes(), pb = c->g.plane_stride;
    const u8* q = c->snap_state.as<u8>();
    HIPCHK(hipMemcpyAsync(c->fixed_persist.p, q, fb, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->misc_persist.p, q + fb, mb, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->colour_persist.p, q + fb + mb, cb, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->arena_top.p, q + fb + mb + cb, 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->planes.as<u8>() + (size_t)c->pslot * pb, q + fb + mb + cb + 16, pb, hipMemcpyDeviceToDevice, st));  // (ensure_planes carries this slot along when the planes grow)
    if (c->snap_tables) HIPCHK(hipMemcpyAsync(c->arena.p, q + fb + mb + cb + 16 + pb, c->snap_tables * sizeof(DenseTab), hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(sync_out(c, st));
  set_host_state(c, hs);
  return SCPR_OK;
}
// the most a frame's packet can take: header, at most 2 bytes per coder entry (a 12-bit interval moves the 31-bit state by at most
// 12 bits, bytes leave it 8 at a time: ransmt.h:120 sizes its scratch the same way), 4 bytes of state per block of 131072 entries
static u64 packet_bound(u32 hdr_len, u64 nsyms) { return (u64)hdr_len + 2 * nsyms + 4 * ((nsyms + kBlockEntries - 1) / kBlockEntries); }

// Block compare and motion search of the P-frames `pfr` of a chunk (DecideBlockTypes / FindMV, screencap.cpp:928-1087, :684-814):
// leaves binfo / btype / bmv / pinfo for the symbol stages and the vector memory mvs[] as the reference leaves it.  Shared by
// the encoder and by scpr_motion_prepass, which needs nothing else of a P-frame.
static int motion_stage(scpr_codec* c, const std::vector<PFrame>& pfr) {
  const Geom& g = c->g;
  hipStream_t st = c->stream;
  const u8* planes = c->planes.as<u8>();
  const int nbx = (g.W + 15) / 16, nby = (g.H + 15) / 16, nblocks = nbx * nby, np = (int)pfr.size();
  const MvParams mp{(int)std::min<u32>(c->prm.high_range_x, 256), (int)std::min<u32>(c->prm.high_range_y, 256), (int)c->prm.low_range_x, (int)c->prm.low_range_y};
  const size_t pb = (size_t)np * nblocks;
  HIPCHK(c->binfo.reserve(pb * 4));
  HIPCHK(c->smv.reserve(pb * 4));
  HIPCHK(c->btype.reserve(pb));
  HIPCHK(c->bmv.reserve(pb * 4));
  HIPCHK(h2d(c, c->pframes.p, pfr.data(), np * sizeof(PFrame), st));
  HIPCHK(hipMemsetAsync(c->pflag.p, 0, (size_t)np * 4, st));
  const int grp = std::min(64, nbx), ngrp = (nblocks + grp - 1) / grp;
  HIPCHK(c->gmask.reserve((size_t)np * ngrp * 8));
  HIPCHK(hipMemsetAsync(c->gmask.p, 0, (size_t)np * ngrp * 8, st));
  HIPCHK(hipMemsetAsync(c->btype.p, 0, pb, st));       // untouched groups are not visited by k_mvresolve
  HIPCHK(hipMemsetAsync(c->bmv.p, 0, pb * 4, st));
  hipLaunchKernelGGL(k_pblocks, dim3((nblocks + 3) / 4, np), dim3(64), 0, st, planes, g, c->pframes.as<PFrame>(), c->binfo.as<u32>(), c->pflag.as<u32>(),
                     c->gmask.as<unsigned long long>());
  hipLaunchKernelGGL(k_mvsearch, dim3(nblocks, np), dim3(64), 0, st, planes, g, c->pframes.as<PFrame>(), c->binfo.as<u32>(), mp, c->smv.as<u32>());
  HIPCHK(c->mvdict.reserve((size_t)np * MVDICT * 4));
  HIPCHK(c->mvpre.reserve(pb * 4));
  hipLaunchKernelGGL(k_mvdict, dim3(np), dim3(256), 0, st, g, c->binfo.as<u32>(), c->smv.as<u32>(), c->mvdict.as<u32>());
  hipLaunchKernelGGL(k_mvpretest, dim3((nblocks + 3) / 4, np), dim3(64), 0, st, planes, g, c->pframes.as<PFrame>(), c->binfo.as<u32>(), mp, c->mvdict.as<u32>(), c->mvpre.as<u32>());
  if ((size_t)nblocks * 4 <= 150 * 1024 && nblocks + nbx < 0xFFFF) {  // frames pipelined over sixteen waves, the vector memory in LDS
    const size_t lds = (size_t)nblocks * 4;
    HIPCHK(hipFuncSetAttribute((const void*)k_mvresolve_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mvresolve_pipe, dim3(1), dim3(64 * MVP_WAVES), lds, st, planes, g, c->pframes.as<PFrame>(), np, c->binfo.as<u32>(), c->smv.as<u32>(), c->mvdict.as<u32>(),
                       c->mvpre.as<u32>(), mp, c->mvs.as<u32>(), c->btype.as<u8>(), c->bmv.as<u32>(), c->pinfo.as<int>(), c->gmask.as<unsigned long long>(), c->err.as<u32>());
  } else {
    const size_t lds = (size_t)nblocks * 16;  // the frame's block arrays + the vector memory in LDS when they fit
    const int use_lds = lds <= 150 * 1024 ? 1 : 0;
    if (use_lds) HIPCHK(hipFuncSetAttribute((const void*)k_mvresolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mvresolve, dim3(1), dim3(256), use_lds ? lds : 0, st, planes, g, c->pframes.as<PFrame>(), np, c->binfo.as<u32>(), c->smv.as<u32>(), c->mvdict.as<u32>(),
                       c->mvpre.as<u32>(), mp, c->mvs.as<u32>(), c->btype.as<u8>(), c->bmv.as<u32>(), c->pinfo.as<int>(), c->gmask.as<unsigned long long>(), use_lds);
  }
  return SCPR_OK;
}
// The fixed-alphabet chains over one list (scpr_fixed.hpp): partition by model, then one workgroup per generation, one wave per model.
template <class SRC, int MAXSYM>
static int fixed_chains(scpr_codec* c, hipStream_t s2, FixBufs& fb, const u32* el, const u32* elpos, size_t n, const uint2* ranges, int ngens, bool load_first,
                        const FixedPersist* pin, FixedPersist* pout, int phase = 0) {  // phase 1: the partition only, 2: the chains only (0: both)
  constexpr int NC = SRC::NCLS;
  const u32 nblk = (u32)((n + PART_B - 1) / PART_B);
  HIPCHK(fb.cnt.reserve((size_t)NC * (nblk + 1) * 4));
  HIPCHK(fb.off.reserve((size_t)NC * (nblk + 1) * 4));
  HIPCHK(fb.tot.reserve(64));
  HIPCHK(fb.sym.reserve((2 * n + 64) * 2));
  HIPCHK(fb.pos.reserve((2 * n + 16) * 4));
  HIPCHK(fb.gen.reserve((size_t)NC * ngens * 2 * 4));
  if (phase != 2) {
  if (nblk) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_count<SRC>), dim3(nblk), dim3(256), 0, s2, el, (u32)n, fb.cnt.as<u32>(), nblk);
  hipLaunchKernelGGL(k_part_scan, dim3(NC), dim3(SCAN_T), 0, s2, fb.cnt.as<u32>(), fb.off.as<u32>(), nblk, fb.tot.as<u32>());
  if (nblk) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_scatter<SRC>), dim3(nblk), dim3(256), 0, s2, el, elpos, (u32)n, fb.off.as<u32>(), nblk, fb.tot.as<u32>(), fb.sym.as<u16>(), fb.pos.as<u32>());
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_genstart<SRC>), dim3(ngens), dim3(64), 0, s2, el, (u32)n, ranges, ngens, fb.off.as<u32>(), nblk, fb.tot.as<u32>(), fb.gen.as<u32>());
  }
  if (phase == 1) return SCPR_OK;
  // How far apart (in coder entries) a generation's model waves may be, scpr_fixed.hpp.  Measured on the headline batch (300
  // generations), bytes written by the kernel / its time: no window 3.74 GB / 2.16 ms, 16384 entries 1.56 / -, 4096 0.95 / 1.28,
  // 2048 0.84 GB / 1.28 ms - 0.83 GB is every 32-byte sector of the entry array written once (the colour entries between the
  // fixed models' are written by another kernel at another time).  With few generations there is nothing to hold back for (one
  // 300-frame GOP is ONE workgroup: in step it takes 15.6-19.8 ms instead of 12.3, its waves waiting for the model with the
  // most symbols window after window): no window.  SCPR_FIXED_WINDOW overrides (0: none).
  static const long window_env = getenv("SCPR_FIXED_WINDOW") ? strtol(getenv("SCPR_FIXED_WINDOW"), nullptr, 0) : -1;
  const u32 auto_window = ngens >= 16 ? 2048u : 0u;
  const u32 fixed_window = window_env >= 0 ? (u32)window_env : auto_window;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fixed_chain2<SRC, MAXSYM>), dim3(ngens), dim3(64 * NC), (size_t)NC * 2 * MAXSYM * 4, s2, fb.sym.as<u16>(), fb.pos.as<u32>(), fb.gen.as<u32>(), ngens,
                     load_first ? 1 : 0, pin, pout, c->entries.as<u32>(), fixed_window);
  (void)c;
  return SCPR_OK;
}
// Blocks per call up to which the rANS stage runs one wave per block with the state on the scalar unit (k_rans_s,
// scpr_rans_s.hpp).  A wave of that form alone on its SIMD takes ~67 cycles per entry (3.7 ms for a full block), two on one SIMD
// take turns at the scalar unit (6.4 ms), three 9.0; k_rans' vector form takes 7.5 ms whatever the count.  The card has 1024
// SIMDs and the workgroups (four waves, one per SIMD of a CU) spread evenly enough for 2040 blocks to stay at two per SIMD
// (tools/exp_rans.py).  SCPR_RANS_SCALAR_MAX overrides it (0: always the vector form; tests and A/B timing).
constexpr int kRansScalarMax = 2048;
static u64 chunk_total_limit() {  // SCPR_DEBUG_CHUNK_LIMIT: tests reach the re-cut of a chunk with small frames (read per chunk: a test may set it)
  const char* e = getenv("SCPR_DEBUG_CHUNK_LIMIT");
  return e ? strtoull(e, nullptr, 0) : kChunkTotalLimit;
}
constexpr int kMaxChunkGens = 1023;  // generations per encode chunk: a colour key names its generation in ten bits (emit_colour)
constexpr int kRecut = 1;  // encode_chunk: the chunk's symbol totals pass 32 bits - nothing has been coded, *nfit frames would fit
static int encode_chunk(scpr_codec* c, int n, std::vector<ChunkFrame>& cf, int ngens, bool load_first, std::vector<FrameBase>& hb, std::vector<u32>& pchanged, int* nfit, u64 room,
                        const EncHostState& hs0) {
  const Geom& g = c->g;
  hipStream_t st = c->stream;
  const u8* planes = c->planes.as<u8>();
  const int nbx = (g.W + 15) / 16, nby = (g.H + 15) / 16, nblocks = nbx * nby;
  const MvParams mp{(int)std::min<u32>(c->prm.high_range_x, 256), (int)std::min<u32>(c->prm.high_range_y, 256), (int)c->prm.low_range_x, (int)c->prm.low_range_y};
  std::vector<int> islots, igens, ifidx, kinds(n), pidx(n, -1), slots(n);
  std::vector<PFrame> pfr;
  std::vector<int> pgen, pfidx, lossslots;
  for (int i = 0; i < n; i++) {
    kinds[i] = cf[i].kind;
    slots[i] = i;
    if (cf[i].kind == 0) {
      islots.push_back(i);
      igens.push_back(cf[i].gen);
      ifidx.push_back(i);
      lossslots.push_back(i);
    } else if (cf[i].kind == 2) {
      pidx[i] = (int)pfr.size();
      pfr.push_back({i, i > 0 ? i - 1 : c->pslot});
      pgen.push_back(cf[i].gen);
      pfidx.push_back(i);
      lossslots.push_back(i);
    }
  }
  const int ni = (int)islots.size(), np = (int)pfr.size();
  hb.assign(n, FrameBase{});
  pchanged.assign(np, 0);
  if (ni + np == 0) return SCPR_OK;
  HIPCHK(h2d(c, c->kinds.p, kinds.data(), n * 4, st));
  HIPCHK(h2d(c, c->pidx.p, pidx.data(), n * 4, st));
  if (c->loss_mask != 0xFFFFFFFFu) {  // DoLoss: coded frames only (flat frames keep the source bytes, screencap.cpp:1488-1499)
    HIPCHK(h2d(c, c->slotlist.p, lossslots.data(), lossslots.size() * 4, st));
    dim3 gl((g.H * (g.S >> 2) + 255) / 256, (unsigned)lossslots.size());
    hipLaunchKernelGGL(k_loss, gl, dim3(256), 0, st, c->planes.as<u8>(), g, c->slotlist.as<int>(), c->loss_mask, c->corr_mask);
    HIPCHK(sync_out(c, st));  // slotlist is reused below
  }
  const int* d_slots = c->slotlist.as<int>();
  if (ni) {
    HIPCHK(h2d(c, c->slotlist.p, islots.data(), ni * 4, st));
    HIPCHK(h2d(c, c->genlist.p, igens.data(), ni * 4, st));
    HIPCHK(h2d(c, c->fidx.p, ifidx.data(), ni * 4, st));
    stage_begin(c, ST_CLASSIFY);
    hipLaunchKernelGGL(k_tiles, dim3(g.ntiles, ni), dim3(256), 0, st, planes, g, d_slots, c->exitmap.as<u8>(), c->tnmap.as<u16>());
    hipLaunchKernelGGL(k_entries, dim3(ni), dim3(256), 0, st, c->exitmap.as<u8>(), c->entry.as<u8>(), g, d_slots);
    // k_runs walks one tile per lane, a cache line of the type/length map at a time; a fixed number of workgroups take the groups
    // of 256 tiles grid-stride and hand their tiles to their lanes one at a time (scpr_kernels.hpp).  Workgroups per CU, 1080p x 300:
    // 2 / 4 / 8 = 1.49 / 1.37 / 1.28 ms (tiles fixed to lanes, the form before: 1.59 / 1.45 / 1.50).  Four: with eight the map's lines
    // in use no longer fit the L2 and the kernel fetches 3.9 GB instead of 1.8 for its 0.1 ms.
    static const int runs_per_cu = getenv("SCPR_RUNS_PER_CU") ? atoi(getenv("SCPR_RUNS_PER_CU")) : 4;
    if (ni > KRUNS_MAXSLOTS) return SCPR_E_PARAM;  // (a chunk holds at most kMaxChunkGens generations)
    const int run_groups = (int)(((size_t)g.ntiles * ni + 255) / 256);
    hipLaunchKernelGGL(k_runs, dim3((unsigned)std::max(1, std::min(run_groups, runs_per_cu * c->ncu))), dim3(256), 0, st, g, d_slots, ni, c->entry.as<u8>(), c->tnmap.as<u16>(),
                       c->runrec.as<u32>(), c->tilecnt.as<u32>(), (u32)std::min<unsigned long long>((0x100000000ull + (u32)g.ntiles - 1u) / (u32)g.ntiles, 0xFFFFFFFFull));
    hipLaunchKernelGGL(k_header, dim3(ni), dim3(64), 0, st, planes, g, d_slots, c->hdrrec.as<u32>(), c->hdrcnt.as<u32>());
    stage_end(c, ST_CLASSIFY);
    stage_begin(c, ST_SCAN);
    hipLaunchKernelGGL(k_scan_tiles, dim3(ni), dim3(256), 0, st, c->tilecnt.as<u32>(), c->tileoff.as<u32>(), c->frametot.as<u32>(), g, d_slots);
    stage_end(c, ST_SCAN);
  }
  if (np) {
    const size_t pb = (size_t)np * nblocks;
    HIPCHK(c->bcnt.reserve(pb * 4));
    HIPCHK(c->boff.reserve(pb * sizeof(BOff)));
    HIPCHK(c->bflag.reserve(pb * 4));
    HIPCHK(c->plist.reserve(pb * 4 + 64));  // the chunk's pixel-coded blocks, behind their number (k_pactive; k_pcount walks it)
    stage_begin(c, ST_INTER);
    {
      int rc = motion_stage(c, pfr);
      if (rc != SCPR_OK) return rc;
    }
    HIPCHK(hipMemsetAsync(c->plist.p, 0, 4, st));
    hipLaunchKernelGGL(k_pactive, dim3((nblocks + 1023) / 1024, np), dim3(1024), 0, st, nblocks, c->btype.as<u8>(), c->plist.as<u32>(), c->bcnt.as<u32>());
    hipLaunchKernelGGL(k_pcount, dim3((unsigned)std::min<size_t>((pb + 63) / 64, 4096)), dim3(64), 0, st, planes, g, c->pframes.as<PFrame>(), c->binfo.as<u32>(), c->plist.as<u32>(),
                       c->bcnt.as<u32>());
    hipLaunchKernelGGL(k_pscan, dim3(np), dim3(64), 0, st, g, np, c->btype.as<u8>(), c->bmv.as<u32>(), c->bcnt.as<u32>(), c->pinfo.as<int>(), c->boff.as<BOff>(),
                       c->bflag.as<u32>(), c->ptot.as<u32>());
    stage_end(c, ST_INTER);
  }
  hipLaunchKernelGGL(k_bases, dim3(1), dim3(64), 0, st, c->kinds.as<int>(), c->pidx.as<int>(), n, c->frametot.as<u32>(),
                     c->hdrcnt.as<u32>(), c->ptot.as<u32>(), c->bases.as<FrameBase>(), c->totals.as<u32>(), chunk_total_limit());
  u32 tot[5];
  HIPCHK(d2h(c, hb.data(), c->bases.p, n * sizeof(FrameBase), st));
  HIPCHK(d2h(c, tot, c->totals.p, sizeof tot, st));
  if (np) HIPCHK(d2h(c, pchanged.data(), c->pflag.p, (size_t)np * 4, st));
  HIPCHK(sync_out(c, st));
  *nfit = (int)tot[4];
  if (*nfit < n) return kRecut;  // (the 32-bit bases of the frames past *nfit have wrapped: the caller cuts the chunk there)
  // Where the colour pairs go (scpr_ctxsort.hpp): a generation's share of the pair arrays is three PLANE arrays of L pairs each
  // (L = its literals: frames of a generation are consecutive, so its pairs are [col_base of its first frame, + 3 L)); a frame
  // writes its literals behind those of the generation's earlier frames.  pad0 = the frame's first pair, pad1 = L.
  std::vector<CsSeg> segs;
  std::vector<CsBlock> cblocks;
  segs.reserve((size_t)3 * ngens);
  cblocks.reserve((size_t)tot[2] / CS_B + (size_t)3 * ngens + 8);  // (this sits between a read-back and k_symbols on the critical path)
  {
    std::vector<u32> gbase(ngens, 0), glit(ngens, 0);
    std::vector<bool> seen(ngens, false);
    for (int i = 0; i < n; i++) {
      const int gq = cf[i].gen;
      if (gq < 0 || cf[i].kind == 1 || hb[i].ncol == 0) continue;
      if (!seen[gq]) gbase[gq] = hb[i].col_base, seen[gq] = true;
      hb[i].pad0 = gbase[gq] + glit[gq];
      glit[gq] += hb[i].ncol / 3;
    }
    for (int i = 0; i < n; i++)
      if (cf[i].gen >= 0 && cf[i].kind != 1) hb[i].pad1 = glit[cf[i].gen];
    for (int gq = 0; gq < ngens; gq++)
      for (u32 p = 0; p < 3 && glit[gq]; p++) {
        const u32 base = gbase[gq] + p * glit[gq];
        CsSeg sg{(u32)cblocks.size(), 0, base, 0};
        for (u32 o = 0; o < glit[gq]; o += CS_B) cblocks.push_back(CsBlock{base + o, std::min<u32>(CS_B, glit[gq] - o), (u32)segs.size(), 0});
        sg.blk_end = (u32)cblocks.size();
        segs.push_back(sg);
      }
    HIPCHK(h2d(c, c->bases.p, hb.data(), n * sizeof(FrameBase), st));
  }
  {  // may the chunk's packets not fit?  Then the state the chains are about to change is kept first (nothing has changed it yet).
    u64 bound = 0;
    for (int i = 0; i < n; i++) bound += packet_bound(cf[i].hdr_len, cf[i].kind == 1 ? 0 : hb[i].nsyms);
    if (bound > room) {
      int rc = snap_take(c, hs0);
      if (rc != SCPR_OK) return rc;
    }
  }
  const size_t Rtot = tot[0], Ttot = tot[1], Ctot = tot[2], Mtot = tot[3];
  const size_t nchains = (size_t)ngens * NCOLCTX;
  HIPCHK(c->runs.reserve(Rtot * 4 + 64));
  HIPCHK(c->runpos.reserve(Rtot * 4 + 64));
  HIPCHK(c->misc.reserve(Mtot * 4 + 64));
  HIPCHK(c->miscpos.reserve(Mtot * 4 + 64));
  for (int k = 0; k < 2; k++) {
    HIPCHK(c->keys[k].reserve(Ctot * 4 + 64));
    HIPCHK(c->vals[k].reserve(Ctot * 4 + 64));
  }
  HIPCHK(c->cstart.reserve((nchains + 1) * 4));
  HIPCHK(c->entries.reserve(Ttot * 4 + 64));
  // dense-table arena: tables of the live generation stay valid while it continues
  if (!load_first) {
    c->arena_used_bound = 0;
    c->live_symbols = 0;
    HIPCHK(hipMemsetAsync(c->arena_top.p, 0, 4, st));
  }
  // A context gets its table once, after at least 16 of its symbols (ans_contexts.cpp:3-50) - counted over the whole
  // generation, not over this call: a context may meet its 16th symbol here after 15 in earlier calls.  So the tables
  // alive at the end of this call are at most (symbols of the live generation so far + this call's) / 16, and 12288 per generation.
  c->live_symbols += Ctot;
  // (the arena holds the live generation's tables and nothing else when a call starts: compact_tables after every call with several)
  const size_t arena_cap = std::min<size_t>((size_t)ngens * NCOLCTX, c->arena_used_bound + c->live_symbols / 16 + (size_t)ngens) + 64;
  HIPCHK(c->arena.reserve_keep(arena_cap * sizeof(DenseTab), c->arena_used_bound * sizeof(DenseTab), st));
  if (ngens > 1) c->live_symbols = Ctot;  // (an upper bound for the generation that is live after this call)
  c->dbg_entries = (int64_t)Ttot;

  stage_begin(c, ST_SYMBOLS);
  if (ni)
    hipLaunchKernelGGL(k_symbols, dim3((g.ntiles + 3) / 4 + 1, ni), dim3(256), 0, st, planes, g, d_slots, c->genlist.as<int>(), c->fidx.as<int>(), c->bases.as<FrameBase>(), c->runrec.as<u32>(),
                       c->tilecnt.as<u32>(), c->tileoff.as<u32>(), c->entry.as<u8>(), c->hdrrec.as<u32>(), c->runs.as<u32>(), c->runpos.as<u32>(), c->keys[0].as<u32>(),
                       c->vals[0].as<u32>());
  if (np) {
    std::vector<PBase> pbv(np);
    for (int k = 0; k < np; k++) {
      const FrameBase& b = hb[pfidx[k]];
      pbv[k] = PBase{b.run_base, b.sym_base, b.pad0, b.misc_base, b.nbt, (u32)pgen[k], b.pad1, 0};
    }
    HIPCHK(h2d(c, c->pbase.p, pbv.data(), np * sizeof(PBase), st));
    hipLaunchKernelGGL(k_pemit, dim3((nblocks + 63) / 64 + 1, np), dim3(64), 0, st, planes, g, c->pframes.as<PFrame>(), c->pbase.as<PBase>(), c->binfo.as<u32>(), c->btype.as<u8>(),
                       c->bmv.as<u32>(), c->boff.as<BOff>(), c->bflag.as<u32>(), c->pinfo.as<int>(), mp, c->runs.as<u32>(), c->runpos.as<u32>(), c->keys[0].as<u32>(),
                       c->vals[0].as<u32>(), c->misc.as<u32>(), c->miscpos.as<u32>(), c->entries.as<u32>());
    HIPCHK(sync_out(c, st));  // pbv is host memory
  }
  stage_end(c, ST_SYMBOLS);

  if (getenv("SCPR_DEBUG_KEYS") && Ctot) {  // design aid: every colour key must name a generation of this chunk and a context below NCOLCTX
    std::vector<u32> hk(Ctot);
    HIPCHK(d2h(c, hk.data(), c->keys[0].p, Ctot * 4, st));
    HIPCHK(sync_out(c, st));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < Ctot; i++)
      if ((hk[i] >> 22) >= (u32)ngens || ((hk[i] >> 8) & 0x3FFFu) >= (u32)NCOLCTX) {
        if (!bad) first = i;
        bad++;
      }
    fprintf(stderr, "[scpr debug] chunk n=%d ngens=%d ni=%d np=%d Ctot=%zu bad keys=%zu first=%zu key=%08x\n", n, ngens, ni, np, (size_t)Ctot, bad, first, bad ? hk[first] : 0u);
    if (bad) return SCPR_E_DEVICE;
  }
  // per-generation ranges of the run list and of the misc list (frames of a generation are consecutive)
  std::vector<GenRange>& rg = c->h_ranges;
  std::vector<MiscRange>& mr = c->h_miscranges;
  rg.assign(ngens, GenRange{0, 0});
  mr.assign(ngens, MiscRange{0, 0});
  {
    std::vector<bool> seen(ngens, false);
    for (int i = 0; i < n; i++) {
      const int gq = cf[i].gen;
      if (gq < 0 || cf[i].kind == 1) continue;
      if (!seen[gq]) {
        rg[gq].run_begin = hb[i].run_base;
        mr[gq].begin = hb[i].misc_base;
        seen[gq] = true;
      }
      rg[gq].run_end = hb[i].run_base + hb[i].nruns;
      mr[gq].end = hb[i].misc_base + hb[i].nmisc;
    }
  }
  HIPCHK(h2d(c, c->ranges.p, rg.data(), ngens * sizeof(GenRange), st));
  HIPCHK(h2d(c, c->miscranges.p, mr.data(), ngens * sizeof(MiscRange), st));
  // The fixed-model chains (run lengths, pixel types, P-frame symbols) and the colour chains read the same lists
  // and write disjoint coder entries: they run side by side on two streams and join before the coder.  The fork is in front of
  // the colour symbols' partition since round 5 (the fixed models need the run list only): beside rocPRIM's radix sort the fork
  // had cost more than it gave (rounds 2 and 3: the sort took twice as long, 3.7 -> 7.2 ms at 4K); the repo's own partition is a
  // third of that traffic, and the fixed branch was the longer one by 1.4 ms (kernel trace of the headline).  1080p encode
  // 18.0 -> 17.25 ms; with only the chains forked and the run list's partition in front of the colour one on the main stream:
  // 17.5 (k_fixed_chain2's waves wait for each other within their window and cost the partition beside them 0.9 ms); a higher
  // stream priority for the main stream changes nothing (tools/r5/enc_stages.sh).
  const int buf_in = c->live_buf, buf_out = ngens > 1 ? 1 - c->live_buf : c->live_buf;
  // Where exactly: in FRONT of the colour partition since its passes and k_chain_starts got faster (tools/r5/fork_ab.sh, 1080p encode:
  // in front of the partition 15.2 ms, behind its first pass 15.6-15.7, behind both 15.7; 4K key frames 40.7 / 40.1 / 40.9 GPix/s;
  // with the slower partition of earlier in the round the middle position had been the best: 17.35 / 17.05 / 17.75 ms).
  // SCPR_FORK_AT = 0 / 1 / 2 for A/B timing.
  static const int fork_at = getenv("SCPR_FORK_AT") ? atoi(getenv("SCPR_FORK_AT")) : 0;
  // The fixed models' PARTITION runs at the fork, their CHAINS are launched only behind the colour partition: k_fixed_chain2's waves
  // (twelve to a workgroup, waiting for each other within their window) beside the colour partition cost it 0.7 ms (its two
  // scatters 0.9 + 0.7 instead of 0.33 + 0.6), beside the colour chains they cost less.  1080p encode 41.1-41.7 -> 41.6-42.5 GPix/s,
  // 4K 40.1-40.7 -> 40.7-40.8, I+P unchanged.  SCPR_FIXED_GATE=0: the whole branch at the fork.
  static const bool fixed_gate = !(getenv("SCPR_FIXED_GATE") && atoi(getenv("SCPR_FIXED_GATE")) == 0);
  auto fork_fixed_branch = [&](int phase = 0) -> int {
    hipStream_t s2 = getenv("SCPR_SERIAL_CHAINS") ? st : c->stream2;  // (design aid: the two chain stages one after the other, to time each alone)
    HIPCHK(hipEventRecord(c->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(s2, c->ev_fork, 0));
    if (phase != 2) stage_begin(c, ST_FIXED, s2);
    // the run list, then the list of P-frame symbols: partitioned by model (scpr_fixed.hpp), one wave per (generation, model)
    {
      int rc = fixed_chains<RunItems, 256>(c, s2, c->fixr, c->runs.as<u32>(), c->runpos.as<u32>(), Rtot, (const uint2*)c->ranges.p, ngens, load_first,
                                           c->fixed_persist.as<FixedPersist>() + buf_in * 12, c->fixed_persist.as<FixedPersist>() + buf_out * 12, phase);
      if (rc != SCPR_OK) return rc;
      if (Mtot) {
        rc = fixed_chains<MiscItems, 512>(c, s2, c->fixm, c->misc.as<u32>(), c->miscpos.as<u32>(), Mtot, (const uint2*)c->miscranges.p, ngens, load_first,
                                          c->misc_persist.as<FixedPersist>() + buf_in * MC_COUNT, c->misc_persist.as<FixedPersist>() + buf_out * MC_COUNT, phase);
        if (rc != SCPR_OK) return rc;
      } else if (phase != 1 && !(load_first && ngens == 1)) {  // a new generation without any P-frame symbol: its P-frame models are the renewed ones, not the kept ones
        HIPCHK(hipMemsetAsync(c->misc_persist.as<FixedPersist>() + buf_out * MC_COUNT, 0, MC_COUNT * sizeof(FixedPersist), s2));
      }
    }
    if (phase == 1) return SCPR_OK;
    stage_end(c, ST_FIXED, s2);
    HIPCHK(hipEventRecord(c->ev_join, s2));
    return SCPR_OK;
  };
  if (fork_at == 0 || Ctot == 0) {
    const int rc = fork_fixed_branch(fixed_gate && Ctot ? 1 : 0);
    if (rc != SCPR_OK) return rc;
  }
  stage_begin(c, ST_SORT);
  {
    // stable partition of every (generation, plane) segment by its 12 context bits: two counting passes of six bits, least
    // significant first (scpr_ctxsort.hpp); keys[0] -> keys[1] -> keys[0]
    if (Ctot > 0) {
      const unsigned nsb = (unsigned)cblocks.size(), nsg = (unsigned)segs.size();
      HIPCHK(c->csblocks.reserve((size_t)nsb * sizeof(CsBlock)));
      HIPCHK(c->cssegs.reserve((size_t)nsg * sizeof(CsSeg)));
      HIPCHK(c->cscnt.reserve((size_t)nsb * 64 * 4));
      HIPCHK(c->csoff.reserve((size_t)nsb * 64 * 4));
      HIPCHK(h2d(c, c->csblocks.p, cblocks.data(), (size_t)nsb * sizeof(CsBlock), st));
      HIPCHK(h2d(c, c->cssegs.p, segs.data(), (size_t)nsg * sizeof(CsSeg), st));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_count<8>), dim3(nsb), dim3(256), 0, st, c->keys[0].as<u32>(), c->csblocks.as<CsBlock>(), c->cscnt.as<u32>());
      hipLaunchKernelGGL(k_cs_scan, dim3(nsg), dim3(64), 0, st, c->cssegs.as<CsSeg>(), c->cscnt.as<u32>(), c->csoff.as<u32>());
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_scatter<8>), dim3(nsb), dim3(256), 0, st, c->keys[0].as<u32>(), c->vals[0].as<u32>(), c->csblocks.as<CsBlock>(), c->csoff.as<u32>(),
                         c->keys[1].as<u32>(), c->vals[1].as<u32>());
      if (fork_at == 1) {
        const int rc = fork_fixed_branch();
        if (rc != SCPR_OK) return rc;
      }
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_count<14>), dim3(nsb), dim3(256), 0, st, c->keys[1].as<u32>(), c->csblocks.as<CsBlock>(), c->cscnt.as<u32>());
      hipLaunchKernelGGL(k_cs_scan, dim3(nsg), dim3(64), 0, st, c->cssegs.as<CsSeg>(), c->cscnt.as<u32>(), c->csoff.as<u32>());
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_scatter<14>), dim3(nsb), dim3(256), 0, st, c->keys[1].as<u32>(), c->vals[1].as<u32>(), c->csblocks.as<CsBlock>(), c->csoff.as<u32>(),
                         c->keys[0].as<u32>(), c->vals[0].as<u32>());
      if (fork_at >= 2 || (fork_at == 0 && fixed_gate)) {
        const int rc = fork_fixed_branch(fork_at == 0 ? 2 : 0);
        if (rc != SCPR_OK) return rc;
      }
    }
    // (tests: two keys of the partitioned array change places - the order proof below must end the call)
    if (c->dbg_inject == 2 && Ctot > 1) {
      c->dbg_inject = 0;
      hipLaunchKernelGGL(k_debug_swap, dim3(1), dim3(1), 0, st, c->keys[0].as<u32>(), 0u, (u32)Ctot - 1u);
    }
    // chain starts from the sorted keys - and the proof that they ARE sorted: a pair out of order sets bit 5 of the error word, which
    // keeps the chain kernels from following chain lengths made of garbage (k_colour_chain_w, k_chain_lists) and ends the call
    hipLaunchKernelGGL(k_chain_starts, dim3((unsigned)((Ctot + 1024) / 1024)), dim3(256), 0, st, c->keys[0].as<u32>(), (u32)Ctot, (u32)nchains, c->cstart.as<u32>(), c->err.as<u32>());
  }
  stage_end(c, ST_SORT);
  if (getenv("SCPR_DEBUG_KEYS") && Ctot) {
    std::vector<u32> hk(Ctot), hc(nchains + 1);
    HIPCHK(d2h(c, hk.data(), c->keys[0].p, Ctot * 4, st));
    HIPCHK(d2h(c, hc.data(), c->cstart.p, (nchains + 1) * 4, st));
    HIPCHK(sync_out(c, st));
    size_t unsorted = 0, firstu = 0, badc = 0, firstc = 0;
    for (size_t i = 1; i < Ctot; i++)
      if ((hk[i] >> 8) < (hk[i - 1] >> 8)) {
        if (!unsorted) firstu = i;
        unsorted++;
      }
    for (size_t q = 0; q < nchains; q++)
      if (hc[q + 1] < hc[q] || hc[q + 1] > Ctot) {
        if (!badc) firstc = q;
        badc++;
      }
    fprintf(stderr, "[scpr debug] sorted: out-of-order keys=%zu (first %zu: %08x after %08x), bad chain starts=%zu (first q=%zu: %u then %u), last=%u\n", unsorted, firstu,
            unsorted ? hk[firstu] : 0u, unsorted ? hk[firstu - 1] : 0u, badc, firstc, badc ? hc[firstc] : 0u, badc ? hc[firstc + 1] : 0u, hc[nchains]);
    if (unsorted || badc) return SCPR_E_DEVICE;
  }

  stage_begin(c, ST_COLOUR);
  {
    Arena ar{c->arena.as<DenseTab>(), c->arena_top.as<u32>(), (u32)arena_cap - 1u, c->err.as<u32>()};  // (the last table allocated is the sink)
    const u32 stamp_out = (load_first && ngens == 1) ? c->live_stamp : c->next_stamp++;
    ChainPersist cp{c->colour_persist.as<ColState>() + (size_t)buf_in * NCOLCTX, c->colour_persist.as<ColState>() + (size_t)buf_out * NCOLCTX, c->live_stamp, stamp_out, load_first ? 1 : 0, ngens};
    c->live_stamp = stamp_out;
    c->live_buf = buf_out;
    const u32 cap = (u32)std::min<size_t>(nchains, Ctot + 1);
    HIPCHK(c->chainlists.reserve((size_t)cap * 4 * CHAIN_CLASSES + 64));
    HIPCHK(c->chaincounts.reserve(16));
    HIPCHK(hipMemsetAsync(c->chaincounts.p, 0, 4 * CHAIN_CLASSES, st));
    hipLaunchKernelGGL(k_chain_lists, dim3((unsigned)((nchains + 1023) / 1024)), dim3(1024), 0, st, c->cstart.as<u32>(), (int)nchains, c->chainlists.as<u32>(), cap,
                       c->chaincounts.as<u32>());
    const unsigned grid = (unsigned)std::min<u32>(cap, 24576u);
    if (grid)
      hipLaunchKernelGGL(k_colour_chain_w, dim3(grid), dim3(64), 0, st, c->keys[0].as<u32>(), c->vals[0].as<u32>(), c->cstart.as<u32>(), c->chainlists.as<u32>(),
                         c->chaincounts.as<u32>(), cap, c->f0, ar, cp, c->entries.as<u32>());
  }
  stage_end(c, ST_COLOUR);
  HIPCHK(hipStreamWaitEvent(st, c->ev_join, 0));  // the coder needs the entries of both
  return SCPR_OK;
}

// ---------------------------------------------------------------------------
extern "C" {

const char* scpr_version(void) { return "screenpressor_amd 0.1 (gfx950)"; }
const char* scpr_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }

scpr_codec* scpr_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    fprintf(stderr, "[scpr] no usable HIP device (count=%d, asked %d): this library has no CPU path\n", ndev, device);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  scpr_codec* c = new scpr_codec;
  c->device = device;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->ncu = v;
  }
  // (SCPR_S2_PRIORITY, design aid: the fixed models' branch on a stream of another priority - -1 high, 1 low, as far as the device
  // has them.  Measured, tools/r5/s2prio_ab.sh: 1080p encode 39.4-40.5 / 40.1-40.7 / 40.1-40.8 GPix/s, 4K 40.2 / 40.0 / 39.7: nothing.)
  int s2prio = getenv("SCPR_S2_PRIORITY") ? atoi(getenv("SCPR_S2_PRIORITY")) : 0, plo = 0, phi = 0;
  (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
  s2prio = s2prio < 0 ? phi : s2prio > 0 ? plo : 0;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, s2prio) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return nullptr;
  }
  for (int s = 0; s < ST_COUNT + 1; s++)
    for (int k = 0; k < 2; k++) (void)hipEventCreate(&c->ev[s][k]);
  timing_reset(c);
  {  // read once, here: the form of the coder and whether the fault injections of the tests are armed
    const char* e = getenv("SCPR_RANS_SCALAR_MAX");
    c->rans_scalar_max = e ? atoi(e) : kRansScalarMax;
    e = getenv("SCPR_ENABLE_DEBUG_INJECT");
    c->dbg_armed = e && atoi(e) != 0;
  }
  return c;
}

int scpr_init(scpr_codec* c, const scpr_params* p) {  // ScreenCodec::Init, screencap.cpp:1565-1584
  if (!c || !p) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  if (c->inited) scpr_deinit(c);
  c->prm = *p;
  if (c->prm.workers < 1) c->prm.workers = 1;
  c->bpp = (int)p->bits_per_pixel / 8;
  // input rows: RGB32 and RGB16 are read back to back (screencap.cpp:1655 `i = y*X*4`, :1668 `i = y*X*2`: no row padding,
  // also for odd widths of RGB16), RGB24 rows are padded to 4 bytes (the plane's own stride, :1592)
  c->pitch_in = p->bits_per_pixel == 24 ? (((int)p->width * 3 + 3) & ~3) : (int)p->width * c->bpp;
  c->last_loss = (int)p->loss;
  c->rs = c->gs = c->bs = 0;
  if (p->bits_per_pixel == 16) {
    while (c->rs < 16 && !((1u << c->rs) & p->red_mask)) c->rs++;
    while (c->gs < 16 && !((1u << c->gs) & p->green_mask)) c->gs++;
    while (c->bs < 16 && !((1u << c->bs) & p->blue_mask)) c->bs++;
  }
  c->inited = true;
  c->have_codec = false;
  c->crashed = false;
  c->frames_done = 0;
  c->last_flat = false;
  c->last_flat_rgb = 0;
  return SCPR_OK;
}

void scpr_deinit(scpr_codec* c) {  // ScreenCodec::Deinit, screencap.cpp:1619-1629
  if (!c || c->crashed) return;
  c->have_codec = false;
  c->frames_done = 0;
  c->last_flat = false;
}

void scpr_destroy(scpr_codec* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  pin_reset(c);  // (read-backs a failed call left queued are dropped, not delivered: their destinations are gone)
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);  // (before the ranges its copies read are unregistered)
  for (const auto& r : c->host_ranges)
    if (r.owned) (void)hipHostUnregister(r.base);  // (what scpr_host_pin registered and nobody took back)
  (void)hipGetLastError();
  c->host_ranges.clear();
  if (c->stream3) {
    (void)hipStreamSynchronize(c->stream3);
    (void)hipStreamDestroy(c->stream3);
    for (int k = 0; k < 2; k++) (void)hipEventDestroy(c->ev_in[k]);
  }
  DevBuf* all[] = {&c->hb_frames, &c->hb_packets, &c->hb_list, &c->snap_mvs, &c->snap_state, &c->planes, &c->tnmap, &c->exitmap, &c->entry, &c->runrec, &c->tilecnt, &c->tileoff, &c->hdrrec, &c->hdrcnt, &c->frametot, &c->flags, &c->slotlist, &c->genlist,
                   &c->bases, &c->totals, &c->runs, &c->runpos, &c->keys[0], &c->keys[1], &c->vals[0], &c->vals[1], &c->cstart, &c->csblocks, &c->cssegs, &c->cscnt, &c->csoff,
                   &c->entries, &c->ranges, &c->rblocks, &c->rscratch, &c->rrec, &c->rsize, &c->packets, &c->pktoff, &c->blkdst, &c->outsizes, &c->total64, &c->arena,
                   &c->arena_top, &c->err, &c->rcp, &c->decframes, &c->decstates, &c->hoststage_in, &c->hoststage_out, &c->chainlists, &c->chaincounts, &c->fixr.cnt, &c->fixr.off, &c->fixr.tot, &c->fixr.sym, &c->fixr.pos, &c->fixr.gen, &c->fixm.cnt, &c->fixm.off, &c->fixm.tot, &c->fixm.sym, &c->fixm.pos, &c->fixm.gen, &c->kinds, &c->pidx, &c->fidx, &c->pframes, &c->pflag, &c->binfo, &c->smv, &c->btype, &c->bmv, &c->bcnt, &c->boff, &c->bflag, &c->plist, &c->pinfo, &c->ptot, &c->pbase, &c->misc, &c->miscpos, &c->miscranges, &c->mvs, &c->mvs_keep, &c->fixed_persist, &c->misc_persist, &c->colour_persist, &c->decgops, &c->decfixed, &c->dec_fixed_persist, &c->dec_colour_persist, &c->dec_arena, &c->dec_arena2, &c->arena2, &c->dec_arena_top, &c->mvdict, &c->mvpre, &c->gmask};
  for (DevBuf* b : all) b->release();
  for (int s = 0; s < ST_COUNT + 1; s++)
    for (int k = 0; k < 2; k++) (void)hipEventDestroy(c->ev[s][k]);
  (void)hipStreamDestroy(c->stream);
  (void)hipStreamDestroy(c->stream2);
  if (c->pin) (void)hipHostFree(c->pin);
  (void)hipEventDestroy(c->ev_fork);
  (void)hipEventDestroy(c->ev_join);
  delete c;
}

void scpr_crash_happened(scpr_codec* c) {
  if (c) c->crashed = true;
}

// Sharding support: the state CScreenCapt carries ACROSS key frames (fn > 0, screencap.cpp:1504; last_was_flat /
// last_flat_clr with prev holding that flat picture, :1490-1497), so that a shard that starts after a flat frame
// produces what the single stream produces there.
int scpr_seed_shard(scpr_codec* c, uint32_t frames_before, int last_was_flat, uint32_t last_flat_rgb) {
  if (!c || !c->inited) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  c->frames_done = frames_before;
  c->last_flat = last_was_flat != 0;
  if (last_was_flat) {
    c->last_flat_rgb = last_flat_rgb & 0xFFFFFFu;
    const Geom& g = c->g;
    hipLaunchKernelGGL(k_fill_flat, dim3((g.H * g.S + 255) / 256), dim3(256), 0, c->stream, c->planes.as<u8>(), g, c->pslot, c->last_flat_rgb);
    HIPCHK(sync_out(c, c->stream));
    c->live_valid = true;       // that flat frame renewed the models ...
    c->live_has_state = false;  // ... and nothing has been coded with them
    c->dec_live = false;
  }
  return SCPR_OK;
}

// The vector memory mvs[] (screencap.cpp:96-97): calloc'd by Init, written by every motion search that finds a vector, read as
// "the vector of the block above" by every later P-frame (:726-735) and never reset - RenewI (:178-198) touches models only.  It is
// the one piece of encoder state that crosses key frames besides the flat-frame memory, so a shard that does not start the stream
// needs it handed over (sharding.py, handover_mv_memory).
int scpr_export_mv_memory(scpr_codec* c, int32_t* mx, int32_t* my) {
  if (!c || !c->inited || !mx || !my) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  const int nblocks = ((c->g.W + 15) / 16) * ((c->g.H + 15) / 16);
  std::vector<u32> h(nblocks);
  HIPCHK(d2h(c, h.data(), c->mvs.p, (size_t)nblocks * 4, c->stream));
  HIPCHK(sync_out(c, c->stream));
  for (int i = 0; i < nblocks; i++) mx[i] = (int16_t)(h[i] & 0xFFFF), my[i] = (int16_t)(h[i] >> 16);
  return nblocks;
}

int scpr_import_mv_memory(scpr_codec* c, const int32_t* mx, const int32_t* my) {
  if (!c || !c->inited || !mx || !my) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  const int nblocks = ((c->g.W + 15) / 16) * ((c->g.H + 15) / 16);
  std::vector<u32> h(nblocks);
  for (int i = 0; i < nblocks; i++) {
    if (mx[i] < -256 || mx[i] > 256 || my[i] < -256 || my[i] > 256) return SCPR_E_PARAM;  // (msr_x = msr_y = min(high_range, 256), :76-80)
    h[i] = ((u32)(mx[i] & 0xFFFF)) | ((u32)(my[i] & 0xFFFF) << 16);
  }
  HIPCHK(h2d(c, c->mvs.p, h.data(), (size_t)nblocks * 4, c->stream));
  HIPCHK(sync_out(c, c->stream));
  return nblocks;
}

// What scpr_compress_batch would leave in mvs[] after these frames, without coding them: conversion, loss mask, frame-type
// decisions, block compare and the motion search - the only stages mvs[] depends on (DecideBlockTypes / FindMV read planes and
// mvs[] and nothing of the models).  The codec is left exactly as it was (planes of the chunk slots are scratch).
int scpr_motion_prepass(scpr_codec* c, const void* d_frames, int nframes, const int* ftypes, int loss, int32_t* mx, int32_t* my) {
  if (!c || !c->inited || !d_frames || !ftypes || !mx || !my || nframes < 0) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  if (c->have_codec && c->version == 2) return SCPR_E_BAD_VERSION;
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  const Geom& g = c->g;
  hipStream_t st = c->stream;
  const int nblocks = ((g.W + 15) / 16) * ((g.H + 15) / 16);
  const int loss_before = c->last_loss;
  if (loss != c->last_loss) setup_loss(c, loss);
  const size_t frame_bytes = (size_t)c->pitch_in * g.H;
  // saved: the vector memory and the previous frame of the stream
  DevBuf keep;
  HIPCHK(keep.reserve((size_t)nblocks * 4 + g.plane_stride));
  HIPCHK(hipMemcpyAsync(keep.p, c->mvs.p, (size_t)nblocks * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(sync_out(c, st));
  struct Restore {  // (every return below leaves the codec as it was)
    scpr_codec* c;
    DevBuf& keep;
    int nblocks, loss_before;
    bool planes_saved = false;
    ~Restore() {
      (void)hipMemcpyAsync(c->mvs.p, keep.p, (size_t)nblocks * 4, hipMemcpyDeviceToDevice, c->stream);
      if (planes_saved)
        (void)hipMemcpyAsync(c->planes.as<u8>() + (size_t)c->pslot * c->g.plane_stride, (u8*)keep.p + (size_t)nblocks * 4, c->g.plane_stride, hipMemcpyDeviceToDevice, c->stream);
      (void)sync_out(c, c->stream);
      keep.release();
      if (loss_before != c->last_loss) setup_loss(c, loss_before);
    }
  } restore{c, keep, nblocks, loss_before};
  u32 frames_done = c->frames_done;
  HIPCHK(hipMemsetAsync(c->err.p, 0, 32, st));
  // chunks no larger than the encoder's, and small enough that the block arrays of the chunk stay modest
  const int chunk = std::max(1, std::min(c->slots, 256));
  for (int f0 = 0; f0 < nframes; f0 += chunk) {
    const int n = std::min(chunk, nframes - f0);
    if ((rc = ensure_planes(c, (size_t)n)) != SCPR_OK) return rc;  // (carries the previous frame over when it grows)
    if (!restore.planes_saved) {
      HIPCHK(hipMemcpyAsync((u8*)keep.p + (size_t)nblocks * 4, c->planes.as<u8>() + (size_t)c->pslot * g.plane_stride, g.plane_stride, hipMemcpyDeviceToDevice, st));
      restore.planes_saved = true;
    }
    const size_t ns = (size_t)n + 1;
    HIPCHK(c->flags.reserve(ns * 8));
    HIPCHK(c->slotlist.reserve(ns * 4));
    HIPCHK(c->pframes.reserve(ns * sizeof(PFrame)));
    HIPCHK(c->pflag.reserve(ns * 4));
    HIPCHK(c->pinfo.reserve(ns * 8));
    const u8* src = (const u8*)d_frames + (size_t)f0 * frame_bytes;
    u32* d_nonflat = c->flags.as<u32>();
    u32* d_first = d_nonflat + n;
    HIPCHK(hipMemsetAsync(d_nonflat, 0, (size_t)n * 8, st));
    if (c->bpp == 4) {
      dim3 gr((g.H * ((g.W + 3) >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack32, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else if (c->bpp == 3) {
      dim3 gr((g.H * (g.S >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack24, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else {
      dim3 gr((g.H * g.W + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack16, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first, c->prm.red_mask, c->prm.green_mask, c->prm.blue_mask, c->rs, c->gs,
                         c->bs);
    }
    std::vector<u32> hflags((size_t)n * 2);
    HIPCHK(d2h(c, hflags.data(), d_nonflat, (size_t)n * 8, st));
    HIPCHK(sync_out(c, st));
    // P-frame iff the picture is not flat, a frame has been coded and the caller allows it (screencap.cpp:1488-1511)
    std::vector<PFrame> pfr;
    std::vector<int> lossslots;
    for (int i = 0; i < n; i++) {
      if (hflags[i] == 0) continue;  // flat frame: 4 bytes, fn not incremented
      lossslots.push_back(i);
      if (frames_done && ftypes[f0 + i]) pfr.push_back({i, i > 0 ? i - 1 : c->pslot});
      frames_done++;
    }
    if (c->loss_mask != 0xFFFFFFFFu && !lossslots.empty()) {
      HIPCHK(h2d(c, c->slotlist.p, lossslots.data(), lossslots.size() * 4, st));
      dim3 gl((g.H * (g.S >> 2) + 255) / 256, (unsigned)lossslots.size());
      hipLaunchKernelGGL(k_loss, gl, dim3(256), 0, st, c->planes.as<u8>(), g, c->slotlist.as<int>(), c->loss_mask, c->corr_mask);
    }
    if (!pfr.empty() && (rc = motion_stage(c, pfr)) != SCPR_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->planes.as<u8>() + (size_t)c->pslot * g.plane_stride, c->planes.as<u8>() + (size_t)(n - 1) * g.plane_stride, g.plane_stride, hipMemcpyDeviceToDevice, st));
    HIPCHK(sync_out(c, st));  // (pfr / lossslots are host memory)
    HIPCHK(hipGetLastError());
  }
  u32 err = 0;
  std::vector<u32> h(nblocks);
  HIPCHK(d2h(c, h.data(), c->mvs.p, (size_t)nblocks * 4, st));
  HIPCHK(d2h(c, &err, c->err.p, 4, st));
  HIPCHK(sync_out(c, st));
  if (err & 8) {
    fprintf(stderr, "[scpr] motion-vector pipeline stalled\n");
    return SCPR_E_DEVICE;
  }
  for (int i = 0; i < nblocks; i++) mx[i] = (int16_t)(h[i] & 0xFFFF), my[i] = (int16_t)(h[i] >> 16);
  return nblocks;
}

// A compress call that can be taken back: txn_begin() at its start (host state kept, vector memory copied, and - a call of several
// frames whose buffer is below the closed-form worst case - the device state too: such a call may be cut into chunks, and a later
// chunk cannot take back an earlier one's changes), compress_core() once or, for the host-pointer form, once per sub-batch, and
// txn_refuse() when the core says SCPR_E_CAPACITY.
struct EncTxn {
  EncHostState hs0;
  std::vector<int> ftypes0;
};
static int txn_begin(scpr_codec* c, EncTxn& t, const int* ftypes, int nframes, size_t out_capacity) {
  const Geom& g = c->g;
  t.hs0 = host_state(c);
  t.ftypes0.assign(ftypes, ftypes + nframes);
  c->snap_taken = false;
  const size_t nblk0 = (size_t)((g.W + 15) / 16) * ((g.H + 15) / 16);
  HIPCHK(c->snap_mvs.reserve(nblk0 * 4));
  HIPCHK(hipMemcpyAsync(c->snap_mvs.p, c->mvs.p, nblk0 * 4, hipMemcpyDeviceToDevice, c->stream));
  // (at most 5 coder entries per pixel + 16 per block: Appendix A of SURVEY.md)
  const u64 worst = packet_bound(4, 5ull * g.NP + 16ull * nblk0 + 16);
  // (Round 5 tried to leave calls of ONE chunk to encode_chunk's exact bound: tests/stress_cases.py case 1 - a refused call of ten
  // noise P-frames - then coded other bytes afterwards.  A chunk of several frames has moved state by the time its totals are
  // known; the copy at the call's start is tens of microseconds.  Kept as it was.)
  if (nframes > 1 && (u64)out_capacity < worst * (u64)nframes) return snap_take(c, t.hs0);
  return SCPR_OK;
}
static int64_t txn_refuse(scpr_codec* c, const EncTxn& t, int* ftypes, int code) {  // the codec as the call found it
  for (size_t i = 0; i < t.ftypes0.size(); i++) ftypes[i] = t.ftypes0[i];
  const int r2 = snap_restore(c, t.hs0);
  return (int64_t)(r2 != SCPR_OK ? r2 : code);
}
// Any OTHER failure inside a compress call (a HIP error, the dense-table arena, the sortedness proof, a stalled pipeline) finds the
// codec half-way: frame count, flat-frame memory, ftypes, mvs[] and the fixed models already moved, the colour models and the
// previous frame perhaps not.  If the call kept a snapshot it is taken back whole like a refused one; otherwise the codec is
// marked as the reference marks itself after an exception in Compress (`crashed`, screencap.cpp:1634-1644): every later frame is
// refused (0 bytes) until the caller calls Init again.  Never a codec that goes on coding P-frames against models that no
// decoder will have.
static int64_t enc_failed(scpr_codec* c, const EncTxn& t, int* ftypes, int64_t r) {
  pin_reset(c);  // (read-backs the failed call left queued point into its stack)
  if (c->snap_taken && txn_refuse(c, t, ftypes, (int)r) == r) return r;
  c->crashed = true;
  return r;
}
// frames resident on the device -> packets at d_out (device memory, or host memory mapped into the device's address space).
// Returns the bytes written, or < 0; SCPR_E_CAPACITY leaves the taking back to the caller (txn_refuse).
static int64_t compress_core(scpr_codec* c, const void* d_frames, int nframes, int* ftypes, void* d_out, size_t out_capacity, uint32_t* sizes, const EncHostState& hs0) {
  int rc = SCPR_OK;
  const Geom& g = c->g;
  hipStream_t st = c->stream;
  const size_t frame_bytes = (size_t)c->pitch_in * g.H;
  int64_t written = 0;
  HIPCHK(hipMemsetAsync(c->err.p, 0, 32, st));
  for (int f0 = 0, used = 0; f0 < nframes; f0 += used) {
    int n = std::min(c->slots, nframes - f0);  // (may shrink below: at most kMaxChunkGens generations per chunk)
    const int npacked = n;
    if ((rc = ensure_planes(c, (size_t)n)) != SCPR_OK) return rc;
    if ((rc = ensure_enc_scratch(c, (size_t)n)) != SCPR_OK) return rc;
    const u8* src = (const u8*)d_frames + (size_t)f0 * frame_bytes;
    u32* d_nonflat = c->flags.as<u32>();
    u32* d_first = d_nonflat + n;
    HIPCHK(hipMemsetAsync(d_nonflat, 0, (size_t)n * 8, st));
    stage_begin(c, ST_PACK);
    if (c->bpp == 4) {
      dim3 gr((g.H * ((g.W + 3) >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack32, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else if (c->bpp == 3) {
      dim3 gr((g.H * (g.S >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack24, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else {
      dim3 gr((g.H * g.W + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_pack16, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first, c->prm.red_mask, c->prm.green_mask, c->prm.blue_mask, c->rs, c->gs,
                         c->bs);
    }
    stage_end(c, ST_PACK);
    std::vector<u32> hflags((size_t)n * 2);
    HIPCHK(d2h(c, hflags.data(), d_nonflat, (size_t)n * 8, st));
    HIPCHK(sync_out(c, st));

    // frame-type decisions: CScreenCapt::CompressFrame, screencap.cpp:1488-1511 - and once more over fewer frames if the
    // chunk's symbol totals turn out to pass 32 bits (k_bases; noise-like content only: ~5 symbols per pixel)
    std::vector<ChunkFrame> cf;
    std::vector<FrameBase> hb;
    std::vector<u32> pchanged;
    int ngens = 0;
    bool load_first = false;
    const u32 s_frames_done = c->frames_done, s_flat_rgb = c->last_flat_rgb;
    const bool s_flat = c->last_flat, s_live_valid = c->live_valid, s_live_has_state = c->live_has_state;
    const int nblk = ((g.W + 15) / 16) * ((g.H + 15) / 16);
    HIPCHK(c->mvs_keep.reserve((size_t)nblk * 4));
    HIPCHK(hipMemcpyAsync(c->mvs_keep.p, c->mvs.p, (size_t)nblk * 4, hipMemcpyDeviceToDevice, st));  // (the motion stage of a chunk that is cut again has moved it on)
    for (;;) {
      cf.assign(n, ChunkFrame{});
      ngens = 0;
      load_first = false;
      for (int i = 0; i < n; i++) {
        ChunkFrame& fr = cf[i];
        const bool flat = hflags[i] == 0;
        const u32 rgb = hflags[npacked + i] & 0xFFFFFFu;
        // (a chunk stops before its 1024th generation: kMaxChunkGens.  Rounds 1-4 stopped at 512 to keep rocPRIM's radix sort off a
        // bit range on which it returned unsorted output, tools/rocprim_sort_check.hip; the sort is the repo's own now.)
        const bool starts_gen = flat ? !(c->last_flat && c->last_flat_rgb == rgb) : !(c->frames_done && ftypes[f0 + i]);
        if (starts_gen && ngens == kMaxChunkGens) {
          n = i;
          break;
        }
        if (flat) {
          fr.kind = 1;
          fr.hdr_len = 4;
          fr.hdr = (u32)(1 + (c->version - 1) * 16) | (rgb << 8);
          if (!(c->last_flat && c->last_flat_rgb == rgb)) {  // :1490-1494: prev := this frame, models renewed
            fr.gen = ngens++;
            c->live_valid = true;
          } else {
            fr.gen = -1;
          }
          c->last_flat = true;
          c->last_flat_rgb = rgb;
          ftypes[f0 + i] = 0;
          continue;
        }
        c->last_flat = false;
        if (c->frames_done && ftypes[f0 + i]) {
          fr.kind = 2;
          fr.hdr_len = 1;
          fr.hdr = 1;
          if (ngens == 0) {  // continues the generation that was live when the call started
            ngens = 1;
            load_first = c->live_valid && c->live_has_state;
          }
          fr.gen = ngens - 1;
          ftypes[f0 + i] = 1;
        } else {
          fr.kind = 0;
          fr.hdr_len = 1;
          fr.hdr = (u32)(2 + (c->version - 1) * 16);
          fr.gen = ngens++;
          c->live_valid = true;
          ftypes[f0 + i] = 0;
        }
        c->frames_done++;
      }
      cf.resize(n);
      used = n;
      // a flat frame that renews the models starts a generation of its own; if it is first in the chunk nothing is loaded
      const bool any_gen = ngens > 0;
      if (ngens == 0) ngens = 1, load_first = c->live_valid && c->live_has_state;
      if (any_gen) {  // does the generation that is live after this chunk hold coded symbols?
        bool coded = false;
        for (int i = 0; i < n; i++) coded |= cf[i].kind != 1 && cf[i].gen == ngens - 1;
        c->live_has_state = coded;
      }
      int nfit = n;
      rc = encode_chunk(c, n, cf, ngens, load_first, hb, pchanged, &nfit, (u64)(out_capacity - (size_t)written), hs0);
      if (rc == kRecut) {
        if (nfit < 1) return SCPR_E_DEVICE;  // (one frame always fits: 5 symbols per pixel of at most 8000 x 8191 pixels)
        c->frames_done = s_frames_done, c->last_flat_rgb = s_flat_rgb, c->last_flat = s_flat, c->live_valid = s_live_valid, c->live_has_state = s_live_has_state;
        HIPCHK(hipMemcpyAsync(c->mvs.p, c->mvs_keep.p, (size_t)nblk * 4, hipMemcpyDeviceToDevice, st));
        n = nfit;
        continue;
      }
      if (rc != SCPR_OK) return rc;
      break;
    }
    // rANS blocks and packets
    std::vector<RansBlock> blocks;
    std::vector<Packet> pk(n);
    {
      int pk_i = 0;
      for (int i = 0; i < n; i++) {
        pk[i].hdr_len = cf[i].hdr_len;
        pk[i].hdr = cf[i].hdr;
        pk[i].blk_begin = (u32)blocks.size();
        pk[i].blk_count = 0;
        if (cf[i].kind == 2) {
          if (!pchanged[pk_i]) pk[i].hdr = 0;  // nothing changed: the frame is the single byte 0 (:1113-1116)
          pk_i++;
        }
        if (cf[i].kind != 1) {
          const FrameBase& b = hb[i];
          for (u32 o = 0; o < b.nsyms; o += kBlockEntries) {
            blocks.push_back({b.sym_base + o, std::min<u32>(kBlockEntries, b.nsyms - o)});
            pk[i].blk_count++;
          }
        }
      }
    }
    const int nb = (int)blocks.size();
    HIPCHK(c->rblocks.reserve((size_t)nb * sizeof(RansBlock) + 16));
    HIPCHK(c->rscratch.reserve((size_t)nb * RANS_SCRATCH + 16));
    HIPCHK(c->rsize.reserve((size_t)nb * 4 + 16));
    HIPCHK(c->blkdst.reserve((size_t)nb * 8 + 16));
    HIPCHK(c->packets.reserve((size_t)n * sizeof(Packet)));
    HIPCHK(c->pktoff.reserve((size_t)n * 8));
    HIPCHK(c->outsizes.reserve((size_t)n * 4));
    if (nb) HIPCHK(h2d(c, c->rblocks.p, blocks.data(), (size_t)nb * sizeof(RansBlock), st));
    HIPCHK(h2d(c, c->packets.p, pk.data(), (size_t)n * sizeof(Packet), st));
    u64 chunk_total = 0;
    u32 err = 0, atop = 0;
    // the coder, the packets' assembly and the read-backs; scalar: k_rans_s (its hand-over is checked by the kernel itself: bit 64)
    auto code_chunk = [&](bool scalar) -> int {
      if (nb) {
        stage_begin(c, ST_RANS);
        if (scalar) {
          HIPCHK(c->rrec.reserve((size_t)nb * RANS_S_RING * RANS_S_TRIP * sizeof(rans_rec_t)));  // a ring of records per block
          size_t lds = 0;
#ifdef SCPR_EXPERIMENT  // (tools/exp_rans.py: LDS nobody uses, so that a CU takes one workgroup only)
          if (const char* e = getenv("SCPR_RANS_LDS")) lds = (size_t)atoi(e);
          if (lds > 65536) HIPCHK(hipFuncSetAttribute((const void*)k_rans_s<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
          if (c->dbg_inject == 3) {  // (tests: one trip's records are not laid - a stale line, as the scalar unit would see it)
            c->dbg_inject = 0;
            hipLaunchKernelGGL(k_rans_s<true>, dim3((nb + 3) / 4), dim3(512), lds, st, c->entries.as<u32>(), c->rblocks.as<RansBlock>(), c->rcp.as<RansRcp>(), c->rrec.as<rans_rec_t>(), nb,
                               c->rscratch.as<u8>(), c->rsize.as<u32>(), c->err.as<u32>(), 9);
          } else {
            hipLaunchKernelGGL(k_rans_s<false>, dim3((nb + 3) / 4), dim3(512), lds, st, c->entries.as<u32>(), c->rblocks.as<RansBlock>(), c->rcp.as<RansRcp>(), c->rrec.as<rans_rec_t>(), nb,
                               c->rscratch.as<u8>(), c->rsize.as<u32>(), c->err.as<u32>(), -1);
          }
        } else {
          if (c->dbg_inject == 3) c->dbg_inject = 0;  // (nothing to poison in the vector form)
          hipLaunchKernelGGL(k_rans, dim3((nb + 63) / 64), dim3(256), 0, st, c->entries.as<u32>(), c->rblocks.as<RansBlock>(), nb, c->rcp.as<RansRcp>(), c->rscratch.as<u8>(),
                             c->rsize.as<u32>(), c->err.as<u32>());
        }
        stage_end(c, ST_RANS);
      }
      stage_begin(c, ST_GATHER);
      hipLaunchKernelGGL(k_offsets, dim3(1), dim3(256), 0, st, c->packets.as<Packet>(), n, c->rsize.as<u32>(), c->outsizes.as<u32>(), c->pktoff.as<u64>(), c->blkdst.as<u64>(),
                         c->total64.as<u64>());
      hipLaunchKernelGGL(k_gather, dim3(nb + n), dim3(256), 0, st, c->packets.as<Packet>(), n, nb, c->rscratch.as<u8>(), c->rsize.as<u32>(), c->pktoff.as<u64>(),
                         c->blkdst.as<u64>(), (u8*)d_out + written, (u64)(out_capacity - (size_t)written), c->err.as<u32>());
      stage_end(c, ST_GATHER);
      HIPCHK(d2h(c, &atop, c->arena_top.p, 4, st));
      HIPCHK(d2h(c, sizes + f0, c->outsizes.p, (size_t)n * 4, st));
      HIPCHK(d2h(c, &chunk_total, c->total64.p, 8, st));
      HIPCHK(d2h(c, &err, c->err.p, 4, st));
      if (c->dbg_inject == 1) {  // (tests: a failure between a read-back and its hand-over - the pool then holds pointers into this frame)
        c->dbg_inject = 0;
        return SCPR_E_DEVICE;
      }
      HIPCHK(sync_out(c, st));
      HIPCHK(hipGetLastError());  // a kernel that could not be launched (the launches themselves are not checked one by one)
      timing_collect(c);
      return SCPR_OK;
    };
    const bool scalar_form = nb > 0 && nb <= c->rans_scalar_max;
    if ((rc = code_chunk(scalar_form)) != SCPR_OK) return rc;
    if (err & 64) {
      // A step of the scalar form did not come out as its entry says (scpr_rans_s.hpp): the blocks are independent of everything
      // but the entries, so the stage is simply run again in the vector form (whatever the first run wrote is overwritten:
      // scratch, sizes, packets, the capacity verdict), and the codec stays with the vector form.
      fprintf(stderr, "[scpr] k_rans_s: a record did not reach the scalar unit as it was laid; the call's blocks are coded again with k_rans, and this codec keeps to it\n");
      c->rans_scalar_max = 0;
      c->rans_recoded++;
      const u32 keep = err & ~(64u | 2u);
      HIPCHK(h2d(c, c->err.p, &keep, 4, st));
      if ((rc = code_chunk(false)) != SCPR_OK) return rc;
    }
    if (err & 32) {
      fprintf(stderr, "[scpr] the colour symbols are not grouped by context in front of the chains (k_chain_starts' proof): nothing was coded with them\n");
      return SCPR_E_DEVICE;
    }
    if (err & 2) return SCPR_E_CAPACITY;  // (the bound said this could happen: the state was kept - the caller takes the call back)
    c->arena_used_bound = atop;  // what the arena really holds: it does not grow with the number of calls
    // the last plane of the chunk is the "previous frame" of the next call
    HIPCHK(hipMemcpyAsync(c->planes.as<u8>() + (size_t)c->pslot * g.plane_stride, c->planes.as<u8>() + (size_t)(n - 1) * g.plane_stride, g.plane_stride, hipMemcpyDeviceToDevice, st));
    if (err & 1) {
      fprintf(stderr, "[scpr] dense-table arena overflow\n");
      return SCPR_E_DEVICE;
    }
    if (err & 8) {
      fprintf(stderr, "[scpr] motion-vector pipeline stalled\n");
      return SCPR_E_DEVICE;
    }
    if (ngens > 1 && atop > 0) {  // only the last generation's tables are needed again: to the bottom, the rest is dropped
      u32 ntop = 0;
      rc = compact_tables(c, c->arena, c->arena2, c->arena_top, c->colour_persist.as<ColState>() + (size_t)c->live_buf * NCOLCTX, (int)(sizeof(ColState) / 4), 3, c->live_stamp, 0u, atop,
                          &ntop);
      if (rc != SCPR_OK) return rc;
      c->arena_used_bound = ntop;
    }
    written += (int64_t)chunk_total;
  }
  return written;
}

int64_t scpr_compress_batch(scpr_codec* c, const void* d_frames, int nframes, int* ftypes, int loss, void* d_out, size_t out_capacity, uint32_t* sizes) {
  if (!c || !c->inited || !d_frames || !ftypes || !d_out || !sizes || nframes < 0) return SCPR_E_PARAM;
  if (c->crashed) return 0;  // screencap.cpp:1634
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  if (c->have_codec && c->version == 2) return SCPR_E_BAD_VERSION;  // a codec that has decoded version 2 cannot encode (no version 2 encoder here)
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);  // the encoder always writes v4 (screencap.cpp:1646-1648)
  if (rc != SCPR_OK) return rc;
  if (loss != c->last_loss) setup_loss(c, loss);
  timing_reset(c);
  EncTxn txn;
  if ((rc = txn_begin(c, txn, ftypes, nframes, out_capacity)) != SCPR_OK) return rc;
  const int64_t r = compress_core(c, d_frames, nframes, ftypes, d_out, out_capacity, sizes, txn.hs0);
  if (r == SCPR_E_CAPACITY) return txn_refuse(c, txn, ftypes, SCPR_E_CAPACITY);
  return r < 0 ? enc_failed(c, txn, ftypes, r) : r;
}

// packets resident on the device -> frames at d_frames_out: device memory, or (out_is_host) the host's buffer mapped into the
// device's address space - then the chains of coded key frames send every finished row there themselves (decode_intra_frame's
// hdst: the pictures cross PCIe while the chains run) and only what they did not send (P-frames, flat frames; every frame of
// other pixel formats or of a version 2 stream) is unpacked afterwards, by kernels that write to the host directly.
static int decompress_core(scpr_codec* c, const void* d_packets, const uint32_t* sizes, const int* ftypes, int nframes, void* d_frames_out, int pitch, bool out_is_host) {
  hipStream_t st = c->stream;
  // SCPR_DEV_STREAMER=0: the device-resident output never goes through the row streamer (A/B timing; see rows_by_streamer below)
  static const bool dev_streamer = !(getenv("SCPR_DEV_STREAMER") && atoi(getenv("SCPR_DEV_STREAMER")) == 0);
  timing_reset(c);
  // first bytes of every packet decide version / flat / coded (screencap.cpp:1700, :1536)
  std::vector<u64> offs(nframes + 1, 0);
  for (int i = 0; i < nframes; i++) offs[i + 1] = offs[i] + sizes[i];
  std::vector<u32> heads(nframes, 0);
  if (nframes) {
    HIPCHK(c->pktoff.reserve((size_t)(nframes + 1) * 8));
    HIPCHK(c->outsizes.reserve((size_t)nframes * 4));
    HIPCHK(h2d(c, c->pktoff.p, offs.data(), (size_t)(nframes + 1) * 8, st));
    hipLaunchKernelGGL(k_heads, dim3((nframes + 255) / 256), dim3(256), 0, st, (const u8*)d_packets, c->pktoff.as<u64>(), nframes, c->outsizes.as<u32>());
    HIPCHK(d2h(c, heads.data(), c->outsizes.p, (size_t)nframes * 4, st));
  }
  HIPCHK(sync_out(c, st));
  int done = 0;
  HIPCHK(c->err.reserve(64));
  for (int f0 = 0; f0 < nframes;) {
    if (!c->have_codec) {
      if (ftypes[f0] > 0) return done;  // P-frame before any key frame (:1699)
      int version = (int)((heads[f0] & 0xFF) >> 4) + 1;
      int rc = ensure_codec(c, version);
      if (rc != SCPR_OK) return rc;
    }
    const Geom& g = c->g;
    // a chunk: as many frames as there are planes for, and no more GOPs than the GPU runs side by side (768 chains)
    int n = 0;
    for (int keys = 0; f0 + n < nframes && n < c->dslots; n++)
      if (ftypes[f0 + n] == 0 && ++keys > 768) break;
    {
      int rc = ensure_planes(c, (size_t)n);
      if (rc != SCPR_OK) return rc;
    }
    // A chunk of fresh GOPs is decoded with a dense-table arena sized from its packets (a table per 12 bytes of stream, below);
    // should that not be enough the device reports an overflow (nothing is lost: the contexts past the end share the sink
    // table) and the chunk is decoded again with the true bound (every context of every GOP dense, 12288 tables per GOP).
    const bool host_crashed = c->crashed, host_flat = c->last_flat;
    const u32 host_flat_rgb = c->last_flat_rgb, host_frames_done = c->frames_done;
    u32 errv[8] = {0};
    bool rows_by_streamer = out_is_host;  // (decided per chunk, below)
    bool streamed_rgb24 = false;          // ... and whether a chunk of RGB24 output went out that way (nothing left to copy then)
    for (int attempt = 0;; attempt++) {
    c->crashed = host_crashed, c->last_flat = host_flat, c->last_flat_rgb = host_flat_rgb, c->frames_done = host_frames_done;
    std::vector<DecFrame> fr;
    std::vector<DecGop> gops;
    HIPCHK(hipMemsetAsync(c->err.p, 0, 32, st));
    stage_begin(c, ST_DECODE);
    u64 gop_bytes = 0;
    for (int i = 0; i < n; i++) {
      const int fi = f0 + i;
      if (c->crashed && ftypes[fi] > 0) return done;  // (:1697)
      c->crashed = false;
      c->frames_done++;
      DecFrame d{offs[fi], sizes[fi], i, 0, i > 0 ? i - 1 : c->pslot};
      bool new_gop = false;
      if (ftypes[fi]) {  // DecompressP
        d.kind = 2;
        c->last_flat = false;
        if (gops.empty()) gops.push_back({(int)fr.size(), 0, c->dec_live ? 1 : 0, 0});
      } else if ((heads[fi] & 15) == 1) {  // flat key frame (:1537-1553)
        const u32 rgb = (heads[fi] >> 8) & 0xFFFFFFu;
        d.kind = 1;
        hipLaunchKernelGGL(k_fill_flat, dim3((g.H * g.S + 255) / 256), dim3(256), 0, st, c->planes.as<u8>(), g, i, rgb);
        new_gop = !(c->last_flat && c->last_flat_rgb == rgb);  // models renewed unless the colour repeats
        c->last_flat = true;
        c->last_flat_rgb = rgb;
        if (!new_gop && gops.empty()) gops.push_back({(int)fr.size(), 0, c->dec_live ? 1 : 0, 0});
      } else {
        d.kind = 0;
        c->last_flat = false;
        new_gop = true;
      }
      if (new_gop) gops.push_back({(int)fr.size(), 0, 0, 0});
      fr.push_back(d);
      gops.back().count++;
      gop_bytes += sizes[fi];
    }
    const size_t ng = gops.size();
    // Key frames of RGB32 output leave through their workgroup's row streamer (a second wave that converts the rows the chain has
    // finished, scpr_wave.hpp) while the chain runs: always when the pictures go to the host (round 4), and since round 5 also
    // when they stay on the device, as long as the chunk's workgroups are at most two to a CU - chain and streamer then have a
    // SIMD each, the decode launch takes what it took (114.9 ms) and k_unpack32's 0.8 ms behind it are gone (headline 133.1 ->
    // 132.1 ms).  With three workgroups on a CU (more than 512 chains in the chunk) streamers share SIMDs with chains and cost
    // more than the kernel they spare (2400 key frames: 449.8 ms against 422.9 + 6.1): those chunks are unpacked afterwards.
    static const size_t dev_streamer_max = getenv("SCPR_DEV_STREAMER_MAX") ? (size_t)atoi(getenv("SCPR_DEV_STREAMER_MAX")) : 512;  // (A/B timing)
    rows_by_streamer = out_is_host || (dev_streamer && ng <= dev_streamer_max);
    streamed_rgb24 = false;
    if (ng) {
      HIPCHK(c->decframes.reserve(fr.size() * sizeof(DecFrame)));
      HIPCHK(c->decgops.reserve(ng * sizeof(DecGop)));
      const bool v2 = c->version == 2;  // range-coder streams: count tables instead of context records (scpr_v2.hpp)
      const size_t state_bytes = v2 ? (size_t)NCOLCTX * V2_COLTAB * 4 : (size_t)NCOLCTX * sizeof(DecRec);
      const size_t blob_bytes = v2 ? sizeof(V2Fixed) : sizeof(FixedBlob);
      HIPCHK(c->decstates.reserve(ng * state_bytes));
      HIPCHK(c->decfixed.reserve(ng * blob_bytes));
      HIPCHK(c->dec_fixed_persist.reserve(blob_bytes));
      HIPCHK(c->dec_colour_persist.reserve(state_bytes));
      if (!v2) HIPCHK(hipMemsetAsync(c->decstates.p, 0, ng * state_bytes, st));  // kind 0 everywhere (RenewI)
      const bool cont = gops[0].load != 0;
      if (cont && !v2 && getenv("SCPR_DEBUG_KEYS")) {  // design aid: every table index of the kept records lies inside the kept part of the arena
        std::vector<u32> hr((size_t)NCOLCTX * DECREC_WORDS);
        HIPCHK(d2h(c, hr.data(), c->dec_colour_persist.p, hr.size() * 4, st));
        HIPCHK(sync_out(c, st));
        size_t bad = 0, dense = 0, firstbad = 0;
        for (size_t r = 0; r < (size_t)NCOLCTX; r++) {
          const u32 kind = hr[r * DECREC_WORDS] & 255u;
          if (kind == 6 || kind == 7) {
            dense++;
            if (hr[r * DECREC_WORDS + 2] == 0 || hr[r * DECREC_WORDS + 2] >= c->dec_arena_used) {
              if (!bad) firstbad = r;
              bad++;
            }
          }
        }
        fprintf(stderr, "[scpr debug] continuing GOP: %zu dense records, arena used %zu, bad indices %zu (first record %zu: kind %u index %u)\n", dense, c->dec_arena_used, bad, firstbad,
                bad ? hr[firstbad * DECREC_WORDS] & 255u : 0u, bad ? hr[firstbad * DECREC_WORDS + 2] : 0u);
        if (bad) return SCPR_E_DEVICE;
      }
      if (cont) {  // the first GOP continues the state kept from the previous call
        HIPCHK(hipMemcpyAsync(c->decstates.p, c->dec_colour_persist.p, state_bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(c->decfixed.p, c->dec_fixed_persist.p, blob_bytes, hipMemcpyDeviceToDevice, st));
      } else {
        c->dec_arena_used = 1;
        c->dec_live_bytes = 0;
      }
      // Tables alive at the end of this chunk.  The true bound is 12288 per GOP.  In practice a context goes dense once, after at
      // least 16 of its symbols, each of which cost the stream about a byte or more (raw bytes while nothing repeats, one-slot
      // intervals of a small table after that: ans_contexts.cpp:3-50; a table per 12 bytes of stream has never been seen
      // exceeded) - counted over the whole GOP, earlier calls included.  The first attempt of a chunk of fresh GOPs gets that
      // estimate (round 2 gave it a flat 1024 tables per GOP: a GOP of 300 desktop frames overflowed at frame ~220 and was decoded
      // twice, 8.2 s instead of 4.2), the second attempt - and a chunk that continues a GOP, which cannot be decoded twice: its
      // run changes the tables that GOP already owns - the true bound.
      const size_t true_bound = ng * (size_t)NCOLCTX;
      size_t estimate = std::min<size_t>(true_bound, (size_t)((c->dec_live_bytes + gop_bytes) / 12) + 8 * ng);
      if (const char* dbg = getenv("SCPR_DEBUG_DEC_ARENA")) estimate = std::min<size_t>(estimate, (size_t)strtoull(dbg, nullptr, 0));  // (tests reach the second attempt)
      const size_t budget = v2 ? 0 : (attempt || cont) ? true_bound : estimate;
      // (the arena holds table 0 and the live GOP's tables when a chunk starts: compact_tables after every chunk with several GOPs;
      // + 1: the sink of an overflow, + slack)
      const size_t arena_cap = std::min<size_t>(1 + true_bound, c->dec_arena_used + budget) + 1 + 64;
      HIPCHK(c->dec_arena.reserve_keep(arena_cap * sizeof(DenseTab), c->dec_arena_used * sizeof(DenseTab), st));
      c->h_dec_top0 = (u32)c->dec_arena_used;  // (a member: the source of an asynchronous copy must outlive the call)
      HIPCHK(h2d(c, c->dec_arena_top.p, &c->h_dec_top0, 4, st));
      HIPCHK(h2d(c, c->decframes.p, fr.data(), fr.size() * sizeof(DecFrame), st));
      HIPCHK(h2d(c, c->decgops.p, gops.data(), ng * sizeof(DecGop), st));
      Arena ar{c->dec_arena.as<DenseTab>(), c->dec_arena_top.as<u32>(), (u32)arena_cap - 1u, c->err.as<u32>()};  // (the last table allocated is the sink)
      // LDS ring of 32-bit pixels: the predictors look back one row + 1 pixel, a finished row is flushed at most
      // one run after it ends, and a run writes up to 255 pixels ahead: a power of two >= W + 512 pixels.
      // (Keeping static + dynamic LDS under 80 KiB lets two GOPs share a CU.)
      int ring = 4096;
      while (ring < 4 * (g.W + 512)) ring <<= 1;
      const int nblocks = ((g.W + 15) / 16) * ((g.H + 15) / 16);
      bool has_p = false;
      for (const DecFrame& d : fr) has_p |= d.kind == 2;
      int dyn = ring + (has_p ? ((nblocks + 15) & ~15) : 0);  // + one byte per block for P-frames
      // + the dense-table cache (scpr_wave.hpp, WaveModel::tab_of) with what LDS is left: of the whole CU (160 KiB) when every
      // GOP gets a CU to itself anyway, of half a CU otherwise (two GOPs per CU: a batch of key frames lives on GOPs in flight)
      int ndc = 0;
      const int dcache_off = dyn;
      if (!v2) {
        const size_t lds_static = has_p ? sizeof(WaveLds) : offsetof(WaveLds, fp);  // (a batch of key frames: without the P-frame tables)
        const size_t lds_budget = (ng <= 256 ? 160 * 1024 : (has_p || ng <= 512) ? 80 * 1024 : 53 * 1024) - lds_static - 1024;  // one, two or three workgroups per CU
        while (ndc < 32 && lds_budget > (size_t)dyn && (size_t)dyn + (size_t)(ndc ? 2 * ndc : 1) * sizeof(DenseTab) <= lds_budget) ndc = ndc ? 2 * ndc : 1;
        if (ndc < 16) ndc = 0;  // a handful of slots would only be evicted all the time (a miss moves two tables, an uncached symbol 32 bytes per lane)
        dyn += ndc * (int)sizeof(DenseTab);
      }
      const u8* pk = (const u8*)d_packets;
      const u8* pk_end = pk + offs[nframes];  // nothing is read at or past this address (the reader supplies 0xFF there)
      // the chunk's frames in the host's buffer, for the chains to send their rows to (RGB32 of version 3 / 4 streams)
      // (RGB24 output in the plane's own layout - pitch = stride, the k_copy_planes case - is streamed too when the chunk is nothing
      // but coded key frames and stays on the device: BASELINE configs[4])
      bool all_coded_keys = true;
      for (int i = 0; i < n; i++) all_coded_keys &= !ftypes[f0 + i] && (heads[f0 + i] & 15) != 1;
      streamed_rgb24 = rows_by_streamer && !out_is_host && !v2 && c->bpp == 3 && pitch == g.S && ((size_t)d_frames_out & 3) == 0 && all_coded_keys;
      u8* hout = (rows_by_streamer && !v2 && (c->bpp == 4 || streamed_rgb24)) ? (u8*)d_frames_out + (size_t)f0 * pitch * g.H : nullptr;
      if (v2) {
        auto kern = has_p ? k_decode_gop_v2<true> : k_decode_gop_v2<false>;
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
        hipLaunchKernelGGL(kern, dim3((unsigned)ng), dim3(64), dyn, st, pk, pk_end, c->decframes.as<DecFrame>(), c->decgops.as<DecGop>(), c->planes.as<u8>(), g,
                           c->decstates.as<u32>(), c->err.as<u32>(), ring, c->decfixed.as<V2Fixed>(), (int)c->prm.high_range_x, (int)c->prm.high_range_y);  // the caller's range, unclamped (:76-77)
      } else {
        auto kern = has_p ? k_decode_gop_w<true> : k_decode_gop_w<false>;  // key-frame-only GOPs: smaller kernel (instruction cache)
        HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
        // P-frame GOPs run as a workgroup: the chain's wave + helper waves for the bulk copies (scpr_wave.hpp, helper_loop) -
        // eight waves when every GOP has a CU to itself (six helpers: the wave that would share the chain's SIMD leaves at
        // once), four when CUs are shared (three helpers; two waves per SIMD at most: the chain keeps its 256 registers)
        static const bool no_helpers = getenv("SCPR_NO_HELPERS") != nullptr;  // (design aid: the chain's wave alone, for the counters of tools/decoder_pmc2.sh)
        // (a batch of key frames for the host: a second wave per workgroup sends the rows - row_streamer)
        const unsigned threads = has_p ? (no_helpers ? 64u : ng <= 256 ? 512u : 256u) : hout ? 128u : 64u;
        hipLaunchKernelGGL(kern, dim3((unsigned)ng), dim3(threads), dyn, st, pk, pk_end, c->decframes.as<DecFrame>(), c->decgops.as<DecGop>(), c->planes.as<u8>(), g,
                           c->decstates.as<DecRec>(), ar, c->f0, c->err.as<u32>(), ring, c->decfixed.as<FixedBlob>(), (int)std::min<u32>(c->prm.high_range_x, 256),
                           (int)std::min<u32>(c->prm.high_range_y, 256), ndc, dcache_off, hout, pitch, n - 1, c->bpp);
      }
    }
    stage_end(c, ST_DECODE);
    u32 atop = 0;
    HIPCHK(d2h(c, errv, c->err.p, 32, st));
    if (ng) HIPCHK(d2h(c, &atop, c->dec_arena_top.p, 4, st));
    HIPCHK(sync_out(c, st));  // also covers fr / gops (host memory)
    HIPCHK(hipGetLastError());         // a kernel that could not be launched
    // A record that names a table beyond the arena's sink (bit 16; refused by the table cache, never followed) is an error of
    // whatever attempt shows it, and always said: an overflow hands out the sink, so no state the decoder makes by itself holds
    // such an index.
    if (errv[0] & 16) fprintf(stderr, "[scpr] decoder: a colour record named a dense table outside the arena (flags %u, attempt %d, %zu GOPs, continued %d)\n", errv[0], attempt, ng,
                              ng ? gops[0].load : 0);
    if ((errv[0] & 1) && attempt == 0 && !(ng && gops[0].load)) continue;  // the estimate was too small for this stream: once more with the true bound
    if (ng && !(errv[0] & 5)) {
      // keep the state of the last GOP for the next call; with one GOP in the chunk its tables stay where they are
      // (top read back: the arena does not grow with the number of calls), with several the next call's first
      // P-frame would need the last GOP's tables only, which sit anywhere below the top
      const bool v2 = c->version == 2;
      const size_t state_bytes = v2 ? (size_t)NCOLCTX * V2_COLTAB * 4 : (size_t)NCOLCTX * sizeof(DecRec);
      const size_t blob_bytes = v2 ? sizeof(V2Fixed) : sizeof(FixedBlob);
      HIPCHK(hipMemcpyAsync(c->dec_colour_persist.p, (const u8*)c->decstates.p + (ng - 1) * state_bytes, state_bytes, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(c->dec_fixed_persist.p, (const u8*)c->decfixed.p + (ng - 1) * blob_bytes, blob_bytes, hipMemcpyDeviceToDevice, st));
      c->dec_arena_used = std::max<size_t>(atop, 1);
      c->dec_live_bytes = ng > 1 ? gop_bytes : c->dec_live_bytes + gop_bytes;  // (an upper bound for the GOP that is live after this chunk)
      c->dec_live = true;
      if (ng > 1 && !v2 && atop > 1) {  // only the last GOP's tables are needed again: to the bottom (behind table 0), the rest is dropped
        u32 ntop = 1;
        int rc = compact_tables(c, c->dec_arena, c->dec_arena2, c->dec_arena_top, c->dec_colour_persist.p, DECREC_WORDS, -1, 0u, 1u, atop, &ntop);
        if (rc != SCPR_OK) return rc;
        c->dec_arena_used = std::max<size_t>(ntop, 1);
      }
    }
    break;
    }
    HIPCHK(hipMemcpyAsync(c->planes.as<u8>() + (size_t)c->pslot * g.plane_stride, c->planes.as<u8>() + (size_t)(n - 1) * g.plane_stride, g.plane_stride, hipMemcpyDeviceToDevice, st));
    stage_begin(c, ST_UNPACK);
    u8* out = (u8*)d_frames_out + (size_t)f0 * pitch * g.H;
    if (c->bpp == 4 && rows_by_streamer && c->version != 2) {
      // a chunk of key frames only: the coded ones are at the host already (their workgroups' row streamers sent them); a chunk
      // with P-frames runs the workgroup form of the kernel, which sends nothing
      bool chunk_has_p = false;
      for (int i = 0; i < n; i++) chunk_has_p |= ftypes[f0 + i] != 0;
      std::vector<int> rest;
      for (int i = 0; i < n; i++) {
        const bool flat_key = !ftypes[f0 + i] && (heads[f0 + i] & 15) == 1;
        if (chunk_has_p || flat_key) rest.push_back(i);
      }
      if (!rest.empty()) {
        HIPCHK(c->hb_list.reserve(rest.size() * 4));
        HIPCHK(h2d(c, c->hb_list.p, rest.data(), rest.size() * 4, st));
        dim3 gr((g.H * ((g.W + 3) >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), (unsigned)rest.size());
        hipLaunchKernelGGL(k_unpack32_list, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g, pitch, c->hb_list.as<int>());
        HIPCHK(sync_out(c, st));  // (`rest` is host memory)
      }
    } else if (c->bpp == 4) {
      dim3 gr((g.H * ((g.W + 3) >> 2) + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_unpack32, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g, pitch);
    } else if (c->bpp == 3 && streamed_rgb24) {
      // (the chains' row streamers wrote every picture of the chunk)
    } else if (c->bpp == 3 && pitch == g.S && ((size_t)out & 3) == 0) {
      dim3 gr((unsigned)(((size_t)g.H * g.S / 4 + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS)), n);
      hipLaunchKernelGGL(k_copy_planes, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g);
    } else {
      dim3 gr((g.H * g.W + 256 * PACK_ITEMS - 1) / (256 * PACK_ITEMS), n);
      hipLaunchKernelGGL(k_unpack_rows, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g, pitch, c->bpp, c->rs, c->gs, c->bs);
    }
    stage_end(c, ST_UNPACK);
    HIPCHK(sync_out(c, st));
    HIPCHK(hipGetLastError());
    timing_collect(c);
    const u32 err = errv[0];
    if (err & (1 | 16)) return SCPR_E_DEVICE;  // (an overflow of the arena in a chunk that could not be decoded again also sets 4)
    if (err & 4) return SCPR_E_STREAM;
    done += n;
    f0 += n;
  }
  return done;
}

int scpr_decompress_batch(scpr_codec* c, const void* d_packets, const uint32_t* sizes, const int* ftypes, int nframes, void* d_frames_out, int pitch) {
  if (!c || !c->inited || !d_packets || !sizes || !ftypes || !d_frames_out || nframes < 0) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  return decompress_core(c, d_packets, sizes, ftypes, nframes, d_frames_out, pitch, false);
}

// ---- batch entry points with HOST pointers ------------------------------------------------------------------------------------
// The reference's boundary hands over host memory (ScreenCodec::CompressFrame / DecompressFrame, screencap.cpp:1632, :1695).  These
// are the batch calls in that shape, with the PCIe crossings taken beside the kernels instead of around them:
//   compress: the frames come over in sub-batches on a copy stream, sub-batch k + 1 while sub-batch k is being coded; the packets
//             (~1 % of the frames) are written by the gather kernel straight into the host's buffer;
//   decompress: the packets go over first (small); every coded key frame's rows leave for the host's buffer as the chain
//             finishes them (decompress_core), so the 8 MB per picture cross while the 110 ms of chain run, not after.
// The fast path wants PINNED host memory (hipHostMalloc / hipHostRegister / torch pin_memory, or scpr_host_pin below for buffers the
// caller reuses - a capture loop does).  Pageable memory works too, through the runtime's staging copies: same results, no overlap.
// Where host memory `p` can be reached from the device without a copy by the runtime: memory the runtime has pinned (hipHostMalloc,
// hipHostRegister by the caller, torch's pin_memory) or that the caller has handed to scpr_host_pin.  Anything else - malloc,
// numpy - is NOT registered behind the caller's back (a registration outlives the memory it names unless somebody takes it back):
// the calls then go through the runtime's own staging copies, with the same results and no overlap.
static int host_view(scpr_codec* c, const void* p, size_t bytes, void** dev) {
  *dev = nullptr;
  if (!bytes) return SCPR_E_PARAM;
  for (const auto& r : c->host_ranges)
    if ((const u8*)p >= (const u8*)r.base && (const u8*)p + bytes <= (const u8*)r.base + r.bytes) {
      void* d0 = nullptr;
      if (hipHostGetDevicePointer(&d0, r.base, 0) == hipSuccess && d0) {
        *dev = (u8*)d0 + ((const u8*)p - (const u8*)r.base);
        return SCPR_OK;
      }
      (void)hipGetLastError();
    }
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost) {
    if (hipHostGetDevicePointer(dev, const_cast<void*>(p), 0) == hipSuccess && *dev) return SCPR_OK;
  }
  (void)hipGetLastError();  // (an unknown pointer is an "invalid value" to the query: not an error of ours, and not to be found by the next caller of hipGetLastError)
  *dev = nullptr;
  return SCPR_E_DEVICE;
}

int scpr_host_pin(scpr_codec* c, void* p, size_t bytes) {
  if (!c || !p || !bytes) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  for (const auto& r : c->host_ranges)
    if (r.base == p && r.bytes == bytes) return SCPR_OK;
  if (hipHostRegister(p, bytes, hipHostRegisterMapped) == hipSuccess) {
    c->host_ranges.push_back({p, bytes, true});
    return SCPR_OK;
  }
  (void)hipGetLastError();
  // somebody has pinned it already (another codec of the process, the caller): fine, and not this codec's to release
  hipPointerAttribute_t a{};
  void* d0 = nullptr;
  if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && hipHostGetDevicePointer(&d0, p, 0) == hipSuccess && d0) {
    c->host_ranges.push_back({p, bytes, false});
    return SCPR_OK;
  }
  (void)hipGetLastError();
  return SCPR_E_DEVICE;
}

int scpr_host_unpin(scpr_codec* c, void* p) {
  if (!c || !p) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  for (size_t i = 0; i < c->host_ranges.size(); i++)
    if (c->host_ranges[i].base == p) {
      (void)hipStreamSynchronize(c->stream);
      if (c->stream3) (void)hipStreamSynchronize(c->stream3);
      const bool owned = c->host_ranges[i].owned;
      c->host_ranges.erase(c->host_ranges.begin() + (long)i);
      // (a registration another codec of the process made over the same memory and has released already is not an error of
      // this call: the codec no longer uses the range, which is what was asked)
      if (owned && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
      return SCPR_OK;
    }
  return SCPR_E_PARAM;
}

int64_t scpr_compress_batch_host(scpr_codec* c, const void* h_frames, int nframes, int* ftypes, int loss, void* h_out, size_t out_capacity, uint32_t* sizes) {
  if (!c || !c->inited || !h_frames || !ftypes || !h_out || !sizes || nframes < 0) return SCPR_E_PARAM;
  if (c->crashed) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  if (c->have_codec && c->version == 2) return SCPR_E_BAD_VERSION;
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  if (loss != c->last_loss) setup_loss(c, loss);
  timing_reset(c);
  if (nframes == 0) return 0;
  const Geom& g = c->g;
  const size_t frame_bytes = (size_t)c->pitch_in * g.H;
  if (!c->stream3) {
    // A stream of ANOTHER PRIORITY than the codec's two: the runtime maps streams onto a few hardware queues, streams of one
    // priority share them round robin, and the third stream of this codec landed on the queue of the second - its kernels (the
    // fixed-model chains) then waited for the whole upload in front of them (rocprofv3 --memory-copy-trace: a 10 ms hole in every
    // sub-batch).  Queues of different priorities are different queues.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (least != greatest) HIPCHK(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, least));
    else HIPCHK(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming));
  }
  // the frames: from pinned memory the copies below are DMA transfers that run beside the kernels; from pageable memory the
  // runtime stages them itself (same results, little overlap)
  // the packets: straight into the host's buffer when it can be mapped, else into a device buffer that is copied back at the end
  void* out_dev = nullptr;
  const bool out_mapped = host_view(c, h_out, out_capacity, &out_dev) == SCPR_OK && out_dev;
  // (not mapped: a device buffer for what the reference's own caller provides per frame, W*H*6 - CompressGetSize, screenpressor.cpp:386-388;
  // should noise-like pictures need more and the host's buffer have it, the call is made once more with the closed-form worst case)
  const size_t nblk0 = (size_t)((g.W + 15) / 16) * ((g.H + 15) / 16);
  const size_t most = (size_t)g.W * g.H * 6 + 64, worst = (size_t)packet_bound(4, 5ull * g.NP + 16ull * nblk0 + 16);
  size_t dev_cap = std::min(out_capacity, most * (size_t)nframes);
  for (int attempt = 0;; attempt++) {
  if (!out_mapped) {
    HIPCHK(c->hb_packets.reserve(dev_cap + 64));
    out_dev = c->hb_packets.p;
  }
  // sub-batches: the call ends one sub-batch's coding after the last upload, so the smaller the better - until a sub-batch's
  // coding (its front end + ONE rANS block chain, 3.7 ms in the scalar form, + the call's read-backs: 0.045 ms per 1080p frame +
  // 4.7 ms) takes longer than the next one's upload (0.145 ms per frame at 57 GB/s): about 400 MB of frames, 48 at 1080p, 12 at 4K
  // (tools/exp_host.py: 300 frames of 1080p in sub-batches of 100 / 75 / 60 / 50 / 43 / 30: 52.8 / 52.0 / 51.2 / 50.9 / 54.4 / 68.3 ms)
  const int sub_env = getenv("SCPR_HOST_SUB") ? atoi(getenv("SCPR_HOST_SUB")) : 0;  // (tests and tuning: frames per sub-batch)
  const int sub_auto = (int)std::min<size_t>((size_t)nframes, ((size_t)400 << 20) / frame_bytes + 1);
  const int sub = std::max(1, std::min(nframes, sub_env > 0 ? sub_env : sub_auto));
  const int nsub = (nframes + sub - 1) / sub;
  HIPCHK(c->hb_frames.reserve((size_t)std::min(nframes, 2 * sub) * frame_bytes));
  EncTxn txn;
  if ((rc = txn_begin(c, txn, ftypes, nframes, out_mapped ? out_capacity : dev_cap)) != SCPR_OK) return rc;
  auto upload = [&](int k) {
    const int a = k * sub, n = std::min(sub, nframes - a);
    hipError_t e = hipMemcpyAsync(c->hb_frames.as<u8>() + (size_t)(k & 1) * sub * frame_bytes, (const u8*)h_frames + (size_t)a * frame_bytes, (size_t)n * frame_bytes, hipMemcpyHostToDevice, c->stream3);
    if (e == hipSuccess) e = hipEventRecord(c->ev_in[k & 1], c->stream3);
    return e;
  };
  // one way out for every failure from here on: the copy stream may still be reading the caller's frames, and sub-batches that
  // were coded have moved the codec (enc_failed)
  auto fail = [&](int64_t r) -> int64_t {
    (void)hipStreamSynchronize(c->stream3);
    return enc_failed(c, txn, ftypes, r);
  };
#define HB_CHK(x)                                          \
  do {                                                     \
    if ((x) != hipSuccess) return fail(SCPR_E_DEVICE);     \
  } while (0)
  HB_CHK(upload(0));
  int64_t written = 0;
  const size_t cap = out_mapped ? out_capacity : dev_cap;
  for (int k = 0; k < nsub; k++) {
    const int a = k * sub, n = std::min(sub, nframes - a);
    if (k + 1 < nsub) HB_CHK(upload(k + 1));  // (its half of the staging was read by sub-batch k - 1, whose call has returned)
    HB_CHK(hipStreamWaitEvent(c->stream, c->ev_in[k & 1], 0));
    const int64_t r = compress_core(c, c->hb_frames.as<u8>() + (size_t)(k & 1) * sub * frame_bytes, n, ftypes + a, (u8*)out_dev + written, cap - (size_t)written, sizes + a, txn.hs0);
    if (r < 0) {
      if (r != SCPR_E_CAPACITY) return fail(r);
      (void)hipStreamSynchronize(c->stream3);
      const int64_t r2 = txn_refuse(c, txn, ftypes, SCPR_E_CAPACITY);
      if (r2 != SCPR_E_CAPACITY || out_mapped || attempt || dev_cap >= out_capacity) return r2;
      written = -1;  // the device buffer was the limit, not the host's: once more with room for the worst case
      break;
    }
    written += r;
  }
  if (written < 0) {
    dev_cap = std::min(out_capacity, worst * (size_t)nframes);
    continue;
  }
  if (!out_mapped && written > 0) HB_CHK(hipMemcpy(h_out, out_dev, (size_t)written, hipMemcpyDeviceToHost));
#undef HB_CHK
  return written;
  }
}

int scpr_decompress_batch_host(scpr_codec* c, const void* h_packets, const uint32_t* sizes, const int* ftypes, int nframes, void* h_frames_out, int pitch) {
  if (!c || !c->inited || !h_packets || !sizes || !ftypes || !h_frames_out || nframes < 0) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  if (nframes == 0) return 0;
  size_t total = 0;
  for (int i = 0; i < nframes; i++) total += sizes[i];
  const size_t ob = (size_t)pitch * c->prm.height * (size_t)nframes;
  HIPCHK(c->hb_packets.reserve(total + 64));
  HIPCHK(hipMemcpyAsync(c->hb_packets.p, h_packets, total, hipMemcpyHostToDevice, c->stream));  // ~1 % of the pictures: not worth a pipeline
  void* out_dev = nullptr;
  if (host_view(c, h_frames_out, ob, &out_dev) == SCPR_OK && out_dev) return decompress_core(c, c->hb_packets.p, sizes, ftypes, nframes, out_dev, pitch, true);
  // the host's buffer cannot be mapped: decode to the device, copy back
  HIPCHK(c->hb_frames.reserve(ob));
  const int r = decompress_core(c, c->hb_packets.p, sizes, ftypes, nframes, c->hb_frames.p, pitch, false);
  if (r > 0) HIPCHK(hipMemcpy(h_frames_out, c->hb_frames.p, (size_t)pitch * c->prm.height * (size_t)r, hipMemcpyDeviceToHost));
  return r;
}

// ---- per-frame entry points with host pointers (the reference's own shape) ----
int scpr_compress_frame(scpr_codec* c, const void* src, void* dst, int dst_len, int* ftype, int loss) {
  if (!c || !c->inited || !src || !dst || !ftype) return SCPR_E_PARAM;
  if (c->crashed) return 0;
  if (dst_len < 0) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  // (the caller's room is the batch call's capacity: a packet that does not fit is refused there with the codec left as it was)
  const size_t fb = (size_t)c->pitch_in * c->g.H, cap = std::min<size_t>((size_t)c->g.W * c->g.H * 6 + 64, (size_t)dst_len);
  HIPCHK(c->hoststage_in.reserve(fb));
  HIPCHK(c->hoststage_out.reserve(cap + 64));
  HIPCHK(h2d(c, c->hoststage_in.p, src, fb, c->stream));
  uint32_t sz = 0;
  int64_t r = scpr_compress_batch(c, c->hoststage_in.p, 1, ftype, loss, c->hoststage_out.p, cap, &sz);
  if (r <= 0) return (int)r;
  HIPCHK(hipMemcpy(dst, c->hoststage_out.p, (size_t)r, hipMemcpyDeviceToHost));
  return (int)r;
}

int scpr_decompress_frame(scpr_codec* c, const void* src, int src_len, void* dst, int pitch, int ftype) {
  if (!c || !c->inited || !src || !dst || src_len < 1) return SCPR_E_PARAM;
  if (c->crashed && ftype > 0) return 0;
  if (!c->have_codec && ftype > 0) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  pin_reset(c);
  const size_t ob = (size_t)pitch * c->prm.height;
  HIPCHK(c->hoststage_in.reserve((size_t)src_len + 64));
  HIPCHK(c->hoststage_out.reserve(ob));
  HIPCHK(h2d(c, c->hoststage_in.p, src, (size_t)src_len, c->stream));
  uint32_t sz = (uint32_t)src_len;
  int r = scpr_decompress_batch(c, c->hoststage_in.p, &sz, &ftype, 1, c->hoststage_out.p, pitch);
  if (r < 0) return r;
  if (r != 1) return 0;
  HIPCHK(hipMemcpy(dst, c->hoststage_out.p, ob, hipMemcpyDeviceToHost));
  return 1;
}

int scpr_last_timing(scpr_codec* c, float* total_ms, float* stage_ms, int cap) {
  if (!c) return 0;
  if (total_ms) *total_ms = c->total_ms;
  for (int s = 0; s < ST_COUNT && s < cap; s++) stage_ms[s] = c->stage_ms[s];
  return ST_COUNT;
}

#ifdef SCPR_PROFILE
// design work only (libscpr_amd_prof.so): the records of the long colour chains since the last call (tools/profile_chains.py)
extern "C" int scpr_debug_chains(unsigned* out, int cap) {
  unsigned n = 0, zero = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(scpr::g_chainrec_n), 4) != hipSuccess) return -1;
  const unsigned m = n < 8192u ? n : 8192u;
  const unsigned take = m < (unsigned)cap ? m : (unsigned)cap;
  if (take && hipMemcpyFromSymbol(out, HIP_SYMBOL(scpr::g_chainrec), (size_t)take * 32) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(scpr::g_chainrec_n), &zero, 4) != hipSuccess) return -1;
  return (int)n;
}
// design work only (libscpr_amd_prof.so): s_memtime ticks per decoder section summed over all GOPs since the last call
extern "C" int scpr_debug_profile(unsigned long long* out) {
  unsigned long long z[24] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(scpr::g_prof), sizeof z) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(scpr::g_prof), z, sizeof z) != hipSuccess) return -1;
  return 24;
}
#endif
#ifdef SCPR_PROFILE
// design work only: colour symbols of the decoder by class since the last call (ticks, count) x 8 - scpr_wave.hpp, WaveDec::cls_end
extern "C" int scpr_debug_cprof(unsigned long long* out) {
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(scpr::g_cprof), sizeof z) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(scpr::g_cprof), z, sizeof z) != hipSuccess) return -1;
  return 32;
}
#endif
int scpr_debug_colour_chain(int device, const uint8_t* syms, int n, int f0, uint16_t* out) {
  if (hipSetDevice(device) != hipSuccess) return SCPR_E_DEVICE;
  std::vector<u32> keys(n), vals(n);
  for (int i = 0; i < n; i++) {
    keys[i] = syms[i];
    vals[i] = (u32)i;
  }
  const u32 cst[2] = {0, (u32)n}, lst[1] = {0}, cnt[1] = {1}, zero[2] = {0, 0};
  DevBuf dk, dv, dc, dl, dn, de, da, dt;
  const size_t acap = 4;
  HIPCHK(dk.reserve((size_t)n * 4));
  HIPCHK(dv.reserve((size_t)n * 4));
  HIPCHK(dc.reserve(8));
  HIPCHK(dl.reserve(4));
  HIPCHK(dn.reserve(4 * CHAIN_CLASSES));
  HIPCHK(de.reserve((size_t)n * 4));
  HIPCHK(da.reserve(acap * sizeof(DenseTab)));
  HIPCHK(dt.reserve(8));
  HIPCHK(hipMemcpy(dk.p, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dv.p, vals.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dc.p, cst, 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dl.p, lst, 4, hipMemcpyHostToDevice));
  const u32 cnt4[CHAIN_CLASSES] = {cnt[0], 0, 0, 0};
  HIPCHK(hipMemcpy(dn.p, cnt4, sizeof cnt4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, zero, 8, hipMemcpyHostToDevice));
  Arena ar{da.as<DenseTab>(), dt.as<u32>(), (u32)acap, dt.as<u32>() + 1};
  ChainPersist cp{nullptr, nullptr, 0, 0, 0, 0};  // nothing loaded, nothing kept
  hipLaunchKernelGGL(k_colour_chain_w, dim3(1), dim3(64), 0, 0, dk.as<u32>(), dv.as<u32>(), dc.as<u32>(), dl.as<u32>(), dn.as<u32>(), 1u, f0, ar, cp, de.as<u32>());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, de.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (DevBuf* b : {&dk, &dv, &dc, &dl, &dn, &de, &da, &dt}) b->release();
  return n;
}

// Test hook: the next scpr_compress_batch fails on purpose.  1: returns SCPR_E_DEVICE between the read-backs of its results and
// their hand-over (what a HIP error there leaves behind); 2: two colour keys change places behind the partition by context (the
// order proof of k_chain_starts must end the call with SCPR_E_DEVICE, no chain is followed); 3: see scpr_rans_s.hpp.
int scpr_debug_inject(scpr_codec* c, int what) {
  if (!c) return SCPR_E_PARAM;
  if (!c->dbg_armed) return SCPR_E_PARAM;  // (a codec created without SCPR_ENABLE_DEBUG_INJECT=1 cannot be made to fail)
  c->dbg_inject = what;
  return SCPR_OK;
}

int scpr_debug_rans_recoded(scpr_codec* c) { return c ? c->rans_recoded : SCPR_E_PARAM; }

int scpr_debug_arena(scpr_codec* c, uint64_t* enc_bytes, uint64_t* dec_bytes) {
  if (!c) return SCPR_E_PARAM;
  if (enc_bytes) *enc_bytes = (uint64_t)c->arena.cap;
  if (dec_bytes) *dec_bytes = (uint64_t)c->dec_arena.cap;
  return SCPR_OK;
}

int64_t scpr_debug_entries(scpr_codec* c, uint16_t* out, int64_t cap) {
  if (!c) return SCPR_E_PARAM;
  int64_t n = std::min<int64_t>(c->dbg_entries, cap);
  if (out && n > 0 && hipMemcpy(out, c->entries.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return SCPR_E_DEVICE;
  return c->dbg_entries;
}

}  // extern "C"
