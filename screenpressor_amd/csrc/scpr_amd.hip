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
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/scpr_amd.h"
#include "scpr_kernels.hpp"
#include "scpr_wave.hpp"

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

struct DevBuf {  // grow-only device allocation
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
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

enum Stage { ST_PACK, ST_CLASSIFY, ST_SCAN, ST_SYMBOLS, ST_SORT, ST_FIXED, ST_COLOUR, ST_RANS, ST_GATHER, ST_DECODE, ST_UNPACK, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"pack", "classify", "scan", "symbols", "sort", "fixed_chain", "colour_chain", "rans", "gather", "decode", "unpack"};

}  // namespace

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
  int slots = 0;  // frames processed per chunk
  // per-slot worst-case buffers
  DevBuf planes, exitmap, entry, runrec, tilecnt, tileoff, hdrrec, hdrcnt, frametot;
  // per-batch buffers
  DevBuf flags, slotlist, genlist, bases, totals, runs, runpos, keys[2], vals[2], hist, cstart, sorttmp, scantmp, entries, ranges;
  DevBuf rblocks, rscratch, rsize, packets, pktoff, blkdst, outsizes, total64, arena, arena_top, err, rcp;
  DevBuf decframes, decstates, hoststage_in, hoststage_out, chainlists, chaincounts;
  // timing
  hipEvent_t ev[ST_COUNT + 1][2];
  bool ev_used[ST_COUNT];
  float stage_ms[ST_COUNT];
  float total_ms = 0;
  int64_t dbg_entries = 0;
};

static void stage_begin(scpr_codec* c, int s) {
  if (!c->ev_used[s]) (void)hipEventRecord(c->ev[s][0], c->stream);
}
static void stage_end(scpr_codec* c, int s) {
  (void)hipEventRecord(c->ev[s][1], c->stream);
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

static int ensure_codec(scpr_codec* c, int version) {  // CreateCodec + CScreenCapt::Init, screencap.cpp:1587-1617, :69-124
  if (c->have_codec) return SCPR_OK;
  if (version < 3 || version > 4) return SCPR_E_BAD_VERSION;
  const scpr_params& p = c->prm;
  if (p.bits_per_pixel != 16 && p.bits_per_pixel != 24 && p.bits_per_pixel != 32) return SCPR_E_BAD_VERSION;
  if (p.width < 3 || p.height < 2 || p.width > 8000 || p.workers < 1 || p.height < 2 * p.workers) return SCPR_E_PARAM;
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
  // chunk size: keep the per-slot worst-case buffers within ~6 GiB
  size_t per_slot = (size_t)g.plane_stride + (size_t)g.ntiles * (512 + 2 + TILE * 4 + 16) + (size_t)(g.W + 2) * 4;
  size_t s = (12ull << 30) / per_slot;
  c->slots = (int)std::min<size_t>(std::max<size_t>(s, 1), 512);
  const size_t ns = (size_t)c->slots + 1;  // +1: slot `slots` holds the previous frame of the stream
  HIPCHK(c->planes.reserve(ns * g.plane_stride));
  HIPCHK(hipMemsetAsync(c->planes.p, 0, ns * g.plane_stride, c->stream));
  HIPCHK(c->exitmap.reserve(ns * g.ntiles * 512));
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
  HIPCHK(c->totals.reserve(64));
  HIPCHK(c->ranges.reserve(ns * sizeof(GenRange)));
  HIPCHK(c->arena_top.reserve(16));
  HIPCHK(c->err.reserve(64));
  HIPCHK(c->total64.reserve(16));
  // reciprocal table for every frequency on the 12-bit scale
  std::vector<RansRcp> tab(kProbScale + 1);
  for (u32 f = 0; f <= (u32)kProbScale; f++) tab[f] = rans_rcp(f ? f : 1);
  HIPCHK(c->rcp.reserve(tab.size() * sizeof(RansRcp)));
  HIPCHK(hipMemcpyAsync(c->rcp.p, tab.data(), tab.size() * sizeof(RansRcp), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  setup_loss(c, (int)p.loss);
  c->have_codec = true;
  return SCPR_OK;
}

// ---------------------------------------------------------------------------
// Encode one chunk of key frames (all in `slot_ids`, generation = index).
// ---------------------------------------------------------------------------
struct ChunkFrame {
  int slot;
  int kind;      // 0 coded I, 1 flat, 2 coded P, 3 unchanged P
  u32 hdr, hdr_len;
};

static int encode_intra_frames(scpr_codec* c, const std::vector<int>& islots, std::vector<FrameBase>& hb) {
  const Geom& g = c->g;
  const int n = (int)islots.size();
  hb.clear();
  if (!n) return SCPR_OK;
  hipStream_t st = c->stream;
  HIPCHK(hipMemcpyAsync(c->slotlist.p, islots.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
  std::vector<int> gens(n);
  for (int i = 0; i < n; i++) gens[i] = i;
  HIPCHK(hipMemcpyAsync(c->genlist.p, gens.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
  const int* d_slots = c->slotlist.as<int>();
  const u8* planes = c->planes.as<u8>();

  if (c->loss_mask != 0xFFFFFFFFu) {
    dim3 gl((g.H * (g.S >> 2) + 255) / 256, n);
    hipLaunchKernelGGL(k_loss, gl, dim3(256), 0, st, c->planes.as<u8>(), g, d_slots, c->loss_mask, c->corr_mask);
  }
  stage_begin(c, ST_CLASSIFY);
  hipLaunchKernelGGL(k_tiles<false>, dim3(g.ntiles, n), dim3(256), 0, st, planes, g, d_slots, c->exitmap.as<u8>(), (const u8*)nullptr, (u32*)nullptr, (u32*)nullptr);
  hipLaunchKernelGGL(k_entries, dim3(n), dim3(64), 0, st, c->exitmap.as<u8>(), c->entry.as<u8>(), g, d_slots);
  hipLaunchKernelGGL(k_tiles<true>, dim3(g.ntiles, n), dim3(256), 0, st, planes, g, d_slots, (u8*)nullptr, c->entry.as<u8>(), c->runrec.as<u32>(), c->tilecnt.as<u32>());
  hipLaunchKernelGGL(k_header, dim3(n), dim3(64), 0, st, planes, g, d_slots, c->hdrrec.as<u32>(), c->hdrcnt.as<u32>());
  stage_end(c, ST_CLASSIFY);
  stage_begin(c, ST_SCAN);
  hipLaunchKernelGGL(k_scan_tiles, dim3(n), dim3(256), 0, st, c->tilecnt.as<u32>(), c->tileoff.as<u32>(), c->frametot.as<u32>(), g, d_slots);
  hipLaunchKernelGGL(k_bases, dim3(1), dim3(64), 0, st, c->frametot.as<u32>(), c->hdrcnt.as<u32>(), d_slots, n, c->bases.as<FrameBase>(), c->totals.as<u32>());
  stage_end(c, ST_SCAN);
  hb.resize(n);
  u32 tot[3];
  HIPCHK(hipMemcpyAsync(hb.data(), c->bases.p, n * sizeof(FrameBase), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(tot, c->totals.p, sizeof tot, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const size_t Rtot = tot[0], Ttot = tot[1], Ctot = tot[2];
  const size_t nchains = (size_t)n * NCOLCTX;
  HIPCHK(c->runs.reserve(Rtot * 4 + 64));
  HIPCHK(c->runpos.reserve(Rtot * 4 + 64));
  for (int k = 0; k < 2; k++) {
    HIPCHK(c->keys[k].reserve(Ctot * 4 + 64));
    HIPCHK(c->vals[k].reserve(Ctot * 4 + 64));
  }
  HIPCHK(c->hist.reserve((nchains + 1) * 4));
  HIPCHK(c->cstart.reserve((nchains + 1) * 4));
  HIPCHK(c->entries.reserve(Ttot * 4 + 64));
  const size_t arena_cap = Ctot / 16 + 64;
  HIPCHK(c->arena.reserve(arena_cap * sizeof(DenseTab)));
  HIPCHK(hipMemsetAsync(c->hist.p, 0, (nchains + 1) * 4, st));
  HIPCHK(hipMemsetAsync(c->arena_top.p, 0, 4, st));
  c->dbg_entries = (int64_t)Ttot;

  stage_begin(c, ST_SYMBOLS);
  hipLaunchKernelGGL(k_symbols, dim3(g.ntiles + 1, n), dim3(256), 0, st, planes, g, d_slots, c->genlist.as<int>(), c->bases.as<FrameBase>(), c->runrec.as<u32>(),
                     c->tilecnt.as<u32>(), c->tileoff.as<u32>(), c->entry.as<u8>(), c->hdrrec.as<u32>(), c->runs.as<u32>(), c->runpos.as<u32>(), c->keys[0].as<u32>(),
                     c->vals[0].as<u32>(), c->hist.as<u32>());
  stage_end(c, ST_SYMBOLS);

  stage_begin(c, ST_SORT);
  {
    size_t tmp = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp, c->hist.as<u32>(), c->cstart.as<u32>(), 0u, nchains + 1, rocprim::plus<u32>(), st));
    HIPCHK(c->scantmp.reserve(tmp));
    HIPCHK(rocprim::exclusive_scan(c->scantmp.p, tmp, c->hist.as<u32>(), c->cstart.as<u32>(), 0u, nchains + 1, rocprim::plus<u32>(), st));
    int genbits = 1;
    while ((1 << genbits) < n) genbits++;
    if (Ctot > 0) {
      size_t stmp = 0;
      HIPCHK(rocprim::radix_sort_pairs(nullptr, stmp, c->keys[0].as<u32>(), c->keys[1].as<u32>(), c->vals[0].as<u32>(), c->vals[1].as<u32>(), Ctot, 8, 22 + genbits, st));
      HIPCHK(c->sorttmp.reserve(stmp));
      HIPCHK(rocprim::radix_sort_pairs(c->sorttmp.p, stmp, c->keys[0].as<u32>(), c->keys[1].as<u32>(), c->vals[0].as<u32>(), c->vals[1].as<u32>(), Ctot, 8, 22 + genbits, st));
    }
  }
  stage_end(c, ST_SORT);

  std::vector<GenRange> rg(n);
  for (int i = 0; i < n; i++) {
    rg[i].run_begin = hb[i].run_base;
    rg[i].run_end = hb[i].run_base + hb[i].nruns;
  }
  HIPCHK(hipMemcpyAsync(c->ranges.p, rg.data(), n * sizeof(GenRange), hipMemcpyHostToDevice, st));
  stage_begin(c, ST_FIXED);
  hipLaunchKernelGGL(k_fixed_chain, dim3(NFIXED_I, n), dim3(64), 0, st, c->runs.as<u32>(), c->runpos.as<u32>(), c->ranges.as<GenRange>(), c->entries.as<u32>());
  stage_end(c, ST_FIXED);
  stage_begin(c, ST_COLOUR);
  {
    Arena ar{c->arena.as<DenseTab>(), c->arena_top.as<u32>(), (u32)arena_cap, c->err.as<u32>()};
    const u32 cap = (u32)std::min<size_t>(nchains, Ctot + 1);
    HIPCHK(c->chainlists.reserve((size_t)cap * 8 + 64));
    HIPCHK(c->chaincounts.reserve(16));
    HIPCHK(hipMemsetAsync(c->chaincounts.p, 0, 8, st));
    hipLaunchKernelGGL(k_chain_lists, dim3((unsigned)((nchains + 255) / 256)), dim3(256), 0, st, c->cstart.as<u32>(), (int)nchains, 192u, c->chainlists.as<u32>(), cap,
                       c->chaincounts.as<u32>());
    for (int which = 0; which < 2; which++) {  // long chains first
      const unsigned grid = (unsigned)std::min<u32>(cap, which == 0 ? 16384u : 32768u);
      if (grid)
        hipLaunchKernelGGL(k_colour_chain_w, dim3(grid), dim3(64), 0, st, c->keys[1].as<u32>(), c->vals[1].as<u32>(), c->cstart.as<u32>(), c->chainlists.as<u32>() + (size_t)which * cap,
                           c->chaincounts.as<u32>() + which, c->f0, ar, c->entries.as<u32>());
    }
  }
  stage_end(c, ST_COLOUR);
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
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  for (int s = 0; s < ST_COUNT + 1; s++)
    for (int k = 0; k < 2; k++) (void)hipEventCreate(&c->ev[s][k]);
  timing_reset(c);
  return c;
}

int scpr_init(scpr_codec* c, const scpr_params* p) {  // ScreenCodec::Init, screencap.cpp:1565-1584
  if (!c || !p) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  if (c->inited) scpr_deinit(c);
  c->prm = *p;
  if (c->prm.workers < 1) c->prm.workers = 1;
  c->bpp = (int)p->bits_per_pixel / 8;
  c->pitch_in = p->bits_per_pixel == 32 ? (int)p->width * 4 : (((int)p->width * c->bpp + 3) & ~3);
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
  (void)hipStreamSynchronize(c->stream);
  DevBuf* all[] = {&c->planes, &c->exitmap, &c->entry, &c->runrec, &c->tilecnt, &c->tileoff, &c->hdrrec, &c->hdrcnt, &c->frametot, &c->flags, &c->slotlist, &c->genlist,
                   &c->bases, &c->totals, &c->runs, &c->runpos, &c->keys[0], &c->keys[1], &c->vals[0], &c->vals[1], &c->hist, &c->cstart, &c->sorttmp, &c->scantmp,
                   &c->entries, &c->ranges, &c->rblocks, &c->rscratch, &c->rsize, &c->packets, &c->pktoff, &c->blkdst, &c->outsizes, &c->total64, &c->arena,
                   &c->arena_top, &c->err, &c->rcp, &c->decframes, &c->decstates, &c->hoststage_in, &c->hoststage_out, &c->chainlists, &c->chaincounts};
  for (DevBuf* b : all) b->release();
  for (int s = 0; s < ST_COUNT + 1; s++)
    for (int k = 0; k < 2; k++) (void)hipEventDestroy(c->ev[s][k]);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

void scpr_crash_happened(scpr_codec* c) {
  if (c) c->crashed = true;
}

int64_t scpr_compress_batch(scpr_codec* c, const void* d_frames, int nframes, int* ftypes, int loss, void* d_out, size_t out_capacity, uint32_t* sizes) {
  if (!c || !c->inited || !d_frames || !ftypes || !d_out || !sizes || nframes < 0) return SCPR_E_PARAM;
  if (c->crashed) return 0;  // screencap.cpp:1634
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);  // the encoder always writes v4 (screencap.cpp:1646-1648)
  if (rc != SCPR_OK) return rc;
  if (loss != c->last_loss) setup_loss(c, loss);
  timing_reset(c);
  const Geom& g = c->g;
  hipStream_t st = c->stream;
  const size_t frame_bytes = (size_t)c->pitch_in * g.H;
  int64_t written = 0;
  HIPCHK(hipMemsetAsync(c->err.p, 0, 4, st));
  for (int f0 = 0; f0 < nframes; f0 += c->slots) {
    const int n = std::min(c->slots, nframes - f0);
    const u8* src = (const u8*)d_frames + (size_t)f0 * frame_bytes;
    u32* d_nonflat = c->flags.as<u32>();
    u32* d_first = d_nonflat + n;
    HIPCHK(hipMemsetAsync(d_nonflat, 0, (size_t)n * 8, st));
    stage_begin(c, ST_PACK);
    if (c->bpp == 4) {
      dim3 gr((g.H * ((g.W + 3) >> 2) + 255) / 256, n);
      hipLaunchKernelGGL(k_pack32, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else if (c->bpp == 3) {
      dim3 gr((g.H * (g.S >> 2) + 255) / 256, n);
      hipLaunchKernelGGL(k_pack24, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first);
    } else {
      dim3 gr((g.H * g.W + 255) / 256, n);
      hipLaunchKernelGGL(k_pack16, gr, dim3(256), 0, st, src, c->planes.as<u8>(), g, d_nonflat, d_first, c->prm.red_mask, c->prm.green_mask, c->prm.blue_mask, c->rs, c->gs,
                         c->bs);
    }
    stage_end(c, ST_PACK);
    std::vector<u32> hflags((size_t)n * 2);
    HIPCHK(hipMemcpyAsync(hflags.data(), d_nonflat, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));

    // frame-type decisions: CScreenCapt::CompressFrame, screencap.cpp:1488-1511
    std::vector<ChunkFrame> cf(n);
    std::vector<int> islots;
    for (int i = 0; i < n; i++) {
      ChunkFrame& fr = cf[i];
      fr.slot = i;
      const bool flat = hflags[i] == 0;
      const u32 rgb = hflags[n + i] & 0xFFFFFFu;
      if (flat) {
        fr.kind = 1;
        fr.hdr_len = 4;
        fr.hdr = (u32)(1 + (c->version - 1) * 16) | (rgb << 8);
        // prev/model refresh when the colour differs from the last flat frame (:1490-1494):
        // it matters to the next P-frame only; key-frame streams need no action here.
        c->last_flat = true;
        c->last_flat_rgb = rgb;
        ftypes[f0 + i] = 0;
        continue;
      }
      c->last_flat = false;
      if (c->frames_done && ftypes[f0 + i]) {
        fprintf(stderr, "[scpr] P-frames are not implemented in this build\n");
        return SCPR_E_PARAM;
      }
      fr.kind = 0;
      fr.hdr_len = 1;
      fr.hdr = (u32)(2 + (c->version - 1) * 16);
      ftypes[f0 + i] = 0;
      c->frames_done++;
      islots.push_back(i);
    }
    std::vector<FrameBase> hb;
    rc = encode_intra_frames(c, islots, hb);
    if (rc != SCPR_OK) return rc;

    // rANS blocks and packets
    std::vector<RansBlock> blocks;
    std::vector<Packet> pk(n);
    {
      int k = 0;
      for (int i = 0; i < n; i++) {
        pk[i].hdr_len = cf[i].hdr_len;
        pk[i].hdr = cf[i].hdr;
        pk[i].blk_begin = (u32)blocks.size();
        pk[i].blk_count = 0;
        if (cf[i].kind == 0) {
          const FrameBase& b = hb[k++];
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
    if (nb) HIPCHK(hipMemcpyAsync(c->rblocks.p, blocks.data(), (size_t)nb * sizeof(RansBlock), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->packets.p, pk.data(), (size_t)n * sizeof(Packet), hipMemcpyHostToDevice, st));
    if (nb) {
      stage_begin(c, ST_RANS);
      hipLaunchKernelGGL(k_rans, dim3((nb + 63) / 64), dim3(64), 0, st, c->entries.as<u32>(), c->rblocks.as<RansBlock>(), nb, c->rcp.as<RansRcp>(), c->rscratch.as<u8>(),
                         c->rsize.as<u32>());
      stage_end(c, ST_RANS);
    }
    stage_begin(c, ST_GATHER);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(256), 0, st, c->packets.as<Packet>(), n, c->rsize.as<u32>(), c->outsizes.as<u32>(), c->pktoff.as<u64>(), c->blkdst.as<u64>(),
                       c->total64.as<u64>());
    hipLaunchKernelGGL(k_gather, dim3(nb + n), dim3(256), 0, st, c->packets.as<Packet>(), n, nb, c->rscratch.as<u8>(), c->rsize.as<u32>(), c->pktoff.as<u64>(),
                       c->blkdst.as<u64>(), (u8*)d_out + written, (u64)(out_capacity - (size_t)written), c->err.as<u32>());
    stage_end(c, ST_GATHER);
    u64 chunk_total = 0;
    u32 err = 0;
    HIPCHK(hipMemcpyAsync(sizes + f0, c->outsizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&chunk_total, c->total64.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&err, c->err.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    timing_collect(c);
    if (err & 2) return SCPR_E_CAPACITY;
    if (err & 1) {
      fprintf(stderr, "[scpr] dense-table arena overflow\n");
      return SCPR_E_DEVICE;
    }
    written += (int64_t)chunk_total;
  }
  return written;
}

int scpr_decompress_batch(scpr_codec* c, const void* d_packets, const uint32_t* sizes, const int* ftypes, int nframes, void* d_frames_out, int pitch) {
  if (!c || !c->inited || !d_packets || !sizes || !ftypes || !d_frames_out || nframes < 0) return SCPR_E_PARAM;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  hipStream_t st = c->stream;
  timing_reset(c);
  // first bytes of every packet decide version / flat / coded (screencap.cpp:1700, :1536)
  std::vector<u64> offs(nframes + 1, 0);
  for (int i = 0; i < nframes; i++) offs[i + 1] = offs[i] + sizes[i];
  std::vector<u32> heads(nframes, 0);
  for (int i = 0; i < nframes; i++)
    HIPCHK(hipMemcpyAsync(&heads[i], (const u8*)d_packets + offs[i], std::min<u32>(4, sizes[i]), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int done = 0;
  HIPCHK(c->err.reserve(64));
  for (int f0 = 0; f0 < nframes;) {
    if (!c->have_codec) {
      if (c->crashed && ftypes[f0] > 0) return done;
      if (ftypes[f0] > 0) return done;  // P-frame before any key frame (:1699)
      int version = (int)((heads[f0] & 0xFF) >> 4) + 1;
      int rc = ensure_codec(c, version);
      if (rc != SCPR_OK) return rc;
    }
    const Geom& g = c->g;
    const int n = std::min(c->slots, nframes - f0);
    std::vector<DecFrame> coded;
    HIPCHK(hipMemsetAsync(c->err.p, 0, 32, st));
    stage_begin(c, ST_DECODE);
    for (int i = 0; i < n; i++) {
      const int fi = f0 + i;
      if (ftypes[fi]) {
        fprintf(stderr, "[scpr] P-frames are not implemented in this build\n");
        return SCPR_E_PARAM;
      }
      c->crashed = false;
      c->frames_done++;
      const int alg = heads[fi] & 15;
      if (alg == 1) {
        const u32 rgb = (heads[fi] >> 8) & 0xFFFFFFu;
        hipLaunchKernelGGL(k_fill_flat, dim3((g.H * g.S + 255) / 256), dim3(256), 0, st, c->planes.as<u8>(), g, i, rgb);
        c->last_flat = true;
        c->last_flat_rgb = rgb;
      } else {
        c->last_flat = false;
        coded.push_back({offs[fi], sizes[fi], i, 0});
      }
    }
    if (!coded.empty()) {
      const size_t nc = coded.size();
      HIPCHK(c->decframes.reserve(nc * sizeof(DecFrame)));
      HIPCHK(c->decstates.reserve(nc * NCOLCTX * sizeof(ColState)));
      HIPCHK(hipMemsetAsync(c->decstates.p, 0, nc * NCOLCTX * sizeof(ColState), st));  // kind 0 everywhere (RenewI)
      const size_t arena_cap = nc * 4096 + 64;  // dense tables per key frame; overflow is reported
      HIPCHK(c->arena.reserve(arena_cap * sizeof(DenseTab)));
      HIPCHK(hipMemsetAsync(c->arena_top.p, 0, 4, st));
      HIPCHK(hipMemcpyAsync(c->decframes.p, coded.data(), nc * sizeof(DecFrame), hipMemcpyHostToDevice, st));
      Arena ar{c->arena.as<DenseTab>(), c->arena_top.as<u32>(), (u32)arena_cap, c->err.as<u32>()};
      int ring = 4096;
      while (ring < 2 * g.S + 1024) ring <<= 1;
      HIPCHK(hipFuncSetAttribute((const void*)k_decode_intra_w, hipFuncAttributeMaxDynamicSharedMemorySize, ring));
      hipLaunchKernelGGL(k_decode_intra_w, dim3((unsigned)nc), dim3(64), ring, st, (const u8*)d_packets, (const u8*)d_packets + offs[nframes] + 8, c->decframes.as<DecFrame>(),
                         c->planes.as<u8>(), g, c->decstates.as<ColState>(), ar, c->f0, c->err.as<u32>(), ring);
    }
    stage_end(c, ST_DECODE);
    stage_begin(c, ST_UNPACK);
    u8* out = (u8*)d_frames_out + (size_t)f0 * pitch * g.H;
    if (c->bpp == 4) {
      dim3 gr((g.H * ((g.W + 3) >> 2) + 255) / 256, n);
      hipLaunchKernelGGL(k_unpack32, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g, pitch);
    } else {
      dim3 gr((g.H * g.W + 255) / 256, n);
      hipLaunchKernelGGL(k_unpack_rows, gr, dim3(256), 0, st, c->planes.as<u8>(), out, g, pitch, c->bpp, c->rs, c->gs, c->bs);
    }
    stage_end(c, ST_UNPACK);
    u32 errv[8] = {0};
    HIPCHK(hipMemcpyAsync(errv, c->err.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    timing_collect(c);
    const u32 err = errv[0];
    if (getenv("SCPR_DEBUG")) fprintf(stderr, "[scpr] decode stats: record misses %u, colour symbols %u, dense %u, raw %u\n", errv[1], errv[2], errv[3], errv[4]);
    if (err & 4) return SCPR_E_STREAM;
    if (err & 1) return SCPR_E_DEVICE;
    done += n;
    f0 += n;
  }
  return done;
}

// ---- per-frame entry points with host pointers (the reference's own shape) ----
int scpr_compress_frame(scpr_codec* c, const void* src, void* dst, int dst_len, int* ftype, int loss) {
  if (!c || !c->inited || !src || !dst || !ftype) return SCPR_E_PARAM;
  if (c->crashed) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  int rc = ensure_codec(c, c->have_codec ? c->version : 4);
  if (rc != SCPR_OK) return rc;
  const size_t fb = (size_t)c->pitch_in * c->g.H, cap = (size_t)c->g.W * c->g.H * 6 + 64;
  HIPCHK(c->hoststage_in.reserve(fb));
  HIPCHK(c->hoststage_out.reserve(cap));
  HIPCHK(hipMemcpyAsync(c->hoststage_in.p, src, fb, hipMemcpyHostToDevice, c->stream));
  uint32_t sz = 0;
  int64_t r = scpr_compress_batch(c, c->hoststage_in.p, 1, ftype, loss, c->hoststage_out.p, cap, &sz);
  if (r <= 0) return (int)r;
  if ((int64_t)dst_len < r) return SCPR_E_CAPACITY;
  HIPCHK(hipMemcpy(dst, c->hoststage_out.p, (size_t)r, hipMemcpyDeviceToHost));
  return (int)r;
}

int scpr_decompress_frame(scpr_codec* c, const void* src, int src_len, void* dst, int pitch, int ftype) {
  if (!c || !c->inited || !src || !dst || src_len < 1) return SCPR_E_PARAM;
  if (c->crashed && ftype > 0) return 0;
  if (!c->have_codec && ftype > 0) return 0;
  if (hipSetDevice(c->device) != hipSuccess) return SCPR_E_DEVICE;
  const size_t ob = (size_t)pitch * c->prm.height;
  HIPCHK(c->hoststage_in.reserve((size_t)src_len + 64));
  HIPCHK(c->hoststage_out.reserve(ob));
  HIPCHK(hipMemsetAsync((u8*)c->hoststage_in.p + src_len, 0, 16, c->stream));
  HIPCHK(hipMemcpyAsync(c->hoststage_in.p, src, (size_t)src_len, hipMemcpyHostToDevice, c->stream));
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
  HIPCHK(dn.reserve(4));
  HIPCHK(de.reserve((size_t)n * 4));
  HIPCHK(da.reserve(acap * sizeof(DenseTab)));
  HIPCHK(dt.reserve(8));
  HIPCHK(hipMemcpy(dk.p, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dv.p, vals.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dc.p, cst, 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dl.p, lst, 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dn.p, cnt, 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dt.p, zero, 8, hipMemcpyHostToDevice));
  Arena ar{da.as<DenseTab>(), dt.as<u32>(), (u32)acap, dt.as<u32>() + 1};
  hipLaunchKernelGGL(k_colour_chain_w, dim3(1), dim3(64), 0, 0, dk.as<u32>(), dv.as<u32>(), dc.as<u32>(), dl.as<u32>(), dn.as<u32>(), f0, ar, de.as<u32>());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, de.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (DevBuf* b : {&dk, &dv, &dc, &dl, &dn, &de, &da, &dt}) b->release();
  return n;
}

int64_t scpr_debug_entries(scpr_codec* c, uint16_t* out, int64_t cap) {
  if (!c) return SCPR_E_PARAM;
  int64_t n = std::min<int64_t>(c->dbg_entries, cap);
  if (out && n > 0 && hipMemcpy(out, c->entries.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return SCPR_E_DEVICE;
  return c->dbg_entries;
}

}  // extern "C"
