// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points (ctypes) over the CPU
// restatement, including stage-level taps used by the parity tests.
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "spo_codec.h"

using namespace spo;

extern "C" {

struct spo_params {
  uint32_t width, height, bits_per_pixel;
  uint32_t red_mask, green_mask, blue_mask;
  uint32_t high_range_x, high_range_y, low_range_x, low_range_y;
  uint32_t loss, workers, version;
};

void* spo_create(const spo_params* p) {
  Params q;
  q.width = p->width;
  q.height = p->height;
  q.bits_per_pixel = p->bits_per_pixel;
  q.red_mask = p->red_mask;
  q.green_mask = p->green_mask;
  q.blue_mask = p->blue_mask;
  q.high_range_x = p->high_range_x;
  q.high_range_y = p->high_range_y;
  q.low_range_x = p->low_range_x;
  q.low_range_y = p->low_range_y;
  q.loss = p->loss;
  q.workers = p->workers;
  q.version = p->version;
  ScreenCodec* c = new ScreenCodec;
  c->init(q);
  return c;
}
void spo_destroy(void* h) { delete (ScreenCodec*)h; }

int spo_compress_frame(void* h, uint8_t* src, uint8_t* dst, int dst_len, int* ftype, int loss) {
  return ((ScreenCodec*)h)->compress_frame(src, dst, dst_len, ftype, loss);
}
int spo_decompress_frame(void* h, const uint8_t* src, int src_len, uint8_t* dst, int pitch, int ftype) {
  return ((ScreenCodec*)h)->decompress_frame(src, src_len, dst, pitch, ftype);
}
void spo_crash_happened(void* h) { ((ScreenCodec*)h)->crash_happened(); }
void spo_set_threads(void* h, int n) { ((ScreenCodec*)h)->set_threads(n); }
void spo_seed_shard(void* h, uint32_t frames_before, int last_flat, uint32_t rgb) {
  const uint8_t c[3] = {(uint8_t)rgb, (uint8_t)(rgb >> 8), (uint8_t)(rgb >> 16)};
  ((ScreenCodec*)h)->seed_shard(frames_before, last_flat != 0, c);
}

// Sharding support: the vector memory mvs[] (screencap.cpp:96-97; never reset, read by FindMV :726-735) out of / into the
// codec, so that the CPU ranks of tests/test_sharding.py can hand it from shard to shard the way the GPU ranks do.
int spo_export_mv_memory(void* h, int32_t* mx, int32_t* my, int nblocks) {
  FrameCodec* f = ((ScreenCodec*)h)->ensure_inner();
  if ((int)f->mv[0].size() != nblocks) return -1;
  for (int i = 0; i < nblocks; i++) mx[i] = f->mv[0][i], my[i] = f->mv[1][i];
  return nblocks;
}
int spo_import_mv_memory(void* h, const int32_t* mx, const int32_t* my, int nblocks) {
  FrameCodec* f = ((ScreenCodec*)h)->ensure_inner();
  if ((int)f->mv[0].size() != nblocks) return -1;
  for (int i = 0; i < nblocks; i++) f->mv[0][i] = mx[i], f->mv[1][i] = my[i];
  return nblocks;
}

// ---- taps on the last compressed frame -------------------------------------
int spo_tap_entries(void* h, uint16_t* out, int cap_entries) {
  FrameCodec* f = ((ScreenCodec*)h)->inner();
  if (!f) return -1;
  int n = (int)f->last_entries.size();
  if (out) {
    int m = n < cap_entries ? n : cap_entries;
    for (int i = 0; i < m; i++) {
      out[2 * i] = f->last_entries[i].freq;
      out[2 * i + 1] = f->last_entries[i].cum;
    }
  }
  return n;
}
int spo_tap_tags(void* h, uint16_t* out, int cap) {
  FrameCodec* f = ((ScreenCodec*)h)->inner();
  if (!f) return -1;
  int n = (int)f->last_tags.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = f->last_tags[i];
  return n;
}
int spo_tap_blocks(void* h, uint8_t* types, int32_t* rect4, int32_t* mv2, int nblocks) {
  FrameCodec* f = ((ScreenCodec*)h)->inner();
  if (!f || (int)f->blk_types.size() != nblocks) return -1;
  memcpy(types, f->blk_types.data(), nblocks);
  for (int k = 0; k < 4; k++) memcpy(rect4 + (size_t)k * nblocks, f->rect_xy[k].data(), sizeof(int32_t) * nblocks);
  for (int k = 0; k < 2; k++) memcpy(mv2 + (size_t)k * nblocks, f->mv[k].data(), sizeof(int32_t) * nblocks);
  return 0;
}
int spo_tap_prev(void* h, uint8_t* out, int nbytes) {  // RGB24 plane kept as `prev`
  FrameCodec* f = ((ScreenCodec*)h)->inner();
  if (!f) return -1;
  memcpy(out, f->prev_plane(), (size_t)nbytes);
  return f->stride();
}
int spo_tap_records(void* h, uint8_t* out, int nbytes) {  // raw run-record buffer (rleData)
  FrameCodec* f = ((ScreenCodec*)h)->inner();
  if (!f) return -1;
  int n = (int)f->last_records.size();
  memcpy(out, f->last_records.data(), (size_t)(n < nbytes ? n : nbytes));
  return n;
}

// ---- stand-alone model chains (one context fed a symbol sequence) ----------
// out[2i] = freq (0 = raw byte), out[2i+1] = cum
void spo_chain_colour(const uint8_t* syms, int n, int f0, uint16_t* out) {
  ColourCtx cx;
  for (int i = 0; i < n; i++) {
    Ivl e;
    if (!cx.encode(syms[i], e, f0)) {
      e.freq = 0;
      e.cum = syms[i];
    }
    out[2 * i] = e.freq;
    out[2 * i + 1] = e.cum;
  }
}
// decoder-side model walk: given the coder values v[i] (any value inside the
// interval; for raw entries the byte itself), reproduce symbols + intervals.
// Returns the number of mismatching symbols vs `expect` (0 = consistent).
int spo_chain_colour_dec(const uint16_t* ivl, const uint8_t* expect, int n, int f0, int probe) {
  ColourCtx cx;
  int bad = 0;
  for (int i = 0; i < n; i++) {
    Ivl e;
    uint8_t c;
    int fr = ivl[2 * i], cf = ivl[2 * i + 1];
    int v = fr ? cf + (probe ? fr - 1 : 0) : 0;
    if (cx.decode(v, c, e)) {
      if (!fr || c != expect[i] || e.freq != fr || e.cum != cf) bad++;
    } else {
      if (fr) bad++;
      cx.note_raw(expect[i], true, f0);
    }
  }
  return bad;
}
void spo_chain_fixed(int nsym, const uint16_t* syms, int n, uint16_t* out) {
  FixedModel m;
  m.reset(nsym);
  for (int i = 0; i < n; i++) {
    Ivl e = m.encode(syms[i]);
    out[2 * i] = e.freq;
    out[2 * i + 1] = e.cum;
  }
}

// ---- stand-alone rANS block (ransmt.h:116-134) ------------------------------
// entries: n pairs (freq,cum); returns bytes written to out (cap >= 2n+4)
void spo_kind_hist(uint64_t* out, int reset) {  // colour symbols seen per context kind since the last reset
  for (int i = 0; i < 8; i++) {
    out[i] = kind_hist()[i];
    if (reset) kind_hist()[i] = 0;
  }
}
int spo_rans_block(const uint16_t* entries, int n, uint8_t* out) {
  std::vector<uint8_t> tmp((size_t)2 * n + 8);
  uint8_t* end = tmp.data() + tmp.size();
  uint8_t* p = end;
  uint32_t x = kRansL;
  for (int i = n - 1; i >= 0; i--) {
    uint32_t fr = entries[2 * i], cf = entries[2 * i + 1];
    if (fr)
      RansEnc::put(x, p, cf, fr);
    else
      *--p = (uint8_t)cf;
  }
  p -= 4;
  p[0] = (uint8_t)x;
  p[1] = (uint8_t)(x >> 8);
  p[2] = (uint8_t)(x >> 16);
  p[3] = (uint8_t)(x >> 24);
  int sz = (int)(end - p);
  memcpy(out, p, sz);
  return sz;
}

// ---- CPU baseline timing helper --------------------------------------------
// Encodes then decodes `nframes` frames (RGB32 or RGB24, back to back in
// `frames`), returns seconds spent in each leg and the total compressed bytes.
// key_interval: 1 = every frame is a key frame; K = key frame every K frames.
uint64_t spo_fnv1a(const uint8_t* p, uint64_t n) {
  uint64_t h = 1469598103934665603ull;
  for (uint64_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
// threads: 1 = everything on the calling thread; > 1 = the reference's shape (band pool of `threads` for key frames
// + one coder thread).  frame_sizes / frame_fnv (optional, nframes entries): size and FNV-1a of every packet.
int spo_time_stream2(const spo_params* p, uint8_t* frames, int nframes, int key_interval, int threads, double* t_enc, double* t_dec,
                     uint64_t* out_bytes, uint64_t* fnv, uint32_t* frame_sizes, uint64_t* frame_fnv);
int spo_time_stream(const spo_params* p, uint8_t* frames, int nframes, int key_interval, double* t_enc, double* t_dec,
                    uint64_t* out_bytes, uint64_t* fnv) {
  return spo_time_stream2(p, frames, nframes, key_interval, 1, t_enc, t_dec, out_bytes, fnv, nullptr, nullptr);
}
int spo_time_stream2(const spo_params* p, uint8_t* frames, int nframes, int key_interval, int threads, double* t_enc, double* t_dec,
                     uint64_t* out_bytes, uint64_t* fnv, uint32_t* frame_sizes, uint64_t* frame_fnv) {
  size_t pitch = ((size_t)p->width * (p->bits_per_pixel / 8) + 3) & ~(size_t)3;
  if (p->bits_per_pixel == 32) pitch = (size_t)p->width * 4;
  size_t fsz = (p->bits_per_pixel == 16 ? (size_t)p->width * 2 : pitch) * p->height;  // RGB16 input rows back to back (screencap.cpp:1668)
  std::vector<std::vector<uint8_t>> packets(nframes);
  std::vector<int> types(nframes);
  std::vector<uint8_t> dst((size_t)p->width * p->height * 6 + 64);
  void* enc = spo_create(p);
  spo_set_threads(enc, threads);
  auto t0 = std::chrono::steady_clock::now();
  uint64_t total = 0, hsh = 1469598103934665603ull;
  for (int i = 0; i < nframes; i++) {
    int ft = (key_interval > 0 && i % key_interval == 0) ? 0 : 1;
    int n = spo_compress_frame(enc, frames + (size_t)i * fsz, dst.data(), (int)dst.size(), &ft, (int)p->loss);
    if (n <= 0) return -1;
    packets[i].assign(dst.begin(), dst.begin() + n);
    types[i] = ft;
    total += (uint64_t)n;
  }
  auto t1 = std::chrono::steady_clock::now();
  spo_destroy(enc);
  for (int i = 0; i < nframes; i++) {
    for (uint8_t b : packets[i]) hsh = (hsh ^ b) * 1099511628211ull;
    if (frame_sizes) frame_sizes[i] = (uint32_t)packets[i].size();
    if (frame_fnv) frame_fnv[i] = spo_fnv1a(packets[i].data(), packets[i].size());
  }
  void* dec = spo_create(p);
  std::vector<uint8_t> outf(fsz);
  int bad = 0;
  auto t2 = std::chrono::steady_clock::now();
  for (int i = 0; i < nframes; i++) {
    if (spo_decompress_frame(dec, packets[i].data(), (int)packets[i].size(), outf.data(), (int)pitch, types[i]) != 1) bad++;
  }
  auto t3 = std::chrono::steady_clock::now();
  spo_destroy(dec);
  *t_enc = std::chrono::duration<double>(t1 - t0).count();
  *t_dec = std::chrono::duration<double>(t3 - t2).count();
  *out_bytes = total;
  *fnv = hsh;
  return bad;
}

}  // extern "C"
