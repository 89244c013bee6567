// ORACLE — TEST INFRASTRUCTURE ONLY (see spo_model.h for the parity note).
//
// CPU restatement of the ScreenPressor frame codec: ScreenCodec /
// CScreenCapt<UseANS> (screencap.h, screencap.cpp), RansMTCoder (ransmt.h), the
// byte-wise rANS primitives (rans_byte.h), and for version 2 streams UseRC
// (screencap.h:105-265) over RangeCoderSub (sub.h, sub.cpp).  Citations are relative to
// /root/reference.
#pragma once
#include <stdint.h>
#include <future>
#include <vector>
#include "spo_model.h"

namespace spo {

struct Params {  // CodecParameters, screencap.h:49-55, plus the worker count
  uint32_t width, height;
  uint32_t bits_per_pixel;  // 16, 24, 32
  uint32_t red_mask, green_mask, blue_mask;
  uint32_t high_range_x, high_range_y, low_range_x, low_range_y;
  uint32_t loss;
  uint32_t workers;  // size of the reference's CSquad pool; bitstream-visible for I-frames
  uint32_t version;  // 4 (default), 3 or 2; encoder side only (the reference itself only ever encodes 4)
};

// ---- version 2: range coder + plain adaptive count tables (UseRC, screencap.h:105-265) ----
// RangeCoderSub, sub.h:20-57 / sub.cpp:13-60.  All arithmetic is the reference's: 32-bit
// unsigned `code`/`range`, 64-bit `low`, carry propagation through Cache/FFNum.
struct RangeCoderV2 {
  uint32_t code = 0, range = 0, ffnum = 0, cache = 0;
  int64_t low = 0;
  uint8_t* out = nullptr;
  const uint8_t* in = nullptr;
  const uint8_t* in_end = nullptr;
  bool overrun = false;  // the reference throws std::length_error (sub.cpp:52-53)
  void enc_begin(uint8_t* dst);
  uint8_t* enc_end();
  void shift_low();
  void encode(uint32_t cum, uint32_t freq, uint32_t tot);
  void dec_begin(const uint8_t* src, int len);
  uint32_t get_freq(uint32_t tot) {  // sub.cpp:43-46; a damaged stream can drive the range to zero (the reference would divide by it)
    range /= tot;
    if (!range) {
      overrun = true;
      return 0;
    }
    return code / range;
  }
  void decode(uint32_t cum, uint32_t freq, uint32_t tot);
  // EncodeVal / DecodeVal / EncodeValUni / DecodeValUni, sub.cpp:63-177 (cnt[maxc] is the total)
  void enc_val(int c, uint32_t* cnt, uint32_t maxc, uint32_t step);
  int dec_val(uint32_t* cnt, uint32_t maxc, uint32_t step);
  void enc_uni(int c, uint32_t* cnt, uint32_t step);
  int dec_uni(uint32_t* cnt, uint32_t step);
};
// the tables of UseRC (screencap.h:133-260): index = position of the matching FixedModel in Models
struct V2Tables {
  std::vector<uint32_t> fixed[21];
  uint32_t maxc[21], step[21];
  std::vector<uint32_t> colour;  // [3][4096][256 + 16 + 1]
  void init(uint32_t msr_x, uint32_t msr_y);
  void reset();  // RenewI with the renew* of UseRC
  uint32_t* col(int plane, uint32_t ctx) { return &colour[((size_t)plane * 4096 + ctx) * 273]; }
};

enum { kRansL = 1u << 23, kBlockEntries = 128 * 1024 };  // rans_byte.h:47, ransmt.h:38

// rANS primitives, restated (rans_byte.h:59-146).
struct RansEnc {
  uint32_t x;
  static inline void put(uint32_t& x, uint8_t*& p, uint32_t start, uint32_t freq) {
    uint32_t x_max = ((kRansL >> kProbBits) << 8) * freq;
    while (x >= x_max) {
      *--p = (uint8_t)(x & 0xff);
      x >>= 8;
    }
    x = ((x / freq) << kProbBits) + (x % freq) + start;
  }
};

// All model tables of one codec instance (screencap.h:436-443).
struct Models {
  ColourCtx colour[3][4096];
  FixedModel run_len[6];   // ntab
  FixedModel blk_run;      // ntab2
  FixedModel blk_type;     // bttab
  FixedModel rect[4];      // sxytab
  FixedModel motion[2];    // mvtab
  FixedModel pix_type[6];  // ptypetab
  FixedModel blk_index;    // xxtab
  void reset();            // RenewI, screencap.cpp:178-198
};

struct RunRec {  // debugging tap: one pixel run as classified
  uint8_t type;
  uint8_t rgb[3];
  uint16_t n;
};

class FrameCodec {  // CScreenCapt<UseANS>
 public:
  FrameCodec(const Params& p, int version);
  int compress(uint8_t* src, uint8_t* dst, int dst_len, int& ftype);
  int decompress(const uint8_t* src, int src_len, uint8_t* dst, int ftype);
  void set_loss(int loss);
  void set_threads(int n) { threads_ = n < 1 ? 1 : n; }
  // Sharding support (not in the reference): the cross-GOP state a single stream would have where a shard starts -
  // whether a frame has been coded before (fn > 0, screencap.cpp:1504) and the flat-frame memory (last_was_flat /
  // last_flat_clr, :1490-1497, with prev holding that flat picture).
  void seed_shard(uint32_t frames_before, bool last_flat, const uint8_t rgb[3]);

  // debugging taps (stage-level known-answer tests)
  std::vector<Ivl> last_entries;     // every coder entry of the last compressed frame
  std::vector<uint16_t> last_tags;   // model id of each entry: plane*4096+ctx, or 12288+fixed id
  std::vector<uint8_t> last_records; // I: run records; P: per-block run records
  std::vector<uint8_t> blk_types;    // bts
  std::vector<int> rect_xy[4];       // sxy
  std::vector<int> mv[2];            // mvs
  const uint8_t* prev_plane() const { return prev_.data(); }
  int stride() const { return stride_; }

 private:
  int W, H, stride_, nbx, nby, version_, f0_, workers_;
  uint32_t far_x, far_y, near_x, near_y;
  uint32_t frames_ = 0;
  int loss_mask_ = -1, corr_mask_ = 0;
  bool last_flat_ = false;
  uint8_t last_flat_rgb_[4] = {0, 0, 0, 0};
  std::vector<uint8_t> prev_;
  Models* m_;
  uint32_t cx_ = 0, cx1_ = 0;

  // encoder
  std::vector<Ivl> out_;
  std::vector<uint32_t> band_start_, band_size_;  // tls[]
  std::vector<uint16_t> tags_;
  void put(Ivl e, int tag) {
    out_.push_back(e);
    tags_.push_back((uint16_t)tag);
    if (threads_ > 1 && (out_.size() & (size_t)(kBlockEntries - 1)) == 0) submit_block();  // RansMTCoder::put, ransmt.h:73-81
  }
  // The reference's two-stage shape for the CPU baseline (same bytes as the serial form): the row bands of a key
  // frame classified by a pool of threads (CSquad::RunParallel, squad.cpp:116-130) and ONE coder thread that takes
  // every full block of 131072 entries while the model thread goes on (RansMTCoder::threadProc, ransmt.h:92-105).
  int threads_ = 1;
  std::vector<std::future<std::vector<uint8_t>>> block_jobs_;
  void submit_block();
  static std::vector<uint8_t> encode_block(const Ivl* e, size_t len);
  void put_sym(FixedModel& m, int sym, int tag);  // encodeF / the EncodeVal family of version 2
  void reset_models();
  RangeCoderV2 rc_;
  V2Tables* v2_ = nullptr;
  int v2_index(const FixedModel& m) const { return (int)(&m - m_->run_len); }  // the fixed models are contiguous in Models
  void put_colour(int plane, uint8_t c);
  void put_rgb(const uint8_t* px);           // EncodeRGB
  void put_pixel(int t, int last_t, const uint8_t* px);  // WritePixel
  void apply_loss(uint8_t* src);
  bool is_flat(const uint8_t* src) const;
  void classify_intra(int worker, int y0, int ysize, const uint8_t* src);
  void decide_blocks(const uint8_t* src, int& bx1, int& bx2, int& by1, int& by2);
  bool find_motion(const uint8_t* src, int bi, int& lmx, int& lmy, int upper);
  bool same_rect(const uint8_t* src, int is, int ip, int wbytes, int h) const;
  int encode_intra(uint8_t* src, uint8_t* dst);
  int encode_inter(uint8_t* src, uint8_t* dst);
  uint8_t* flush_entries(uint8_t* dst);

  // decoder
  const uint8_t* in_ = nullptr;
  uint32_t rx_ = 0;
  int n_dec_ = 0;
  int dec_len_ = 0;
  void dec_begin(const uint8_t* p);
  void dec_count();
  int get_fixed(FixedModel& m);
  int get_colour(int plane);
  void get_rgb(int& r, int& g, int& b);
  bool get_bool();
  int decode_intra(const uint8_t* src, uint8_t* dst);
  int decode_inter(const uint8_t* src, uint8_t* dst);

 public:
  ~FrameCodec();
};

class ScreenCodec {  // screencap.h:519-541, screencap.cpp:1560-1743
 public:
  ScreenCodec() {}
  ~ScreenCodec() { deinit(); }
  void init(const Params& p);
  void deinit();
  // returns compressed size; 0 when refused; negative on parameter errors
  int compress_frame(uint8_t* src, uint8_t* dst, int dst_len, int* ftype, int loss);
  int decompress_frame(const uint8_t* src, int src_len, uint8_t* dst, int pitch, int ftype);
  void crash_happened() { crashed_ = true; }
  void set_threads(int n) {
    threads_ = n;
    if (fc_) fc_->set_threads(n);
  }
  void seed_shard(uint32_t frames_before, bool last_flat, const uint8_t rgb[3]) {
    if (!fc_) create(p_.version == 3 ? 3 : p_.version == 2 ? 2 : 4);
    fc_->seed_shard(frames_before, last_flat, rgb);
  }
  FrameCodec* inner() { return fc_; }
  FrameCodec* ensure_inner() {  // the lazily created codec object (CreateCodec, screencap.cpp:1646-1648), for the sharding calls
    if (!fc_) create(p_.version == 3 ? 3 : p_.version == 2 ? 2 : 4);
    return fc_;
  }

 private:
  void create(int version);
  Params p_{};
  FrameCodec* fc_ = nullptr;
  bool rgb32_ = false, rgb16_ = false, crashed_ = false;
  int threads_ = 1;
  uint32_t W = 0, H = 0, stride_ = 0, bpp_ = 0;
  int rs_ = 0, gs_ = 0, bs_ = 0, last_loss_ = 0;
  std::vector<uint8_t> buf_;
};

}  // namespace spo
