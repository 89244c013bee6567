// ORACLE — TEST INFRASTRUCTURE ONLY (see spo_model.h for the parity note).
// CPU restatement of screencap.cpp / ransmt.h / rans_byte.h of the reference.
#include "spo_codec.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>

namespace spo {

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

// CSquadWorker::GetSegment, squad.cpp:16-31
static void segment(int total, int k, int nw, int& start, int& size) {
  if (total >= nw) {
    start = (int)((long long)total * k / nw);
    int end = (int)((long long)total * (k + 1) / nw);
    if (end > total) end = total;
    size = end - start;
  } else if (k < total) {
    start = k;
    size = 1;
  } else {
    start = size = 0;
  }
}

void Models::reset() {  // screencap.cpp:178-198
  for (int p = 0; p < 3; p++)
    for (int j = 0; j < 4096; j++) colour[p][j].reset();
  blk_run.reset(256);
  blk_index.reset(256);
  for (int i = 0; i < 6; i++) run_len[i].reset(256);
  blk_type.reset(5);
  for (int i = 0; i < 4; i++) rect[i].reset(16);
  motion[0].reset(512);
  motion[1].reset(512);
  for (int i = 0; i < 6; i++) pix_type[i].reset(6);
}

// ---------------------------------------------------------------------------
FrameCodec::FrameCodec(const Params& p, int version) {  // Init, screencap.cpp:69-124
  W = (int)p.width;
  H = (int)p.height;
  stride_ = (W * 3 + 3) & ~3;
  version_ = version;
  f0_ = version == 3 ? 64 : 32;  // screencap.cpp:1613-1614
  workers_ = p.workers < 1 ? 1 : (int)p.workers;
  far_x = version < 3 ? p.high_range_x : std::min<uint32_t>(p.high_range_x, 256);  // :76-80: version 2 takes the caller's range as it is
  far_y = version < 3 ? p.high_range_y : std::min<uint32_t>(p.high_range_y, 256);
  near_x = p.low_range_x;
  near_y = p.low_range_y;
  nbx = (W + 15) / 16;
  nby = (H + 15) / 16;
  prev_.assign((size_t)H * stride_, 0);
  blk_types.assign((size_t)nbx * nby, 0);
  for (auto& v : rect_xy) v.assign((size_t)nbx * nby, 0);
  for (auto& v : mv) v.assign((size_t)nbx * nby, 0);
  last_records.assign((size_t)W * H * 5, 0);
  band_start_.assign((size_t)imax(nby, workers_), 0);
  band_size_.assign((size_t)imax(nby, workers_), 0);
  m_ = new Models;
  if (version_ == 2) {
    v2_ = new V2Tables;
    v2_->init(far_x, far_y);
  }
  set_loss((int)p.loss);
}

FrameCodec::~FrameCodec() {
  delete m_;
  delete v2_;
}

void FrameCodec::reset_models() {  // RenewI, screencap.cpp:178-198
  if (version_ == 2) v2_->reset();
  else m_->reset();
}

// ------------------------------ version 2: coder and tables ------------------
void RangeCoderV2::enc_begin(uint8_t* dst) {  // UseRC::encodeBegin (screencap.h:111-117) + EncodeBegin (sub.h:26-28)
  out = dst;
  low = 0;
  ffnum = cache = 0;
  range = 0xFFFFFFFFu;
}
void RangeCoderV2::shift_low() {  // sub.cpp:30-41
  if ((low >> 24) != 0xFF) {
    *out++ = (uint8_t)(cache + (uint32_t)(low >> 32));
    const int c = 0xFF + (int)(low >> 32);
    while (ffnum) {
      *out++ = (uint8_t)c;
      ffnum--;
    }
    cache = (uint32_t)low >> 24;
  } else
    ffnum++;
  low = (int64_t)(uint32_t)((uint32_t)low << 8);
}
uint8_t* RangeCoderV2::enc_end() {  // sub.cpp:13-19
  low += 1;
  for (int i = 0; i < 5; i++) shift_low();
  return out;
}
void RangeCoderV2::encode(uint32_t cum, uint32_t freq, uint32_t tot) {  // sub.cpp:21-27
  range /= tot;
  low += (uint32_t)(cum * range);
  range *= freq;
  while (range < (1u << 24)) {
    shift_low();
    range <<= 8;
  }
}
void RangeCoderV2::dec_begin(const uint8_t* src, int len) {  // sub.h:30-42
  code = 0;
  range = 0xFFFFFFFFu;
  in = src;
  in_end = src + len;
  overrun = len < 5;
  for (int i = 0; i < 5 && in < in_end; i++) code = (code << 8) | *in++;
}
void RangeCoderV2::decode(uint32_t cum, uint32_t freq, uint32_t tot) {  // sub.cpp:48-60 (range was divided by tot in get_freq)
  (void)tot;
  code -= cum * range;
  range *= freq;
  while (range < (1u << 24)) {
    if (in >= in_end) {
      overrun = true;
      return;
    }
    code = (code << 8) | *in++;
    range <<= 8;
  }
}
static inline void v2_bump(uint32_t* cnt, uint32_t maxc, int c, uint32_t step) {  // the tail of EncodeVal / DecodeVal, sub.cpp:73-82
  cnt[c] += step;
  cnt[maxc] += step;
  if (cnt[maxc] > (1u << 16)) {  // BOT_C
    uint32_t t = 0;
    for (uint32_t i = 0; i < maxc; i++) {
      cnt[i] = (cnt[i] >> 1) + 1;
      t += cnt[i];
    }
    cnt[maxc] = t;
  }
}
void RangeCoderV2::enc_val(int c, uint32_t* cnt, uint32_t maxc, uint32_t step) {  // sub.cpp:63-83
  uint32_t cum = 0;
  for (int i = 0; i < c; i++) cum += cnt[i];
  encode(cum, cnt[c], cnt[maxc]);
  v2_bump(cnt, maxc, c, step);
}
int RangeCoderV2::dec_val(uint32_t* cnt, uint32_t maxc, uint32_t step) {  // sub.cpp:86-110
  const uint32_t value = get_freq(cnt[maxc]);
  uint32_t cum = 0, c = 0;
  for (; c < maxc; c++) {
    if (value >= cum + cnt[c]) cum += cnt[c];
    else break;
  }
  if (c >= maxc) {  // only a damaged stream: the reference reads cnt[maxc] here; stop instead
    overrun = true;
    c = maxc - 1;
    cum -= cnt[c];
  }
  decode(cum, cnt[c], cnt[maxc]);
  v2_bump(cnt, maxc, (int)c, step);
  return (int)c;
}
static inline void v2_bump_uni(uint32_t* cnt, int c, uint32_t x, uint32_t step) {  // sub.cpp:126-139
  cnt[c] += step;
  cnt[256 + x] += step;
  cnt[272] += step;
  if (cnt[272] > (1u << 16)) {
    uint32_t t = 0;
    for (int i = 0; i < 256; i++) {
      cnt[i] = (cnt[i] >> 1) + 1;
      t += cnt[i];
    }
    cnt[272] = t;
    for (int i = 0; i < 16; i++) {
      cnt[256 + i] = 0;
      for (int j = 0; j < 16; j++) cnt[256 + i] += cnt[i * 16 + j];
    }
  }
}
void RangeCoderV2::enc_uni(int c, uint32_t* cnt, uint32_t step) {  // sub.cpp:113-141
  uint32_t cum = 0, x = 0;
  for (; x < (uint32_t)c / 16; x++) cum += cnt[256 + x];
  for (uint32_t i = x * 16; i < (uint32_t)c; i++) cum += cnt[i];
  encode(cum, cnt[c], cnt[272]);
  v2_bump_uni(cnt, c, x, step);
}
int RangeCoderV2::dec_uni(uint32_t* cnt, uint32_t step) {  // sub.cpp:144-177
  const uint32_t value = get_freq(cnt[272]);
  uint32_t cum = 0, x = 0;
  for (; x < 16; x++) {
    if (value >= cum + cnt[256 + x]) cum += cnt[256 + x];
    else break;
  }
  uint32_t c = x * 16;
  for (; c < 256; c++) {
    if (value >= cum + cnt[c]) cum += cnt[c];
    else break;
  }
  if (c >= 256) {  // damaged stream
    overrun = true;
    c = 255;
    x = 15;
    cum -= cnt[c];
  }
  decode(cum, cnt[c], cnt[272]);
  v2_bump_uni(cnt, (int)c, x, step);
  return (int)c;
}
void V2Tables::init(uint32_t msr_x, uint32_t msr_y) {
  // order of the FixedModel members of Models: run_len[6], blk_run, blk_type, rect[4], motion[2], pix_type[6], blk_index
  for (int i = 0; i < 6; i++) maxc[i] = 256, step[i] = 400;       // ntab, SC_NSTEP (screencap.h:34, :136-149)
  maxc[6] = 256, step[6] = 20;                                    // ntab2, SC_BTNSTEP (:41, :202-211)
  maxc[7] = 5, step[7] = 10;                                      // bttab, SC_BTSTEP (:39, :213-222)
  for (int i = 8; i < 12; i++) maxc[i] = 16, step[i] = 100;       // sxytab, SC_SXYSTEP (:42, :224-233)
  maxc[12] = msr_x * 2, maxc[13] = msr_y * 2;                     // mvtab sized by the motion range (:235-260)
  step[12] = step[13] = 100;                                      // SC_MSTEP (:43)
  for (int i = 14; i < 20; i++) maxc[i] = 6, step[i] = 1000;      // ptypetab, SC_UNSTEP (:44, :160-169)
  maxc[20] = 256, step[20] = 1;                                   // xxtab, SC_XXSTEP (:45, :191-200)
  for (int i = 0; i < 21; i++) fixed[i].assign(maxc[i] + 1, 0);
  colour.assign((size_t)3 * 4096 * 273, 0);
  reset();
}
void V2Tables::reset() {
  for (int i = 0; i < 21; i++) {  // renewN / FixedTab::renew / renewM: every count 1, total = alphabet size
    for (uint32_t j = 0; j < maxc[i]; j++) fixed[i][j] = 1;
    fixed[i][maxc[i]] = maxc[i];
  }
  for (size_t t = 0; t < (size_t)3 * 4096; t++) {  // renewC, screencap.h:182-189
    uint32_t* c = &colour[t * 273];
    for (int n = 0; n < 256; n++) c[n] = 1;
    for (int n = 0; n < 16; n++) c[256 + n] = 16;
    c[272] = 256;
  }
}

void FrameCodec::put_sym(FixedModel& m, int sym, int tag) {
  if (version_ == 2) {
    const int k = v2_index(m);
    rc_.enc_val(sym, v2_->fixed[k].data(), v2_->maxc[k], v2_->step[k]);
  } else {
    put(m.encode(sym), tag);
  }
}

void FrameCodec::set_loss(int loss) {  // SetupLossMask, screencap.cpp:127-139
  int mask = 0;
  for (int i = 0; i < loss; i++) mask = (mask << 1) | 1;
  mask = (mask << 8) + mask;
  mask = (int)(((uint32_t)mask << 16) + (uint32_t)mask);
  loss_mask_ = ~mask;
  int c = (1 << loss) >> 1;
  c = (c << 8) + c;
  corr_mask_ = (int)(((uint32_t)c << 16) + (uint32_t)c);
}

void FrameCodec::apply_loss(uint8_t* src) {  // DoLoss + CMD_DOLOSS, :201-220, :852-861
  if (loss_mask_ != -1) {
    size_t n = (size_t)H * stride_ / 4;
    uint32_t lm = (uint32_t)loss_mask_, cm = (uint32_t)corr_mask_;
    for (size_t i = 0; i < n; i++) {
      uint32_t v;
      memcpy(&v, src + 4 * i, 4);
      v = (v & lm) | cm;
      memcpy(src + 4 * i, &v, 4);
    }
  }
  if (W & 3) {
    int pad = stride_ - W * 3;
    for (int y = 0; y < H; y++) memset(src + (size_t)y * stride_ + W * 3, 0, pad);
  }
}

bool FrameCodec::is_flat(const uint8_t* src) const {  // IsFlat, :1436-1444
  if (W & 3)
    return !memcmp(src, src + 3, (size_t)(W - 1) * 3) && !memcmp(src, src + stride_, (size_t)(H - 1) * stride_);
  return !memcmp(src, src + 3, (size_t)W * H * 3 - 3);
}

// ------------------------------ encoder: symbols ---------------------------
void FrameCodec::put_colour(int plane, uint8_t c) {  // UseANS::encodeC, screencap.h:311-317; UseRC::encodeC, :172-175
  if (version_ == 2) {
    rc_.enc_uni(c, v2_->col(plane, cx_ + cx1_), 400);  // SC_STEP
    return;
  }
  Ivl e;
  if (!m_->colour[plane][cx_ + cx1_].encode(c, e, f0_)) {
    e.freq = 0;
    e.cum = c;
  }
  put(e, plane * 4096 + (int)(cx_ + cx1_));
}

#define SPO_NEXT_CX(v)            \
  cx1_ = (cx_ << 6) & 0xFC0;      \
  cx_ = (uint32_t)(v) >> 2;

void FrameCodec::put_rgb(const uint8_t* px) {  // EncodeRGB, :631-643
  put_colour(0, px[0]);
  SPO_NEXT_CX(px[0]);
  put_colour(1, px[1]);
  SPO_NEXT_CX(px[1]);
  put_colour(2, px[2]);
  SPO_NEXT_CX(px[2]);
}

void FrameCodec::put_pixel(int t, int last_t, const uint8_t* px) {  // WritePixel, :609-627
  put_sym(m_->pix_type[last_t], t, 12294 + last_t);
  if (t) return;
  put_colour(0, px[0]);
  SPO_NEXT_CX(px[0]);
  put_colour(1, px[1]);
  SPO_NEXT_CX(px[1]);
  put_colour(2, px[2]);
}

// RansMTCoder::writeBlock per 131072 entries, ransmt.h:116-134
std::vector<uint8_t> FrameCodec::encode_block(const Ivl* e, size_t len) {
  std::vector<uint8_t> tmp((size_t)kBlockEntries * 2 + 8);
  uint32_t x = kRansL;
  uint8_t* end = tmp.data() + tmp.size();
  uint8_t* p = end;
  for (size_t i = len; i-- > 0;) {
    if (e[i].freq)
      RansEnc::put(x, p, e[i].cum, e[i].freq);
    else
      *--p = (uint8_t)e[i].cum;
  }
  p -= 4;  // RansEncFlush, rans_byte.h:90-102
  p[0] = (uint8_t)x;
  p[1] = (uint8_t)(x >> 8);
  p[2] = (uint8_t)(x >> 16);
  p[3] = (uint8_t)(x >> 24);
  return std::vector<uint8_t>(p, end);
}

void FrameCodec::submit_block() {  // the block that has just filled up goes to the coder thread (one at a time, like the reference's single worker)
  if (!block_jobs_.empty()) block_jobs_.back().wait();
  std::vector<Ivl> blk(out_.end() - kBlockEntries, out_.end());
  block_jobs_.push_back(std::async(std::launch::async, [b = std::move(blk)]() { return encode_block(b.data(), b.size()); }));
}

uint8_t* FrameCodec::flush_entries(uint8_t* dst) {
  const size_t n = out_.size();
  size_t bi = 0;
  for (size_t b0 = 0; b0 < n; b0 += kBlockEntries, bi++) {
    const size_t len = std::min<size_t>(kBlockEntries, n - b0);
    // full blocks handed over on the way come back from the coder thread; the tail is coded here (RansMTCoder::finish, ransmt.h:83-90)
    const std::vector<uint8_t> chunk = bi < block_jobs_.size() ? block_jobs_[bi].get() : encode_block(&out_[b0], len);
    memcpy(dst, chunk.data(), chunk.size());
    dst += chunk.size();
  }
  block_jobs_.clear();
  last_entries = out_;
  last_tags = tags_;
  return dst;
}

// ------------------------------ intra classification -----------------------
// predictor tests on packed RGB24; `last` = previous pixel in raster order,
// off = -stride-3 (top-left).  screencap.cpp:502-521, :560-574
static inline bool eq3(const uint8_t* a, const uint8_t* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }
static inline bool grad3(const uint8_t* p, const uint8_t* left, int off) {
  return p[0] == (int)left[0] + (int)p[off + 3] - (int)p[off] && p[1] == (int)left[1] + (int)p[off + 4] - (int)p[off + 1] &&
         p[2] == (int)left[2] + (int)p[off + 5] - (int)p[off + 2];
}
static inline int intra_type(const uint8_t* p, const uint8_t* last, int off) {
  if (eq3(p, last)) return 1;
  if (eq3(p, p + off)) return 5;
  if (eq3(p, p + off + 3)) return 2;
  if (grad3(p, last, off)) return 4;
  return 0;
}
static inline bool intra_fits(int t, const uint8_t* p, const uint8_t* last, int off) {
  switch (t) {
    case 0:
    case 1: return eq3(p, last);
    case 2: return eq3(p, p + off + 3);
    case 4: return grad3(p, last, off);
    case 5: return eq3(p, p + off);
  }
  return false;
}

void FrameCodec::classify_intra(int worker, int y0, int ysize, const uint8_t* src) {  // :876-919
  uint8_t* rec = last_records.data();
  size_t j = (size_t)y0 * W * 5;
  band_start_[worker] = (uint32_t)j;
  int x = 0, y = y0, lasti = (y0 - 1) * stride_ + (W - 1) * 3;
  if (y0 == 0) {
    x = 1;
    y = 1;
    lasti = stride_;
  }
  const int yend = y0 + ysize, off = -stride_ - 3;
  const int i0 = y * stride_ + x * 3;
  int t = intra_type(src + i0, src + lasti, off);
  rec[j++] = (uint8_t)t;
  if (!t) {
    rec[j++] = src[i0];
    rec[j++] = src[i0 + 1];
    rec[j++] = src[i0 + 2];
  }
  int n = 1;
  x++;
  lasti = i0;
  if (x >= W) {  // (reference assumes W >= 3; keep the walk well-defined for W == 2)
    x = 0;
    y++;
  }
  while (y < yend) {
    const int i = y * stride_ + x * 3;
    if (n < 255 && intra_fits(t, src + i, src + lasti, off))
      n++;
    else {
      rec[j++] = (uint8_t)n;
      t = intra_type(src + i, src + lasti, off);
      rec[j++] = (uint8_t)t;
      if (!t) {
        rec[j++] = src[i];
        rec[j++] = src[i + 1];
        rec[j++] = src[i + 2];
      }
      n = 1;
    }
    lasti = i;
    if (++x >= W) {
      x = 0;
      y++;
    }
  }
  rec[j++] = (uint8_t)n;
  band_size_[worker] = (uint32_t)(j - band_start_[worker]);
}

int FrameCodec::encode_intra(uint8_t* src, uint8_t* dst) {  // CompressI, :319-403
  apply_loss(src);
  cx_ = cx1_ = 0;
  if (threads_ > 1 && workers_ > 1) {  // CMD_CLASSIFYPIXELSI on the pool (:334, squad.cpp:116-130): bands are independent
    const int nt = std::min(threads_, workers_);
    std::vector<std::thread> pool;
    for (int q = 0; q < nt; q++)
      pool.emplace_back([&, q]() {
        for (int k = q; k < workers_; k += nt) {
          int y0 = 0, ys = 1;
          segment(H, k, workers_, y0, ys);
          classify_intra(k, y0, ys, src);
        }
      });
    for (std::thread& t : pool) t.join();
  } else {
    for (int k = 0; k < workers_; k++) {
      int y0 = 0, ys = 1;
      segment(H, k, workers_, y0, ys);
      classify_intra(k, y0, ys, src);
    }
  }
  out_.clear();
  tags_.clear();
  if (version_ == 2) rc_.enc_begin(dst);
  reset_models();
  put_rgb(src);

  int t = 0, last_t = 0, n = 1, lasti = 0;
  for (int k = 1; k < W + 1; k++) {  // rest of row 0 and pixel (0,1)
    int i = (k / W) * stride_ + (k % W) * 3;
    if (eq3(src + i, src + lasti) && n < 255)
      n++;
    else {
      put_sym(m_->run_len[0], n, 12288);
      put_rgb(src + i);
      n = 1;
    }
    lasti = i;
  }
  put_sym(m_->run_len[0], n, 12288);
  int x = 0, y = 1;
  const uint8_t* rec = last_records.data();
  for (int band = 0; band < workers_; band++) {
    size_t j = band_start_[band], jend = j + band_size_[band];
    while (j < jend) {
      t = rec[j];
      cx1_ = ((uint32_t)(src[lasti + 1] >> 2) << 6) & 0xFC0;
      cx_ = src[lasti + 2] >> 2;
      put_pixel(t, last_t, rec + j + 1);
      last_t = t;
      if (!t) j += 3;
      n = rec[j + 1];
      put_sym(m_->run_len[t], n, 12288 + t);
      j += 2;
      x += n;
      while (x >= W) {
        x -= W;
        y++;
      }
      lasti = y * stride_ + x * 3;
    }
  }
  uint8_t* end = version_ == 2 ? rc_.enc_end() : flush_entries(dst);
  if (version_ == 2) last_entries.clear();
  memcpy(prev_.data(), src, (size_t)H * stride_);
  return (int)(end - dst);
}

// ------------------------------ inter: block analysis ----------------------
bool FrameCodec::same_rect(const uint8_t* src, int is, int ip, int wbytes, int h) const {  // :817-825
  const uint8_t* pv = prev_.data();
  for (int y = 0; y < h; y++) {
    if (memcmp(src + is, pv + ip, wbytes)) return false;
    is += stride_;
    ip += stride_;
  }
  return true;
}

// FindMV, :684-814.  First exact match in a fixed candidate order.
bool FrameCodec::find_motion(const uint8_t* src, int bi, int& lmx, int& lmy, int upper) {
  const int X = W, Y = H;
  int x1 = rect_xy[0][bi], y1 = rect_xy[1][bi], x2 = rect_xy[2][bi], y2 = rect_xy[3][bi];
  int rx1 = x1 - (int)near_x, rx2 = x1 + (int)near_x, ry1 = y1 - (int)near_y, ry2 = y1 + (int)near_y;
  if (rx1 < 0) rx1 = 0;
  if (ry1 < 0) ry1 = 0;
  if (rx2 + x2 - x1 > X) rx2 = X - x2 + x1 + 1;
  if (ry2 + y2 - y1 > Y) ry2 = Y - y2 + y1 + 1;
  int fx1 = x1 - (int)far_x, fx2 = x1 + (int)far_x, fy1 = y1 - (int)far_y, fy2 = y1 + (int)far_y;
  if (fx1 < 0) fx1 = 0;
  if (fy1 < 0) fy1 = 0;
  if (fx2 + x2 - x1 > X) fx2 = X - x2 + x1 + 1;
  if (fy2 + y2 - y1 > Y) fy2 = Y - y2 + y1 + 1;

  const int is = y1 * stride_ + x1 * 3, wb = (x2 - x1) * 3, h = y2 - y1;
  auto at = [&](int x, int y) { return same_rect(src, is, y * stride_ + x * 3, wb, h); };
  auto hit = [&](int x, int y, bool remember) {
    mv[0][bi] = x - x1;
    mv[1][bi] = y - y1;
    if (remember) {
      lmx = x - x1;
      lmy = y - y1;
    }
    return true;
  };
  {  // 1. the last vector found by search (not refreshed by predicted hits)
    int sx = x1 + lmx, sy = y1 + lmy;
    if (sx >= fx1 && sx < fx2 && sy >= fy1 && sy < fy2 && at(sx, sy)) return hit(sx, sy, false);
  }
  if (upper >= 0 && (mv[0][upper] != lmx || mv[1][upper] != lmy)) {  // 2. block above
    int x = x1 + mv[0][upper], y = y1 + mv[1][upper];
    if (x >= fx1 && x < fx2 && y >= fy1 && y < fy2 && at(x, y)) return hit(x, y, false);
  }
  int common = imin(y1 - fy1, fy2 - y1 - 1);
  int yup = y1 - 1, ydn = y1 + 1;
  for (int k = 0; k < common; k++, yup--, ydn++) {  // 3. vertical, alternating
    if (at(x1, yup)) return hit(x1, yup, true);
    if (at(x1, ydn)) return hit(x1, ydn, true);
  }
  for (; yup >= fy1; yup--)
    if (at(x1, yup)) return hit(x1, yup, true);
  for (; ydn < fy2; ydn++)
    if (at(x1, ydn)) return hit(x1, ydn, true);
  for (int x = x1; x >= fx1; x--)  // 4. horizontal
    if (at(x, y1)) return hit(x, y1, true);
  for (int x = x1; x < fx2; x++)
    if (at(x, y1)) return hit(x, y1, true);
  for (int x = x1; x >= rx1; x--) {  // 5. near 2-D window
    for (int y = y1; y >= ry1; y--)
      if (at(x, y)) return hit(x, y, true);
    for (int y = y1 + 1; y < ry2; y++)
      if (at(x, y)) return hit(x, y, true);
  }
  for (int x = x1 + 1; x < rx2; x++) {
    for (int y = y1; y >= ry1; y--)
      if (at(x, y)) return hit(x, y, true);
    for (int y = y1 + 1; y < ry2; y++)
      if (at(x, y)) return hit(x, y, true);
  }
  return false;
}

// P-frame predictors, :525-556, :578-604
static inline int inter_type(const uint8_t* p, const uint8_t* pr, int off) {
  if (eq3(p, p - 3)) return 1;
  if (eq3(p, pr)) return 3;
  if (eq3(p, p + off)) return 5;
  if (eq3(p, p + off + 3)) return 2;
  if (grad3(p, p - 3, off)) return 4;
  return 0;
}
static inline bool inter_fits(int t, const uint8_t* p, const uint8_t* pr, const uint8_t* last, int off) {
  switch (t) {
    case 0: return eq3(p, last);
    case 1: return eq3(p, p - 3);
    case 2: return eq3(p, p + off + 3);
    case 3: return eq3(p, pr);
    case 4: return grad3(p, p - 3, off);
    case 5: return eq3(p, p + off);
  }
  return false;
}
static inline int edge_type(const uint8_t* p, const uint8_t* pr) { return eq3(p, pr) ? 3 : 0; }
static inline bool edge_fits(int t, const uint8_t* p, const uint8_t* pr, const uint8_t* last) {
  if (t == 0) return eq3(p, last);
  if (t == 3) return eq3(p, pr);
  return false;
}

// DecideBlockTypes in canonical order (one worker, block rows top to bottom),
// :928-1087.  With one worker every row above is Done, so the upper block's
// vector is always a candidate for by > 0.
void FrameCodec::decide_blocks(const uint8_t* src, int& obx1, int& obx2, int& oby1, int& oby2) {
  int bx1 = nbx, bx2 = -1, by1 = nby, by2 = -1;
  int lmx = 0, lmy = 0;
  const int off = -stride_ - 3;
  const uint8_t* pv = prev_.data();
  uint8_t* rec = last_records.data();
  for (int by = 0; by < nby; by++) {
    size_t j = (size_t)by * 16 * W * 5;
    band_start_[by] = (uint32_t)j;
    for (int bx = 0; bx < nbx; bx++) {
      const int x1 = bx * 16, x2 = imin(bx * 16 + 16, W), y1 = by * 16, y2 = imin(by * 16 + 16, H);
      const int bi = by * nbx + bx, upper = by > 0 ? bi - nbx : -1;
      const int bw = (x2 - x1) * 3, xb = x1 * 3;
      int cp = 0;
      bool changed = false;
      auto differs = [&](int x, int y) { return !eq3(src + y * stride_ + x * 3, pv + y * stride_ + x * 3); };
      for (int y = y1; y < y2; y++) {
        if (!memcmp(src + y * stride_ + xb, pv + y * stride_ + xb, bw)) continue;
        changed = true;
        int sx1 = x2, sx2 = x1, sy1 = y, sy2 = y;
        for (int yy = y2 - 1; yy > sy1; yy--)
          if (memcmp(src + yy * stride_ + xb, pv + yy * stride_ + xb, bw)) {
            sy2 = yy;
            break;
          }
        for (int x = x1; x < x2; x++)
          if (differs(x, sy2)) {
            sx1 = x;
            break;
          }
        sx2 = sx1;
        for (int x = x2 - 1; x > sx1; x--)
          if (differs(x, sy2)) {
            sx2 = x;
            break;
          }
        for (int yy = sy1; yy < sy2; yy++) {
          for (int x = x1; x < sx1; x++)
            if (differs(x, yy)) {
              sx1 = x;
              break;
            }
          for (int x = x2 - 1; x > sx2; x--)
            if (differs(x, yy)) {
              sx2 = x;
              break;
            }
        }
        sx2++;
        sy2++;
        if (sx1 > x1 || sy1 > y1 || sx2 < x2 || sy2 < y2) {
          cp = 2;
          rect_xy[0][bi] = sx1;
          rect_xy[1][bi] = sy1;
          rect_xy[2][bi] = sx2;
          rect_xy[3][bi] = sy2;
        } else {
          cp = 1;
          rect_xy[0][bi] = x1;
          rect_xy[1][bi] = y1;
          rect_xy[2][bi] = x2;
          rect_xy[3][bi] = y2;
        }
        if (find_motion(src, bi, lmx, lmy, upper))
          cp += 2;
        else {
          int n = 333, lasti = 0, t = 0;
          for (int yy = sy1; yy < sy2; yy++) {
            int i = yy * stride_ + sx1 * 3;
            for (int x = sx1; x < sx2; x++) {
              const bool inner = x > 0 && yy > 0;
              bool fits = n < 255 && (inner ? inter_fits(t, src + i, pv + i, src + lasti, off) : edge_fits(t, src + i, pv + i, src + lasti));
              if (fits)
                n++;
              else {
                if (n != 333) rec[j++] = (uint8_t)n;
                t = inner ? inter_type(src + i, pv + i, off) : edge_type(src + i, pv + i);
                rec[j++] = (uint8_t)t;
                n = 1;
              }
              lasti = i;
              i += 3;
            }
          }
          rec[j++] = (uint8_t)n;
        }
        break;
      }
      blk_types[bi] = (uint8_t)cp;
      if (changed) {
        bx1 = imin(bx, bx1);
        by1 = imin(by, by1);
        bx2 = imax(bx, bx2);
        by2 = imax(by, by2);
      }
    }
    band_size_[by] = (uint32_t)(j - band_start_[by]);
  }
  obx1 = bx1 == nbx ? -1 : bx1;
  oby1 = by1 == nby ? -1 : by1;
  obx2 = bx2;
  oby2 = by2;
}

int FrameCodec::encode_inter(uint8_t* src, uint8_t* dst0) {  // CompressP, :1091-1271
  uint8_t* dst = dst0;
  apply_loss(src);
  if (!memcmp(src, prev_.data(), (size_t)H * stride_)) {
    *dst = 0;
    last_entries.clear();
    return 1;
  }
  *dst++ = 1;
  out_.clear();
  tags_.clear();
  if (version_ == 2) rc_.enc_begin(dst);
  int bx1, bx2, by1, by2;
  decide_blocks(src, bx1, bx2, by1, by2);

  int xx1 = by1 * nbx + bx1, xx2 = by2 * nbx + bx2;
  put_sym(m_->blk_index, xx1 & 255, 12300);
  put_sym(m_->blk_index, (xx1 >> 8) & 255, 12300);
  put_sym(m_->blk_index, xx2 & 255, 12300);
  put_sym(m_->blk_index, (xx2 >> 8) & 255, 12300);

  int oldt = -1, n = -1;
  for (int b = xx1; b <= xx2; b++) {  // block-type RLE, :1155-1169
    if (blk_types[b] == oldt && n < 255)
      n++;
    else {
      if (n > 0) put_sym(m_->blk_run, n, 12301);
      put_sym(m_->blk_type, blk_types[b], 12302);
      oldt = blk_types[b];
      n = 1;
    }
  }
  put_sym(m_->blk_run, n, 12301);

  cx_ = cx1_ = 0;
  int lastmx = 0, lastmy = 0;
  const uint8_t* rec = last_records.data();
  for (int by = 0; by < nby; by++) {
    size_t j = band_start_[by];
    for (int bx = 0; bx < nbx; bx++) {
      int bi = by * nbx + bx, bt = blk_types[bi];
      if (!bt) continue;
      int x1 = rect_xy[0][bi], x2 = rect_xy[2][bi], y1 = rect_xy[1][bi], y2 = rect_xy[3][bi];
      if ((bt - 1) & 1) {
        put_sym(m_->rect[0], x1 - bx * 16, 12303);
        put_sym(m_->rect[1], y1 - by * 16, 12304);
        put_sym(m_->rect[2], x2 - 1 - bx * 16, 12305);
        put_sym(m_->rect[3], y2 - 1 - by * 16, 12306);
      }
      if ((bt - 1) & 2) {  // motion vector, :1199-1214
        if (version_ == 2) {  // no "same vector" flag before version 3 (canEncodeBool, screencap.h:262)
          put_sym(m_->motion[0], mv[0][bi] + (int)far_x, 12307);
          put_sym(m_->motion[1], mv[1][bi] + (int)far_y, 12308);
        } else if (bi > 0 && mv[0][bi] == lastmx && mv[1][bi] == lastmy) {
          Ivl e = {kProbScale / 2, kProbScale / 2};
          put(e, 12309);
        } else {
          Ivl e = {kProbScale / 2, 0};
          put(e, 12309);
          put_sym(m_->motion[0], mv[0][bi] + (int)far_x, 12307);
          put_sym(m_->motion[1], mv[1][bi] + (int)far_y, 12308);
          lastmx = mv[0][bi];
          lastmy = mv[1][bi];
        }
      } else {  // pixel runs over the rect, :1215-1245
        int y = y1, x = x1, last_t = 0;
        while (y < y2) {
          int t = rec[j++];
          int rn = rec[j++];
          int i = y * stride_ + x * 3;
          put_pixel(t, last_t, src + i);
          last_t = t;
          put_sym(m_->run_len[t], rn, 12288 + t);
          if (rn > 1) {
            int q = x - x1 + rn - 1;
            x = q % (x2 - x1) + x1;
            y += q / (x2 - x1);
            i = y * stride_ + x * 3;
          }
          cx1_ = ((uint32_t)(src[i + 1] >> 2) << 6) & 0xFC0;
          cx_ = src[i + 2] >> 2;
          if (++x == x2) {
            x = x1;
            y++;
          }
        }
      }
    }
  }
  uint8_t* end = version_ == 2 ? rc_.enc_end() : flush_entries(dst);
  if (version_ == 2) last_entries.clear();
  memcpy(prev_.data(), src, (size_t)H * stride_);
  return (int)(end - dst0);
}

void FrameCodec::seed_shard(uint32_t frames_before, bool last_flat, const uint8_t rgb[3]) {
  frames_ = frames_before;
  last_flat_ = last_flat;
  if (last_flat) {  // the state the flat frame left behind (:1490-1494): prev := that picture, models renewed
    memcpy(last_flat_rgb_, rgb, 3);
    for (int y = 0; y < H; y++) {
      uint8_t* row = prev_.data() + (size_t)y * stride_;
      for (int x = 0; x < W; x++) memcpy(row + 3 * x, rgb, 3);
      memset(row + 3 * W, 0, (size_t)stride_ - 3 * W);
    }
    reset_models();
  }
}

int FrameCodec::compress(uint8_t* src, uint8_t* dst, int /*dst_len*/, int& ftype) {  // :1456-1518
  if (is_flat(src)) {
    ftype = 0;
    if (!(last_flat_ && !memcmp(src, last_flat_rgb_, 3))) {
      memcpy(prev_.data(), src, (size_t)H * stride_);
      reset_models();
      memcpy(last_flat_rgb_, src, 3);
    }
    dst[0] = (uint8_t)(1 + (version_ - 1) * 16);
    memcpy(dst + 1, src, 3);
    last_flat_ = true;
    last_entries.clear();
    return 4;
  }
  last_flat_ = false;
  if (frames_ && ftype) {
    ftype = 1;
    frames_++;
    return encode_inter(src, dst);
  }
  ftype = 0;
  frames_++;
  dst[0] = (uint8_t)(2 + (version_ - 1) * 16);
  return encode_intra(src, dst + 1) + 1;
}

// ------------------------------ decoder -----------------------------------
void FrameCodec::dec_begin(const uint8_t* p) {  // decodeBegin, screencap.h:295-301; UseRC::decodeBegin, :123-125
  if (version_ == 2) {
    rc_.dec_begin(p, dec_len_);
    return;
  }
  in_ = p;
  n_dec_ = 0;
  rx_ = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8) | ((uint32_t)in_[2] << 16) | ((uint32_t)in_[3] << 24);
  in_ += 4;
}
void FrameCodec::dec_count() {  // screencap.h:327-331
  if (++n_dec_ == kBlockEntries) {
    rx_ = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8) | ((uint32_t)in_[2] << 16) | ((uint32_t)in_[3] << 24);
    in_ += 4;
    n_dec_ = 0;
  }
}
static inline void rans_advance(uint32_t& x, const uint8_t*& p, uint32_t start, uint32_t freq) {  // rans_byte.h:130-146
  x = freq * (x >> kProbBits) + (x & (kProbScale - 1)) - start;
  while (x < kRansL) x = (x << 8) | *p++;
}
int FrameCodec::get_fixed(FixedModel& m) {  // decodeF, screencap.h:346-359; the DecodeVal family of version 2
  if (version_ == 2) {
    const int k = v2_index(m);
    return rc_.dec_val(v2_->fixed[k].data(), v2_->maxc[k], v2_->step[k]);
  }
  Ivl e;
  int c = m.decode((int)(rx_ & (kProbScale - 1)), e);
  rans_advance(rx_, in_, e.cum, e.freq);
  dec_count();
  return c;
}
int FrameCodec::get_colour(int plane) {  // decodeC, screencap.h:318-333; UseRC::decodeC, :176-180
  if (version_ == 2) return rc_.dec_uni(v2_->col(plane, cx_ + cx1_), 400);
  ColourCtx& cx = m_->colour[plane][cx_ + cx1_];
  Ivl e;
  uint8_t c;
  if (cx.decode((int)(rx_ & (kProbScale - 1)), c, e))
    rans_advance(rx_, in_, e.cum, e.freq);
  else {
    c = *in_++;
    cx.note_raw(c, true, f0_);
  }
  dec_count();
  return c;
}
void FrameCodec::get_rgb(int& r, int& g, int& b) {  // DecodeRGB, :662-679
  r = get_colour(0);
  SPO_NEXT_CX(r);
  g = get_colour(1);
  SPO_NEXT_CX(g);
  b = get_colour(2);
  SPO_NEXT_CX(b);
}
bool FrameCodec::get_bool() {  // decodeBool, screencap.h:411-421 (version 2: never the same vector, :263-264)
  if (version_ == 2) return false;
  bool flag = (rx_ & (kProbScale - 1)) >= kProbScale / 2;
  rans_advance(rx_, in_, flag ? kProbScale / 2 : 0, kProbScale / 2);
  dec_count();
  return flag;
}

int FrameCodec::decode_intra(const uint8_t* src, uint8_t* dst) {  // DecompressI, :414-498
  int r = 0, g = 0, b = 0;
  dec_begin(src);
  reset_models();
  cx_ = cx1_ = 0;
  int t = 0, last_t = 0, i = 0, n = 1, k = 0, lasti = 0;
  while (k < W + 1) {
    get_rgb(r, g, b);
    n = get_fixed(m_->run_len[t]);
    for (int q = 0; q < n; q++) {
      dst[i] = (uint8_t)r;
      dst[i + 1] = (uint8_t)g;
      dst[i + 2] = (uint8_t)b;
      k++;
      lasti = i;
      i += 3;
      if (i % stride_ >= W * 3) i = (i / stride_ + 1) * stride_;
    }
  }
  const int off = -stride_ - 3;
  int x = (i % stride_) / 3, y = i / stride_;
  while (y < H) {
    last_t = t;
    t = get_fixed(m_->pix_type[last_t]);
    if (!t) get_rgb(r, g, b);
    n = get_fixed(m_->run_len[t]);
    i = y * stride_ + x * 3;
    while (n-- > 0) {
      switch (t) {
        case 0:
          dst[i] = (uint8_t)r;
          dst[i + 1] = (uint8_t)g;
          dst[i + 2] = (uint8_t)b;
          break;
        case 1:
          dst[i] = dst[lasti];
          dst[i + 1] = dst[lasti + 1];
          dst[i + 2] = dst[lasti + 2];
          break;
        case 2:
          dst[i] = dst[i + off + 3];
          dst[i + 1] = dst[i + off + 4];
          dst[i + 2] = dst[i + off + 5];
          break;
        case 4:
          dst[i] = (uint8_t)((int)dst[lasti] + (int)dst[i + off + 3] - (int)dst[i + off]);
          dst[i + 1] = (uint8_t)((int)dst[lasti + 1] + (int)dst[i + off + 4] - (int)dst[i + off + 1]);
          dst[i + 2] = (uint8_t)((int)dst[lasti + 2] + (int)dst[i + off + 5] - (int)dst[i + off + 2]);
          break;
        case 5:
          dst[i] = dst[i + off];
          dst[i + 1] = dst[i + off + 1];
          dst[i + 2] = dst[i + off + 2];
          break;
        default:  // type 3 never occurs in a valid I-frame; the reference writes nothing
          break;
      }
      lasti = i;  // GO_NEXT_PIXEL, :405-410
      x++;
      i += 3;
      if (x >= W) {
        x = 0;
        y++;
        i = y * stride_;
      }
      if (y >= H && n > 0) return 0;  // corrupt stream guard (not in the reference)
    }
    g = dst[lasti + 1];
    b = dst[lasti + 2];
    cx_ = (uint32_t)g >> 2;
    SPO_NEXT_CX(b);
  }
  memcpy(prev_.data(), dst, (size_t)H * stride_);
  return 1;
}

int FrameCodec::decode_inter(const uint8_t* src, uint8_t* dst) {  // DecompressP, :1275-1432
  uint8_t* pv = prev_.data();
  int first = *src++;
  if (!(first & 1)) {
    memcpy(dst, pv, (size_t)H * stride_);
    return 1;
  }
  dec_begin(src);
  int lo = get_fixed(m_->blk_index), hi = get_fixed(m_->blk_index);
  int xx1 = (hi << 8) + lo;
  lo = get_fixed(m_->blk_index);
  hi = get_fixed(m_->blk_index);
  int xx2 = (hi << 8) + lo;
  if (xx2 >= nbx * nby || xx1 > xx2) return 0;  // guard (not in the reference)
  std::fill(blk_types.begin(), blk_types.end(), 0);
  for (int b = xx1; b <= xx2;) {
    int c = get_fixed(m_->blk_type);
    int n = get_fixed(m_->blk_run);
    if (n <= 0 || b + n > nbx * nby) return 0;  // guard
    for (int q = 0; q < n; q++) blk_types[b++] = (uint8_t)c;
  }
  const int off = -stride_ - 3;
  cx_ = cx1_ = 0;
  int lastmx = 0, lastmy = 0;
  for (int by = 0; by < nby; by++)
    for (int bx = 0; bx < nbx; bx++) {
      int x16 = bx * 16, y16 = by * 16;
      int x1 = x16, x2 = imin(x16 + 16, W), y1 = y16, y2 = imin(y16 + 16, H);
      int bi = by * nbx + bx, bt = blk_types[bi];
      if (!bt) {
        for (int y = y1; y < y2; y++) memcpy(dst + y * stride_ + x1 * 3, pv + y * stride_ + x1 * 3, (x2 - x1) * 3);
        continue;
      }
      if ((bt - 1) & 1) {
        for (int y = y1; y < y2; y++) memcpy(dst + y * stride_ + x1 * 3, pv + y * stride_ + x1 * 3, (x2 - x1) * 3);
        x1 = get_fixed(m_->rect[0]) + x16;
        y1 = get_fixed(m_->rect[1]) + y16;
        x2 = get_fixed(m_->rect[2]) + x16 + 1;
        y2 = get_fixed(m_->rect[3]) + y16 + 1;
        if (x2 > W || y2 > H || x1 >= x2 || y1 >= y2) return 0;  // guard
      }
      if ((bt - 1) & 2) {
        int mx, my;
        if (get_bool()) {
          mx = lastmx;
          my = lastmy;
        } else {
          mx = get_fixed(m_->motion[0]) - (int)far_x;
          my = get_fixed(m_->motion[1]) - (int)far_y;
        }
        lastmx = mx;
        lastmy = my;
        if (x1 + mx < 0 || y1 + my < 0 || x2 + mx > W || y2 + my > H) return 0;  // guard
        for (int y = y1; y < y2; y++) memcpy(dst + y * stride_ + x1 * 3, pv + (y + my) * stride_ + (x1 + mx) * 3, (x2 - x1) * 3);
      } else {
        int x = x1, y = y1, t = 0, last_t = 0, r = 0, g = 0, b = 0;
        while (y < y2) {
          int i = y * stride_ + x * 3;
          last_t = t;
          t = get_fixed(m_->pix_type[last_t]);
          if (!t) get_rgb(r, g, b);
          int n = get_fixed(m_->run_len[t]);
          for (int q = 0; q < n; q++) {
            if (y >= y2) return 0;  // guard
            switch (t) {
              case 1: r = dst[i - 3]; g = dst[i - 2]; b = dst[i - 1]; break;
              case 2: r = dst[i + off + 3]; g = dst[i + off + 4]; b = dst[i + off + 5]; break;
              case 3: r = pv[i]; g = pv[i + 1]; b = pv[i + 2]; break;
              case 4:
                r = (int)dst[i - 3] + (int)dst[i + off + 3] - (int)dst[i + off];
                g = (int)dst[i - 2] + (int)dst[i + off + 4] - (int)dst[i + off + 1];
                b = (int)dst[i - 1] + (int)dst[i + off + 5] - (int)dst[i + off + 2];
                break;
              case 5: r = dst[i + off]; g = dst[i + off + 1]; b = dst[i + off + 2]; break;
            }
            dst[i] = (uint8_t)r;
            dst[i + 1] = (uint8_t)g;
            dst[i + 2] = (uint8_t)b;
            i += 3;
            if (++x >= x2) {
              x = x1;
              y++;
              i = y * stride_ + x * 3;
            }
          }
          cx_ = (uint32_t)(g & 255) >> 2;
          SPO_NEXT_CX(b & 255);
        }
      }
    }
  memcpy(pv, dst, (size_t)H * stride_);
  return 1;
}

int FrameCodec::decompress(const uint8_t* src, int src_len, uint8_t* dst, int ftype) {  // :1522-1557
  dec_len_ = src_len;  // version 2 hands the packet length to its coder as it is (one more than what follows the header byte)
  if (W & 3) {
    int pad = stride_ - W * 3;
    for (int y = 0; y < H; y++) memset(dst + (size_t)y * stride_ + W * 3, 0, pad);
  }
  frames_++;
  if (ftype) {
    last_flat_ = false;
    return decode_inter(src, dst);
  }
  int alg = *src++ & 15;
  if (alg == 1) {
    for (int x = 0; x < W; x++) memcpy(dst + x * 3, src, 3);
    for (int y = 1; y < H; y++) memcpy(dst + (size_t)y * stride_, dst, (size_t)W * 3);
    if (!(last_flat_ && !memcmp(last_flat_rgb_, src, 3))) {
      memcpy(prev_.data(), dst, (size_t)H * stride_);
      reset_models();
    }
    last_flat_ = true;
    memcpy(last_flat_rgb_, src, 3);
    return 1;
  }
  last_flat_ = false;
  return decode_intra(src, dst);
}

// ------------------------------ ScreenCodec --------------------------------
void ScreenCodec::init(const Params& p) {  // :1565-1584
  p_ = p;
  W = p.width;
  H = p.height;
  bpp_ = p.bits_per_pixel / 8;
  stride_ = (W * bpp_ + 3) & ~3u;
  rgb32_ = p.bits_per_pixel == 32;
  rgb16_ = p.bits_per_pixel == 16;
  last_loss_ = (int)p.loss;
  rs_ = gs_ = bs_ = 0;
  if (rgb16_) {
    while (rs_ < 16 && !((1u << rs_) & p.red_mask)) rs_++;
    while (gs_ < 16 && !((1u << gs_) & p.green_mask)) gs_++;
    while (bs_ < 16 && !((1u << bs_) & p.blue_mask)) bs_++;
  }
}

void ScreenCodec::create(int version) {  // CreateCodec, :1587-1617
  const uint32_t stride24 = (W * 3 + 3) & ~3u;
  if (rgb32_ || rgb16_) buf_.assign((size_t)stride24 * H, 0);
  fc_ = new FrameCodec(p_, version);
  fc_->set_threads(threads_);
}

void ScreenCodec::deinit() {  // :1619-1629
  if (crashed_) return;
  delete fc_;
  fc_ = nullptr;
  buf_.clear();
  rgb32_ = rgb16_ = false;
}

int ScreenCodec::compress_frame(uint8_t* src, uint8_t* dst, int dst_len, int* ftype, int loss) {  // :1632-1692
  if (crashed_) return 0;
  if (p_.bits_per_pixel != 16 && p_.bits_per_pixel != 24 && p_.bits_per_pixel != 32) return -48;
  if (!fc_) create(p_.version == 3 ? 3 : p_.version == 2 ? 2 : 4);
  if (loss != last_loss_) {
    fc_->set_loss(loss);
    last_loss_ = loss;
  }
  const uint32_t stride24 = (W * 3 + 3) & ~3u;
  if (rgb32_) {
    for (uint32_t y = 0; y < H; y++) {
      const uint8_t* s = src + (size_t)y * W * 4;
      uint8_t* d = buf_.data() + (size_t)y * stride24;
      for (uint32_t x = 0; x < W; x++, s += 4, d += 3) {
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
      }
    }
    src = buf_.data();
  } else if (rgb16_) {
    for (uint32_t y = 0; y < H; y++) {
      const uint8_t* s = src + (size_t)y * W * 2;
      uint8_t* d = buf_.data() + (size_t)y * stride24;
      for (uint32_t x = 0; x < W; x++, s += 2, d += 3) {
        uint16_t w = (uint16_t)(s[0] | (s[1] << 8));
        d[0] = (uint8_t)((w & p_.red_mask) >> rs_);
        d[1] = (uint8_t)((w & p_.green_mask) >> gs_);
        d[2] = (uint8_t)((w & p_.blue_mask) >> bs_);
      }
    }
    src = buf_.data();
  }
  return fc_->compress(src, dst, dst_len, *ftype);
}

int ScreenCodec::decompress_frame(const uint8_t* src, int src_len, uint8_t* dst, int pitch, int ftype) {  // :1695-1743
  if (crashed_ && ftype > 0) return 0;
  if (!fc_) {
    if (ftype > 0) return 0;
    int version = (src[0] >> 4) + 1;
    if (version < 2 || version > 4) return -version;  // BadVersionException, :1589-1590
    if (p_.bits_per_pixel != 16 && p_.bits_per_pixel != 24 && p_.bits_per_pixel != 32) return -48;
    create(version);
  }
  const uint32_t stride24 = (W * 3 + 3) & ~3u;
  bool use_buf = rgb32_ || rgb16_;
  if ((uint32_t)pitch != stride_) {
    buf_.resize((size_t)stride24 * H, 0);
    use_buf = true;
  }
  crashed_ = false;
  if (!use_buf) return fc_->decompress(src, src_len, dst, ftype);
  int ret = fc_->decompress(src, src_len, buf_.data(), ftype);
  for (uint32_t y = 0; y < H; y++) {
    const uint8_t* s = buf_.data() + (size_t)y * stride24;
    uint8_t* d = dst + (size_t)y * pitch;
    if (bpp_ == 4) {
      for (uint32_t x = 0; x < W; x++, s += 3, d += 4) {
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
        d[3] = 255;
      }
    } else if (bpp_ == 2) {
      for (uint32_t x = 0; x < W; x++, s += 3, d += 2) {
        uint16_t w = (uint16_t)((s[0] << rs_) + (s[1] << gs_) + (s[2] << bs_));
        d[0] = (uint8_t)w;
        d[1] = (uint8_t)(w >> 8);
      }
    } else {
      memcpy(d, s, (size_t)W * 3);
    }
  }
  return ret;
}

}  // namespace spo
