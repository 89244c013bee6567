// scpr_avi.cpp — RIFF AVI 1.0 reader / writer for one video stream (include/scpr_avi.h).
// Host-side I/O only; no codec logic lives here.
#include "../../include/scpr_avi.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t fcc(char a, char b, char c, char d) { return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24); }
constexpr uint32_t kRIFF = fcc('R', 'I', 'F', 'F'), kAVI = fcc('A', 'V', 'I', ' '), kLIST = fcc('L', 'I', 'S', 'T'), kHdrl = fcc('h', 'd', 'r', 'l'),
                   kAvih = fcc('a', 'v', 'i', 'h'), kStrl = fcc('s', 't', 'r', 'l'), kStrh = fcc('s', 't', 'r', 'h'), kStrf = fcc('s', 't', 'r', 'f'),
                   kMovi = fcc('m', 'o', 'v', 'i'), kIdx1 = fcc('i', 'd', 'x', '1'), kVids = fcc('v', 'i', 'd', 's'), k00dc = fcc('0', '0', 'd', 'c'),
                   k00db = fcc('0', '0', 'd', 'b'), kRec = fcc('r', 'e', 'c', ' ');
constexpr uint32_t kAvifHasIndex = 0x10;
constexpr uint64_t kMaxFile = 0xFFF00000ull;  // stay clear of the 32-bit RIFF size

struct IndexEntry {
  uint32_t ckid, flags, offset, size;
};
struct Frame {
  uint64_t pos;  // file offset of the chunk data
  uint32_t size, flags;
};

void put32(std::vector<uint8_t>& b, uint32_t v) {
  for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
}
void put16(std::vector<uint8_t>& b, uint32_t v) {
  b.push_back((uint8_t)v);
  b.push_back((uint8_t)(v >> 8));
}
uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t get16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

bool is_uncompressed(uint32_t c) { return c == SCPR_BI_RGB || c == SCPR_BI_BITFIELDS; }

}  // namespace

struct scpr_avi_writer {
  FILE* f = nullptr;
  scpr_format fmt{};
  uint32_t rate = 25, scale = 1, ckid = k00dc;
  std::vector<IndexEntry> index;
  uint64_t movi_fourcc_pos = 0, pos = 0;
  uint32_t max_chunk = 0;
  bool failed = false;

  // headers for the current frame count; called twice (provisional, final): the layout has a fixed size
  std::vector<uint8_t> headers(uint32_t riff_size, uint32_t movi_size) const {
    const bool bitfields = fmt.bit_count == 16;
    const uint32_t strf_size = 40 + (bitfields ? 12 : 0);
    const uint32_t strl_size = 4 + (8 + 56) + (8 + strf_size);
    const uint32_t hdrl_size = 4 + (8 + 56) + (8 + strl_size);
    const uint32_t frames = (uint32_t)index.size();
    std::vector<uint8_t> b;
    put32(b, kRIFF);
    put32(b, riff_size);
    put32(b, kAVI);
    put32(b, kLIST);
    put32(b, hdrl_size);
    put32(b, kHdrl);
    put32(b, kAvih);
    put32(b, 56);
    put32(b, (uint32_t)(1000000ull * scale / (rate ? rate : 1)));  // dwMicroSecPerFrame
    put32(b, 0);                                                   // dwMaxBytesPerSec
    put32(b, 0);                                                   // dwPaddingGranularity
    put32(b, kAvifHasIndex);                                       // dwFlags
    put32(b, frames);                                              // dwTotalFrames
    put32(b, 0);                                                   // dwInitialFrames
    put32(b, 1);                                                   // dwStreams
    put32(b, max_chunk);                                           // dwSuggestedBufferSize
    put32(b, fmt.width);
    put32(b, fmt.height);
    for (int i = 0; i < 4; i++) put32(b, 0);
    put32(b, kLIST);
    put32(b, strl_size);
    put32(b, kStrl);
    put32(b, kStrh);
    put32(b, 56);
    put32(b, kVids);
    put32(b, is_uncompressed(fmt.compression) ? SCPR_FOURCC_DIB : fmt.compression);  // fccHandler
    put32(b, 0);        // dwFlags
    put16(b, 0);        // wPriority
    put16(b, 0);        // wLanguage
    put32(b, 0);        // dwInitialFrames
    put32(b, scale);    // dwScale
    put32(b, rate);     // dwRate
    put32(b, 0);        // dwStart
    put32(b, frames);   // dwLength
    put32(b, max_chunk);
    put32(b, 0xFFFFFFFFu);  // dwQuality: default
    put32(b, 0);            // dwSampleSize: variable
    put16(b, 0);
    put16(b, 0);
    put16(b, fmt.width);
    put16(b, fmt.height);
    put32(b, kStrf);
    put32(b, strf_size);
    put32(b, strf_size);  // biSize covers the masks (screenpressor.cpp:328)
    put32(b, fmt.width);
    put32(b, fmt.height);
    put16(b, 1);
    put16(b, fmt.bit_count);
    put32(b, fmt.compression);
    put32(b, fmt.size_image ? fmt.size_image : ((fmt.width * fmt.bit_count / 8 + 3) & ~3u) * fmt.height);
    put32(b, 0);
    put32(b, 0);
    put32(b, 0);
    put32(b, 0);
    if (bitfields)
      for (int i = 0; i < 3; i++) put32(b, fmt.masks[i]);
    put32(b, kLIST);
    put32(b, movi_size);
    put32(b, kMovi);
    return b;
  }
};

struct scpr_avi_reader {
  FILE* f = nullptr;
  scpr_avi_info info{};
  std::vector<Frame> frames;
};

extern "C" {

scpr_avi_writer* scpr_avi_create(const char* path, const scpr_format* fmt, uint32_t rate, uint32_t scale) {
  if (!path || !fmt || !fmt->width || !fmt->height) return nullptr;
  FILE* f = std::fopen(path, "wb");
  if (!f) return nullptr;
  scpr_avi_writer* w = new scpr_avi_writer;
  w->f = f;
  w->fmt = *fmt;
  w->rate = rate ? rate : 25;
  w->scale = scale ? scale : 1;
  w->ckid = is_uncompressed(fmt->compression) ? k00db : k00dc;
  const std::vector<uint8_t> h = w->headers(0, 0);
  if (std::fwrite(h.data(), 1, h.size(), f) != h.size()) {
    std::fclose(f);
    delete w;
    return nullptr;
  }
  w->pos = h.size();
  w->movi_fourcc_pos = h.size() - 4;
  return w;
}

int scpr_avi_write(scpr_avi_writer* w, const void* data, uint32_t size, uint32_t flags) {
  if (!w || w->failed || (!data && size)) return SCPR_E_PARAM;
  const uint64_t need = 8ull + size + (size & 1) + 16ull * (w->index.size() + 1) + 8;
  if (w->pos + need > kMaxFile) return SCPR_E_CAPACITY;
  uint8_t hd[8];
  std::memcpy(hd, &w->ckid, 4);
  for (int i = 0; i < 4; i++) hd[4 + i] = (uint8_t)(size >> (8 * i));
  bool ok = std::fwrite(hd, 1, 8, w->f) == 8 && (size == 0 || std::fwrite(data, 1, size, w->f) == size);
  if (ok && (size & 1)) ok = std::fputc(0, w->f) != EOF;  // chunks are word aligned
  if (!ok) {
    w->failed = true;
    return SCPR_E_DEVICE;
  }
  w->index.push_back({w->ckid, flags & SCPR_FRAME_KEY, (uint32_t)(w->pos - w->movi_fourcc_pos), size});
  w->pos += 8ull + size + (size & 1);
  if (size > w->max_chunk) w->max_chunk = size;
  return SCPR_OK;
}

int scpr_avi_finish(scpr_avi_writer* w) {
  if (!w) return SCPR_E_PARAM;
  bool ok = !w->failed;
  const uint32_t movi_size = (uint32_t)(w->pos - w->movi_fourcc_pos);
  std::vector<uint8_t> idx;
  put32(idx, kIdx1);
  put32(idx, (uint32_t)(16 * w->index.size()));
  for (const IndexEntry& e : w->index) {
    put32(idx, e.ckid);
    put32(idx, e.flags);
    put32(idx, e.offset);
    put32(idx, e.size);
  }
  ok = ok && std::fwrite(idx.data(), 1, idx.size(), w->f) == idx.size();
  const uint64_t total = w->pos + idx.size();
  const std::vector<uint8_t> h = w->headers((uint32_t)(total - 8), movi_size);
  ok = ok && std::fseek(w->f, 0, SEEK_SET) == 0 && std::fwrite(h.data(), 1, h.size(), w->f) == h.size();
  ok = (std::fclose(w->f) == 0) && ok;
  delete w;
  return ok ? SCPR_OK : SCPR_E_DEVICE;
}

scpr_avi_reader* scpr_avi_open(const char* path) {
  if (!path) return nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return nullptr;
  scpr_avi_reader* r = new scpr_avi_reader;
  r->f = f;
  auto fail = [&]() -> scpr_avi_reader* {
    std::fclose(f);
    delete r;
    return nullptr;
  };
  uint8_t b[12];
  if (std::fread(b, 1, 12, f) != 12 || get32(b) != kRIFF || get32(b + 8) != kAVI) return fail();
  std::fseek(f, 0, SEEK_END);
  const uint64_t file_size = (uint64_t)std::ftell(f);
  uint64_t movi_fourcc = 0, movi_end = 0, idx_pos = 0;
  uint32_t idx_size = 0;
  bool have_vids = false, in_vids = false, have_fmt = false;
  int stream = -1, vid_stream = -1;
  // walk the top-level chunks; descend into hdrl / strl
  struct Range {
    uint64_t pos, end;
  };
  std::vector<Range> stack{{12, file_size}};
  while (!stack.empty()) {
    Range& top = stack.back();
    if (top.pos + 8 > top.end) {
      stack.pop_back();
      continue;
    }
    std::fseek(f, (long)top.pos, SEEK_SET);
    if (std::fread(b, 1, 8, f) != 8) break;
    const uint32_t id = get32(b), size = get32(b + 4);
    const uint64_t data = top.pos + 8, next = data + size + (size & 1);
    top.pos = next;
    if (id == kLIST) {
      if (size < 4 || std::fread(b, 1, 4, f) != 4) break;
      const uint32_t type = get32(b);
      if (type == kMovi) {
        movi_fourcc = data;
        movi_end = data + size;
      } else if (type == kHdrl || type == kStrl) {
        if (type == kStrl) {
          stream++;
          in_vids = false;
        }
        stack.push_back({data + 4, data + size});
      }
    } else if (id == kAvih && size >= 56) {
      uint8_t h[56];
      if (std::fread(h, 1, 56, f) != 56) break;
      r->info.frames = get32(h + 16);
    } else if (id == kStrh && size >= 48) {
      uint8_t h[56] = {0};
      if (std::fread(h, 1, size < 56 ? size : 56, f) < 48) break;
      if (get32(h) == kVids && !have_vids) {
        have_vids = in_vids = true;
        vid_stream = stream;
        r->info.handler = get32(h + 4);
        r->info.scale = get32(h + 20);
        r->info.rate = get32(h + 24);
        r->info.frames = get32(h + 32);
      }
    } else if (id == kStrf && in_vids && !have_fmt && size >= 40) {
      uint8_t h[52] = {0};
      const uint32_t n = size < 52 ? size : 52;
      if (std::fread(h, 1, n, f) != n) break;
      scpr_format& fm = r->info.format;
      fm.width = get32(h + 4);
      const uint32_t hh = get32(h + 8);  // biHeight is signed (negative: a top-down DIB); its magnitude, also for the most negative value
      fm.height = (hh & 0x80000000u) ? 0u - hh : hh;
      fm.bit_count = get16(h + 14);
      fm.compression = get32(h + 16);
      fm.size_image = get32(h + 20);
      if (n >= 52)
        for (int i = 0; i < 3; i++) fm.masks[i] = get32(h + 40 + 4 * i);
      have_fmt = true;
    } else if (id == kIdx1) {
      idx_pos = data;
      idx_size = size;
    }
  }
  if (!have_fmt || !movi_fourcc) return fail();
  const uint32_t want_hi = (uint32_t)('0' + vid_stream / 10) | ((uint32_t)('0' + vid_stream % 10) << 8);
  auto is_video_chunk = [&](uint32_t id) { return (id & 0xFFFF) == want_hi && ((id >> 16) == (('d') | ('c' << 8)) || (id >> 16) == (('d') | ('b' << 8))); };
  if (idx_pos && idx_size >= 16 && idx_pos + idx_size <= file_size) {  // (an index that claims more bytes than the file has is no index: the movi list is walked)
    std::vector<uint8_t> idx(idx_size);
    std::fseek(f, (long)idx_pos, SEEK_SET);
    if (std::fread(idx.data(), 1, idx_size, f) != idx_size) return fail();
    // offsets are from the 'movi' fourcc in files written by AVIFile, from the file start in some others:
    // the first video entry tells which, by where its chunk header really is
    int64_t base = -1;
    for (uint32_t i = 0; i + 16 <= idx_size; i += 16) {
      const uint32_t id = get32(&idx[i]), flags = get32(&idx[i + 4]), off = get32(&idx[i + 8]), size = get32(&idx[i + 12]);
      if (!is_video_chunk(id)) continue;
      if (base < 0) {
        for (uint64_t cand : {movi_fourcc, (uint64_t)0}) {
          std::fseek(f, (long)(cand + off), SEEK_SET);
          if (std::fread(b, 1, 8, f) == 8 && get32(b) == id && get32(b + 4) == size) {
            base = (int64_t)cand;
            break;
          }
        }
        if (base < 0) break;
      }
      r->frames.push_back({(uint64_t)base + off + 8, size, flags});
    }
  }
  if (r->frames.empty()) {  // no usable index: walk the movi list (every frame is taken as a key frame candidate: flags 0)
    std::vector<Range> st{{movi_fourcc + 4, movi_end}};
    while (!st.empty()) {
      Range& top = st.back();
      if (top.pos + 8 > top.end) {
        st.pop_back();
        continue;
      }
      std::fseek(f, (long)top.pos, SEEK_SET);
      if (std::fread(b, 1, 8, f) != 8) break;
      const uint32_t id = get32(b), size = get32(b + 4);
      const uint64_t data = top.pos + 8;
      top.pos = data + size + (size & 1);
      if (id == kLIST) {
        if (std::fread(b, 1, 4, f) == 4 && get32(b) == kRec) st.push_back({data + 4, data + size});
      } else if (is_video_chunk(id)) {
        r->frames.push_back({data, size, 0});
      }
    }
  }
  r->info.frames = (uint32_t)r->frames.size();
  return r;
}

int scpr_avi_get_info(const scpr_avi_reader* r, scpr_avi_info* info) {
  if (!r || !info) return SCPR_E_PARAM;
  *info = r->info;
  return SCPR_OK;
}

int64_t scpr_avi_frame_size(const scpr_avi_reader* r, uint32_t index, uint32_t* flags) {
  if (!r || index >= r->frames.size()) return SCPR_E_PARAM;
  if (flags) *flags = r->frames[index].flags;
  return r->frames[index].size;
}

int64_t scpr_avi_read(scpr_avi_reader* r, uint32_t index, void* buf, uint64_t capacity, uint32_t* flags) {
  if (!r || index >= r->frames.size() || !buf) return SCPR_E_PARAM;
  const Frame& fr = r->frames[index];
  if (fr.size > capacity) return SCPR_E_CAPACITY;
  if (std::fseek(r->f, (long)fr.pos, SEEK_SET) != 0 || std::fread(buf, 1, fr.size, r->f) != fr.size) return SCPR_E_DEVICE;
  if (flags) *flags = fr.flags;
  return fr.size;
}

void scpr_avi_close(scpr_avi_reader* r) {
  if (!r) return;
  std::fclose(r->f);
  delete r;
}

}  // extern "C"
