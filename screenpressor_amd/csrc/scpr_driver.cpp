// scpr_driver.cpp — codec-instance policy in front of the GPU codec (include/scpr_driver.h).
//
// Restates the decisions of the reference's CodecInst (screenpressor.cpp:276-650): which formats
// are accepted, when a key frame is asked for, how the host's quality becomes a loss level, how
// the frame type of a packet is found on decode.  Host-side control logic only; every frame goes
// through scpr_compress_frame / scpr_decompress_frame of the GPU library.
#include "../../include/scpr_driver.h"

#include <algorithm>
#include <cstring>

struct scpr_driver {
  scpr_codec* codec = nullptr;
  scpr_driver_config cfg{500, 0, 0, 0, 1};
  uint32_t masks[3] = {0x7C00, 0x3E0, 0x1F};  // rmask/gmask/bmask members (screenpressor.h)
  uint32_t npframes = 0;                       // P-frames since the last key frame (screenpressor.cpp:367, :425-431)
  bool compressing = false, decompressing = false;
  scpr_format dec_in{};                        // stream format given to DecompressBegin
  uint32_t dec_size_image = 0;
};

namespace {

bool uncompressed(uint32_t fourcc) { return fourcc == SCPR_BI_RGB || fourcc == SCPR_FOURCC_DIB || fourcc == SCPR_BI_BITFIELDS; }

// CanCompress (screenpressor.cpp:276-305).  For 16-bit input it also latches the colour masks,
// as the reference does as a side effect.
bool can_compress(scpr_driver* d, const scpr_format* f) {
  if (!f || !uncompressed(f->compression)) return false;
  if (f->bit_count == 24 || f->bit_count == 32) return true;
  if (f->bit_count != 16) return false;
  if (f->compression == SCPR_BI_BITFIELDS) {
    std::memcpy(d->masks, f->masks, sizeof d->masks);
  } else {
    d->masks[0] = 0x7C00;
    d->masks[1] = 0x3E0;
    d->masks[2] = 0x1F;
  }
  return true;
}

// CanDecompress (:449-493)
bool can_decompress(scpr_driver* d, const scpr_format* in, const scpr_format* out) {
  if (!in) return false;
  if (!out) return in->compression == SCPR_FOURCC_SCPR;
  if (out->width != in->width || out->height != in->height) return false;  // 1:1 only
  if (in->bit_count > 16 && out->bit_count != 24 && out->bit_count != 32) return false;
  if (in->bit_count != out->bit_count) return false;
  if (in->bit_count == 16 && out->compression != SCPR_BI_BITFIELDS) {
    // without BITFIELDS the output can only be 555: the stream must be 555 too
    return in->masks[0] == 0x7C00 && in->masks[1] == 0x3E0 && in->masks[2] == 0x1F;
  }
  return can_compress(d, out);
}

uint32_t dib_stride(uint32_t width, uint32_t bits) { return (width * bits / 8 + 3) & ~3u; }

scpr_params make_params(const scpr_driver* d, const scpr_format* f, uint32_t loss) {
  scpr_params p{};
  p.width = f->width;
  p.height = f->height;
  p.bits_per_pixel = f->bit_count;
  p.red_mask = d->masks[0];
  p.green_mask = d->masks[1];
  p.blue_mask = d->masks[2];
  p.high_range_x = p.high_range_y = 256;  // screenpressor.cpp:378-379, :571-572
  p.low_range_x = p.low_range_y = 8;
  p.loss = loss;
  p.workers = d->cfg.workers ? d->cfg.workers : 1;
  return p;
}

}  // namespace

extern "C" {

scpr_driver* scpr_driver_open(int device) {
  scpr_codec* c = scpr_create(device);
  if (!c) return nullptr;
  scpr_driver* d = new scpr_driver;
  d->codec = c;
  return d;
}

void scpr_driver_close(scpr_driver* d) {
  if (!d) return;
  scpr_destroy(d->codec);
  delete d;
}

void scpr_driver_configure(scpr_driver* d, const scpr_driver_config* cfg) {
  if (!d) return;
  d->cfg = cfg ? *cfg : scpr_driver_config{500, 0, 0, 0, 1};
  if (d->cfg.key_frame_interval == 0) d->cfg.key_frame_interval = 1;
}

int scpr_driver_compress_query(scpr_driver* d, const scpr_format* in) {
  if (!d) return SCPR_E_PARAM;
  return can_compress(d, in) ? SCPR_OK : SCPR_E_BADFORMAT;
}

int scpr_driver_compress_get_format(scpr_driver* d, const scpr_format* in, scpr_format* out) {
  if (!d || !out) return SCPR_E_PARAM;
  if (!can_compress(d, in)) return SCPR_E_BADFORMAT;
  *out = *in;  // width, height, bit count and size carried over (:327)
  out->compression = SCPR_FOURCC_SCPR;
  if (in->bit_count == 16) std::memcpy(out->masks, d->masks, sizeof d->masks);  // the three DWORDs after the header (:330-336)
  return SCPR_OK;
}

uint32_t scpr_driver_compress_get_size(const scpr_format* in) { return in ? in->width * in->height * 6 : 0; }

int scpr_driver_compress_begin(scpr_driver* d, const scpr_format* in) {
  if (!d) return SCPR_E_PARAM;
  if (!can_compress(d, in)) return SCPR_E_BADFORMAT;
  scpr_driver_compress_end(d);  // free resources if necessary (:353)
  d->npframes = 0;
  const scpr_params p = make_params(d, in, d->cfg.loss);
  const int rc = scpr_init(d->codec, &p);
  if (rc == SCPR_E_BAD_VERSION) return SCPR_E_BADFORMAT;
  if (rc != SCPR_OK) return rc;
  d->compressing = true;
  return SCPR_OK;
}

int scpr_driver_compress_end(scpr_driver* d) {
  if (!d) return SCPR_E_PARAM;
  scpr_deinit(d->codec);
  d->compressing = false;
  return SCPR_OK;
}

int scpr_driver_compress(scpr_driver* d, const void* in, void* out, uint32_t out_capacity, uint32_t quality, int host_keyframe, uint32_t* out_size,
                         uint32_t* out_flags) {
  if (!d || !d->compressing || !in || !out || !out_size || !out_flags) return SCPR_E_PARAM;
  // key-frame policy (:402-406): either the configured interval rules, or the host's flag does
  int ftype = 1;
  const bool forced_kf = d->cfg.force_interval && d->npframes + 1 >= d->cfg.key_frame_interval;
  const bool host_kf = !d->cfg.force_interval && host_keyframe;
  if (host_kf || forced_kf) ftype = 0;
  // quality -> loss (:410-422): 0-2000 -> 4, ..., 8001-10000 -> 0
  uint32_t loss = d->cfg.loss;
  if (!d->cfg.force_loss) loss = std::min((10000u - std::min(quality, 10000u)) / 2000u, 4u);
  const int sz = scpr_compress_frame(d->codec, in, out, (int)std::min<uint32_t>(out_capacity, 0x7FFFFFFFu), &ftype, (int)loss);
  if (sz < 0) return sz;
  if (!ftype) {
    *out_flags = SCPR_FRAME_KEY;
    d->npframes = 0;
  } else {
    *out_flags = 0;
    d->npframes++;
  }
  *out_size = (uint32_t)sz;
  return SCPR_OK;
}

int scpr_driver_decompress_query(scpr_driver* d, const scpr_format* in, const scpr_format* out) {
  if (!d) return SCPR_E_PARAM;
  return can_decompress(d, in, out) ? SCPR_OK : SCPR_E_BADFORMAT;
}

int scpr_driver_decompress_get_format(scpr_driver* d, const scpr_format* in, scpr_format* out) {
  if (!d || !out) return SCPR_E_PARAM;
  if (!can_decompress(d, in, nullptr)) return SCPR_E_BADFORMAT;
  *out = *in;  // masks copied also (:520)
  out->compression = in->bit_count == 16 ? SCPR_BI_BITFIELDS : SCPR_BI_RGB;
  const uint32_t bpp = out->bit_count / 8;
  out->size_image = ((in->width * bpp + 3) & ~3u) * in->height;
  d->dec_size_image = out->size_image;
  return SCPR_OK;
}

int scpr_driver_decompress_begin(scpr_driver* d, const scpr_format* in, const scpr_format* out) {
  if (!d) return SCPR_E_PARAM;
  scpr_driver_decompress_end(d);  // free resources if necessary (:538)
  if (!can_decompress(d, in, out)) return SCPR_E_BADFORMAT;
  if (out && out->bit_count == 16) std::memcpy(d->masks, in->masks, sizeof d->masks);  // the stream's masks rule (:548-553)
  d->dec_in = *in;
  d->dec_size_image = dib_stride(in->width, in->bit_count) * in->height;
  const scpr_params p = make_params(d, in, 0);
  const int rc = scpr_init(d->codec, &p);
  if (rc == SCPR_E_BAD_VERSION) return SCPR_E_BADFORMAT;
  if (rc != SCPR_OK) return rc;
  d->decompressing = true;
  return SCPR_OK;
}

int scpr_driver_decompress_end(scpr_driver* d) {
  if (!d) return SCPR_E_PARAM;
  scpr_deinit(d->codec);
  d->decompressing = false;
  return SCPR_OK;
}

int scpr_infer_frame_type(uint8_t first_byte, uint32_t data_size) {
  switch (first_byte) {  // headers of the v1/v2 streams; v3/v4 headers (0x2x, 0x3x) are not told apart here
    case 0x00: return 1;
    case 0x01: return data_size <= 4 ? 0 : 1;
    case 0x02:
    case 0x11:
    case 0x12: return 0;
    default: return -1;
  }
}

int scpr_driver_decompress(scpr_driver* d, const void* in, uint32_t in_size, void* out, int not_keyframe) {
  if (!d || !d->decompressing || !in || !out || in_size < 1) return SCPR_E_PARAM;
  int ftype = not_keyframe ? 1 : 0;
  const int inferred = scpr_infer_frame_type(*(const uint8_t*)in, in_size);
  if (inferred >= 0) ftype = inferred;
  const int stride = (int)dib_stride(d->dec_in.width, d->dec_in.bit_count);
  const int r = scpr_decompress_frame(d->codec, in, (int)in_size, out, stride, ftype);
  if (r == SCPR_E_BAD_VERSION) return SCPR_E_BADFORMAT;  // BadVersionException -> ICERR_BADFORMAT (:621-636)
  if (r < 0) return r;
  return SCPR_OK;  // a refused frame (0) is not an error to the host either (:620)
}

}  // extern "C"
