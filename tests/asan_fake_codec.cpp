// Test infrastructure (CPU sanitizer build only, tests/test_sanitizers.py): the entry points of include/scpr_amd.h that
// csrc/scpr_driver.cpp calls, as a RECORDING FAKE - no device, no codec.  It lets the policy layer (key-frame interval, quality ->
// loss, frame-type inference, format negotiation; CodecInst, screenpressor.cpp:276-638) run under AddressSanitizer /
// UndefinedBehaviorSanitizer on the CPU box.  A "packet" of the fake is 6 bytes: header byte (0x32 key / 0x01 P), the loss it was
// asked for, and the frame number; "decoding" one fills the picture with the frame number.
#include <cstdint>
#include <cstring>
#include "../include/scpr_amd.h"

struct scpr_codec {
  scpr_params p{};
  bool inited = false;
  uint32_t frames = 0;
};

extern "C" {
scpr_codec* scpr_create(int device) { return device == 0 ? new scpr_codec : nullptr; }
void scpr_destroy(scpr_codec* c) { delete c; }
int scpr_init(scpr_codec* c, const scpr_params* p) {
  if (!c || !p) return SCPR_E_PARAM;
  if (p->width < 3 || p->height < 2) return SCPR_E_PARAM;
  c->p = *p;
  c->inited = true;
  c->frames = 0;
  return SCPR_OK;
}
void scpr_deinit(scpr_codec* c) {
  if (c) c->inited = false;
}
int scpr_compress_frame(scpr_codec* c, const void* src, void* dst, int dst_len, int* ftype, int loss) {
  if (!c || !c->inited || !src || !dst || !ftype || dst_len < 6) return SCPR_E_PARAM;
  if (c->frames == 0) *ftype = 0;  // the first frame is a key frame whatever is asked (screencap.cpp:1504)
  uint8_t* o = (uint8_t*)dst;
  o[0] = *ftype == 0 ? 0x32 : 0x01;
  o[1] = (uint8_t)loss;
  std::memcpy(o + 2, &c->frames, 4);
  c->frames++;
  return 6;
}
int scpr_decompress_frame(scpr_codec* c, const void* src, int src_len, void* dst, int pitch, int ftype) {
  if (!c || !c->inited || !src || !dst || src_len < 1) return SCPR_E_PARAM;
  (void)ftype;
  const uint8_t* s = (const uint8_t*)src;
  std::memset(dst, src_len >= 3 ? s[2] : 0, (size_t)pitch * c->p.height);
  return 1;
}
}
