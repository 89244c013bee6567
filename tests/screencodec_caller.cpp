// TEST: a caller written against the reference's ScreenCodec class (the way CodecInst drives it,
// screenpressor.cpp:343-439, :535-638) compiled against include/scpr_screencodec.hpp.  Encodes a key frame and
// two P-frames of a small synthetic picture, decodes them, and checks the round trip; prints the packet sizes and an
// FNV-1a hash of the packets so the Python side can compare them with the oracle's.
#include <stdio.h>
#include <string.h>
#include <vector>
#include "scpr_screencodec.hpp"

int main(int argc, char** argv) {
  const uint W = 100, H = 37;
  CodecParameters prm;
  prm.width = W, prm.height = H, prm.bits_per_pixel = 32;
  prm.redmask = 0x7C00, prm.greenmask = 0x3E0, prm.bluemask = 0x1F;
  prm.high_range_x = prm.high_range_y = 256, prm.low_range_x = prm.low_range_y = 8;  // screenpressor.cpp:377-378
  prm.loss = 0;
  ScreenCodec enc, dec;
  enc.Init(&prm);
  dec.Init(&prm);
  std::vector<BYTE> src(W * H * 4), dst(W * H * 6), out(W * H * 4);  // CompressGetSize = W*H*6 (screenpressor.cpp:386-388)
  unsigned long long fnv = 1469598103934665603ull;
  for (int t = 0; t < 3; t++) {
    for (uint y = 0; y < H; y++)
      for (uint x = 0; x < W; x++) {
        BYTE* p = &src[(y * W + x) * 4];
        const bool box = x >= 10 + 3u * t && x < 40 + 3u * t && y >= 5 && y < 20;  // a box that moves 3 pixels per frame
        p[0] = box ? (BYTE)(x * 7 + y) : 200;
        p[1] = box ? (BYTE)(y * 5) : 180;
        p[2] = box ? 30 : (BYTE)(160 + (y & 1));
        p[3] = 255;
      }
    int ftype = t == 0 ? 0 : 1;
    const int n = enc.CompressFrame(&src[0], &dst[0], (int)dst.size(), ftype, 0);
    if (n <= 0) return 2;
    for (int i = 0; i < n; i++) fnv = (fnv ^ dst[i]) * 1099511628211ull;
    if (dec.DecompressFrame(&dst[0], n, &out[0], W * 4, ftype) != 1) return 3;
    if (memcmp(&src[0], &out[0], src.size()) != 0) return 4;
    printf("frame %d ftype %d bytes %d\n", t, ftype, n);
  }
  // an unknown stream version is an exception, as in the reference (screencap.cpp:1609)
  BYTE bad[8] = {0x52, 0, 0, 0, 0, 0, 0, 0};  // version 6
  ScreenCodec d2;
  d2.Init(&prm);
  bool threw = false;
  try {
    d2.DecompressFrame(bad, 8, &out[0], W * 4, 0);
  } catch (BadVersionException& e) {
    threw = e.version == 6;
  }
  if (!threw) return 5;
  // a parameter set the library refuses (near window wider than the far one): Init does not throw (the reference's only stores
  // the parameters), the frame calls return 0 as for any refusal, and LastError() names the code
  CodecParameters odd = prm;
  odd.high_range_x = 4, odd.low_range_x = 8;
  ScreenCodec e3;
  e3.Init(&odd);
  int ft3 = 0;
  if (e3.CompressFrame(&src[0], &dst[0], (int)dst.size(), ft3, 0) != 0 || e3.LastError() != SCPR_E_PARAM) return 6;
  BYTE key4[8] = {0x32, 0, 0, 0, 0, 0, 0, 0};  // a version 4 key frame header
  if (e3.DecompressFrame(key4, 8, &out[0], W * 4, 0) != 0 || e3.LastError() != SCPR_E_PARAM) return 7;
  if (enc.LastError() != SCPR_OK) return 8;
  printf("fnv %llu\nok\n", fnv);
  return 0;
}
