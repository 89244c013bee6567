// scpr_screencodec.hpp - the reference's `class ScreenCodec` (screencap.h:519-541) as a thin C++
// wrapper over the C ABI of include/scpr_amd.h, so that code written against the reference's class
// (CodecInst::CompressBegin/Compress/DecompressBegin/Decompress, screenpressor.cpp:343-638) compiles
// unchanged against the MI355X library: same type names, same member signatures, same error
// behaviour (0 when refused, BadVersionException for a stream version or pixel depth the codec
// does not know, screencap.h:86-90, thrown where the reference throws it: screencap.cpp:1589-1590,
// :1609).  Init never throws, as in the reference (it only stores the parameters, :1565-1584): a
// failure there (no GPU, a parameter this library refuses - INTEGRATION.md "range limits") shows as 0
// from the first CompressFrame / DecompressFrame, and LastError() (an addition) says which SCPR_E_*
// code it was.  Include this INSTEAD of the reference's screencap.h and link libscpr_amd.so.
#ifndef SCPR_SCREENCODEC_HPP
#define SCPR_SCREENCODEC_HPP

#include "scpr_amd.h"

#ifndef SCPR_NO_WIN_TYPES  // the reference takes these from <windows.h> / defines.h:14-15
typedef unsigned char BYTE;
typedef unsigned short WORD;
typedef unsigned int uint;
#endif

struct CodecParameters {  // screencap.h:49-55
  uint width, height;
  BYTE bits_per_pixel;  // 16, 24 or 32
  WORD redmask, greenmask, bluemask;
  uint high_range_x, high_range_y, low_range_x, low_range_y;
  uint loss;  // bits, 0..5
};

class BadVersionException {  // screencap.h:86-90
 public:
  BadVersionException(int v) : version(v) {}
  int version;
};

class ScreenCodec {
  scpr_codec* h;
  ScreenCodec(const ScreenCodec&);             // one GPU codec per object, like the reference's owned pSC
  ScreenCodec& operator=(const ScreenCodec&);

 public:
  // `device`: HIP ordinal.  `workers`: the size of the worker pool whose key-frame bitstream is to be
  // reproduced (the CPU build takes the machine's CPU count, screencap.cpp:1459-1461); 1 is canonical.
  explicit ScreenCodec(int device = 0, unsigned workers = 1) : h(scpr_create(device)), workers_(workers) {}
  ~ScreenCodec() {
    Deinit();
    scpr_destroy(h);
  }
  void Init(CodecParameters* p) {  // screencap.cpp:1565-1584
    scpr_params q = {p->width,        p->height,       p->bits_per_pixel, p->redmask,     p->greenmask, p->bluemask,
                     p->high_range_x, p->high_range_y, p->low_range_x,    p->low_range_y, p->loss,      workers_};
    bpp_ = p->bits_per_pixel;
    init_error_ = h ? scpr_init(h, &q) : (int)SCPR_E_DEVICE;  // kept for the first frame call: the reference's Init cannot fail
    last_error_ = init_error_;
  }
  void Deinit() {  // :1619-1629
    if (h) scpr_deinit(h);
  }
  // frame type 0 = I, 1 = P; returns the compressed size, 0 when nothing was written (:1632-1692)
  int CompressFrame(BYTE* pSrc, BYTE* pDst, int dstLength, int& ftype, int loss) {
    const int r = init_error_ != SCPR_OK ? init_error_ : scpr_compress_frame(h, pSrc, pDst, dstLength, &ftype, loss);
    last_error_ = r < 0 ? r : (int)SCPR_OK;
    if (r == SCPR_E_BAD_VERSION) throw BadVersionException(bpp_);  // CreateCodec: bits per pixel not 16/24/32 (:1589-1590)
    return r < 0 ? 0 : r;
  }
  // returns 1, or 0 when refused (:1695-1743)
  int DecompressFrame(BYTE* pSrc, int srcLength, BYTE* pDst, int pitch, int ftype) {
    const int r = init_error_ != SCPR_OK ? init_error_ : scpr_decompress_frame(h, pSrc, srcLength, pDst, pitch, ftype);
    last_error_ = r < 0 ? r : (int)SCPR_OK;
    if (r == SCPR_E_BAD_VERSION) throw BadVersionException((pSrc[0] >> 4) + 1);  // caught at screenpressor.cpp:621-636
    return r < 0 ? 0 : r;
  }
  void CrashHappened() { scpr_crash_happened(h); }  // screencap.h:540
  // (an addition) the SCPR_E_* code behind the last 0 a frame call returned, SCPR_OK after a call that worked
  int LastError() const { return last_error_; }

 private:
  unsigned workers_;
  int bpp_ = 0, init_error_ = SCPR_E_PARAM, last_error_ = SCPR_OK;  // (a frame call before any Init is refused)
};

#endif  // SCPR_SCREENCODEC_HPP
