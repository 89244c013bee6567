/* scpr_amd.h — C ABI of the MI355X-native ScreenPressor v4 encode/decode path.
 *
 * Drop-in boundary: the reference's `class ScreenCodec` (screencap.h:519-541 in
 * the reference tree), which CodecInst owns by value (screenpressor.h:15) and
 * drives from Compress/Decompress (screenpressor.cpp:392-439, :591-638).
 * Every entry point below names the member it replaces.  No C++ types, no
 * exceptions and no torch types cross this boundary.
 *
 * The library is GPU-only: there is no CPU fallback.  Every call fails with
 * SCPR_E_DEVICE when no gfx950 device is usable.
 */
#ifndef SCPR_AMD_H
#define SCPR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CodecParameters (screencap.h:49-55) + `workers`.
 * `workers` is the size of the reference's CSquad pool (= CPU count of the
 * capture machine, screencap.cpp:1459-1461).  It is bitstream-visible: key
 * frames cut their pixel runs at the start of each of `workers` row bands
 * (screencap.cpp:365-388, squad.cpp:16-31).  1 is the canonical value.
 * P-frames are always produced in the canonical (single worker) block order. */
typedef struct scpr_params {
    uint32_t width, height;
    uint32_t bits_per_pixel;              /* 16, 24 or 32 */
    uint32_t red_mask, green_mask, blue_mask; /* RGB16 only, e.g. 0x7C00,0x3E0,0x1F */
    uint32_t high_range_x, high_range_y;  /* motion search, far window (256,256) */
    uint32_t low_range_x, low_range_y;    /* near window (8,8); at most min(high_range, 256): beyond that the reference writes motion symbols below zero */
    uint32_t loss;                        /* 0..5 bits dropped per channel */
    uint32_t workers;                     /* >= 1 */
} scpr_params;

typedef struct scpr_codec scpr_codec;

enum {
    SCPR_OK = 0,
    SCPR_E_DEVICE = -1,      /* no usable gfx950 device / HIP error */
    SCPR_E_PARAM = -2,       /* bad argument */
    SCPR_E_BAD_VERSION = -3, /* BadVersionException (screencap.h:86-90): stream version not 2/3/4 (version 2 is decode-only) or bpp not 16/24/32 */
    SCPR_E_CAPACITY = -4,    /* destination too small */
    SCPR_E_STREAM = -5       /* corrupt stream detected by the decoder */
};

/* ScreenCodec::ScreenCodec() (screencap.cpp:1560).  `device` = HIP ordinal. */
scpr_codec* scpr_create(int device);
/* ScreenCodec::~ScreenCodec() (screencap.h:535) */
void scpr_destroy(scpr_codec* c);
/* ScreenCodec::Init (screencap.cpp:1565-1584) */
int scpr_init(scpr_codec* c, const scpr_params* p);
/* ScreenCodec::Deinit (screencap.cpp:1619-1629) */
void scpr_deinit(scpr_codec* c);
/* ScreenCodec::CrashHappened (screencap.h:540) */
void scpr_crash_happened(scpr_codec* c);

/* ScreenCodec::CompressFrame (screencap.cpp:1632-1692).  Host pointers.
 * src: RGB32 rows of width*4 bytes; RGB24 rows padded to 4 bytes; RGB16 rows of
 *      width*2 bytes back to back (the reference indexes them `y*X*2`,
 *      screencap.cpp:1668 - NOT DWORD-aligned, which only differs for odd widths;
 *      the decompress side writes RGB16 rows at the caller's `pitch`, :1726-1734).
 * *ftype in: 0 = key frame wanted, 1 = P allowed; out: type produced.
 * Returns the compressed size, 0 when refused (crashed), < 0 on error.
 * dst_len: room at dst (the reference's dstLength; its own guard, CheckDstLength,
 *      screencap.cpp:300-314, is commented out and CodecInst provides W*H*6 bytes,
 *      screenpressor.cpp:386-388).  A packet that does not fit is REFUSED with
 *      SCPR_E_CAPACITY and the codec is left exactly as the call found it (models,
 *      previous frame, motion-vector memory, flat-frame memory, frame count): the
 *      same frame may be given again with more room. */
int scpr_compress_frame(scpr_codec* c, const void* src, void* dst, int dst_len, int* ftype, int loss);

/* ScreenCodec::DecompressFrame (screencap.cpp:1695-1743).  Host pointers.
 * Returns 1 on success, 0 when refused, < 0 on error. */
int scpr_decompress_frame(scpr_codec* c, const void* src, int src_len, void* dst, int pitch, int ftype);

/* ---- batch entry points (an addition: many frames per call, data resident in
 * HBM).  Semantics are exactly those of calling the per-frame functions on the
 * frames in order, including all cross-frame state. -------------------------- */

/* d_frames: nframes frames back to back in device memory (same layout as src
 *           above).
 * ftypes:   host array, in/out as *ftype above.
 * d_out:    device buffer receiving the packets back to back in frame order.
 * sizes:    host array receiving each packet's size.
 * Returns the total number of bytes written, or < 0.
 * SCPR_E_CAPACITY (the packets do not fit out_capacity): the WHOLE call is taken
 * back - codec state and `ftypes` are as before it, d_out holds nothing of use.
 * How: a chunk's packets are bounded before anything is coded (2 bytes per coder
 * entry + 4 per block of 131072 + headers); only when that bound exceeds the room
 * left is the state the chains are about to change copied aside first (models,
 * live dense tables, previous frame: ~10-30 MB device to device).  A call of
 * several frames copies that state at its start unless out_capacity covers the
 * closed-form worst case, about 10 bytes per pixel and frame (a later chunk of the
 * call could not take back an earlier one's changes) - so a multi-frame call with
 * the reference's W*H*6 bytes per frame does pay for the copy (tens of
 * microseconds; more with a long live GOP's dense tables); a one-frame call pays
 * only when the exact bound says so.
 * Any OTHER error (< 0) from a compress call: if the call had kept that copy it is
 * taken back whole as well; if not, the codec is left as the reference leaves
 * itself after an exception (screencap.cpp:1634-1644): scpr_compress_* return 0
 * for every later frame until scpr_init() is called again. */
int64_t scpr_compress_batch(scpr_codec* c, const void* d_frames, int nframes, int* ftypes, int loss,
                            void* d_out, size_t out_capacity, uint32_t* sizes);

/* d_packets: the packets back to back in device memory (exactly sum(sizes)
 * bytes: the decoder touches no address past the aligned 4-byte word that holds
 * the last byte, so no slack is needed behind the buffer); sizes/ftypes: host
 * arrays.  d_frames_out: nframes frames of `pitch`-byte rows in device
 * memory.  Returns the number of frames decoded, or < 0. */
int scpr_decompress_batch(scpr_codec* c, const void* d_packets, const uint32_t* sizes, const int* ftypes,
                          int nframes, void* d_frames_out, int pitch);

/* ---- batch entry points with HOST pointers (an addition) --------------------
 * The reference's boundary hands over host memory (ScreenCodec::CompressFrame,
 * screencap.cpp:1632: pSrc / pDst; DecompressFrame, :1695).  These are the two
 * batch calls in that shape - same arguments, same results, same state rules,
 * h_* pointing to host memory - with the PCIe crossings taken BESIDE the kernels:
 *   compress:   the frames come over in sub-batches on a copy stream (sub-batch
 *               k + 1 while k is coded); the packets, ~1 % of the frames, are
 *               written by the gather kernel straight into h_out.
 *   decompress: the packets go over first; the chain of every coded key frame
 *               sends each finished row to h_frames_out itself (RGB32 output of
 *               version 3 / 4 streams), so the pictures cross while the chains
 *               run; P-frames, flat frames and the other pixel formats are
 *               unpacked into h_frames_out by kernels once their GOPs are done.
 * The overlap needs PINNED host memory: memory the HIP runtime has pinned
 * (hipHostMalloc, hipHostRegister, torch pin_memory) is recognised and used as it
 * is; a buffer the caller reuses - a capture loop's frame and packet buffers - is
 * pinned and mapped once with scpr_host_pin and released with scpr_host_unpin
 * (or by scpr_destroy).  Pageable memory (malloc, numpy) is never registered behind
 * the caller's back - a registration would outlive the memory it names - and goes
 * through the runtime's staging copies: same results, no overlap.
 * scpr_compress_batch_host is taken back whole on SCPR_E_CAPACITY, like
 * scpr_compress_batch. */
int64_t scpr_compress_batch_host(scpr_codec* c, const void* h_frames, int nframes, int* ftypes, int loss,
                                 void* h_out, size_t out_capacity, uint32_t* sizes);
int scpr_decompress_batch_host(scpr_codec* c, const void* h_packets, const uint32_t* sizes, const int* ftypes,
                               int nframes, void* h_frames_out, int pitch);
/* hipHostRegister(p, bytes, mapped) / hipHostUnregister(p) kept by the codec: the batch_host calls take their fast path for
 * pointers inside a pinned range.  The memory must stay allocated until scpr_host_unpin or scpr_destroy.  0 or < 0. */
int scpr_host_pin(scpr_codec* c, void* p, size_t bytes);
int scpr_host_unpin(scpr_codec* c, void* p);

/* ---- sharding support (an addition) ----------------------------------------
 * GOPs are independent except for what CScreenCapt keeps ACROSS key frames:
 * `fn > 0` (a frame has been coded: P-frames are allowed, screencap.cpp:1504)
 * and the flat-frame memory last_was_flat / last_flat_clr with `prev` holding
 * that flat picture (:1490-1497).  Call after scpr_init on the codec of a shard
 * that does not start the stream: frames_before = frames coded before the
 * shard's first frame, last_was_flat / last_flat_rgb (b0 | b1<<8 | b2<<16 of the
 * RGB24 pixel) = whether the frame just before the shard was a flat one.  The
 * third piece of state that crosses key frames, the motion-vector memory, has
 * its own calls below. */
int scpr_seed_shard(scpr_codec* c, uint32_t frames_before, int last_was_flat, uint32_t last_flat_rgb);

/* The motion-vector memory `int *mvs[2]` of CScreenCapt (screencap.h:450;
 * calloc'd by Init, screencap.cpp:96-97; written by FindMV :720-735, :740-811;
 * read as "the vector of the block above" by every later P-frame, :726-735;
 * never reset - RenewI :178-198 touches models only).  mx / my: host arrays of
 * ceil(width/16) * ceil(height/16) ints in block raster order, components in
 * [-256, 256].  Both return the number of blocks, or < 0.
 * A shard that does not start the stream imports what the frames before it
 * leave behind; with that (and scpr_seed_shard) its packets are the single
 * stream's packets. */
int scpr_export_mv_memory(scpr_codec* c, int32_t* mx, int32_t* my);
int scpr_import_mv_memory(scpr_codec* c, const int32_t* mx, const int32_t* my);

/* What mvs[] holds after scpr_compress_batch of these frames (same arguments),
 * WITHOUT coding them and without changing the codec: conversion to RGB24,
 * loss mask, frame-type decisions, block compare and motion search only
 * (DecideBlockTypes :928-1087 and FindMV :684-814 read the planes and mvs[],
 * nothing of the models).  A rank runs it over its own shard to produce the
 * memory the next shard starts from, before any rank has coded anything.
 * ftypes: host array (in only).  Returns the number of blocks, or < 0. */
int scpr_motion_prepass(scpr_codec* c, const void* d_frames, int nframes, const int* ftypes, int loss,
                        int32_t* mx, int32_t* my);

/* ---- instrumentation ------------------------------------------------------ */
/* Kernel time of the last batch call, measured with HIP events on the codec's
 * own stream: total milliseconds and, per stage, milliseconds in `stage_ms`
 * (up to `cap` entries, names via scpr_stage_name).  Returns the number of
 * stages. */
int scpr_last_timing(scpr_codec* c, float* total_ms, float* stage_ms, int cap);
const char* scpr_stage_name(int stage);

/* Debug taps on the last compress call (tests only): copies up to cap coder
 * entries ({freq, cum} uint16 pairs, stream order, all frames of the batch)
 * to host memory; returns the entry count. */
int64_t scpr_debug_entries(scpr_codec* c, uint16_t* out, int64_t cap);

/* Debug tap (tests only): bytes currently allocated for the dense-table arenas of
 * the compress side and of the decompress side. */
int scpr_debug_arena(scpr_codec* c, uint64_t* enc_bytes, uint64_t* dec_bytes);

/* Test hook, inert (SCPR_E_PARAM) unless the codec was created with
 * SCPR_ENABLE_DEBUG_INJECT=1 in the environment: the next scpr_compress_batch
 * misbehaves on purpose.  1: SCPR_E_DEVICE between the read-backs of its results and
 * their hand-over to the caller's variables (what a HIP error there leaves queued);
 * 2: two colour keys change places in front of the chains (the order proof must end
 * the call with SCPR_E_DEVICE, no chain is followed); 3: one trip's records of the
 * scalar-unit rANS coder are not laid, i.e. the scalar unit reads a lap-old line
 * (the coder's own check must notice, the call's blocks are coded again in the
 * vector form and the call returns the right bytes). */
int scpr_debug_inject(scpr_codec* c, int what);

/* Test tap: how many compress calls of this codec had their rANS blocks coded a
 * second time because the scalar-unit coder's check (see above) spoke. */
int scpr_debug_rans_recoded(scpr_codec* c);

/* Test hook: runs ONE colour context over `n` symbols through the wave-per-chain
 * encoder kernel and returns the coder entries ({freq, cum} pairs; freq 0 = raw). */
int scpr_debug_colour_chain(int device, const uint8_t* syms, int n, int f0, uint16_t* out);

const char* scpr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SCPR_AMD_H */
