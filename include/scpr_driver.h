/* scpr_driver.h — the codec-instance policy layer in front of the codec.
 *
 * The reference wraps `ScreenCodec` in `CodecInst` (screenpressor.h:9-60, screenpressor.cpp),
 * which a Video-for-Windows host drives through ICM messages.  This is that layer as a portable
 * C API: the same decisions (format negotiation, key-frame policy, quality -> loss, frame-type
 * inference on decode, buffer sizing), the same order of calls, no Win32 types.  Every function
 * names the CodecInst member it stands for.  Frames cross this API in host memory, one at a
 * time, exactly as they cross ICM; the work is done by the GPU codec of scpr_amd.h.
 */
#ifndef SCPR_DRIVER_H
#define SCPR_DRIVER_H

#include <stdint.h>
#include "scpr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SCPR_FOURCC_SCPR 0x52504353u /* 'SCPR' (screenpressor.h:6) */
#define SCPR_FOURCC_DIB 0x20424944u  /* 'DIB ' */
#define SCPR_BI_RGB 0u
#define SCPR_BI_BITFIELDS 3u
#define SCPR_FRAME_KEY 0x10u         /* AVIIF_KEYFRAME, what Compress reports for a key frame (screenpressor.cpp:427) */

/* the fields of BITMAPINFOHEADER (+ the three colour masks that follow it for 16-bit formats)
 * that CodecInst reads or writes */
typedef struct scpr_format {
    uint32_t width, height;
    uint32_t bit_count;     /* 16, 24, 32 */
    uint32_t compression;   /* SCPR_BI_RGB, SCPR_BI_BITFIELDS, SCPR_FOURCC_DIB or SCPR_FOURCC_SCPR */
    uint32_t size_image;
    uint32_t masks[3];      /* red, green, blue; 16-bit formats only */
} scpr_format;

/* what CodecInst takes from the registry (conf.h / Configuration::GetCurConfig) */
typedef struct scpr_driver_config {
    uint32_t key_frame_interval; /* KeyFrameInterval, default 500 */
    uint32_t force_interval;     /* 1: ignore the host's key-frame flag, force one every interval */
    uint32_t force_loss;         /* 1: use `loss` below, ignore the host's quality */
    uint32_t loss;               /* 0..4 */
    uint32_t workers;            /* see scpr_params.workers; 0 = 1 */
} scpr_driver_config;

enum { SCPR_E_BADFORMAT = -16 /* ICERR_BADFORMAT */ };

typedef struct scpr_driver scpr_driver;

/* DriverProc DRV_OPEN / DRV_CLOSE (drvproc.cpp): one codec instance on HIP device `device` */
scpr_driver* scpr_driver_open(int device);
void scpr_driver_close(scpr_driver* d);
/* registry values; NULL restores the defaults (interval 500, host-driven key frames, quality-driven loss) */
void scpr_driver_configure(scpr_driver* d, const scpr_driver_config* cfg);

/* CodecInst::CompressQuery (screenpressor.cpp:308-313): SCPR_OK or SCPR_E_BADFORMAT */
int scpr_driver_compress_query(scpr_driver* d, const scpr_format* in);
/* CodecInst::CompressGetFormat (:316-337): the stream format for an input format */
int scpr_driver_compress_get_format(scpr_driver* d, const scpr_format* in, scpr_format* out);
/* CodecInst::CompressGetSize (:386-388): bytes the host must provide per output frame */
uint32_t scpr_driver_compress_get_size(const scpr_format* in);
/* CodecInst::CompressBegin (:343-384) / CompressEnd (:442-446) */
int scpr_driver_compress_begin(scpr_driver* d, const scpr_format* in);
int scpr_driver_compress_end(scpr_driver* d);
/* CodecInst::Compress (:392-439).  quality 0..10000, host_keyframe = ICCOMPRESS_KEYFRAME.
 * *out_size = bytes written, *out_flags = SCPR_FRAME_KEY or 0. */
int scpr_driver_compress(scpr_driver* d, const void* in, void* out, uint32_t out_capacity, uint32_t quality, int host_keyframe,
                         uint32_t* out_size, uint32_t* out_flags);

/* CodecInst::DecompressQuery / CanDecompress (:449-500): stream format -> wanted output format */
int scpr_driver_decompress_query(scpr_driver* d, const scpr_format* in, const scpr_format* out);
/* CodecInst::DecompressGetFormat (:503-533): the natural output format of a stream */
int scpr_driver_decompress_get_format(scpr_driver* d, const scpr_format* in, scpr_format* out);
/* CodecInst::DecompressBegin (:535-577) / DecompressEnd (:644-650) */
int scpr_driver_decompress_begin(scpr_driver* d, const scpr_format* in, const scpr_format* out);
int scpr_driver_decompress_end(scpr_driver* d);
/* CodecInst::Decompress (:591-638).  not_keyframe = ICDECOMPRESS_NOTKEYFRAME; the first byte of the
 * data overrides it where InferFrameType can tell.  Output rows are (width*bit_count/8+3)&~3 bytes. */
int scpr_driver_decompress(scpr_driver* d, const void* in, uint32_t in_size, void* out, int not_keyframe);

/* CodecInst::InferFrameType (:579-589): 0 = key frame, 1 = P-frame, -1 = cannot tell (v3/v4 headers) */
int scpr_infer_frame_type(uint8_t first_byte, uint32_t data_size);

#ifdef __cplusplus
}
#endif
#endif /* SCPR_DRIVER_H */
