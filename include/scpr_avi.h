/* scpr_avi.h — AVI container reader / writer for 'SCPR' streams.
 *
 * The reference has no container code: the capturing application writes the AVI and hands the
 * codec one frame at a time through ICM (screenpressor.cpp:392-439 sets the chunk id to 'SCPR' and
 * reports AVIIF_KEYFRAME for key frames).  This is the step either side of the codec so that real
 * captures can be read and written: classic RIFF AVI 1.0 (one video stream, `idx1` index, files
 * below 4 GiB — no OpenDML extension).  The stream format is the BITMAPINFOHEADER the codec
 * negotiates (scpr_driver_compress_get_format): biCompression 'SCPR', and for 16-bit video the
 * three colour masks after the header.
 */
#ifndef SCPR_AVI_H
#define SCPR_AVI_H

#include <stdint.h>
#include "scpr_driver.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scpr_avi_info {
    scpr_format format;     /* stream format ('strf') */
    uint32_t handler;       /* fccHandler of the stream header */
    uint32_t rate, scale;   /* frames per second = rate / scale */
    uint32_t frames;
} scpr_avi_info;

typedef struct scpr_avi_writer scpr_avi_writer;
typedef struct scpr_avi_reader scpr_avi_reader;

/* Creates the file and writes provisional headers.  `fmt->compression` is the chunk type: 'SCPR'
 * (or any other fourcc) gives '00dc' chunks, BI_RGB / BI_BITFIELDS give '00db'.  NULL on failure. */
scpr_avi_writer* scpr_avi_create(const char* path, const scpr_format* fmt, uint32_t rate, uint32_t scale);
/* Appends one frame.  flags: SCPR_FRAME_KEY for key frames.  SCPR_E_CAPACITY when the file would pass 4 GiB. */
int scpr_avi_write(scpr_avi_writer* w, const void* data, uint32_t size, uint32_t flags);
/* Writes the index, patches the sizes and frame counts, closes the file and frees the writer. */
int scpr_avi_finish(scpr_avi_writer* w);

/* Opens a file and indexes the frames of its first video stream (from `idx1`, or by walking `movi`). */
scpr_avi_reader* scpr_avi_open(const char* path);
int scpr_avi_get_info(const scpr_avi_reader* r, scpr_avi_info* info);
/* Size of frame `index` (and its flags), or < 0. */
int64_t scpr_avi_frame_size(const scpr_avi_reader* r, uint32_t index, uint32_t* flags);
/* Reads frame `index` into buf; returns its size, or < 0 (SCPR_E_CAPACITY when buf is too small). */
int64_t scpr_avi_read(scpr_avi_reader* r, uint32_t index, void* buf, uint64_t capacity, uint32_t* flags);
void scpr_avi_close(scpr_avi_reader* r);

#ifdef __cplusplus
}
#endif
#endif /* SCPR_AVI_H */
