// Test infrastructure (CPU sanitizer build only, tests/test_sanitizers.py): drives csrc/scpr_avi.cpp and the policy layer of
// csrc/scpr_driver.cpp (over tests/asan_fake_codec.cpp) from a plain executable built with -fsanitize=address,undefined.
//   harness write <path>      a small valid AVI through scpr_avi_create / write / finish
//   harness read <path>...    opens each file like an untrusted capture: refused, or every frame sized and read
//   harness policy            a capture-shaped session through scpr_driver_*; prints the decisions, one line per frame
// Exit code 0 unless a call misbehaves; a sanitizer report aborts the process.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../include/scpr_avi.h"
#include "../include/scpr_driver.h"

static int do_write(const char* path) {
  scpr_format fmt{64, 48, 32, SCPR_FOURCC_SCPR, 64 * 48 * 4, {0, 0, 0}};
  scpr_avi_writer* w = scpr_avi_create(path, &fmt, 25, 1);
  if (!w) return 2;
  for (uint32_t i = 0; i < 7; i++) {
    std::vector<uint8_t> d(5 + 37 * i, (uint8_t)(0x30 + i));  // odd and even sizes (chunks are word aligned)
    if (i == 5) d.clear();                                      // a frame of no bytes (a dropped frame)
    if (scpr_avi_write(w, d.data(), (uint32_t)d.size(), i % 3 == 0 ? SCPR_FRAME_KEY : 0) != SCPR_OK) return 3;
  }
  return scpr_avi_finish(w) == SCPR_OK ? 0 : 4;
}

static int do_read(const char* path) {
  scpr_avi_reader* r = scpr_avi_open(path);
  if (!r) {
    std::printf("%s: refused\n", path);
    return 0;
  }
  scpr_avi_info info{};
  if (scpr_avi_get_info(r, &info) != SCPR_OK) return 5;
  uint64_t bytes = 0;
  uint32_t read_ok = 0, read_bad = 0;
  std::vector<uint8_t> buf;
  for (uint32_t i = 0; i < info.frames; i++) {
    uint32_t flags = 0;
    const int64_t sz = scpr_avi_frame_size(r, i, &flags);
    if (sz < 0) return 6;
    if (sz > (64 << 20)) {  // (a size field of a damaged file: the reader must refuse a buffer that is too small, not read past it)
      uint8_t tiny[16];
      if (scpr_avi_read(r, i, tiny, sizeof tiny, &flags) != SCPR_E_CAPACITY) return 7;
      read_bad++;
      continue;
    }
    buf.assign((size_t)sz + 1, 0xEE);
    const int64_t got = scpr_avi_read(r, i, buf.data(), (uint64_t)sz, &flags);
    if (got >= 0 && got != sz) return 8;
    if (buf[(size_t)sz] != 0xEE) return 9;  // wrote past the size it reported
    if (got >= 0) read_ok++, bytes += (uint64_t)got;
    else read_bad++;
    if (sz > 0 && scpr_avi_read(r, i, buf.data(), (uint64_t)sz - 1, &flags) != SCPR_E_CAPACITY) return 10;
  }
  if (scpr_avi_frame_size(r, info.frames, nullptr) >= 0) return 11;
  std::printf("%s: %ux%u bpp %u fourcc %08x rate %u/%u frames %u (read %u, short %u, %llu bytes)\n", path, info.format.width, info.format.height, info.format.bit_count,
              info.format.compression, info.rate, info.scale, info.frames, read_ok, read_bad, (unsigned long long)bytes);
  scpr_avi_close(r);
  return 0;
}

static int do_policy() {
  scpr_driver* d = scpr_driver_open(0);
  if (!d) return 20;
  if (scpr_driver_open(7) != nullptr) return 21;  // (the fake has one device)
  scpr_driver_config cfg{4, 0, 0, 0, 1};  // key frame every 4 frames unless the host asks earlier; quality decides the loss
  scpr_driver_configure(d, &cfg);
  const scpr_format good{64, 48, 32, SCPR_BI_RGB, 64 * 48 * 4, {0, 0, 0}};
  scpr_format bad = good;
  bad.bit_count = 8;
  scpr_format huge = good;
  huge.width = 0xFFFFFFFFu, huge.height = 0xFFFFFFFFu;
  scpr_format rgb16{64, 48, 16, SCPR_BI_BITFIELDS, 64 * 48 * 2, {0xF800, 0x07E0, 0x001F}};
  std::printf("query good %d bad %d huge %d rgb16 %d null %d\n", scpr_driver_compress_query(d, &good), scpr_driver_compress_query(d, &bad), scpr_driver_compress_query(d, &huge),
              scpr_driver_compress_query(d, &rgb16), scpr_driver_compress_query(d, nullptr));
  scpr_format out{};
  if (scpr_driver_compress_get_format(d, &rgb16, &out) != SCPR_OK) return 22;
  std::printf("get_format rgb16 -> fourcc %08x masks %x %x %x; size %u / huge %u\n", out.compression, out.masks[0], out.masks[1], out.masks[2], scpr_driver_compress_get_size(&good),
              scpr_driver_compress_get_size(&huge));
  if (scpr_driver_compress_get_format(d, &good, &out) != SCPR_OK || scpr_driver_compress_begin(d, &good) != SCPR_OK) return 23;
  std::vector<uint8_t> frame(64 * 48 * 4, 7), pkt(scpr_driver_compress_get_size(&good));
  std::vector<std::vector<uint8_t>> packets;
  const uint32_t quality[] = {10000, 10000, 7500, 5000, 2500, 0, 10000, 9999, 10000, 10000};
  for (int i = 0; i < 10; i++) {
    uint32_t size = 0, flags = 0;
    const int rc = scpr_driver_compress(d, frame.data(), pkt.data(), (uint32_t)pkt.size(), quality[i], i == 6, &size, &flags);
    if (rc != SCPR_OK || size != 6) return 24;
    std::printf("frame %d quality %u host_key %d -> key %d loss %u\n", i, quality[i], i == 6, (flags & SCPR_FRAME_KEY) ? 1 : 0, (unsigned)pkt[1]);
    packets.emplace_back(pkt.begin(), pkt.begin() + size);
  }
  uint32_t size = 0, flags = 0;
  if (scpr_driver_compress(d, frame.data(), pkt.data(), 2, 10000, 0, &size, &flags) == SCPR_OK) return 25;  // (a buffer of two bytes)
  scpr_driver_compress_end(d);
  if (scpr_driver_compress(d, frame.data(), pkt.data(), (uint32_t)pkt.size(), 10000, 0, &size, &flags) == SCPR_OK) return 26;  // (no session)
  // the registry's other mode: the interval rules (force_interval), the configured loss rules (force_loss)
  scpr_driver_config forced{3, 1, 1, 2, 1};
  scpr_driver_configure(d, &forced);
  if (scpr_driver_compress_begin(d, &good) != SCPR_OK) return 30;
  for (int i = 0; i < 8; i++) {
    if (scpr_driver_compress(d, frame.data(), pkt.data(), (uint32_t)pkt.size(), 10000, i == 1, &size, &flags) != SCPR_OK) return 31;
    std::printf("forced frame %d host_key %d -> key %d loss %u\n", i, i == 1, (flags & SCPR_FRAME_KEY) ? 1 : 0, (unsigned)pkt[1]);
  }
  scpr_driver_compress_end(d);
  scpr_driver_configure(d, nullptr);
  // the other direction
  scpr_format natural{};
  if (scpr_driver_decompress_get_format(d, &out, &natural) != SCPR_OK) return 27;
  std::printf("decompress query natural %d wrong-size %d; natural bpp %u fourcc %08x\n", scpr_driver_decompress_query(d, &out, &natural), scpr_driver_decompress_query(d, &out, &rgb16) == SCPR_OK ? 0 : 1,
              natural.bit_count, natural.compression);
  if (scpr_driver_decompress_begin(d, &out, &natural) != SCPR_OK) return 28;
  std::vector<uint8_t> pic(64 * 48 * 4 + 1, 0xEE);
  for (size_t i = 0; i < packets.size(); i++) {
    const int rc = scpr_driver_decompress(d, packets[i].data(), (uint32_t)packets[i].size(), pic.data(), i != 0);
    if (rc != SCPR_OK || pic[0] != (uint8_t)i || pic.back() != 0xEE) return 29;
  }
  scpr_driver_decompress_end(d);
  std::printf("infer %d %d %d %d %d\n", scpr_infer_frame_type(0x00, 100), scpr_infer_frame_type(0x01, 4), scpr_infer_frame_type(0x12, 9), scpr_infer_frame_type(0x32, 9), scpr_infer_frame_type(0xFF, 0));
  scpr_driver_close(d);
  scpr_driver_close(nullptr);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "write")) return do_write(argv[2]);
  if (argc >= 3 && !std::strcmp(argv[1], "read")) {
    for (int i = 2; i < argc; i++) {
      const int rc = do_read(argv[i]);
      if (rc) {
        std::fprintf(stderr, "%s: the reader misbehaved (%d)\n", argv[i], rc);
        return rc;
      }
    }
    return 0;
  }
  if (argc >= 2 && !std::strcmp(argv[1], "policy")) return do_policy();
  std::fprintf(stderr, "usage: harness write <path> | read <path>... | policy\n");
  return 64;
}
