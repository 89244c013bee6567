"""Python face of include/scpr_driver.h and include/scpr_avi.h: the codec-instance policy of the
reference's CodecInst (screenpressor.cpp:276-650) and an AVI container for 'SCPR' streams.

Thin ctypes bindings; all logic is in libscpr_amd.so (csrc/scpr_driver.cpp, csrc/scpr_avi.cpp).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .codec import load_library

FOURCC_SCPR = 0x52504353
FOURCC_DIB = 0x20424944
BI_RGB = 0
BI_BITFIELDS = 3
FRAME_KEY = 0x10
E_BADFORMAT = -16


class Format(C.Structure):
    """scpr_format: the BITMAPINFOHEADER fields CodecInst uses"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bit_count", C.c_uint32), ("compression", C.c_uint32),
                ("size_image", C.c_uint32), ("masks", C.c_uint32 * 3)]

    @classmethod
    def make(cls, width, height, bit_count=32, compression=BI_RGB, masks=(0, 0, 0)):
        f = cls(width, height, bit_count, compression, ((width * bit_count // 8 + 3) & ~3) * height)
        f.masks[:] = masks
        return f


class DriverConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("key_frame_interval", "force_interval", "force_loss", "loss", "workers")]


class AviInfo(C.Structure):
    _fields_ = [("format", Format), ("handler", C.c_uint32), ("rate", C.c_uint32), ("scale", C.c_uint32), ("frames", C.c_uint32)]


_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, fp = C.c_void_p, C.POINTER(Format)
        L.scpr_driver_open.restype = vp
        L.scpr_driver_open.argtypes = [C.c_int]
        L.scpr_driver_close.argtypes = [vp]
        L.scpr_driver_configure.argtypes = [vp, C.POINTER(DriverConfig)]
        L.scpr_driver_compress_query.argtypes = [vp, fp]
        L.scpr_driver_compress_get_format.argtypes = [vp, fp, fp]
        L.scpr_driver_compress_get_size.restype = C.c_uint32
        L.scpr_driver_compress_get_size.argtypes = [fp]
        L.scpr_driver_compress_begin.argtypes = [vp, fp]
        L.scpr_driver_compress_end.argtypes = [vp]
        L.scpr_driver_compress.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.scpr_driver_decompress_query.argtypes = [vp, fp, fp]
        L.scpr_driver_decompress_get_format.argtypes = [vp, fp, fp]
        L.scpr_driver_decompress_begin.argtypes = [vp, fp, fp]
        L.scpr_driver_decompress_end.argtypes = [vp]
        L.scpr_driver_decompress.argtypes = [vp, vp, C.c_uint32, vp, C.c_int]
        L.scpr_infer_frame_type.argtypes = [C.c_uint8, C.c_uint32]
        L.scpr_avi_create.restype = vp
        L.scpr_avi_create.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32]
        L.scpr_avi_write.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
        L.scpr_avi_finish.argtypes = [vp]
        L.scpr_avi_open.restype = vp
        L.scpr_avi_open.argtypes = [C.c_char_p]
        L.scpr_avi_get_info.argtypes = [vp, C.POINTER(AviInfo)]
        L.scpr_avi_frame_size.restype = C.c_int64
        L.scpr_avi_frame_size.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32)]
        L.scpr_avi_read.restype = C.c_int64
        L.scpr_avi_read.argtypes = [vp, C.c_uint32, vp, C.c_uint64, C.POINTER(C.c_uint32)]
        L.scpr_avi_close.argtypes = [vp]
        _bound = True
    return L


def infer_frame_type(first_byte: int, data_size: int) -> int:
    """CodecInst::InferFrameType (screenpressor.cpp:579-589)"""
    return _lib().scpr_infer_frame_type(first_byte, data_size)


class Driver:
    """One codec instance as a VfW host sees it (CodecInst)."""

    def __init__(self, device: int = 0):
        self._L = _lib()
        self._h = self._L.scpr_driver_open(device)
        if not self._h:
            raise RuntimeError("scpr_driver_open failed: no usable gfx950 device (there is no CPU fallback)")
        self._in = None

    def close(self):
        if self._h:
            self._L.scpr_driver_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def configure(self, key_frame_interval=500, force_interval=False, force_loss=False, loss=0, workers=1):
        cfg = DriverConfig(key_frame_interval, int(force_interval), int(force_loss), loss, workers)
        self._L.scpr_driver_configure(self._h, C.byref(cfg))

    def compress_query(self, fmt: Format) -> int:
        return self._L.scpr_driver_compress_query(self._h, C.byref(fmt))

    def compress_get_format(self, fmt: Format) -> Format:
        out = Format()
        rc = self._L.scpr_driver_compress_get_format(self._h, C.byref(fmt), C.byref(out))
        if rc:
            raise ValueError(f"bad format ({rc})")
        return out

    def compress_begin(self, fmt: Format) -> int:
        self._in = fmt
        self._cap = self._L.scpr_driver_compress_get_size(C.byref(fmt))
        return self._L.scpr_driver_compress_begin(self._h, C.byref(fmt))

    def compress(self, frame: np.ndarray, quality: int = 10000, keyframe: bool = False):
        """-> (packet bytes, flags)"""
        src = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        out = np.empty(self._cap, dtype=np.uint8)
        sz, fl = C.c_uint32(), C.c_uint32()
        rc = self._L.scpr_driver_compress(self._h, src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), self._cap, quality, int(keyframe),
                                          C.byref(sz), C.byref(fl))
        if rc:
            raise RuntimeError(f"scpr_driver_compress: {rc}")
        return out[: sz.value].tobytes(), fl.value

    def compress_end(self):
        return self._L.scpr_driver_compress_end(self._h)

    def decompress_query(self, fin: Format, fout: Format | None) -> int:
        return self._L.scpr_driver_decompress_query(self._h, C.byref(fin), C.byref(fout) if fout is not None else None)

    def decompress_get_format(self, fin: Format) -> Format:
        out = Format()
        rc = self._L.scpr_driver_decompress_get_format(self._h, C.byref(fin), C.byref(out))
        if rc:
            raise ValueError(f"bad format ({rc})")
        return out

    def decompress_begin(self, fin: Format, fout: Format) -> int:
        self._dec_bytes = ((fin.width * fin.bit_count // 8 + 3) & ~3) * fin.height
        return self._L.scpr_driver_decompress_begin(self._h, C.byref(fin), C.byref(fout))

    def decompress(self, packet: bytes, not_keyframe: bool = False) -> np.ndarray:
        buf = np.frombuffer(packet, dtype=np.uint8)
        out = np.zeros(self._dec_bytes, dtype=np.uint8)
        rc = self._L.scpr_driver_decompress(self._h, buf.ctypes.data_as(C.c_void_p), len(packet), out.ctypes.data_as(C.c_void_p), int(not_keyframe))
        if rc:
            raise RuntimeError(f"scpr_driver_decompress: {rc}")
        return out

    def decompress_end(self):
        return self._L.scpr_driver_decompress_end(self._h)


class AviWriter:
    def __init__(self, path: str, fmt: Format, rate: int = 25, scale: int = 1):
        self._L = _lib()
        self._h = self._L.scpr_avi_create(path.encode(), C.byref(fmt), rate, scale)
        if not self._h:
            raise OSError(f"cannot create {path}")

    def write(self, data: bytes, flags: int = 0):
        buf = np.frombuffer(data, dtype=np.uint8)
        rc = self._L.scpr_avi_write(self._h, buf.ctypes.data_as(C.c_void_p), len(data), flags)
        if rc:
            raise OSError(f"scpr_avi_write: {rc}")

    def finish(self):
        if self._h:
            rc = self._L.scpr_avi_finish(self._h)
            self._h = None
            if rc:
                raise OSError(f"scpr_avi_finish: {rc}")


class AviReader:
    def __init__(self, path: str):
        self._L = _lib()
        self._h = self._L.scpr_avi_open(path.encode())
        if not self._h:
            raise OSError(f"cannot open {path} as AVI")
        self.info = AviInfo()
        self._L.scpr_avi_get_info(self._h, C.byref(self.info))

    def __len__(self):
        return self.info.frames

    def read(self, i: int):
        """-> (bytes, flags)"""
        fl = C.c_uint32()
        n = self._L.scpr_avi_frame_size(self._h, i, C.byref(fl))
        if n < 0:
            raise IndexError(i)
        buf = np.empty(max(int(n), 1), dtype=np.uint8)
        got = self._L.scpr_avi_read(self._h, i, buf.ctypes.data_as(C.c_void_p), int(n), C.byref(fl))
        if got != n:
            raise OSError(f"scpr_avi_read: {got}")
        return buf[:n].tobytes(), fl.value

    def close(self):
        if self._h:
            self._L.scpr_avi_close(self._h)
            self._h = None

    def __del__(self):
        self.close()
