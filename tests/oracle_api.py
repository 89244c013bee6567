"""ctypes binding of the ORACLE (oracle/libspo.so).  Test infrastructure only:
nothing under screenpressor_amd/ may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class SpoParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "bits_per_pixel", "red_mask", "green_mask", "blue_mask",
        "high_range_x", "high_range_y", "low_range_x", "low_range_y", "loss", "workers", "version")]


def make_params(w, h, bpp=32, loss=0, workers=1, version=4, masks=(0x7C00, 0x3E0, 0x1F), high_range=(256, 256), low_range=(8, 8)):
    return SpoParams(w, h, bpp, masks[0], masks[1], masks[2], high_range[0], high_range[1], low_range[0], low_range[1], loss, workers, version)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("SPO_LIB") or os.path.join(ORACLE_DIR, "libspo.so")  # (SPO_LIB: the sanitizer build, tests/test_sanitizers.py)
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("spo_codec.cpp", "spo_capi.cpp", "spo_codec.h", "spo_model.h")]
        if not os.environ.get("SPO_LIB") and (not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libspo.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.spo_create.restype = C.c_void_p
        L.spo_create.argtypes = [C.POINTER(SpoParams)]
        L.spo_destroy.argtypes = [C.c_void_p]
        L.spo_compress_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.spo_decompress_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.spo_crash_happened.argtypes = [C.c_void_p]
        L.spo_seed_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
        L.spo_export_mv_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.spo_import_mv_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.spo_tap_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.spo_tap_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.spo_tap_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.spo_tap_prev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.spo_tap_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.spo_chain_colour.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.spo_chain_colour_dec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.spo_chain_fixed.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.spo_rans_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.spo_time_stream.argtypes = [C.POINTER(SpoParams), C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.spo_time_stream2.argtypes = [C.POINTER(SpoParams), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
        L.spo_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.spo_fnv1a.restype = C.c_uint64
        L.spo_fnv1a.argtypes = [C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleCodec:
    """Mirror of ScreenCodec (screencap.h:519-541) over the CPU restatement."""

    def __init__(self, w, h, bpp=32, loss=0, workers=1, version=4, high_range=(256, 256), low_range=(8, 8)):
        self.w, self.h, self.bpp, self.loss = w, h, bpp, loss
        self.params = make_params(w, h, bpp, loss, workers, version, high_range=high_range, low_range=low_range)
        self.h_ = lib().spo_create(C.byref(self.params))
        self.cap = w * h * 6 + 64
        self.pitch = w * 4 if bpp == 32 else ((w * (bpp // 8) + 3) & ~3)
        self.in_pitch = w * 2 if bpp == 16 else self.pitch  # RGB16 input rows back to back (screencap.cpp:1668)
        self.stride24 = (w * 3 + 3) & ~3
        self.nblocks = ((w + 15) // 16) * ((h + 15) // 16)

    def close(self):
        if self.h_ and _lib is not None:  # (at interpreter shutdown the module globals may be gone already)
            _lib.spo_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def compress(self, frame, key=False, loss=None):
        src = np.ascontiguousarray(frame, dtype=np.uint8).copy()  # the codec may write into src
        assert src.size == self.in_pitch * self.h, (src.size, self.in_pitch, self.h)
        dst = np.empty(self.cap, dtype=np.uint8)
        ft = C.c_int(0 if key else 1)
        n = lib().spo_compress_frame(self.h_, _ptr(src), _ptr(dst), self.cap, C.byref(ft), self.loss if loss is None else loss)
        return bytes(dst[:n]), ft.value

    def set_threads(self, n):
        lib().spo_set_threads(self.h_, n)

    def seed_shard(self, frames_before, last_flat, rgb):
        lib().spo_seed_shard(self.h_, frames_before, 1 if last_flat else 0, rgb)

    def export_mv_memory(self):
        mv = np.zeros((2, self.nblocks), dtype=np.int32)
        assert lib().spo_export_mv_memory(self.h_, _ptr(mv[0]), _ptr(mv[1]), self.nblocks) == self.nblocks
        return mv

    def import_mv_memory(self, mv):
        mv = np.ascontiguousarray(mv, dtype=np.int32).reshape(2, self.nblocks)
        assert lib().spo_import_mv_memory(self.h_, _ptr(mv[0]), _ptr(mv[1]), self.nblocks) == self.nblocks

    def decompress(self, data, ftype, pitch=None):
        pitch = self.pitch if pitch is None else pitch
        out = np.zeros(pitch * self.h, dtype=np.uint8)
        buf = np.frombuffer(bytes(data) + b"\0" * 16, dtype=np.uint8)  # 4-byte over-read guard
        r = lib().spo_decompress_frame(self.h_, _ptr(buf), len(data), _ptr(out), pitch, ftype)
        return r, out

    def entries(self):
        n = lib().spo_tap_entries(self.h_, None, 0)
        out = np.zeros((max(n, 0), 2), dtype=np.uint16)
        if n > 0:
            lib().spo_tap_entries(self.h_, _ptr(out), n)
        return out

    def tags(self):
        n = lib().spo_tap_tags(self.h_, None, 0)
        out = np.zeros(max(n, 0), dtype=np.uint16)
        if n > 0:
            lib().spo_tap_tags(self.h_, _ptr(out), n)
        return out

    def blocks(self):
        t = np.zeros(self.nblocks, dtype=np.uint8)
        r = np.zeros((4, self.nblocks), dtype=np.int32)
        m = np.zeros((2, self.nblocks), dtype=np.int32)
        lib().spo_tap_blocks(self.h_, _ptr(t), _ptr(r), _ptr(m), self.nblocks)
        return t, r, m

    def prev(self):
        out = np.zeros(self.stride24 * self.h, dtype=np.uint8)
        lib().spo_tap_prev(self.h_, _ptr(out), out.size)
        return out.reshape(self.h, self.stride24)


def chain_colour(syms, f0=32):
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    out = np.zeros((len(syms), 2), dtype=np.uint16)
    lib().spo_chain_colour(_ptr(syms), len(syms), f0, _ptr(out))
    return out


def chain_colour_dec(ivl, syms, f0=32, probe=0):
    ivl = np.ascontiguousarray(ivl, dtype=np.uint16)
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    return lib().spo_chain_colour_dec(_ptr(ivl), _ptr(syms), len(syms), f0, probe)


def chain_fixed(nsym, syms):
    syms = np.ascontiguousarray(syms, dtype=np.uint16)
    out = np.zeros((len(syms), 2), dtype=np.uint16)
    lib().spo_chain_fixed(nsym, _ptr(syms), len(syms), _ptr(out))
    return out


def rans_block(entries):
    entries = np.ascontiguousarray(entries, dtype=np.uint16)
    n = len(entries)
    out = np.zeros(2 * n + 16, dtype=np.uint8)
    sz = lib().spo_rans_block(_ptr(entries), n, _ptr(out))
    return bytes(out[:sz])


def time_stream(frames, w, h, bpp, key_interval, loss=0, workers=1, threads=1):
    """encode + decode `frames` on the host CPU.  workers = row bands of a key frame (bitstream-visible, as in the
    reference); threads = 1: one thread; > 1: the reference's two-stage shape (band pool + one coder thread)."""
    p = make_params(w, h, bpp, loss, workers)
    te, td, nb, hv = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
    frames = np.ascontiguousarray(frames, dtype=np.uint8).copy()
    n = frames.shape[0]
    sizes, fnvs = np.zeros(n, np.uint32), np.zeros(n, np.uint64)
    bad = lib().spo_time_stream2(C.byref(p), _ptr(frames), n, key_interval, threads, C.byref(te), C.byref(td), C.byref(nb), C.byref(hv), _ptr(sizes), _ptr(fnvs))
    return dict(bad=bad, t_enc=te.value, t_dec=td.value, bytes=nb.value, fnv=hv.value, sizes=sizes, frame_fnv=fnvs)


def fnv1a(data) -> int:
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().spo_fnv1a(_ptr(a), a.size))
