"""Host-side mirror of the reference's ScreenCodec (screencap.h:519-541) over the
C ABI in include/scpr_amd.h.  Same method names, argument meaning and return
conventions; device memory is held in torch tensors (plumbing only).

There is no CPU implementation behind this class: if the HIP library is missing
or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("SCPR_AMD_LIB", os.path.join(_PKG, "libscpr_amd.so"))  # the override is for build experiments

SCPR_OK, SCPR_E_DEVICE, SCPR_E_PARAM, SCPR_E_BAD_VERSION, SCPR_E_CAPACITY, SCPR_E_STREAM = 0, -1, -2, -3, -4, -5
EXPORTS = ["scpr_create", "scpr_destroy", "scpr_init", "scpr_deinit", "scpr_crash_happened", "scpr_compress_frame",
           "scpr_decompress_frame", "scpr_compress_batch", "scpr_decompress_batch", "scpr_compress_batch_host", "scpr_decompress_batch_host", "scpr_host_pin", "scpr_host_unpin", "scpr_last_timing", "scpr_stage_name",
           "scpr_seed_shard", "scpr_export_mv_memory", "scpr_import_mv_memory", "scpr_motion_prepass", "scpr_debug_entries", "scpr_debug_arena", "scpr_debug_colour_chain", "scpr_debug_inject", "scpr_debug_rans_recoded", "scpr_version",
           # include/scpr_driver.h, include/scpr_avi.h
           "scpr_driver_open", "scpr_driver_close", "scpr_driver_configure", "scpr_driver_compress_query", "scpr_driver_compress_get_format",
           "scpr_driver_compress_get_size", "scpr_driver_compress_begin", "scpr_driver_compress_end", "scpr_driver_compress",
           "scpr_driver_decompress_query", "scpr_driver_decompress_get_format", "scpr_driver_decompress_begin", "scpr_driver_decompress_end",
           "scpr_driver_decompress", "scpr_infer_frame_type", "scpr_avi_create", "scpr_avi_write", "scpr_avi_finish", "scpr_avi_open",
           "scpr_avi_get_info", "scpr_avi_frame_size", "scpr_avi_read", "scpr_avi_close"]


class ScprParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "bits_per_pixel", "red_mask", "green_mask", "blue_mask",
        "high_range_x", "high_range_y", "low_range_x", "low_range_y", "loss", "workers")]


class BadVersionException(Exception):
    """screencap.h:86-90"""


class CapacityError(RuntimeError):
    """SCPR_E_CAPACITY: the packets do not fit the destination; the codec is as the call found it"""


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build it with `python -m screenpressor_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch ships its own libamdhip64.so.7; it must be the one HIP runtime of the process, so it is
        # loaded first and the library's DT_NEEDED entry binds to it (two runtimes cannot share the GPU).
        import torch  # noqa: F401
        L = C.CDLL(_LIB_PATH)
        L.scpr_create.restype = C.c_void_p
        L.scpr_create.argtypes = [C.c_int]
        L.scpr_destroy.argtypes = [C.c_void_p]
        L.scpr_init.argtypes = [C.c_void_p, C.POINTER(ScprParams)]
        L.scpr_deinit.argtypes = [C.c_void_p]
        L.scpr_crash_happened.argtypes = [C.c_void_p]
        L.scpr_compress_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.scpr_decompress_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.scpr_compress_batch.restype = C.c_int64
        L.scpr_compress_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.scpr_decompress_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_int]
        L.scpr_compress_batch_host.restype = C.c_int64
        L.scpr_compress_batch_host.argtypes = L.scpr_compress_batch.argtypes
        L.scpr_decompress_batch_host.argtypes = L.scpr_decompress_batch.argtypes
        L.scpr_host_pin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.scpr_host_unpin.argtypes = [C.c_void_p, C.c_void_p]
        L.scpr_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.scpr_stage_name.restype = C.c_char_p
        L.scpr_stage_name.argtypes = [C.c_int]
        L.scpr_debug_entries.restype = C.c_int64
        L.scpr_debug_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.scpr_seed_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
        L.scpr_export_mv_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.scpr_import_mv_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.scpr_motion_prepass.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]
        L.scpr_debug_arena.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.scpr_debug_colour_chain.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.scpr_debug_inject.argtypes = [C.c_void_p, C.c_int]
        L.scpr_debug_rans_recoded.argtypes = [C.c_void_p]
        L.scpr_version.restype = C.c_char_p
        _lib = L
    return _lib


class ScreenCodec:
    def __init__(self, device: int = 0):
        self._L = load_library()
        self._h = self._L.scpr_create(device)
        if not self._h:
            raise RuntimeError("scpr_create failed: no usable MI355X/HIP device (this codec has no CPU path)")
        self.device = device
        self.params = None

    # ScreenCodec::Init
    def Init(self, width, height, bits_per_pixel=32, loss=0, workers=1, masks=(0x7C00, 0x3E0, 0x1F),
             high_range=(256, 256), low_range=(8, 8)):
        self.params = ScprParams(width, height, bits_per_pixel, masks[0], masks[1], masks[2],
                                 high_range[0], high_range[1], low_range[0], low_range[1], loss, workers)
        rc = self._L.scpr_init(self._h, C.byref(self.params))
        if rc != SCPR_OK:
            raise ValueError(f"scpr_init failed: {rc}")
        self.width, self.height, self.bpp, self.loss = width, height, bits_per_pixel, loss
        self.pitch = width * 4 if bits_per_pixel == 32 else ((width * (bits_per_pixel // 8) + 3) & ~3)  # decode side: DIB rows
        # compress side: RGB16 rows are read back to back (screencap.cpp:1668), which differs from the DIB pitch for odd widths
        self.in_pitch = width * 2 if bits_per_pixel == 16 else self.pitch
        self.frame_bytes = self.in_pitch * height
        self.max_packet = width * height * 6 + 64  # CompressGetSize, screenpressor.cpp:386-388
        return self

    def Deinit(self):
        self._L.scpr_deinit(self._h)

    def SeedShard(self, frames_before: int, last_was_flat: bool, last_flat_rgb: int = 0):
        """scpr_seed_shard: the cross-GOP state of the single stream where this codec's shard starts"""
        self._check(self._L.scpr_seed_shard(self._h, frames_before, 1 if last_was_flat else 0, last_flat_rgb))
        return self

    # the motion-vector memory mvs[] (screencap.cpp:96-97), the other state that crosses key frames: (2, blocks) int32
    @property
    def nblocks(self):
        return ((self.width + 15) // 16) * ((self.height + 15) // 16)

    def ExportMvMemory(self) -> np.ndarray:
        mv = np.zeros((2, self.nblocks), dtype=np.int32)
        self._check(self._L.scpr_export_mv_memory(self._h, mv[0].ctypes.data_as(C.c_void_p), mv[1].ctypes.data_as(C.c_void_p)))
        return mv

    def ImportMvMemory(self, mv):
        mv = np.ascontiguousarray(mv, dtype=np.int32).reshape(2, self.nblocks)
        self._check(self._L.scpr_import_mv_memory(self._h, mv[0].ctypes.data_as(C.c_void_p), mv[1].ctypes.data_as(C.c_void_p)))
        return self

    def MotionPrepass(self, frames, ftypes, loss: int | None = None) -> np.ndarray:
        """scpr_motion_prepass: mvs[] as CompressBatch(frames, ftypes) would leave it; the codec itself is not changed"""
        import torch
        n = frames.shape[0]
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.numel() == n * self.frame_bytes
        ft = (C.c_int * n)(*[int(x) for x in ftypes])
        mv = np.zeros((2, self.nblocks), dtype=np.int32)
        torch.cuda.synchronize(frames.device)
        self._check(self._L.scpr_motion_prepass(self._h, C.c_void_p(frames.data_ptr()), n, ft, self.loss if loss is None else loss,
                                                mv[0].ctypes.data_as(C.c_void_p), mv[1].ctypes.data_as(C.c_void_p)))
        return mv

    def CrashHappened(self):
        self._L.scpr_crash_happened(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.scpr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(rc):
        if rc == SCPR_E_BAD_VERSION:
            raise BadVersionException(rc)
        if rc == SCPR_E_CAPACITY:
            raise CapacityError(f"scpr error {rc}: destination too small (nothing was coded)")
        if rc < 0:
            raise RuntimeError(f"scpr error {rc}")
        return rc

    # ScreenCodec::CompressFrame: (bytes, ftype_out); ftype_in 0 = key frame wanted, 1 = P allowed
    def CompressFrame(self, frame: np.ndarray, ftype: int = 1, loss: int | None = None, dst_len: int | None = None):
        src = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        assert src.size == self.frame_bytes, (src.size, self.frame_bytes)
        dst = np.empty(self.max_packet if dst_len is None else dst_len, dtype=np.uint8)
        ft = C.c_int(ftype)
        n = self._check(self._L.scpr_compress_frame(self._h, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                                    dst.size, C.byref(ft), self.loss if loss is None else loss))
        return bytes(dst[:n]), ft.value

    # ScreenCodec::DecompressFrame: (ret, frame bytes)
    def DecompressFrame(self, data: bytes, ftype: int, pitch: int | None = None):
        pitch = self.pitch if pitch is None else pitch
        out = np.zeros(pitch * self.height, dtype=np.uint8)
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        r = self._check(self._L.scpr_decompress_frame(self._h, buf.ctypes.data_as(C.c_void_p), len(data),
                                                      out.ctypes.data_as(C.c_void_p), pitch, ftype))
        return r, out

    # batch entry points: torch uint8 CUDA tensors in, out
    def CompressBatch(self, frames, ftypes, loss: int | None = None, out=None, sync: bool = True):
        """sync=False: the caller vouches that `frames` is complete (the C side works on the codec's own stream and waits for
        nothing else); the default waits for the whole device, which also waits for another codec's call in another thread"""
        import torch
        n = frames.shape[0]
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.numel() == n * self.frame_bytes
        own = out is None
        if own:
            # room for incompressible pictures (a literal pixel costs a little over three bytes); a batch that does not fit ends in
            # CapacityError with the codec left as the call found it.  Pass `out` to reuse a smaller buffer
            # (kept on the object and grown when a call needs more: 2.5 GB for 300 frames of 1080p is not allocated per call)
            need = min(n * self.max_packet, max(64 << 20, n * (self.width * self.height * 4 + 1024)))
            if getattr(self, "_out", None) is None or self._out.numel() < need or self._out.device != frames.device:
                self._out = None
                self._out = torch.empty(need, dtype=torch.uint8, device=frames.device)
            out = self._out
        ft = (C.c_int * n)(*[int(x) for x in ftypes])
        sizes = (C.c_uint32 * n)()
        if sync:
            torch.cuda.synchronize(frames.device)
        total = self._check(self._L.scpr_compress_batch(self._h, C.c_void_p(frames.data_ptr()), n, ft, self.loss if loss is None else loss,
                                                        C.c_void_p(out.data_ptr()), out.numel(), sizes))
        # (from the object's own buffer the packets are copied out - they are ~1 % of it - so that the next call cannot change them)
        return (out[:total].clone() if own else out[:total]), np.frombuffer(sizes, dtype=np.uint32).copy(), list(ft)

    def DecompressBatch(self, packets, sizes, ftypes, pitch: int | None = None, out=None, sync: bool = True):
        import torch
        n = len(sizes)
        pitch = self.pitch if pitch is None else pitch
        total = int(np.sum(sizes))
        assert packets.is_cuda and packets.dtype == torch.uint8 and packets.numel() >= total  # exact size: the C side needs no slack
        if out is None:
            out = torch.empty(n * pitch * self.height, dtype=torch.uint8, device=packets.device)
        sz = (C.c_uint32 * n)(*[int(x) for x in sizes])
        ft = (C.c_int * n)(*[int(x) for x in ftypes])
        if sync:  # (see CompressBatch)
            torch.cuda.synchronize(packets.device)
        r = self._check(self._L.scpr_decompress_batch(self._h, C.c_void_p(packets.data_ptr()), sz, ft, n, C.c_void_p(out.data_ptr()), pitch))
        return r, out

    # the batch calls with HOST memory (numpy arrays or CPU torch tensors, pinned or not): the reference's boundary hands over host
    # pointers (screencap.cpp:1632, :1695); the transfers run beside the kernels (scpr_compress_batch_host / scpr_decompress_batch_host)
    @staticmethod
    def _host_ptr(a):
        if isinstance(a, np.ndarray):
            assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
            return a.ctypes.data, a.size
        assert not a.is_cuda and a.is_contiguous() and a.element_size() == 1  # a CPU torch tensor
        return a.data_ptr(), a.numel()

    def HostPin(self, a):
        """scpr_host_pin: a host buffer the caller will reuse (numpy array / CPU tensor), pinned and mapped until HostUnpin or close()"""
        ptr, nbytes = self._host_ptr(a)
        self._check(self._L.scpr_host_pin(self._h, C.c_void_p(ptr), nbytes))
        return self

    def HostUnpin(self, a):
        ptr, _ = self._host_ptr(a)
        self._check(self._L.scpr_host_unpin(self._h, C.c_void_p(ptr)))
        return self

    def CompressBatchHost(self, frames, ftypes, loss: int | None = None, out=None):
        """frames: n * frame_bytes bytes of host memory; out: host buffer for the packets (default: a numpy array of the safe size).
        Returns (packets: the used part of `out`, sizes, ftypes produced)."""
        n = len(ftypes)
        ptr, nbytes = self._host_ptr(frames)
        assert nbytes == n * self.frame_bytes, (nbytes, n, self.frame_bytes)
        if out is None:
            out = np.empty(min(n * self.max_packet, max(64 << 20, n * (self.width * self.height * 4 + 1024))), dtype=np.uint8)
        optr, ocap = self._host_ptr(out)
        ft = (C.c_int * n)(*[int(x) for x in ftypes])
        sizes = (C.c_uint32 * n)()
        total = self._check(self._L.scpr_compress_batch_host(self._h, C.c_void_p(ptr), n, ft, self.loss if loss is None else loss, C.c_void_p(optr), ocap, sizes))
        return out[:total], np.frombuffer(sizes, dtype=np.uint32).copy(), list(ft)

    def DecompressBatchHost(self, packets, sizes, ftypes, pitch: int | None = None, out=None):
        """packets: host memory (exactly sum(sizes) bytes or more); out: host buffer for n frames of `pitch`-byte rows (default: numpy).
        Returns (frames decoded, out)."""
        n = len(sizes)
        pitch = self.pitch if pitch is None else pitch
        ptr, nbytes = self._host_ptr(packets)
        assert nbytes >= int(np.sum(sizes))
        if out is None:
            out = np.empty(n * pitch * self.height, dtype=np.uint8)
        optr, ocap = self._host_ptr(out)
        assert ocap >= n * pitch * self.height
        sz = (C.c_uint32 * n)(*[int(x) for x in sizes])
        ft = (C.c_int * n)(*[int(x) for x in ftypes])
        r = self._check(self._L.scpr_decompress_batch_host(self._h, C.c_void_p(ptr), sz, ft, n, C.c_void_p(optr), pitch))
        return r, out

    def last_timing(self):
        tot = C.c_float()
        st = (C.c_float * 32)()
        k = self._L.scpr_last_timing(self._h, C.byref(tot), st, 32)
        return tot.value, {self._L.scpr_stage_name(i).decode(): st[i] for i in range(k)}

    def debug_inject(self, what: int):
        """scpr_debug_inject (tests; the codec must have been created with SCPR_ENABLE_DEBUG_INJECT=1 in the environment): the next
        CompressBatch misbehaves on purpose - 1 fails between read-back and hand-over, 2 unsorted keys, 3 a stale record in k_rans_s"""
        self._check(self._L.scpr_debug_inject(self._h, what))

    def debug_rans_recoded(self) -> int:
        """calls whose rANS blocks were coded again by k_rans because k_rans_s' own check spoke"""
        return int(self._L.scpr_debug_rans_recoded(self._h))

    def debug_arena(self):
        """(compress side, decompress side) bytes allocated for dense tables"""
        a, b = C.c_uint64(), C.c_uint64()
        self._L.scpr_debug_arena(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def debug_entries(self):
        n = self._L.scpr_debug_entries(self._h, None, 0)
        out = np.zeros((max(n, 0), 2), dtype=np.uint16)
        if n > 0:
            self._L.scpr_debug_entries(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out


def debug_colour_chain(syms, f0: int = 32, device: int = 0) -> np.ndarray:
    """one colour context fed `syms` through the wave-per-chain encoder kernel -> (n, 2) uint16 entries"""
    L = load_library()
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    out = np.zeros((len(syms), 2), dtype=np.uint16)
    rc = L.scpr_debug_colour_chain(device, syms.ctypes.data_as(C.c_void_p), len(syms), f0, out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise RuntimeError(f"scpr_debug_colour_chain: {rc}")
    return out
