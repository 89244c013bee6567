"""Seeded synthetic "desktop" sequences.

Every pixel is a pure function of (seed, frame index, x, y) built from an
integer hash, so the same frames can be regenerated anywhere (CPU, GPU box)
without shipping data.  Content mix, chosen to exercise every part of the
ScreenPressor path:

* flat background and 12 flat-filled windows with 1-px borders and title bars
  (long predictor runs, 255-run caps);
* text-like 1-px glyph noise in windows (literal pixels, small colour contexts);
* a 2-axis additive gradient panel (predictor types 4/5, dense contexts);
* a rectangle translating a few px per frame and a vertically scrolling text
  window (exact-match motion search: vertical and 2-D hits);
* ~200 "sparkle" pixels per frame (partial blocks, changed-rect coding);
* optional noise patch (fraction of the frame) to force > 131072 coder entries.

Frames are RGB32 (B,G,R,A=255 byte order is irrelevant to the codec; alpha is
dropped on encode).
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _h32(a: np.ndarray) -> np.ndarray:
    """lowbias32 integer hash on uint64 arrays holding 32-bit values."""
    a = a & _M32
    a = ((a ^ (a >> np.uint64(16))) * np.uint64(0x7FEB352D)) & _M32
    a = ((a ^ (a >> np.uint64(15))) * np.uint64(0x846CA68B)) & _M32
    return a ^ (a >> np.uint64(16))


def _hs(*vals: int) -> int:
    """scalar hash chain"""
    acc = np.uint64(0x9E3779B9)
    for v in vals:
        acc = _h32(np.asarray(acc ^ np.uint64(v & 0xFFFFFFFF), dtype=np.uint64))
    return int(acc)


def _field(seed: int, y0: int, x0: int, h: int, w: int) -> np.ndarray:
    """h x w array of 32-bit hashes keyed by absolute coordinates"""
    ys = (np.arange(y0, y0 + h, dtype=np.int64) & 0xFFFF).astype(np.uint64)[:, None]
    xs = (np.arange(x0, x0 + w, dtype=np.int64) & 0xFFFF).astype(np.uint64)[None, :]
    return _h32((ys * np.uint64(0x10001) + xs * np.uint64(0x9E37) + np.uint64(seed & 0xFFFFFFFF)))


def _colour(seed: int, k: int) -> np.ndarray:
    v = _hs(seed, 0xC0104, k)
    return np.array([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], dtype=np.uint8)


def _text(seed: int, y0: int, x0: int, h: int, w: int, fg: np.ndarray, bg: np.ndarray, yshift: int = 0) -> np.ndarray:
    """text-like glyph noise: 9-px lines every 14 px, 5-px glyph cells + 1-px gap"""
    ys = np.arange(y0 + yshift, y0 + yshift + h, dtype=np.int64)[:, None]
    xs = np.arange(x0, x0 + w, dtype=np.int64)[None, :]
    in_line = (ys % 14) < 9
    in_cell = (xs % 6) < 5
    word_gap = (_h32(((xs // 36).astype(np.uint64) * np.uint64(77) + (ys // 14).astype(np.uint64) * np.uint64(1315423911) + np.uint64(seed & 0xFFFFFFFF))) % np.uint64(5)) == 0
    bits = _field(seed ^ 0x7E47, y0 + yshift, x0, h, w)
    ink = ((bits % np.uint64(100)) < 38) & in_line & in_cell & ~word_gap
    out = np.empty((h, w, 3), dtype=np.uint8)
    out[:] = bg
    out[ink] = fg
    # anti-aliased edge: a second shade on a few ink neighbours
    soft = ((bits >> np.uint64(8)) % np.uint64(100) < 12) & in_line & in_cell & ~word_gap & ~ink
    out[soft] = ((fg.astype(np.int32) + bg.astype(np.int32)) // 2).astype(np.uint8)
    return out


class DesktopSequence:
    """frame(t) -> (H, W, 4) uint8 RGB32 frame of the synthetic desktop."""

    def __init__(self, width: int, height: int, seed: int = 1, noise_fraction: float = 0.0, sparkles: int = 200, static: bool = False):
        self.W, self.H, self.seed = int(width), int(height), int(seed)
        self.noise_fraction = float(noise_fraction)
        self.sparkles = int(sparkles)
        self.static = bool(static)  # static=True: only sparkles/noise change between frames
        W, H = self.W, self.H
        self.windows = []
        for k in range(12):
            v = _hs(seed, 0xB0B, k)
            ww = max(8, W // 6 + (v & 0xFF) * W // 1024)
            wh = max(8, H // 6 + ((v >> 8) & 0xFF) * H // 1024)
            wx = (v >> 16) % max(1, W - ww)
            wy = (_hs(seed, 0xB0C, k)) % max(1, H - wh)
            kind = k % 4  # 0 flat, 1 text, 2 gradient, 3 scrolling text
            self.windows.append((wx, wy, ww, wh, kind, k))
        self.base = self._render_base()

    def _render_base(self) -> np.ndarray:
        W, H, seed = self.W, self.H, self.seed
        img = np.empty((H, W, 3), dtype=np.uint8)
        img[:] = _colour(seed, 100)
        # taskbar
        tb = max(2, H // 27)
        img[H - tb:, :] = _colour(seed, 101)
        for (wx, wy, ww, wh, kind, k) in self.windows:
            self._draw_window(img, wx, wy, ww, wh, kind, k, 0)
        return img

    def _draw_window(self, img, wx, wy, ww, wh, kind, k, t):
        seed = self.seed
        x2, y2 = min(self.W, wx + ww), min(self.H, wy + wh)
        if x2 - wx < 4 or y2 - wy < 4:
            return
        body = _colour(seed, 200 + k)
        img[wy:y2, wx:x2] = body
        title = min(y2 - wy - 2, max(3, self.H // 60))
        img[wy:wy + title, wx:x2] = _colour(seed, 300 + k)
        img[wy:y2, wx] = 0
        img[wy:y2, x2 - 1] = 0
        img[wy, wx:x2] = 0
        img[y2 - 1, wx:x2] = 0
        ix1, iy1, ix2, iy2 = wx + 2, wy + title + 1, x2 - 2, y2 - 2
        if ix2 <= ix1 or iy2 <= iy1:
            return
        h, w = iy2 - iy1, ix2 - ix1
        if kind == 1:
            img[iy1:iy2, ix1:ix2] = _text(seed + k, iy1, ix1, h, w, _colour(seed, 400 + k), body)
        elif kind == 2:
            ys = np.arange(h, dtype=np.int32)[:, None]
            xs = np.arange(w, dtype=np.int32)[None, :]
            g = np.empty((h, w, 3), dtype=np.uint8)
            g[..., 0] = ((xs // 2 + ys // 3 + 16 * k) & 0xFF).astype(np.uint8)
            g[..., 1] = ((xs // 3 + 2 * ys // 5 + 40) & 0xFF).astype(np.uint8)
            g[..., 2] = ((xs // 4 + ys // 2 + 90) & 0xFF).astype(np.uint8)
            img[iy1:iy2, ix1:ix2] = g
        elif kind == 3:
            shift = 0 if self.static else 2 * t
            img[iy1:iy2, ix1:ix2] = _text(seed + 31 * k, iy1, ix1, h, w, _colour(seed, 500 + k), body, yshift=shift)

    def frame24(self, t: int) -> np.ndarray:
        """(H, W, 3) uint8"""
        W, H, seed = self.W, self.H, self.seed
        img = self.base.copy()
        if not self.static:
            # scrolling windows are re-rendered with the current shift
            for (wx, wy, ww, wh, kind, k) in self.windows:
                if kind == 3:
                    self._draw_window(img, wx, wy, ww, wh, kind, k, t)
            # translating rectangle with a text texture (2-D motion hits)
            rw, rh = min(W, max(8, W // 8)), min(H, max(8, H // 8))
            rx = (W // 10 + 3 * t) % max(1, W - rw)
            ry = (H // 3 + t) % max(1, H - rh)
            img[ry:ry + rh, rx:rx + rw] = _text(seed ^ 0x5151, 0, 0, rh, rw, _colour(seed, 600), _colour(seed, 601))
        # sparkles
        if self.sparkles > 0:
            idx = np.arange(self.sparkles, dtype=np.uint64)
            hx = _h32(idx * np.uint64(2654435761) + np.uint64((seed * 7919 + t * 104729) & 0xFFFFFFFF))
            hy = _h32(hx + np.uint64(0x1234567))
            hc = _h32(hy + np.uint64(0x89ABCDE))
            xs = (hx % np.uint64(W)).astype(np.int64)
            ys = (hy % np.uint64(H)).astype(np.int64)
            img[ys, xs, 0] = (hc & np.uint64(0xFF)).astype(np.uint8)
            img[ys, xs, 1] = ((hc >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
            img[ys, xs, 2] = ((hc >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
        # noise patch (bottom-right), re-randomised each frame
        if self.noise_fraction > 0:
            nh = max(1, int(H * self.noise_fraction ** 0.5))
            nw = max(1, int(W * self.noise_fraction ** 0.5))
            f = _field(_hs(seed, 0xA015E, t), H - nh, W - nw, nh, nw)
            img[H - nh:, W - nw:, 0] = (f & np.uint64(0xFF)).astype(np.uint8)
            img[H - nh:, W - nw:, 1] = ((f >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
            img[H - nh:, W - nw:, 2] = ((f >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
        return img

    def frame(self, t: int) -> np.ndarray:
        """(H, W, 4) uint8, alpha = 255"""
        out = np.empty((self.H, self.W, 4), dtype=np.uint8)
        out[..., :3] = self.frame24(t)
        out[..., 3] = 255
        return out

    def frames(self, n: int, start: int = 0) -> np.ndarray:
        return np.stack([self.frame(start + i) for i in range(n)])


def pack24(frame24: np.ndarray) -> np.ndarray:
    """(H, W, 3) -> (H, stride) uint8 with DWORD-aligned rows (RGB24 DIB layout)"""
    h, w, _ = frame24.shape
    stride = (w * 3 + 3) & ~3
    out = np.zeros((h, stride), dtype=np.uint8)
    out[:, : w * 3] = frame24.reshape(h, w * 3)
    return out
