#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

Provenance: the reference cannot be built in this image (every translation unit needs
<windows.h>), so these streams come from the ORACLE (oracle/libspo.so, the
CPU restatement of the reference algorithm) on seeded synthetic input.  They pin the
oracle against regressions and give the GPU box byte-level targets that do not depend on
the oracle being rebuilt there.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as O  # noqa: E402
from screenpressor_amd.synth import DesktopSequence, pack24  # noqa: E402

CASES = [
    # name, width, height, bpp, frames, key frames, seed, noise, workers, loss, keep_bytes
    ("desktop_64x48_ip", 64, 48, 32, 8, (0,), 3, 0.0, 1, 0, True),
    ("desktop_100x37_ip_padded", 100, 37, 32, 6, (0,), 4, 0.0, 1, 0, True),
    ("desktop_100x37_rgb24", 100, 37, 24, 6, (0,), 4, 0.0, 1, 0, False),
    ("desktop_128x96_workers4_keys", 128, 96, 32, 3, (0, 1, 2), 9, 0.0, 4, 0, True),
    ("desktop_100x37_loss2", 100, 37, 32, 4, (0,), 6, 0.0, 1, 2, True),
    ("noise_320x240_multiblock", 320, 240, 32, 2, (0,), 5, 0.6, 1, 0, False),
    ("desktop_640x480_c1", 640, 480, 32, 60, (0,), 1, 0.0, 1, 0, False),      # BASELINE configs[0]
    ("desktop_1080p_keys", 1920, 1080, 32, 3, (0, 1, 2), 1, 0.0, 1, 0, False),  # BASELINE configs[1] content
    ("desktop_1080p_ip", 1920, 1080, 32, 4, (0,), 1, 0.0, 1, 0, False),        # BASELINE configs[2] content
    ("desktop_1080p_rgb24", 1920, 1080, 24, 3, (0, 1, 2), 1, 0.0, 1, 0, False),  # BASELINE configs[4]: same stream as desktop_1080p_keys
    ("desktop_33x21_rgb16_odd", 33, 21, 16, 4, (0,), 12, 0.0, 1, 0, True),      # RGB16 rows back to back, odd width (screencap.cpp:1668)
]
# streams of the legacy version 2 format (range coder): the product only decodes these
V2_CASES = [("desktop_100x37_v2_ip", 100, 37, 32, 7, (0, 4), 7, 0.0, 1, 0, True), ("desktop_320x240_v2_ip", 320, 240, 32, 5, (0,), 8, 0.0, 1, 0, True)]
# version 3 streams (f0 = 64, header 0x22/0x21, screencap.cpp:1613, :1700): decode only as well (the compress side writes version 4)
V3_CASES = [("desktop_100x37_v3_ip", 100, 37, 32, 7, (0, 4), 11, 0.0, 1, 0, True), ("chan0noise_160x120_v3_keys", 160, 120, 32, 2, (0, 1), 13, 0.0, 1, 0, True)]
CASES_ALL = CASES + V2_CASES + V3_CASES


def to_rgb16(f24, w, h):
    """RGB555 rows back to back (the layout the reference's compress side reads)"""
    c = f24.astype(np.uint16) >> 3
    px = ((c[..., 2] << 10) | (c[..., 1] << 5) | c[..., 0]).astype(np.uint16)
    return np.ascontiguousarray(px).view(np.uint8).reshape(h, w * 2)


def frames_of(case):
    name, w, h, bpp, n, keys, seed, noise, workers, loss, keep = case
    if name.startswith("chan0noise"):
        # byte 0 of every pixel random, bytes 1 and 2 from two values each: a handful of colour contexts that each see
        # more than 14 different symbols before the first repeat, i.e. the Cx2 -> Cx6 promotion (Cx6::create23,
        # ans_contexts.h:492-531), the one place where f0 (32 for version 4, 64 for version 3) changes intervals
        rng = np.random.default_rng(seed)
        for t in range(n):
            f = np.full((h, w, 4), 255, np.uint8)
            f[..., 0] = rng.integers(0, 256, (h, w))
            f[..., 1] = 40 + 8 * rng.integers(0, 2, (h, w))
            f[..., 2] = 80 + 8 * rng.integers(0, 2, (h, w))
            yield t, f
        return
    seq = DesktopSequence(w, h, seed=seed, noise_fraction=noise)
    for t in range(n):
        yield t, (seq.frame(t) if bpp == 32 else pack24(seq.frame24(t)) if bpp == 24 else to_rgb16(seq.frame24(t), w, h))


def main():
    manifest = {}
    for case in CASES_ALL:
        name, w, h, bpp, n, keys, seed, noise, workers, loss, keep = case
        version = 2 if case in V2_CASES else 3 if case in V3_CASES else 4
        enc = O.OracleCodec(w, h, bpp, loss=loss, workers=workers, version=version)
        packets, types = [], []
        for t, f in frames_of(case):
            data, ft = enc.compress(f, key=(t in keys))
            packets.append(data)
            types.append(ft)
        blob = b"".join(packets)
        entry = {"width": w, "height": h, "bpp": bpp, "frames": n, "keys": list(keys), "seed": seed, "noise": noise,
                 "workers": workers, "loss": loss, "version": version, "sizes": [len(p) for p in packets], "ftypes": types,
                 "sha256": hashlib.sha256(blob).hexdigest(),
                 "frame_sha256": [hashlib.sha256(p).hexdigest() for p in packets]}
        if keep:
            with open(os.path.join(HERE, name + ".bin"), "wb") as fh:
                fh.write(blob)
            entry["file"] = name + ".bin"
        manifest[name] = entry
        print(name, len(blob), entry["sha256"][:16])
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)


if __name__ == "__main__":
    main()
