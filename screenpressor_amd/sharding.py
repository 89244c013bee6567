"""Frame sharding across the GPUs of one node.

The unit of independence is the GOP (a key frame and the P-frames that depend on it): RenewI
resets every model at a key frame (screencap.cpp:343).  A stream of G GOPs is cut into
contiguous GOP ranges, one per rank; each rank encodes its range as an independent stream and
the packets are gathered to rank 0 in frame order.  No collective touches the data path until
that gather.

Three pieces of state do cross key frames in the reference.  (1, 2) The flat-frame rule compares with the previous
flat frame and P-frames need "a frame has been coded" (last_was_flat / last_flat_clr / fn, screencap.cpp:1490-1504):
`shard_seed` computes that state from the frames right before a cut and `scpr_seed_shard` installs it, so a shard
that starts at or after a repeated flat colour produces the single stream's bytes.  (3) The motion search remembers, per
block, the last vector found in ANY earlier frame (mvs[], screencap.cpp:96-97; read as "the vector of the block above",
:726-735; never reset - RenewI :178-198 touches models only).  `handover_mv_memory` passes it down the ranks before
anything is coded: rank r takes the memory rank r-1's shard leaves behind, runs the motion-only pre-pass over its own
shard (scpr_motion_prepass: conversion, block compare, motion search) and passes the result on; then every rank imports
what it received and all ranks code their shards side by side.  With (1)-(3) seeded a shard's packets are the single
stream's packets.

What the hand-over costs: the pre-pass is the motion stage of the encoder and that is about HALF of an I+P encode (measured:
34 ms against 70 ms for 150 frames of 3840x2160; 21.5 of ~37 ms at 1080p), and the chain is serial - rank r waits
for r pre-passes, so at N = 8 the last rank starts coding after ~7 x 34 ms, several encode-times.  It is real serial work
inside the timed configs[3] step and grows linearly with N; bench.py reports it as its own field (`handover_ms`) so that the
strong-scaling number can be read for what it is.
"""
from __future__ import annotations

import numpy as np


def gop_starts(ftypes_in) -> list[int]:
    """indices where the caller asks for a key frame (ftype 0)"""
    return [i for i, t in enumerate(ftypes_in) if t == 0]


def shard_gops(ftypes_in, world: int) -> list[tuple[int, int]]:
    """contiguous [lo, hi) frame ranges, one per rank, cut at GOP starts and balanced by frame count"""
    n = len(ftypes_in)
    starts = gop_starts(ftypes_in)
    if not starts or starts[0] != 0:
        starts = [0] + starts
    cuts = [0]
    for r in range(1, world):
        target = n * r / world
        best = min(starts, key=lambda s: (abs(s - target), s))
        cuts.append(max(best, cuts[-1]))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def flat_colour(frame: np.ndarray, width: int, height: int, bpp: int, masks=(0x7C00, 0x3E0, 0x1F)):
    """b0 | b1 << 8 | b2 << 16 of the picture's one colour, or None if it is not a flat picture (IsFlat,
    screencap.cpp:1436-1444, on the colour bytes the codec sees: RGB32 drops byte 3, :1652-1664; RGB16 is converted to
    RGB24 with the caller's masks first, :1665-1678 - rows of width * 2 bytes back to back, :1668 - and the flat rule
    :1488-1500 applies to what comes out)"""
    assert bpp in (16, 24, 32)
    if bpp == 16:
        w16 = np.asarray(frame, dtype=np.uint8).reshape(-1)[: height * width * 2].view("<u2").astype(np.uint32)
        sh = [next((k for k in range(16) if (m >> k) & 1), 16) for m in masks]  # ScreenCodec::Init's shifts, screencap.cpp:1572-1583
        c = [(w16 & np.uint32(m)) >> np.uint32(q) if q < 16 else np.zeros_like(w16) for m, q in zip(masks, sh)]
        v = (c[0] & 255) | ((c[1] & 255) << 8) | ((c[2] & 255) << 16)
        return int(v[0]) if (v == v[0]).all() else None
    px = bpp // 8
    pitch = width * 4 if bpp == 32 else (width * 3 + 3) & ~3
    rows = np.asarray(frame, dtype=np.uint8).reshape(height, pitch)[:, : width * px].reshape(height, width, px)[..., :3]
    c = rows[0, 0]
    if not (rows == c).all():
        return None
    return int(c[0]) | (int(c[1]) << 8) | (int(c[2]) << 16)


def shard_seed(get_frame, lo: int, width: int, height: int, bpp: int, masks=(0x7C00, 0x3E0, 0x1F)):
    """Arguments of scpr_seed_shard / ScreenCodec.SeedShard for a shard whose first frame is `lo` of the stream:
    (frames_before, last_was_flat, last_flat_rgb).  `get_frame(t)` returns input frame t; only the frames right
    before the cut are looked at (one, unless they are flat).  GOPs share nothing else except the motion-vector
    memory (module docstring)."""
    if lo <= 0:
        return 0, False, 0
    last = flat_colour(get_frame(lo - 1), width, height, bpp, masks)
    # fn counts coded (non-flat) frames only (flat frames return before fn++, screencap.cpp:1488-1500)
    coded, t = 0, lo - 1
    while t >= 0 and coded == 0:
        coded += flat_colour(get_frame(t), width, height, bpp, masks) is None
        t -= 1
    return coded, last is not None, (last or 0)


def handover_mv_memory(dist, rank: int, world: int, nblocks: int, prepass, device=None) -> np.ndarray:
    """The motion-vector memory this rank's shard starts from, as a (2, nblocks) int32 array (x row, y row).

    `prepass(mv_in) -> mv_out`: the memory after this rank's shard when it starts from `mv_in` (ScreenCodec.ImportMvMemory +
    MotionPrepass on the GPU; any codec that can do the same in the CPU tests).  The last rank never runs it.  The chain
    is serial by nature (shard r's vectors depend on shard r-1's); one broadcast per link rather than send/recv pairs: a
    collective every backend has, no pairwise communicators to set up, and the payload is 8 bytes per block (260 KB at
    3840x2160).  Rank 0 starts from zeros (calloc, screencap.cpp:96-97)."""
    import torch
    buf = torch.zeros(2 * nblocks, dtype=torch.int32, device=device if device is not None else "cpu")
    mine = np.zeros((2, nblocks), dtype=np.int32)
    for src in range(world - 1):  # link src -> src + 1
        if rank == src:
            out = np.ascontiguousarray(prepass(mine), dtype=np.int32).reshape(-1)
            assert out.size == 2 * nblocks
            buf.copy_(torch.from_numpy(out))
        dist.broadcast(buf, src=src)
        if rank == src + 1:
            mine = buf.cpu().numpy().reshape(2, nblocks).copy()
    return mine


def gather_packets_begin(dist, rank: int, world: int, payload: np.ndarray, sizes: np.ndarray, device=None):
    """Starts the gather of gather_packets() and returns a handle for gather_packets_end(): the transfers run beside whatever
    the caller does in between (bench.py: the decode of the same step - the packets are final once the encoder returns).
    `payload` must stay untouched until the end call."""
    import torch
    t_payload = payload if isinstance(payload, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(payload))
    t_sizes = torch.as_tensor(np.asarray(sizes, dtype=np.int64))
    if device is not None:
        t_payload, t_sizes = t_payload.to(device), t_sizes.to(device)
    meta = torch.tensor([t_payload.numel(), t_sizes.numel()], dtype=torch.int64, device=t_payload.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)  # every rank learns every rank's byte and frame count (16 bytes each) ...
    counts = [(int(m[0]), int(m[1])) for m in metas]
    max_bytes = max(c[0] for c in counts)
    max_frames = max(c[1] for c in counts)
    pad_p = torch.zeros(max_bytes, dtype=torch.uint8, device=t_payload.device)
    pad_p[: t_payload.numel()] = t_payload
    pad_s = torch.zeros(max_frames, dtype=torch.int64, device=t_payload.device)
    pad_s[: t_sizes.numel()] = t_sizes
    # ... and the packets themselves travel to rank 0 only (sizes, then a gather of the chunks to rank 0)
    gp = [torch.empty_like(pad_p) for _ in range(world)] if rank == 0 else None
    gs = [torch.empty_like(pad_s) for _ in range(world)] if rank == 0 else None
    works = [dist.gather(pad_p, gather_list=gp, dst=0, async_op=True), dist.gather(pad_s, gather_list=gs, dst=0, async_op=True)]
    return {"rank": rank, "world": world, "counts": counts, "gp": gp, "gs": gs, "works": works, "keep": (pad_p, pad_s, t_payload, t_sizes)}


def gather_packets_end(handle):
    """Waits for the transfers of gather_packets_begin(); rank 0 gets (packets, sizes) in rank (= frame) order, the others
    (None, None)."""
    import torch
    for w in handle["works"]:
        w.wait()
    if handle["rank"] != 0:
        return None, None
    counts, gp, gs = handle["counts"], handle["gp"], handle["gs"]
    out_p = torch.cat([gp[r][: counts[r][0]] for r in range(handle["world"])])
    out_s = torch.cat([gs[r][: counts[r][1]] for r in range(handle["world"])])
    return out_p, out_s


def gather_packets(dist, rank: int, world: int, payload: np.ndarray, sizes: np.ndarray, device=None):
    """Rank 0 receives every rank's packets and per-frame sizes, in rank (= frame) order.
    `payload` is a uint8 array (numpy on CPU/gloo, or a torch tensor on the GPU for RCCL)."""
    return gather_packets_end(gather_packets_begin(dist, rank, world, payload, sizes, device=device))
