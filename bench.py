#!/usr/bin/env python3
"""bench.py - MPix/s encode+decode of the ScreenPressor hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch resident in HBM: compress the frames to packets
(scpr_compress_batch), decompress them back (scpr_decompress_batch), and - with more than one rank - gather
the compressed chunks to rank 0 over RCCL.  value = pixels of all ranks / max-over-ranks step time
(W*H*N / (t_enc + t_dec)).

Workloads (config.workload names the one measured):
  keys  BASELINE configs[1]: 1920x1080 RGB32, every frame a key frame, 300 frames per GPU - the default, at every N
        (--bpp 24: configs[4]).  Multi-GPU: every rank its own 300 frames, weak scaling, no data-path collective
        but the final gather (frames are independent units: the task's rule for paths that partition).
  ip    configs[2]: the same frames, key frame every --gop frames.
  c4    configs[3]: 3840x2160, ONE stream of 1200 frames with a key frame every 150, cut at key frames into one
        contiguous GOP range per rank (screenpressor_amd.sharding.shard_gops), strong scaling.
With N = 1 and no --no-others the line also carries config.others: configs[2] (K = 50 and one 300-frame GOP),
configs[3]'s one-GPU share (4K x 150, as key frames and as one GOP) and configs[4], each with encode / decode /
combined MPix/s, compressed bytes, sha256 of the stream, a bounded CPU sample and the parity flag against it; and
(unless --no-n8) the work of an N = 8 headline run - the eight 300-frame streams of seeds 1..8 - timed on this one
GPU: the same-stream denominator for the driver's N = 8 line (n8_same_stream_entry).

Launching: `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: it starts N
ranks (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU and
relays rank 0's line.  Under `python -m torch.distributed.run` (WORLD_SIZE set) the process is a rank.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
METRIC = "MPix/s encode+decode, 1080p/4K RGB32; bitstream byte-identical to ref"  # BASELINE.json's metric; see "parity" in the line for what was checked


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["keys", "ip", "c4"], default="keys")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (keys, ip: default 300) or of the whole stream (c4: default 1200)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bpp", type=int, choices=[32, 24], default=32, help="24: packed 3-byte pixels, rows padded to 4 bytes (BASELINE configs[4])")
    ap.add_argument("--gop", type=int, default=None, help="key frame interval (ip: default 50; c4: default 150)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the workload timed on the host CPU (default: the whole workload at 1080p keys, a bounded sample otherwise)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--c4-leg", action="store_true", help="run the configs[3] leg of an N > 1 run at N = 1 too (to rehearse it)")
    ap.add_argument("--c4-frames", type=int, default=None, help="frames of the configs[3] leg's stream (default 1200)")
    ap.add_argument("--no-others", action="store_true", help="N = 1: do not measure the other single-GPU configs")
    ap.add_argument("--no-n8", action="store_true", help="N = 1: do not time the 8 x 300-frame stream set an N = 8 headline run shards (config.others' same-stream denominator; ~1.5 min of host rendering)")
    ap.add_argument("--no-host-boundary", action="store_true", help="N = 1: skip config.host_boundary and the per-frame calls (profiling runs: their launches would mix into the per-kernel figures of the HBM-resident step)")
    ap.add_argument("--rehearse-one-gpu", action="store_true", help="N > 1 on a one-GPU box: every rank on GPU 0, collectives over gloo on CPU tensors (rehearses the multi-rank path with the real codecs; not a scaling measurement)")
    ap.add_argument("--selftest-launcher", action="store_true", help="no codec, no GPU: ranks exchange synthetic packets over gloo (tests the launcher, the sharding and the gather)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher ---
def spawn_ranks(args, argv):
    """the parent of a `--gpus N` run: starts N ranks (fresh child processes, before anything here touches a GPU), relays rank 0's
    JSON line, passes rank 0's stderr through as it comes (a long run shows its progress; nothing is lost if the launcher is
    killed), keeps every other rank's stderr in a NAMED file under gpurun_out/ranks/ (it survives the launcher) and, when a rank
    dies, says which one and ends the others - a rank that is gone before the first collective would otherwise leave the rest
    waiting for the RCCL timeout with nothing said"""
    import threading
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = os.path.join(ROOT, "gpurun_out", "ranks")
    os.makedirs(logdir, exist_ok=True)
    procs, errs, paths = [], [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        path = os.path.join(logdir, f"rank{r}_of{n}_pid{os.getpid()}.stderr")
        paths.append(path)
        ef = open(path, "w+b")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=subprocess.PIPE if r == 0 else ef))
    out = []

    def relay0():  # rank 0's stderr: to ours line by line, and to its file
        for line in iter(procs[0].stderr.readline, b""):
            errs[0].write(line)
            errs[0].flush()
            sys.stderr.write(line.decode(errors="replace"))
            sys.stderr.flush()
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    relay = threading.Thread(target=relay0, daemon=True)
    relay.start()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            time.sleep(2.0)  # (ranks that are about to fail by themselves get to say why)
            for p in procs:
                if p.poll() is None:
                    p.kill()  # the exact processes started above
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    relay.join(timeout=10)
    sys.stdout.write((out[0] if out else b"").decode())
    sys.stdout.flush()

    def tail(r, nbytes=3000):
        errs[r].flush()
        errs[r].seek(0, os.SEEK_END)
        size = errs[r].tell()
        errs[r].seek(max(0, size - nbytes))
        return errs[r].read().decode(errors="replace")
    rc = 0
    if failed is not None:
        sys.stderr.write(f"[bench] rank {failed} of {n} exited with code {procs[failed].returncode}; the other ranks were stopped. Its stderr ({paths[failed]}) ends:\n{tail(failed)}\n")
        for r in range(n):
            if r != failed and procs[r].returncode not in (0, -9):
                sys.stderr.write(f"[bench] rank {r} exited with code {procs[r].returncode}; its stderr ({paths[r]}) ends:\n{tail(r, 1500)}\n")
        rc = 1
    for ef in errs:
        ef.close()
    if rc == 0:
        for q in paths:  # (a clean run leaves nothing behind)
            try:
                os.remove(q)
            except OSError:
                pass
    return rc


# ------------------------------------------------------------------------------------------------ workloads ---
def _render(job):
    """worker of the frame pool (spawned processes: numpy only)"""
    w, h, seed, bpp, t0, t1 = job
    import numpy as np
    sys.path.insert(0, ROOT)
    from screenpressor_amd.synth import DesktopSequence, pack24
    seq = DesktopSequence(w, h, seed=seed)
    return np.stack([np.ascontiguousarray(seq.frame(t) if bpp == 32 else pack24(seq.frame24(t))).reshape(-1) for t in range(t0, t1)])


def host_cores(world=1):
    """cores ONE rank may use: its affinity mask (not the machine's count) shared by the `world` ranks of the job, at most
    one GPU's share of the box (16)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    return max(1, min(n // max(1, world), 16))


def make_frames(w, h, seed, bpp, t0, t1, dev=None, world=1):
    """frames t0..t1-1 of the seeded synthetic desktop, rendered by a pool of host processes (a 4K frame takes
    0.4 s of numpy); returned as one torch uint8 tensor (n, frame_bytes) on `dev` (or a numpy array).  The pool is this
    rank's share of the cores the job may use: `world` ranks render at the same time, and the box limits processes."""
    import multiprocessing as mp
    import numpy as np
    n = t1 - t0
    nproc = max(1, min(16 if n > 600 else 8, host_cores(world), n // 4))
    pitch = w * 4 if bpp == 32 else (w * 3 + 3) & ~3
    if dev is not None:
        import torch
        out = torch.empty((n, h * pitch), dtype=torch.uint8, device=dev)
    else:
        out = np.empty((n, h * pitch), dtype=np.uint8)
    per = max(1, min(16, (n + nproc - 1) // nproc))  # frames per job (bounded: 16 4K frames are 0.5 GB through a pipe)
    jobs = [(w, h, seed, bpp, a, min(t1, a + per)) for a in range(t0, t1, per)]

    def put(a, block):
        if dev is not None:
            out[a - t0:a - t0 + len(block)] = torch.from_numpy(block).to(dev)
        else:
            out[a - t0:a - t0 + len(block)] = block
    if nproc == 1:
        for j in jobs:
            put(j[4], _render(j))
    else:
        # (closed and joined, not left to Pool.__exit__: that one calls terminate(), the workers get SIGTERM, and under rocprofv3 its
        # chained signal handler prints an abort banner per worker into the profile's log)
        pool = mp.get_context("spawn").Pool(nproc)
        try:
            for j, block in zip(jobs, pool.imap(_render, jobs)):
                put(j[4], block)
        finally:
            pool.close()
            pool.join()
    return out


class Workload:
    """what one rank encodes: frames [lo, hi) of stream `seed`, with the caller's key-frame requests"""

    def __init__(self, name, w, h, bpp, seed, lo, hi, ftypes, scaling, total_frames, note=""):
        self.name, self.w, self.h, self.bpp, self.seed, self.lo, self.hi = name, w, h, bpp, seed, lo, hi
        self.ftypes, self.scaling, self.total_frames, self.note = ftypes, scaling, total_frames, note
        self.n = hi - lo
        self.pitch = w * 4 if bpp == 32 else (w * 3 + 3) & ~3


def describe(args, rank, world):
    sys.path.insert(0, ROOT)
    from screenpressor_amd.sharding import shard_gops
    wl = args.workload
    if wl == "c4":
        w, h, total, k = args.width or 3840, args.height or 2160, args.frames or 1200, args.gop or 150
        ft = [0 if t % k == 0 else 1 for t in range(total)]
        lo, hi = shard_gops(ft, world)[rank]
        name = (f"BASELINE configs[3]: {w}x{h} RGB{args.bpp}, one stream of {total} frames, key frame every {k}, cut at key frames into "
                f"{world} contiguous GOP range(s), synthetic desktop seed 1")
        return Workload(name, w, h, args.bpp, 1, lo, hi, ft[lo:hi], "strong", total)
    w, h, n = args.width or 1920, args.height or 1080, args.frames or 300
    if wl == "keys":
        ft = [0] * n
        name = f"BASELINE configs[{1 if args.bpp == 32 else 4}]: {w}x{h} RGB{args.bpp} key-frame-only (I-frames), {n} frames per GPU, synthetic desktop seed 1+rank"
    else:
        k = args.gop or 50
        ft = [0 if t % k == 0 else 1 for t in range(n)]
        name = f"BASELINE configs[2]: {w}x{h} RGB{args.bpp} I+P, key frame every {k}, {n} frames per GPU, synthetic desktop seed 1+rank"
    return Workload(name, w, h, args.bpp, 1 + rank, 0, n, ft, "weak", n * world)


# ------------------------------------------------------------------------------------------------ one config ---
class Runner:
    """codec pair + device buffers for one frame geometry; run() = timed passes of the hot path over a batch"""

    def __init__(self, dev, local_rank, w, h, bpp, n):
        import torch
        from screenpressor_amd.codec import ScreenCodec
        self.torch, self.dev, self.w, self.h, self.bpp, self.n = torch, dev, w, h, bpp, n
        self.pitch = w * 4 if bpp == 32 else (w * 3 + 3) & ~3
        self.enc = ScreenCodec(local_rank).Init(w, h, bpp)
        self.dec = ScreenCodec(local_rank).Init(w, h, bpp)
        self.packets = torch.empty(max(256 << 20, n * w * h // 2), dtype=torch.uint8, device=dev)
        self.decoded = torch.empty(n * h * self.pitch, dtype=torch.uint8, device=dev)

    def reset(self):
        W, H, BPP = self.w, self.h, self.bpp
        self.enc.Deinit(); self.enc.Init(W, H, BPP)
        self.dec.Deinit(); self.dec.Init(W, H, BPP)

    def same(self, dec, frames):
        return bool(self.torch.equal(dec.reshape(frames.shape[0], -1), frames))

    def step(self, frames, ftypes, seed=None, after=None, reset=True):
        """one pass of the hot path: fresh codecs, `seed(enc)` (a shard that does not start the stream: what crosses key
        frames, sharding.py), compress, decompress, and the exchange step: `after(packets, sizes)` once both are done, or - a
        pair (begin, end) - `begin(packets, sizes)` as soon as the packets exist and `end()` after the decode, so that the
        transfers run beside the decode"""
        if reset:
            self.reset()
            if seed:
                seed(self.enc)
        begin, end = after if isinstance(after, tuple) else (None, None)
        t0 = time.perf_counter()
        out, sizes, ft = self.enc.CompressBatch(frames, ftypes, out=self.packets)
        t1 = time.perf_counter()
        _, se = self.enc.last_timing()
        if begin:
            begin(out, sizes)
            t1b = time.perf_counter()
        else:
            t1b = t1
        r, dec = self.dec.DecompressBatch(out, sizes, ft, out=self.decoded)
        t2 = time.perf_counter()
        _, sd = self.dec.last_timing()
        assert r == len(ftypes)
        if end:
            end()
        elif after:
            after(out, sizes)
        return out, sizes, ft, dec, t1 - t0, t2 - t1b, {k: v for k, v in list(se.items()) + list(sd.items()) if v > 0}


def shard_seeder(env, wl, frames):
    """seed(enc) for a rank whose shard is frames [wl.lo, wl.hi) of a stream cut over env.world ranks: fn > 0 (the synthetic
    desktop has no flat frames) and the motion-vector memory handed down the ranks (sharding.handover_mv_memory: a broadcast
    per link, the motion-only pre-pass of this rank's shard in between).  Errors of the pre-pass are kept in `seed.error`
    instead of raised: every rank must reach every collective of the chain."""
    from screenpressor_amd.sharding import handover_mv_memory
    import numpy as np
    if env.world == 1 and not wl.lo:
        return None

    def seed(enc):
        seed.error = None
        if wl.lo:
            enc.SeedShard(wl.lo, False, 0)

        def prepass(mv_in):
            try:
                enc.ImportMvMemory(mv_in)
                return enc.MotionPrepass(frames, wl.ftypes)
            except Exception as e:  # noqa: BLE001
                seed.error = repr(e)
                return np.zeros_like(mv_in)
        mv = handover_mv_memory(env.dist, env.rank, env.world, enc.nblocks, prepass, device=env.cdev) if env.world > 1 else None
        if mv is not None:
            enc.ImportMvMemory(mv)
    seed.error = None
    return seed


def csrc_digest():
    """sha256 over the kernels' sources (screenpressor_amd/csrc, sorted by name): what a recorded profile is a profile OF.
    tools/summarize_profiles.py writes it into profiles/r*_pmc_hbm_traffic.json; a record whose digest differs from the
    sources of this run predates a change to the kernels and is not quoted"""
    hh = hashlib.sha256()
    d = os.path.join(ROOT, "screenpressor_amd", "csrc")
    for name in sorted(os.listdir(d)):
        hh.update(name.encode())
        hh.update(open(os.path.join(d, name), "rb").read())
    return hh.hexdigest()


def stream_sha256(packets_host):
    return hashlib.sha256(packets_host.tobytes()).hexdigest()


def cpu_sample(wl, frames_host, nf, gop, threads_all):
    """the CPU port (oracle/libspo.so) over the first nf frames of the workload: the one-thread canonical run (workers = 1:
    the stream the GPU is compared with) and, for key frames, the reference's two-stage shape on every host core
    (workers = threads = cores: another, equally valid stream - the band count is bitstream-visible)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    pix = nf * wl.w * wl.h / 1e6
    one = O.time_stream(frames_host[:nf], wl.w, wl.h, wl.bpp, key_interval=gop, workers=1, threads=1)
    assert one["bad"] == 0
    res = {"one_thread": {"value": round(pix / (one["t_enc"] + one["t_dec"]), 2), "enc_MPix_s": round(pix / one["t_enc"], 2), "dec_MPix_s": round(pix / one["t_dec"], 2), "cores": 1, "workers": 1}}
    if threads_all and threads_all > 1:
        nfa = max(1, min(nf, 100))
        allc = O.time_stream(frames_host[:nfa], wl.w, wl.h, wl.bpp, key_interval=gop, workers=threads_all, threads=threads_all)
        assert allc["bad"] == 0
        pa = nfa * wl.w * wl.h / 1e6
        res["all_cores"] = {"value": round(pa / (allc["t_enc"] + allc["t_dec"]), 2), "enc_MPix_s": round(pa / allc["t_enc"], 2), "dec_MPix_s": round(pa / allc["t_dec"], 2),
                            "cores": threads_all, "workers": threads_all, "frames": nfa,
                            "shape": "row bands of a key frame on a pool of `cores` threads + one rANS thread (squad.cpp:116-130, ransmt.h:92-105); the model stage and the whole decoder are one thread, as in the reference"}
    return one, res


def parity_of(packets_host, sizes, one, nf):
    """packets of the first nf frames against the oracle's (size and FNV-1a of every packet)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as O
    offs = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
    ok = [int(s) for s in sizes[:nf]] == [int(s) for s in one["sizes"][:nf]]
    if ok:
        for t in range(nf):
            if O.fnv1a(packets_host[offs[t]:offs[t + 1]]) != int(one["frame_fnv"][t]):
                ok = False
                break
    return ok


def measure(runner, wl, frames, steps, warmup, barrier=None, seed=None, after=None):
    tor = runner.torch
    for _ in range(warmup):
        runner.step(frames, wl.ftypes, seed, after)
    if barrier:
        barrier()
    acc, te, td, samples = {}, 0.0, 0.0, []
    t0 = time.perf_counter()
    for _ in range(steps):
        out, sizes, ft, dec, a, b, st = runner.step(frames, wl.ftypes, seed, after)
        te, td = te + a, td + b
        samples.append((a, b))
        for k, v in st.items():
            acc[k] = acc.get(k, 0.0) + v
    if barrier:
        barrier()
    else:
        tor.cuda.synchronize(runner.dev)
    elapsed = time.perf_counter() - t0
    assert runner.same(dec, frames), "the decoded frames differ from the input: the round trip is not lossless"
    return dict(out=out, sizes=sizes, ft=ft, elapsed=elapsed, t_enc=te / steps, t_dec=td / steps, samples=samples, stage_ms={k: v / steps for k, v in acc.items()})


def golden_stream(name):
    """the committed sha256 of a full bench stream (tests/golden/manifest.json, made by the oracle: make_golden.py --streams)"""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json"))).get(name)
    except Exception:  # noqa: BLE001
        return None


def golden_stream_check(entry, packets_host, sizes, first_frame=0):
    """packets (one uint8 array, in frame order, starting at frame `first_frame` of the stream - a GOP start) against the
    committed hashes: the whole stream when all of it is here, otherwise GOP by GOP (a shorter run of the same stream, or
    one rank's shard).  True / False, or a string saying why nothing could be compared."""
    import numpy as np
    if not entry:
        return "not checked: no committed hash for this stream"
    n, k = len(sizes), int(entry["key_interval"])
    if first_frame == 0 and n == entry["frames"]:
        return stream_sha256(packets_host) == entry["sha256"] and int(packets_host.size) == entry["bytes"]
    if not entry.get("gop_sha256") or first_frame % k or n % k or first_frame + n > entry["frames"]:
        return "not checked: the run is not a whole number of GOPs of the committed stream"
    offs = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
    g0 = first_frame // k
    return all(hashlib.sha256(packets_host[offs[g * k]:offs[(g + 1) * k]].tobytes()).hexdigest() == entry["gop_sha256"][g0 + g] for g in range(n // k))


def other_config(runner_cache, dev, local_rank, label, wl, frames, gop, cpu_frames, no_cpu, stream=None, threads_all=0, passes=3):
    """one of the single-GPU configs that is not the headline: 1 warm-up + `passes` timed passes (a P-frame chain moves a few
    per cent from pass to pass and more from box to box: median and best are both given), the full stream against its
    committed hash, a bounded CPU sample and the parity flag against it"""
    import statistics
    key = (wl.w, wl.h, wl.bpp)
    if key not in runner_cache or runner_cache[key].n < wl.n:
        runner_cache[key] = Runner(dev, local_rank, wl.w, wl.h, wl.bpp, wl.n)
    r = measure(runner_cache[key], wl, frames, passes, 1)
    host = r["out"].cpu().numpy()
    pix = wl.n * wl.w * wl.h / 1e6
    te, td = [a for a, _ in r["samples"]], [b for _, b in r["samples"]]
    tc = [a + b for a, b in r["samples"]]
    res = {"config": label, "workload": wl.name, "frames": wl.n, "passes": passes, "statistic": "median of the timed passes (best in *_best)",
           "enc_MPix_s": round(pix / statistics.median(te), 1), "dec_MPix_s": round(pix / statistics.median(td), 1),
           "combined_MPix_s": round(pix / statistics.median(tc), 1),
           "enc_MPix_s_best": round(pix / min(te), 1), "dec_MPix_s_best": round(pix / min(td), 1), "combined_MPix_s_best": round(pix / min(tc), 1),
           "compressed_bytes": int(host.size), "sha256": stream_sha256(host), "lossless_roundtrip": True,
           "golden_stream_ok": golden_stream_check(golden_stream(stream), host, r["sizes"]) if stream else None, "golden_stream": stream,
           "stage_ms": {k: round(v, 2) for k, v in r["stage_ms"].items()}}
    if not no_cpu:
        nf = min(cpu_frames, wl.n)
        one, cpu = cpu_sample(wl, frames[:nf].cpu().numpy(), nf, gop, threads_all)
        res["cpu_baseline"] = dict(cpu["one_thread"], sample=f"first {nf} frames, oracle/libspo.so, one thread")
        if "all_cores" in cpu:
            res["cpu_baseline"]["all_cores"] = cpu["all_cores"]
        res["parity"] = {"vs": "oracle (CPU restatement; pinned to the reference only for rANS)", "frames_checked": nf, "ok": parity_of(host, r["sizes"], one, nf)}
    return res


def n8_same_stream_entry(dev, local_rank, w, h, n, headline_ms, headline_sha):
    """The default multi-GPU workload is WEAK scaling of the headline: rank r codes the 300 key frames of synthetic desktop seed 1 + r.
    This times the work of an N = 8 run - the eight streams, 2400 key frames - through the same Runner.step on ONE GPU, so that a
    later SCALE line has a same-stream denominator: (value at N = 8) / (this value) is the strong-scaling speed-up of the job the
    eight ranks do together.  (One GPU holds 1024 key-frame chains at once, one per SIMD: 2400 chains are two and a bit rounds, so
    the figure to expect is well below 8 x 300 frames' time.)"""
    import torch
    ranks = 8
    frames = torch.empty((ranks * n, h * w * 4), dtype=torch.uint8, device=dev)
    for r in range(ranks):
        frames[r * n:(r + 1) * n] = make_frames(w, h, 1 + r, 32, 0, n, dev)
    runner = Runner(dev, local_rank, w, h, 32, ranks * n)
    wl = Workload("8 x %d key frames" % n, w, h, 32, 1, 0, ranks * n, [0] * (ranks * n), "weak", ranks * n)
    m = measure(runner, wl, frames, 2, 1)
    host = m["out"].cpu().numpy()
    first = int(sum(int(x) for x in m["sizes"][:n]))
    pix = w * h * ranks * n / 1e6
    ms = m["elapsed"] / 2 * 1e3
    return {"config": "the N = 8 headline run's work on ONE GPU: 8 streams (synthetic desktop seeds 1..8) x %d key frames of %dx%d RGB32, one CompressBatch + one DecompressBatch" % (n, w, h),
            "frames": ranks * n, "passes": 2, "ms_per_step": round(ms, 2), "combined_MPix_s": round(pix / (ms * 1e-3), 1), "enc_MPix_s": round(pix / m["t_enc"], 1),
            "dec_MPix_s": round(pix / m["t_dec"], 1), "stage_ms": {k: round(v, 2) for k, v in m["stage_ms"].items()}, "lossless_roundtrip": True,
            "first_stream_is_the_headline_stream": stream_sha256(host[:first]) == headline_sha, "compressed_bytes": int(host.size),
            "time_vs_one_stream": round(ms / headline_ms, 2),
            "use": "same-stream denominator for the driver's N = 8 line: strong-scaling speed-up = (N = 8 value) / combined_MPix_s; time_vs_one_stream is how many "
                   "headline steps these 8 streams cost one GPU (8 would be no overlap at all)"}


def per_frame_api_ms(local_rank, frames_host, w, h, bpp):
    """the drop-in pair as a VfW host drives it (host pointers, one frame per call, PCIe included): median milliseconds of
    CompressFrame / DecompressFrame for key frames and P-frames (tools/perframe_latency.py)"""
    import statistics
    from screenpressor_amd.codec import ScreenCodec
    enc, dec = ScreenCodec(local_rank).Init(w, h, bpp), ScreenCodec(local_rank).Init(w, h, bpp)
    enc.CompressFrame(frames_host[0], 0)  # warm-up (allocations)
    rows, pk = {}, []
    for t, f in enumerate(frames_host):
        t0 = time.perf_counter()
        p, ft = enc.CompressFrame(f, 0 if t % 4 == 0 else 1)
        rows.setdefault("encode_key" if ft == 0 else "encode_p", []).append((time.perf_counter() - t0) * 1e3)
        pk.append((p, ft))
    dec.DecompressFrame(pk[0][0], 0)
    for p, ft in pk:
        t0 = time.perf_counter()
        r, _ = dec.DecompressFrame(p, ft)
        assert r == 1
        rows.setdefault("decode_key" if ft == 0 else "decode_p", []).append((time.perf_counter() - t0) * 1e3)
    out = {k: round(statistics.median(v), 2) for k, v in sorted(rows.items())}
    out["samples_ms"] = {k: [round(x, 2) for x in v] for k, v in sorted(rows.items())}
    return out


# ------------------------------------------------------------------------------------------------ a rank ---
def selftest_rank(args, rank, world):
    """launcher / sharding / gather without a codec (CPU, gloo): every rank makes the synthetic packets of its GOP range"""
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from screenpressor_amd.sharding import gather_packets_begin, gather_packets_end, shard_gops
    if os.environ.get("SCPR_SELFTEST_DIE_RANK") == str(rank):  # (tests: a rank that is gone before the first collective)
        sys.stderr.write("selftest: this rank gives up before the rendezvous\n")
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    total, k = args.frames or 24, args.gop or 4
    ft = [0 if t % k == 0 else 1 for t in range(total)]
    lo, hi = shard_gops(ft, world)[rank]
    pk = [bytes([(7 * t + j) & 255 for j in range(5 + t % 3)]) for t in range(lo, hi)]
    payload = np.frombuffer(b"".join(pk), dtype=np.uint8).copy()
    sizes = [len(p) for p in pk]
    if world > 1:  # (in two halves, as the timed step does it: the transfers run beside the work in between)
        h = gather_packets_begin(dist, rank, world, payload, sizes)
        local_sum = int(payload.sum())
        out_p, out_s = gather_packets_end(h)
        assert local_sum == int(payload.sum())
    else:
        out_p, out_s = payload, sizes
    if rank == 0:
        blob = bytes(np.asarray(out_p).tobytes()) if not hasattr(out_p, "numpy") else out_p.numpy().tobytes()
        want = b"".join(bytes([(7 * t + j) & 255 for j in range(5 + t % 3)]) for t in range(total))
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "frames": total, "ranges": shard_gops(ft, world), "gathered_ok": blob == want,
                          "sizes_ok": [int(s) for s in (out_s.tolist() if hasattr(out_s, "tolist") else out_s)] == [5 + t % 3 for t in range(total)]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


class Env:
    """what a rank runs on: one GPU + RCCL in the bench; CPU tensors + gloo + a stand-in codec in tests/test_sharding.py, which
    drives c4_leg() through the same collectives"""

    def __init__(self, rank, world, dev, torch, dist, make_runner, render, cdev=None):
        self.rank, self.world, self.dev, self.torch, self.dist, self.make_runner, self.render = rank, world, dev, torch, dist, make_runner, render
        self.cdev = cdev if cdev is not None else dev  # where tensors handed to collectives live (the GPU under RCCL; the CPU under gloo)

    def sync(self):
        if self.dev is not None and self.dev.type == "cuda":
            self.torch.cuda.synchronize(self.dev)

    def barrier(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def all_reduce(self, t, op):
        if self.world > 1:
            self.dist.all_reduce(t, op=op)

    def all_ok(self, ok):
        t = self.torch.tensor([1 if ok else 0], device=self.cdev, dtype=self.torch.int32)
        self.all_reduce(t, self.dist.ReduceOp.MIN)
        return int(t.item()) == 1


def c4_leg(env, wl4, golden_name="stream_4k_ip_k150_1200"):
    """BASELINE configs[3] on `env.world` ranks: ONE stream cut at key frames into contiguous GOP ranges (wl4 is this rank's),
    every shard seeded with what crosses key frames - fn and the motion-vector memory handed down the ranks - so that the
    gathered packets are the single stream's; one warm-up and one timed pass between barriers, the hand-over and the gather
    of the packets on rank 0 inside the step; rank 0 compares the gathered stream with the committed oracle hash.
    No collective sits inside a try: whatever happens on one rank, every rank reaches the same collectives in the same
    order (a rank that skipped one would leave the others waiting, and the headline line would be lost with it).
    Returns the entry for config.others (meaningful on rank 0)."""
    from screenpressor_amd.sharding import gather_packets
    torch, dist, rank, world, dev = env.torch, env.dist, env.rank, env.world, env.cdev
    ready, c4, f4, r4 = 1, None, None, None
    try:
        f4 = env.render(wl4)
        r4 = env.make_runner(wl4)
    except Exception as e:  # noqa: BLE001  (every rank still takes part in the collective below)
        ready, c4 = 0, {"config": wl4.name, "error": repr(e)}
    if not env.all_ok(ready == 1):
        return c4 or {"config": wl4.name, "error": "another rank could not set the workload up"}
    seed = shard_seeder(env, wl4, f4)
    gathered = {}

    def one_pass():
        """(result or None, error): reset + seed (collectives) outside the try, the codec calls inside"""
        r4.reset()
        th = time.perf_counter()
        if seed:
            seed(r4.enc)
        one_pass.handover_s = time.perf_counter() - th  # this rank's wait for the ranks before it + its own motion pre-pass
        err = seed.error if seed else None
        if err is None:
            try:
                out, sizes, _, dec, a, b, _ = r4.step(f4, wl4.ftypes, reset=False)
                assert r4.same(dec, f4), "the decoded frames differ from the input"
                return (out, sizes, a, b), None
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        return None, err

    _, err4 = one_pass()  # warm-up
    if env.all_ok(err4 is None):
        env.barrier()
        t0 = time.perf_counter()
        res, err4 = one_pass()
        good = env.all_ok(res is not None)
        if good:  # (all ranks, all fine: the gather is part of the step)
            if world > 1:
                gathered["p"], gathered["s"] = gather_packets(dist, rank, world, res[0], res[1], device=dev)
            else:
                gathered["p"], gathered["s"] = res[0], torch.as_tensor([int(x) for x in res[1]])
        env.barrier()
        t4 = torch.tensor([time.perf_counter() - t0, getattr(one_pass, "handover_s", 0.0)], device=dev, dtype=torch.float64)
        env.all_reduce(t4, dist.ReduceOp.MAX)
        handover_ms = float(t4[1].item()) * 1e3
        t4 = t4[:1]
        if good:
            pix4 = wl4.total_frames * wl4.w * wl4.h / 1e6
            c4 = {"config": f"configs[3]: ONE {wl4.w}x{wl4.h} stream, {wl4.total_frames} frames, GOP-sharded over the ranks (strong scaling), every shard seeded with the "
                            "motion-vector memory of the shards before it, packets gathered on rank 0 in the step",
                  "workload": wl4.name, "scaling": "strong", "frames_total": wl4.total_frames, "frames_rank0": wl4.n, "gops_rank0": sum(1 for f in wl4.ftypes if f == 0),
                  "value_MPix_s": round(pix4 / float(t4.item()), 2), "ms_per_step": round(float(t4.item()) * 1e3, 1),
                  "handover_ms": round(handover_ms, 1),
                  "handover_note": "inside ms_per_step: the serial chain of motion pre-passes that hands the vector memory down the ranks before anything is coded "
                                   "(max over ranks = what the last rank waited; about half an encode per rank before it, sharding.py)",
                  "enc_MPix_s_rank0": round(wl4.n * wl4.w * wl4.h / 1e6 / res[2], 1), "dec_MPix_s_rank0": round(wl4.n * wl4.w * wl4.h / 1e6 / res[3], 1),
                  "lossless_roundtrip": True,
                  "note": "a GOP is one serial chain (one wave): 8 GOPs run side by side on ONE GPU already, so this stream gains nothing from more GPUs - "
                          "GPU count pays when the stream has more GOPs than one GPU has chain slots (768 at 1080p), see DESIGN.md 7"}
            if rank == 0:
                host = gathered["p"].cpu().numpy()
                sizes = [int(x) for x in gathered["s"].cpu().tolist()]
                c4.update({"gathered_frames_rank0": len(sizes), "compressed_bytes": int(host.size), "sha256": stream_sha256(host), "golden_stream": golden_name,
                           "golden_stream_ok": golden_stream_check(golden_stream(golden_name), host, sizes),
                           "parity": {"vs": "sha256 of ONE oracle codec over the whole stream (tests/golden/manifest.json), whole or GOP by GOP", "sharded_equals_single_stream": None}})
                c4["parity"]["sharded_equals_single_stream"] = c4["golden_stream_ok"] if isinstance(c4["golden_stream_ok"], bool) else None
    if c4 is None:
        c4 = {"config": wl4.name, "error": err4 or "the pass failed on another rank"}
    return c4


def run_rank(args):
    rc_final = 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.selftest_launcher:
        return selftest_rank(args, rank, world)
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from screenpressor_amd.sharding import gather_packets_begin, gather_packets_end

    cdev = None
    if world > 1 and args.rehearse_one_gpu:
        # every rank on GPU 0 (RCCL refuses two ranks on one device), the collectives over gloo on CPU tensors: the real codecs,
        # sharding, hand-over and gather with more than one process, on a box that has one GPU
        local_rank = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cdev = torch.device("cpu")
    elif world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # the ranks RCCL actually has
    dev = torch.device("cuda", local_rank)
    wl = describe(args, rank, world)
    W, H, BPP, N = wl.w, wl.h, wl.bpp, wl.n
    gop = 1 if args.workload == "keys" else (args.gop or (50 if args.workload == "ip" else 150))
    env = Env(rank, world, dev, torch, dist, lambda q: Runner(dev, local_rank, q.w, q.h, q.bpp, q.n),
              lambda q: make_frames(q.w, q.h, q.seed, q.bpp, q.lo, q.hi, dev, world), cdev)
    cdev = env.cdev

    # synthetic input, resident in HBM before the timed region
    frames = env.render(wl)
    runner = env.make_runner(wl)
    seed = shard_seeder(env, wl, frames) if args.workload == "c4" else None  # (keys / ip: every rank has a stream of its own)
    barrier = env.barrier
    gathered = {}

    # the exchange step: compressed chunks + sizes to rank 0 in frame order (RCCL over xGMI), started when the encoder has
    # returned and waited for after the decode of the same step - the chunks travel while the decoder's chains run
    def exchange_begin(out, sizes):
        if world > 1:
            gathered["h"] = gather_packets_begin(dist, rank, world, out, sizes, device=cdev)

    def exchange_end():
        if world > 1:
            gathered["p"], gathered["s"] = gather_packets_end(gathered.pop("h"))

    m = measure(runner, wl, frames, args.steps, args.warmup, barrier, seed, (exchange_begin, exchange_end))
    if seed is not None and seed.error:
        raise RuntimeError("motion pre-pass: " + seed.error)
    tmax = torch.tensor([m["elapsed"]], device=cdev, dtype=torch.float64)
    env.all_reduce(tmax, dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_frames = wl.total_frames
    value = total_frames * W * H / 1e6 / (elapsed / args.steps)

    # N > 1: configs[3] as well (c4_leg); reported under config.others, never as `value`.
    c4 = None
    head_gathered = (int(gathered["s"].numel()), int(gathered["p"].numel())) if gathered.get("s") is not None else None
    if (world > 1 or args.c4_leg) and not args.no_others and args.workload == "keys" and (W, H, BPP) == (1920, 1080, 32) and not args.frames:
        import copy
        a4 = copy.copy(args)
        a4.workload, a4.width, a4.height, a4.frames, a4.gop = "c4", None, None, args.c4_frames, None
        wl4 = describe(a4, rank, world)
        headline_out = m["out"].clone() if rank == 0 else None
        del frames
        runner.packets = runner.decoded = None
        runner = None
        m["out"] = headline_out
        torch.cuda.empty_cache()
        c4 = c4_leg(env, wl4)
        frames = None

    if rank == 0:
        host = m["out"].cpu().numpy()
        comp_bytes = int(host.size)
        per_step = m["stage_ms"]
        dom = max(per_step, key=per_step.get)
        raw = N * H * wl.pitch
        # algorithmic bytes per SURVEY.md 8(d): encode I = raw + c, encode P = 2 raw + c; decode I = c + raw, decode P = c + 2 raw
        np_frames = sum(1 for f in m["ft"] if f)
        is_dec = dom in ("decode", "unpack")
        alg_bytes = raw + comp_bytes + np_frames * H * wl.pitch
        dom_ms = per_step[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel: PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, FETCH doubled for gfx950), recorded under profiles/; quoted only for the workload they were taken on
        # The PMC passes cannot run inside this process (rocprofv3 wraps the command), so the figure is READ from the last record - and
        # only when that record says it was taken on these very kernel sources (csrc_digest) and on this workload; otherwise null.
        traffic, traffic_src = None, None
        kernel_of = {"decode": "k_decode_gop_w", "rans": "k_rans", "colour_chain": "k_colour_chain_w", "pack": "k_pack32", "classify": "k_tiles"}
        try:
            import glob
            recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
            rec = json.load(open(recs[-1]))
            if rec.get("csrc_sha256") != csrc_digest():
                traffic_src = f"null: {os.path.relpath(recs[-1], ROOT)} (recorded {rec.get('recorded', 'undated')}) predates the last change to screenpressor_amd/csrc"
            elif args.workload == "keys" and N == 300 and (W, H, BPP) == (1920, 1080, 32):
                for kq in rec["kernels"]:  # (a kernel template has one row per instance: the timed launch is the big one)
                    if kernel_of.get(dom, "?") in kq["kernel"] and round(kq["hbm_bytes_per_launch"]) > (traffic or 0):
                        traffic = round(kq["hbm_bytes_per_launch"])
                        traffic_src = (f"{os.path.relpath(recs[-1], ROOT)}: rocprofv3 --pmc passes of this command, recorded {rec.get('recorded', 'undated')} on kernel sources "
                                       f"{rec['csrc_sha256'][:12]} (= this run's)")
        except Exception:
            traffic = None
        # The dominant kernel of the headline is the decoder: one wave per key frame, a serial chain of ~700 000 symbols.  Its
        # limit is how fast ONE wave issues dependent instructions, not a memory system, so that is what `bound` says; the HBM
        # fraction the contract asks for stays beside it (frac / achieved / peak) and cycles per symbol is the figure to watch.
        cyc = None
        if is_dec:
            try:
                nsym = int(runner.enc._L.scpr_debug_entries(runner.enc._h, None, 0)) if runner is not None else 0  # coder entries of the last compress call = symbols the decoder takes
                clock_hz = 2.4e9  # MI355X_MICROARCH.md: max clock 2400 MHz (a lone wave per SIMD on a cool card holds it: s_memtime stamps give 2.40-2.43)
                gops = sum(1 for f in m["ft"] if f == 0) or 1
                if nsym:
                    cyc = {"symbols_per_launch": nsym, "chains_per_launch": gops, "clock_GHz": round(clock_hz / 1e9, 3),
                           "cycles_per_symbol": round(dom_ms * 1e-3 * clock_hz / (nsym / gops), 1),
                           "note": "launch time x clock / symbols of one chain (the chains run side by side, one wave each); instructions per symbol: profiles/r*_pmc_decoder.json"}
            except Exception as e:  # noqa: BLE001
                cyc = {"error": repr(e)}
        roofline = {"bound": "issue-latency" if is_dec else "hbm", "kernel": kernel_of.get(dom, dom), "stage": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "launch_ms": round(dom_ms, 3),
                    "algorithmic_bytes_per_launch": alg_bytes, "chain": cyc,
                    "note": ("the dominant kernel is a serial per-GOP chain: bound by the issue rate and latencies of one wave (`chain`), not by bandwidth - "
                             "achieved / peak / frac are its HBM figures all the same; DESIGN.md 5" if is_dec else "see DESIGN.md for what bounds this stage")}
        parity = {"vs": "oracle (CPU restatement of the reference; pinned to the reference itself only for rANS, see DESIGN.md 1)", "ok": None,
                  "lossless_roundtrip": True, "sha256": stream_sha256(host), "compressed_bytes": comp_bytes}
        if (args.workload, W, H, wl.seed) == ("keys", 1920, 1080, 1) and N >= 3:  # the committed fixture holds the hashes of the first three packets
            try:
                man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["desktop_1080p_keys" if BPP == 32 else "desktop_1080p_rgb24"]
                offs = np.concatenate([[0], np.cumsum(m["sizes"].astype(np.int64))])
                parity["golden_fixture_ok"] = all(hashlib.sha256(host[offs[t]:offs[t + 1]].tobytes()).hexdigest() == man["frame_sha256"][t] for t in range(3))
            except Exception as e:  # noqa: BLE001
                parity["golden_fixture_ok"] = f"not checked: {e}"
        # the whole timed stream against the committed sha256 of ONE oracle codec over it (tests/golden/manifest.json)
        gname = None
        if (W, H, wl.seed, wl.lo) == (1920, 1080, 1, 0) and N == 300:
            gname = {"keys": "stream_1080p_keys_300", "ip": {50: "stream_1080p_ip_k50_300", 300: "stream_1080p_ip_onegop_300"}.get(gop)}.get(args.workload)
        elif args.workload == "c4" and (W, H, wl.seed, gop) == (3840, 2160, 1, 150):
            gname = "stream_4k_ip_k150_1200"
        if gname:
            if args.workload == "c4" and world > 1 and gathered.get("p") is not None:  # the sharded stream as rank 0 gathered it
                parity["golden_stream_ok"] = golden_stream_check(golden_stream(gname), gathered["p"].cpu().numpy(), [int(x) for x in gathered["s"].cpu().tolist()])
            else:
                parity["golden_stream_ok"] = golden_stream_check(golden_stream(gname), host, m["sizes"], wl.lo)
            parity["golden_stream"] = gname
        cpu = None
        if not args.no_cpu and world == 1 and frames is not None:  # the CPU baseline is a rank-0, one-GPU measurement
            big = W * H > 1920 * 1080 or args.workload != "keys"
            nf = min(args.cpu_frames or (40 if big else N), N)
            ncores = host_cores()  # the cores this process may run on, at most one GPU's share of the box
            one, res = cpu_sample(wl, frames[:nf].cpu().numpy(), nf, gop, ncores if args.workload == "keys" else 0)
            parity["ok"] = parity_of(host, m["sizes"], one, nf)
            parity["frames_checked"] = nf
            best = res.get("all_cores", res["one_thread"])
            if res["one_thread"]["value"] >= best["value"]:
                best = res["one_thread"]
            cpu = {"value": best["value"], "unit": "MPix/s", "cores": best["cores"], "kind": "port",
                   "sample": (f"the whole workload ({nf} frames)" if nf == N else f"first {nf} frames of the same workload") + ", encode+decode, oracle/libspo.so; "
                   "`value` is the faster of one thread and the all-cores two-stage shape", "host_cores": ncores, **res}
            assert parity["ok"], "the packets of the timed run differ from the oracle's"
        config = {"workload": wl.name, **({"rehearsal": f"{world} ranks on ONE GPU, collectives over gloo on CPU tensors: not a scaling measurement"} if args.rehearse_one_gpu and world > 1 else {}),
                  "frames_per_gpu": N, "frames_total": total_frames, "parallelism": f"GOP/frame-sharded x{world}, one process per GPU",
                  "compressed_bytes_rank0": comp_bytes,
                  "enc_MPix_s_rank0": round(N * W * H / 1e6 / m["t_enc"], 2), "dec_MPix_s_rank0": round(N * W * H / 1e6 / m["t_dec"], 2),
                  "stage_ms_per_step": {k: round(v, 3) for k, v in per_step.items()}}
        if world > 1 and head_gathered is not None:
            config["gathered_frames_rank0"], config["gathered_bytes_rank0"] = head_gathered
        if world == 1 and frames is not None and not args.no_host_boundary:
            # The boundary hands over HOST buffers (ScreenCodec::CompressFrame takes host pointers, screencap.cpp:1632; DecompressFrame
            # :1695).  config.host_boundary: the same pass through scpr_compress_batch_host / scpr_decompress_batch_host - frames in
            # pinned host memory, packets into host memory, pictures back into host memory - where the transfers run beside the kernels
            # (frames uploaded in sub-batches under the coding of the one before; every key frame's rows leave for the host from the
            # decoder's chains while they run).  Never `value`: that stays the HBM-resident step.
            try:
                h_in = frames.cpu().pin_memory()
                h_pk = torch.empty(max(2 * comp_bytes, 64 << 20), dtype=torch.uint8).pin_memory()
                h_out = torch.empty_like(h_in).pin_memory()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                d_in = h_in.to(dev, non_blocking=True)
                out, sizes, ft, dec, _, _, _ = runner.step(d_in, wl.ftypes)
                h_pk[:out.numel()].copy_(out, non_blocking=True)
                h_out.copy_(dec.reshape(N, -1), non_blocking=True)
                torch.cuda.synchronize(dev)
                seq_s = time.perf_counter() - t0
                del d_in
                rows = []
                for it in range(3):  # (the first pass allocates the staging buffers)
                    runner.reset()
                    h_out.zero_()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    pk, hsizes, hft = runner.enc.CompressBatchHost(h_in.reshape(-1), wl.ftypes, out=h_pk)
                    t1 = time.perf_counter()
                    _, se = runner.enc.last_timing()
                    r, got = runner.dec.DecompressBatchHost(pk, hsizes, hft, out=h_out.reshape(-1))
                    t2 = time.perf_counter()
                    assert r == N
                    rows.append((t1 - t0, t2 - t1))
                best = min(rows[1:], key=lambda q: q[0] + q[1])
                same_packets = stream_sha256(pk.numpy()) == parity["sha256"] and [int(x) for x in hsizes] == [int(x) for x in m["sizes"]]
                lossless = bool(torch.equal(h_out, h_in))
                config["host_boundary"] = {"MPix_s": round(N * W * H / 1e6 / (best[0] + best[1]), 1), "compress_ms": round(best[0] * 1e3, 2), "decompress_ms": round(best[1] * 1e3, 2),
                                           "passes_ms": [[round(a * 1e3, 1), round(b * 1e3, 1)] for a, b in rows], "statistic": "best of the two passes after the first",
                                           "packets_equal_the_device_path": same_packets, "lossless_roundtrip": lossless,
                                           "sequential_MPix_s": round(N * W * H / 1e6 / seq_s, 1),
                                           "method": "scpr_compress_batch_host + scpr_decompress_batch_host on pinned host tensors (frames, packets, pictures all in host memory), "
                                                     "wall clock around the two calls; sequential_MPix_s: upload, the device-resident step, download, one after the other"}
                config["incl_host_transfer_MPix_s"] = config["host_boundary"]["MPix_s"] if same_packets and lossless else "host path differs from the device path"
                del h_in, h_out, h_pk
            except Exception as e:  # noqa: BLE001
                config["incl_host_transfer_MPix_s"] = f"not measured: {e}"
        if world == 1 and frames is not None and (W, H) == (1920, 1080) and args.workload == "keys" and not args.no_host_boundary:
            try:
                config["per_frame_api_ms"] = dict(per_frame_api_ms(local_rank, [f.reshape(-1) for f in frames[:8].cpu().numpy()], W, H, BPP),
                                                  note="ScreenCodec::CompressFrame / DecompressFrame one frame per call, host pointers, PCIe included (median of 8 frames: 2 key, 6 P)")
            except Exception as e:  # noqa: BLE001
                config["per_frame_api_ms"] = f"not measured: {e}"
        if c4 is not None:
            config["others"] = [c4]
        if world == 1 and c4 is None and not args.no_others and args.workload == "keys" and (W, H, BPP) == (1920, 1080, 32) and not args.frames:
            others, cache = [], {(W, H, BPP): runner}
            cf = args.cpu_frames or 60

            def sub(label, w, h, bpp, n, ft, fr, k, cpu_n, stream=None, threads_all=0):
                owl = Workload(label, w, h, bpp, 1, 0, n, ft, "weak", n)
                try:
                    others.append(other_config(cache, dev, local_rank, label, owl, fr, k, cpu_n, args.no_cpu, stream, threads_all))
                except Exception as e:  # noqa: BLE001
                    others.append({"config": label, "error": repr(e)})
            sub("configs[2]: 1920x1080 RGB32 I+P, key frame every 50 (6 GOPs), 300 frames", W, H, 32, N, [0 if t % 50 == 0 else 1 for t in range(N)], frames, 50, cf, "stream_1080p_ip_k50_300")
            sub("configs[2] as ONE GOP: key frame 0 then 299 P-frames (the reference's default key interval is 500, conf.h:7)", W, H, 32, N, [0] + [1] * (N - 1), frames, N, cf, "stream_1080p_ip_onegop_300")
            f24 = make_frames(W, H, 1, 24, 0, N, dev)
            sub("configs[4]: 1920x1080 RGB24 (3-byte pixels, pitch 5760) key-frame-only, 300 frames", W, H, 24, N, [0] * N, f24, 1, cf, "stream_1080p_keys_300")
            others[-1]["stream_equals_rgb32_stream"] = others[-1].get("sha256") == parity["sha256"]
            del f24
            del frames
            runner.packets = runner.decoded = None
            cache.clear()
            torch.cuda.empty_cache()
            f4k = make_frames(3840, 2160, 1, 32, 0, 150, dev)
            sub("configs[3], one GPU's share of 8: 3840x2160 RGB32, 150 frames as key frames (a frame-sharded stream)", 3840, 2160, 32, 150, [0] * 150, f4k, 1, 10, "stream_4k_keys_150", host_cores())
            sub("configs[3], one GPU's share of 8: 3840x2160 RGB32, ONE GOP of 150 frames (key frame every 150)", 3840, 2160, 32, 150, [0] + [1] * 149, f4k, 150, 10, "stream_4k_ip_k150_1200", host_cores())
            del f4k
            cache.clear()
            torch.cuda.empty_cache()
            if not args.no_n8:
                try:
                    others.append(n8_same_stream_entry(dev, local_rank, W, H, N, ms_per_step, parity["sha256"]))
                except Exception as e:  # noqa: BLE001
                    others.append({"config": "the N = 8 headline run's streams on ONE GPU", "error": repr(e)})
            config["others"] = others
        # north_star's target - at least 10x the host-CPU encoder on 4K RGB32 at one GPU, bit-identical output - as ONE field, from the
        # 4K entries above (the CPU side is the oracle port, kind "port": the reference itself cannot be built here)
        try:
            t4k = {}
            for o in config.get("others", []):
                if "3840x2160" in o.get("config", "") and "cpu_baseline" in o and "enc_MPix_s" in o:
                    cb = o["cpu_baseline"]
                    allc = cb.get("all_cores", {})
                    which = "key_frames" if "as key frames" in o["config"] else "one_gop"
                    t4k[which] = {"gpu_enc_MPix_s": o["enc_MPix_s"], "cpu_enc_MPix_s_one_thread": cb.get("enc_MPix_s"), "cpu_enc_MPix_s_all_cores": allc.get("enc_MPix_s"),
                                  "cores": allc.get("cores", cb.get("cores")),
                                  "enc_4k_vs_cpu_all_cores": round(o["enc_MPix_s"] / max(allc.get("enc_MPix_s") or 0, cb.get("enc_MPix_s") or 0, 1e-9), 1),
                                  "bit_identical": bool(o.get("golden_stream_ok") is True and o.get("parity", {}).get("ok", False))}
            if t4k:
                config["targets"] = {"north_star": ">= 10x the host-CPU encoder on 4K RGB32 at 1 GPU, bit-identical", "cpu_kind": "port (oracle/libspo.so, the faster of one thread and all cores)",
                                     **t4k, "met": all(v["enc_4k_vs_cpu_all_cores"] >= 10 and v["bit_identical"] for v in t4k.values()),
                                     "eight_gpu_target": "unmeasured here: the driver's N = 8 run"}
        except Exception as e:  # noqa: BLE001
            config["targets"] = f"not computed: {e}"
        line = {"metric": METRIC, "value": round(value, 2), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": config,
                "roofline": roofline, "cpu_baseline": cpu, "parity": parity}
        print(json.dumps(line), flush=True)
        # a timed stream that is not the committed stream is not a measurement of this codec: the line says so AND the run fails
        bad = [q.get("config", "?") for q in [dict(config=wl.name, **parity)] + [o for o in config.get("others", []) if isinstance(o, dict)] if q.get("golden_stream_ok") is False]
        if bad:
            sys.stderr.write("[bench] golden_stream_ok is false for: " + "; ".join(bad) + "\n")
            rc_final = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc_final


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)  # (before anything imports torch or touches the GPU)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
