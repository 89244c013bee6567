#!/usr/bin/env python3
"""bench.py — MPix/s encode+decode on the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W

Workload (config.workload): 1920x1080 RGB32, every frame a key frame, 300 frames per
GPU (BASELINE.json configs[1]).  One step = one pass of the hot path over the batch:
compress the 300 frames (resident in HBM) to packets in HBM, then decompress them back
to RGB32 in HBM.  value = pixels of all ranks / max-over-ranks step time, i.e. the
combined figure W*H*N/(t_enc+t_dec) of SURVEY.md §8d.  Multi-GPU: frames are sharded
across ranks (weak scaling, one process per GPU); every step ends with the gather of the
compressed chunks to rank 0 over RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-frames", type=int, default=300, help="frames of the same workload timed on the host CPU (oracle): the default is the whole 300-frame workload, ~8 s of one core")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=["keys", "ip"], default="keys",
                    help="keys: every frame a key frame (BASELINE configs[1], the headline); ip: key frame every --gop frames (configs[2])")
    ap.add_argument("--gop", type=int, default=50)
    ap.add_argument("--bpp", type=int, choices=[32, 24], default=32, help="24: packed 3-byte pixels, rows padded to 4 bytes (BASELINE configs[4])")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from screenpressor_amd.codec import ScreenCodec
    from screenpressor_amd.synth import DesktopSequence, pack24

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    W, H, N = args.width, args.height, args.frames

    # synthetic input, resident in HBM before the timed region (rank r: its own seed/shard)
    seq = DesktopSequence(W, H, seed=1 + rank)
    BPP = args.bpp
    pitch = W * 4 if BPP == 32 else (W * 3 + 3) & ~3
    frames = torch.empty((N, H * pitch), dtype=torch.uint8, device=dev)
    for t in range(N):
        f = seq.frame(t) if BPP == 32 else pack24(seq.frame24(t))
        frames[t] = torch.from_numpy(np.ascontiguousarray(f).reshape(-1)).to(dev)
    codec_e = ScreenCodec(local_rank).Init(W, H, BPP)
    codec_d = ScreenCodec(local_rank).Init(W, H, BPP)
    packets = torch.empty(max(256 << 20, N * W * H // 2), dtype=torch.uint8, device=dev)
    decoded = torch.empty(N * H * pitch, dtype=torch.uint8, device=dev)
    ftypes = [0] * N if args.workload == "keys" else [0 if t % args.gop == 0 else 1 for t in range(N)]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    stage_acc, t_enc_acc, t_dec_acc, comp_bytes = {}, 0.0, 0.0, 0

    def step(timed):
        nonlocal t_enc_acc, t_dec_acc, comp_bytes
        codec_e.Deinit(); codec_e.Init(W, H, BPP)
        codec_d.Deinit(); codec_d.Init(W, H, BPP)
        t0 = time.perf_counter()
        out, sizes, ft = codec_e.CompressBatch(frames, ftypes, out=packets)
        t1 = time.perf_counter()
        te, se = codec_e.last_timing()
        r, dec = codec_d.DecompressBatch(out, sizes, ft, out=decoded)
        t2 = time.perf_counter()
        td, sd = codec_d.last_timing()
        assert r == N
        if world > 1:  # exchange step: compressed chunks + sizes to rank 0, frame order (RCCL over xGMI)
            from screenpressor_amd.sharding import gather_packets
            gather_packets(dist, rank, world, out, sizes, device=dev)
        if timed:
            t_enc_acc += t1 - t0
            t_dec_acc += t2 - t1
            comp_bytes = int(out.numel())
            for k, v in list(se.items()) + list(sd.items()):
                if v > 0:
                    stage_acc[k] = stage_acc.get(k, 0.0) + v
        return out, sizes, dec

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, sizes, dec = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    lossless = bool(torch.equal(dec.reshape(N, -1), frames))

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * N * W * H / 1e6 / (elapsed / args.steps)

    if rank == 0:
        # dominant kernel stage by device time (HIP events on the codec's stream, scpr_last_timing)
        per_step = {k: v / args.steps for k, v in stage_acc.items()}
        dom = max(per_step, key=per_step.get)
        raw = N * H * pitch
        # algorithmic bytes per SURVEY.md §8(d): encode I = raw + c, decode I = c + raw, per frame
        alg_bytes = raw + comp_bytes if args.workload == "keys" else 2 * raw + comp_bytes  # P-frames also read the previous frame
        dom_ms = per_step[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the PMC passes recorded under profiles/ (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs of this same command, FETCH doubled for gfx950); only
        # quoted when the recorded run used this workload
        traffic = None
        kernel_of = {"decode": "k_decode_gop_w", "rans": "k_rans", "colour_chain": "k_colour_chain_w", "pack": "k_pack32"}
        try:
            import glob
            rec = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))[-1]))  # the latest recorded pass
            if args.workload == "keys" and N == 300 and (W, H) == (1920, 1080):
                for kq in rec["kernels"]:
                    if kernel_of.get(dom, "?") in kq["kernel"]:
                        traffic = round(kq["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "launch_ms": round(dom_ms, 3),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": "the dominant kernel is a serial per-GOP chain (issue-bound), not bandwidth-shaped; see DESIGN.md §5"}
        cpu = None
        if not args.no_cpu and world == 1:  # the CPU baseline is a rank-0, one-GPU measurement
            import oracle_api as O
            nf = min(args.cpu_frames, N)
            sample = np.stack([np.ascontiguousarray(seq.frame(t) if BPP == 32 else pack24(seq.frame24(t))).reshape(-1) for t in range(nf)])
            r = O.time_stream(sample, W, H, BPP, key_interval=1 if args.workload == "keys" else args.gop)
            assert r["bad"] == 0
            cpu = {"value": round(nf * W * H / 1e6 / (r["t_enc"] + r["t_dec"]), 2), "unit": "MPix/s", "cores": 1, "kind": "port",
                   "sample": (f"the whole workload ({nf} frames)" if nf == N else f"first {nf} frames of the same workload") + ", encode+decode, oracle/libspo.so (single thread)",
                   "enc_MPix_s": round(nf * W * H / 1e6 / r["t_enc"], 2), "dec_MPix_s": round(nf * W * H / 1e6 / r["t_dec"], 2)}
        line = {
            "metric": "MPix/s encode+decode, 1080p RGB32; bitstream byte-identical to ref", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"{W}x{H} RGB{BPP} key-frame-only (I-frames), {N} frames per GPU, synthetic desktop seed 1+rank" if args.workload == "keys"
                                    else f"{W}x{H} RGB{BPP} I+P (key frame every {args.gop}), {N} frames per GPU, synthetic desktop seed 1+rank"),
                       "frames_per_gpu": N, "parallelism": f"frame-sharded x{world}", "lossless_roundtrip": lossless,
                       "compressed_bytes_per_gpu": comp_bytes,
                       "enc_MPix_s_rank0": round(N * W * H / 1e6 / (t_enc_acc / args.steps), 2),
                       "dec_MPix_s_rank0": round(N * W * H / 1e6 / (t_dec_acc / args.steps), 2),
                       "stage_ms_per_step": {k: round(v, 3) for k, v in per_step.items()}},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
