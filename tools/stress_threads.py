"""Design tool (GPU box): two codecs compressing at the same time from two host threads (their kernels share CUs, LDS and the
scalar data cache that k_rans_s invalidates per trip) - every packet against the oracle.  `python tools/stress_threads.py [rounds]`."""
import sys, os, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle_api as O
from screenpressor_amd.codec import ScreenCodec
from screenpressor_amd.synth import DesktopSequence


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda", 0)
    jobs = []
    for k, (w, h, n, key_every, noise) in enumerate([(640, 360, 24, 6, 0.0), (480, 270, 30, 30, 0.3), (800, 450, 12, 1, 0.1), (320, 240, 40, 8, 0.6)]):
        seq = DesktopSequence(w, h, seed=40 + k, noise_fraction=noise)
        frames = np.stack([seq.frame(t) for t in range(n)])
        ft = [0 if t % key_every == 0 else 1 for t in range(n)]
        ora = O.OracleCodec(w, h, 32)
        want = [ora.compress(f, key=q == 0)[0] for f, q in zip(frames, ft)]
        jobs.append((w, h, torch.from_numpy(frames).to(dev), ft, want))
    bad = []

    def worker(idx):
        for r in range(rounds):
            w, h, fr, ft, want = jobs[(idx + 2 * r) % len(jobs)]
            c = ScreenCodec(0).Init(w, h, 32)
            pk, sizes, _ = c.CompressBatch(fr, ft, sync=False)
            pk = pk.cpu().numpy().tobytes()
            o = 0
            for i, sz in enumerate(sizes):
                if pk[o:o + int(sz)] != want[i]:
                    bad.append((idx, r, i))
                    break
                o += int(sz)
            c.Deinit()

    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    print("%d rounds x 2 threads, %d bad %s, %.0f s" % (rounds, len(bad), bad[:3], time.time() - t0))
    print("ALL OK" if not bad else "FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
