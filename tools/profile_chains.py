"""Design tool: the slowest colour chains of an encode (profile build): length, cycles, cycles per symbol, final kind, which path
the symbols took.  usage: profile_chains.py [keys4k|gop1080|keys1080|gop4k] [frames]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "screenpressor_amd", "libscpr_amd_prof.so")


def main():
    import numpy as np
    import torch
    import bench as B
    from screenpressor_amd import codec as K
    K._LIB_PATH = LIB
    which = sys.argv[1] if len(sys.argv) > 1 else "keys4k"
    w, h, n, k = {"keys1080": (1920, 1080, 300, 1), "gop1080": (1920, 1080, 300, 300), "keys4k": (3840, 2160, 150, 1), "gop4k": (3840, 2160, 150, 150)}[which]
    if len(sys.argv) > 2:
        n = int(sys.argv[2])
    dev = torch.device("cuda", 0)
    f = B.make_frames(w, h, 1, 32, 0, n, dev)
    c = K.ScreenCodec(0).Init(w, h, 32)
    L = K.load_library()
    out = torch.empty(max(256 << 20, n * w * h // 2), dtype=torch.uint8, device=dev)
    buf = (C.c_uint32 * (8192 * 8))()
    for _ in range(2):
        c.Deinit(); c.Init(w, h, 32)
        L.scpr_debug_chains(buf, 0)
        c.CompressBatch(f, [0 if t % k == 0 else 1 for t in range(n)], out=out)
        st = c.last_timing()[1]
    nrec = L.scpr_debug_chains(buf, 8192)
    r = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 8)[:min(nrec, 8192)]
    print(which, n, "frames; colour_chain %.2f ms; chains >= 2048 symbols: %d" % (st["colour_chain"], nrec))
    order = np.argsort(-r[:, 1].astype(np.int64))
    print("   len     cycles  cyc/sym kind   d   slow      par      raw  dser   batch  ctx(plane,cx) gen")
    for i in order[:25]:
        ln, cy, kd, slow, par, raw, dser, q = [int(x) for x in r[i]]
        print("%7d %10d %7.1f %4d %4d %7d %8d %7d %5d %7d  %d,%d  %d" % (ln, cy, cy / ln, kd & 255, kd >> 8, slow, par, raw, dser & 255, dser >> 8, (q % 12288) // 4096, q % 4096, q // 12288))
    print("sum of cycles of all long chains: %.1f M; longest %.2f M cycles = %.2f ms at 2.4 GHz" % (r[:, 1].sum() / 1e6, r[order[0], 1] / 1e6, r[order[0], 1] / 2.4e6))


if __name__ == "__main__":
    if "--build" in sys.argv or not os.path.exists(LIB):
        src = os.path.join(ROOT, "screenpressor_amd", "csrc", "scpr_amd.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-fPIC", "-shared", "-DSCPR_PROFILE", "-mllvm", "-align-all-nofallthru-blocks=6",
                               "-mllvm", "-enable-post-misched=false", "-Wno-unused-result", "-o", LIB, src, os.path.join(ROOT, "screenpressor_amd", "csrc", "scpr_driver.cpp"),
                               os.path.join(ROOT, "screenpressor_amd", "csrc", "scpr_avi.cpp")])
        if "--build" in sys.argv:
            sys.exit(0)
    main()
