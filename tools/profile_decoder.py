"""Design tool: where one decoder wave spends its time (s_memtime ticks per section).

Builds screenpressor_amd/libscpr_amd_prof.so (-DSCPR_PROFILE), decodes N synthetic 1080p key
frames and prints the share of each section of decode_intra_frame.  Not part of the product.
"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("SCPR_PROF_LIB", os.path.join(ROOT, "screenpressor_amd", "libscpr_amd_prof.so"))  # (the override: a profile build of an experiment)
src = os.path.join(ROOT, "screenpressor_amd", "csrc", "scpr_amd.hip")
if "--build" in sys.argv or not os.path.exists(LIB):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-fPIC", "-shared", "-DSCPR_PROFILE", "-mllvm", "-align-all-nofallthru-blocks=6", "-mllvm", "-enable-post-misched=false", "-mllvm", "-structurizecfg-skip-uniform-regions=true",
                           "-Wno-unused-result", "-o", LIB, src])
    if "--build" in sys.argv:
        sys.exit(0)
import torch
from screenpressor_amd import codec as K
from screenpressor_amd.synth import DesktopSequence
K._LIB_PATH = LIB
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
skip = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0  # --ip: P-frames decoded before the profiled ones (an aged GOP)
W, H = 1920, 1080
seq = DesktopSequence(W, H, seed=1)
frames = torch.from_numpy(seq.frames(n)).cuda().reshape(n, -1)
c = K.ScreenCodec()
c.Init(W, H, 32)
IP = "--ip" in sys.argv  # one GOP: a key frame and n-1 P-frames (sections 5-7 of decode_inter_frame)
pk, sizes, ft = c.CompressBatch(frames, [0] + [1] * (n - 1) if IP else [0] * n)
L = K.load_library()
out = (C.c_ulonglong * 24)()
cp = (C.c_ulonglong * 32)()
if IP and skip:  # age the GOP: the first skip + 1 frames are decoded before the counters are cleared
    off = int(np.sum(sizes[:skip + 1]))
    d = K.ScreenCodec()
    d.Init(W, H, 32)
    r, dec0 = d.DecompressBatch(pk[:off], sizes[:skip + 1], ft[:skip + 1])
    L.scpr_debug_profile(out)
    L.scpr_debug_cprof(cp)
    r, dec = d.DecompressBatch(pk[off:].clone(), sizes[skip + 1:], ft[skip + 1:])
    torch.cuda.synchronize()
    assert torch.equal(dec.reshape(-1), frames[skip + 1:].reshape(-1))
    n = n - skip - 1
else:
    L.scpr_debug_profile(out)
    L.scpr_debug_cprof(cp)
    r, dec = c.DecompressBatch(pk, sizes, ft)
    torch.cuda.synchronize()
    assert torch.equal(dec.reshape(-1), frames.reshape(-1))
L.scpr_debug_profile(out)
ev = np.array(list(out)[8:16], dtype=np.float64)
ex = np.array(list(out)[16:], dtype=np.float64)
v = np.array(list(out)[:8], dtype=np.float64)
names = ["P", "colour", "N", "fill (key frames) / wait for motion copies and own stores (P-frames)", "rows/loop", "P-frame: plane copy, header, block types", "P-frame: rect header symbols", "P-frame: runs"]
tot = v.sum()
print("ticks per frame: %.0f" % (tot / n))
for nm, x in zip(names, v):
    if x:
        print("%-50s %5.1f %%   %.0f ticks/frame" % (nm, 100 * x / tot, x / n))
for nm, x in zip(["colour symbols", "record cache misses", "dense-table symbols", "raw symbols", "small-table slow path", "runs", "literal runs", "dense-table cache misses"], ev):
    print("%-50s %.0f per frame" % (nm, x / n))
print("dense fast-path hits per frame: %.0f; small-table hits on the top entry: %.0f, in one-symbol tables: %.0f" % (ex[0] / n, ex[1] / n, ex[2] / n))
if True:
    L.scpr_debug_cprof(cp)
    cpv = list(cp)
    cls_names = ["top entry of a small table", "small table, any other way (whole symbol)", "[of which: small_op alone, the general path]", "dense table (whole symbol but the tail)", "[of which: dense_op alone, the general path]",
                 "raw byte, kinds 0-3 (but the tail)", "[record-cache miss alone]", "[dense-table cache miss alone]"]
    print("colour symbols by class (ticks include the record fetch; a miss is also inside its symbol's class):")
    classes = {}
    for k, nm in enumerate(cls_names):
        t, cnt = cpv[2 * k], cpv[2 * k + 1]
        if cnt:
            print("  %-48s %9.0f per frame x %7.0f ticks = %10.0f ticks/frame" % (nm, cnt / n, t / cnt, t / n))
            classes[nm] = {"per_frame": cnt / n, "ticks_each": t / cnt, "ticks_per_frame": t / n}
    if os.environ.get("SCPR_PROFILE_JSON"):
        import json
        json.dump({"argv": sys.argv[1:], "frames_profiled": n, "ticks_per_frame": tot / n,
                   "sections_ticks_per_frame": {nm: x / n for nm, x in zip(names, v) if x},
                   "events_per_frame": {nm: x / n for nm, x in zip(["colour symbols", "record cache misses", "dense-table symbols", "raw symbols", "small-table slow path", "runs", "literal runs", "dense-table cache misses"], ev)},
                   "extra_ticks_per_frame": {"rect write-back": ex[3] / n, "motion blocks": ex[4] / n, "run length end to end of single-row fill": ex[5] / n, "general fills": ex[6] / n},
                   "colour_classes": classes}, open(os.environ["SCPR_PROFILE_JSON"], "w"), indent=1)
if not IP:
    rp = [x / n for x in cpv[16:24]]
    print("key-frame runs by pixel type, per frame: literal %.0f, previous pixel %.0f, above %.0f, gradient %.0f, above-left %.0f; longer than 64 pixels: %.0f; %.1f pixels per run" % (rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7] / max(rp[1] + rp[2] + rp[3] + rp[4] + rp[5], 1)))
if IP:
    rp = [x / n for x in cpv[16:24]]
    print("P-frame runs through the general fills, per frame: %.0f (literal %.0f, left %.0f, above %.0f, previous frame %.0f, above-left / gradient %.0f; longer than 64: %.0f; %.1f pixels each)" % (rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7] / max(rp[0], 1)))
print("P-frame, more sections (ticks/frame): rect write-back %.0f, motion blocks (symbols, hand-over) %.0f; after the run length to the end of the single-row fill %.0f, general fills %.0f ('runs' above is then the loop between runs)" % (ex[3] / n, ex[4] / n, ex[5] / n, ex[6] / n))
