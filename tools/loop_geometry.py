#!/usr/bin/env python3
"""Design tool (no GPU needed): where the compiler put the key-frame decoder's fast run loop.

Compiles the device code with the product's flags, disassembles k_decode_gop_w<false> and reports the back edge of the
fast-run loop (found by the block-counter test in front of it): the loop's length in bytes and where its first
instruction sits in a 64-byte line.  The kernel is one function of ~19 000 instructions and a change anywhere in it moves
register allocation and block placement everywhere: the same run loop has been seen 1260 bytes long (its rare ways out of
line: 114.5 ms per 1080p key frame) and 4176 bytes long (rare ways inside the loop, the common path jumping over them:
120.3 ms) from sources that mean the same.  A look at this before a GPU run says which of the two a build is.

    python tools/loop_geometry.py [extra compiler flags]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-fPIC", "-Wno-unused-result", "-mllvm", "-align-all-nofallthru-blocks=6",
         "-mllvm", "-enable-post-misched=false", "-mllvm", "-structurizecfg-skip-uniform-regions=true"]


def main():
    src = os.path.join(ROOT, "screenpressor_amd", "csrc", "scpr_amd.hip")
    with tempfile.TemporaryDirectory() as d:
        o, dev = os.path.join(d, "a.o"), os.path.join(d, "dev.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["--cuda-device-only", "-c", "-o", o, src], stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + o, "--output=" + dev])
        dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", dev], text=True).split("\n")
    for kern in ("k_decode_gop_wILb0", "k_decode_gop_wILb1"):
        start = [i for i, l in enumerate(dis) if kern in l and l.rstrip().endswith(">:")][0]
        ins = []
        for l in dis[start + 1:]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*// ([0-9A-F]{12}):", l)
            if not m:
                if l.startswith("0000"):
                    break
                continue
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        print(f"{kern}: {len(ins)} instructions, {ins[-1][0] - ins[0][0]} bytes")
        for i, (a, op, args) in enumerate(ins):
            if op == "s_add_i32" and re.search(r"0xfffe000[0-9a-f]$", args):  # ndec - (kBlockEntries - k): the fast loops' block-counter test
                for aa, oo, ar in ins[i:i + 8]:
                    if oo.startswith("s_cbranch"):
                        off = int(ar.split()[0])
                        off = off - 65536 if off >= 32768 else off
                        tg = aa + 4 + 4 * off
                        if off < 0:
                            print(f"   fast-run loop: back edge at {aa:#x} -> {tg:#x}: {aa + 4 - tg} bytes long, head at offset {tg % 64} of its 64-byte line")
                        break


if __name__ == "__main__":
    main()
