#!/bin/bash
# Runs ON THE GPU BOX: kernel + memory-copy trace of the host-pointer compress call with sub-batches of $1 frames (default 100)
set -e
R=$PWD
OUT=$R/gpurun_out/r4/trace_host
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o run --output-format csv -- python3 $R/tools/exp_host.py 300 ${1:-100} > $OUT/log.txt 2>&1
cd $R
python3 - <<PY
import csv, glob
k = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:]))
for r in csv.DictReader(open(m)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
ev.sort()
# the last compress call: find the last three big H2D copies
big = [i for i, e in enumerate(ev) if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2] and int(e[2].split()[-1] or 0) > 100e6]
print("big H2D copies:", len(big))
i0 = big[-3] if len(big) >= 3 else 0
t0 = ev[i0][0]
with open("$OUT/timeline.txt", "w") as f:
    for s, e, n in ev[i0:]:
        if (e - s) > 50000 or n.startswith("C"):
            f.write("%9.3f ms  +%8.3f ms  %s\n" % ((s - t0) / 1e6, (e - s) / 1e6, n))
print(open("$OUT/timeline.txt").read()[:6000])
PY
