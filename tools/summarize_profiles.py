"""Turns gpurun_out/prof/<tag>/ (tools/collect_profiles.sh) into the committed profiles/<tag>_* files."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r1e"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof", tag)
dst = os.path.join(ROOT, "profiles")

def find(d, pat):
    r = glob.glob(os.path.join(src, d, "**", pat), recursive=True)
    return r[0] if r else None

# 1. kernel stats
st = find("stats", "*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)
line = [l for l in open(os.path.join(src, "bench_stats.log")).read().splitlines() if l.startswith("{")]
if line:
    open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line[-1] + "\n")

# 2. HBM traffic per kernel launch
def pmc(d):
    f = find(d, "*counter_collection.csv")
    acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
    if f:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    return acc, cnt
fa, fc = pmc("pmc_FETCH_SIZE")
wa, wc = pmc("pmc_WRITE_SIZE")
ker = []
for k in fa:
    if not k.startswith("scpr::") and "rocprim" not in k:
        continue
    n = max(len(fc[k]), 1)
    fkb, wkb = fa[k] / n, wa.get(k, 0.0) / max(len(wc.get(k, [1])), 1)
    ker.append({"kernel": k, "launches": n, "FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb, "hbm_bytes_per_launch": (2 * fkb + wkb) * 1024})
json.dump({"command": "bench.py --frames 300 --steps 1 --warmup 0 --no-cpu (1920x1080 key frames)",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE; counters are KB; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is",
           "kernels": ker}, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

# 3. decoder instruction mix
tot = collections.defaultdict(float)
for d in glob.glob(os.path.join(src, "dec_*")):
    if os.path.isdir(d):
        f = find(os.path.basename(d), "*counter_collection.csv")
        if f:
            for r in csv.DictReader(open(f)):
                if "decode_gop" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
nsym = 16 * 700818  # tools/decode_only.py 16: coder entries of the 16 synthetic key frames (oracle tap)
json.dump({"command": "tools/decode_only.py 16 (16 x 1080p key frames), kernel k_decode_gop_w<false>", "symbols": nsym,
           "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles",
           "per_symbol": {k: v / nsym for k, v in sorted(tot.items())}}, open(os.path.join(dst, f"{tag}_pmc_decoder.json"), "w"), indent=1)
print("written", sorted(os.listdir(dst))[-6:])
