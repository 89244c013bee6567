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
for sub in ("ip", "4k"):  # the I+P (configs[2]) and 4K (configs[3] share) runs
    st2 = find("stats_" + sub, "*kernel_stats.csv")
    if st2:
        rows = list(csv.DictReader(open(st2)))
        with open(os.path.join(dst, f"{tag}_{sub}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
    lg = os.path.join(src, f"bench_stats_{sub}.log")
    if os.path.exists(lg):
        line = [l for l in open(lg).read().splitlines() if l.startswith("{")]
        if line:
            open(os.path.join(dst, f"{tag}_{sub}_bench_line.json"), "w").write(line[-1] + "\n")

# 2. HBM traffic per kernel launch: the LARGEST launch of each kernel - the 300-frame batch of the timed step (the same command
# also makes a pass with host transfers and a handful of one-frame calls for the per-frame latencies: small launches that an
# average over dispatches would mix in)
def pmc(d):
    f = find(d, "*counter_collection.csv")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    if f:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: max(v.values()) for k, v in per.items()}, {k: len(v) for k, v in per.items()}
fa, fc = pmc("pmc_FETCH_SIZE")
wa, wc = pmc("pmc_WRITE_SIZE")
ker = []
for k in fa:
    if not k.startswith("scpr::") and "rocprim" not in k:
        continue
    fkb, wkb = fa[k], wa.get(k, 0.0)
    ker.append({"kernel": k, "launches_seen": fc[k], "FETCH_SIZE_KB_largest_launch": fkb, "WRITE_SIZE_KB_largest_launch": wkb, "hbm_bytes_per_launch": (2 * fkb + wkb) * 1024})
prov = open(os.path.join(src, "csrc_sha256.txt")).read().split() if os.path.exists(os.path.join(src, "csrc_sha256.txt")) else [None, None]
json.dump({"command": "bench.py --no-others --no-host-boundary --steps 1 --warmup 0 --no-cpu (configs[1]: 300 x 1920x1080 key frames, the HBM-resident step only)",
           "csrc_sha256": prov[0], "recorded": prov[1],
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE; counters are KB; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is; per kernel the largest dispatch (the 300-frame launch of the timed step)",
           "kernels": ker}, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

# 3. decoder instruction mix and waits (tools/decoder_pmc2.sh)
summ = os.path.join(src, "dec", "summary.json")
if os.path.exists(summ):
    d = json.load(open(summ))
    d["note"] = ("SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles; ACTIVE_INST_ANY == INSTS: every instruction is 'active' for one unit, "
                 "WAIT_ANY is everything else of the wave's life - for this lone wave the fifth cycle of every instruction (tools/fetch_bench.hip: 5.0 cycles per "
                 "instruction, dependent or not), LDS round trips, values crossing between the vector and the scalar unit, branches: see DESIGN.md 3")
    json.dump(d, open(os.path.join(dst, f"{tag}_pmc_decoder.json"), "w"), indent=1)
print("written", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
