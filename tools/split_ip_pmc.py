#!/usr/bin/env python3
"""Split the decoder's PMC counters of an I+P GOP (tools/decoder_pmc2.sh N <dir> --ip) into key-frame and P-frame
per-symbol figures, using the key-frames-only collection (profiles/<tag>_pmc_decoder.json) for the key frame's share:

    P-frame figure = (stream total - key-frame symbols x key-frame per-symbol figure) / P-frame symbols

usage: tools/split_ip_pmc.py <tag> [--p-ms MS_PER_P_FRAME]      (reads gpurun_out/<tag>_decpmc_ip/summary.json,
                                                                 writes profiles/<tag>_pmc_decoder_ip.json)
--p-ms: the wall-clock time of one P-frame WITH helper waves (tools/exp_dec_gop.py), for the production cycles per symbol.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ = 2.43  # the chain's wave runs at the card's sustained clock under this load (MI355X_MICROARCH.md: 2.4 GHz peak)


def main():
    tag = sys.argv[1]
    p_ms = float(sys.argv[sys.argv.index("--p-ms") + 1]) if "--p-ms" in sys.argv else None
    ip = json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_decpmc_ip", "summary.json")))
    key = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_decoder.json")))
    st = ip["stream"]
    ksyms = st["symbols_first_frame"]
    psyms = st["symbols"] - ksyms
    nP = st["frames"] - 1
    kps = key["per_symbol"]
    pfr, kfr = {}, {}
    for name, total in sorted(ip["totals"].items()):
        if name not in kps:
            continue
        kfr[name] = round(kps[name], 3)
        pfr[name] = round((total - ksyms * kps[name]) / psyms, 3)
    out = {
        "command": f"tools/decoder_pmc2.sh {st['frames']} {tag}_decpmc_ip --ip  (tools/decode_only.py {st['frames']} --ip with "
                   "SCPR_NO_HELPERS=1: ONE GOP of the 1080p synthetic desktop, a key frame and "
                   f"{nP} P-frames, k_decode_gop_w<true> with the chain's wave alone - helper waves poll LDS words and would drown "
                   "the chain's instruction counts; the chain therefore also copies the previous plane itself, which the helpers "
                   "take off it in production)",
        "stream": st,
        "per_symbol_whole_stream": ip["per_symbol"],
        "per_symbol_p_frames": pfr,
        "per_symbol_key_frames": kfr,
        "derivation": f"P-frame figures = (stream totals - key-frame symbols x the key-frame per-symbol figures of {tag}_pmc_decoder.json)"
                      " / P-frame symbols  (tools/split_ip_pmc.py)",
        "summary": {
            "p_frame_symbols_per_frame": round(psyms / nP, 1),
            "p_frame_instructions_per_symbol": pfr.get("SQ_INSTS"),
            "p_frame_cycles_per_symbol_no_helpers": round(4 * pfr.get("SQ_WAVE_CYCLES", 0), 1),
            "key_frame_instructions_per_symbol": kfr.get("SQ_INSTS"),
            "key_frame_cycles_per_symbol": round(4 * kfr.get("SQ_WAVE_CYCLES", 0), 1),
        },
        "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles",
    }
    if p_ms is not None:
        cyc = p_ms * 1e-3 * CLOCK_GHZ * 1e9 / (psyms / nP)
        out["summary"]["p_frame_cycles_per_symbol_with_helpers_from_wall_clock"] = (
            f"{p_ms} ms per frame (tools/exp_dec_gop.py) x {CLOCK_GHZ} GHz / {psyms / nP:.0f} symbols = {cyc:.0f}")
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_decoder_ip.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("written", dst)
    print(json.dumps(out["summary"], indent=1))


if __name__ == "__main__":
    main()
