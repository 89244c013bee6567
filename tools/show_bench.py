"""Design tool: the numbers of a bench.py line (file argument) in a few rows."""
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = l["config"]
print("headline", l["value"], "MPix/s", l["ms_per_step"], "ms; parity", {k: l["parity"].get(k) for k in ("ok", "golden_fixture_ok", "golden_stream_ok")}, "frac", l["roofline"]["frac"])
print("  stages", c["stage_ms_per_step"])
print("  per-frame", c.get("per_frame_api_ms"), "host-incl", c.get("incl_host_transfer_MPix_s"))
print("  host boundary", c.get("host_boundary"))
print("  chain", l["roofline"].get("chain"), "bound", l["roofline"].get("bound"), "traffic", l["roofline"].get("traffic"), l["roofline"].get("traffic_source"))
print("  targets", c.get("targets"))
for o in c.get("others", []):
    if "error" in o:
        print("ERR", o)
        continue
    cb = o.get("cpu_baseline") or {}
    print(o["config"][:60].ljust(60), "comb", o.get("combined_MPix_s", o.get("value_MPix_s")), "enc", o.get("enc_MPix_s", o.get("enc_MPix_s_rank0")), "dec", o.get("dec_MPix_s", o.get("dec_MPix_s_rank0")),
          "gold", o.get("golden_stream_ok"), "par", (o.get("parity") or {}).get("ok"), "cpu", cb.get("value"), (cb.get("all_cores") or {}).get("value"))
    if o.get("stage_ms"):
        print("     ", {k: v for k, v in o["stage_ms"].items() if v >= 0.5})
