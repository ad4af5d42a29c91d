#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (gpurun_out/pmc{1,2,3}/p_counter_collection.csv) into profiles/.

Passes (collected separately, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
  pmc1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pmc2: FETCH_SIZE      pmc3: WRITE_SIZE
gfx950 corrections applied: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B requests at 64 B for wide
(16 B/lane) coalesced reads, which is what every conv staging load is, so read bytes = 2 x FETCH_SIZE x 1024.
GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.
usage: python tools/summarize_pmc.py <round-tag> [dominant-kernel-substring]
"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dom = sys.argv[2] if len(sys.argv) > 2 else "attn_kernel"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in ("pmc1", "pmc2", "pmc3"):
    path = os.path.join(ROOT, "gpurun_out", p, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p == "pmc1" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, d in agg.items():
    if not k.startswith("void dani::") and not k.startswith("dani::"):
        continue
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches_sampled"] = max(len(v) for v in d.values())
    if dur[k]:
        e["avg_duration_ns_profiled"] = sum(dur[k]) / len(dur[k])
    if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["kernel_cycles_per_xcd"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
        # GRBM_GUI_ACTIVE / 8 / duration is only a clock for dispatches of >= 0.3 ms (MI355X_MICROARCH.md, DVFS
        # give-back: the quotient reads high on shorter ones), so it is reported for those alone
        if e.get("avg_duration_ns_profiled", 0) >= 300000:
            e["effective_clock_ghz_long_dispatch"] = cyc / e["avg_duration_ns_profiled"]
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = 2 * e.get("FETCH_SIZE", 0) * 1024 + e.get("WRITE_SIZE", 0) * 1024
    out[k] = e
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
# the dominant kernel may have several template instantiations (linear_rs_kernel<NCB, UQ>): launch-weighted average
match = [(k, e) for k, e in out.items() if dom in k]
if match:
    n = sum(e["launches_sampled"] for _, e in match)
    hbm = sum(e.get("hbm_bytes_per_launch", 0) * e["launches_sampled"] for _, e in match) / n
    busy = [e["mfma_busy_frac"] * e["launches_sampled"] for _, e in match if "mfma_busy_frac" in e]
    json.dump({"kernel": " | ".join(k for k, _ in match), "hbm_bytes_per_launch": round(hbm),
               "source": f"profiles/{tag}_pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, KiB->B, launch-weighted average over the instantiations listed)",
               "mfma_busy_frac": (sum(busy) / n) if busy else None,
               "per_instantiation": {k: {"launches_sampled": e["launches_sampled"], "hbm_bytes_per_launch": round(e.get("hbm_bytes_per_launch", 0)),
                                         "mfma_busy_frac": e.get("mfma_busy_frac"), "avg_duration_ns_profiled": e.get("avg_duration_ns_profiled")} for k, e in match}},
              open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("dominant:", [k for k, _ in match], "hbm bytes per launch", round(hbm))
print("wrote profiles/%s_pmc_summary.json" % tag)
