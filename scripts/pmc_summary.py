"""Per-kernel PMC summary from separate rocprofv3 --pmc passes (one counter set per pass, as the hardware guide
prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains).

usage: python scripts/pmc_summary.py <dir with pmc_<SET>/.../*_counter_collection.csv> <out.json>

Per kernel name (averaged over its dispatches):
  hbm_bytes_per_launch_corrected = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
      (rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide streaming reads at 64 B, so
       it is doubled - MI355X_MICROARCH.md, HBM / rocprofv3 section)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
"""
import glob
import json
import os
import sys

import pandas as pd


def load(root, counter_set):
    files = glob.glob(os.path.join(root, f"pmc_{counter_set}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    return pd.concat([pd.read_csv(f) for f in files])


def per_kernel(df, counter):
    d = df[df.Counter_Name == counter]
    # one row per (dispatch, counter instance): sum instances within a dispatch, then average over dispatches
    per_dispatch = d.groupby(["Kernel_Name", "Dispatch_Id"]).Counter_Value.sum().reset_index()
    return per_dispatch.groupby("Kernel_Name").Counter_Value.agg(["mean", "count"])


def main():
    root, out = sys.argv[1], sys.argv[2]
    tabs = {}
    for cset, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                           ("SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]),
                           ("GRBM_GUI_ACTIVE", ["GRBM_GUI_ACTIVE"])):
        df = load(root, cset)
        if df is None:
            continue
        for c in counters:
            tabs[c] = per_kernel(df, c)
    names = set()
    for t in tabs.values():
        names |= set(t.index)
    recs = []
    for n in sorted(names):
        if "dic::" not in n:
            continue
        g = lambda c: float(tabs[c].loc[n, "mean"]) if c in tabs and n in tabs[c].index else None
        rec = {"kernel": n, "dispatches": int(tabs["FETCH_SIZE"].loc[n, "count"]) if "FETCH_SIZE" in tabs and n in tabs["FETCH_SIZE"].index else None,
               "FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
               "SQ_VALU_MFMA_BUSY_CYCLES": g("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CYCLES": g("SQ_BUSY_CYCLES"),
               "GRBM_GUI_ACTIVE": g("GRBM_GUI_ACTIVE")}
        if rec["FETCH_SIZE_KB"] is not None and rec["WRITE_SIZE_KB"] is not None:
            rec["hbm_bytes_per_launch_corrected"] = 2.0 * rec["FETCH_SIZE_KB"] * 1024 + rec["WRITE_SIZE_KB"] * 1024
        if rec["SQ_VALU_MFMA_BUSY_CYCLES"] and rec["GRBM_GUI_ACTIVE"]:
            rec["mfma_util"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        recs.append(rec)
    json.dump(recs, open(out, "w"), indent=1)
    for r in sorted(recs, key=lambda r: -(r.get("hbm_bytes_per_launch_corrected") or 0))[:12]:
        print(r["kernel"][:70], r.get("hbm_bytes_per_launch_corrected"), r.get("mfma_util"))


if __name__ == "__main__":
    main()
