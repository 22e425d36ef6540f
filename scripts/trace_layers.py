"""Post-process a rocprofv3 --kernel-trace CSV of scripts/run_resnet_fwd.py: the dispatches of the LAST forward in order.
usage: python scripts/trace_layers.py <dir or kernel_trace.csv> [--all]   (default: contraction kernels only + a per-kind summary)"""
import glob
import os
import sys

import pandas as pd

src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
df = pd.concat([pd.read_csv(f) for f in files]).sort_values("Start_Timestamp").reset_index(drop=True)
names = df.Kernel_Name.astype(str)
# forwards start with the stem's image packing kernel
starts = df.index[names.str.contains("stem_pack_image_kernel")].tolist()
lo = starts[-1]
fwd = df.iloc[lo:].copy()
fwd["dur_us"] = (fwd.End_Timestamp - fwd.Start_Timestamp) / 1e3
fwd["gap_us"] = (fwd.Start_Timestamp - fwd.End_Timestamp.shift(1)) / 1e3
t0, t1 = fwd.Start_Timestamp.iloc[0], fwd.End_Timestamp.max()
print(f"last forward: {len(fwd)} dispatches, {(t1 - t0) / 1e6:.3f} ms wall, busy {fwd.dur_us.sum() / 1e3:.3f} ms, "
      f"gaps {fwd.gap_us.iloc[1:].clip(lower=0).sum() / 1e3:.3f} ms")


def short(n):
    n = n.replace("void dic::", "").replace("dic::", "")
    return n.split("(")[0][:48]


fwd["k"] = names.iloc[lo:].map(short)
print(fwd.groupby("k").dur_us.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False).head(20).to_string())
show_all = "--all" in sys.argv
for _, r in fwd.iterrows():
    if show_all or "gemm" in r.k or "conv" in r.k:
        g = int(r.Grid_Size_X) // max(1, int(r.Workgroup_Size_X)) if "Grid_Size_X" in r else -1
        print(f"{r.k:48s} {r.dur_us:8.1f} us  wgs {g:6d}  gap {r.gap_us:6.1f}")
