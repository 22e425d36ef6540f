"""Aggregate the last forward of a `rocprofv3 --kernel-trace --output-format csv` run of scripts/run_resnet_fwd.py by kernel kind and
grid size: python3 scripts/agg_trace.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[[i for i, r in enumerate(rows) if "bn_relu_maxpool" in r["Kernel_Name"]][-1]:]
KINDS = (("astat", "astat"), ("ws256", "ws256"), ("bn_finalize_train", "bnfin"), ("bn_finalize_from", "bnfs"), ("bn_stats_slice", "bnss"), ("tail_fixup", "tailfx"),
         ("persist_ws_kernel<6", "ws6"), ("persist_ws_kernel<0", "ws0"), ("persist_ws_kernel<2", "ws2"), ("halo", "halo"),
         ("bn_apply_planes", "bnp"), ("bn_apply_kernel", "bna"), ("gemm_bf3_kernel", "g64"))
agg = collections.OrderedDict()
tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    short = next((s for k, s in KINDS if k in r["Kernel_Name"]), "other")
    agg.setdefault((short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))), []).append(d)
for k, v in agg.items():
    print(f"{k[0]:6s} grid {k[1]:6d}  x{len(v):3d}  avg {sum(v) / len(v):7.1f} us  total {sum(v) / 1e3:7.3f} ms")
print(f"kernel time of the forward: {tot / 1e3:.3f} ms in {len(rows)} launches")
