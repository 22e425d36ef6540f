"""Print the headline fields of a bench.py JSON line: python3 scripts/bench_line.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
print("stages_ms", d.get("stages_ms"))
for k in ("roofline", "resnet_forward", "decoder_roofline", "decoder_roofline_batch256", "parity", "prefetch_dropped"):
    print(k, d.get(k))
