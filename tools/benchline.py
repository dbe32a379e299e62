#!/usr/bin/env python3
"""stdin: the JSON line of bench.py -> value, ms/step and the phase times (for shell sweeps)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line); print(tag, "%.2f M rays/s" % (j["value"] / 1e6), "%.4f ms" % j["ms_per_step"], {k: round(v, 4) for k, v in j["phase_ms"].items()}, "colour %.4f" % j["roofline"]["avg_kernel_ms"]["colour_pass"], "cks", {k: "%.6g" % v for k, v in (j.get("checksums") or {}).items()})
