#!/usr/bin/env python3
"""profiles/valu_issue.json from the output of build/ubench_stream (tools/ubench/gen_stream_ubench.py): what the final-stage
kernel's own VALU instruction stream sustains per SIMD at its occupancy, tied to the kernels' source hash.
    python tools/make_issue_json.py gpurun_out/<tag>_ubench_stream.txt <tag>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import _native  # noqa: E402

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\w+)\s+(\d+) waves/SIMD:\s+([\d.]+) cycles per trip and SIMD = ([\d.]+) per VALU instruction \((\d+) VALU, (\d+) instructions in all\); clock ([\d.]+) GHz", line)
    if m:
        rows[(m.group(1), int(m.group(2)))] = {"cycles_per_trip": float(m.group(3)), "cycles_per_valu_inst": float(m.group(4)), "valu_insts": int(m.group(5)),
                                              "clock_ghz": float(m.group(7))}
v = rows[("valu", 4)]
rec = {"source_hash": _native.source_hash(), "tag": sys.argv[2] if len(sys.argv) > 2 else "round",
       "what": "stage_tube2_kernel<rgb>: the VALU instructions of pairs 0..3 of a channel (8 passes), same registers, modifiers and order, as a loop of "
               "their own (no LDS, no waits); one 1024-thread workgroup per CU = 4 waves per SIMD; cycles = s_memtime, clock = s_memtime / s_memrealtime",
       "stream_cycles_per_valu_inst": v["cycles_per_valu_inst"], "stream_clock_ghz": v["clock_ghz"], "stream_valu_insts_per_8_passes": v["valu_insts"],
       "mac_only": rows.get(("mac", 4)), "two_waves_per_simd": rows.get(("valu", 2)), "guide_cycles_per_wave64_valu_inst": 2.0}
json.dump(rec, open(os.path.join(ROOT, "profiles", "valu_issue.json"), "w"), indent=1)
print(json.dumps(rec))
