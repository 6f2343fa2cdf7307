#!/usr/bin/env python3
"""Turns a PMC summary (tools/summarize_pmc.py) into profiles/traffic.json, the per-launch HBM traffic that bench.py
reports as roofline.traffic.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte fabric requests at
64 bytes, so the read side is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is."""
import json
import sys

summary = json.load(open(sys.argv[1]))
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "config3", "size": [int(sys.argv[3]), int(sys.argv[4])] if len(sys.argv) > 4 else [1920, 1080],
       "fetch_size_kib": summary["FETCH_SIZE"], "write_size_kib": summary["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2.0 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0,
       "l2_hit_rate": summary["TCC_HIT_sum"] / max(summary["TCC_HIT_sum"] + summary["TCC_MISS_sum"], 1.0),
       "valu_lane_utilisation": summary["SQ_THREAD_CYCLES_VALU"] / max(summary["SQ_ACTIVE_INST_VALU"] * 64.0, 1.0),
       # share of the SIMDs' issue cycles taken by vector ALU instructions: a wave64 instruction occupies its 16-lane SIMD for
       # 4 cycles; 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
       "valu_issue_busy": summary["SQ_INSTS_VALU"] * 4.0 / max(1024.0 * summary["GRBM_GUI_ACTIVE"] / 8.0, 1.0),
       "valu_instructions_per_launch": summary["SQ_INSTS_VALU"],
       # shader-clock cycles the launch lasted (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
       "gpu_cycles_per_launch": summary["GRBM_GUI_ACTIVE"] / 8.0,
       "source": "rocprofv3 --pmc, separate passes (tools/profile_pmc.sh), mean over the renderKernel<false> dispatches"}
print(json.dumps(out, indent=1))
