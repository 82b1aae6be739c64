#!/usr/bin/env python3
"""tools/update_traffic.py <workload> <batch> <profile-tag> — fold the output of tools/pmc_traffic.sh (gpurun_out/traffic_<w>.json,
one JSON line: fetch_kb_per_step / write_kb_per_step) into profiles/hbm_traffic.json, together with the sha256 of the kernel
sources the workload runs on, so that bench.py reports the figure only while those sources are unchanged."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

KERNEL_SOURCES = {
    "c3hdr": ["vp_fused_up2x.h", "vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "c1": ["vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "c3": ["vp_fused_up2x.h", "vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "c5": ["vp_fused_up2x.h", "vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "hdr4k": ["vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "up1440": ["vp_fused_period.h", "vp_fused_period.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "down1440": ["vp_fused_period.h", "vp_fused_period.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "up2160": ["vp_fused_period.h", "vp_fused_period.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "jinc1080": ["vp_fused_jinc.hip", "vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "dovi4k": ["vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
    "c4ed": ["vp_errdiff.hip", "vp_errdiff_core.h", "vp_fused_up2x.h", "vp_fused.hip", "vp_fused_dev.h", "vp_device.h", "vp_params.h"],
}
ALGO = {"c3": 157593600, "c5": 157593600, "c3hdr": 157593600, "c1": 11404800, "hdr4k": 58060800, "up1440": 20966400, "down1440": 39628800, "up2160": 35942400, "c4ed": 157593600, "jinc1080": 39398400, "dovi4k": 58060800}

w, batch, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
m = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", f"traffic_{w}.json")) if l.startswith("{")][-1])
doc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = json.load(open(doc_path))
fetch, write = m["fetch_kb_per_step"], m["write_kb_per_step"]
doc[w] = {"batch": batch, "fetch_kb": round(fetch), "write_kb": round(write),
          "bytes_per_launch": int((2 * fetch + write) * 1024), "algorithmic_bytes_per_launch": ALGO[w] * batch,
          "profile": tag, "sources": KERNEL_SOURCES[w], "csrc_sha256": bench.csrc_digest(KERNEL_SOURCES[w]),
          "valu_issue_frac": (round(m["valu_issue_frac"], 4) if m.get("valu_issue_frac") else None),
          "wait_inst_any_share": (round(m["wait_inst_any_share"], 4) if m.get("wait_inst_any_share") else None),
          "sustained_mhz": (round(m["sustained_mhz"]) if m.get("sustained_mhz") else None),
          "note": f"tools/pmc_traffic.sh {w}: FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate rocprofv3 --pmc passes, per {batch}-frame step"}
json.dump(doc, open(doc_path, "w"), indent=1)
print(w, "traffic / algorithmic =", round(doc[w]["bytes_per_launch"] / doc[w]["algorithmic_bytes_per_launch"], 3), "valu_issue_frac =", doc[w]["valu_issue_frac"])
