#!/usr/bin/env python3
"""Writes profiles/r03/k2_traffic.json from two rocprofv3 --pmc passes over tools/kbench (tools/pmc.sh):
    bash tools/pmc.sh fetch "library ogd_remap" FETCH_SIZE TCC_EA0_RDREQ_32B_sum
    bash tools/pmc.sh write "library ogd_remap" WRITE_SIZE
    python tools/k2_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write
Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: FETCH_SIZE and WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE counts a 128-B fabric request as 64 B, so with no 32-B requests it is doubled."""
import csv
import glob
import json
import sys
from pathlib import Path

KERNEL = "ogd::remap_u8_wave_kernel<1, 1, 16, 2, 4096, ogd::MapGlobal>"
W, H, CELL = 6000, 8000, 25


def per_launch(d, counter):
	vals = {}
	# gpurun merges every call's files into the same local directory: take the newest pass only (dispatch ids repeat)
	files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=lambda f: Path(f).stat().st_mtime)
	for f in files[-1:]:
		for row in csv.DictReader(open(f)):
			if "MapGlobal" in row["Kernel_Name"] and "<1, 1, 16, 2, 4096" in row["Kernel_Name"] and row["Counter_Name"] == counter:
				vals.setdefault(row["Dispatch_Id"], 0.0)
				vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
	if not vals:
		return None, 0
	return sum(vals.values()) / len(vals), len(vals)


fetch_kb, nf = per_launch(sys.argv[1], "FETCH_SIZE")
rd32, _ = per_launch(sys.argv[1], "TCC_EA0_RDREQ_32B_sum")
write_kb, nw = per_launch(sys.argv[2], "WRITE_SIZE")
assert fetch_kb and write_kb, "counters not found"
h, w = -(-H // CELL) + 1 + 2, -(-W // CELL) + 1 + 2  # kbench's node grid
Hd, Wd = h * CELL, w * CELL
out = {
	"kernel": KERNEL,
	"workload": "one %dx%d gray page, cell %d, dense map %dx%d (tools/kbench, same launch as bench.py's K2)" % (W, H, CELL, Wd, Hd),
	"command": "bash tools/pmc.sh fetch 'library ogd_remap' FETCH_SIZE TCC_EA0_RDREQ_32B_sum ; bash tools/pmc.sh write 'library ogd_remap' WRITE_SIZE (separate rocprofv3 --pmc passes) ; python tools/k2_traffic.py",
	"FETCH_SIZE_KB_per_launch": fetch_kb,
	"WRITE_SIZE_KB_per_launch": write_kb,
	"TCC_EA0_RDREQ_32B_per_launch": rd32,
	"launches_averaged": [nf, nw],
	"correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B fabric requests at 64 B on gfx950 (no 32-B requests from this kernel) -> doubled; WRITE_SIZE used as is",
	"hbm_bytes_per_launch": int(round(fetch_kb * 1024 * 2 + write_kb * 1024)),
	"algorithmic_bytes_per_launch": Hd * Wd * 9 + H * W,
}
# bench.py reports these counters only for the source they were measured on
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
out["source_files"] = list(bench.K2_SOURCES)
out["source_sha256"] = bench.k2_source_hash()
p = Path(__file__).resolve().parent.parent / "profiles" / "r03" / "k2_traffic.json"
p.write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
