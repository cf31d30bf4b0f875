#!/usr/bin/env python3
"""Write one entry of profiles/hbm_traffic.json from a tools/prof_pmc.py result, STAMPED with the kernel sources it was measured on.
usage: python tools/update_hbm_traffic.py <cfg> <pmc_prefix>.json [--source profiles/rNN_pmc_default_<cfg>.json]
bench.py quotes the entry as roofline.traffic only while the forward kernel's sources still have that SHA-256 (bench.kernel_sources_sha256:
a kernel change makes the entry stale; hipcc objects are not bit-reproducible, so the library file itself cannot be the stamp)."""
import argparse, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("cfg")
ap.add_argument("pmc_json")
ap.add_argument("--source", default=None)
a = ap.parse_args()
r = json.load(open(a.pmc_json))
fetch_kb, write_kb = r["FETCH_SIZE"], r["WRITE_SIZE"]
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256
h = kernel_sources_sha256()
try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    head = "unknown"
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
hit = r.get("TCC_HIT_sum", 0.0) / max(r.get("TCC_HIT_sum", 0.0) + r.get("TCC_MISS_sum", 0.0), 1.0)
tj[a.cfg] = {
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "bytes_uncorrected": (fetch_kb + write_kb) * 1024.0,
    "bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
    "tcc_hit_rate": hit,
    "kernel_sources_sha256": h, "git_head": head,
    "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.py), mean over {int(r.get('dispatches', 0))} dispatches of the default "
            f"kernel on `bench.py --config {a.cfg}`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); "
            "WRITE_SIZE matches the algorithmic O+LSE bytes",
    "source": a.source or os.path.relpath(os.path.abspath(a.pmc_json), ROOT),
}
json.dump(tj, open(path, "w"), indent=1)
print(f"profiles/hbm_traffic.json[{a.cfg}]: {tj[a.cfg]['bytes'] / 1e6:.1f} MB per launch, L2 hit {hit:.3f}, kernel sources {h[:12]} ({head})")
