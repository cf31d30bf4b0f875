#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for the attention kernel in separate passes (one `--pmc` group
per run, never combined with tracing — see the guide's HBM/rocprofv3 section) and write a
per-launch summary (mean over dispatches of the fwd kernel) to <out>.json / <out>.txt.

usage: python tools/prof_pmc.py <out_prefix> -- <command ...>
"""
import csv
import glob
import json
import os
import subprocess
import sys

GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT",
     "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_TRANS_F32"],
    ["GRBM_GUI_ACTIVE", "FETCH_SIZE"],
    ["GRBM_GUI_ACTIVE", "WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_VALU_MFMA_COEXEC_CYCLES"],
]


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    kern_filter = os.environ.get("KERNEL_FILTER", "fwd_kernel")
    res = {}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    for gi, grp in enumerate(GROUPS):
        d = f"/tmp/pmc_{os.getpid()}_{gi}"
        full = ["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            res[f"group{gi}_error"] = r.stdout[-2000:]
            continue
        acc, cnt = {}, {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if kern_filter not in row.get("Kernel_Name", ""):
                    continue
                name, val = row["Counter_Name"], float(row["Counter_Value"])
                acc[name] = acc.get(name, 0.0) + val
                cnt[name] = cnt.get(name, 0) + 1
        for k in acc:
            res[k] = acc[k] / cnt[k]
        res.setdefault("dispatches", max(cnt.values()) if cnt else 0)
    json.dump(res, open(out + ".json", "w"), indent=1, sort_keys=True)
    with open(out + ".txt", "w") as f:
        f.write("# per-launch means over dispatches of kernels matching '%s'\n# command: %s\n" % (kern_filter, " ".join(cmd)))
        for k in sorted(res):
            f.write(f"{k:34s} {res[k]}\n")
        if "SQ_WAVE_CYCLES" in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res:
            f.write("\n# derived (per guide: SQ_* wave counters are quad-cycles, MFMA_BUSY is cycles)\n")
            if res.get("SQ_BUSY_CYCLES"):
                f.write(f"mfma_busy_frac_of_sq_busy          {res['SQ_VALU_MFMA_BUSY_CYCLES'] / res['SQ_BUSY_CYCLES']:.4f}\n")
            if res.get("GRBM_GUI_ACTIVE"):
                f.write(f"mfma_busy_per_simd / gui_active    {res['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * res['GRBM_GUI_ACTIVE']):.4f}\n")
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
