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
    if sys.argv[1] == "--rederive":                  # rewrite <prefix>.txt from <prefix>.json (no GPU needed)
        for pre in sys.argv[2:]:
            res = json.load(open(pre + ".json"))
            head = open(pre + ".txt").read().splitlines()
            write_txt(pre, res, head[0].split("'")[1], head[1].split("command: ", 1)[1])
        return
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
    write_txt(out, res, kern_filter, " ".join(cmd))


N_SIMD, N_XCD, MFMA_CYCLES = 1024, 8, 32   # MI355X: 256 CUs x 4 SIMDs in 8 XCDs; v_mfma_f32_32x32x16_{bf16,f16} holds the pipe 32 cycles


def write_txt(out, res, kern_filter, cmd):
    with open(out + ".txt", "w") as f:
        f.write("# per-launch means over dispatches of kernels matching '%s'\n# command: %s\n" % (kern_filter, cmd))
        for k in sorted(res):
            f.write(f"{k:34s} {res[k]}\n")
        gui = res.get("GRBM_GUI_ACTIVE")
        if gui:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (cfg3: 7.78e6 = 8 x 0.97e6 cycles = 0.52 ms at 1.88 GHz); the SQ
            # counters are summed over all SIMDs.  SQ_VALU_MFMA_BUSY_CYCLES equals 32 x SQ_INSTS_MFMA exactly where it does
            # not saturate (it is a 31-bit counter: cfg4 reads 2147483648), so the instruction count is the robust source.
            f.write("\n# derived: matrix-pipe busy fraction over the whole launch, per SIMD = 32 cycles x MFMA instructions / 1024 SIMDs\n"
                    "#          divided by the launch's shader cycles = GRBM_GUI_ACTIVE / 8 XCDs\n")
            if res.get("SQ_INSTS_MFMA"):
                f.write(f"mfma_pipe_busy_from_insts          {MFMA_CYCLES * res['SQ_INSTS_MFMA'] / N_SIMD / (gui / N_XCD):.4f}\n")
            b = res.get("SQ_VALU_MFMA_BUSY_CYCLES")
            if b and b < 2147483648.0:
                f.write(f"mfma_pipe_busy_from_busy_cycles    {b / N_SIMD / (gui / N_XCD):.4f}\n")
            if res.get("SQ_INSTS_VALU") and res.get("SQ_INSTS_MFMA"):
                f.write(f"valu_insts_per_mfma (non-MFMA)     {(res['SQ_INSTS_VALU'] - res['SQ_INSTS_MFMA']) / res['SQ_INSTS_MFMA']:.2f}\n")
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
