"""Board power / power cap / shader clock sampled while the headline kernel runs.

A sampler thread polls the SMI library (amdsmi python package; rocm_smi through ctypes as the fallback) as fast as it answers
while the main thread launches back-to-back forwards of one BASELINE shape for `--seconds` per data set.  Data sets: the
reference's normal(0, 0.5), all zeros, normal(0, 0.01) — the same instruction stream with different switching activity.
Prints one summary line per data set (TF/s, mean / p95 power, cap, mean / min / max gfx clock, throttle residency deltas when
the firmware reports them) and writes every sample to --out (CSV).

usage: python tools/power_trace.py [--cfg cfg3] [--seconds 3] [--out gpurun_out/r03_power_trace.csv]
"""
import argparse, ctypes as C, math, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiny_flash_attention_amd import _lib, ops

CFG = {"cfg3": (4, 32, 4096, 128, torch.bfloat16, True), "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
       "cfg4": (1, 16, 16384, 128, torch.bfloat16, False)}


class Smi:
    """whatever SMI interface answers on this box; every getter returns None when its quantity is not available"""

    def __init__(self):
        self.kind, self.h, self.rsmi = None, None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            if hs:
                self.kind, self.h, self.a = "amdsmi", hs[0], amdsmi
        except Exception as e:   # noqa: BLE001
            self.err = repr(e)
        if self.kind is None:
            try:
                r = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
                if r.rsmi_init(C.c_uint64(0)) == 0:
                    self.kind, self.rsmi = "rsmi", r
            except Exception as e:   # noqa: BLE001
                self.err = repr(e)

    def cap_w(self):
        try:
            if self.kind == "amdsmi":
                d = self.a.amdsmi_get_power_cap_info(self.h)
                v = d.get("power_cap")
                return float(v) / (1e6 if v and v > 100000 else 1.0)
            if self.kind == "rsmi":
                v = C.c_uint64()
                if self.rsmi.rsmi_dev_power_cap_get(0, 0, C.byref(v)) == 0:
                    return v.value / 1e6
        except Exception:   # noqa: BLE001
            pass
        return None

    def sample(self):
        """(power W, gfx clock MHz, extras dict)"""
        pw = clk = None
        ex = {}
        try:
            if self.kind == "amdsmi":
                try:
                    m = self.a.amdsmi_get_gpu_metrics_info(self.h)
                    for key in ("current_socket_power", "average_socket_power"):
                        v = m.get(key)
                        if isinstance(v, (int, float)) and 0 < v < 65535:
                            pw = float(v)
                            break
                    g = m.get("current_gfxclks") or m.get("current_gfxclk")
                    if isinstance(g, (list, tuple)):
                        g = [x for x in g if isinstance(x, (int, float)) and 0 < x < 65535]
                        clk = sum(g) / len(g) if g else None
                        if g:
                            ex["gfxclk_min"], ex["gfxclk_max"] = min(g), max(g)
                    elif isinstance(g, (int, float)) and 0 < g < 65535:
                        clk = float(g)
                    for key in ("accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc",
                                "vr_thm_residency_acc", "hbm_thm_residency_acc", "temperature_hotspot", "throttle_status", "indep_throttle_status",
                                "average_gfx_activity", "energy_accumulator", "voltage_gfx", "average_gfxclk_frequency", "firmware_timestamp"):
                        v = m.get(key)
                        if isinstance(v, (int, float)):
                            ex[key] = v
                except Exception:   # noqa: BLE001
                    pass
                if pw is None:
                    d = self.a.amdsmi_get_power_info(self.h)
                    for key in ("current_socket_power", "average_socket_power", "socket_power"):
                        v = d.get(key)
                        if isinstance(v, (int, float)) and 0 < v < 65535:
                            pw = float(v)
                            break
                if clk is None:
                    d = self.a.amdsmi_get_clock_info(self.h, self.a.AmdSmiClkType.GFX)
                    v = d.get("clk") or d.get("cur_clk")
                    clk = float(v) if isinstance(v, (int, float)) else None
            elif self.kind == "rsmi":
                v = C.c_uint64()
                if self.rsmi.rsmi_dev_current_socket_power_get(0, C.byref(v)) == 0 or self.rsmi.rsmi_dev_power_ave_get(0, 0, C.byref(v)) == 0:
                    pw = v.value / 1e6
        except Exception as e:   # noqa: BLE001
            ex["err"] = repr(e)[:80]
        return pw, clk, ex


def run_cmd(cmd, out):
    """--cmd mode: run a command that prints ARM_BEGIN <tag> / ARM_END <tag> lines; report power and clock per arm"""
    import subprocess
    smi = Smi()
    print(f"smi interface: {smi.kind}  power cap: {smi.cap_w()} W", flush=True)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            pw, clk, ex = smi.sample()
            samples.append((time.perf_counter(), pw, clk, ex))
            time.sleep(0.002)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    arms, cur, lines = [], None, []
    pr = subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1)
    for line in pr.stdout:
        t = time.perf_counter()
        line = line.rstrip()
        if line.startswith("ARM_BEGIN"):
            cur = (line[len("ARM_BEGIN"):].strip(), t)
        elif line.startswith("ARM_END") and cur:
            arms.append((cur[0], cur[1], t, line[len("ARM_END"):].strip()))
            cur = None
        else:
            print(line, flush=True)
    pr.wait()
    stop.set()
    th.join()
    for tag, t0, t1, text in arms:
        mine = [x for x in samples if t0 + 0.5 <= x[0] <= t1]
        pw = sorted(x[1] for x in mine if x[1] is not None)
        ck = [x[2] for x in mine if x[2] is not None]
        ppt = [x[3].get("ppt_residency_acc") for x in mine if "ppt_residency_acc" in x[3]]
        acc = [x[3].get("accumulation_counter") for x in mine if "accumulation_counter" in x[3]]
        res = f"{ppt[-1] - ppt[0]}/{acc[-1] - acc[0]}" if len(ppt) > 1 and len(acc) > 1 else "n/a"
        line = (f"{text}  | power mean {(sum(pw) / len(pw)) if pw else float('nan'):6.1f} W max {pw[-1] if pw else float('nan'):6.1f} W (cap {smi.cap_w()} W)"
                f"  gfxclk mean {(sum(ck) / len(ck)) if ck else float('nan'):6.0f} MHz  PPT-limited firmware samples {res}")
        print(line, flush=True)
        lines.append(line)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out.replace(".csv", ".txt"), "w") as f:
        f.write(f"command: {cmd}\nsmi interface: {smi.kind}  power cap: {smi.cap_w()} W\n" + "\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cmd", default="", help="sample while this shell command runs (it prints ARM_BEGIN/ARM_END lines) instead of the attention kernel")
    ap.add_argument("--cfg", default="cfg3")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--idle", type=float, default=1.0, help="idle seconds sampled between data sets")
    ap.add_argument("--data", default="normal,zeros,small,normal")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_power_trace.csv"))
    a = ap.parse_args()
    if a.cmd:
        return run_cmd(a.cmd, a.out)
    dev = torch.device("cuda:0")
    L = _lib.lib()
    smi = Smi()
    print(f"smi interface: {smi.kind}  power cap: {smi.cap_w()} W", flush=True)
    B, H, N, D, dt, causal = CFG[a.cfg]
    samples = []          # (t, phase, power, clk, extras)
    phase = ["idle"]
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            pw, clk, ex = smi.sample()
            samples.append((time.perf_counter(), phase[0], pw, clk, ex))
            time.sleep(0.002)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.set_variant(a.variant)
    lines = []
    for i, data in enumerate(a.data.split(",")):
        mk = {"normal": lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.5).to(dt),
              "small": lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0, 0.01).to(dt),
              "zeros": lambda: torch.zeros((B, H, N, D), dtype=dt, device=dev),
              "ones": lambda: torch.ones((B, H, N, D), dtype=dt, device=dev)}[data]
        q, k, v = mk(), mk(), mk()
        out = torch.empty_like(q)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)
        p = ops.make_params(q, k, v, out, lse, causal, 1 / math.sqrt(D))
        fl, by = C.c_double(), C.c_double()
        L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
        torch.cuda.synchronize()
        phase[0] = f"idle{i}"
        time.sleep(a.idle)
        tag = f"{data}{i}"
        phase[0] = tag
        ms = C.c_float()
        t0 = time.perf_counter()
        rates = []
        while time.perf_counter() - t0 < a.seconds:
            _lib.check(L.tfa_fwd_time(C.byref(p), 0, 100, s, C.byref(ms)))
            rates.append(fl.value / (ms.value * 1e-3) / 1e12)
        phase[0] = f"idle{i}b"
        mine = [x for x in samples if x[1] == tag]
        # drop the first 0.5 s (the DVFS ramp) from the summary
        tcut = mine[0][0] + 0.5 if mine else 0
        pw = sorted(x[2] for x in mine if x[2] is not None and x[0] >= tcut)
        ck = [x[3] for x in mine if x[3] is not None and x[0] >= tcut]
        first, last = (mine[0][4], mine[-1][4]) if mine else ({}, {})
        resid = {k_: last[k_] - first[k_] for k_ in last if k_.endswith("_acc") and k_ in first}
        acc = (last.get("accumulation_counter", 0) - first.get("accumulation_counter", 0)) if mine else 0
        tf_last = sorted(rates[len(rates) // 2:])[len(rates[len(rates) // 2:]) // 2] if rates else float("nan")
        line = (f"{a.cfg} [{data}] {tf_last:7.1f} TF (median of the second half)  samples {len(mine)}  power mean "
                f"{(sum(pw) / len(pw)) if pw else float('nan'):6.1f} W p95 {pw[int(0.95 * (len(pw) - 1))] if pw else float('nan'):6.1f} W max {pw[-1] if pw else float('nan'):6.1f} W"
                f"  cap {smi.cap_w()} W  gfxclk mean {(sum(ck) / len(ck)) if ck else float('nan'):6.0f} min {min(ck) if ck else float('nan'):6.0f} max {max(ck) if ck else float('nan'):6.0f} MHz"
                f"  residency deltas {resid} over {acc} firmware samples  hotspot {last.get('temperature_hotspot')} C")
        print(line, flush=True)
        lines.append(line)
    time.sleep(a.idle)
    stop.set()
    th.join()
    _lib.set_variant(-1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    t00 = samples[0][0] if samples else 0
    keys = sorted({k_ for x in samples for k_ in x[4]})
    with open(a.out, "w") as f:
        f.write("t_s,phase,power_w,gfxclk_mhz," + ",".join(keys) + "\n")
        for t, ph, pw, clk, ex in samples:
            f.write(f"{t - t00:.4f},{ph},{'' if pw is None else pw},{'' if clk is None else round(clk, 1)}," + ",".join(str(ex.get(k_, "")) for k_ in keys) + "\n")
    with open(a.out.replace(".csv", ".txt"), "w") as f:
        f.write(f"smi interface: {smi.kind}  power cap: {smi.cap_w()} W  ({len(samples)} samples, {a.out})\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
