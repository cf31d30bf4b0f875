#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract for the FlashAttention-2 forward hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the hot path (tfa_fwd, include/tfa.h) over one synthetic batch
already resident in HBM.  At N=1 the workload is BASELINE.json's headline configuration
(config 3: B=4, H=32, N=4096, D=128, bf16, causal).  For N>1 every rank runs ITS SHARD of BASELINE
config 5 (B=64 over 8 GPUs: per-GPU B=8, global batch 8*N — at N=8 exactly config 5) on its own GPU
(batch sharding, no data-path collective: each (b,h) pair is an independent problem,
flash_attention_cutlass/csrc/flash_attention.cu:382,409,698) -> weak scaling; `--config` overrides.
For N>1 (or with `--gather`) the same JSON line carries BOTH numbers: `value` = kernel only, and the
sub-dict `gather` = the step plus the RCCL all-gather of the output shards (north_star's "trivial
gather"), chunked over the batch so that chunk c's gather (side stream) overlaps chunk c+1's kernel.
`--gpus N` without a torchrun environment re-launches itself under torch.distributed.run.

Protocol (what happens, in order; all of it is reported in the JSON line):
  1. PRE-CONDITIONING (`--precondition-s`, default 2 s, untimed, disclosed as `preconditioning`):
     back-to-back launches of the same step until the time is up.  A fresh MI355X sits at its idle
     clock / power state; the first few hundred milliseconds of any launch sequence measure the DVFS
     ramp, not the kernel (round 1: the same binary gave 909 TFLOP/s with 5 warm-up launches and
     1.06-1.10 PFLOP/s after 100).  The launch rate at the start and at the end of the phase and the
     shader clock before/after are emitted so the effect is visible.  `--precondition-s 0` disables it.
  2. W warm-up steps (untimed), barrier + synchronize.
  3. EXACTLY K timed steps between HIP events on the launch stream and a host timer, barrier +
     synchronize, max over ranks -> `value`, `ms_per_step`, `roofline.achieved`.
  4. K more steps with one event pair per launch -> `per_launch_ms` {median, min, max, mean}.
  5. K steps in the reference's own harness style: host timer, synchronize after EVERY call
     (flash_attention_cutlass/test.py:30-40) -> `reference_harness_ms`.
  6. one traced launch -> sustained shader clock.
  7. rank 0, N=1, forward mode: `secondary` — <= 11 s of driver-timed twins of the other numbers quoted in DESIGN.md: BASELINE
     configs 2 and 4, the headline shape non-causal, one GQA decode shape (HBM-bound) and the headline backward.
  8. rank 0, N=1: the reference's CPU paths on the host cores (`cpu_baseline`: the C path on a bounded
     sample of the same workload; `cpu_baseline_python`: the pure-Python path on its own config 1).

Rank 0 prints ONE JSON line.  `value` = algorithmic flops of all ranks / wall time of the K
steps (max over ranks).  `roofline.achieved` = algorithmic flops per launch / average launch
duration measured with HIP events on the launch stream over the same timed region.
"""
import argparse
import ctypes as C
import gc
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_BF16 = 2500.0   # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def box_info():
    """Which box produced this line: host name, the GPU's id and its power cap — boxes of the pool run the same binaries 4-8 % apart on random data
    (a lower power budget; equally fast on zeros), so a headline is attributable only with these beside it.  Best effort: None where the SMI
    library does not answer."""
    info = {"hostname": socket.gethostname(), "gpu_name": None, "gpu_id": None, "power_cap_w": None, "smi": None}
    try:
        info["gpu_name"] = torch.cuda.get_device_name(0)
    except Exception:
        pass
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        info["smi"] = "amdsmi"
        try:
            v = amdsmi.amdsmi_get_power_cap_info(h).get("power_cap")
            info["power_cap_w"] = float(v) / (1e6 if v and v > 100000 else 1.0)
        except Exception:
            pass
        for fn in ("amdsmi_get_gpu_device_uuid", "amdsmi_get_gpu_device_bdf"):
            try:
                info["gpu_id"] = str(getattr(amdsmi, fn)(h))
                break
            except Exception:
                continue
    except Exception:
        try:
            r = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
            if r.rsmi_init(C.c_uint64(0)) == 0:
                info["smi"] = "rsmi"
                v = C.c_uint64()
                if r.rsmi_dev_power_cap_get(0, 0, C.byref(v)) == 0:
                    info["power_cap_w"] = v.value / 1e6
                if r.rsmi_dev_unique_id_get(0, C.byref(v)) == 0:
                    info["gpu_id"] = hex(v.value)
        except Exception:
            pass
    return info


CONFIGS = {
    # name: (B, H, N, D, dtype, causal)  — BASELINE.json configs 2..5 (per-GPU shapes)
    "cfg2": (4, 8, 1024, 64, torch.float16, False),
    "cfg3": (4, 32, 4096, 128, torch.bfloat16, True),     # headline
    "cfg3nc": (4, 32, 4096, 128, torch.bfloat16, False),
    "cfg4": (1, 16, 16384, 128, torch.bfloat16, False),
    "cfg4c": (1, 16, 16384, 128, torch.bfloat16, True),
    "cfg5": (8, 32, 4096, 128, torch.bfloat16, True),     # per-GPU shard of B=64 over 8 GPUs
}


def cpu_baseline(B, H, N, D, causal, target_s=15.0):
    """Time the reference CPU path on a bounded sample (heads of the same workload)."""
    from oracle import oracle as O   # checker/baseline only — never on the product path

    ref = O.ref_kernels()
    kind = "reference" if ref is not None else "port"
    fn = (lambda q, k, v, c, s: ref.flash_attn(q, k, v, c, s)) if ref is not None else O.flash_attn
    cores = torch.get_num_threads() if ref is not None else O.num_threads()
    sc = 1.0 / math.sqrt(D)

    def run(heads):
        q, k, v = O.make_inputs(1, heads, N, D, torch.float32, seed=0)
        t0 = time.perf_counter()
        fn(q, k, v, causal, sc)
        return time.perf_counter() - t0

    run(1)                                   # warm-up (OpenMP pool)
    probe_heads = max(1, min(2, B * H))
    t_probe = run(probe_heads)
    per_head = t_probe / probe_heads
    heads = int(max(probe_heads, min(B * H, target_s / max(per_head, 1e-6))))
    t = run(heads) if heads > probe_heads else t_probe
    flops = 4.0 * heads * N * N * D * (0.5 if causal else 1.0)
    return {
        "value": flops / t / 1e12,
        "unit": "TFLOP/s",
        "cores": int(os.cpu_count() if ref is not None else cores),
        "threads": int(cores),
        "kind": kind,
        "sample": f"{heads} of {B * H} (b,h) heads of the same workload (N={N}, D={D}, causal={causal}), fp32, "
                  f"{t:.2f} s; flash_attention_c flash_attn (attn.cpp:101-169)",
    }


def cpu_baseline_python(reps=3):
    """The reference's pure-Python path (flash_attention_py/tiny_flash_attn.py:137-196) on ITS configuration —
    BASELINE config 1: B=1 H=2 N=128 D=64 fp32, no scale, no mask, BLOCK_M=4.  At the headline size its
    Python double loop is ~1e6 iterations per head (infeasible), so it is timed where the reference
    runs it.  /root/reference does not exist on the GPU box: this is the line-cited restatement in oracle/
    (kind "port"), pinned bit-for-bit-close (1e-6) to the reference's own output by tests/test_oracle.py."""
    from oracle import oracle as O

    q, k, v = O.make_inputs(1, 2, 128, 64, torch.float32, seed=0)
    O.tiny_py_multihead(q, k, v, 4)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.tiny_py_multihead(q, k, v, 4)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    flops = 4.0 * 1 * 2 * 128 * 128 * 64
    return {
        "value": flops / t / 1e12,
        "unit": "TFLOP/s",
        "ms": t * 1e3,
        "cores": int(os.cpu_count()),
        "threads": int(torch.get_num_threads()),
        "kind": "port",
        "sample": f"BASELINE config 1 whole (B=1 H=2 N=128 D=64 fp32, scale 1, non-causal, BLOCK_M=4), best of {reps}; "
                  "tiny_flash_attn.py flash_attn_v2_multihead (:137-196) restated in oracle/oracle.py",
    }


def secondary_measurements(dev, budget_s=11.0):
    """Driver-timed twins of the builder-run numbers: after the headline's timed region, <= budget_s in
    total, the other BASELINE configs, one decode shape and the headline backward — each: 0.25 s of untimed launches, then
    one HIP-event pair around as many launches as fit its share of the budget.  `frac` = fraction of the 2.5 PF bf16 MFMA
    peak, `hbm_frac` = algorithmic bytes / time against 8 TB/s (the bound of the decode shape)."""
    from tiny_flash_attention_amd import _lib, ops

    L = _lib.lib()
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    cases = [
        ("cfg2", "fwd", (4, 8, 8, 1024, 1024, 64, torch.float16, False)),
        ("cfg3nc", "fwd", (4, 32, 32, 4096, 4096, 128, torch.bfloat16, False)),
        # the headline with TFA_FWD_EXACT_MAX: P rounded at the reference's own points (element-wise rtol 1e-3, tests/test_parity_gpu.py)
        ("cfg3_exact_max", "fwd_exact", (4, 32, 32, 4096, 4096, 128, torch.bfloat16, True)),
        ("cfg4", "fwd", (1, 16, 16, 16384, 16384, 128, torch.bfloat16, False)),
        ("decode_B64_H32_Hk8_Nq1_Nk8192", "fwd", (64, 32, 8, 1, 8192, 128, torch.bfloat16, True)),
        ("cfg3_bwd", "bwd", (4, 32, 32, 4096, 4096, 128, torch.bfloat16, True)),
        # head dims above 128 (the reference's "classic" 8 heads x 256, flash_attention_cutlass/test.py:44-48): forward, backward, decode
        ("d256_fwd", "fwd", (4, 8, 8, 4096, 4096, 256, torch.bfloat16, True)),
        ("d256_bwd", "bwd", (4, 8, 8, 4096, 4096, 256, torch.bfloat16, True)),
        ("decode_d256_B1_H16_Nq1_Nk65536", "split", (1, 16, 16, 1, 65536, 256, torch.bfloat16, True)),
    ]
    share = budget_s / len(cases)
    res = {}
    for name, mode, (B, H, Hk, Nq, Nk, D, dt, causal) in cases:
        try:
            mk = lambda h, n: torch.empty((B, h, n, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5).to(dt)
            q, k, v = mk(H, Nq), mk(Hk, Nk), mk(Hk, Nk)
            sc = 1.0 / math.sqrt(D)
            out = torch.empty_like(q)
            lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
            p = ops.make_params(q, k, v, out, lse, causal, sc)
            if mode == "fwd_exact":
                p.flags = _lib.TFA_FWD_EXACT_MAX
            fl, by = C.c_double(), C.c_double()
            if mode == "bwd":
                _lib.check(L.tfa_fwd(C.byref(p), sptr))
                dout = mk(H, Nq)
                dq, dk, dv, delta = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(lse)
                pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
                call = lambda: _lib.check(L.tfa_bwd(C.byref(pb), sptr))
                L.tfa_bwd_work(C.byref(pb), C.byref(fl), C.byref(by))
            elif mode == "split":                            # what the reference-named entry points do for decode-like calls
                splits = int(L.tfa_fwd_suggest_splits(C.byref(p)))
                L.tfa_fwd_splitkv_workspace.restype = C.c_longlong
                ws = torch.empty((max(int(L.tfa_fwd_splitkv_workspace(C.byref(p), splits)), 4),), dtype=torch.float32, device=dev)
                call = (lambda: _lib.check(L.tfa_fwd_splitkv(C.byref(p), splits, C.c_void_p(ws.data_ptr()), sptr))) if splits > 1 else (lambda: _lib.check(L.tfa_fwd(C.byref(p), sptr)))
                L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
            else:
                call = lambda: _lib.check(L.tfa_fwd(C.byref(p), sptr))
                L.tfa_fwd_work(C.byref(p), C.byref(fl), C.byref(by))
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_warm = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            while time.perf_counter() - t0 < 0.25:          # untimed: clocks settle on this shape; also sizes the timed batch
                e0.record(stream)
                for _ in range(10):
                    call()
                e1.record(stream)
                e1.synchronize()
                n_warm += 10
            per = max(e0.elapsed_time(e1) / 10.0, 1e-3)     # ms per launch, roughly
            n = int(max(10, min(2000, (share - 0.35) * 1e3 / per)))
            e0.record(stream)
            for _ in range(n):
                call()
            e1.record(stream)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / n
            tfs = fl.value / (ms * 1e-3) / 1e12
            gbs = by.value / (ms * 1e-3) / 1e9
            res[name] = {"shape": f"B{B} H{H} Hk{Hk} Nq{Nq} Nk{Nk} D{D} {'bf16' if dt == torch.bfloat16 else 'fp16'} {'causal' if causal else 'full'}",
                         "mode": mode, "ms": ms, "launches": n, "tflops": tfs, "frac": tfs / PEAK_TFLOPS_BF16,
                         "algorithmic_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS,
                         "bound": "hbm" if name.startswith("decode") else "mfma",
                         "kernel_variant": _lib.variant_name(L.tfa_fwd_variant(C.byref(p))) if mode in ("fwd", "fwd_exact") else
                                           (f"tfa_fwd_splitkv, {splits} chunks in one launch + merge" if mode == "split" else
                                            ("tfa_bwd: dQ launch (forms delta) + fused dK/dV launch" if D <= 128 else "tfa_bwd: dQ (forms delta), dK and dV launches (256-wide kernels)"))}
            del q, k, v, out, lse
        except Exception as e:   # a report, never a reason to lose the headline line
            res[name] = {"error": repr(e)}
    return res


def workload_text(cfg_name, world, bwd=False, bwd_form="default"):
    B, H, N, D, dtype, causal = CONFIGS[cfg_name]
    what = ("backward (" + bwd_form + " form" + {"default": ": dQ launch, which forms delta, + fused dK/dV launch", "split": ": delta, dQ, dK, dV launches",
                                                       "workspace": ": delta, fused dK/dV launch keeping dS, dQ from dS"}[bwd_form] + ")") if bwd else "forward"
    shard = (f" = rank's shard of BASELINE config 5 (B=64 over 8 GPUs; here global B={B * world} over {world})" if cfg_name == "cfg5" else "")
    return (f"{cfg_name}: FlashAttention-2 {what}, per-GPU B={B} H={H} N={N} D={D} {'causal' if causal else 'full'}{shard}, "
            f"q/k/v normal(0,0.5) resident in HBM, scale=1/sqrt(D)")


# the sources that define what the headline kernel executes: a change to any of them makes a recorded HBM-traffic figure stale
KERNEL_SOURCES = ["tfa_fwd_kernel.h", "tfa_fwd_kernel_dma.h", "tfa_fwd_kernel_il.h", "tfa_fwd_il_regs.h", "tfa_fwd_il_pass_prologue.inc",
                  "tfa_fwd_il_tile_loop.inc", "tfa_fwd_il_asm_loop.inc", "tfa_fwd_il_epilogue.inc", "tfa_fwd_inst.inc", "tfa_launch.h", "tfa_api.hip", "Makefile"]


def kernel_sources_sha256():
    """SHA-256 over the forward kernel's sources (csrc/: KERNEL_SOURCES, in that order).  (Not over libtfa_hip.so: hipcc's objects are not
    bit-reproducible, so a rebuild of unchanged sources would look like a new kernel.)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "tiny-flash-attention_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def hbm_traffic_for(cfg_name, variant, bwd):
    """HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC passes perturb timing and run separately:
    tools/prof_pmc.py -> tools/update_hbm_traffic.py -> profiles/hbm_traffic.json).  Every entry is stamped with the SHA-256 of the
    kernel sources it was measured on (and the git head); sources that changed since give traffic = null and say why."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if variant >= 0 or bwd or cfg_name not in tj:
            return None, None
        e = tj[cfg_name]
        have = kernel_sources_sha256()
        if e.get("kernel_sources_sha256") != have:
            return None, (f"stale: profiles/hbm_traffic.json[{cfg_name}] was measured on kernel sources {str(e.get('kernel_sources_sha256'))[:12]} "
                          f"({e.get('git_head', '?')}), this tree has {have[:12]} — re-run tools/prof_pmc.py + tools/update_hbm_traffic.py")
        return e["bytes"], ("static: profiles/hbm_traffic.json (" + e.get("source", "rocprofv3 --pmc passes of this command") +
                            f"), measured on these very kernel sources ({have[:12]}, {e.get('git_head', '?')}), not in this run")
    except Exception as ex:
        return None, f"unavailable: {ex!r}"


def fake_run(args, world, rank, cfg_name, do_gather):
    """--fake-step-ms: bench.py's multi-rank control flow on CPU tensors over gloo (tests/test_dist_cpu.py).  The HIP launch is a sleep;
    everything a scaling run depends on — the configuration chosen for this world size, barriers, max over ranks, whole-job aggregation, the
    gather leg through dist.OverlappedGather, the JSON fields — is the real code path's logic.  Never a measurement."""
    import torch.distributed as dist
    from tiny_flash_attention_amd import dist as tdist   # (imports torch only: no HIP library needed for the schedule itself)

    if world > 1 or do_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
    B, H, N, D, dtype, causal = CONFIGS[cfg_name]
    sc = 1.0 / math.sqrt(D)
    flops_step_rank = 4.0 * B * H * N * N * D * (0.5 if causal else 1.0)

    def barrier():
        if dist.is_initialized():
            dist.barrier()

    def region(fn, join=None):
        for _ in range(args.warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        if join:
            join()
        barrier()
        wall = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([wall], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall

    step = lambda: time.sleep(args.fake_step_ms * 1e-3)
    wall = region(step)
    total = flops_step_rank * world * args.steps
    line = {"metric": f"fwd TFLOPS + achieved %MFMA-roofline, (B={B * world},H={H},N={N},D={D}) bf16", "value": total / wall / 1e12, "unit": "TFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "fake (control-flow test: the step is a CPU sleep, the collective is gloo)",
            "config": {"workload": workload_text(cfg_name, world), "global_batch": B * world, "per_gpu_batch": B,
                       "parallelism": f"batch-sharded x{world}, no data-path collective"}}
    if do_gather:
        # tiny CPU stand-ins with the real batch dimension, so that the chunking over B is the real one
        q = torch.zeros((B, 1, 2, 8), dtype=torch.bfloat16)
        fn = lambda a, b_, c, cz, s_, o: (time.sleep(args.fake_step_ms * 1e-3 * a.shape[0] / B), o.fill_(float(rank + 1)))[1]
        gth = tdist.OverlappedGather(q, q, q, causal, sc, world, rank, chunks=args.gather_chunks, fn=fn)
        wall_g = region(gth.step, gth.join)
        res = gth.result()                          # zero-copy view over the per-chunk gather buffers
        ok = len(res) == world * B and all(bool((t == float(row // B + 1)).all()) for row, t in res)
        line["gather"] = {"value": total / wall_g / 1e12, "unit": "TFLOP/s", "ms_per_step": wall_g / args.steps * 1e3, "chunks": gth.nchunks,
                          "gathered_rows_ok": ok}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return line


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _self_spawn(n):
    """`python bench.py --gpus N` outside torchrun: run the same command line as N ranks of one node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: cfg3 (the headline) on one GPU, the per-GPU shard of cfg5 (B=8) on several")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (-1 = library default)")
    ap.add_argument("--gather", action="store_true", help="one GPU: also time the step + all-gather of the output (world 1 communicator); "
                                                          "several GPUs: always on unless --no-gather")
    ap.add_argument("--no-gather", action="store_true", help="several GPUs: skip the gather leg (kernel-only number alone)")
    ap.add_argument("--fake-step-ms", type=float, default=0.0,
                    help="TEST HOOK (tests/test_dist_cpu.py): replace the HIP launch by a CPU sleep of this many ms and RCCL by gloo, so that the "
                         "multi-rank control flow (config choice, barriers, max over ranks, both legs, the JSON line) runs without GPUs; "
                         "the line says data: fake")
    ap.add_argument("--gather-chunks", type=int, default=4, help="batch chunks of the overlapped gather")
    ap.add_argument("--precondition-s", type=float, default=2.0,
                    help="untimed, disclosed clock/power pre-conditioning before the warm-up steps (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other configs, decode, backward: ~8 s)")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "bwd"],
                    help="fwd (default, the BASELINE metric) or bwd: a step is one tfa_bwd call")
    ap.add_argument("--bwd-form", default="default", choices=["default", "split", "workspace"],
                    help="bwd mode: tfa_bwd's form — default (7 GEMM units), split (dK and dV as two launches: round 2's 8 units), "
                         "workspace (dS kept in scratch memory: 5 units)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(_self_spawn(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    fake = args.fake_step_ms > 0
    if not fake and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the hot path)")

    # one GPU: the headline (config 3).  Several: every rank runs its shard of config 5 (B=64 over 8 GPUs -> B=8 per GPU)
    cfg_name = args.config or ("cfg3" if world == 1 else "cfg5")
    B, H, N, D, dtype, causal = CONFIGS[cfg_name]
    sc = 1.0 / math.sqrt(D)
    do_gather = args.gather or (world > 1 and not args.no_gather)
    bwd = args.mode == "bwd"

    if fake:
        line = fake_run(args, world, rank, cfg_name, do_gather)
        if rank == 0:
            print(json.dumps(line), flush=True)
        return

    import tiny_flash_attention_amd as tfa   # noqa: F401  (fails loudly if the HIP library is missing)
    from tiny_flash_attention_amd import _lib, ops
    from tiny_flash_attention_amd import dist as tdist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or do_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    mk = lambda: torch.empty((B, H, N, D), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(dtype)
    q, k, v = mk(), mk(), mk()                     # reference input recipe (test.py:13-17), resident in HBM
    out = torch.empty_like(q)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=dev)

    _lib.set_variant(args.variant)
    p = ops.make_params(q, k, v, out, lse, causal, sc)
    L = _lib.lib()
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    pref = C.byref(p)
    if bwd:
        _lib.check(L.tfa_fwd(pref, sptr))                      # out, lse of this q,k,v
        dout = mk()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        pb = ops.make_bwd_params(q, k, v, out, lse, dout, dq, dk, dv, delta, causal, sc)
        if args.bwd_form == "workspace":
            bwd_ws = torch.empty((max(16, ops.bwd_workspace_bytes(pb)),), dtype=torch.uint8, device=dev)
            pb.workspace, pb.workspace_bytes = bwd_ws.data_ptr(), bwd_ws.numel()
        _lib.debug_bwd_split(args.bwd_form == "split")
        pbref = C.byref(pb)

    def step():
        if bwd:
            _lib.check(L.tfa_bwd(pbref, sptr))
        else:
            _lib.check(L.tfa_fwd(pref, sptr))

    def shader_clock_mhz():
        """one traced launch: median over workgroups of (s_memtime cycles) / (s_memrealtime 100 MHz ticks)"""
        if bwd:
            return None
        try:
            g_, b_, l_ = C.c_int(), C.c_int(), C.c_int()
            _lib.check(L.tfa_fwd_plan(pref, C.byref(g_), C.byref(b_), C.byref(l_)))
            tb = torch.zeros((g_.value, 8), dtype=torch.int64, device=dev)
            L.tfa_debug_set_trace(C.c_void_p(tb.data_ptr()))
            try:
                _lib.check(L.tfa_fwd(pref, sptr))
                torch.cuda.synchronize()
            finally:
                L.tfa_debug_set_trace(None)
            tr = tb.cpu()
            cyc, ticks = (tr[:, 3] - tr[:, 0]).double(), tr[:, 6].double()
            ok = ticks > 0
            return float((cyc[ok] / ticks[ok] * 100.0).median()) if bool(ok.any()) else None
        except Exception:
            return None

    def timed_chunk(n, fn=None):
        fn = fn or step
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n):
            fn()
        b.record(stream)
        b.synchronize()
        return a.elapsed_time(b) / n

    def timed_region(fn, join=None):
        """W warm-up steps, barrier + synchronize, EXACTLY K timed steps between HIP events and a host timer, barrier + synchronize,
        max over ranks.  Returns (wall seconds of the K steps, event ms per step on this rank)."""
        for _ in range(args.warmup):
            fn()
        if join is not None:
            join()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        # (the K launches are enqueued ~50x faster than they execute; a host stall longer than that head start opens a gap in
        #  the stream and lands in `value` — seen once in round 2: 962 TF with every kernel at its usual 0.525 ms.  The garbage
        #  collector is the one avoidable source; one replay of a HIP graph holding the K launches was tried and is 2.5 % SLOWER
        #  than the loop: the runtime leaves a gap between graph kernel nodes.)
        gc_was = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            fn()
        if join is not None:
            join()                                 # the side stream's last gather belongs to the timed region
        e1.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        ev = e0.elapsed_time(e1) / args.steps
        if dist is not None:
            tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall, ev

    # ---- 1. pre-conditioning (untimed, disclosed) -----------------------------------------------------------
    precond = {"seconds": 0.0, "launches": 0}
    if args.precondition_s > 0:
        step()
        torch.cuda.synchronize()                   # first-launch one-offs (module load, LDS opt-in) stay out of the rates
        clk0 = shader_clock_mhz() if rank == 0 else None
        t_begin = time.perf_counter()
        first = last = timed_chunk(10)
        n_launch = 10
        while time.perf_counter() - t_begin < args.precondition_s:
            last = timed_chunk(50)
            n_launch += 50
        clk1 = shader_clock_mhz() if rank == 0 else None
        precond = {"seconds": time.perf_counter() - t_begin, "launches": n_launch,
                   "ms_per_launch_first10": first, "ms_per_launch_last50": last,
                   "shader_clock_mhz_before": clk0, "shader_clock_mhz_after": clk1}

    # ---- 2. warm-up, 3. the timed region (kernel only: `value`) ----------------------------------------------
    wall, ev_ms = timed_region(step)

    # ---- 4. per-launch durations (one event pair per launch) -------------------------------------------------
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    evs[0].record(stream)
    for i in range(args.steps):
        step()
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    per_launch = {"median": statistics.median(per), "min": min(per), "max": max(per), "mean": sum(per) / len(per), "n": len(per)}

    # ---- 5. the reference's harness: host timer, synchronize after every call (test.py:30-40) ----------------
    torch.cuda.synchronize()
    th = time.time()
    for _ in range(args.steps):
        step()
        torch.cuda.synchronize()
    ref_harness_ms = (time.time() - th) * 1e3 / args.steps

    # ---- 6. sustained shader clock right behind the measurements ----------------------------------------------
    clk_mhz = None
    if rank == 0 and not bwd:
        for _ in range(20):
            step()
        clk_mhz = shader_clock_mhz()

    # ---- 6a. what nothing but MFMAs sustains on this box, on this q tensor's values (2 s, rank 0, one GPU): quoted beside the nominal peak
    mfma_ceiling = None
    if rank == 0 and world == 1 and not bwd and dtype == torch.bfloat16:
        try:
            tf_ = C.c_double()
            _lib.check(L.tfa_debug_mfma_ceiling(C.c_void_p(q.data_ptr()), C.c_ulonglong(q.numel() * q.element_size()), C.c_double(2.0), sptr, C.byref(tf_)))
            mfma_ceiling = tf_.value
        except Exception:
            mfma_ceiling = None

    fl, by = C.c_double(), C.c_double()
    if bwd:
        L.tfa_bwd_work(pbref, C.byref(fl), C.byref(by))
    else:
        L.tfa_fwd_work(pref, C.byref(fl), C.byref(by))
    flops_step_rank = fl.value
    total_flops = flops_step_rank * world * args.steps
    value = total_flops / wall / 1e12
    achieved = flops_step_rank / (ev_ms * 1e-3) / 1e12

    # ---- 6b. the gather leg: the same K steps, each followed by the chunked, overlapped all-gather of the output shards -------
    gather_line = None
    if do_gather and not bwd:
        gatherer = tdist.OverlappedGather(q, k, v, causal, sc, world, rank, chunks=args.gather_chunks)
        wall_g, ev_g = timed_region(gatherer.step, gatherer.join)
        gather_line = {
            "value": total_flops / wall_g / 1e12, "unit": "TFLOP/s", "ms_per_step": wall_g / args.steps * 1e3, "launch_ms": ev_g,
            "chunks": gatherer.nchunks,
            "gathered_bytes_per_rank_per_step": int(out.numel() * out.element_size() * world),
            "what": "step = forward of this rank's batch slab in batch chunks + one all_gather_into_tensor per chunk on a side stream "
                    "(chunk c's gather overlaps chunk c+1's kernel; the last gather is exposed); value = the same algorithmic flops / this time",
        }
        del gatherer

    if rank == 0:
        grid, block, ldsb = C.c_int(), C.c_int(), C.c_int()
        L.tfa_fwd_plan(pref, C.byref(grid), C.byref(block), C.byref(ldsb))
        traffic, traffic_source = hbm_traffic_for(cfg_name, args.variant, bwd)
        tf = lambda ms: flops_step_rank / (ms * 1e-3) / 1e12
        line = {
            "metric": ("bwd TFLOPS (2.5 x fwd flops)" if bwd else "fwd TFLOPS") + f" + achieved %MFMA-roofline, (B={B * world},H={H},N={N},D={D}) "
                      + ("bf16" if dtype == torch.bfloat16 else "fp16"),
            "value": value,
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if dtype == torch.bfloat16 else "f16",
            "data": "synthetic",
            "config": {
                "workload": workload_text(cfg_name, world, bwd, args.bwd_form),
                "global_batch": B * world,
                "per_gpu_batch": B,
                "parallelism": f"batch-sharded x{world}, no data-path collective" + (" (gather leg reported separately under `gather`)" if gather_line else ""),
                "kernel_variant": _lib.variant_name(args.variant) if args.variant >= 0 else _lib.variant_name(L.tfa_fwd_variant(pref)),
                "launch": {"grid": grid.value, "block": block.value, "lds_bytes": ldsb.value},
                "flops_per_step_per_gpu": flops_step_rank,
                "algorithmic_bytes_per_step_per_gpu": by.value,
            },
            "box": box_info(),
            "preconditioning": precond,
            "per_launch_ms": per_launch,
            "per_launch_tflops": {"median": tf(per_launch["median"]), "best": tf(per_launch["min"])},
            "reference_harness_ms": ref_harness_ms,
            "reference_harness_tflops": tf(ref_harness_ms),
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_TFLOPS_BF16,
                "unit": "TFLOP/s",
                # ONE number for the judged fraction: the per-GPU share of `value` (the host-timed, barrier-bracketed K steps) over the nominal peak;
                # the HIP-event clock over the same K launches (`achieved`, `launch_ms`) stays beside it as `frac_event_clock` (the two differ by < 0.5 %)
                "frac": value / world / PEAK_TFLOPS_BF16,
                "frac_event_clock": achieved / PEAK_TFLOPS_BF16,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "launch_ms": ev_ms,
                "nominal_clock_mhz": 2400.0,
                "sustained_clock_mhz": clk_mhz,
                "peak_at_sustained_clock": (PEAK_TFLOPS_BF16 * clk_mhz / 2400.0) if clk_mhz else None,
                "frac_at_sustained_clock": (achieved / (PEAK_TFLOPS_BF16 * clk_mhz / 2400.0)) if clk_mhz else None,
                # what nothing-but-MFMA sustains on THIS box on this q tensor's normal(0,0.5) values under the board's power cap — measured in this
                # run (tfa_debug_mfma_ceiling: median of 2 s of launches behind the timed regions; boxes differ by +-5 %): beside the nominal peak, never instead of it
                "mfma_only_ceiling_random_data": mfma_ceiling,
                "mfma_only_ceiling_source": "measured in this run: tfa_debug_mfma_ceiling on the q tensor (csrc/tfa_probe.hip)" if mfma_ceiling else None,
                "frac_of_mfma_only_ceiling": (achieved / mfma_ceiling) if mfma_ceiling else None,
                "algorithmic_hbm_GBs": by.value / (ev_ms * 1e-3) / 1e9,
                "hbm_frac": by.value / (ev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            },
        }
        if gather_line is not None:
            line["gather"] = gather_line
        if world == 1 and not bwd and args.variant < 0 and not args.no_secondary:
            line["secondary"] = secondary_measurements(dev)
            # which number belongs to which guarantee (north_star: "matching reference output to rtol=1e-3"): `value` is the default kernel, whose P is
            # rounded against a lazily re-based row reference (parity: the reference's atol 1e-2 and the rigorous 2^-8 * A bound); the SAME configuration
            # with TFA_FWD_EXACT_MAX rounds P at the reference's own points (flash_attention.cu:263-316, main_torch_only.py:240-260) and holds plain
            # element-wise rtol 1e-3 on whole heads (tests/test_parity_gpu.py::test_exact_running_max_flag_at_baseline_sizes)
            ex = line["secondary"].get("cfg3_exact_max", {})
            if cfg_name == "cfg3" and "tflops" in ex:
                line["value_at_reference_rounding_points"] = {
                    "value": ex["tflops"], "unit": "TFLOP/s", "frac": ex["frac"], "ms": ex["ms"], "kernel_variant": ex["kernel_variant"],
                    "flag": "TFA_FWD_EXACT_MAX", "guarantee": "element-wise rtol 1e-3 against the reference's tile loop (where |ref| > 0.05 * A; tests/test_parity_gpu.py)",
                    "guarantee_of_value": "atol 1e-2 (the reference's bar, flash_attention_cutlass/test.py:87) and |d| <= 2^-8 * A against fp64",
                    "ratio_to_value": ex["tflops"] / (value / world) if value else None}
        if world == 1 and not args.no_cpu_baseline and not bwd:
            try:
                line["cpu_baseline"] = cpu_baseline(B, H, N, D, causal)
            except Exception as e:  # the baseline is a report, not the product: never fail the bench on it
                line["cpu_baseline"] = {"value": None, "unit": "TFLOP/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"unavailable: {e!r}"}
            try:
                line["cpu_baseline_python"] = cpu_baseline_python()
            except Exception as e:
                line["cpu_baseline_python"] = {"value": None, "unit": "TFLOP/s", "cores": os.cpu_count(), "kind": "port",
                                               "sample": f"unavailable: {e!r}"}
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
