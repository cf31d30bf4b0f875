"""GPU tests of the multi-GPU path that can run on ONE GPU (SURVEY section 8(e) "test without a cluster"):
an RCCL communicator with world_size = 1 under the sharded forward and the overlapped gather, bench.py's --gather
leg and its self-spawn, and a single-process loop over every visible device (per-device LDS opt-in / CU count)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_NCCL_WS1 = r"""
import math, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", %(port)r)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # nccl == RCCL on ROCm
import tiny_flash_attention_amd as tfa
from tiny_flash_attention_amd import dist as tdist
g = torch.Generator(device=dev).manual_seed(3)
mk = lambda: torch.empty((4, 4, 512, 128), dtype=torch.float32, device=dev).normal_(0.0, 0.5, generator=g).to(torch.bfloat16)
q, k, v = mk(), mk(), mk()
sc = 1.0 / math.sqrt(128)
want, _ = tfa.flash_attention_v2_cutlass(q, k, v, True, sc)
full = tdist.sharded_forward(q, k, v, True, sc, gather=True)
assert torch.equal(full, want), "sharded_forward(gather=True) on nccl ws=1"
og = tdist.OverlappedGather(q, k, v, True, sc, 1, 0, chunks=4)
og.step(); og.step(); og.join(); torch.cuda.synchronize()
assert og.nchunks == 4 and torch.equal(og.full, want), "OverlappedGather on nccl ws=1"
t = torch.ones(8, device=dev); dist.all_reduce(t); assert float(t.sum()) == 8.0
dist.barrier(); dist.destroy_process_group()
print("NCCL_WS1_OK")
"""


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_rccl_world_size_one_sharded_forward_and_overlapped_gather():
    r = subprocess.run([sys.executable, "-c", _NCCL_WS1 % {"root": ROOT, "port": _free_port()}], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert r.returncode == 0 and "NCCL_WS1_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gather_leg_on_rccl_world_size_one():
    j = _bench("--gpus", "1", "--steps", "4", "--warmup", "2", "--precondition-s", "0.2", "--no-cpu-baseline", "--gather", "--config", "cfg3")
    # ONE line carries both numbers: `value` = kernel only, `gather` = the step + the chunked, overlapped all-gather (RCCL communicator, world 1)
    assert j["n_gpus"] == 1 and j["steps"] == 4 and "gather" in j["config"]["parallelism"]
    assert j["value"] > 100.0 and j["per_launch_ms"]["n"] == 4
    g = j["gather"]
    assert g["chunks"] == 4 and 0.0 < g["value"] <= j["value"] * 1.05 and g["ms_per_step"] >= j["ms_per_step"] * 0.95
    assert g["gathered_bytes_per_rank_per_step"] == 4 * 32 * 4096 * 128 * 2


def test_bench_contract_fields_and_preconditioning():
    j = _bench("--steps", "5", "--warmup", "2", "--precondition-s", "0.3", "--no-cpu-baseline")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "preconditioning", "per_launch_ms", "reference_harness_ms"):
        assert key in j, key
    assert j["steps"] == 5 and j["warmup"] == 2 and j["dtype"] == "bf16" and j["roofline"]["bound"] == "mfma"
    assert j["preconditioning"]["seconds"] >= 0.3 and j["preconditioning"]["launches"] >= 10
    assert j["per_launch_ms"]["min"] <= j["per_launch_ms"]["median"] <= j["per_launch_ms"]["max"]
    # traffic is quoted only while the library still is the one profiles/hbm_traffic.json was measured on (SHA-256 stamp); else null + reason
    assert ("static" in j["roofline"]["traffic_source"]) if j["roofline"]["traffic"] is not None else ("stale" in j["roofline"]["traffic_source"] or "unavailable" in j["roofline"]["traffic_source"])
    assert "cfg3_exact_max" in j.get("secondary", {"cfg3_exact_max": 0})
    # ONE judged fraction: value / peak (the host-timed K steps); the HIP-event clock over the same launches is a second field
    assert abs(j["roofline"]["frac"] - j["value"] / 2500.0) < 1e-9
    assert abs(j["roofline"]["frac_event_clock"] - j["roofline"]["achieved"] / 2500.0) < 1e-9
    assert abs(j["roofline"]["frac"] / j["roofline"]["frac_event_clock"] - 1.0) < 0.05
    # the number that carries the rtol-1e-3 guarantee stands beside `value` in the line the driver records
    if "secondary" in j and "tflops" in j["secondary"].get("cfg3_exact_max", {}):
        r = j["value_at_reference_rounding_points"]
        assert r["flag"] == "TFA_FWD_EXACT_MAX" and r["value"] == j["secondary"]["cfg3_exact_max"]["tflops"] and 0.5 < r["ratio_to_value"] < 1.1


def test_single_process_loop_over_visible_devices():
    """One process, every visible GPU in turn: each device needs its own dynamic-LDS opt-in (hipFuncSetAttribute is per
    device) and its own CU count; the result must be bit-identical on every device."""
    import tiny_flash_attention_amd as tfa
    from oracle import oracle as O

    q, k, v = O.make_inputs(2, 8, 768, 128, torch.bfloat16, seed=44)
    sc = 1.0 / math.sqrt(128)
    ref = O.exact64(q, k, v, True, sc, p_round=torch.bfloat16)
    first = None
    for d in range(torch.cuda.device_count()):
        dev = torch.device("cuda", d)
        out, lse = tfa.flash_attention_v2_cutlass(q.to(dev), k.to(dev), v.to(dev), True, sc)
        big, _ = tfa.flash_attention_v2_cutlass(*(t.to(dev).repeat(2, 8, 1, 1) for t in (q, k, v)), True, sc)   # 8-wave kernel (128 KiB LDS)
        torch.cuda.synchronize(dev)
        assert (out.float().cpu() - ref).abs().max().item() <= 1e-2
        assert torch.isfinite(big.float()).all()
        first = out.cpu() if first is None else first
        assert torch.equal(out.cpu(), first)
