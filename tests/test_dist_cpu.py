"""Multi-process CPU test of the batch x head sharding (gloo, world_size=2): shard plan, zero-copy
slabs, the gather, and that the assembled result equals the unsharded one.  The compute function
injected here is the CPU oracle (as the checker); on GPUs the default is the HIP operator."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, causal, q_out):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from tiny_flash_attention_amd import dist as tdist

    B, H, Hk, N, D = shape if len(shape) == 5 else (shape[0], shape[1], shape[1], shape[2], shape[3])
    q, k, v = O.make_inputs(B, H, N, D, torch.float32, seed=17, Hk=Hk)   # same seed on every rank
    sc = 1.0 / math.sqrt(D)
    fn = lambda a, b, c, cz, s: O.flash_attn(a, b, c, cz, s)
    full = tdist.sharded_forward(q, k, v, causal, sc, fn=fn, gather=True)
    local = tdist.sharded_forward(q, k, v, causal, sc, fn=fn, gather=False)
    ref = O.flash_attn(q, k, v, causal, sc)
    axis = tdist.shard_axis(B, Hk, world)
    if axis == "batch" or Hk == H:
        ref_local = tdist.local_shard(ref, world, rank, axis)
    else:                                                     # GQA on the (batch x K/V head) axis: whole K/V heads with their query heads
        lo, hi = tdist.shard_bounds(B * Hk, world, rank)
        G = H // Hk
        ref_local = ref.reshape(1, B * H, N, D)[:, lo * G:hi * G]
    ok = bool(torch.equal(full, ref)) and bool(torch.equal(local.reshape(ref_local.shape), ref_local))
    # zero-copy: the slab is a view into the full tensor
    ok = ok and tdist.local_shard(q, world, rank, "batch" if axis == "batch" else "bh").data_ptr() >= q.data_ptr()
    # callers that hold SHARDS (no rank has the full tensors): every rank passes its own batch slab, GQA as it stands
    if B % world == 0:
        lo, hi = tdist.shard_bounds(B, world, rank)
        full2 = tdist.sharded_forward_local(q[lo:hi].contiguous(), k[lo:hi].contiguous(), v[lo:hi].contiguous(), causal, sc, fn=fn, gather=True)
        mine = tdist.sharded_forward_local(q[lo:hi].contiguous(), k[lo:hi].contiguous(), v[lo:hi].contiguous(), causal, sc, fn=fn, gather=False)
        ok = ok and bool(torch.equal(full2, ref)) and bool(torch.equal(mine, ref[lo:hi]))
    q_out.put((rank, ok, axis))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,causal", [((4, 2, 64, 32), True), ((1, 6, 48, 32), False),
                                          ((2, 4, 2, 48, 32), True),      # GQA, batch-sharded (K/V heads stay whole)
                                          ((1, 8, 2, 48, 32), False)])    # GQA, B < world: whole K/V heads with their query heads
def test_sharded_forward_gloo_world2(shape, causal):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, causal, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert {a for _, _, a in res} == {"batch" if shape[0] >= 2 else "bh"}


def test_shard_bounds():
    import sys

    sys.path.insert(0, ROOT)
    from tiny_flash_attention_amd.dist import shard_axis, shard_bounds

    assert [shard_bounds(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]          # BASELINE config 5: B=64 over 8 GPUs
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_axis(64, 32, 8) == "batch" and shard_axis(1, 16, 8) == "bh" and shard_axis(6, 4, 4) == "bh"


def _kv_worker(rank, world, port, shape, causal, q_out):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from tiny_flash_attention_amd import dist as tdist

    B, H, Nq, Nk, D = shape
    q, k, v = O.make_inputs(B, H, Nq, D, torch.float32, seed=23, Nk=Nk)   # same seed on every rank
    sc = 1.0 / math.sqrt(D)
    n = Nk // world
    lo = rank * n
    out, lse = tdist.kv_sharded_forward(q, k[:, :, lo:lo + n], v[:, :, lo:lo + n], causal, sc, lo, Nk,
                                        partial_fn=O.partial_attn, merge_fn=O.merge_partials, out_dtype=torch.float32)
    ref, lse_ref = O.sdpa_reference(q, k, v, causal, sc)
    fin = torch.isfinite(lse_ref)
    ok = (out - ref).abs().max().item() < 2e-6 and bool((torch.isinf(lse) == ~fin).all()) and (lse[fin] - lse_ref[fin]).abs().max().item() < 1e-5
    q_out.put((rank, ok, "kv"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,causal", [((2, 2, 48, 128, 32), True), ((1, 3, 100, 64, 32), False), ((1, 2, 16, 96, 32), True)])
def test_kv_sharded_forward_gloo_world2(shape, causal):
    """split-KV across ranks: partials over each rank's key chunk (global causal positions), all-gather, merge."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kv_worker, args=(r, 2, port, shape, causal, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _og_worker(rank, world, port, shape, causal, chunks, q_out):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from tiny_flash_attention_amd import dist as tdist

    Bl, H, N, D = shape                                           # per-rank batch (weak scaling, as bench.py --gather)
    q, k, v = O.make_inputs(Bl, H, N, D, torch.float32, seed=100 + rank)
    sc = 1.0 / math.sqrt(D)

    def fn(a, b, c, cz, s, o):
        o.copy_(O.flash_attn(a.contiguous(), b.contiguous(), c.contiguous(), cz, s))

    og = tdist.OverlappedGather(q, k, v, causal, sc, world, rank, chunks=chunks, fn=fn)
    og.step()
    og.step()                                                      # a second step re-uses the slabs
    og.join()
    ok = og.nchunks == min(chunks, Bl)
    res = og.result()                                              # zero-copy view over the chunk buffers the collectives wrote
    ok = ok and len(res) == world * Bl
    for r in range(world):                                         # every rank's slab, recomputed locally from its seed
        qr, kr, vr = O.make_inputs(Bl, H, N, D, torch.float32, seed=100 + r)
        want = O.flash_attn(qr, kr, vr, causal, sc)
        ok = ok and bool(torch.equal(og.full[r * Bl:(r + 1) * Bl], want))              # the assembled copy ...
        for lo, hi, view in res.slab(r):                                               # ... and the views: same values, and they ARE the gather buffers
            ok = ok and bool(torch.equal(view, want[lo:hi])) and any(view.data_ptr() == og.parts[c][r].data_ptr() for c in range(og.nchunks))
        ok = ok and all(bool(torch.equal(res.batch(r, b), want[b])) for b in range(Bl))
        # tensor-style indexing (callers written against the assembled tensor): a rank's rows, one row, a slice across ranks — same values; an int is a view
        ok = ok and bool(torch.equal(res[r * Bl:(r + 1) * Bl], want)) and bool(torch.equal(res[r * Bl], want[0])) and res[r * Bl].data_ptr() == res.batch(r, 0).data_ptr()
    ok = ok and tuple(res.shape) == (world * Bl, H, N, D) and bool(torch.equal(res[0:world * Bl], og.full)) and bool(torch.equal(res[-1], og.full[-1]))
    ok = ok and [row for row, _ in res] == list(range(world * Bl))
    q_out.put((rank, ok, "og"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,causal,chunks", [((4, 2, 64, 32), True, 4), ((3, 2, 48, 32), False, 2), ((1, 1, 32, 32), True, 4)])
def test_overlapped_gather_gloo_world2(shape, causal, chunks):
    """bench.py --gather's schedule (batch chunks: kernel c+1 beside gather c) assembles every rank's slab bit-exactly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_og_worker, args=(r, 2, port, shape, causal, chunks, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


# ---- bench.py's multi-rank control flow without GPUs (the step is a sleep, the collective is gloo: --fake-step-ms) -----------------
def _run_bench(argv, world):
    import json
    import subprocess
    import sys

    env = dict(os.environ, PYTHONPATH=ROOT)
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + argv
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print exactly ONE JSON line, got {len(lines)}: {r.stdout[-500:]}"
    return json.loads(lines[0])


def test_bench_two_ranks_run_the_cfg5_shard_and_report_both_legs():
    """`bench.py --gpus 2` as the driver launches it: every rank runs the per-GPU shard of BASELINE config 5 (B=8), the line names it,
    `value` is the whole job's kernel-only rate over the max-over-ranks time and the gather leg sits in the same line."""
    line = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--fake-step-ms", "20"], 2)
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["per_gpu_batch"] == 8 and line["config"]["global_batch"] == 16
    assert "cfg5" in line["config"]["workload"] and "BASELINE config 5" in line["config"]["workload"]
    assert "(B=16,H=32,N=4096,D=128)" in line["metric"]
    per_step = 4.0 * 8 * 32 * 4096 * 4096 * 128 * 0.5
    assert abs(line["value"] - 2 * per_step / (line["ms_per_step"] * 1e-3) / 1e12) <= 1e-6 * line["value"]   # whole-job aggregate
    assert 20.0 <= line["ms_per_step"] <= 60.0
    g = line["gather"]
    assert g["gathered_rows_ok"] is True and g["chunks"] == 4 and g["ms_per_step"] >= line["ms_per_step"] * 0.9
    assert g["value"] <= line["value"] * 1.1


def test_bench_one_rank_keeps_the_headline_config():
    line = _run_bench(["--steps", "3", "--warmup", "1", "--fake-step-ms", "5"], 1)
    assert line["n_gpus"] == 1 and line["config"]["per_gpu_batch"] == 4 and line["config"]["workload"].startswith("cfg3:")
    assert line["metric"] == "fwd TFLOPS + achieved %MFMA-roofline, (B=4,H=32,N=4096,D=128) bf16"      # BASELINE.json's metric, verbatim
    assert "gather" not in line
    line = _run_bench(["--steps", "2", "--warmup", "0", "--fake-step-ms", "5", "--gather"], 1)
    assert line["gather"]["gathered_rows_ok"] is True


def test_bench_self_spawns_its_ranks_outside_torchrun():
    line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--fake-step-ms", "5", "--no-gather"], 1)
    assert line["n_gpus"] == 2 and "gather" not in line and line["config"]["global_batch"] == 16


def test_bench_quotes_hbm_traffic_only_for_the_kernel_sources_it_was_measured_on(monkeypatch):
    """roofline.traffic comes from profiles/hbm_traffic.json, stamped with the SHA-256 of the forward kernel's sources at measurement time
    (tools/update_hbm_traffic.py): a tree whose kernel sources differ gets null and the reason, never a figure measured on another kernel."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    entry = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["cfg3"]
    assert len(entry["kernel_sources_sha256"]) == 64 and entry["bytes"] > 5.39e8          # at least the algorithmic bytes
    monkeypatch.setattr(bench, "kernel_sources_sha256", lambda: entry["kernel_sources_sha256"])
    traffic, src = bench.hbm_traffic_for("cfg3", -1, False)
    assert traffic == entry["bytes"] and src.startswith("static")
    monkeypatch.setattr(bench, "kernel_sources_sha256", lambda: "0" * 64)
    traffic, src = bench.hbm_traffic_for("cfg3", -1, False)
    assert traffic is None and src.startswith("stale")
    assert bench.hbm_traffic_for("cfg3", 30, False) == (None, None)                    # a forced kernel variant: not the measured kernel
