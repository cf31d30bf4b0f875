"""Batch x head sharding of the forward pass across GPUs (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm).

Every (b,h) pair is an independent attention problem — the reference's grid already treats
blockIdx.y = b*H + h as independent (flash_attention_cutlass/csrc/flash_attention.cu:382,409,698) —
so the path shards with NO data-path collective: each rank runs the kernel on a contiguous slab
of the batch (or of the flattened batch*head axis when B < world).  The only optional collective
is the "trivial gather" of the output slabs (all_gather_into_tensor).
"""
import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n, world, rank):
    """Contiguous [lo, hi) of `n` units for `rank` of `world` (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_axis(B, H, world):
    """'batch' when every rank can get whole batches, else 'bh' (flattened batch*head units)."""
    return "batch" if B >= world and B % world == 0 else "bh"


def local_shard(t, world, rank, axis):
    """Zero-copy view of this rank's slab of a contiguous (B,H,N,D) tensor.
    axis='batch' -> (B/world, H, N, D); axis='bh' -> (1, n_local, N, D)."""
    B, H = t.shape[0], t.shape[1]
    if axis == "batch":
        lo, hi = shard_bounds(B, world, rank)
        return t[lo:hi]
    flat = t.reshape(1, B * H, *t.shape[2:])
    lo, hi = shard_bounds(B * H, world, rank)
    return flat[:, lo:hi]


def local_shard_gqa(q, k, v, world, rank):
    """This rank's slab of a GQA problem on the flattened (batch x K/V head) axis: whole K/V heads with all of their query heads.
    q (B,H,N,D), k/v (B,Hk,Nk,D), contiguous -> views (1, n*G, N, D), (1, n, Nk, D), (1, n, Nk, D)."""
    B, H, Hk = q.shape[0], q.shape[1], k.shape[1]
    G = H // Hk
    lo, hi = shard_bounds(B * Hk, world, rank)
    qf = q.reshape(1, B * H, *q.shape[2:])[:, lo * G:hi * G]
    kf = k.reshape(1, B * Hk, *k.shape[2:])[:, lo:hi]
    vf = v.reshape(1, B * Hk, *v.shape[2:])[:, lo:hi]
    return qf, kf, vf


def _gather_slabs(out_l, world, group):
    """all-gather of equal contiguous slabs into ONE contiguous buffer (world, *slab): all_gather_into_tensor, not the
    list form (which stages through per-rank tensors)."""
    full = torch.empty((world,) + tuple(out_l.shape), dtype=out_l.dtype, device=out_l.device)
    dist.all_gather_into_tensor(full.view(-1), out_l.contiguous().view(-1), group=group)
    return full


def sharded_forward_local(q_l, k_l, v_l, is_causal, softmax_scale, group=None, gather=True, fn=None):
    """The multi-GPU entry for callers whose tensors are ALREADY sharded: every rank passes ITS batch slab
    q_l (B_l,H,N,D), k_l / v_l (B_l,Hk,Nk,D) — no rank ever holds the full tensors — runs the kernel on it (GQA as it stands:
    a batch slab keeps every K/V head whole) and, with gather=True, returns the (world*B_l, H, N, D) output of all ranks
    (equal slabs; one all_gather_into_tensor), else its own slab.  The partitioning is the reference's own independence
    of (b,h) problems (flash_attention_cutlass/csrc/flash_attention.cu:382,409,698): no data-path collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if fn is None:
        fn = lambda a, b_, c, causal, sc: ops.flash_attn_fwd(a, b_, c, causal, sc, return_lse=False)[0]
    out_l = fn(q_l, k_l, v_l, is_causal, softmax_scale)
    if not gather or world == 1:
        return out_l
    full = _gather_slabs(out_l, world, group)
    return full.reshape((world * out_l.shape[0],) + tuple(out_l.shape[1:]))


def sharded_forward(q, k, v, is_causal, softmax_scale, group=None, gather=True, fn=None):
    """q,k,v: the FULL (B,H,N,D) / (B,Hk,Nk,D) tensors, identical on every rank (e.g. regenerated from a seed — the
    synthetic bench; real callers hold shards and use sharded_forward_local).  Each rank computes its slab: whole batches
    when B divides over the ranks, else flattened (batch x head) units — with GQA whole K/V heads together with their query
    heads.  gather=True: every rank returns the full output (equal slabs required), else its local slab.
    `fn(q,k,v,is_causal,scale) -> out` defaults to the HIP operator; tests inject a CPU function."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if fn is None:
        fn = lambda a, b_, c, causal, sc: ops.flash_attn_fwd(a, b_, c, causal, sc, return_lse=False)[0]
    B, H, Hk = q.shape[0], q.shape[1], k.shape[1]
    if H % Hk != 0:
        raise ValueError(f"H={H} is not a multiple of Hk={Hk}")
    axis = shard_axis(B, Hk if Hk != H else H, world)
    if axis == "batch":
        ql, kl, vl = (local_shard(t, world, rank, axis) for t in (q, k, v))
        units = B
    elif Hk == H:
        ql, kl, vl = (local_shard(t, world, rank, axis) for t in (q, k, v))
        units = B * H
    else:
        ql, kl, vl = local_shard_gqa(q, k, v, world, rank)
        units = B * Hk
    out_l = fn(ql.contiguous(), kl.contiguous(), vl.contiguous(), is_causal, softmax_scale)
    if not gather or world == 1:
        return out_l if world > 1 else out_l.reshape(q.shape)
    if units % world != 0:
        raise ValueError("gather needs equal slabs: units % world_size != 0")
    return _gather_slabs(out_l, world, group).reshape(q.shape)


def kv_sharded_forward(q, k_local, v_local, is_causal, softmax_scale, kv_offset, nk_total, group=None,
                       partial_fn=None, merge_fn=None, out_dtype=None):
    """Split-KV across ranks (long context): every rank holds the same queries
    ``q`` (B,H,Nq,D) and ITS chunk ``k_local``/``v_local`` = keys [kv_offset, kv_offset+Nk_local) of a sequence of
    ``nk_total`` keys (equal chunk sizes across ranks).  Each rank computes the partial attention of all queries
    over its chunk (fp32 O + LSE; causal mask against global key positions), the partials are all-gathered
    (the one collective: world x (B*H*Nq*(D+1)) floats per rank over xGMI) and merged locally with the
    online-softmax rule, so every rank returns the full ``(out, lse)``.
    ``partial_fn(q,k,v,causal,scale,kv_offset,nk_total) -> (o32, lse)`` and ``merge_fn(o_parts, lse_parts, dtype)``
    default to the HIP operators; the CPU tests inject oracle functions."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if partial_fn is None:
        partial_fn = lambda a, b_, c, cz, sc, off, tot: ops.flash_attn_fwd(a, b_, c, cz, sc, out_f32=True, kv_offset=off, nk_total=tot)
    if merge_fn is None:
        merge_fn = ops.merge_partials
    out_dtype = q.dtype if out_dtype is None else out_dtype
    o_l, l_l = partial_fn(q, k_local, v_local, is_causal, softmax_scale, kv_offset, nk_total)
    o_l, l_l = o_l.contiguous(), l_l.contiguous()
    if world == 1:
        return merge_fn(o_l.unsqueeze(0), l_l.unsqueeze(0), out_dtype)
    o_all = torch.empty((world,) + tuple(o_l.shape), dtype=o_l.dtype, device=o_l.device)
    l_all = torch.empty((world,) + tuple(l_l.shape), dtype=l_l.dtype, device=l_l.device)
    dist.all_gather_into_tensor(o_all.view(-1), o_l.view(-1), group=group)
    dist.all_gather_into_tensor(l_all.view(-1), l_l.view(-1), group=group)
    return merge_fn(o_all, l_all, out_dtype)


class Gathered:
    """Zero-copy view of an OverlappedGather's output: rank r's batch b is global row r*B + b; chunk c holds the batch rows bounds[c] of every rank."""

    def __init__(self, parts, bounds, world, B):
        self.parts, self.bounds, self.world, self.B = parts, bounds, world, B

    def slab(self, r):
        """rank r's (B, H, N, D) output as a list of (lo, hi, view) — one view per chunk, no copy"""
        return [(lo, hi, self.parts[c][r]) for c, (lo, hi) in enumerate(self.bounds)]

    def batch(self, r, b):
        """rank r's batch element b: an (H, N, D) view"""
        for c, (lo, hi) in enumerate(self.bounds):
            if lo <= b < hi:
                return self.parts[c][r, b - lo]
        raise IndexError(b)

    def __len__(self):
        return self.world * self.B

    def __iter__(self):
        """(global row, (H, N, D) view) in global row order"""
        for r in range(self.world):
            for b in range(self.B):
                yield r * self.B + b, self.batch(r, b)

    @property
    def shape(self):
        """shape of the assembled tensor: (world*B, H, N, D)"""
        return (self.world * self.B,) + tuple(self.parts[0].shape[2:])

    def __getitem__(self, idx):
        """Tensor-style indexing over the global batch rows, for callers written against the assembled tensor (`full[r*B:(r+1)*B]`): an int or a
        contiguous slice that stays inside one rank's rows of one chunk is a zero-copy VIEW of the gather buffer; anything else indexes `.cat()`
        (a copy of the whole output)."""
        n = len(self)
        if isinstance(idx, int):
            if not -n <= idx < n:
                raise IndexError(idx)
            r, b = divmod(idx % n, self.B)
            return self.batch(r, b)
        if isinstance(idx, slice):
            start, stop, step = idx.indices(n)
            if step == 1 and start < stop:
                (r0, b0), (r1, b1) = divmod(start, self.B), divmod(stop - 1, self.B)
                if r0 == r1:
                    for c, (lo, hi) in enumerate(self.bounds):
                        if lo <= b0 and b1 < hi:
                            return self.parts[c][r0, b0 - lo:b1 - lo + 1]
        return self.cat()[idx]

    def cat(self):
        """the assembled (world*B, H, N, D) tensor — a COPY when there is more than one chunk"""
        full = self.parts[0] if len(self.parts) == 1 else torch.cat(self.parts, dim=1)
        return full.reshape((self.world * self.B,) + tuple(full.shape[2:]))


class OverlappedGather:
    """Forward over this rank's batch slab + the "trivial gather" of all ranks' outputs, with the gather hidden behind
    the compute ("overlap the gather with compute by chunking over B").

    The slab is cut into `chunks` contiguous batch chunks.  `step()` launches chunk c's kernel on the current stream and
    queues chunk c's all-gather on a side stream behind an event, so that RCCL moves chunk c over xGMI while chunk c+1
    computes; only the last chunk's gather is exposed.  (b,h) problems are independent
    (flash_attention_cutlass/csrc/flash_attention.cu:382,409,698), so chunking changes no result bit.  Each chunk is gathered
    with ONE all_gather_into_tensor into its own contiguous buffer `parts[c]` (world, rows, H, N, D); `result()` joins and returns
    a zero-copy `Gathered` view of those buffers (rank r's batch b = global row r*B + b; `.cat()` / the `full` property assemble
    one tensor — a copy); `join()` makes the current stream wait for the outstanding gathers.
    `fn(q,k,v,is_causal,scale,out) -> None` defaults to the HIP operator writing into `out`; `device='cpu'` tensors
    (the gloo tests) run the same schedule without streams."""

    def __init__(self, q, k, v, is_causal, softmax_scale, world, rank, chunks=4, group=None, fn=None):
        self.q, self.k, self.v = q, k, v
        self.causal, self.scale = is_causal, softmax_scale
        self.world, self.rank, self.group = world, rank, group
        B = q.shape[0]
        self.nchunks = max(1, min(int(chunks), B))
        self.bounds = [shard_bounds(B, self.nchunks, c) for c in range(self.nchunks)]
        self.out = torch.empty_like(q)
        # one CONTIGUOUS (world, chunk rows, H, N, D) buffer per chunk: a chunk's gather is a single all_gather_into_tensor
        self.parts = [torch.empty((world, hi - lo) + tuple(q.shape[1:]), dtype=q.dtype, device=q.device) for lo, hi in self.bounds]
        self.cuda = q.is_cuda
        if fn is None:
            fn = lambda a, b_, c, causal, sc, o: ops.flash_attn_fwd(a, b_, c, causal, sc, return_lse=False, out=o)
        self.fn = fn
        if self.cuda:
            self.side = torch.cuda.Stream(device=q.device)
            # allocated on the caller's stream, used on the side stream (and RCCL's): tell the caching allocator, so that
            # dropping this object while gathers are in flight cannot hand the memory out early
            self.out.record_stream(self.side)
            for t in self.parts:
                t.record_stream(self.side)
            self.computed = [torch.cuda.Event() for _ in self.bounds]
            self.gathered = [None for _ in self.bounds]      # event of the chunk's last gather (its out slab is being read)

    def _gather(self, c, lo, hi):
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.parts[c][0].copy_(self.out[lo:hi])
            return
        dist.all_gather_into_tensor(self.parts[c].view(-1), self.out[lo:hi].reshape(-1), group=self.group)

    def result(self):
        """The gathered output of all ranks after join(), WITHOUT copying it: a `Gathered` view over the per-chunk buffers the collectives wrote
        (`parts[c]`, shape (world, rows of chunk c, H, N, D)).  `res.slab(r)` / `res.batch(r, b)` / iteration hand out views; `res.cat()` — and
        the `full` property — assemble the (world*B, H, N, D) tensor, which is a copy of the whole output (2.1 GB per call at BASELINE config 5 on
        8 ranks) and is there for callers that really need one contiguous tensor.  `Gathered` also answers `shape`, `len()` and tensor-style
        `res[i]` / `res[a:b]` over the global batch rows (views where the rows lie in one buffer, else an index into `.cat()`), so that callers written
        against the assembled tensor of the earlier versions keep working."""
        self.join()
        return Gathered(self.parts, self.bounds, self.world, self.q.shape[0])

    @property
    def full(self):
        return self.result().cat()

    def step(self):
        if not self.cuda:
            for c, (lo, hi) in enumerate(self.bounds):
                self.fn(self.q[lo:hi], self.k[lo:hi], self.v[lo:hi], self.causal, self.scale, self.out[lo:hi])
                self._gather(c, lo, hi)
            return
        main = torch.cuda.current_stream(self.q.device)
        for c, (lo, hi) in enumerate(self.bounds):
            if self.gathered[c] is not None:
                main.wait_event(self.gathered[c])            # the previous step's gather still reads this out slab
            self.fn(self.q[lo:hi], self.k[lo:hi], self.v[lo:hi], self.causal, self.scale, self.out[lo:hi])
            self.computed[c].record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.computed[c])
                self._gather(c, lo, hi)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.gathered[c] = ev

    def join(self):
        if self.cuda:
            torch.cuda.current_stream(self.q.device).wait_stream(self.side)
