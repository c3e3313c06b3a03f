"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

Shapes, as SURVEY.md section 8e lays out:
  * batches of independent proofs / signatures: replicas -- `shard_range` splits the item index range, no exchange step,
    the caller concatenates per-rank result arrays (`gather_results`);
  * one large multi-scalar multiplication: term sharding -- every rank runs a complete bucket MSM on its slice of the terms
    and emits ONE Jacobian partial (28 uint32: x, y, z limbs + infinity flag).  EC addition is not an RCCL reduction
    operator, so the collective is an all-gather of the raw limb buffers followed by a local tree sum
    (`s2k_gej_sum_dev`).  Payload: world_size x 112 bytes -- latency, not bandwidth, is what it costs.
  * the same sum with the bucket *windows* sharded (BASELINE config 5 as worded): every rank holds all terms and owns a
    contiguous share of the signed-digit windows; its partial is  sum_{w in share} 2^(c w) S_w , so the exchange and the final
    sum are exactly those of term sharding (`msm_window_sharded`).  Which is faster depends on n and on the rank count:
    term sharding divides all per-term work by the world size, window sharding repeats the decode / GLV split on every rank
    but needs no slicing of the inputs and gives each rank whole windows; `msm_auto` picks by a size rule.
  * K independent sums (`s2k_ecmult_multi_many`): independent objects again -- `shard_sums` cuts them into contiguous ranges of about
    equal numbers of TERMS, every rank runs its range as one launch chain, and the only exchange is the gather of the K x 68 result
    bytes (`msm_many_sharded`).
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous [lo, hi) slice of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sums(offsets, world):
    """K sums with terms back to back (offsets[K + 1]) -> cut[world + 1]: rank r takes sums [cut[r], cut[r + 1]), contiguous, about equal
    numbers of terms (a rank's range ends with the first sum whose end passes its share of the terms)."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    k = off.size - 1
    n = int(off[-1]) if off.size else 0
    cut = [0] + [int(np.searchsorted(off[1:], (n * (r + 1)) // world, side="left")) + 1 if r + 1 < world else k for r in range(world)]
    cut = [min(max(c, 0), max(k, 0)) for c in cut]
    for i in range(1, len(cut)):
        cut[i] = max(cut[i], cut[i - 1])
    return cut


class _NoStream:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def _chain(backend):
    """the stream context in which partial -> all-gather -> sum form ONE stream-ordered chain (EngineBackend: its own torch stream, which
    the RCCL collective then also runs on; CPU test backends: nothing)"""
    return backend.chain() if hasattr(backend, "chain") else _NoStream()


def msm_sharded(backend, sc, pt_xy, g_sc=None, pt_inf=None, group=None, to_host=True):
    """r = g_sc*G + sum sc_i*pt_i with the terms sharded over the ranks of `group`.

    `backend` provides  msm_partial(sc, pt_xy, g_sc, pt_inf) -> torch uint32[28] (on its device)  and
    gej_sum(parts uint32[world,28], to_host) -> (xy bytes[64], inf).  On a GPU rank that is `EngineBackend(engine)`.
    Every rank passes the full input (already resident); each computes only its slice.  Returns (xy, inf) on all ranks; with
    to_host=False the two are device tensors and NOTHING waits for the GPU (the caller synchronises when it needs the values)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = sc.shape[0] if hasattr(sc, "shape") and len(sc.shape) > 1 else len(sc) // 32
    lo, hi = shard_range(n, rank, world)
    if world == 1 and hasattr(backend, "msm_whole"):
        with _chain(backend):                                           # (on the backend's stream like the sharded chain: no cross-stream hops per call)
            return backend.msm_whole(sc, pt_xy, g_sc, pt_inf, to_host)  # nothing to exchange: the engine's complete call (no partial / sum stage)
    with _chain(backend):
        part = backend.msm_partial(sc[lo:hi], pt_xy[lo:hi], g_sc if rank == 0 else None, None if pt_inf is None else pt_inf[lo:hi])
        return _gather_and_sum(backend, part, group, to_host)


def _gather_and_sum(backend, part, group, to_host=True):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        parts = part.reshape(1, 28)
    else:
        bufs = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(bufs, part, group=group)
        parts = torch.stack(bufs)
    return backend.gej_sum(parts, to_host) if _takes_to_host(backend) else backend.gej_sum(parts)


def _takes_to_host(backend):
    import inspect
    try:
        return "to_host" in inspect.signature(backend.gej_sum).parameters
    except (TypeError, ValueError):
        return False


def msm_window_sharded(backend, sc, pt_xy, g_sc=None, pt_inf=None, group=None, to_host=True):
    """r = g_sc*G + sum sc_i*pt_i with the Pippenger digit windows sharded over the ranks of `group`: rank r computes
    sum_{w in share r} 2^(c w) S_w over ALL terms (`backend.msm_window_partial`), the Jacobian partials are all-gathered as raw
    limb buffers and summed locally.  Every rank passes the same full input.  Returns (xy, inf) on all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    with _chain(backend):
        part = backend.msm_window_partial(sc, pt_xy, g_sc, pt_inf, rank, world)
        return _gather_and_sum(backend, part, group, to_host)


# Terms at which one bucket MSM call stops being pure latency on MI355X (the call costs 0.45-0.75 ms up to 2^16 terms and grows from there:
# profiles/r03*_msm_sweep.txt); a backend may carry its own measured value as `floor_terms`.
MSM_FLOOR_TERMS = 1 << 16


def msm_auto(backend, sc, pt_xy, g_sc=None, pt_inf=None, group=None, to_host=True):
    """term sharding unless the per-rank slice would fall under the size at which a call is pure latency (each rank would then pay the
    whole floor for a sliver of the terms) while whole windows are still available to hand out."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n = sc.shape[0] if hasattr(sc, "shape") and len(sc.shape) > 1 else len(sc) // 32
    floor_terms = getattr(backend, "floor_terms", MSM_FLOOR_TERMS)
    if world > 1 and n // world < floor_terms and n >= (1 << 12):
        return msm_window_sharded(backend, sc, pt_xy, g_sc, pt_inf, group, to_host)
    return msm_sharded(backend, sc, pt_xy, g_sc, pt_inf, group, to_host)


def msm_many_sharded(backend, sc, pt_xy, offsets, g_sc=None, pt_inf=None, group=None):
    """K independent sums (terms back to back, host array offsets[K + 1], optional g_sc[K, 32]) sharded over the ranks of `group` as objects.

    `backend.msm_many(sc, pt_xy, offsets, g_sc, pt_inf) -> (xy uint8[k, 64], inf int32[k])` runs a range of sums (EngineBackend: one
    s2k_ecmult_multi_many_dev launch chain).  Every rank passes the full input and computes only its range; the results are gathered
    (K x 68 bytes: the path's only exchange) and returned on all ranks as (xy uint8[K, 64], inf int32[K]) tensors on the backend's device."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    k = off.size - 1
    cut = shard_sums(off, world)
    a, b = cut[rank], cut[rank + 1]
    t0, t1 = (int(off[a]), int(off[b])) if k > 0 else (0, 0)
    with _chain(backend):
        xy, inf = backend.msm_many(sc[t0:t1], pt_xy[t0:t1], off[a:b + 1] - (off[a] if k > 0 else 0), None if g_sc is None else g_sc[a:b],
                                   None if pt_inf is None else pt_inf[t0:t1])
        if world == 1:
            return xy, inf
        # one buffer per rank: 64 result bytes + the flag as 4 bytes, padded to the longest range
        rec = torch.cat([xy.reshape(-1, 64), inf.reshape(-1, 1).to(torch.int32).contiguous().view(torch.uint8).reshape(-1, 4)], dim=1)
        longest = max(cut[r + 1] - cut[r] for r in range(world))
        pad = torch.zeros((longest, 68), dtype=torch.uint8, device=rec.device); pad[: rec.shape[0]] = rec
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        full = torch.cat([bufs[r][: cut[r + 1] - cut[r]] for r in range(world)])
        return full[:, :64].contiguous(), full[:, 64:].contiguous().view(torch.int32).reshape(-1)


def gather_results(local, n_total, group=None):
    """concatenate per-rank result arrays of a replica-sharded batch (rank order = index order)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(maxlen, dtype=local.dtype, device=local.device); pad[: local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)])


class EngineBackend:
    """adapter: Engine (HIP) -> the backend interface of msm_sharded; tensors live in HBM of the engine's GPU.

    Stream discipline: the engine's own stream is created non-blocking, i.e. it does NOT order itself against torch's default
    stream, and the collectives (RCCL) order themselves against torch's *current* stream only.  So the engine work is put on a
    dedicated torch stream that first waits for the current stream (inputs ready) and that the current stream then waits for
    (the partial is ready before the all-gather reads it).  Passing stream 0 to a *_dev entry point means "the engine's stream",
    never torch's default stream."""

    def __init__(self, engine):
        import torch
        self.engine = engine
        self.dev = torch.device("cuda", engine.device)
        self.stream = torch.cuda.Stream(device=self.dev)

    def chain(self):
        """context: everything inside -- engine calls and torch.distributed collectives alike -- runs on this backend's stream, in order"""
        import torch
        return torch.cuda.stream(self.stream)

    def _enter(self):
        import torch
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
        return self.stream.cuda_stream

    def _leave(self):
        import torch
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def msm_partial(self, sc, pt_xy, g_sc, pt_inf):
        import torch
        out = torch.zeros(28, dtype=torch.int32, device=self.dev)
        if sc.numel() == 0 and g_sc is None:
            out[27] = 1
            return out
        sc = sc.contiguous(); pt_xy = pt_xy.contiguous()
        h = self._enter()
        self.engine.ecmult_multi_partial_dev(out, sc, pt_xy, g_sc, pt_inf, stream=h)
        self._leave()
        return out

    def msm_whole(self, sc, pt_xy, g_sc, pt_inf, to_host=True):
        """the whole sum on this GPU (s2k_ecmult_multi_dev): what a one-rank job runs"""
        import torch
        r = torch.empty(64, dtype=torch.uint8, device=self.dev); inf = torch.empty(1, dtype=torch.int32, device=self.dev)      # (the call writes both, always)
        if sc.numel() == 0 and g_sc is None:
            r.zero_(); inf.fill_(1)
        else:
            sc = sc.contiguous(); pt_xy = pt_xy.contiguous()
            h = self._enter()
            self.engine.ecmult_multi_dev(r, inf, sc, pt_xy, g_sc=g_sc, pt_inf=pt_inf, stream=h)
            self._leave()
        if not to_host:
            return r, inf
        self.stream.synchronize()
        return r.cpu().numpy(), int(inf.item())

    def msm_many(self, sc, pt_xy, offsets, g_sc, pt_inf):
        """a range of independent sums as one launch chain (s2k_ecmult_multi_many_dev); nothing waits for the GPU"""
        import torch
        k = len(offsets) - 1
        r = torch.empty((max(k, 0), 64), dtype=torch.uint8, device=self.dev); inf = torch.empty(max(k, 0), dtype=torch.int32, device=self.dev)
        if k > 0:
            sc = sc.contiguous(); pt_xy = pt_xy.contiguous()
            h = self._enter()
            self.engine.ecmult_multi_many_dev(r, inf, sc, pt_xy, offsets, g_sc=None if g_sc is None else g_sc.contiguous(),
                                              pt_inf=None if pt_inf is None else pt_inf.contiguous(), stream=h)
            self._leave()
        return r, inf

    def msm_window_partial(self, sc, pt_xy, g_sc, pt_inf, part, parts):
        import torch
        out = torch.zeros(28, dtype=torch.int32, device=self.dev)
        sc = sc.contiguous(); pt_xy = pt_xy.contiguous()
        h = self._enter()
        self.engine.ecmult_multi_window_partial_dev(out, sc, pt_xy, part, parts, g_sc=g_sc, pt_inf=pt_inf, stream=h)
        self._leave()
        return out

    def gej_sum(self, parts, to_host=True):
        """sum of Jacobian partials; to_host=False returns the device tensors (xy uint8[64], inf int32[1]) without waiting for anything"""
        import torch
        r = torch.zeros(64, dtype=torch.uint8, device=self.dev); inf = torch.zeros(1, dtype=torch.int32, device=self.dev)
        parts = parts.contiguous()
        h = self._enter()
        self.engine.gej_sum_dev(r, inf, parts, parts.shape[0], stream=h)
        self._leave()
        if not to_host:
            return r, inf
        self.stream.synchronize()
        return r.cpu().numpy(), int(inf.item())
