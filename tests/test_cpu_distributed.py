"""N > 1 path on CPU: world_size-2 `gloo` processes run the sharded-MSM protocol of secp256k1_zkp_amd/parallel.py
(term sharding, all-gather of raw Jacobian limb buffers, local sum) and the replica result gather, with the
host-compiled device arithmetic (tests/host_emul) standing in for the HIP engine.  The result must equal the reference's
single-process secp256k1_ecmult_multi_var."""
import ctypes
import os
import socket

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "host_emul", "libs2k_hostemu.so")


class EmuBackend:
    def __init__(self):
        self.lib = ctypes.CDLL(EMU)

    def msm_partial(self, sc, pt_xy, g_sc, pt_inf):
        import torch
        sc = np.ascontiguousarray(sc.numpy()); pt = np.ascontiguousarray(pt_xy.numpy())
        out = np.zeros(28, np.uint32)
        g = None if g_sc is None else np.ascontiguousarray(g_sc.numpy()).tobytes()
        self.lib.emu_msm_partial(out.ctypes.data_as(ctypes.c_void_p), g, sc.tobytes(), pt.tobytes(), None, ctypes.c_size_t(sc.shape[0]))
        return torch.from_numpy(out.view(np.int32).copy())

    def msm_window_partial(self, sc, pt_xy, g_sc, pt_inf, part, parts, direct=0, force_c=0):
        import torch
        sc = np.ascontiguousarray(sc.numpy()); pt = np.ascontiguousarray(pt_xy.numpy())
        out = np.zeros(28, np.uint32)
        g = None if g_sc is None else np.ascontiguousarray(g_sc.numpy()).tobytes()
        rc = self.lib.emu_msm_window_partial(out.ctypes.data_as(ctypes.c_void_p), g, sc.tobytes(), pt.tobytes(), None, ctypes.c_size_t(sc.shape[0]),
                                             ctypes.c_uint(part), ctypes.c_uint(parts), ctypes.c_int(direct), ctypes.c_int(force_c))
        assert rc == 0
        return torch.from_numpy(out.view(np.int32).copy())

    def gej_sum(self, parts):
        p = np.ascontiguousarray(parts.numpy()).view(np.uint32)
        out = ctypes.create_string_buffer(64)
        inf = self.lib.emu_gej_sum(out, p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.shape[0]))
        return np.frombuffer(out.raw, np.uint8), inf

    def msm_many(self, sc, pt_xy, offsets, g_sc, pt_inf):
        """a range of independent sums (the role of s2k_ecmult_multi_many_dev): each one through the emulated bucket MSM and to-affine"""
        import torch
        k = len(offsets) - 1
        xy = np.zeros((max(k, 0), 64), np.uint8); inf = np.zeros(max(k, 0), np.int32)
        for s in range(k):
            lo, hi = int(offsets[s]), int(offsets[s + 1])
            part = self.msm_partial(sc[lo:hi], pt_xy[lo:hi], None if g_sc is None else g_sc[s], None)
            r, f = self.gej_sum(part.reshape(1, 28))
            xy[s] = r; inf[s] = f
        return torch.from_numpy(xy), torch.from_numpy(inf)


def _worker(rank, world, port, sc, pts, g, q):
    import torch
    import torch.distributed as dist
    from secp256k1_zkp_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xy, inf = parallel.msm_sharded(EmuBackend(), torch.from_numpy(sc), torch.from_numpy(pts), torch.from_numpy(g))
        xy2, inf2 = parallel.msm_window_sharded(EmuBackend(), torch.from_numpy(sc), torch.from_numpy(pts), torch.from_numpy(g))
        assert inf2 == inf and xy2.tobytes() == xy.tobytes()
        lo, hi = parallel.shard_range(11, rank, world)
        local = torch.arange(lo, hi, dtype=torch.int32)
        allres = parallel.gather_results(local, 11)
        q.put((rank, xy.tobytes(), inf, allres.tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_msm_gloo_world2(ref):
    if not os.path.exists(EMU):
        pytest.skip("host emulation library not built")
    import torch.multiprocessing as mp
    rng = np.random.default_rng(21)
    n = 37
    pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    g = rng.integers(0, 256, 32, dtype=np.uint8)
    exp, einf = ref.ecmult_multi(sc, pts, g.tobytes(), None)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sc, pts, g, q)) for r in range(2)]
    for p in procs: p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, xy, inf, allres in outs:
        assert inf == einf and xy == exp.tobytes()
        assert allres == list(range(11))


def _worker_many(rank, world, port, sc, pts, off, g, q):
    import torch
    import torch.distributed as dist
    from secp256k1_zkp_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xy, inf = parallel.msm_many_sharded(EmuBackend(), torch.from_numpy(sc), torch.from_numpy(pts), off, torch.from_numpy(g))
        q.put((rank, xy.numpy().tobytes(), inf.numpy().tolist(), parallel.shard_sums(off, world)))
    finally:
        dist.destroy_process_group()


def test_many_sums_gloo_world2(ref):
    """K independent sums sharded over two ranks as objects (parallel.msm_many_sharded: term-balanced contiguous ranges, gather of the
    K x 68 result bytes): every rank ends with every sum, each equal to the reference's secp256k1_ecmult_multi_var (src/ecmult_impl.h:822-867);
    ragged sizes with empty sums at both ends and one sum that is most of the terms."""
    if not os.path.exists(EMU):
        pytest.skip("host emulation library not built")
    import torch.multiprocessing as mp
    rng = np.random.default_rng(23)
    sizes = [0, 3, 1, 40, 0, 2, 7, 0]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1]); k = len(sizes)
    pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[4] = 0
    g = rng.integers(0, 256, (k, 32), dtype=np.uint8)
    exp = [ref.ecmult_multi(sc[int(off[s]):int(off[s + 1])], pts[int(off[s]):int(off[s + 1])], g[s].tobytes(), None) for s in range(k)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_many, args=(r, 2, port, sc, pts, off, g, q)) for r in range(2)]
    for p in procs: p.start()
    outs = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, xy, inf, cut in outs:
        assert cut[0] == 0 and cut[-1] == k and 0 < cut[1] < k          # both ranks had work
        assert inf == [e[1] for e in exp]
        assert xy == b"".join(e[0].tobytes() for e in exp)


def test_shard_sums():
    from secp256k1_zkp_amd.parallel import shard_sums
    rng = np.random.default_rng(24)
    for _ in range(500):
        k = int(rng.integers(0, 40)); off = np.concatenate([[0], np.cumsum(rng.integers(0, 50, k))]).astype(np.uint64); w = int(rng.integers(1, 9))
        c = shard_sums(off, w)
        assert len(c) == w + 1 and c[0] == 0 and c[-1] == k and all(c[i] <= c[i + 1] for i in range(w))
    c = shard_sums(np.arange(0, 257 * 1024, 1024), 8)
    assert c == [0, 32, 64, 96, 128, 160, 192, 224, 256]
    c = shard_sums(np.array([0, 1000, 1001, 1002, 1003]), 2)              # one sum holds nearly all the terms: it is a range of its own
    assert c == [0, 1, 4]


def test_window_shares_add_up(ref):
    """sum over the shares of  sum_{w in share} 2^(c w) S_w  == the reference's ecmult_multi_var, for share counts that divide the
    windows evenly, unevenly, and exceed them; the bucket path and the bucket-free exact path (k_msm_direct with
    msm_share_scalar, what an overflowing launch falls back to) give the same group element share by share."""
    if not os.path.exists(EMU):
        pytest.skip("host emulation library not built")
    import torch
    be = EmuBackend()
    rng = np.random.default_rng(22)
    for n, force_c in ((9, 0), (40, 5), (40, 13)):
        pts = np.stack([np.frombuffer(ref.rand_point(rng), np.uint8) for _ in range(n)])
        sc = rng.integers(0, 256, (n, 32), dtype=np.uint8); sc[1] = 0; sc[2, :16] = 0
        g = rng.integers(0, 256, 32, dtype=np.uint8)
        exp, einf = ref.ecmult_multi(sc, pts, g.tobytes(), None)
        tsc, tpt, tg = torch.from_numpy(sc), torch.from_numpy(pts), torch.from_numpy(g)
        for parts in (1, 2, 3, 8, 40):
            shares = [be.msm_window_partial(tsc, tpt, tg, None, p, parts, direct=0, force_c=force_c) for p in range(parts)]
            xy, inf = be.gej_sum(torch.stack(shares))
            assert inf == einf and xy.tobytes() == exp.tobytes(), (n, force_c, parts)
            if parts in (2, 3):
                for p in range(parts):
                    d = be.msm_window_partial(tsc, tpt, tg, None, p, parts, direct=1, force_c=force_c)
                    a = be.gej_sum(shares[p].reshape(1, 28)); b = be.gej_sum(d.reshape(1, 28))
                    assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes(), (n, force_c, parts, p)


def test_shard_range():
    from secp256k1_zkp_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
