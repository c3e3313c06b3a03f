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

    def gej_sum(self, parts):
        p = np.ascontiguousarray(parts.numpy()).view(np.uint32)
        out = ctypes.create_string_buffer(64)
        inf = self.lib.emu_gej_sum(out, p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.shape[0]))
        return np.frombuffer(out.raw, np.uint8), inf


def _worker(rank, world, port, sc, pts, g, q):
    import torch
    import torch.distributed as dist
    from secp256k1_zkp_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xy, inf = parallel.msm_sharded(EmuBackend(), torch.from_numpy(sc), torch.from_numpy(pts), torch.from_numpy(g))
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


def test_shard_range():
    from secp256k1_zkp_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
