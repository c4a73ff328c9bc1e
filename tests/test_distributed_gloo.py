"""world_size-2 `gloo` coverage of the N>1 path (runs on CPU): block partition, the log-weight
all-gather + normalisation, ancestor resampling and descriptor exchange."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from oracle import oracle_np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nowcastautogp_amd import distributed as D
    P, S = 10, 3
    rng = np.random.Generator(np.random.PCG64(42))
    logw_all = -500 + 3 * rng.standard_normal((P, S))
    sl = D.shard(P)
    assert P % world == 0
    w_loc, ess = D.normalize_log_weights(logw_all[sl])
    w1, ess1 = D.normalize_log_weights(logw_all[sl][:, 0])
    wall = D.all_gather_rows(w_loc)
    anc = D.resample_ancestors(wall[:, 0], seed=7)
    descr = [("particle", int(i)) for i in range(P)][sl]
    mine = D.exchange_particles(descr, anc)
    q.put((rank, sl.start, sl.stop, w_loc, ess, w1, ess1, anc, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_normalisation_and_resampling_over_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.Generator(np.random.PCG64(42))
    logw_all = -500 + 3 * rng.standard_normal((10, 3))
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 10)]
    for s in range(3):
        w_ref, ess_ref, _ = oracle_np.weights_normalize(logw_all[:, s])
        got = np.concatenate([r[3][:, s] for r in res])
        assert np.allclose(got, w_ref, rtol=1e-14, atol=0)
        for r in res:
            assert abs(r[4][s] - ess_ref) < 1e-12 * ess_ref
    assert np.array_equal(res[0][7], res[1][7])          # same ancestors on every rank
    anc = res[0][7]
    assert res[0][8] == [("particle", int(a)) for a in anc[:5]]
    assert res[1][8] == [("particle", int(a)) for a in anc[5:]]
    assert abs(res[0][6] - oracle_np.weights_normalize(logw_all[:, 0])[1]) < 1e-12


def test_shard_partition_covers_everything():
    from nowcastautogp_amd.distributed import shard
    for P in (1, 7, 64, 257):
        for size in (1, 2, 3, 8):
            idx = np.concatenate([np.arange(P)[shard(P, r, size)] for r in range(size)])
            assert np.array_equal(idx, np.arange(P))
