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


def _fit_and_predict(P, seed):
    """make_and_fit_model -> forced maybe_resample -> mcmc -> predict_mvn, on whatever process
    group is (or is not) initialised; returns the FULL mixture and this rank's programs."""
    from nowcastautogp_amd import autogp
    from nowcastautogp_amd import nowcast as nc
    from tests import mirror_contracts as mc
    from tests.engine_oracle import OracleEngine
    data = nc.create_transformed_data(mc.days(0, 20), mc.series20(), transformation=lambda v: v)
    model = nc.make_and_fit_model(data, engine=OracleEngine(), seed=seed, n_particles=P, n_mcmc=2,
                                  n_hmc=1)
    model.log_weights = model.log_weights - 3.0 * np.arange(len(model.particles))
    assert autogp.maybe_resample(model, float(P))          # ESS < P: always resamples
    autogp.mcmc_structure(model, 1, 1)
    mix = autogp.predict_mvn(model, mc.days(20, 24))
    draws = mix.rand(7)
    return dict(means=mix.means, covs=mix.covs, w=mix.weights, draws=draws,
                programs=[(p.program()[0].tolist(), p.program()[1].tolist(), p.noise)
                          for p in model.particles], lw=model.log_weights.copy())


def _worker_model(rank, world, port, q, P, seed):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, _fit_and_predict(P, seed)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_sharded_model_reproduces_the_single_rank_model():
    """GPModel sharding (ragged: 5 particles over 2 ranks), fit_smc, maybe_resample with the
    descriptor exchange, structure + HMC moves and predict_mvn's gather, end to end over gloo:
    every particle has its own random stream keyed by its global index, so the two-rank run must
    give the single-rank mixture (this path had never been executed)."""
    P, seed = 5, 3
    ref = _fit_and_predict(P, seed)                      # this process: no process group
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_model, args=(r, world, port, q, P, seed))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [len(res[r]["programs"]) for r in range(world)] == [3, 2]
    assert res[0]["programs"] + res[1]["programs"] == ref["programs"]
    for r in range(world):                               # every rank holds the FULL mixture
        assert np.allclose(res[r]["means"], ref["means"], rtol=1e-12, atol=1e-12)
        assert np.allclose(res[r]["covs"], ref["covs"], rtol=1e-12, atol=1e-12)
        assert np.allclose(res[r]["w"], ref["w"], rtol=1e-13)
        assert np.allclose(res[r]["draws"], ref["draws"], rtol=1e-10, atol=1e-10)
    assert np.array_equal(np.concatenate([res[0]["lw"], res[1]["lw"]]), ref["lw"])


def _fit_and_nowcast(P, seed):
    """make_and_fit_model, then forecast_with_nowcasts in the shared-K mode and in two lockstep
    refinement modes (one with every scenario resampling), on whatever process group there is."""
    from nowcastautogp_amd import distributed as D
    from nowcastautogp_amd import nowcast as nc
    from tests import mirror_contracts as mc
    from tests.engine_oracle import OracleEngine
    calls = {"gather": 0, "objects": 0}
    real_rows, real_many = D.all_gather_rows, D.exchange_particles_many

    def rows(*a, **k):
        calls["gather"] += 1
        return real_rows(*a, **k)

    def many(*a, **k):
        calls["objects"] += 1
        return real_many(*a, **k)

    D.all_gather_rows, D.exchange_particles_many = rows, many
    data = nc.create_transformed_data(mc.days(0, 20), mc.series20(), transformation=lambda v: v)
    model = nc.make_and_fit_model(data, engine=OracleEngine(), seed=seed, n_particles=P, n_mcmc=1,
                                  n_hmc=1)
    scen = nc.create_nowcast_data([[101.0, 102.5], [99.0, 104.0], [103.0, 100.5]], mc.days(20, 22))
    dates = mc.days(22, 25)
    out = {}
    for name, mode in (("shared_k", dict(ess_threshold=0.5)), ("hmc", dict(n_hmc=1)),
                       ("structure_resampled", dict(n_mcmc=1, n_hmc=1, ess_threshold=1.0))):
        calls["gather"] = calls["objects"] = 0
        out[name] = nc.forecast_with_nowcasts(model, scen, dates, 4, **mode)
        out[name + "_collectives"] = dict(calls)
    D.all_gather_rows, D.exchange_particles_many = real_rows, real_many
    return out


def _worker_nowcast(rank, world, port, q, P, seed):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, _fit_and_nowcast(P, seed)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forecast_with_nowcasts_batches_scenarios_and_reproduces_one_rank():
    """with the particles sharded over ranks forecast_with_nowcasts keeps
    its batched forms — the shared-K call in the default mode, lockstep P x D calls in the
    refinement modes — with ONE log-weight all-gather per weight update for ALL scenarios and ONE
    descriptor exchange when scenarios resample; the draws are the single-rank draws."""
    P, seed = 5, 4
    ref = _fit_and_nowcast(P, seed)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nowcast, args=(r, world, port, q, P, seed))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for name in ("shared_k", "hmc", "structure_resampled"):
            assert res[r][name].shape == (3, 12)
            assert np.allclose(res[r][name], ref[name], rtol=1e-9, atol=1e-9), name
        # collectives of one forecast over D = 3 scenarios: independent of D
        assert res[r]["shared_k_collectives"] == {"gather": 2, "objects": 0}
        # add_data (0) + maybe_resample (1) + predict (weights 1 + mixtures 1)
        assert res[r]["hmc_collectives"] == {"gather": 3, "objects": 0}
        assert res[r]["structure_resampled_collectives"] == {"gather": 3, "objects": 1}


def test_shard_partition_covers_everything():
    from nowcastautogp_amd import _lib
    from nowcastautogp_amd.distributed import shard
    for P in (1, 7, 64, 257):
        for size in (1, 2, 3, 8):
            idx = np.concatenate([np.arange(P)[shard(P, r, size)] for r in range(size)])
            assert np.array_equal(idx, np.arange(P))
            # the C-ABI's collective (ngp_weights_allgather_normalize) partitions the same way
            for r in range(size):
                if P >= size:
                    sl = shard(P, r, size)
                    assert _lib.shard(P, size, r) == (sl.start, sl.stop - sl.start)
