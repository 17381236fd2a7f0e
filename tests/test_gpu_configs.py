"""GPU parity tests on the REAL shapes of BASELINE.json's configs (run with `-m gpu` on an MI355X).

    config 2   n=1024, 64 particles, depth-3 trees           (hybrid schedule of the 49-255 particle regime)
    config 3   n=2048, 512 particles, linear_schedule(2048, .10) annealing sequence   (the benchmarked population)
    config 4   n=4096, 128 particles, forced depth-6 trees   (32-leaf programs, depth-8 evaluation stack)
    config 5   online: n = 128 k, k = 1..16, 256 particles, resampling between steps (duplicate particles)

Every call goes through the C ABI; the checker is oracle/fast.py (C restatement of eval_cov + LAPACK
dpotrf/dtrtrs, itself pinned against oracle/oracle.py in tests/test_oracle.py), one particle per host thread.
Tolerance: |gpu - ref| <= 1e-8 max(1, |ref|) (BASELINE.json north_star: "logpdf within 1e-8 rel. of CPU reference").
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import fast as F

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
LP_TOL = 1e-8


def lp_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


BORDERLINE = 1e-12      # min pivot^2 / max diagonal below which fp64 factorisations may legitimately disagree about PD-ness


def check_against_oracle(pkg, lp, info, nodes, noises, ts, xs, idx=None, min_ok_frac=0.97, expect_all_pd=False):
    """Compare the particles `idx` (default: all) with the oracle, in BOTH directions of the PosDefException semantics of
    the reference (src/Model.jl:136 -> LAPACK dpotrf info): a particle the GPU rejects must be rejected by LAPACK too or be
    numerically borderline there (min pivot^2 / max diagonal < 1e-12 in the CPU factor); at most one particle the GPU accepts may
    be rejected by LAPACK (an indefinite-to-rounding matrix); everything both accept agrees within LP_TOL.  expect_all_pd: the seeded
    population is known to be positive definite throughout — any rejection is a failure."""
    idx = np.arange(len(nodes)) if idx is None else np.asarray(idx)
    programs = pkg.encode_batch(nodes)
    op_off, ops, prm_off, prm = programs
    ref, rinfo = F.gp_logpdf_many(programs, noises, ts, xs, indices=idx)
    ok = (info[idx] == 0)
    if expect_all_pd:
        assert ok.all(), f"GPU flags particles {idx[~ok].tolist()} not positive definite (info {info[idx][~ok].tolist()})"
        assert (rinfo == 0).all()
    assert ok.mean() >= min_ok_frac, f"{(~ok).sum()} of {len(idx)} particles not PD on the GPU"

    def ratio(i):
        return F.gp_pivot_ratio(ops[op_off[i]:op_off[i + 1]], prm[prm_off[i]:prm_off[i + 1]], float(noises[i]), ts)[1]
    for k in np.flatnonzero(~ok & (rinfo == 0)):          # GPU rejects, LAPACK accepts: only if borderline
        r = ratio(int(idx[k]))
        assert r < BORDERLINE, f"particle {int(idx[k])}: GPU info {int(info[idx[k]])} but LAPACK factors it with pivot ratio {r:.3e}"
    # GPU accepts, LAPACK rejects: dpotrf gives no factor to measure the margin on; at most one such particle per call
    assert (ok & (rinfo != 0)).sum() <= 1, "GPU accepts particles LAPACK rejects"
    both = ok & (rinfo == 0)
    err = lp_err(lp[idx][both], ref[both])
    assert err.max() <= LP_TOL, (int(idx[both][err.argmax()]), float(err.max()))
    assert np.isnan(lp[idx][~ok]).all()
    return float(err.max())


# ---------------------------------------------------------------------------------------------------------
def test_config2_n1024_P64_depth3(pkg, engine):
    """BASELINE configs[1]: n=1024, 64 particles, random depth-3 trees, all 64 against the oracle."""
    n, P = 1024, 64
    ts, xs = pkg.prior.synthetic_series(n, seed=1024, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(1024), P, max_depth=3)
    engine.set_data(ts, xs)
    lp, info = engine.logpdf_batch(nodes, noises, check=False)
    check_against_oracle(pkg, lp, info, nodes, noises, ts, xs)
    # regular (unshuffled) time grid, SURVEY.md §8(d) names both variants
    ts2, xs2 = pkg.prior.synthetic_series(n, seed=1024, shuffle=False)
    engine.set_data(ts2, xs2)
    lp2, info2 = engine.logpdf_batch(nodes, noises, check=False)
    check_against_oracle(pkg, lp2, info2, nodes, noises, ts2, xs2)


def test_config3_n2048_P512_every_particle(pkg, engine):
    """BASELINE configs[2] at its real shape — the population bench.py times (same seeds): n=2048, 512 particles from
    the prior, EVERY particle against the oracle.  This is the >= 256-particle schedule (in-kernel tile evaluation,
    k_chol_diag + sub-diagonal k_chol_update with the in-register solve)."""
    n, P = 2048, 512
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
    engine.set_data(ts, xs)
    lp, info = engine.logpdf_batch(nodes, noises, check=False)
    check_against_oracle(pkg, lp, info, nodes, noises, ts, xs, expect_all_pd=True)     # (bench.py prints not_positive_definite: 0)
    # the device-output entry the benchmark calls returns the same bits
    import torch
    d_lp = torch.zeros(P, dtype=torch.float64, device="cuda:0"); d_info = torch.zeros(P, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    engine.logpdf_batch_device(pkg.encode_batch(nodes), noises, n, d_lp.data_ptr(), d_info.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(d_lp.cpu().numpy(), lp, equal_nan=True) and np.array_equal(d_info.cpu().numpy(), info)


def test_config3_annealing_schedule(pkg, engine):
    """The full data-annealing n-sequence of config 3, linear_schedule(2048, .10) = 205, 410, ..., 1845, 2048
    (src/Schedule.jl:24-39; the reweight step evaluates on ts[1:step], src/inference_smc_anneal_data.jl:206-217):
    all 512 particles swept at every step, a 64-particle subset against the oracle at every step, and the 64-particle
    shard (one rank's share on 8 GPUs) evaluated on its own at every step must equal the big sweep to rounding — one
    rank's shard at the intermediate steps, ALL EIGHT ranks' shards at the last step (the benchmarked one)."""
    n, P = 2048, 512
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
    engine.set_data(ts, xs)
    sub = np.arange(0, P, 8)              # 64 particles
    steps = pkg.schedule.linear_schedule(n, 0.10)
    assert steps[0] == 205 and steps[-1] == 2048 and len(steps) == 10
    for step in steps:
        lp, info = engine.logpdf_batch(nodes, noises, n=step, check=False)
        check_against_oracle(pkg, lp, info, nodes, noises, ts[:step], xs[:step], idx=sub)
        for r in (range(8) if step == steps[-1] else (3,)):
            lo, hi = pkg.shard_range(P, r, 8)
            lps, infos = engine.logpdf_batch(nodes[lo:hi], noises[lo:hi], n=step, check=False)
            assert np.array_equal(infos, info[lo:hi]), r
            ok = infos == 0
            assert lp_err(lps[ok], lp[lo:hi][ok]).max() <= 1e-10, r


def test_config4_n4096_P128_depth6(pkg, engine):
    """BASELINE configs[3]: n=4096, 128 particles, forced depth-6 composite kernels (shallower prior samples
    rejected).  Oracle parity on every particle when the host has the cores for it (>= 64), else on 16; plus the
    size-independent properties on ALL particles: permutation invariance and batch-composition invariance."""
    n, P = 4096, 128
    ts, xs = pkg.prior.synthetic_series(n, seed=4096, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(4096), P, max_depth=6, min_depth=6, max_size=63)
    assert all(nd.depth() == 6 for nd in nodes)
    engine.set_data(ts, xs)
    lp, info = engine.logpdf_batch(nodes, noises, check=False)
    idx = np.arange(P) if F.host_cores() >= 64 else np.arange(0, P, 8)
    check_against_oracle(pkg, lp, info, nodes, noises, ts, xs, idx=idx, min_ok_frac=0.9, expect_all_pd=True)
    ok = info == 0
    perm = np.random.default_rng(7).permutation(n)
    engine.set_data(ts[perm], xs[perm])
    lp2, info2 = engine.logpdf_batch(nodes, noises, check=False)
    assert np.array_equal(info2 == 0, ok)
    assert lp_err(lp2[ok], lp[ok]).max() <= LP_TOL
    # batch composition: the same particles in another order and batch size give the same values
    sel = np.flatnonzero(ok)[::3][::-1]
    lp3, _ = engine.logpdf_batch([nodes[i] for i in sel], noises[sel], check=False)
    assert lp_err(lp3, lp2[sel]).max() <= 1e-10
    # a full 63-node / 32-leaf tree is among the population sizes the prior produces here
    assert max(nd.size() for nd in nodes) >= 31


def test_config5_online_stream_with_resampling(pkg, engine):
    """BASELINE configs[4] (scripts/online.jl:168-244): n grows 128 -> 2048 in 16 steps, 256 particles; after each
    reweight the ESS is checked and the population resampled (Gen.maybe_resample! semantics, adaptive threshold
    P/2, src/inference_smc_anneal_data.jl:230-232), so later sweeps carry duplicate particles and take the
    evaluate-once path.  Incremental log-weights follow src/inference_smc_anneal_data.jl:134-136
    (w += logpdf_n - logpdf_{n_prev} for an unchanged particle).  A 32-particle subset is checked against the oracle
    at every step, all 256 at the last."""
    n_max, P = 2048, 256
    ts, xs = pkg.prior.synthetic_series(n_max, seed=128, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(128), P, max_depth=4, max_size=31)
    engine.set_data(ts, xs)
    lw = np.zeros(P); lml = 0.0
    prev = np.zeros(P)                      # logpdf at n = 0 is 0 (src/inference_smc_anneal_data.jl:185-187)
    n_resampled = 0
    seen0, run0 = engine.dedup_stats()
    for k in range(1, 17):
        n = 128 * k
        lp, info = engine.logpdf_batch(nodes, noises, n=n, check=False)
        sub = np.arange(P) if k == 16 else np.arange(k % 8, P, 8)
        check_against_oracle(pkg, lp, info, nodes, noises, ts[:n], xs[:n], idx=sub, min_ok_frac=0.9)
        cur = np.where(info == 0, lp, -np.inf)
        lw = lw + (cur - prev)
        lw = np.where(np.isnan(lw), -np.inf, lw)
        ess = pkg.dist.effective_sample_size(lw)
        assert 0.0 < ess <= P + 1e-9
        did, parents, lw, lml = pkg.dist.maybe_resample(lw, lml, P / 2, seed=1000 + k)
        if did and k < 16:
            n_resampled += 1
            nodes = [nodes[i] for i in parents]; noises = noises[parents]; cur = cur[parents]
        prev = cur
    assert n_resampled >= 1, "the stream never resampled: the duplicate-particle path was not exercised"
    seen1, run1 = engine.dedup_stats()
    assert (seen1 - seen0) == 16 * P and (run1 - run0) < (seen1 - seen0)
    assert np.isfinite(lml)


# ---------------------------------------------------------------------------------------------------------
# the collective behind the C ABI
# ---------------------------------------------------------------------------------------------------------
def test_allgather_through_c_entry_single_rank(pkg):
    """agp_comm_get_unique_id / agp_comm_init_rank / agp_allgather_logweights{,_device} on a one-rank RCCL
    communicator (a one-GPU box cannot host two RCCL ranks): the id is 128 bytes, the communicator initialises, the
    device form copies this rank's shard into the full vector on the caller's stream, the host form is a no-op."""
    import torch
    eng = pkg.GPEngine(0)
    try:
        assert eng.comm_info() == (False, 0, 1)
        cid = pkg.GPEngine.comm_unique_id()
        assert len(cid) == 128
        eng.comm_init_rank(cid, 1, 0)
        assert eng.comm_info() == (True, 0, 1)
        ts, xs = pkg.prior.synthetic_series(300, seed=9)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(9), 11, max_depth=3)
        eng.set_data(ts, xs)
        ref, _ = eng.logpdf_batch(nodes, noises)
        d_lp = torch.zeros(11, dtype=torch.float64, device="cuda:0"); d_info = torch.zeros(11, dtype=torch.int32, device="cuda:0")
        d_all = torch.full((11,), float("nan"), dtype=torch.float64, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        eng.logpdf_batch_device(pkg.encode_batch(nodes), noises, 300, d_lp.data_ptr(), d_info.data_ptr(), st)
        eng.allgather_logweights_device(d_lp.data_ptr(), 11, d_all.data_ptr(), st)      # chained on the same stream
        torch.cuda.synchronize()
        assert np.array_equal(d_all.cpu().numpy(), ref)
        assert np.array_equal(eng.allgather_logweights(ref), ref)
        with pytest.raises(pkg.AGPError):
            eng.comm_init_rank(cid, 1, 0)            # one communicator per context
    finally:
        eng.close()


def test_unequal_shards_are_unpadded_correctly(pkg, engine):
    """Blocks of unequal length travel padded to the largest (ncclAllGather moves equal counts) and are compacted on
    the device: the kernel against the host partition for every (P, R) shape class, incl. P < R."""
    for P, R in ((11, 2), (13, 8), (512, 8), (257, 4), (5, 8), (1, 3), (1000, 7)):
        mx = (P + R - 1) // R
        padded = np.full(mx * R, np.nan)
        want = np.arange(P, dtype=np.float64) + 0.5
        for r in range(R):
            lo, hi = pkg.shard_range(P, r, R)
            assert (lo, hi) == pkg.dist.shard_range(P, r, R)
            padded[r * mx: r * mx + (hi - lo)] = want[lo:hi]
        assert np.array_equal(engine.debug_compact_shards(padded, P, R), want), (P, R)


def test_multi_device_context_pool(pkg):
    """agp_init_multi + agp_logpdf_batch_multi with the devices this box has (one): the single-process deployment
    (one Julia process driving the node) returns the same values as the plain context."""
    import torch
    ndev = torch.cuda.device_count()
    multi = pkg.GPEngineMulti(list(range(ndev)))
    try:
        ts, xs = pkg.prior.synthetic_series(700, seed=21, shuffle=True)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(21), 37, max_depth=3)
        multi.set_data(ts, xs)
        lp, info = multi.logpdf_batch(nodes, noises, check=False)
        has, rank, size = multi.engines[0].comm_info()
        assert has and rank == 0 and size == ndev
        eng = multi.engines[0]
        ref, rinfo = eng.logpdf_batch(nodes, noises, check=False)
        assert np.array_equal(info, rinfo)
        ok = info == 0
        assert lp_err(lp[ok], ref[ok]).max() <= 1e-10
        # the same with resident factors per device (agp_logpdf_batch_extend_multi) along a short schedule
        for step in (300, 520, 700):
            lpe, infoe = multi.logpdf_batch(nodes, noises, n=step, check=False, extend=True)
            refe, rinfoe = eng.logpdf_batch(nodes, noises, n=step, check=False)
            oke = infoe == 0
            assert np.array_equal(infoe, rinfoe) and lp_err(lpe[oke], refe[oke]).max() <= 1e-10
        st = eng.extend_stats()
        U = len({(pkg.encode(nd)[0].tobytes(), pkg.encode(nd)[1].tobytes(), float(z)) for nd, z in zip(nodes, noises)})
        assert st["extended"] == 2 * U and st["from_scratch"] == U
        with pytest.raises(pkg.AGPError):
            pkg.GPEngineMulti([0, 0])
    finally:
        multi.close()


def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run with both ranks on cuda:0 (AGP_BENCH_SHARE_GPU=1, gloo stands in
    for RCCL, which refuses two ranks on one device): the population stays 512... here 64 IN TOTAL, split 32 + 32,
    every rank reconstructs the same complete log-weight vector, one JSON line comes back."""
    env = dict(os.environ, AGP_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n-obs", "700",
           "--particles", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["particles_total"] == 64 and out["config"]["particles_per_gpu"] == 32
    assert out["config"]["allgather_selfcheck"] is True
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0
    # the gradient sweep split both ways, every rank timing its share: modelled cost beside measured time (what the first real
    # multi-GPU run checks the plan's constants with)
    gs = out["config"]["gradient_sweep_split"]
    assert "error" not in gs and out["leg_errors"] == [], (gs, out["leg_errors"])
    for split in ("block_split", "cost_aware_plan"):
        assert len(gs[split]["modelled_cost_per_rank"]) == 2 and all(t > 0 for t in gs[split]["measured_ms_per_rank"]), gs
    assert out["series_version"] == 2


def _bench_line(argv, extra_env=None, timeout=1200):
    env = dict(os.environ, OMP_NUM_THREADS="1", AGP_BENCH_PREWARM_S="0.2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_plain_invocation_self_launches():
    """`python bench.py --gpus 2` started PLAINLY — no torch.distributed.run around it, exactly as the driver invokes it —
    launches its own two ranks and prints one JSON line (AGP_BENCH_SHARE_GPU=1 puts both ranks on this box's one GPU)."""
    out = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--n-obs", "700", "--particles", "64", "--no-cpu-baseline"],
                      {"AGP_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["config"]["launch"] == "self-launched torch.distributed.run"
    assert out["config"]["particles_total"] == 64 and out["config"]["particles_per_gpu"] == 32
    assert out["config"]["allgather_selfcheck"] is True and len(out["config"]["per_rank_ms_per_step"]) == 2
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0


def test_bench_single_process_mode():
    """`python bench.py --single-process`: one host process drives the box's device(s) through agp_init_multi +
    agp_logpdf_batch_multi (the deployment of a single Julia process) and reports the same line."""
    out = _bench_line(["--single-process", "--gpus", "1", "--steps", "3", "--warmup", "1", "--n-obs", "700", "--particles", "64"])
    assert out["n_gpus"] == 1 and out["config"]["launch"].startswith("single-process")
    assert out["config"]["rccl_ranks_seen"] == [1] and out["config"]["allgather_selfcheck"] is True
    assert out["config"]["particles_total"] == 64 and out["value"] > 0 and out["roofline"]["achieved"] > 0


def _n_devices():
    import torch
    return torch.cuda.device_count()


needs_two_gpus = pytest.mark.skipif("_n_devices() < 2", reason="needs >= 2 visible devices (real RCCL ranks)")


@needs_two_gpus
def test_bench_two_real_rccl_ranks(tmp_path):
    """The first multi-GPU box verifies itself: `python bench.py --gpus 2` started plainly, one rank per DEVICE, the log-weights
    over the engine's own RCCL communicator (no AGP_BENCH_SHARE_GPU, neither fallback may fire), the gathered vector bit-equal
    to the one-GPU sweep of the same population (src/inference_smc_anneal_data.jl:232: every rank resamples from the same vector)."""
    common = ["--steps", "5", "--warmup", "1", "--n-obs", "700", "--particles", "64", "--no-cpu-baseline", "--no-extra-legs"]
    one = _bench_line(["--gpus", "1", "--dump-logweights", str(tmp_path / "lw1.npy")] + common)
    two = _bench_line(["--gpus", "2", "--dump-logweights", str(tmp_path / "lw2.npy")] + common)
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["launch"] == "self-launched torch.distributed.run"
    assert cfg["rccl_ranks_seen"] == 2 and cfg["allgather_selfcheck"] is True
    assert cfg["collective"].startswith("rccl via C ABI"), cfg["collective"]
    assert cfg["particles_total"] == 64 and cfg["particles_per_gpu"] == 32 and len(cfg["per_rank_ms_per_step"]) == 2
    assert cfg["allgather_us_hip_events"] is not None and 0 < cfg["allgather_us_hip_events"] < 5000
    a, b = np.load(tmp_path / "lw1.npy"), np.load(tmp_path / "lw2.npy")
    assert a.shape == (64,) and np.array_equal(a, b, equal_nan=True)
    assert one["config"]["not_positive_definite"] == two["config"]["not_positive_definite"]


@needs_two_gpus
def test_bench_single_process_two_devices():
    """One host process driving two devices (agp_init_multi + agp_logpdf_batch_multi: ncclCommInitAll, one group call)."""
    out = _bench_line(["--single-process", "--gpus", "2", "--steps", "5", "--warmup", "1", "--n-obs", "700", "--particles", "64"])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["launch"].startswith("single-process")
    assert cfg["rccl_ranks_seen"] == [2, 2] and cfg["allgather_selfcheck"] is True
    assert cfg["collective"].startswith("rccl via C ABI"), cfg["collective"]
    assert cfg["particles_total"] == 64 and out["value"] > 0


@needs_two_gpus
def test_two_rank_stream_on_two_gpus():
    """tools/run_stream.py (config 5's stream) on two real ranks: without moves the log-marginal-likelihood estimate and the
    resampling steps equal the one-rank run's; with the rejuvenation stand-in (blocks exchanged through the host channel,
    log-weights through the engine's RCCL all-gather) it completes and extends resident factors on both ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)

    def run(nproc, extra):
        port = 29100 + (os.getpid() + 7 * nproc + len(extra)) % 400
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)] if nproc > 1 else [sys.executable]) + [str(ROOT / "tools" / "run_stream.py")] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = run(1, []), run(2, [])
    assert two["n_gpus"] == 2 and one["resampled_steps"] == two["resampled_steps"]
    assert abs(one["log_ml_est"] - two["log_ml_est"]) <= 1e-9 * max(1.0, abs(one["log_ml_est"]))
    rj = run(2, ["--rejuvenate"])
    assert rj["n_gpus"] == 2 and rj["store"]["extended"] > 0 and np.isfinite(rj["log_ml_est"])


def test_bench_default_line_has_every_block():
    """The default single-GPU line (short run, small shapes): roofline + the legs beside the value sweep + cpu_baseline."""
    out = _bench_line(["--steps", "8", "--warmup", "1", "--n-obs", "384", "--particles", "512"])
    assert out["config"]["not_positive_definite"] == 0
    for key in ("roofline", "roofline_diag_kernel", "roofline_cov_kernel", "grad", "predict", "cpu_baseline"):
        assert key in out and "error" not in out[key], (key, out.get(key))
    assert out["roofline_cov_kernel"]["unit"] == "GB/s" and out["roofline_cov_kernel"]["leaf_evals_per_s"] > 0
    assert out["grad"]["kernel_ms"]["k_trtri_chain"] > 0 and out["grad"]["kernel_ms"]["k_grad_contract + k_lag_grad"] > 0
    assert out["grad"]["lag_domain_particles"] > 0 and out["grad"]["elementwise"]["ms_per_sweep"] > 0
    assert out["cpu_baseline"]["one_worker_evals_per_s"] > 0 and out["cpu_baseline"]["parity_max_rel_err_vs_gpu"] < 1e-8
    assert out["roofline"]["algorithmic_bytes_per_step"] == 8.0 * 384 * 384 * 512


def test_two_rank_stream_on_one_gpu(tmp_path, pkg):
    """The streaming driver with the ENGINE evaluator on two ranks (both on this box's GPU, log-weights over gloo), with the
    rejuvenation hook: both ranks end with the same population, equal to the single-process engine run to rounding, and
    every rank extended resident factors."""
    sys.path.insert(0, str(ROOT / "tests"))
    import _stream_worker as W
    P = 24
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29400 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_stream_worker.py"), str(tmp_path), str(P), "hook", "engine"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"stream_rank{k}.json").read_text()) for k in range(2)]
    for key in ("hist", "lml", "weights", "noises", "programs", "prev_logpdf", "parents"):
        assert res[0][key] == res[1][key], key
    assert all(x["store"]["from_scratch"] > 0 for x in res)
    ts, xs = pkg.prior.synthetic_series(600, seed=6, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), P, max_depth=3)
    eng = pkg.GPEngine(0)
    try:
        eng.set_data(ts, xs)
        ev = pkg.stream.EngineEvaluator(eng)
        st = pkg.stream.OnlineStream(nodes, noises, ev, seed=5)
        hook = W.make_hook(pkg, ev)
        steps = [100 * k for k in range(1, 7)]
        hist = [st.step(n, last=(n == steps[-1]), rejuvenate=hook) for n in steps]
    finally:
        eng.close()
    assert [h["resampled"] for h in hist] == [h["resampled"] for h in res[0]["hist"]]
    assert abs(st.log_ml_estimate() - res[0]["lml"]) <= 1e-9 * max(1.0, abs(res[0]["lml"]))
    assert [repr(nd.to_tuple()) for nd in st.nodes] == res[0]["programs"]


def test_online_stream_driver_on_gpu(pkg):
    """autogp.jl_amd/stream.py (the control flow of run_smc_anneal_data around the engine: reweight by block-extension
    sweeps, ESS, resampling, per-step predictive callback) against the same driver fed by the CPU oracle: identical
    resampling decisions, log-marginal-likelihood estimate and weights to 1e-8; the per-step predictions of the block
    against the oracle's predictive at the last step."""
    from oracle import oracle as O
    n_max, P = 768, 40
    ts, xs = pkg.prior.synthetic_series(n_max, seed=33, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(33), P, max_depth=3)
    eng = pkg.GPEngine(0)
    try:
        eng.set_data(ts, xs)
        gpu = pkg.stream.OnlineStream(nodes, noises, pkg.stream.EngineEvaluator(eng), seed=3)

        def ev(nd, nz, n):
            lp, info = F.gp_logpdf_many(pkg.encode_batch(nd), nz, ts[:n], xs[:n], threads=min(64, F.host_cores()))
            return lp, info
        cpu = pkg.stream.OnlineStream(nodes, noises, ev, seed=3)
        steps = [96 * k for k in range(1, 9)]
        tq = np.linspace(0.0, 1.1, 40)
        for n in steps:
            a = gpu.step(n, last=(n == steps[-1])); b = cpu.step(n, last=(n == steps[-1]))
            assert a["resampled"] == b["resampled"] and abs(a["ess"] - b["ess"]) <= 1e-6 * max(1.0, b["ess"])
            assert abs(a["log_ml_est"] - b["log_ml_est"]) <= 1e-8 * max(1.0, abs(b["log_ml_est"]))
            mean, var = gpu.predict_block(eng, tq, n)            # the per-step callback
            assert mean.shape == (P, 40) and np.isfinite(mean).all() and (var > 0).all()
        assert any(h["resampled"] for h in gpu.history)
        assert np.abs(gpu.particle_weights() - cpu.particle_weights()).max() <= 1e-7
        st = eng.extend_stats()
        assert st["extended"] > 0 and st["tile_rows_reused"] > 0
        i = int(np.argmax(gpu.particle_weights()))
        mu, cov = O.predict_mvn(gpu.nodes[i].to_tuple(), float(gpu.noises[i]), ts, xs, tq)
        assert np.abs(mean[i] - mu).max() <= 1e-8 * max(1.0, np.abs(mu).max())
        assert np.abs(var[i] - np.diag(cov)).max() <= 1e-8 * max(1.0, np.abs(cov).max())
    finally:
        eng.close()


def test_large_n_8192(pkg, engine):
    """Beyond BASELINE's sizes: n=8192 (64 tile rows, 2 080 tiles = 273 MB per particle), 6 particles — the small-population
    schedules at the largest matrix the suite factors, n not a multiple of the tile size alongside (8100).  logpdf against the
    C + LAPACK oracle, value+gradient sweep's value against the value sweep, predictive variance positive and the predictive
    likelihood identity logpdf(joint) - logpdf(obs) = logpdf(xs_test | obs) with every term from the GPU."""
    n = 8192
    ts, xs = pkg.prior.synthetic_series(n, seed=8192, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(8192), 6, max_depth=3)
    progs = pkg.encode_batch(nodes)
    engine.set_data(ts, xs)
    for m in (8192, 8100):
        lp, info = engine.logpdf_batch(None, noises, n=m, check=False, programs=progs)
        ref, rinfo = F.gp_logpdf_many(progs, noises, ts[:m], xs[:m], indices=np.arange(3))
        ok = (info[:3] == 0) & (rinfo == 0)
        assert ok.any() and np.array_equal(info[:3] == 0, rinfo == 0)
        assert lp_err(lp[:3][ok], ref[ok]).max() <= LP_TOL
    lpg, grads, gn, ig = engine.logpdf_grad_batch(nodes[:2], noises[:2], n=8100, check=False)
    for i in range(2):
        if info[i] == 0:
            assert ig[i] == 0 and abs(lpg[i] - lp[i]) <= 1e-10 * max(1.0, abs(lp[i])) and np.isfinite(grads[i]).all() and np.isfinite(gn[i])
    # predictive identity on the last 92 points given the first 8100
    k = next(nd for nd, i in zip(nodes, info) if i == 0)
    nz = float(noises[[i for i, v in enumerate(info) if v == 0][0]])
    lj = engine.logpdf_batch([k], [nz], n=8192)[0][0]
    lo = engine.logpdf_batch([k], [nz], n=8100)[0][0]
    mean, var, cov, _ = engine.predict_batch([k], [nz], ts[8100:], n=8100, want_cov=True)
    assert (var > 0).all()
    d = xs[8100:] - mean[0]
    L = np.linalg.cholesky(cov[0])
    z = np.linalg.solve(L, d)
    lpred = -0.5 * (len(d) * np.log(2 * np.pi) + 2 * np.log(np.diag(L)).sum() + z @ z)
    assert abs((lj - lo) - lpred) <= 1e-7 * max(1.0, abs(lpred))


def test_large_n_16384(pkg, engine):
    """n=16384: 128 tile rows, 8 256 tiles = 1.08 GB per particle (the tile-row indices of the kernels, the packed offsets and
    the per-column inverse-block store at their largest in this suite).  One particle against the C + LAPACK oracle, two
    through the size-independent identity logpdf(n) = logpdf(n - r) + logpdf(last r | first n - r)."""
    n = 16384
    ts, xs = pkg.prior.synthetic_series(n, seed=16384, shuffle=True)
    ks = [pkg.Linear(0.1, 0.3, 0.7) + pkg.Periodic(0.96, 0.21, 1.1) * pkg.SquaredExponential(0.47, 0.8),
          pkg.GammaExponential(0.2, 1.3, 0.9) + pkg.ChangePoint(pkg.Constant(0.2), pkg.SquaredExponential(0.05, 0.6), 0.5, 0.001)]
    nz = np.array([0.08, 0.05])
    engine.set_data(ts, xs)
    lp, info = engine.logpdf_batch(ks, nz, check=False)
    assert (info == 0).all()
    ref, rinfo = F.gp_logpdf_many(pkg.encode_batch(ks), nz, ts, xs, indices=np.arange(1))
    assert rinfo[0] == 0 and lp_err(lp[:1], ref).max() <= LP_TOL
    r = 70
    lo, _ = engine.logpdf_batch(ks, nz, n=n - r, check=False)
    mean, var, cov, pinfo = engine.predict_batch(ks, nz, ts[n - r:], n=n - r, want_cov=True, check=False)
    assert (pinfo == 0).all() and (var > 0).all()
    for i in range(2):
        d = xs[n - r:] - mean[i]
        L = np.linalg.cholesky(cov[i])
        z = np.linalg.solve(L, d)
        lpred = -0.5 * (r * np.log(2 * np.pi) + 2 * np.log(np.diag(L)).sum() + z @ z)
        assert abs((lp[i] - lo[i]) - lpred) <= 1e-7 * max(1.0, abs(lpred))


def _multi_entry_population(pkg, P=72, seed=12):
    """A resampled population with a skewed class mix: dense-class particles first, Toeplitz-class ones behind, copies in between."""
    sys.path.insert(0, str(ROOT / "tests"))
    import _plan_worker as W
    return W.skewed_population(pkg, P, seed=seed)


def _check_multi_entries(pkg, engines, ts, xs, n_dev):
    from oracle import oracle as O
    nodes, noises = _multi_entry_population(pkg)
    P = len(nodes)
    n = len(ts)
    tq = np.concatenate([ts[:150], ts.max() + (np.sort(ts)[1] - np.sort(ts)[0]) * np.arange(1, 40)])
    single = pkg.GPEngine(0)
    try:
        single.set_data(ts, xs)
        lp0, gr0, gn0, i0 = single.logpdf_grad_batch(nodes, noises, check=False)
        m0, v0, _, pi0 = single.predict_batch(nodes, noises, tq, check=False)
        mc0, vc0, c0, _ = single.predict_batch(nodes[:9], noises[:9], tq[:60], want_cov=True, check=False)
    finally:
        single.close()
    lp, gr, gn, info, owner = pkg.logpdf_grad_batch_multi(engines, nodes, noises, check=False, want_owner=True)
    # the split is agp_shard_plan's for a gradient sweep on this series' lattice kind; copies follow their representatives
    kind = engines[0].lattice_stats()["kind"]
    plan, cost, rc = pkg.shard_plan(pkg.encode_batch(nodes), noises, n, n_dev, sweep=1, lattice_kind=kind)
    assert np.array_equal(owner, plan) and set(np.unique(owner)) == set(range(n_dev))
    assert np.array_equal(info, i0) and (info == 0).all()
    assert np.abs(lp - lp0).max() <= 1e-10 * np.abs(lp0).max()
    for p in range(P):
        sc = max(1.0, np.abs(gr0[p]).max() if gr0[p].size else 0.0, abs(gn0[p]))
        assert gr[p].shape == gr0[p].shape and np.abs(gr[p] - gr0[p]).max(initial=0.0) <= 1e-7 * sc and abs(gn[p] - gn0[p]) <= 1e-7 * sc, p
    for p in (0, 20, P - 1):
        lo_, go_, gno_ = O.gp_logpdf_grad(nodes[p].to_tuple(), float(noises[p]), ts, xs)
        sc = max(1.0, np.abs(go_).max(), abs(gno_))
        assert abs(lp[p] - lo_) <= 1e-8 * max(1.0, abs(lo_)) and np.abs(gr[p] - go_).max() <= 1e-7 * sc and abs(gn[p] - gno_) <= 1e-7 * sc
    m1, v1, _, pi1, owner_p = pkg.predict_batch_multi(engines, nodes, noises, tq, check=False, want_owner=True)
    plan_p, _, _ = pkg.shard_plan(pkg.encode_batch(nodes), noises, n, n_dev, sweep=2, lattice_kind=kind, m_future=39)
    assert np.array_equal(owner_p, plan_p) and np.array_equal(pi1, pi0)
    scm = np.maximum(1.0, np.maximum(np.abs(m0).max(axis=1), np.abs(v0).max(axis=1)))[:, None]
    assert (np.abs(m1 - m0) / scm).max() <= 1e-8 and (np.abs(v1 - v0) / scm).max() <= 1e-8
    mo, co = O.predict_mvn(nodes[3].to_tuple(), float(noises[3]), ts, xs, tq)
    assert np.abs(m1[3] - mo).max() <= 1e-8 * max(1.0, np.abs(mo).max()) and np.abs(v1[3] - np.diag(co)).max() <= 1e-8 * max(1.0, np.abs(co).max())
    mc1, vc1, c1, _ = pkg.predict_batch_multi(engines, nodes[:9], noises[:9], tq[:60], want_cov=True, check=False)
    assert np.abs(c1 - c0).max() <= 1e-8 * max(1.0, np.abs(c0).max()) and np.abs(mc1 - mc0).max() <= 1e-8 * max(1.0, np.abs(mc0).max())
    # malformed calls are refused, not split
    with pytest.raises(pkg.AGPError):
        pkg.logpdf_grad_batch_multi([engines[0], engines[0]], nodes, noises)


def test_multi_context_gradient_and_predictive_entries_on_one_device(pkg):
    """agp_logpdf_grad_batch_multi / agp_predict_batch_multi over THREE contexts of this box's one device (the entries need no
    communicator: results go to the host): the split is agp_shard_plan's, the shares run concurrently on the contexts' persistent host
    threads, and the results come back in the caller's order — equal to one engine's sweep over the whole population at the
    tolerances of the paths involved, and to the oracle."""
    ts, xs = pkg.prior.synthetic_series(700, seed=3, shuffle=True)
    engines = [pkg.GPEngine(0) for _ in range(3)]
    try:
        for e in engines:
            e.set_data(ts, xs)
        _check_multi_entries(pkg, engines, ts, xs, 3)
    finally:
        for e in engines:
            e.close()


@needs_two_gpus
def test_multi_device_gradient_and_predictive_entries(pkg):
    """The same over two DEVICES through agp_init_multi's contexts (one process driving the node)."""
    ts, xs = pkg.prior.synthetic_series(700, seed=3, shuffle=True)
    multi = pkg.GPEngineMulti([0, 1])
    try:
        multi.set_data(ts, xs)
        _check_multi_entries(pkg, multi.engines, ts, xs, 2)
    finally:
        multi.close()
