"""Short versions of the randomised soak runs (tools/gpu_fuzz_stream.py, tools/gpu_fuzz_structured.py, tools/gpu_fuzz.py) inside the
GPU suite: the fixed-shape tests of round 4 were green while runs like these found a resident-sum defect, a cancellation in the
structured predictive pass and a mis-sized launch (NOTES_dead_ends.md).  Seeds are fixed; the long versions run from the tools."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_factor_store_life_cycle_randomised(pkg):
    """Growing prefixes, rejuvenated and copied particles, appended data, resets, chunked workspaces, predictive and gradient calls in
    between — every result against an engine that keeps nothing (value 1e-9, predictive 1e-8, gradient 1e-7 of the scale; above those the
    oracle arbitrates: the store-based engine may be no further from it than the bound or 4x the other engine's distance)."""
    msg = _tool("gpu_fuzz_stream").run(pkg, sequences=40, seed=2024)
    assert msg.startswith("stream fuzz ok"), msg


def test_structured_sweeps_randomised(pkg):
    """Structured value / gradient / predictive sweeps forced on random regular grids (orders, offsets, prefixes, horizons,
    near-singular particles) against dense / element-wise sweeps, the oracle and 80-bit factorisations."""
    msg = _tool("gpu_fuzz_structured").run(pkg, cases=60, seed=2024)
    assert msg.startswith("structured fuzz ok"), msg


def test_sizes_and_schedules_randomised(pkg):
    """Random (n, P, depth) across the schedule thresholds, duplicate time points, store sweeps: oracle parity and reproducibility."""
    eng = pkg.GPEngine(0)
    try:
        msg = _tool("gpu_fuzz").run(pkg, eng, cases=30, seed=2024, big=True)
    finally:
        eng.close()
    assert msg.startswith("fuzz ok"), msg


def _native(name, *args, env_extra=None):
    """Run a native driver of the C ABI (tools/native/*.cpp, built by __graft_entry__.build()) and parse its JSON line."""
    import json, os, subprocess
    exe = ROOT / "tools" / "native" / name
    assert exe.exists(), f"{exe} missing: __graft_entry__.build() compiles the native drivers"
    env = dict(os.environ)
    env.update(env_extra or {})
    env["LD_LIBRARY_PATH"] = str(ROOT / "autogp.jl_amd" / "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    pr = subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert pr.returncode == 0 and lines, (pr.returncode, pr.stdout[-500:], pr.stderr[-500:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("entry", ["value", "grad"])
def test_native_threads_driver(entry):
    """The only non-Python users of the ABI: C++ threads calling agp_logpdf / agp_logpdf_grad one particle at a time, as
    Threads.@threads does in the reference (src/inference_smc_anneal_data.jl:133,240) — SURVEY section 8(b) "Threading".  Pass: no API
    error, and the coalesced single-particle results equal the batch entry's."""
    r = _native("threads_bench", 640, 96, 3, *( ["g"] if entry == "grad" else []))
    assert r["api_errors"] == 0 and r["calls"] == 96 * 3 and r["batches"] >= 1
    assert r["max_rel_diff_vs_batch_entry"] <= 1e-10, r


@pytest.mark.parametrize("points", ["irregular", "grid", "monthly"])
def test_native_hmc_replay(points):
    """Gen.hmc's call pattern (update -> choice_gradients pairs at new parameters, src/inference_smc_anneal_data.jl:63-67) from 64 C++
    threads on irregular times, a shuffled regular grid and a shuffled monthly index.  Pass: no API error, no non-finite value at an
    accepted (info == 0) evaluation, gradient calls start from the resident factor of the value call before them."""
    r = _native("hmc_replay", 512, 64, 1, 10, 0.02, *([] if points == "irregular" else [points]))
    assert r["api_errors"] == 0 and r["non_finite"] == 0, r
    assert r["gradient_calls"] >= 64 * 2 and r["value_calls"] >= 64
    assert r["gradient_particles_from_resident_factor"] >= 0.5 * r["value_calls"], r
    assert (r["gradient_particles_in_lag_domain"] > 0) == (points == "grid"), r


def test_class_aware_pairs_randomised(pkg):
    """Gen.hmc's update -> choice_gradients pairs at the opt-in level AGP_LAG >= 2 from Python threads: random sizes, orders, prefixes,
    forced / heuristic size tests, reserved / unreserved stores, three leapfrog steps each — against the default engine's batch entry."""
    msg = _tool("gpu_fuzz_pairs").run(pkg, cases=25, seed=2024)
    assert msg.startswith("pairs fuzz ok"), msg


def test_factor_store_sizes_itself_for_small_batches():
    """A population of 192 threads whose single-particle calls arrive spread out (random host-side pauses, 20 us coalescing window:
    SMALL batches) and that never announces itself (no agp_extend_reserve): the store sizes itself by the distinct callers — no lookup
    may find that its factor was dropped before anything started from it (agp_extend_stats2: evicted_before_reuse == 0) and the
    gradient sweeps start from resident factors as in the reserved run.  With round 5's sizing rule (AGP_STORE_SELF_SIZE=0) the same
    run falls off the cliff: that run is the control."""
    env = {"HMC_WINDOW_US": "20", "HMC_JITTER_US": "3000"}
    r = _native("hmc_replay", 512, 192, 2, 10, 0.02, env_extra=env)
    assert r["api_errors"] == 0 and r["store_reserved"] is False
    assert r["mean_batch"] < 96, r                       # (the premise: batches well below the population)
    assert r["store_callers_seen"] == 192, r              # (persistent threads, as Julia's pool: the library sees the population)
    assert r["evicted_before_reuse"] == 0, r
    rr = _native("hmc_replay", 512, 192, 2, 10, 0.02, env_extra=dict(env, HMC_RESERVE="1"))
    assert rr["evicted_before_reuse"] == 0 and rr["store_reserved"] is True
    # (the arrivals are random: the few gradient calls that meet the store while it grows vary from run to run — 0 .. 170 of 7 650
    # calls over ten runs; round 5's rule, the control below, refactors thousands)
    assert r["gradient_particles_factored"] <= rr["gradient_particles_factored"] + 0.05 * r["value_calls"], (r, rr)
    old = _native("hmc_replay", 512, 192, 2, 10, 0.02, env_extra=dict(env, AGP_STORE_SELF_SIZE="0"))
    assert old["evicted_before_reuse"] > 0 and old["gradient_particles_factored"] > r["gradient_particles_factored"], (old, r)
