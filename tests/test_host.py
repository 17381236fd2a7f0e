"""CPU tests of the host side: grammar mirror, program encoding, prior / schedule restatements, and
that the C-ABI library loads and exports every symbol include/autogp_hip.h declares (no compute
calls without a GPU)."""
import ctypes
import math
import re
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

ROOT = Path(__file__).resolve().parent.parent


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_grammar_mirror(pkg):
    G = pkg
    k = G.ChangePoint(G.Linear(0.1) + G.Periodic(0.96, 0.21, 1.1), G.SquaredExponential(0.47) * G.GammaExponential(0.42, 0.58, 3.2),
                      0.5, 0.95)
    assert k.size() == 7 and k.depth() == 3                      # src/GP.jl:93-99,475-479
    assert G.Linear(0.1).params() == (0.1, 1, 1)                 # defaults, src/GP.jl:189
    assert G.SquaredExponential(0.47).params() == (0.47, 1)
    with pytest.raises(AssertionError):
        G.GammaExponential(1.0, 2.5)                             # src/GP.jl:274
    seq = G.unroll(k)
    assert [type(s).__name__ for s in seq] == ["Linear", "Periodic", "Plus", "SquaredExponential", "GammaExponential",
                                               "Times", "ChangePoint"]
    ops, prm = G.encode(k)
    assert ops.dtype == np.uint8 and list(ops) == [2, 5, 6, 3, 4, 7, 8]
    assert list(prm) == [0.1, 1, 1, 0.96, 0.21, 1.1, 0.47, 1, 0.42, 0.58, 3.2, 0.5, 0.95]
    # the oracle understands exactly this encoding
    assert O.program_to_tree(ops, prm) == k.to_tuple()
    assert G.from_tuple(k.to_tuple()) == k
    assert G.Linear(1).to_tuple() == ("LIN", 1.0, 1.0, 1.0)       # Int-typed params (test/test_GP.jl:109)


def test_encode_batch(pkg):
    G = pkg
    nodes = [G.Constant(0.5), G.Linear(0.1, 1.3, 0.7) + G.WhiteNoise(2), G.Periodic(1, 2, 3)]
    op_off, ops, prm_off, prm = G.encode_batch(nodes)
    assert list(op_off) == [0, 1, 4, 5] and list(prm_off) == [0, 1, 5, 8]
    assert list(ops) == [1, 2, 0, 6, 5]
    assert ops.flags.c_contiguous and prm.flags.c_contiguous and prm.dtype == np.float64


def test_header_symbols_exported(pkg):
    """Every function declared in include/autogp_hip.h is exported by the built library."""
    hdr = (ROOT / "include" / "autogp_hip.h").read_text()
    declared = sorted(set(re.findall(r"\b(agp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 12
    assert set(declared) == set(pkg.EXPORTED_SYMBOLS)
    assert pkg.LIB_PATH.exists(), "build the engine first: python __graft_entry__.py"
    lib = ctypes.CDLL(str(pkg.LIB_PATH))
    for sym in declared:
        assert getattr(lib, sym) is not None
    lib.agp_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.agp_version()


def test_header_is_plain_c(tmp_path):
    """include/autogp_hip.h is what a C / cgo / ccall binding includes: it must compile as C99 (no C++ in the signatures), and the
    error codes are the documented ones."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "autogp_hip.h"\n'
                   "int main(void) { return (AGP_OK == 0 && AGP_ERR_ARG == -1 && AGP_ERR_HIP == -2 && AGP_ERR_PROGRAM == -3 && AGP_ERR_NODATA == -4 &&\n"
                   "                         AGP_ERR_COMM == -5 && AGP_ERR_HOST == -6 && AGP_COMM_ID_BYTES == 128) ? 0 : 1; }\n")
    exe = tmp_path / "hdr"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_julia_ccall_signatures(tmp_path):
    """The Julia shim has never been executed (no Julia here or on the GPU box): every ccall's type tuple, argument count
    and return type is checked statically against the prototypes of include/autogp_hip.h (tools/check_ccall_signatures.py)
    — and the checker itself is checked on deliberately broken copies of the shim."""
    sys.path.insert(0, str(ROOT / "tools"))
    import check_ccall_signatures as CK
    problems, n_calls, seen, protos = CK.check()
    assert n_calls >= 20 and len(protos) >= 40
    assert problems == [], "\n".join(problems)
    # every entry of the reference's path has a binding
    for sym in ("agp_init", "agp_set_data", "agp_logpdf", "agp_logpdf_grad", "agp_logpdf_batch", "agp_logpdf_batch_extend",
                "agp_predict_batch", "agp_infer_gp_sum", "agp_init_multi", "agp_logpdf_batch_multi", "agp_allgather_logweights",
                "agp_comm_init_rank", "agp_comm_get_unique_id", "agp_shard_range", "agp_destroy", "agp_last_error"):
        assert sym in seen, sym
    shim = (ROOT / "autogp.jl_amd" / "julia" / "src" / "AutoGPHIP.jl").read_text()
    breakages = [
        ("(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), eng.ptr, ts, xs, length(ts))", "(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), eng.ptr, ts, xs, length(ts))"),   # width
        ("(Ptr{Cvoid}, Ptr{Float64}, Int32), eng.ptr, lw, length(lw))", "(Ptr{Cvoid}, Float64, Int32), eng.ptr, lw, length(lw))"),                                          # pointer-ness
        ("(Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), eng.ptr, id, n_ranks, rank)", "(Ptr{Cvoid}, Ptr{UInt8}, Int32), eng.ptr, id, n_ranks)"),                                  # arity
        ("(Ref{Ptr{Cvoid}}, Cint), ref, device)", "(Ref{Ptr{Cvoid}}, Cint), ref)"),                                                                                          # argument count
        ("ccall((:agp_destroy, LIB), Cvoid,", "ccall((:agp_destroy, LIB), Cint,"),                                                                                          # return type
    ]
    for good, bad in breakages:
        assert good in shim, good
        f = tmp_path / "broken.jl"
        f.write_text(shim.replace(good, bad))
        orig_root = CK.ROOT
        try:
            CK.ROOT = tmp_path
            pr, _, _, _ = CK.check(shims=[f])
        finally:
            CK.ROOT = orig_root
        assert len(pr) >= 1, f"checker missed: {bad}"


def test_documented_build_recipes_produce_the_full_library(pkg, tmp_path):
    """The build commands a maintainer is told to run (autogp.jl_amd/julia/deps/build.jl and INTEGRATION.md section 1) are
    extracted, run with their outputs redirected to a temp dir, and every exported symbol is resolved from the results
    (a library linked from a subset of csrc/*.hip links fine and fails at the first ccall)."""
    import shlex, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    csrc = ROOT / "autogp.jl_amd" / "csrc"
    jl = (ROOT / "autogp.jl_amd" / "julia" / "deps" / "build.jl").read_text()
    m = re.search(r"^run\(`([^`]+)`\)", jl, re.M)
    assert m, "deps/build.jl: no run(`...`) build command"
    out_jl = tmp_path / "jl" / "libautogp_hip.so"
    cmd_jl = m.group(1)
    for var, val in (("$CSRC", str(csrc)), ("$hipcc", hipcc), ("$LIBFILE", str(out_jl)), ("$OBJDIR", str(tmp_path / "obj"))):
        assert var in cmd_jl
        cmd_jl = cmd_jl.replace(var, val)
    assert "$" not in cmd_jl
    md = (ROOT / "INTEGRATION.md").read_text()
    blk = re.search(r"## 1\. Build\s+```bash\n(.*?)```", md, re.S).group(1)
    lines = [ln for ln in blk.splitlines() if ln.strip() and not ln.lstrip().startswith("#") and not ln.startswith("export ")]
    assert len(lines) == 1, lines
    out_md = tmp_path / "md" / "libautogp_hip.so"
    cmd_md = lines[0].replace("$PWD/autogp.jl_amd/lib/libautogp_hip.so", str(out_md)).replace("$PWD/autogp.jl_amd/build/obj",
                                                                                                 str(tmp_path / "obj"))
    assert "$" not in cmd_md
    # both recipes share the object directory: the second run only links
    for cmd, out in ((cmd_jl, out_jl), (cmd_md, out_md)):
        subprocess.run(shlex.split(cmd), check=True, cwd=str(ROOT), stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(str(out))
        for sym in pkg.EXPORTED_SYMBOLS:
            assert hasattr(lib, sym), f"{out}: {sym} missing"
    # a loaded library has no unresolved kernel-launch functions either: RTLD_NOW
    ctypes.CDLL(str(out_md), mode=ctypes.RTLD_GLOBAL | 2)
    # and the resource report tool names every kernel unit
    tool = (ROOT / "tools" / "kernel_resources.py").read_text()
    assert "agp_engine.hip" not in tool and 'glob("agp_kernels*.hip")' in tool


def test_lattice_admission_of_calendar_indices(pkg):
    """agp_probe_lattice = the admission test of agp_set_data on its own (host code): date indices as GPModel ingests them
    (datetime2unix + min-max LinearTransform, src/api.jl:49-51,98-101) are lattices with gaps, with the day as spacing — admitted
    up to 4096 lattice points (the LDS budget of a rank table); longer calendar lattices by compact tables (kind 3)."""
    pr = pkg.prior
    for freq, n, gaps in (("B", 2048, {1, 3}), ("B", 2900, {1, 3}), ("M", 134, {28, 29, 30, 31}), ("M", 64, {28, 29, 30, 31})):
        for shuffle in (False, True):
            ts, _ = pr.calendar_series(n, freq, seed=1, shuffle=shuffle)
            r = pkg.probe_lattice(ts)
            assert r["kind"] == 2, (freq, n)
            g = np.sort(r["index"])
            assert g[0] == 0 and r["n_lattice"] == g[-1] + 1 <= 4096 and set(np.diff(g)) == gaps
            days = (pr.datetime2unix(pr.calendar_dates(n, freq)) / 86400.0)
            assert np.array_equal(g, np.sort(days - days.min()).astype(np.int64))           # the lattice index IS the day number
            # |t_a - t_b| = |g_a - g_b| h to the admitted 1e-11 of the smallest gap
            h = r["spacing"]
            assert np.abs(ts - (ts.min() + r["index"] * h)).max() <= 1e-11 * np.diff(np.sort(ts)).min()
    assert pkg.probe_lattice(pr.calendar_series(2048, "D")[0])["kind"] == 1              # calendar days: a regular grid
    assert pkg.probe_lattice(np.linspace(0.0, 1.0, 777))["kind"] == 1
    # raw unix seconds, unscaled (test/test_GP.jl:35-68 style raw-space calls)
    r = pkg.probe_lattice(pr.datetime2unix(pr.calendar_dates(120, "M", "1990-01-01")))
    assert r["kind"] == 2 and r["spacing"] == 86400.0
    # a regular grid with missing observations is a lattice with gaps
    r = pkg.probe_lattice(np.delete(np.linspace(0.0, 1.0, 400), [17, 18, 200, 333]))
    assert r["kind"] == 2 and r["n_lattice"] == 400
    # refused: irregular times; an offset grid (ulp(1000) / h = 4.6e-10); a business-day index with one date moved by 1e-9 gaps;
    # duplicates; a daily index with a few long gaps (the lattice lags at one ordinal difference then spread over too many values)
    rng = np.random.default_rng(0)
    ts, _ = pr.calendar_series(512, "B")
    moved = ts.copy(); moved[100] += 1e-9 * np.diff(ts).min()
    days = np.cumsum(np.where(rng.random(3000) < 0.02, rng.integers(5, 40, 3000), 1)).astype(float)
    for bad in (np.sort(rng.random(500)), np.linspace(1000.0, 1001.0, 2048), moved, np.concatenate([ts, ts[5:6]]), days / days.max()):
        assert pkg.probe_lattice(bad)["kind"] == 0
    # lattices beyond 4096 points (2048 month starts: 62 304 days; 3000 business days) or with more lags than elements
    # (n_lattice > n^2 / 2: 40 month starts) get no table over their LAGS — but in time order the lag of a pair (i, i + od) takes few
    # values (months: <= 6 over 170 years, business days 3, years <= 4 across 2100): kind 3, compact tables indexed by (ordinal difference, lag - base[od])
    for freq, n, wmax in (("M", 2048, 6), ("B", 3000, 3), ("Y", 300, 4), ("M", 40, 5), ("Q", 400, 5)):
        for shuffle in (False, True):
            ts, _ = pr.calendar_series(n, freq, seed=2, shuffle=shuffle)
            r = pkg.probe_lattice(ts)
            assert r["kind"] == 3, (freq, n)
            g = np.sort(r["index"])
            daysq = (pr.datetime2unix(pr.calendar_dates(n, freq)) / 86400.0)
            assert np.array_equal(g, np.sort(daysq - daysq.min()).astype(np.int64))
            assert np.abs(ts - (ts.min() + r["index"] * r["spacing"])).max() <= 1e-11 * np.diff(np.sort(ts)).min()
            w = max(int((g[od:] - g[:-od]).max() - (g[od:] - g[:-od]).min()) + 1 for od in range(1, n))
            assert w == wmax <= 8


def test_julia_block_structure(tmp_path):
    """The Julia sources (shim, its tests, deps/build.jl, the golden-vector script) have never met a Julia parser: a tokenizer
    checks that strings, comments and brackets balance and that every block opener has its `end` (tools/check_julia_blocks.py) —
    and that the checker notices when they do not."""
    sys.path.insert(0, str(ROOT / "tools"))
    import check_julia_blocks as CB
    assert len(CB.FILES) >= 4
    assert CB.main() == []
    src = (ROOT / "autogp.jl_amd" / "julia" / "src" / "AutoGPHIP.jl").read_text()
    broken = {
        "missing_end": src.replace("\nend\n", "\n", 1),
        "stray_end": src.replace("function check(", "end\nfunction check(", 1),
        "open_paren": src.replace("check(eng, ccall((:agp_set_data, LIB), Cint,", "check(eng, ccall(((:agp_set_data, LIB), Cint,", 1),
        "open_string": src.replace('"Pre-size the factor store', 'Pre-size the factor store', 1),
    }
    for name, text in broken.items():
        assert text != src, name
        f = tmp_path / f"{name}.jl"
        f.write_text(text)
        assert CB.main([f]) != [], name
    # constructs the tokenizer must NOT trip over: indexing with end, comprehensions, one-line definitions, symbols, adjoints
    ok = tmp_path / "ok.jl"
    ok.write_text("module M\nf(x) = x[end] + sum(y for y in x if y > 0)\ng = [i for i in 1:3]\nh(a) = a' * a\ns = :end\n"
                  "function k(v)\n  if v > 0 v else -v end\nend\nz = \"a $(f([1])) end\"\nstruct P\n  x::Int\nend\nend\n")
    assert CB.main([ok]) == []


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must fail loudly, never fall back to a CPU implementation."""
    with pytest.raises(pkg.AGPError, match="no CPU fallback"):
        pkg.GPEngine(0)
    with pytest.raises(pkg.AGPError):
        pkg.compute_cov_matrix_vectorized(pkg.Constant(1.0), 0.1, np.linspace(0, 1, 4), engine=None)


def test_product_does_not_import_oracle():
    """The shipped package never references oracle/ (parity claims would be void otherwise)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]oracle|agp_oracle", re.M)
    for f in (ROOT / "autogp.jl_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".h", ".jl"):
            hits = [m.group(0) for m in pat.finditer(f.read_text())]
            # gp.py documents (in a docstring) that to_tuple() feeds the oracle's tree form
            hits = [h for h in hits if not (f.name == "gp.py" and h == "oracle/oracle")]
            assert not hits, (f, hits)


def test_prior_restatement(pkg):
    pr = pkg.prior
    assert pr.NODE_DIST_CP.sum() == pytest.approx(1.0) and pr.NODE_DIST_NOCP.sum() == pytest.approx(1.0)
    assert pr.transform_param("lengthscale", 0.0) == pytest.approx(math.exp(-1.5))      # src/Model.jl:24,44-46
    assert pr.transform_param("gamma", 0.0) == pytest.approx(1.0)                        # 2/(1+e^0)
    assert pr.idx_to_depth(1) == 1 and pr.idx_to_depth(3) == 2 and pr.idx_to_depth(7) == 3
    rng = np.random.default_rng(0)
    nodes, noises = pr.sample_particles(rng, 200, max_depth=3)
    assert max(n.depth() for n in nodes) <= 3 and (noises > 1e-5).all()

    def check(node, under_cp_or_root):
        if isinstance(node, pkg.ChangePoint):
            assert under_cp_or_root, "ChangePoint only at the root or under a ChangePoint (src/Model.jl:103)"
            assert node.scale == 0.001                           # src/Model.jl:121
            check(node.left, True); check(node.right, True)
        elif isinstance(node, (pkg.Plus, pkg.Times)):
            check(node.left, False); check(node.right, False)
        else:
            assert isinstance(node, (pkg.Linear, pkg.GammaExponential, pkg.Periodic))   # default leaf dist
    for nd in nodes:
        check(nd, True)
    deep, _ = pr.sample_particles(np.random.default_rng(1), 5, max_depth=6, min_depth=6)
    assert all(n.depth() == 6 for n in deep)
    ts, xs = pr.synthetic_series(300, seed=1)
    assert ts.min() == 0.0 and ts.max() == 1.0 and abs(xs.mean()) < 1e-12 and xs.max() - xs.min() == pytest.approx(1.0)


def test_linear_schedule(pkg):
    ls = pkg.schedule.linear_schedule
    assert ls(2048, .10) == [205, 410, 615, 820, 1025, 1230, 1435, 1640, 1845, 2048]    # config 3 n-sequence
    assert ls(100, .10) == list(range(10, 101, 10))
    assert ls(105, .10)[-1] == 105 and ls(144, .05)[-1] == 144


def test_dist_helpers(pkg):
    d = pkg.dist
    for P, W in ((512, 8), (10, 4), (3, 8), (64, 1)):
        rngs = [d.shard_range(P, r, W) for r in range(W)]
        assert rngs[0][0] == 0 and rngs[-1][1] == P
        assert all(rngs[i][1] == rngs[i + 1][0] for i in range(W - 1))
        assert d.shard_sizes(P, W) == [b - a for a, b in rngs]
    lw = np.log(np.array([0.1, 0.2, 0.3, 0.4]))
    assert d.effective_sample_size(lw) == pytest.approx(O.effective_sample_size(lw))
    did, parents, nlw, lml = d.maybe_resample(np.array([0.0, -50.0, -50.0, -50.0]), 0.0, 2.0, seed=1)
    assert did and (parents == 0).all() and (nlw == 0).all()
    assert lml == pytest.approx(math.log(1 + 3 * math.exp(-50)) - math.log(4))
    did, parents, nlw, lml = d.maybe_resample(np.zeros(4), 1.5, 2.0, seed=1)
    assert not did and lml == 1.5


def test_no_waterfalled_buffer_accesses():
    """tools/check_isa.py: every buffer load / store of the compiled kernels takes its descriptor from SGPRs (a descriptor the
    compiler cannot prove wave-uniform costs a readfirstlane loop per load — a silent ~10 % in the K-loops)."""
    import shutil, importlib.util
    from pathlib import Path
    if not (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("check_isa", Path(__file__).resolve().parent.parent / "tools" / "check_isa.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.main() == 0
