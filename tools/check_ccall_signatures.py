#!/usr/bin/env python3
"""Static check of the (never executed) Julia shim against the C header.

    python tools/check_ccall_signatures.py            # exit code 1 + one line per mismatch

Julia is not installed in the build image or on the GPU box, so `autogp.jl_amd/julia/src/AutoGPHIP.jl` has never met a
Julia parser.  What CAN be verified without one is the part where a silent mistake corrupts memory instead of raising:
every `ccall((:sym, LIB), ret, (types...), args...)` must agree with the prototype of `sym` in `include/autogp_hip.h` in

  * arity (of the type tuple AND of the argument list behind it),
  * width and kind of every by-value argument and of the return value (Cint / Int32 / Int64 / Float64 / Cvoid),
  * pointer-ness of every argument, and the pointee type where the header names one.

The comparison is on normalised type classes: "i32", "i64", "f64", "void", "cstr", and "ptr:<class>" for pointers
(`Ptr{Cvoid}` / `void*` / `agp_ctx*` are all "ptr:void"; `Ref{T}` and `Ptr{T}` are both pointers to T; a `Ptr{UInt8}` may
be handed to a `void*` parameter).  Also run as a CPU test: tests/test_host.py::test_julia_ccall_signatures."""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "autogp_hip.h"
SHIMS = sorted((ROOT / "autogp.jl_amd" / "julia").rglob("*.jl")) + [ROOT / "tools" / "make_golden_reference.jl"]

C_SCALARS = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void", "uint8_t": "u8", "char": "char"}
JL_SCALARS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Float64": "f64", "Cdouble": "f64", "Cvoid": "void", "UInt8": "u8",
              "Cstring": "cstr", "Clonglong": "i64", "Nothing": "void"}


def split_top(s, sep=","):
    """Split on `sep` at bracket depth 0."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


# ---- C side ----------------------------------------------------------------------------------------
def c_type_class(decl):
    """'const double* noise /* P */' -> 'ptr:f64'."""
    d = re.sub(r"/\*.*?\*/", " ", decl).strip()
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = [t for t in d.split() if t not in ("const", "struct")]
    if not toks:
        raise ValueError(decl)
    base = toks[0]
    if base == "agp_ctx":
        cls = "void"               # opaque handle
    elif base in C_SCALARS:
        cls = C_SCALARS[base]
    else:
        raise ValueError(f"unknown C type in '{decl}'")
    if stars == 0:
        return cls
    if stars == 1:
        if cls == "char":
            return "cstr"
        return "ptr:" + cls
    return "ptr:ptr"               # agp_ctx** / agp_ctx* const*


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)          # preprocessor lines
    text = text.replace('extern "C" {', " ")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(agp_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "define" in ret:
            continue
        argl = [] if args in ("", "void") else [c_type_class(a) for a in split_top(args)]
        protos[name] = (c_type_class(ret + " x") if ret else "void", argl)
    return protos


# ---- Julia side -------------------------------------------------------------------------------------
def jl_type_class(t):
    t = t.strip()
    m = re.fullmatch(r"(Ptr|Ref)\{(.*)\}", t)
    if m:
        inner = m.group(2).strip()
        if re.fullmatch(r"(Ptr|Ref)\{.*\}", inner):
            return "ptr:ptr"
        if inner not in JL_SCALARS:
            raise ValueError(f"unknown Julia pointee type '{inner}'")
        return "ptr:" + JL_SCALARS[inner]
    if t not in JL_SCALARS:
        raise ValueError(f"unknown Julia type '{t}'")
    return JL_SCALARS[t]


def find_ccalls(src):
    """Yield (line, symbol, ret, [types], n_args) for every ccall((:sym, LIB), ...)."""
    for m in re.finditer(r"ccall\(\(:(\w+)\s*,\s*\w+\)\s*,", src):
        i = m.end()
        depth, j = 1, i                      # find the matching ')' of ccall(
        while depth and j < len(src):
            if src[j] in "([{":
                depth += 1
            elif src[j] in ")]}":
                depth -= 1
            j += 1
        body = src[i:j - 1]
        parts = split_top(body)
        ret, tup, args = parts[0], parts[1], parts[2:]
        if not (tup.startswith("(") and tup.endswith(")")):
            raise ValueError(f"ccall of {m.group(1)}: argument types are not a literal tuple: {tup[:60]}")
        types = [x for x in split_top(tup[1:-1]) if x]
        yield src.count("\n", 0, m.start()) + 1, m.group(1), ret, types, len(args)


def compatible(jl, c):
    if jl == c:
        return True
    if c == "ptr:void" and jl.startswith("ptr:") and jl != "ptr:ptr":
        return True                          # any data pointer may be handed to void*
    if c == "ptr:ptr" and jl == "ptr:ptr":
        return True
    if c == "cstr" and jl in ("cstr", "ptr:u8"):
        return True
    return False


def check(header=HEADER, shims=SHIMS):
    protos = parse_header(Path(header).read_text())
    problems, seen = [], set()
    n_calls = 0
    for f in shims:
        f = Path(f)
        if not f.exists():
            continue
        src = f.read_text()
        for line, sym, ret, types, n_args in find_ccalls(src):
            n_calls += 1
            where = f"{f.relative_to(ROOT)}:{line} ccall(:{sym})"
            if sym not in protos:
                problems.append(f"{where}: no such function in {Path(header).name}")
                continue
            seen.add(sym)
            c_ret, c_args = protos[sym]
            try:
                j_ret = jl_type_class(ret); j_args = [jl_type_class(t) for t in types]
            except ValueError as e:
                problems.append(f"{where}: {e}")
                continue
            if not compatible(j_ret, c_ret):
                problems.append(f"{where}: return type {ret} ({j_ret}) but the header returns {c_ret}")
            if len(j_args) != len(c_args):
                problems.append(f"{where}: {len(j_args)} argument types but the prototype has {len(c_args)} parameters")
                continue
            if n_args != len(j_args):
                problems.append(f"{where}: {len(j_args)} argument types but {n_args} arguments passed")
            for k, (a, b) in enumerate(zip(j_args, c_args)):
                if not compatible(a, b):
                    problems.append(f"{where}: argument {k + 1} is {types[k]} ({a}) but the header says {b}")
    return problems, n_calls, seen, protos


def main():
    problems, n_calls, seen, protos = check()
    for p in problems:
        print("MISMATCH", p)
    unbound = sorted(k for k in protos if k not in seen and not k.startswith("agp_debug"))
    print(f"{n_calls} ccall sites checked against {len(protos)} prototypes; {len(problems)} mismatches; "
          f"entry points without a Julia binding: {', '.join(unbound) if unbound else 'none'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
