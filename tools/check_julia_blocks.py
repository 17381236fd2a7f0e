#!/usr/bin/env python3
"""Block-structure check of the (never executed) Julia sources: no Julia parser exists in the build image, so the shim, its tests,
deps/build.jl and tools/make_golden_reference.jl have never been parsed.  What a tokenizer CAN establish: strings, comments and
brackets are balanced, and every block opener (module, function, macro, struct, if, for, while, let, begin, do, try, quote) is closed
by an `end` — a missing or stray `end` is the commonest way an unparsed Julia file is broken.  `end` inside brackets is indexing
(`a[end]`), `for` / `if` inside brackets are comprehension / generator clauses, `f(x) = ...` opens nothing.

    python tools/check_julia_blocks.py          # exit code 1 + one line per problem
Also a CPU test: tests/test_host.py::test_julia_block_structure."""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FILES = sorted((ROOT / "autogp.jl_amd" / "julia").rglob("*.jl")) + [ROOT / "tools" / "make_golden_reference.jl"]
OPENERS = {"module", "baremodule", "function", "macro", "struct", "if", "for", "while", "let", "begin", "do", "try", "quote"}
CONTINUERS = {"elseif", "else", "catch", "finally"}


def tokens(src: str, problems: list, name: str):
    """Yield (line, kind, text) with strings / chars / comments removed; kind in {"word", "open", "close"}."""
    i, n, line = 0, len(src), 1
    while i < n:
        ch = src[i]
        if ch == "\n":
            line += 1; i += 1
        elif src.startswith("#=", i):                       # nested block comment
            depth, j = 1, i + 2
            while j < n and depth:
                if src.startswith("#=", j): depth += 1; j += 2
                elif src.startswith("=#", j): depth -= 1; j += 2
                else:
                    line += src[j] == "\n"; j += 1
            if depth: problems.append(f"{name}:{line}: unterminated #= comment")
            i = j
        elif ch == "#":
            while i < n and src[i] != "\n": i += 1
        elif src.startswith('"""', i) or ch == '"' or ch == "`":
            q = '"""' if src.startswith('"""', i) else ch
            j, l0 = i + len(q), line
            while j < n and not src.startswith(q, j):
                if src[j] == "\\": j += 1
                elif src[j] == "$" and j + 1 < n and src[j + 1] == "(":      # interpolation: skip its balanced parentheses
                    d, j = 1, j + 2
                    while j < n and d:
                        d += (src[j] == "(") - (src[j] == ")"); line += src[j] == "\n"; j += 1
                    continue
                line += src[j] == "\n" if j < n else 0
                j += 1
            if j >= n: problems.append(f"{name}:{l0}: unterminated string")
            i = j + len(q)
        elif ch == "'" and i > 0 and (src[i - 1].isalnum() or src[i - 1] in "_)]}'."):
            i += 1                                            # adjoint operator
        elif ch == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", src[i:])
            i += len(m.group(0)) if m else 1
        elif ch in "([{":
            yield line, "open", ch; i += 1
        elif ch in ")]}":
            yield line, "close", ch; i += 1
        elif ch.isalpha() or ch == "_" or ch == "@":
            m = re.match(r"@?[^\W\d][\w!]*", src[i:]) or re.match(r"@", src[i:])
            w = m.group(0)
            prev = src[i - 1] if i else ""
            if not (prev and prev in ".:") or src[i - 2:i] == "::":      # `x.end`, `:end` are a field / a symbol, not keywords
                yield line, "word", w
            i += len(w)
        else:
            i += 1


def check(path: Path) -> list:
    problems = []
    src = path.read_text()
    name = str(path.relative_to(ROOT)) if str(path.resolve()).startswith(str(ROOT)) else str(path)
    stack = []                     # ("block", word, line) / ("bracket", ch, line)
    pair = {")": "(", "]": "[", "}": "{"}
    prev_word = None
    for line, kind, t in tokens(src, problems, name):
        in_bracket = any(s[0] == "bracket" for s in stack[-1:]) if stack else False
        if kind == "open":
            stack.append(("bracket", t, line))
        elif kind == "close":
            if not stack or stack[-1][0] != "bracket" or stack[-1][1] != pair[t]:
                problems.append(f"{name}:{line}: unbalanced '{t}'" + (f" (open {stack[-1][1]!r} from line {stack[-1][2]})" if stack else ""))
                return problems
            stack.pop()
        else:
            w = t
            if w in ("abstract", "primitive", "mutable"):
                prev_word = w; continue
            if w == "type" and prev_word in ("abstract", "primitive"):
                stack.append(("block", prev_word + " type", line))
            elif w in OPENERS and not in_bracket:
                stack.append(("block", w, line))
            elif w in OPENERS and in_bracket and w in ("begin", "let", "function", "do", "quote", "try"):
                stack.append(("block", w, line))              # (a real block inside an argument list; for / if there are generator clauses)
            elif w == "end" and not in_bracket:
                if not stack or stack[-1][0] != "block":
                    problems.append(f"{name}:{line}: `end` without an open block")
                    return problems
                stack.pop()
            elif w == "end" and in_bracket and stack and stack[-1][0] == "block":
                stack.pop()
            elif w in CONTINUERS and not in_bracket and (not stack or stack[-1][0] != "block"):
                problems.append(f"{name}:{line}: `{w}` outside a block")
            prev_word = w
    for s in stack:
        problems.append(f"{name}:{s[2]}: {'block `' + s[1] + '`' if s[0] == 'block' else 'bracket ' + repr(s[1])} never closed")
    return problems


def main(files=None) -> list:
    out = []
    for f in (files or FILES):
        out += check(Path(f))
    return out


if __name__ == "__main__":
    probs = main([Path(a) for a in sys.argv[1:]] or None)
    for p in probs:
        print(p)
    print(f"checked {len(sys.argv[1:]) or len(FILES)} Julia files: {'ok' if not probs else str(len(probs)) + ' problem(s)'}")
    sys.exit(1 if probs else 0)
