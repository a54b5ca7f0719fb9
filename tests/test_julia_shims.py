"""The Julia side of the boundary (julia/*.jl) cannot be parsed here: there is no Julia in the image.  What CAN be checked
without one: that every file is lexically well formed -- strings and comments closed, brackets balanced, every block opener
(`function`, `if`, `for`, `while`, `let`, `begin`, `do`, `module`, `struct`, `try`, `quote`, `macro`) closed by its `end` --, that
every `@ccall libcnfhip.<name>(...)` names an entry point include/cnfhip.h declares, with as many arguments as the declaration
has, and that the argument count of each ccall matches.  A crude lexer, not a parser: it finds slips of the pen, not type errors."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = sorted(f for f in os.listdir(os.path.join(ROOT, "julia")) if f.endswith(".jl"))

OPENERS = {"function", "if", "for", "while", "let", "begin", "do", "module", "baremodule", "struct", "try", "quote", "macro"}


def _lex(src):
    """Yields (kind, text, line) for identifiers / brackets outside strings, chars and comments."""
    i, n, line = 0, len(src), 1
    out = []
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1; i += 1; continue
        if c == "#":
            if src.startswith("#=", i):
                j = src.find("=#", i + 2)
                assert j >= 0, f"line {line}: unterminated #= comment"
                line += src.count("\n", i, j); i = j + 2
            else:
                j = src.find("\n", i)
                i = n if j < 0 else j
            continue
        if c == '"':
            if src.startswith('"""', i):
                j = src.find('"""', i + 3)
                assert j >= 0, f"line {line}: unterminated triple-quoted string"
                line += src.count("\n", i, j); i = j + 3
                continue
            j = i + 1
            depth = 0
            while j < n:
                if src[j] == "\\":
                    j += 2; continue
                if src[j] == "$" and j + 1 < n and src[j + 1] == "(":
                    depth += 1; j += 2; continue
                if depth and src[j] == "(":
                    depth += 1
                elif depth and src[j] == ")":
                    depth -= 1
                elif src[j] == '"' and depth == 0:
                    break
                elif src[j] == "\n":
                    line += 1
                j += 1
            assert j < n, f"line {line}: unterminated string"
            i = j + 1
            continue
        if c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src.find("'", i + 2) - i <= 4)):
            i = src.find("'", i + 2) + 1          # a character literal (not the adjoint operator, which follows an identifier)
            continue
        if c in "([{":
            out.append(("open", c, line)); i += 1; continue
        if c in ")]}":
            out.append(("close", c, line)); i += 1; continue
        m = re.match(r"[A-Za-z_ -￿][A-Za-z0-9_! -￿]*", src[i:])
        if m:
            prev = src[i - 1] if i else " "
            out.append(("sym" if prev in ":." else "id", m.group(0), line))      # :end / a.end are not keywords
            i += len(m.group(0)); continue
        i += 1
    return out


@pytest.mark.parametrize("name", JL)
def test_julia_file_is_lexically_well_formed(name):
    src = open(os.path.join(ROOT, "julia", name), encoding="utf-8").read()
    toks = _lex(src)
    pairs = {")": "(", "]": "[", "}": "{"}
    br, blocks = [], []
    for kind, text, line in toks:
        if kind == "open":
            br.append((text, line))
        elif kind == "close":
            assert br and br[-1][0] == pairs[text], f"{name}:{line}: unbalanced {text!r}"
            br.pop()
        elif kind == "id":
            in_index = bool(br) and br[-1][0] == "["            # a[end], a[begin:end]
            if text in OPENERS and not (in_index and text == "begin"):
                if text in ("for", "if") and br and br[-1][0] in "([":
                    continue                                   # a generator / comprehension, or a ternary-like filter: no `end`
                blocks.append((text, line))
            elif text in ("mutable", "abstract", "primitive"):
                continue
            elif text == "type" and blocks and toks:           # `abstract type X end`
                blocks.append((text, line))
            elif text == "end" and not in_index:
                assert blocks, f"{name}:{line}: `end` without an opener"
                blocks.pop()
    assert not br, f"{name}: unclosed {br[-1][0]!r} from line {br[-1][1]}"
    assert not blocks, f"{name}: `{blocks[-1][0]}` of line {blocks[-1][1]} has no `end`"


def _header_decls():
    h = open(os.path.join(ROOT, "include", "cnfhip.h"), encoding="utf-8").read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(cnf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return decls


def test_every_ccall_names_a_declared_entry_point_with_the_right_arity():
    decls = _header_decls()
    assert len(decls) >= 50
    seen = 0
    for name in JL:
        src = open(os.path.join(ROOT, "julia", name), encoding="utf-8").read()
        src = re.sub(r"#=.*?=#", " ", src, flags=re.S)
        src = "\n".join(l.split("#", 1)[0] if '"' not in l else l for l in src.splitlines())
        for m in re.finditer(r"@ccall\(?\s*libcnfhip\.(cnf_[a-z0-9_]+)\s*\(", src):
            fn = m.group(1)
            assert fn in decls, f"{name}: @ccall of {fn}, which include/cnfhip.h does not declare"
            i, depth, start = m.end(), 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            args = src[start:i - 1]
            # top-level commas of the argument list
            d = 0; n_args = 1 if args.strip() else 0
            for ch in args:
                d += ch in "([{"
                d -= ch in ")]}"
                if ch == "," and d == 0:
                    n_args += 1
            assert n_args == decls[fn], f"{name}: {fn} is called with {n_args} arguments, the header declares {decls[fn]}"
            seen += 1
    assert seen >= 20
