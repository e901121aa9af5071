"""Static check of the Julia shims (Julia is not installed in the build image, so they never execute here): every `ccall` names a function that
include/dre_hip.h declares, with as many argument types as the C prototype has parameters, and an `Int32` status return where the header says
`int`.  Catches the typos a first `using DREHip` would hit."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "dre_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int|void|double|const char\s*\*)\s+(dre_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        # function-pointer parameters contain parentheses: count top-level commas only
        depth = 0; n = 0 if args in ("", "void") else 1
        for ch in args:
            if ch == "(": depth += 1
            elif ch == ")": depth -= 1
            elif ch == "," and depth == 0: n += 1
        protos[name] = (ret.replace(" ", ""), n)
    return protos


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[": depth += 1
        elif ch in ")}]": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip(): out.append(cur)
    return out


def _julia_ccalls():
    calls = []
    jdir = os.path.join(ROOT, "differentialriccatiequations.jl_amd", "julia")
    for fn in sorted(os.listdir(jdir)):
        if not fn.endswith(".jl"): continue
        src = open(os.path.join(jdir, fn)).read()
        for m in re.finditer(r"ccall\(\(:([A-Za-z0-9_]+),\s*LIB\),\s*([A-Za-z0-9_{}]+),\s*\(", src):
            # the argument-type tuple: from the opening parenthesis to its match
            i = m.end(); depth = 1; j = i
            while depth:
                depth += {"(": 1, ")": -1}.get(src[j], 0); j += 1
            types = [t for t in _split_top(src[i:j - 1]) if t.strip()]
            calls.append((fn, m.group(1), m.group(2), len(types)))
    return calls


def test_every_ccall_matches_a_header_prototype():
    protos = _header_prototypes()
    calls = _julia_ccalls()
    assert len(protos) >= 70 and len(calls) >= 50
    bad = []
    for fn, name, ret, nargs in calls:
        if name not in protos:
            bad.append(f"{fn}: {name} is not declared in dre_hip.h"); continue
        cret, cn = protos[name]
        if cn != nargs:
            bad.append(f"{fn}: {name} takes {cn} parameters, the ccall passes {nargs} types")
        if cret == "int" and ret not in ("Cint", "Int32"):
            bad.append(f"{fn}: {name} returns int, the ccall says {ret}")
    assert not bad, "\n".join(bad)
