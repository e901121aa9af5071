"""The C-ABI library loads on a machine without a GPU and exports every symbol include/dre_hip.h declares."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "dre_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dre_[A-Za-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    import dre_amd as D
    lib = D._lib.load()
    names = _declared()
    assert len(names) >= 55
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dre_hip.h but not exported"
        assert n in D._lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert set(D._lib.PROTOTYPES) == set(names)
    assert lib.dre_version() >= 100


def test_no_cpu_fallback_without_device():
    import torch
    import dre_amd as D
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(D.DREError) as e:
        D.Context(0)
    assert e.value.code == -6 and "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "differentialriccatiequations.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert "dre_oracle" not in src and "mf_emul" not in src, f


def test_adi_options_struct_layout_matches_the_header(tmp_path):
    """`dre_adi_options` crosses the boundary BY LAYOUT (ctypes `AdiOptionsC`, Julia `AdiOptionsC`): its size and the offsets of the fields added last
    (`shift_fn`, `shift_user`: user-defined shift strategies) are compared with what the C compiler makes of include/dre_hip.h, and the defaults
    leave both callbacks unset."""
    import ctypes as C
    import subprocess
    import dre_amd as D
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dre_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(dre_adi_options), '
                   'offsetof(dre_adi_options, shifts_re), offsetof(dre_adi_options, inner_solve), offsetof(dre_adi_options, shift_fn), '
                   'offsetof(dre_adi_options, shift_user)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, o_re, o_inner, o_fn, o_user = (int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    T = D._lib.AdiOptionsC
    assert (C.sizeof(T), T.shifts_re.offset, T.inner_solve.offset, T.shift_fn.offset, T.shift_user.offset) == (size, o_re, o_inner, o_fn, o_user)
    o = T()
    D._lib.load().dre_adi_default_options(C.byref(o))
    assert o.shift_kind == 1 and not o.shift_fn and not o.shift_user and not o.inner_solve
    # the Julia mirror lists the same fields in the same order
    jl = open(os.path.join(ROOT, "differentialriccatiequations.jl_amd", "julia", "DREHip.jl")).read()
    body = jl[jl.index("struct AdiOptionsC"):]
    body = body[:body.index("\nend")]
    assert re.findall(r"^\s+([a-z_]+)::", body, flags=re.M) == [f[0] for f in T._fields_]
