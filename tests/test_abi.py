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
