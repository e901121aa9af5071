import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session.  No skip: on a GPU box a missing device / extension must fail loudly."""
    import dre_amd as D
    c = D.Context(0)
    D.set_default_context(c)
    _session_ctx.append(c)
    return c


# every tunable of dre_ctx_set_option (csrc/api.hip, option_ref): snapshot before a test, restored after it — a test that changes an option and
# "restores" a hard-coded default would otherwise switch the rest of the session back to the defaults, although tools/option_matrix.sh set a
# configuration for the whole process through DRE_OPTIONS
OPTION_NAMES = (
    "dense_inverse_max_n", "compress_direct_max_n", "compress_direct_ratio", "compress_factor_min_n", "compress_factor_min_cols", "compress_sketch",
    "compress_sketch_min_cols", "compress_sketch_extra", "compress_sketch_cholqr", "compress_sketch_sparse", "compress_sketch_ratio",
    "top_inverse_max_rows", "mf_subtree", "setup_streams", "x_side_stream", "side_after_panels", "setup_batched", "dense_warm", "side_prefetch",
    "xwarm_sx", "prefetch_batch", "dense_x_max_n", "dense_x_max_k", "adi_group", "adi_group_max_n", "adi_fan", "ros1_recurrence",
    "adi_fan_max_coef", "shard_min_cols", "x_compress_every", "pivot_growth_warn", "pivot_growth_fail", "pivot_static", "pivot_refine_steps",
    "shard_emulate", "comm_host_async", "side_gate", "recurrence_wide", "gemm_swizzle", "mf_swizzle", "ros2_tight")
_session_ctx = []


@pytest.fixture(autouse=True)
def _options_are_restored_after_every_test(request):
    if "ctx" in request.fixturenames:
        request.getfixturevalue("ctx")          # (the first GPU test of a session creates the context: snapshot it too)
    c = _session_ctx[0] if _session_ctx else None
    before = {k: c.get_option(k) for k in OPTION_NAMES} if c is not None else None
    yield
    c = _session_ctx[0] if _session_ctx else None
    if c is not None and before is not None:
        for k, v in before.items():
            if c.get_option(k) != v:
                c.set_option(k, v)


@pytest.fixture(scope="session")
def rail371():
    import dre_amd as D
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    return d, L, Dm
