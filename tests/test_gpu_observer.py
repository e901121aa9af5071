"""Observer payloads: `observe_gale_step!(observer, i, X, residual, residual_norm)` gets the iterate and the
residual OBJECT at every ADI iteration (src/lyapunov/adi.jl:119, src/Callbacks.jl:97-107).  The loop is device resident, so observers that
want them declare `needs_state = True`; the solve then runs through the stepwise protocol and the hooks fire live with LDLᵀ handles that
materialise on first access (`rank()` never downloads).  Oracle sequences: tests/golden/observer_371.npz (make_fixtures_r03.py)."""
import os

import numpy as np
import pytest

import dre_amd as D
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class Recorder:
    needs_state = True

    def __init__(self):
        self.ranks, self.norms, self.given, self.its, self.shifts, self.events = [], [], [], [], [], []

    def observe_gale_start(self, prob, alg):
        self.events.append("start")

    def observe_gale_metadata(self, desc, mu):
        assert desc == "ADI shifts"
        self.shifts.append(mu)

    def observe_gale_step(self, i, X, residual, residual_norm):
        assert X is not None and residual is not None
        self.its.append(i)
        self.ranks.append(X.rank())
        self.norms.append(D.norm(residual))
        self.given.append(residual_norm)

    def observe_gale_done(self, iters, X, residual, residual_norm):
        self.events.append(("done", iters))

    def observe_gdre_step(self, t, X, K):
        self.events.append(("gdre_step", float(t)))


def _shifts():
    return list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))


def test_observer_sees_rank_and_residual_of_every_adi_iteration_inside_gdre(ctx, rail371):
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "observer_371.npz"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    # literal mode: the reference's arithmetic at every compression, hence the oracle's residual widths and ranks
    rec = Recorder()
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts()), compress_exact=True)), dt=-100.0, observer=rec, return_stats=True)
    its = [int(v) for v in g["iters"]]
    assert [x["iters"] for x in st["gales"]] == its
    assert len(rec.ranks) == len(g["rank_X"]) == sum(its) + 3              # "step 0" of every Lyapunov solve (adi.jl:65) + every iteration
    assert rec.its == [i for n in its for i in range(n + 1)]
    # rank(X) per iteration: the first Lyapunov solve (from X0) is the oracle's sequence exactly; in the warm-started ones the residual
    # factor may be a few columns wider or narrower (eigenvalues at the truncation threshold 100 eps max|lambda|, LDLt.jl:216, fall on
    # either side with a different eigensolver), so: same compression cadence (adi.jl:111-113: the rank drops at the same iterations), the
    # per-iteration growth (= residual width) and every compressed rank within 4 columns of the oracle's
    assert rec.ranks[:its[0] + 1] == [int(v) for v in g["rank_X"][:its[0] + 1]]
    off = 0
    for nit in its:
        mine, ref = np.array(rec.ranks[off:off + nit + 1]), np.array(g["rank_X"][off:off + nit + 1])
        dm, dr = np.diff(mine), np.diff(ref)
        assert np.array_equal(dm < 0, dr < 0), (list(mine), list(ref))
        assert np.max(np.abs(dm[dm > 0] - dr[dr > 0])) <= 4 and np.max(np.abs(mine[1:][dm < 0] - ref[1:][dr < 0]), initial=0) <= 4, (list(mine), list(ref))
        off += nit + 1
    # norm(residual) evaluated by the OBSERVER on the handle equals the norm the solver reports, and both follow the oracle's sequence
    nr, gv, ref = np.array(rec.norms), np.array(rec.given), g["norm_residual"]
    assert np.allclose(nr, gv, rtol=1e-6, atol=1e-3 * gv.min())
    assert np.allclose(gv, ref, rtol=1e-5, atol=0.2 * ref.min())
    assert len(rec.shifts) == sum(its) and [e for e in rec.events if e == "start"] == ["start"] * 3
    assert [e for e in rec.events if isinstance(e, tuple) and e[0] == "gdre_step"] == [("gdre_step", t) for t in (4500.0, 4400.0, 4300.0, 4200.0)]
    # same trajectory as the device-resident loop (no observer)
    ref_sol = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts()), compress_exact=True)), dt=-100.0)
    for a, b in zip(sol.K, ref_sol.K):
        assert D.delta(a, b) < 1e-10 or np.linalg.norm(a - b) == 0.0


def test_live_observer_on_a_single_gale_matches_the_replayed_one(ctx, rail371):
    """Default (Krylov-truncated) mode, one Lyapunov solve: the live hooks report the same iteration numbers, shifts and norms as the
    replayed ones of the device-resident loop, X of iteration i has rank(X0) + i * (residual width) columns, and the residual handle of the
    LAST iteration is the residual of the returned X (lyapunov/residual.jl:3-31)."""
    d, L, Dm = rail371
    tau = 100.0
    X0 = D.lowrank(L, Dm)
    K0 = (d.B.T @ L) @ Dm @ (L.T @ d.E)
    G = np.hstack([d.C.T, d.E.T @ L])
    BtLD = (d.B.T @ L) @ Dm
    S = np.zeros((G.shape[1],) * 2); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + Dm / tau
    F = D.lr_update((d.A - d.E / (2 * tau)).tocsc(), -1.0, d.B, K0)
    prob = D.GALEProblem(d.E, F, D.lowrank(G, S))
    alg = D.ADI(shifts=D.Shifts.Cyclic(_shifts()))

    class Replay:
        def __init__(self):
            self.norms, self.shifts = [], []

        def observe_gale_step(self, i, X, residual, nrm):
            assert X is None and residual is None          # no `needs_state`: scalars only
            self.norms.append(nrm)

        def observe_gale_metadata(self, desc, mu):
            self.shifts.append(mu)
    rp, rec = Replay(), Recorder()
    Xa, ia = D.solve_gale(prob, alg, initial_guess=X0, observer=rp, return_info=True)
    Xb, ib = D.solve_gale(prob, alg, initial_guess=X0, observer=rec, return_info=True)
    # (bit for bit — unless the multifrontal path is forced on this small pencil (tools/option_matrix.sh): the one-shot solve then takes fan groups, the same
    # iterates through partial fractions, which a step-by-step solve cannot)
    fan_forced = ctx.get_option("dense_inverse_max_n") == 0 and ctx.get_option("adi_fan") >= 2
    if fan_forced:
        assert ia["iters"] == ib["iters"] and np.allclose(np.array(rp.norms), np.array(rec.given), rtol=1e-9, atol=0.1 * ib["abstol"])
    else:
        assert ia["iters"] == ib["iters"] and np.array_equal(np.array(rp.norms), np.array(rec.given))
    assert np.allclose(np.array(rp.shifts), np.array(rec.shifts))
    k = ib["rhs_cols"]
    assert rec.ranks == [6 + i * k for i in range(ib["iters"] + 1)]
    assert np.linalg.norm(Xa.dense() - Xb.dense()) <= (1e-10 * np.linalg.norm(Xb.dense()) if fan_forced else 0.0)
    Rtrue = D.residual(prob, Xb)
    assert abs(D.norm(Rtrue) - rec.norms[-1]) < 0.25 * ib["abstol"]          # recurrence vs from-scratch residual: rounding of the size of abstol/10
