"""The CPU oracle against the committed golden vectors (tests/golden/logpdf_golden.npz, generated with scipy by
tests/golden/make_golden.py) -- this is what pins the oracle's log-density VALUES, which no test of the reference
asserts."""
import os

import numpy as np
import pytest

from demc_amd import families as F

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "logpdf_golden.npz"))
RTOL = 1e-10


def _orc(orc, D):
    return orc.Oracle(n_groups=1, Np=4, D=D, n_rows=0, store_history=0)


@pytest.mark.parametrize("kind,a,b,key", [
    (F.PRIOR_NORMAL, 1.0, 2.0, "prior_normal_1_2"), (F.PRIOR_HALFCAUCHY, 0.0, 1.0, "prior_halfcauchy_0_1"),
    (F.PRIOR_HALFCAUCHY, 0.0, 2.5, "prior_halfcauchy_0_2p5"), (F.PRIOR_UNIFORM, -1.0, 2.0, "prior_uniform_m1_2"),
    (F.PRIOR_BETA, 2.0, 3.0, "prior_beta_2_3"), (F.PRIOR_BETA, 1.0, 1.0, "prior_beta_1_1"),
    (F.PRIOR_GAMMA, 2.5, 1.5, "prior_gamma_2p5_1p5"), (F.PRIOR_EXPONENTIAL, 0.0, 0.7, "prior_exponential_0p7"),
    (F.PRIOR_LOGNORMAL, 0.3, 0.8, "prior_lognormal_0p3_0p8"), (F.PRIOR_CAUCHY, -1.0, 2.0, "prior_cauchy_m1_2")])
def test_priors(orc, kind, a, b, key):
    o = _orc(orc, 1)
    o.set_priors([kind], [a], [b])
    got = o.prior(G["prior_x"].reshape(-1, 1))
    exp = G[key]
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], exp[fin], rtol=RTOL, atol=1e-13)


def test_gaussian(orc):
    o = _orc(orc, 2)
    o.set_model(F.FAM_GAUSSIAN, G["gauss_x"], [G["gauss_x"].size])
    np.testing.assert_allclose(o.loglike(G["gauss_theta"]), G["gauss_ll"], rtol=RTOL)


def test_mvn_iso(orc):
    X = G["iso_X"]
    o = _orc(orc, X.shape[1] + 1)
    o.set_model(F.FAM_MVN_ISO, X, list(X.shape))
    np.testing.assert_allclose(o.loglike(G["iso_theta"]), G["iso_ll"], rtol=RTOL)


def test_mvn_full_whitened_and_direct(orc):
    X, S = G["full_X"], G["full_Sigma"]
    o = _orc(orc, X.shape[1])
    o.set_model(F.FAM_MVN_FULL, X, list(X.shape), S)
    np.testing.assert_allclose(o.loglike(G["full_theta"]), G["full_ll"], rtol=RTOL)
    import ctypes as C
    dp = C.POINTER(C.c_double)
    for t, e in zip(G["full_theta"], G["full_ll"]):
        t = np.ascontiguousarray(t)
        got = orc.lib().orc_mvn_full_direct(np.ascontiguousarray(X).ctypes.data_as(dp), X.shape[0], X.shape[1],
                                            np.ascontiguousarray(S).ctypes.data_as(dp), t.ctypes.data_as(dp))
        assert abs(got - e) <= RTOL * abs(e)


def test_binomial_including_edges(orc):
    n, k = G["binom_n"], G["binom_k"]
    o = _orc(orc, 1)
    o.set_model(F.FAM_BINOMIAL, np.concatenate([n, k]), [n.size])
    np.testing.assert_allclose(o.loglike(G["binom_p"].reshape(-1, 1)), G["binom_ll"], rtol=RTOL)


def test_hier_binomial_with_hierarchical_prior(orc):
    k, th = G["hb_k"], G["hb_theta"]
    S = k.size
    o = _orc(orc, S + 2)
    o.set_model(F.FAM_HIER_BINOMIAL, k, [S], G["hb_n"])
    o.set_priors([1, 2] + [5] * S, [1, 0] + [0] * S, [1, 1] + [1] * S, [0, 0] + [1] * S)
    np.testing.assert_allclose(o.loglike(th), G["hb_ll"], rtol=RTOL)
    np.testing.assert_allclose(o.prior(th), G["hb_prior"], rtol=RTOL)


def test_hier_gaussian(orc):
    Y, th = G["hg_Y"], G["hg_theta"]
    o = _orc(orc, Y.shape[0] + 3)
    o.set_model(F.FAM_HIER_GAUSSIAN, Y, list(Y.shape))
    np.testing.assert_allclose(o.loglike(th), G["hg_ll"], rtol=RTOL)


def test_lnr(orc):
    c, rt, th = G["lnr_choice"], G["lnr_rt"], G["lnr_theta"]
    o = _orc(orc, th.shape[1])
    o.set_model(F.FAM_LNR, np.concatenate([c, rt]), [c.size, th.shape[1] - 1], [1.0])
    np.testing.assert_allclose(o.loglike(th), G["lnr_ll"], rtol=1e-9)


def test_lba_and_normalisation(orc):
    c, rt, th = G["lba_choice"], G["lba_rt"], G["lba_theta"]
    o = _orc(orc, th.shape[1])
    o.set_model(F.FAM_LBA, np.concatenate([c, rt]), [c.size, th.shape[1] - 3])
    np.testing.assert_allclose(o.loglike(th), G["lba_ll"], rtol=1e-9)
    assert abs(G["lba_total_mass"][0] - 1.0) < 1e-5  # sum_c int pdf = 1 (validate by integration)


def test_rastrigin(orc):
    o = _orc(orc, 2)
    o.set_model(F.FAM_RASTRIGIN, None, [])
    x = np.array([[0.0, 0.0], [1.0, -1.0], [0.5, 0.25]])
    exp = 20 + (x ** 2 - 10 * np.cos(2 * np.pi * x)).sum(1)
    np.testing.assert_allclose(o.loglike(x), exp, rtol=1e-12, atol=1e-12)
