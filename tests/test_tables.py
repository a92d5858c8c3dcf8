"""The generated tables the kernels read, checked on the CPU: the committed header is parsed and the function evaluated as the
device evaluates it (same operations, same order; numpy has no FMA, which moves last bits only)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "csrc")


def _softplus_header():
    txt = open(os.path.join(CSRC, "demc_softplus_table.hpp")).read()
    consts = {k: float.fromhex(v) for k, v in re.findall(r"constexpr double (kSp\w+) = (-?0x[0-9a-fA-F.]+p[+-]?\d+);", txt)}
    body = txt[txt.index("kSpTable[kSpDoubles] = {"):]
    vals = np.array([float.fromhex(v) for v in re.findall(r"-?0x[0-9a-fA-F.]+p[+-]?\d+", body)])
    ne, nl = (int(v) for v in re.search(r"kSpExpN = (\d+), kSpLogN = (\d+)", txt).groups())
    assert len(vals) == ne + 2 * (nl + 1)
    return consts, vals[:ne], vals[ne:].reshape(nl + 1, 2), ne, nl


def test_softplus_table_of_the_row_streaming_kernel():
    """softplus_tab (demc_device.hpp) on the committed tables (tools/gen_softplus_table.py): relative error below 6e-16 against
    80-bit arithmetic from -690 to 700, the table rows what they say they are, the split of ln2/64 exact enough for k up to 2^17"""
    c, E, L, ne, nl = _softplus_header()
    ld = np.longdouble
    assert np.allclose(E, 2.0 ** (np.arange(ne) / ne), rtol=2e-16, atol=0)
    cj = 1.0 + np.arange(nl + 1) / nl
    assert np.allclose(L[:, 0], np.log(cj), rtol=0, atol=2e-16) and np.allclose(L[:, 1], 1.0 / cj, rtol=2e-16, atol=0)
    assert L[0, 0] == 0.0 and L[0, 1] == 1.0  # (row 0 keeps a tiny exp(-|x|) at full relative accuracy)
    assert abs(c["kSpC"] * np.log(2.0) / ne - 1.0) < 3e-16
    assert int(np.float64(c["kSpLn2Hi"]).view(np.uint64)) & ((1 << 19) - 1) == 0  # 34 significant bits: k * hi is exact
    assert abs(float((ld(c["kSpLn2Hi"]) + ld(c["kSpLn2Lo"])) * ne / np.log(ld(2))) - 1.0) < 1e-18

    def device_form(x):
        a = np.maximum(-np.abs(x), -700.0)
        kf = np.rint(a * c["kSpC"])
        r = (a - kf * c["kSpLn2Hi"]) - kf * c["kSpLn2Lo"]
        k = kf.astype(np.int64)
        p = 1 / 120.0
        for q in (1 / 24.0, 1 / 6.0, 0.5, 1.0, 1.0):
            p = p * r + q
        t = np.ldexp(E[k & (ne - 1)] * p, (k >> 6).astype(np.int32))
        jf = np.rint(t * nl)
        j = jf.astype(np.int64)
        f = (t - jf / nl) * L[j, 1]
        q = 1 / 7.0
        for cc in (-1 / 6.0, 1 / 5.0, -1 / 4.0, 1 / 3.0, -0.5, 1.0):
            q = q * f + cc
        return np.maximum(x, 0.0) + (L[j, 0] + f * q)

    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-40, 40, 300000), rng.uniform(-1, 1, 50000), rng.uniform(-690, -30, 20000), rng.uniform(30, 700, 20000),
                        np.linspace(-0.01, 0.01, 2001), [0.0, -0.0, 700.0, -690.0]])
    ref = np.maximum(x.astype(ld), 0) + np.log1p(np.exp(-np.abs(x.astype(ld))))
    rel = np.abs((device_form(x).astype(ld) - ref) / ref).astype(float)
    assert rel.max() < 6e-16, (rel.max(), x[rel.argmax()])
    # the ends: +-Inf and an argument below the clamp give max(x, 0) + exp(-700)-sized terms, never NaN
    for v, want in ((np.inf, np.inf), (-np.inf, 0.0), (-1e300, 0.0)):
        got = device_form(np.array([v]))[0]
        assert got == want or abs(got - want) < 1e-300, (v, got)


def _table(header, name):
    txt = open(os.path.join(CSRC, header)).read()
    ints = {k: int(v) for k, v in re.findall(r"constexpr int (k\w+) = (-?\d+);", txt)}
    body = txt[txt.index(name + "["):]
    body = body[body.index("{") + 1:body.index("};")]
    vals = np.array([float(v) for v in re.findall(r"-?\d+\.?\d*(?:e[+-]?\d+)?", body)])
    return ints, vals


def test_phi_table_of_the_lba_kernels():
    """phiS_Phi_table (demc_device.hpp) on the committed table (tools/gen_phi_table.py): Phi and the phi recovered from the SAME
    coefficients as the derivative, on a dense grid of [-8.5, 8.5] and clamped beyond, against scipy to 2.5e-16 absolute"""
    from scipy.special import ndtr
    ints, tab = _table("demc_phi_table.hpp", "kPhiTable")
    deg, rows, S, row = ints["kPhiDeg"], ints["kPhiIntervals"], ints["kPhiPerUnit"], ints["kPhiRow"]
    assert tab.size == rows * row and rows == 17 * S + 1
    tab = tab.reshape(rows, row)
    z = np.concatenate([np.linspace(-8.5, 8.5, 400001), [-30.0, 30.0, -8.5, 8.5]])
    zs = np.clip(z * S, -8.5 * S, 8.5 * S)
    i = np.rint(zs + 8.5 * S).astype(np.int64)
    u = zs - (i - 8.5 * S)
    a = tab[i]
    P = a[:, deg].copy()
    dP = a[:, deg].copy()
    P = P * u + a[:, deg - 1]
    for k in range(deg - 2, -1, -1):
        dP = dP * u + P
        P = P * u + a[:, k]
    zc = np.clip(z, -8.5, 8.5)
    assert np.abs(P - ndtr(zc)).max() < 2.5e-16
    assert np.abs(dP * S - np.exp(-0.5 * zc * zc) / np.sqrt(2 * np.pi)).max() < 3e-16


def test_log_phi_table_of_the_lnr_kernel():
    """log_Phi_neg_table (demc_device.hpp) on the committed table (tools/gen_logphi_table.py): log Phi(-z) on [-8.5, 38.5] against
    40-digit arithmetic (scipy's log_ndtr is itself off by 1e-15 near z = 0), error relative to max(1, |g|) below 3e-16"""
    import mpmath
    mpmath.mp.dps = 40
    ints, tab = _table("demc_logphi_table.hpp", "kLogPhiTable")
    deg, rows, per, row = ints["kLogPhiDeg"], ints["kLogPhiRows"], ints["kLogPhiPerUnit"], ints["kLogPhiRow"]
    assert tab.size == rows * row and rows == 47 * per + 1 and row == deg + 1
    tab = tab.reshape(rows, row)
    z = np.concatenate([np.linspace(-8.5, 38.5, 4001), np.random.default_rng(4).uniform(-8.5, 38.5, 4000)])
    zc = np.clip(per * z, -8.5 * per, 38.5 * per)
    i = np.rint(zc + 8.5 * per).astype(np.int64)
    u = zc - (i - 8.5 * per)
    a = tab[i]
    P = a[:, deg].copy()
    for k in range(deg - 1, -1, -1):
        P = P * u + a[:, k]
    g = np.array([float(mpmath.log(mpmath.ncdf(-mpmath.mpf(float(v))))) for v in z])
    assert (np.abs(P - g) / np.maximum(1.0, np.abs(g))).max() < 3e-16
