"""Minimal stand-in for MCMCChains.Chains: value array (iterations x parameters x chains) + names, and the
summary statistics the reference's tests read from describe(chains)[1] (:mean, :std, :rhat)."""
import numpy as np


class Chains:
    def __init__(self, value, names, parameters, internals=("acceptance", "lp")):
        self.value = np.asarray(value)  # [Ns][n_parms + 2][n_chains]
        self.names = list(names)
        self.parameters = list(parameters)
        self.internals = list(internals)

    def __len__(self):  # length(chains) == number of kept iterations (test/utility_tests.jl:34-39)
        return self.value.shape[0]

    def __getitem__(self, name):
        return self.value[:, self.names.index(name), :]

    @staticmethod
    def _rhat(x):
        """split-R-hat over chains (Vehtari et al. 2021, rank-free form)."""
        n, m = x.shape
        h = n // 2
        if h < 2:
            return np.nan
        s = np.concatenate([x[:h], x[h:2 * h]], axis=1)
        w = s.var(axis=0, ddof=1).mean()
        b = h * s.mean(axis=0).var(ddof=1)
        if w == 0:
            return np.nan
        return float(np.sqrt(((h - 1) / h * w + b / h) / w))

    def describe(self):
        out = {}
        for j, nm in enumerate(self.names):
            if nm in self.internals:
                continue
            x = self.value[:, j, :]
            out[nm] = dict(mean=float(x.mean()), std=float(x.std(ddof=1)), rhat=self._rhat(x))
        return out

    def mean(self):
        return {k: v["mean"] for k, v in self.describe().items()}
