"""Config + data/parameter containers mirroring /root/reference/src/Base.pl.jl:45-136.

Field names follow the reference; Julia identifiers that are not valid Python (σ²t) get ASCII names, and every
field is also reachable under its Julia spelling through attribute aliases (Para.θ, Para.Σp, getattr(Para, "σ²t")).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class SimConditions:
    """src/Base.pl.jl:45-56"""
    nSubj: int
    nItem: int
    nFeat: int
    nIter: int
    nChain: int
    nBurnin: int
    nThin: int
    nRep: int
    qRa: float
    qRt: float


def setCond(*, nSubj=2000, nItem=15, nFeat=3, nIter=5000, nChain=4, nBurnin=None, nThin=1, nRep=10, qRa=0.5, qRt=0.5):
    """src/Base.pl.jl:59-62.  As in the reference the nBurnin kwarg is ignored and forced to round(nIter/2)
    (Julia's round: half to even, which Python's round() also implements)."""
    nBurnin = int(round(nIter / 2))
    return SimConditions(int(nSubj), int(nItem), int(nFeat), int(nIter), int(nChain), nBurnin, int(nThin), int(nRep),
                         float(qRa), float(qRt))


class InputData:
    """src/Base.pl.jl:67-78: computes kappa = Y .- 0.5 and logT = log.(T)."""

    def __init__(self, *, Y=(), T=(), X=()):
        self.Y = np.asarray(Y)
        self.κ = self.Y.astype(np.float64) - 0.5 if self.Y.size else np.zeros(0)
        self.T = np.asarray(T, dtype=np.float64)
        self.logT = np.log(self.T) if self.T.size else np.zeros(0)
        self.X = np.asarray(X, dtype=np.float64)

    kappa = property(lambda self: self.κ)


class InputData4R:
    """src/Base.pl.jl:86-95 (all fields given explicitly)."""

    def __init__(self, *, Y=(), κ=(), T=(), logT=(), X=(), kappa=None):
        self.Y = np.asarray(Y)
        self.κ = np.asarray(κ if kappa is None else kappa, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.logT = np.asarray(logT, dtype=np.float64)
        self.X = np.asarray(X, dtype=np.float64)

    kappa = property(lambda self: self.κ)


_ALIASES = {"ω": "omega", "θ": "theta", "ζ": "zeta", "λ": "lam", "σ²t": "sig2t", "σ2t": "sig2t", "ν": "nu", "β": "beta",
            "ρ": "rho", "Σp": "Sigp", "lambda_": "lam"}


class InputPara:
    """src/Base.pl.jl:100-115: mutable bag ω θ a b ζ λ σ²t ν β ρ Σp (ASCII: omega theta a b zeta lam sig2t nu beta rho Sigp)."""
    _fields = ("omega", "theta", "a", "b", "zeta", "lam", "sig2t", "nu", "beta", "rho", "Sigp")

    def __init__(self, **kw):
        for f in self._fields:
            object.__setattr__(self, f, np.zeros(0))
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        k = _ALIASES.get(k, k)
        if k not in self._fields:
            raise AttributeError(f"InputPara has no field {k}")
        object.__setattr__(self, k, np.asarray(v, dtype=np.float64))

    def __getattr__(self, k):
        if k in _ALIASES:
            return object.__getattribute__(self, _ALIASES[k])
        raise AttributeError(k)

    def __repr__(self):
        return "InputPara(" + ", ".join(f"{f}{tuple(getattr(self, f).shape)}" for f in self._fields if getattr(self, f).size) + ")"


class OutputDic:
    """src/Base.pl.jl:130-136"""

    def __init__(self, pD=None, DIC=None):
        self.pD = pD
        self.DIC = DIC

    def __repr__(self):
        return f"OutputDic(pD={self.pD}, DIC={self.DIC})"
