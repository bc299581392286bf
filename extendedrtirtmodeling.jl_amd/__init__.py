"""MI355X-native Gibbs engine for the `sample!` hot path of ExtendedRtIrtModeling.jl.

Export list mirrors the part of /root/reference/src/ExtendedRtIrtModeling.jl:33-77 that belongs to the hot path.
"""
from .base import InputData, InputData4R, InputPara, OutputDic, SimConditions, setCond
from .gibbs import (GibbsMlIrt, GibbsRtIrt, GibbsRtIrtCross, GibbsRtIrtCrossQr, GibbsRtIrtLatent, GibbsRtIrtLatentQr, GibbsRtIrtNull,
                    GibbsRtIrtQuantile, checkConvergence, coef, ess_rhat, getDic, getDicHost, simulateData,
                    getLogLikelihood, precis, sample, sample_b)
from .simtools import (comparePara, getBias, getMetrics, getMetrics2, getRmse, runSimulation, setDataMlIrt, setDataRtIrt, setDataRtIrtCross,
                       setDataRtIrtLatent, setDataRtIrtNull,
                       setTrueParaMlIrt, setTrueParaRtIrt, setTrueParaRtIrtCross, setTrueParaRtIrtLatent)
from . import _lib, parallel

__all__ = [
    "setCond", "SimConditions", "InputData", "InputData4R", "InputPara", "OutputDic",
    "setDataMlIrt", "setDataRtIrt", "setDataRtIrtCross", "setDataRtIrtLatent", "setDataRtIrtNull", "runSimulation", "getMetrics", "getMetrics2",
    "comparePara",
    "setTrueParaMlIrt", "setTrueParaRtIrt", "setTrueParaRtIrtCross", "setTrueParaRtIrtLatent",
    "getBias", "getRmse", "getDic", "getDicHost", "checkConvergence", "ess_rhat", "simulateData", "getLogLikelihood", "sample_b", "sample",
    "GibbsMlIrt", "GibbsRtIrt", "GibbsRtIrtCrossQr", "GibbsRtIrtLatentQr", "GibbsRtIrtQuantile", "GibbsRtIrtNull", "GibbsRtIrtCross", "GibbsRtIrtLatent", "coef", "precis",
]
