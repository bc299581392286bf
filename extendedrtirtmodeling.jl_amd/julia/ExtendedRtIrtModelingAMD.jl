# ExtendedRtIrtModelingAMD.jl -- Julia-side drop-in for the `sample!` hot path of ExtendedRtIrtModeling.jl, backed by
# libertirt.so (include/ertirt.h).  SHIPS AS SOURCE, UNTESTED: no Julia toolchain exists in the build container or on
# the GPU box (SURVEY.md 8(b)); the same C ABI is exercised by the Python ctypes host in this package, whose tests are
# the parity tests.  See INTEGRATION.md for how a maintainer wires this into the reference package.
#
# What it replaces (paths relative to the reference repository):
#   sample!(::GibbsMlIrt)          src/GibbsRtIrt.pl.jl:210-257
#   sample!(::GibbsRtIrt)          src/GibbsRtIrt.pl.jl:278-346
#   sample!(::GibbsRtIrtCrossQr)   src/GibbsRtIrtCross.pl.jl:265-325
#   sample!(::GibbsRtIrtLatentQr)  src/GibbsRtIrtLatent.pl.jl:271-337
#   sample!(::GibbsRtIrtNull)      src/GibbsRtIrt.pl.jl:367-426
#   sample!(::GibbsRtIrtCross)     src/GibbsRtIrtCross.pl.jl:176-235
#   sample!(::GibbsRtIrtLatent)    src/GibbsRtIrtLatent.pl.jl:168-233
# Struct names, fields, constructor behaviour, kwargs, error text and the Post layout are the reference's; coef / precis /
# getDic / comparePara / checkConvergence of the reference keep working on the filled `Post`.
module ExtendedRtIrtModelingAMD

using LinearAlgebra, Random

export sample!, GibbsMlIrt, GibbsRtIrt, GibbsRtIrtCrossQr, GibbsRtIrtLatentQr, GibbsRtIrtQuantile, GibbsRtIrtNull, GibbsRtIrtCross,
       GibbsRtIrtLatent, essRhat, simulateData!, libertirt_path!, rcclUniqueId, getDicDevice, checkConvergenceDevice, setSeed!

const LIB = Ref{String}(get(ENV, "LIBERTIRT", "libertirt.so"))
libertirt_path!(p::AbstractString) = (LIB[] = String(p))
# the 128-byte ncclUniqueId of a subject-sharded chain: made on ONE process, handed to all of them (MPI.Bcast!, a file, ...)
rcclUniqueId() = (u = zeros(UInt8, 128); check(ccall((:erm_rccl_unique_id, LIB[]), Cint, (Ptr{UInt8},), u)); u)

# ---- mirror of erm_config / erm_state / erm_timing (include/ertirt.h); field order and types must match exactly
struct ErmConfig
    model::Int32; n_item::Int32; n_subj::Int64; n_feat::Int32; n_iter::Int32; n_chain::Int32; n_burnin::Int32
    intercept::Int32; one_pl::Int32; cov2one::Int32; sigp_mode::Int32; chain_id::Int32; q_rt::Float64; seed::UInt64
    device::Int32; precision::Int32; trace_mode::Int32; lanes_per_row::Int32; block_threads::Int32; grid_blocks::Int32
    profile::Int32; flags::Int32; nu_trace_max_gb::Float64
end
struct ErmState
    theta::Ptr{Float64}; a::Ptr{Float64}; b::Ptr{Float64}; zeta::Ptr{Float64}; lambda::Ptr{Float64}; sig2t::Ptr{Float64}
    beta::Ptr{Float64}; sigp::Ptr{Float64}; rho::Ptr{Float64}; nu::Ptr{Float64}
end
const MODEL_MLIRT, MODEL_RTIRT, MODEL_CROSSQR, MODEL_LATENTQR = Int32(0), Int32(1), Int32(2), Int32(3)
const MODEL_NULL, MODEL_CROSS, MODEL_LATENT = Int32(4), Int32(5), Int32(6)
const TRACE_RA, TRACE_RT, TRACE_QR, TRACE_LOGLIKE = Int32(0), Int32(1), Int32(2), Int32(3)

lasterr() = unsafe_string(ccall((:erm_last_error, LIB[]), Cstring, ()))
check(rc::Integer) = rc == 0 ? nothing : error("libertirt: " * lasterr())
# the struct mirrors above were written against ERM_ABI_VERSION 4 of include/ertirt.h
const ABI_VERSION = 4
function checkAbi()
    v = ccall((:erm_abi_version, LIB[]), Cint, ())
    v == ABI_VERSION || error("libertirt: the library's struct layout version is $v, this module was written against $ABI_VERSION")
end

# ---- the reference's containers (src/Base.pl.jl:100-115), reproduced so the module is self-contained when used stand-alone;
# inside the reference package these definitions are simply dropped in favour of the existing ones.
Base.@kwdef mutable struct InputPara
    ω::Array = Float64[]; θ::Array = Float64[]; a::Array = Float64[]; b::Array = Float64[]; ζ::Array = Float64[]
    λ::Array = Float64[]; σ²t::Array = Float64[]; ν::Array = Float64[]; β::Array = Float64[]; ρ::Array = Float64[]; Σp::Array = Float64[]
end
mutable struct OutputPost
    ra; rt; qr; logLike; mean
end

abstract type GibbsAMD end
for (T, model) in ((:GibbsMlIrt, MODEL_MLIRT), (:GibbsRtIrt, MODEL_RTIRT), (:GibbsRtIrtCrossQr, MODEL_CROSSQR), (:GibbsRtIrtLatentQr, MODEL_LATENTQR),
                   (:GibbsRtIrtNull, MODEL_NULL), (:GibbsRtIrtCross, MODEL_CROSS), (:GibbsRtIrtLatent, MODEL_LATENT))
    @eval begin
        mutable struct $T <: GibbsAMD
            Cond; Data; truePara; Para; Post
            handle::Ptr{Cvoid}; key::Any; seed::UInt64; device::Int32; precision::Int32
            shard::Any      # nothing, or (rank, count, nSubjTotal, rowBase, uid): Cond.nSubj / Data then describe this process's subjects only
                            # (give every rank the same β / ρ / Σp start, e.g. Random.seed!(s) before the constructor and cut θ, ζ afterwards:
                            #  setInitialValues draws them from the global RNG as the reference does)
            function $T(Cond; Data = [], truePara = [], Para = Float64[], Post = Float64[], seed = 1234, device = 0, precision = 1, shard = nothing)
                obj = new(Cond, Data, truePara, Para, Post, C_NULL, nothing, UInt64(seed), Int32(device), Int32(precision), shard)
                setInitialValues(obj)                      # always overwrites Para, as the reference's constructors do
                obj.Post = OutputPost([], [], [], [], Float64[])
                finalizer(o -> (o.handle != C_NULL && ccall((:erm_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), o.handle)), obj)
                return obj
            end
        end
        modelid(::$T) = $model
    end
end
const GibbsRtIrtQuantile = GibbsRtIrtLatentQr     # README.md:22,95; the reference's export is commented out (src/ExtendedRtIrtModeling.jl:65)

# setInitialValues: src/GibbsRtIrt.pl.jl:84-93,122-133; src/GibbsRtIrtCross.pl.jl:123-134; src/GibbsRtIrtLatent.pl.jl:113-124
function setInitialValues(s::GibbsMlIrt)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), β = randn(C.nFeat + 1)); s
end
function setInitialValues(s::GibbsRtIrt)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), β = randn(C.nFeat + 1, 2), Σp = Matrix{Float64}(I, 2, 2)); s
end
function setInitialValues(s::GibbsRtIrtCrossQr)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), ρ = randn(C.nItem), Σp = Matrix{Float64}(I, 2, 2)); s
end
function setInitialValues(s::GibbsRtIrtLatentQr)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), β = randn(C.nFeat + 2), Σp = Matrix{Float64}(I, 2, 2)); s
end

# src/GibbsRtIrt.pl.jl:159-170; src/GibbsRtIrtCross.pl.jl:85-96; src/GibbsRtIrtLatent.pl.jl:78-89
function setInitialValues(s::GibbsRtIrtNull)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), Σp = Matrix{Float64}(I, 2, 2)); s
end
function setInitialValues(s::GibbsRtIrtCross)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), ρ = randn(C.nItem), Σp = Matrix{Float64}(I, 2, 2)); s
end
function setInitialValues(s::GibbsRtIrtLatent)
    C = s.Cond
    s.Para = InputPara(θ = randn(C.nSubj), a = ones(C.nItem), b = zeros(C.nItem), ζ = randn(C.nSubj), λ = zeros(C.nItem),
                       σ²t = ones(C.nItem), β = randn(C.nFeat + 2), Σp = Matrix{Float64}(I, 2, 2)); s
end

dense(x) = isempty(x) ? Float64[] : Array{Float64}(x)
ptr(x::Array{Float64}) = isempty(x) ? Ptr{Float64}(C_NULL) : pointer(x)

function engine!(M::GibbsAMD, intercept::Bool, onepl::Bool, cov2one::Bool; upload::Bool = true)
    key = (intercept, onepl, cov2one)
    M.handle != C_NULL && M.key == key && return M.handle
    M.handle != C_NULL && ccall((:erm_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), M.handle)
    C = M.Cond
    cfg = ErmConfig(modelid(M), C.nItem, C.nSubj, C.nFeat, C.nIter, C.nChain, C.nBurnin, intercept, onepl, cov2one, 0, 0, C.qRt,
                    M.seed, M.device, M.precision, 1, 0, 0, 0, 0, 0, 0.0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    checkAbi()
    check(ccall((:erm_create, LIB[]), Cint, (Ref{ErmConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    if M.shard !== nothing       # one chain over several devices (include/ertirt.h, erm_set_shard_rccl): collective, before the data
        rank, count, ntot, base, uid = M.shard
        length(uid) == 128 || error("shard: uid must be the 128 bytes of rcclUniqueId()")
        GC.@preserve uid check(ccall((:erm_set_shard_rccl, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{UInt8}), h[], rank, count, ntot, base, uid))
    end
    if !upload                                                  # the data set will be generated on the device (simulateData!)
        M.handle, M.key = h[], key
        return M.handle
    end
    Y = Array{UInt8}(M.Data.Y)                                  # Matrix{Bool} from setData* or a 0/1 numeric matrix
    logT = modelid(M) == MODEL_MLIRT ? Float64[] : dense(M.Data.logT)
    X = (modelid(M) in (MODEL_CROSSQR, MODEL_CROSS, MODEL_NULL) || C.nFeat == 0) ? Float64[] : dense(M.Data.X)
    GC.@preserve Y logT X check(ccall((:erm_set_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}), h[], Y, ptr(logT), ptr(X)))
    M.handle, M.key = h[], key
    return M.handle
end

function state_arrays(P::InputPara)
    (dense(P.θ), dense(P.a), dense(P.b), dense(P.ζ), dense(P.λ), dense(P.σ²t), dense(vec(P.β)), dense(vec(P.Σp)), dense(P.ρ), dense(vec(P.ν)))
end

# nChain INDEPENDENT chains, chain l on GPU devices[l] (erm_farm_*, include/ertirt.h): the library samples them concurrently (one host
# thread per chain) and reduces Post.mean over the devices with one RCCL all-reduce.  Chain 1 starts from MCMC.Para, chain l > 1 from a
# fresh setInitialValues draw, as nChain separately constructed samplers would.
function sampleFarm!(M::GibbsAMD, intercept::Bool, onepl::Bool, cov2one::Bool, devices)
    C = M.Cond
    M.shard === nothing || error("a subject-sharded sampler cannot also farm chains")
    devs = Int32[devices[mod1(l, length(devices))] for l in 1:C.nChain]
    cfg = ErmConfig(modelid(M), C.nItem, C.nSubj, C.nFeat, C.nIter, 1, C.nBurnin, intercept, onepl, cov2one, 0, 0, C.qRt,
                    M.seed, 0, M.precision, 1, 0, 0, 0, 0, 0, 0.0)
    f = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:erm_farm_create, LIB[]), Cint, (Ref{ErmConfig}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}), cfg, devs, C.nChain, f))
    try
        Y = Array{UInt8}(M.Data.Y)
        logT = modelid(M) == MODEL_MLIRT ? Float64[] : dense(M.Data.logT)
        X = (modelid(M) in (MODEL_CROSSQR, MODEL_CROSS, MODEL_NULL) || C.nFeat == 0) ? Float64[] : dense(M.Data.X)
        GC.@preserve Y logT X check(ccall((:erm_farm_set_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}), f[], Y, ptr(logT), ptr(X)))
        first = M.Para
        for l in 1:C.nChain
            l > 1 && setInitialValues(M)                     # a fresh draw for every further chain
            arrs = state_arrays(M.Para)
            GC.@preserve arrs check(ccall((:erm_farm_set_state, LIB[]), Cint, (Ptr{Cvoid}, Int32, Ref{ErmState}), f[], l - 1, ErmState(map(ptr, arrs)...)))
        end
        M.Para = first
        check(ccall((:erm_farm_run, LIB[]), Cint, (Ptr{Cvoid}, Int64), f[], C.nIter))
        h0 = ccall((:erm_farm_engine, LIB[]), Ptr{Cvoid}, (Ptr{Cvoid}, Int32), f[], 0)
        trace(which) = begin
            w = ccall((:erm_trace_width, LIB[]), Int64, (Ptr{Cvoid}, Cint), h0, which)
            out = Array{Float64}(undef, C.nIter, w, C.nChain)
            check(ccall((:erm_farm_get_trace, LIB[]), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), f[], which, out)); out
        end
        Post = M.Post
        Post.ra = trace(TRACE_RA)
        Post.logLike = trace(TRACE_LOGLIKE)
        modelid(M) != MODEL_MLIRT && (Post.rt = trace(TRACE_RT))
        Post.qr = trace(TRACE_QR)
        N, J, F = C.nSubj, C.nItem, C.nFeat
        nb = modelid(M) == MODEL_MLIRT ? F + 1 : modelid(M) in (MODEL_RTIRT, MODEL_NULL) ? 2 * (F + 1) : modelid(M) in (MODEL_LATENTQR, MODEL_LATENT) ? F + 2 : 0
        nnu = modelid(M) == MODEL_LATENTQR ? N : modelid(M) == MODEL_CROSSQR ? N * J : 0
        bufs = (zeros(N), zeros(J), zeros(J), zeros(N), zeros(J), zeros(J), zeros(nb), zeros(4), zeros(J), zeros(nnu))
        GC.@preserve bufs check(ccall((:erm_farm_get_mean, LIB[]), Cint, (Ptr{Cvoid}, Ref{ErmState}), f[], ErmState(map(ptr, bufs)...)))
        Post.mean = InputPara(θ = bufs[1], a = bufs[2], b = bufs[3], ζ = bufs[4], λ = bufs[5], σ²t = bufs[6], β = bufs[7], Σp = bufs[8], ρ = bufs[9], ν = bufs[10])
    finally
        ccall((:erm_farm_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), f[])
    end
    return M
end

"""
    sample!(MCMC; intercept=false, itemtype="2pl", cov2one, devices=nothing)

Same contract as the reference's `sample!`: runs `Cond.nIter * Cond.nChain` sweeps of the interleaved loop, fills
`MCMC.Post.{ra,rt,qr,logLike,mean}`, leaves the final state in `MCMC.Para`, returns `MCMC`.
`devices = 0:7` runs the `Cond.nChain` chains as independent chains, one per listed GPU (cycled), instead: `Post` has the same shapes,
chain `l` in slab `l`; `Post.mean` is the joint mean over iterations and chains.
"""
function sample!(M::GibbsAMD; intercept = false, itemtype::Union{String} = "2pl",
                 cov2one = !(M isa GibbsRtIrtLatentQr || M isa GibbsRtIrtLatent), devices = nothing)
    if !(itemtype in ["1pl", "2pl"])
        error("Invalid input: the item type must be '1pl' or '2pl'.")
    end
    devices === nothing || return sampleFarm!(M, Bool(intercept), itemtype == "1pl", Bool(cov2one), collect(devices))
    C = M.Cond
    h = engine!(M, intercept, itemtype == "1pl", cov2one)
    check(ccall((:erm_reset_trace, LIB[]), Cint, (Ptr{Cvoid},), h))
    arrs = state_arrays(M.Para)
    GC.@preserve arrs begin
        st = ErmState(map(ptr, arrs)...)
        check(ccall((:erm_set_state, LIB[]), Cint, (Ptr{Cvoid}, Ref{ErmState}), h, st))
    end
    check(ccall((:erm_run, LIB[]), Cint, (Ptr{Cvoid}, Int64), h, C.nIter * C.nChain))

    trace(which) = begin
        w = ccall((:erm_trace_width, LIB[]), Int64, (Ptr{Cvoid}, Cint), h, which)
        out = Array{Float64}(undef, C.nIter, w, C.nChain)
        check(ccall((:erm_get_trace, LIB[]), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), h, which, out)); out
    end
    Post = M.Post
    Post.ra = trace(TRACE_RA)
    Post.logLike = trace(TRACE_LOGLIKE)
    modelid(M) != MODEL_MLIRT && (Post.rt = trace(TRACE_RT))
    Post.qr = trace(TRACE_QR)        # CrossQr: errors if vec(nu) per sweep exceeds erm_config.nu_trace_max_gb (then use erm_get_item_trace)

    N, J, F = C.nSubj, C.nItem, C.nFeat
    nb = modelid(M) == MODEL_MLIRT ? F + 1 : modelid(M) in (MODEL_RTIRT, MODEL_NULL) ? 2 * (F + 1) : modelid(M) in (MODEL_LATENTQR, MODEL_LATENT) ? F + 2 : 0
    nnu = modelid(M) == MODEL_LATENTQR ? N : modelid(M) == MODEL_CROSSQR ? N * J : 0
    bufs = (zeros(N), zeros(J), zeros(J), zeros(N), zeros(J), zeros(J), zeros(nb), zeros(4), zeros(J), zeros(nnu))
    GC.@preserve bufs begin
        check(ccall((:erm_get_mean, LIB[]), Cint, (Ptr{Cvoid}, Ref{ErmState}), h, ErmState(map(ptr, bufs)...)))
    end
    Post.mean = InputPara(θ = copy(bufs[1]), a = copy(bufs[2]), b = copy(bufs[3]), ζ = copy(bufs[4]), λ = copy(bufs[5]), σ²t = copy(bufs[6]),
                          β = copy(bufs[7]), Σp = copy(bufs[8]), ρ = copy(bufs[9]), ν = copy(bufs[10]))   # bufs are reused below
    GC.@preserve bufs begin
        check(ccall((:erm_get_state, LIB[]), Cint, (Ptr{Cvoid}, Ref{ErmState}), h, ErmState(map(ptr, bufs)...)))
    end
    P = M.Para
    P.θ, P.a, P.b = copy(bufs[1]), copy(bufs[2]), copy(bufs[3])
    if modelid(M) != MODEL_MLIRT
        P.ζ, P.λ, P.σ²t, P.Σp = copy(bufs[4]), copy(bufs[5]), copy(bufs[6]), reshape(copy(bufs[8]), 2, 2)
    end
    nb > 0 && (P.β = modelid(M) in (MODEL_RTIRT, MODEL_NULL) ? reshape(copy(bufs[7]), F + 1, 2) : copy(bufs[7]))
    modelid(M) in (MODEL_CROSSQR, MODEL_CROSS) && (P.ρ = copy(bufs[9]))
    modelid(M) == MODEL_CROSSQR && (P.ν = reshape(copy(bufs[10]), N, J))
    modelid(M) == MODEL_LATENTQR && (P.ν = copy(bufs[10]))
    return M
end

"""
    essRhat(MCMC, which) -> (ess, rhat)

Effective sample size and split R-hat of every column of `Post.ra` (`which = TRACE_RA`), `Post.rt` or `Post.qr`, computed on the device
from the resident traces (what `checkConvergence`, src/SimTools.jl:419-443, obtains from MCMCChains on the host).  Call after `sample!`.
"""
function essRhat(M::GibbsAMD, which::Integer)
    M.handle == C_NULL && error("run sample! first")
    w = ccall((:erm_trace_width, LIB[]), Int64, (Ptr{Cvoid}, Cint), M.handle, which)
    ess, rhat = zeros(w), zeros(w)
    check(ccall((:erm_get_diagnostics, LIB[]), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}), M.handle, which, ess, rhat))
    return ess, rhat
end

"""
    getDicDevice(MCMC) -> (Dbar, Dhat, pD, DIC)

`getDic` (src/GibbsRtIrt.pl.jl:432-472, src/GibbsRtIrtCross.pl.jl:330-353, src/GibbsRtIrtLatent.pl.jl:342-365) from device-resident state
(`erm_get_dic`): D̄ over every recorded logLike row, D̂ from one evaluation pass at `Post.mean` on the device.  Call after `sample!`.
"""
function getDicDevice(M::GibbsAMD)
    M.handle == C_NULL && error("run sample! first")
    out = zeros(4)
    check(ccall((:erm_get_dic, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), M.handle, out))
    return (Dbar = out[1], Dhat = out[2], pD = out[3], DIC = out[4])
end

"""
    checkConvergenceDevice(MCMC) -> (ess, rhat, essN, rhatN)

`checkConvergence` (src/SimTools.jl:419-443) with the ESS / R-hat of every column computed AND counted on the device (`erm_get_convergence`).
"""
function checkConvergenceDevice(M::GibbsAMD)
    M.handle == C_NULL && error("run sample! first")
    tot = zeros(Int64, 4)
    for which in (TRACE_RA, TRACE_RT, TRACE_QR)
        which == TRACE_RT && M isa GibbsMlIrt && continue
        c = zeros(Int64, 4)
        check(ccall((:erm_get_convergence, LIB[]), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), M.handle, which, c))
        tot .+= c
    end
    return (ess = 100 * tot[2] / max(tot[1], 1), rhat = 100 * tot[4] / max(tot[3], 1), essN = "$(tot[2]) / $(tot[1])", rhatN = "$(tot[4]) / $(tot[3])")
end

"""
    setSeed!(MCMC, seed)

A new seed for the chain's random streams (`erm_set_seed`): one sampler serves every replication of a `runSimulation` condition.
"""
function setSeed!(M::GibbsAMD, seed::Integer)
    M.seed = seed
    M.handle != C_NULL && check(ccall((:erm_set_seed, LIB[]), Cint, (Ptr{Cvoid}, UInt64), M.handle, UInt64(seed)))
    return M
end

"""
    simulateData!(MCMC, truePara; type="norm", seed=4321)

`setData*` on the device (src/SimTools.jl:117-368): draws X, theta, zeta, Y, logT from `truePara` straight into the engine's buffers;
`truePara.θ`, `truePara.ζ` receive the generated truth and `MCMC.Data` the data set.
"""
function simulateData!(M::GibbsAMD, truePara; type::String = "norm", seed = 4321)
    h = engine!(M, false, false, !(M isa GibbsRtIrtLatentQr || M isa GibbsRtIrtLatent); upload = false)
    arrs = state_arrays(truePara)
    GC.@preserve arrs begin
        check(ccall((:erm_simulate_data, LIB[]), Cint, (Ptr{Cvoid}, Ref{ErmState}, UInt64, Cint), h, ErmState(map(ptr, arrs)...), UInt64(seed),
                    type == "norm" ? 0 : type == "tail" ? 1 : 2))
    end
    C = M.Cond
    θ, ζ = zeros(C.nSubj), zeros(C.nSubj)
    check(ccall((:erm_get_truth, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h, θ, ζ))
    truePara.θ, truePara.ζ = θ, ζ
    Y = Array{UInt8}(undef, C.nSubj, C.nItem); logT = zeros(C.nSubj, C.nItem); X = zeros(C.nSubj, max(C.nFeat, 1))
    check(ccall((:erm_get_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}), h, Y, logT, X))
    M.Data = (Y = Y, κ = Y .- 0.5, T = exp.(logT), logT = logT, X = X[:, 1:C.nFeat])
    M.truePara = truePara
    return M
end

end # module
