# ParticleMDIHIP.jl -- Julia glue for libpmdi_hip.so (include/pmdi_hip.h).
#
# STATUS: written against the C ABI but NOT executed anywhere in this repository's CI: neither
# the build container nor the GPU box has a julia binary (DESIGN.md).  The Python twin of this
# file (particlemdi.jl_amd/pmdi.py) drives the same ABI and is what the GPU tests exercise.
#
# Two drivers: `pmdi` replaces only the sweep (the reference's own Julia hyper-parameter functions stay, as the
# north star describes); `pmdi_device` runs the whole iteration on the device (pmdi_gibbs_*, pmdi_csv_*).
#
# What it does: `ParticleMDIHIP.pmdi(...)` has the signature, the asserts and the CSV output of
# `ParticleMDI.pmdi` (src/pmdi.jl:36-40, 50-55, 147-158, 377-383).  The hyper-parameter
# updates and label alignment are the reference's own functions, called from the installed
# ParticleMDI package (update_M!, update_γ!, update_Φ!, update_Z, update_v, align_labels!);
# only the block src/pmdi.jl:165-171,188-370 -- reset, known prefix, the conditional-SMC sweep,
# particle pick and feature selection -- is replaced by two ccalls.  If any dataTypes[k] is
# not one of the three built-in cluster types (a user-defined plugin type, README.md:48-88),
# the call is forwarded unchanged to ParticleMDI.pmdi: user types keep working on the
# reference's CPU loop.
module ParticleMDIHIP

using ParticleMDI
using DelimitedFiles, Distributions, Printf, Random
using NonUniformRandomVariateGeneration: sampleCategorical

const LIB = get(ENV, "PMDI_HIP_LIB", joinpath(@__DIR__, "..", "libpmdi_hip.so"))
const PMDI_ABI_VERSION = Int32(2)

# mirrors of the C structs in include/pmdi_hip.h
struct CDataset
    kind::Int32
    D::Int32
    ld::Int64
    xf::Ptr{Float64}
    xi::Ptr{Int64}
end

struct CConfig
    abi_version::Int32
    device::Int32
    K::Int32
    N::Int32
    P::Int32
    n_chains::Int32
    n::Int64
    seed::UInt64
    q1_mode::Int32
    q2_mode::Int32
    pool_cap::Int64
    block_threads::Int32
    reserved::Int32
    tuning::Ptr{Cvoid}      # const pmdi_tuning * (kernel-selection knobs); C_NULL = all automatic
end

struct CSweepStats
    n_operations::Int64
    n_resamples::Int64
    n_clones::Int64
    max_id::Int64
    sum_classes::Int64
    steps_fast::Int64
    steps_converted::Int64
    steps_fallback::Int64
end

device_kind(::Type{ParticleMDI.GaussianCluster}) = Int32(0)
device_kind(::Type{ParticleMDI.CategoricalCluster}) = Int32(1)
device_kind(::Type{ParticleMDI.NegBinomCluster}) = Int32(2)
device_kind(::Any) = Int32(-1)

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:pmdi_last_error, LIB), Cstring, ()))
    error("libpmdi_hip: [$rc] $msg")
end

"""
    pmdi(dataFiles, dataTypes, N, particles, ρ, iter, outputFile; thin, featureSelect, dataNames,
         seed = rand(UInt64), device = 0)

Drop-in for `ParticleMDI.pmdi`; the sweep runs on an MI355X.
"""
function pmdi(dataFiles, dataTypes, N::Int64, particles::Int64, ρ::Float64, iter::Int64,
              outputFile::String; thin::Int64 = 1, featureSelect::Union{String, Nothing} = nothing,
              dataNames = nothing, seed::UInt64 = rand(UInt64), device::Integer = 0)
    kinds = [device_kind(t) for t in dataTypes]
    if any(k -> k < 0, kinds)
        # a user-defined cluster type: the reference's own loop handles it
        return ParticleMDI.pmdi(dataFiles, dataTypes, N, particles, ρ, iter, outputFile;
                                thin = thin, featureSelect = featureSelect, dataNames = dataNames)
    end
    K = length(dataFiles)
    n_obs = size(dataFiles[1], 1)
    dataNames === nothing && (dataNames = ["K$i" for i in 1:K])
    @assert length(dataTypes) == K "Number of datatypes not equal to number of datasets"
    @assert length(dataNames) == K "Number of data names not equal to number of datasets"
    @assert all(size(d, 1) == n_obs for d in dataFiles) "Datasets don't have same number of observations. Each row must correspond to the same underlying observational unit across datasets."
    @assert 0 < ρ < 1 "ρ must be between 0 and 1"
    @assert 1 < N <= n_obs "Number of clusters must be greater than 1 and not greater than the number of observations"
    @assert particles > 1 "Conditional particle filter requires 2 or more particles"
    n1 = floor(Int64, ρ * n_obs)
    @assert n1 >= 1 "floor(ρ·n) must be at least 1"

    # ---- device handle: data are copied to the GPU once ----
    mats = [kinds[k] == 0 ? convert(Matrix{Float64}, dataFiles[k]) : convert(Matrix{Int64}, dataFiles[k]) for k in 1:K]
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve mats begin
        ds = [CDataset(kinds[k], Int32(size(mats[k], 2)), Int64(n_obs),
                       kinds[k] == 0 ? pointer(mats[k]) : Ptr{Float64}(C_NULL),
                       kinds[k] == 0 ? Ptr{Int64}(C_NULL) : pointer(mats[k])) for k in 1:K]
        cfg = Ref(CConfig(PMDI_ABI_VERSION, Int32(device), Int32(K), Int32(N), Int32(particles), Int32(1),
                          Int64(n_obs), seed, Int32(0), Int32(0), Int64(0), Int32(0), Int32(0), C_NULL))
        check(ccall((:pmdi_create, LIB), Cint, (Ref{CConfig}, Ptr{CDataset}, Ref{Ptr{Cvoid}}), cfg, ds, handle))
    end
    h = handle[]

    try
        # ---- host state, exactly as the reference initialises it (src/pmdi.jl:59-96) ----
        M = fill(2.0, K)
        γc = rand(Gamma(1.0 / N, 1.0), N, K) .+ eps(Float64)
        npairs = K > 1 ? div(K * (K - 1), 2) : 1
        Φ = K > 1 ? rand(Gamma(1, 0.2), npairs) : zeros(1)
        s = Matrix{Int64}(undef, n_obs, K)
        for k in 1:K
            s[:, k] = sampleCategorical(n_obs, γc[:, k])
        end
        c_combn = Matrix{Int64}(undef, N^K, K)
        for k in 1:K
            c_combn[:, K - k + 1] = div.(0:(N^K - 1), N^(K - k)) .% N .+ 1
        end
        Γc = Matrix{Float64}(undef, N^K, K)
        for k in 1:K
            Γc[:, k] = log.(γc[:, k])[c_combn[:, k]]
        end
        Φ_index = K > 1 ? Matrix{Bool}(undef, N^K, npairs) : fill(1, (N, 1))
        if K > 1
            col = 1
            for k1 in 1:(K - 1), k2 in (k1 + 1):K
                Φ_index[:, col] = c_combn[:, k1] .== c_combn[:, k2]
                col += 1
            end
        end
        Z = ParticleMDI.update_Z(Φ, Φ_index, Γc)
        v = ParticleMDI.update_v(n_obs, Z)

        D = [size(d, 2) for d in dataFiles]
        flags = featureSelect === nothing ? ones(UInt8, sum(D)) : UInt8.(rand(Bool, sum(D)))
        featureFile = nothing
        if featureSelect !== nothing
            names = ["$(dataNames[k])_d$d" for k in 1:K for d in 1:D[k]]
            writedlm(featureSelect, reshape(names, 1, :), ',')
            featureFile = open(featureSelect, "a")
            writedlm(featureFile, reshape(Bool.(flags), 1, :), ',')
        end

        Φ_lab = ParticleMDI.calculate_Φ_lab(K)
        header = [[@sprintf("MassParameter_%d", k) for k in 1:K];
                  [@sprintf("phi_%d_%d", Φ_lab[i, 1], Φ_lab[i, 2]) for i in 1:size(Φ_lab, 1)];
                  "ll";
                  ["$(dataNames[k])_n$i" for k in 1:K for i in 1:n_obs]]
        writedlm(outputFile, reshape(header, 1, :), ',')
        fileid = open(outputFile, "a")
        t0 = time_ns()
        writedlm(fileid, [M; Φ; 0; s[1:(n_obs * K)]]', ',')

        order_obs = collect(1:n_obs)
        s_out = similar(s)
        logweight = zeros(Float64, particles)
        p_star = Ref{Int64}(0)
        stats = Ref(CSweepStats(0, 0, 0, 0, 0, 0, 0, 0))
        flags_out = similar(flags)
        for it in 1:iter
            shuffle!(order_obs)                                        # src/pmdi.jl:172
            ParticleMDI.update_M!(M, γc, K, N)                          # :176
            ParticleMDI.update_γ!(γc, Φ, v, M, s, Φ_index, c_combn, Γc, N, K)
            Π = γc ./ sum(γc, dims = 1)                                 # :179
            K > 1 && ParticleMDI.update_Φ!(Φ, v, s, Φ_index, γc, K, Γc)
            Z = ParticleMDI.update_Z(Φ, Φ_index, Γc)
            v = ParticleMDI.update_v(n_obs, Z)

            # ---- replaces src/pmdi.jl:165-171, 188-350, 373 ----
            check(ccall((:pmdi_sweep, LIB), Cint,
                        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8},
                         Float64, Ptr{Int64}, Ptr{Float64}, Ref{Int64}, Ref{CSweepStats}, Ptr{Float64}),
                        h, it, s, order_obs, n1, Π, Φ, flags, it == 1 ? 0.0 : 1.0,
                        s_out, logweight, p_star, stats, C_NULL))
            if featureSelect !== nothing
                # ---- replaces src/pmdi.jl:354-370 ----
                check(ccall((:pmdi_feature_select, LIB), Cint,
                            (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{UInt8}, Ptr{Float64}),
                            h, it, s_out, flags_out, C_NULL))
                flags .= flags_out
            end
            s .= s_out                                                  # :373
            ParticleMDI.align_labels!(s, Φ, γc, N, K)                   # :375

            ll = (time_ns() - t0) / 1.0e9                               # :377
            if it % thin == 0
                writedlm(fileid, [M; Φ; ll; s[1:(n_obs * K)]]', ',')
                featureFile !== nothing && writedlm(featureFile, reshape(Bool.(flags), 1, :), ',')
            end
        end
        close(fileid)
        featureFile !== nothing && close(featureFile)
    finally
        ccall((:pmdi_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    end
    return
end

"""
    pmdi_device(dataFiles, dataTypes, N, particles, ρ, iter, outputFile; thin, featureSelect, dataNames,
                seed = rand(UInt64), device = 0, q2_mode = 0)

The same run with EVERYTHING of an iteration on the MI355X (include/pmdi_hip.h, pmdi_gibbs_*): shuffle!(order_obs),
update_M!, update_γ!, update_Φ!, update_Z, update_v (evaluated without the N^K tables of src/pmdi.jl:69-92), the sweep,
feature selection and align_labels! (contingency tables) are device kernels; the CSV rows are written by the library's
byte-compatible writer (pmdi_csv_*).  Host-side draws become counter-based Philox variates keyed on `seed`
(SURVEY 8 rows f1, f2, f4).  `pmdi` above keeps the reference's own Julia functions for those parts.
"""
function pmdi_device(dataFiles, dataTypes, N::Int64, particles::Int64, ρ::Float64, iter::Int64,
                     outputFile::String; thin::Int64 = 1, featureSelect::Union{String, Nothing} = nothing,
                     dataNames = nothing, seed::UInt64 = rand(UInt64), device::Integer = 0, q2_mode::Integer = 0)
    kinds = [device_kind(t) for t in dataTypes]
    any(k -> k < 0, kinds) && return ParticleMDI.pmdi(dataFiles, dataTypes, N, particles, ρ, iter, outputFile;
                                                      thin = thin, featureSelect = featureSelect, dataNames = dataNames)
    K = length(dataFiles)
    n_obs = size(dataFiles[1], 1)
    dataNames === nothing && (dataNames = ["K$i" for i in 1:K])
    @assert length(dataNames) == K "Number of data names not equal to number of datasets"
    @assert length(dataTypes) == K "Number of datatypes not equal to number of datasets"
    @assert all(size(d, 1) == n_obs for d in dataFiles) "Datasets don't have same number of observations. Each row must correspond to the same underlying observational unit across datasets."
    @assert 0 < ρ < 1 "ρ must be between 0 and 1"
    @assert 1 < N <= n_obs "Number of clusters must be greater than 1 and not greater than the number of observations"
    @assert particles > 1 "Conditional particle filter requires 2 or more particles"
    mats = [kinds[k] == 0 ? convert(Matrix{Float64}, dataFiles[k]) : convert(Matrix{Int64}, dataFiles[k]) for k in 1:K]
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve mats begin
        ds = [CDataset(kinds[k], Int32(size(mats[k], 2)), Int64(n_obs),
                       kinds[k] == 0 ? pointer(mats[k]) : Ptr{Float64}(C_NULL),
                       kinds[k] == 0 ? Ptr{Int64}(C_NULL) : pointer(mats[k])) for k in 1:K]
        cfg = Ref(CConfig(PMDI_ABI_VERSION, Int32(device), Int32(K), Int32(N), Int32(particles), Int32(1),
                          Int64(n_obs), seed, Int32(0), Int32(q2_mode), Int64(0), Int32(0), Int32(0), C_NULL))
        check(ccall((:pmdi_create, LIB), Cint, (Ref{CConfig}, Ptr{CDataset}, Ref{Ptr{Cvoid}}), cfg, ds, handle))
    end
    h = handle[]
    g = Ref{Ptr{Cvoid}}(C_NULL)
    csv = Ref{Ptr{Cvoid}}(C_NULL)
    fcsv = Ref{Ptr{Cvoid}}(C_NULL)
    try
        check(ccall((:pmdi_gibbs_create, LIB), Cint, (Ptr{Cvoid}, Float64, Int32, Ref{Ptr{Cvoid}}),
                    h, ρ, featureSelect === nothing ? 0 : 1, g))                 # src/pmdi.jl:59-66, 95-96, 106-110
        # own the strings whose pointers cross the ABI (String(nm) of a SubString or Symbol is a temporary: rooting
        # `dataNames` would not keep it alive)
        strs = String[String(nm) for nm in dataNames]
        names = [Base.unsafe_convert(Cstring, st) for st in strs]
        GC.@preserve strs begin
            check(ccall((:pmdi_csv_open, LIB), Cint, (Cstring, Int32, Int64, Ptr{Cstring}, Ref{Ptr{Cvoid}}),
                        outputFile, K, n_obs, names, csv))                       # :147-156
            if featureSelect !== nothing
                D = Int32[size(d, 2) for d in dataFiles]
                check(ccall((:pmdi_csv_open_features, LIB), Cint, (Cstring, Int32, Ptr{Int32}, Ptr{Cstring}, Ref{Ptr{Cvoid}}),
                            featureSelect, K, D, names, fcsv))                   # :111
            end
        end
        flags = Vector{UInt8}(undef, sum(size(d, 2) for d in dataFiles))
        write_flags() = begin
            check(ccall((:pmdi_gibbs_get, LIB), Cint,
                        (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                        g[], 0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, flags))
            check(ccall((:pmdi_csv_write_flags, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}), fcsv[], flags))
        end
        featureSelect === nothing || write_flags()                                # :116
        t0 = time_ns()
        check(ccall((:pmdi_csv_write_gibbs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64), csv[], g[], 0, 0.0))   # :158
        stats = Vector{Int64}(undef, 8)
        for it in 1:iter
            check(ccall((:pmdi_gibbs_iterate, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{Cvoid}), g[], 1, C_NULL, C_NULL))   # :165-375
            check(ccall((:pmdi_gibbs_results, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Int64}, Ptr{Float64}),
                        g[], stats, C_NULL, C_NULL, C_NULL))                     # synchronises; a kernel-side error surfaces here
            ll = (time_ns() - t0) / 1.0e9                                        # :377
            if it % thin == 0
                check(ccall((:pmdi_csv_write_gibbs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64), csv[], g[], 0, ll))   # :379
                featureSelect === nothing || write_flags()                       # :381
            end
        end
    finally
        csv[] == C_NULL || ccall((:pmdi_csv_close, LIB), Cint, (Ptr{Cvoid},), csv[])
        fcsv[] == C_NULL || ccall((:pmdi_csv_close, LIB), Cint, (Ptr{Cvoid},), fcsv[])
        g[] == C_NULL || ccall((:pmdi_gibbs_destroy, LIB), Cint, (Ptr{Cvoid},), g[])
        ccall((:pmdi_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    end
    return
end

# ---- generate_psm's reading of an output file (src/output_analysis/consensus_map.jl:32-47), natively ----------
# returns (samples::Array{UInt8,3} of size (n_obs, K, rows kept) -- i.e. [row][k][i] in C order --, names)
function read_allocations(outputFile::String, burnin::Int64 = 0, thin::Int64 = 1)
    K = Ref{Int32}(0); n_obs = Ref{Int64}(0); n_iter = Ref{Int64}(0)
    names = Vector{UInt8}(undef, 1 << 16)
    check(ccall((:pmdi_csv_read_allocations, LIB), Cint,
                (Cstring, Int64, Int64, Ref{Int32}, Ref{Int64}, Ref{Int64}, Ptr{UInt8}, Int64, Ptr{UInt8}, Int32),
                outputFile, burnin, thin, K, n_obs, n_iter, C_NULL, 0, names, length(names)))
    samples = Array{UInt8, 3}(undef, n_obs[], K[], n_iter[])
    check(ccall((:pmdi_csv_read_allocations, LIB), Cint,
                (Cstring, Int64, Int64, Ref{Int32}, Ref{Int64}, Ref{Int64}, Ptr{UInt8}, Int64, Ptr{UInt8}, Int32),
                outputFile, burnin, thin, K, n_obs, n_iter, samples, length(samples), C_NULL, 0))
    return samples, split(unsafe_string(pointer(names)), '\n')
end

# ---- the cluster plugin protocol on the device (unit-level entry points) ----------------------
# calc_logprob / cluster_add! / calc_logmarginal for a batch of stand-alone clusters of dataset k;
# see pmdi_clusters_new, pmdi_cluster_add, pmdi_calc_logprob, pmdi_calc_logmarginal in the header.
function clusters_new(h::Ptr{Cvoid}, k::Integer, B::Integer)
    cb = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmdi_clusters_new, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ref{Ptr{Cvoid}}), h, k - 1, B, cb))
    return cb[]
end
cluster_add!(cb::Ptr{Cvoid}, rows::Vector{Int64}, featureFlag::Vector{UInt8}) =
    check(ccall((:pmdi_cluster_add, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}), cb, rows, featureFlag))
function calc_logprob(cb::Ptr{Cvoid}, rows::Vector{Int64}, featureFlag::Vector{UInt8})
    out = Vector{Float64}(undef, length(rows))
    check(ccall((:pmdi_calc_logprob, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{Float64}), cb, rows, featureFlag, out))
    return out
end

end # module
