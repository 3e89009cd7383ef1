# TTNBackend.jl — the Julia-side binding a TensorTrainNumerics.jl maintainer would add to route the
# Float64 hot path through libttn_hip.so (include/ttn.h).  NOT executed in this repository: the build
# image has no Julia runtime (see DESIGN.md §2); the same C ABI is exercised through ctypes by
# tensortrainnumerics.jl_amd/tt.py, which mirrors this file function for function.
#
# Only `TTvector{Float64}` / `TToperator{Float64}` methods are overloaded; every other eltype
# (ComplexF64, Float32, Int — test/test_tt_tools.jl:576-596) falls through to the generic Julia methods.
module TTNBackend

using TensorTrainNumerics
import TensorTrainNumerics: TTvector, TToperator, orthogonalize, tt_compress!, _tt_bond_truncate!, hadamard, add!, r_and_d_to_rks, zeros_tt,
    _applyH1_lsr, _applyH0, _update_left_env, _update_right_env, _applyH2_lsr
import Base: *, +

const LIB = get(ENV, "TTN_LIB", joinpath(@__DIR__, "..", "tensortrainnumerics.jl_amd", "libttn_hip.so"))

# return-code contract of include/ttn.h: 0 ok, <0 argument error (AssertionError in the reference), >0 HIP error
function _chk(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:ttn_last_error_string, LIB), Cstring, ()))
    (-4 <= rc <= -1) && throw(AssertionError(msg))      # tt_operations.jl:11,102,240,344; tt_tools.jl:513,744,773
    error("ttn error $rc: $msg")
end

_ptrs(v::Vector{<:Array{Float64}}) = Ptr{Float64}[pointer(c) for c in v]
_dims(x) = Int64[x...]

# *(A::TToperator, v::TTvector)  — src/tt_operations.jl:101-111
function *(A::TToperator{Float64, N}, v::TTvector{Float64, N}) where {N}
    @assert A.tto_dims == v.ttv_dims "Incompatible dimensions"
    y = zeros_tt(Float64, A.tto_dims, A.tto_rks .* v.ttv_rks)
    pa, px, py = _ptrs(A.tto_vec), _ptrs(v.ttv_vec), _ptrs(y.ttv_vec)
    GC.@preserve A v y pa px py _chk(ccall((:ttn_apply_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}),
        N, _dims(A.tto_dims), pa, A.tto_rks, px, v.ttv_rks, py))
    return y
end

# dot(A, B) — src/tt_operations.jl:239-250
function TensorTrainNumerics.dot(A::TTvector{Float64, N}, B::TTvector{Float64, N}) where {N}
    @assert A.ttv_dims == B.ttv_dims "TT dimensions are not compatible"
    out = Ref{Float64}(0.0)
    pa, pb = _ptrs(A.ttv_vec), _ptrs(B.ttv_vec)
    GC.@preserve A B pa pb _chk(ccall((:ttn_dot_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ref{Float64}),
        N, _dims(A.ttv_dims), pa, A.ttv_rks, pb, B.ttv_rks, out))
    return out[]
end

# hadamard(x, y) — src/tt_operations.jl:343-361
function hadamard(x::TTvector{Float64, N}, y::TTvector{Float64, N}) where {N}
    @assert x.ttv_dims == y.ttv_dims "Incompatible TT dimensions"
    z = zeros_tt(Float64, x.ttv_dims, x.ttv_rks .* y.ttv_rks)
    px, py, pz = _ptrs(x.ttv_vec), _ptrs(y.ttv_vec), _ptrs(z.ttv_vec)
    GC.@preserve x y z px py pz _chk(ccall((:ttn_hadamard_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}),
        N, _dims(x.ttv_dims), px, x.ttv_rks, py, y.ttv_rks, pz))
    return z
end

# +(x, y) — src/tt_operations.jl:10-35
function +(x::TTvector{Float64, N}, y::TTvector{Float64, N}) where {N}
    @assert x.ttv_dims == y.ttv_dims "Incompatible dimensions"
    rks = x.ttv_rks + y.ttv_rks; rks[1] = 1; rks[end] = 1
    z = zeros_tt(Float64, x.ttv_dims, rks)
    px, py, pz = _ptrs(x.ttv_vec), _ptrs(y.ttv_vec), _ptrs(z.ttv_vec)
    GC.@preserve x y z px py pz _chk(ccall((:ttn_add_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}),
        N, _dims(x.ttv_dims), px, x.ttv_rks, py, y.ttv_rks, pz))
    return z
end

# add!(x, y) — src/tt_operations.jl:37-66 (rebinds the fields of x)
function add!(x::TTvector{Float64, N}, y::TTvector{Float64, N}) where {N}
    z = x + y
    x.ttv_vec = z.ttv_vec; x.ttv_rks = z.ttv_rks; x.ttv_ot = z.ttv_ot
    return x
end

# a * x — src/tt_operations.jl:256-266 (`x * a`, `-`, `/` are one-liners on top of it in the reference, :268-295, and keep dispatching
# here).  a == 0 gives zeros_tt(...) with ot reset; otherwise the first core with ot == 0 (else core 1) is scaled and ot is copied.
function *(a::Float64, x::TTvector{Float64, N}) where {N}
    y = zeros_tt(Float64, x.ttv_dims, x.ttv_rks)
    yot = zeros(Int64, N)
    px, py = _ptrs(x.ttv_vec), _ptrs(y.ttv_vec)
    GC.@preserve x y px py _chk(ccall((:ttn_scale_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Float64, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}),
        N, _dims(x.ttv_dims), a, px, x.ttv_rks, x.ttv_ot, py, yot))
    y.ttv_ot .= yot
    return y
end

# _tt_bond_truncate!(ψ, k; max_bond, truncerr) — src/tt_tools.jl:743-770: mutates cores k, k+1 and ψ.ttv_rks[k+1] in place and
# RETURNS orthogonalize(ψ; i=k) (:769) like the reference (tt_compress! discards that value, so ttn_compress_f64 never computes it).
function _tt_bond_truncate!(ψ::TTvector{Float64, N}, k::Int; max_bond::Int = typemax(Int), truncerr::Real = 0.0) where {N}
    @assert(1 ≤ k < N, "k must be in 1:(N-1)")
    mb = min(max_bond, typemax(Int64) >> 1)
    need = zeros(Int64, N + 1)                      # the rank of a rank-deficient bond may grow to min(n r_left, n r_right, max_bond)
    _chk(ccall((:ttn_compress_rank_bound, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
        N, _dims(ψ.ttv_dims), ψ.ttv_rks, mb, 1, k, need, C_NULL))
    bufs = [zeros(Float64, ψ.ttv_dims[j] * need[j] * need[j + 1]) for j in 1:N]
    for j in 1:N
        copyto!(bufs[j], vec(ψ.ttv_vec[j]))
    end
    rks = copy(ψ.ttv_rks)
    pb = Ptr{Float64}[pointer(c) for c in bufs]
    GC.@preserve bufs pb _chk(ccall((:ttn_bond_truncate_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Int64, Int64, Float64),
        N, _dims(ψ.ttv_dims), pb, rks, k, mb, Float64(truncerr)))
    for j in (k, k + 1)                             # only these two slots change (:764-767)
        n = ψ.ttv_dims[j]
        ψ.ttv_vec[j] = reshape(bufs[j][1:(n * rks[j] * rks[j + 1])], n, rks[j], rks[j + 1])
    end
    ψ.ttv_rks[k + 1] = rks[k + 1]
    return orthogonalize(ψ; i = k)
end

# orthogonalize(x; i=1) — src/tt_tools.jl:511-543
function orthogonalize(x::TTvector{Float64, N}; i = 1::Int) where {N}
    @assert(1 ≤ i ≤ x.N, DimensionMismatch("Impossible orthogonalization"))
    y = zeros_tt(Float64, x.ttv_dims, x.ttv_rks)    # max-size buffers (output ranks never exceed the input's)
    yr = zeros(Int64, N + 1); yot = zeros(Int64, N)
    px, py = _ptrs(x.ttv_vec), _ptrs(y.ttv_vec)
    GC.@preserve x y px py _chk(ccall((:ttn_orthogonalize_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Int64, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Int64}),
        N, _dims(x.ttv_dims), px, x.ttv_rks, i, py, yr, yot))
    # the library writes each core compactly with the NEW ranks at the start of its buffer: re-wrap
    for k in 1:N
        n = x.ttv_dims[k]
        y.ttv_vec[k] = reshape(vec(y.ttv_vec[k])[1:(n * yr[k] * yr[k + 1])], n, yr[k], yr[k + 1])
    end
    y.ttv_rks .= yr; y.ttv_ot .= yot
    return y
end

# tt_compress!(ψ, max_bond; truncerr, sweeps, verbose) — src/tt_tools.jl:772-789.
# Mutates ψ.ttv_vec[k] slots and ψ.ttv_rks IN PLACE (QTTvector wrappers share those arrays,
# src/qtt_tools.jl:783-786) and returns ψ itself (test/test_tt_tools.jl:514).
function tt_compress!(ψ::TTvector{Float64, N}, max_bond::Int; truncerr::Real = 0.0, sweeps::Int = 1, verbose::Bool = false) where {N}
    @assert(sweeps ≥ 1, "sweeps must be >= 1")
    if verbose                               # the two log lines test/test_tt_tools.jl:572 asserts on
        for sw in 1:sweeps
            @info "TT compress: sweep $sw (L→R)"
            @info "TT compress: sweep $sw (R→L)"
        end
    end
    # a rank-deficient bond may grow to min(n r_left, n r_right, max_bond): size the in/out buffers for that
    need = zeros(Int64, N + 1)
    _chk(ccall((:ttn_compress_rank_bound, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}),
        N, _dims(ψ.ttv_dims), ψ.ttv_rks, min(max_bond, typemax(Int64) >> 1), sweeps, 0, need, C_NULL))
    bufs = [zeros(Float64, ψ.ttv_dims[k] * need[k] * need[k + 1]) for k in 1:N]
    for k in 1:N
        copyto!(bufs[k], vec(ψ.ttv_vec[k]))
    end
    rks = copy(ψ.ttv_rks)
    pb = Ptr{Float64}[pointer(c) for c in bufs]
    GC.@preserve bufs pb _chk(ccall((:ttn_compress_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Int64, Float64, Int64),
        N, _dims(ψ.ttv_dims), pb, rks, min(max_bond, typemax(Int64) >> 1), Float64(truncerr), sweeps))
    for k in 1:N
        n = ψ.ttv_dims[k]
        ψ.ttv_vec[k] = reshape(bufs[k][1:(n * rks[k] * rks[k + 1])], n, rks[k], rks[k + 1])
    end
    ψ.ttv_rks .= rks
    return ψ
end

# op = x -> tt_compress!(A * x, max_bond) (src/solvers/euler.jl:55: the operator krylov_linsolve hands to KrylovKit, and the pattern of
# every time stepper) as ONE stateless call: A * x is never materialised, neither in HBM nor over PCIe (16.8 ms against 21.9 ms
# for `*` followed by `tt_compress!` on one d = 30 rank-64 train).  Drop-in for the closure at euler.jl:55:
#     op = x -> apply_compress(A, x, max_bond)
function apply_compress(A::TToperator{Float64, N}, v::TTvector{Float64, N}, max_bond::Int; truncerr::Real = 0.0, sweeps::Int = 1) where {N}
    @assert A.tto_dims == v.ttv_dims "Incompatible dimensions"
    @assert(sweeps ≥ 1, "sweeps must be >= 1")
    mb = min(max_bond, typemax(Int64) >> 1)
    cap = zeros(Int64, N + 1)
    _chk(ccall((:ttn_apply_compress_rank_bound, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Ptr{Int64}),
        N, _dims(v.ttv_dims), A.tto_rks, v.ttv_rks, mb, sweeps, cap))
    bufs = [zeros(Float64, v.ttv_dims[k] * cap[k] * cap[k + 1]) for k in 1:N]
    rks = zeros(Int64, N + 1)
    pa, px = _ptrs(A.tto_vec), _ptrs(v.ttv_vec)
    pb = Ptr{Float64}[pointer(c) for c in bufs]
    GC.@preserve A v bufs pa px pb _chk(ccall((:ttn_apply_compress_f64, LIB), Cint,
        (Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Int64, Float64, Int64),
        N, _dims(v.ttv_dims), pa, A.tto_rks, px, v.ttv_rks, pb, rks, mb, Float64(truncerr), sweeps))
    y = zeros_tt(Float64, v.ttv_dims, rks)
    for k in 1:N
        n = v.ttv_dims[k]
        y.ttv_vec[k] = reshape(bufs[k][1:(n * rks[k] * rks[k + 1])], n, rks[k], rks[k + 1])
    end
    return y
end

# Device-resident chains (Krylov / RK4 inner loops, src/solvers/euler.jl:55,199-204) use the handle API:
#   h = Ref{Ptr{Cvoid}}(); ccall((:ttn_tt_create, LIB), Cint, (Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ref{Ptr{Cvoid}}), ...)
#   ccall((:ttn_apply_compress, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Int64), A, x, y, r, 0.0, 1)
# so that tt_compress!(A*x, r) never crosses PCIe (ttn_apply_compress fuses the apply into the first L->R sweep: A*x is
# never written to HBM).
#
# Core-wise sharded chains (one segment of the chain per GPU / Julia worker, INTEGRATION.md §5): a segment is an ordinary
# handle with open boundary ranks; one direction of a sweep over the local bonds and the boundary-core hand-off are
#   ccall((:ttn_sweep, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Float64), y, k_first, k_last, max_bond, truncerr)
#   ccall((:ttn_tt_core_extent, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), y, k, n, bl, br)
#   ccall((:ttn_tt_core_export, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int64}), y, k, devbuf, devrks)   # -> ncclSend
#   ccall((:ttn_tt_core_import, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Int64}, Int64, Int64), y, k, devbuf, devrks, bl, br)
# (k, k_first, k_last 1-based like _tt_bond_truncate!; devbuf / devrks are device pointers).
#
# Site-swap chains (INTEGRATION.md §2): hadamard_ttm and reorder run on handles,
#   ccall((:ttn_hadamard_ttm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64), x, y, z, tol, rmax, work_cap)
#   ccall((:ttn_swap_sites, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Float64), q, length(swaps), swaps, threshold)
# with swaps = _bubble_sort_swaps(perm) exactly as reorder computes it (src/qtt_tools.jl:759); ttn_compress_status reports a
# rank that outgrew its slot (-5) or a Jacobi SVD that hit its sweep limit (-9).
#
# Two-site solvers on handles (INTEGRATION.md §2): x receives the result, its capacity bounds the adapted ranks,
#   ccall((:ttn_mals_linsolve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64), A, b, x0, x, tol, rmax)
#   ccall((:ttn_dmrg_linsolve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Int64}),
#         A, b, x0, x, tol, length(sweep_schedule), sweep_schedule, rmax_schedule)      # dmrg_linsolve(...; N = 2), dmrg.jl:388
#   ccall((:ttn_dmrg_linsolve_it, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Int64, Float64, Int64),
#         A, b, x0, x, tol, length(sweep_schedule), sweep_schedule, rmax_schedule, it_solver, linsolv_maxiter, linsolv_tol, itslv_thresh)
# (the keyword form, dmrg.jl:392-396: local systems above itslv_thresh unknowns — or all of them with it_solver — by matrix-free CG).
#
# Status is sticky per handle: ttn_compress_status(h, C_NULL) returns the first error any compress / sweep / swap / solver call
# recorded on h since the last query (capacity -5, Jacobi sweep limit -9, singular local system -10) and clears it.
# ccall((:ttn_status_all, LIB), Cint, ()) answers for every live handle and for handles freed with an unread code, in one synchronisation.

# ---- TDVP local contractions (src/solvers/tdvp.jl:29-43, :205-208) ------------------------------------------------------------
# The five @tensor kernels of tdvp1sweep! / tdvp2sweep!, for Float64 and ComplexF64 arrays in the layouts the sweeps hold
# (sites (l,s,r), operator cores (a,s,b,s')).  KrylovKit's exponentiate keeps calling them as closures, unchanged.
const _TE = Union{Float64, ComplexF64}
_cplx(::Type{Float64}) = Cint(0)
_cplx(::Type{ComplexF64}) = Cint(1)
_f64ptr(X::Array{T}) where {T <: _TE} = Ptr{Float64}(pointer(X))

function _tdvp_contract(op::Int, ::Type{T}, dims7::NTuple{7, Int}, FL, FR, X, M1, M2, out::Array{T}) where {T <: _TE}
    d7 = Int64[dims7...]
    p(Z) = Z === nothing ? Ptr{Float64}(C_NULL) : _f64ptr(Z)
    GC.@preserve FL FR X M1 M2 out d7 _chk(ccall((:ttn_tdvp_contract_f64, LIB), Cint,
        (Cint, Cint, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint),
        op, _cplx(T), 1, d7, p(FL), p(FR), p(X), p(M1), p(M2), _f64ptr(out), 0))
    return out
end

function _applyH1_lsr(AC::Array{T, 3}, FL::Array{T, 3}, FR::Array{T, 3}, M::Array{T, 4}) where {T <: _TE}
    Dl, d, Dr = size(AC); a, b = size(M, 1), size(M, 3)
    return _tdvp_contract(0, T, (Dl, d, Dr, a, b, 1, 1), FL, FR, AC, M, nothing, Array{T}(undef, Dl, d, Dr))
end

function _applyH0(C::Array{T, 2}, FL::Array{T, 3}, FR::Array{T, 3}) where {T <: _TE}
    Dl, Dr = size(C); a = size(FL, 2)
    return _tdvp_contract(1, T, (Dl, 1, Dr, a, 1, 1, 1), FL, FR, C, nothing, nothing, Array{T}(undef, Dl, Dr))
end

function _update_left_env(A::Array{T, 3}, M::Array{T, 4}, FL::Array{T, 3}) where {T <: _TE}
    Dl, d, Dr = size(A); a_in, a_out = size(M, 1), size(M, 3)
    return _tdvp_contract(2, T, (Dl, d, Dr, a_in, a_out, 1, 1), FL, nothing, A, M, nothing, Array{T}(undef, Dr, a_out, Dr))
end

function _update_right_env(A::Array{T, 3}, M::Array{T, 4}, FR::Array{T, 3}) where {T <: _TE}
    Dl, d, Dr = size(A); a_out, a_in = size(M, 1), size(M, 3)
    return _tdvp_contract(3, T, (Dl, d, Dr, a_in, a_out, 1, 1), nothing, FR, A, M, nothing, Array{T}(undef, Dl, a_out, Dl))
end

function _applyH2_lsr(AAC::Array{T, 4}, FL::Array{T, 3}, FR::Array{T, 3}, M1::Array{T, 4}, M2::Array{T, 4}) where {T <: _TE}
    Dl, d1, d2, Dr = size(AAC); a, b, c = size(M1, 1), size(M1, 3), size(M2, 3)
    return _tdvp_contract(4, T, (Dl, d1, Dr, a, b, c, d2), FL, FR, AAC, M1, M2, Array{T}(undef, Dl, d1, d2, Dr))
end

end # module
