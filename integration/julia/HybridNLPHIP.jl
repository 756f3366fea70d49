# HybridNLPHIP.jl -- the reference-side binding: a drop-in MOI.AbstractNLPEvaluator that forwards the
# callbacks of /root/reference/src/moi.jl:1-33 to libqln_hip.so (C ABI: include/qln_evaluator.h).
#
# NOT EXECUTED in this pipeline (no Julia on either box) -- kept free of arithmetic so that review by
# reading is credible: every evaluator method is one ccall.  Usage inside the reference's notebook, after
# `include("nlp.jl")`, `include("moi.jl")` etc.:
#     nlp = HybridNLPHIP(model, obj, init_mode, k_trans, N, xinit, xterm)
#     Z_sol, solver = solve(Z0, nlp, c_tol=1e-3, tol=1e-3)    # dispatches to the method at the END OF THIS FILE
# The reference's own `solve` is typed `solve(x0, prob::HybridNLP; ...)` (src/moi.jl:46) and HybridNLP is a concrete
# struct (src/nlp.jl:13), so it cannot take this type; this file therefore adds a second METHOD of the same generic
# function `solve` for HybridNLPHIP (same keywords, same Ipopt options, same bounds) -- src/moi.jl itself is not edited.
# tests/test_julia_binding.py holds every ccall and struct of this file to include/qln_evaluator.h (symbol, argument
# count, pointer / integer / double class of every argument, field order and width of the three structs).
using LinearAlgebra        # diag
using MathOptInterface
const MOI = MathOptInterface
const LIBQLN = get(ENV, "QLN_LIB", joinpath(@__DIR__, "..", "..", "quadruped_landing_amd", "csrc", "libqln_hip.so"))

struct QlnModel; g::Cdouble; mb::Cdouble; mf::Cdouble; lb::Cdouble; l1::Cdouble; l2::Cdouble; end
struct QlnBatchDesc
    B::Int32; N::Int32; model::QlnModel
    k_trans::Ptr{Int32}; init_mode::Ptr{Int32}; x0::Ptr{Cdouble}; xf::Ptr{Cdouble}; cost::Ptr{Cdouble}
    cost_batch::Int32; z_stride::Int64; align::Int32
    jac_format::Int32      # 0 = dense 15x20 step blocks, 1 = structural non-zeros only (QLN_JAC_FORMAT_*)
end

qln_check(rc) = rc == 0 || error("libqln_hip: " * unsafe_string(ccall((:qln_last_error, LIBQLN), Cstring, ())))

mutable struct HybridNLPHIP <: MOI.AbstractNLPEvaluator
    handle::Ptr{Cvoid}
    N::Int; k_trans::Int; init_mode::Int
    lb::Vector{Float64}; ub::Vector{Float64}      # fields solve() reads (src/moi.jl:69)
    n_nlp::Int; m_nlp::Int
    use_sparse_jacobian::Bool
end

# HybridNLP(model, obj, init_mode, k_trans, N, x0, xf) -- src/nlp.jl:34-37.  `obj` is the reference's
# Vector{QuadraticCost}; it is flattened to the 41-double records [Q(15) R(5) q(15) r(5) c].
function HybridNLPHIP(model, obj, init_mode, k_trans, N, x0, xf; use_sparse_jacobian=false, device=0)
    # a sparse solve wants only the entries that can be non-zero; the dense callback needs neither format in particular
    jac_format = use_sparse_jacobian ? 1 : 0
    cost = vcat([[diag(o.Q); diag(o.R); o.q; o.r; o.c] for o in obj]...)
    kt = Int32[k_trans]; im = Int32[init_mode]; x0v = collect(Float64, x0); xfv = collect(Float64, xf)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve cost kt im x0v xfv begin
        d = Ref(QlnBatchDesc(1, N, QlnModel(model.g, model.mb, model.mf, model.lb, model.l1, model.l2),
                             pointer(kt), pointer(im), pointer(x0v), pointer(xfv), pointer(cost), 1, 0, 0, jac_format))
        qln_check(ccall((:qln_create, LIBQLN), Cint, (Ref{QlnBatchDesc}, Cint, Ref{Ptr{Cvoid}}), d, device, h))
    end
    m = Ref{Int32}(0); nnz = Ref{Int32}(0)
    qln_check(ccall((:qln_problem_dims, LIBQLN), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ref{Int32}), h[], 0, m, nnz))
    lb = zeros(m[]); ub = zeros(m[])
    qln_check(ccall((:qln_constraint_bounds, LIBQLN), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Cdouble}), h[], 0, lb, ub))
    nlp = HybridNLPHIP(h[], N, k_trans, init_mode, lb, ub, 20N - 5, m[], use_sparse_jacobian)
    finalizer(p -> ccall((:qln_destroy, LIBQLN), Cint, (Ptr{Cvoid},), p.handle), nlp)
end

num_primals(nlp::HybridNLPHIP) = nlp.n_nlp      # src/nlp.jl:86
num_duals(nlp::HybridNLPHIP) = nlp.m_nlp        # src/nlp.jl:87

function MOI.eval_objective(prob::HybridNLPHIP, x)                      # src/moi.jl:1-3
    f = Ref{Cdouble}(0.0)
    qln_check(ccall((:qln_eval_objective_host, LIBQLN), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{Cdouble}), prob.handle, x, f))
    return f[]
end
function MOI.eval_objective_gradient(prob::HybridNLPHIP, grad_f, x)     # src/moi.jl:5-8
    qln_check(ccall((:qln_eval_objective_gradient_host, LIBQLN), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), prob.handle, x, grad_f))
    return nothing
end
function MOI.eval_constraint(prob::HybridNLPHIP, g, x)                  # src/moi.jl:10-13
    qln_check(ccall((:qln_eval_constraint_host, LIBQLN), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), prob.handle, x, g))
    return nothing
end
function MOI.eval_constraint_jacobian(prob::HybridNLPHIP, vec, x)       # src/moi.jl:15-24
    if prob.use_sparse_jacobian   # vec has one slot per entry of jacobian_structure (block-COO order)
        qln_check(ccall((:qln_eval_constraint_jacobian_host, LIBQLN), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), prob.handle, x, vec))
    else                          # dense column-major m_nlp x n_nlp; only jac_c!'s write-set is assigned
        qln_check(ccall((:qln_eval_constraint_jacobian_dense_host, LIBQLN), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Cdouble}), prob.handle, 0, x, vec))
    end
    return nothing
end
MOI.features_available(prob::HybridNLPHIP) = [:Grad, :Jac]              # src/moi.jl:26-28
MOI.initialize(prob::HybridNLPHIP, features) = nothing                  # src/moi.jl:30
function MOI.jacobian_structure(nlp::HybridNLPHIP)                      # src/moi.jl:31-33
    if !nlp.use_sparse_jacobian
        return vec(Tuple.(CartesianIndices(zeros(num_duals(nlp), num_primals(nlp)))))
    end
    nnz = Ref{Int32}(0)
    qln_check(ccall((:qln_problem_dims, LIBQLN), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ref{Int32}), nlp.handle, 0, C_NULL, nnz))
    rows = zeros(Int32, nnz[]); cols = zeros(Int32, nnz[])
    qln_check(ccall((:qln_jacobian_structure, LIBQLN), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}), nlp.handle, 0, rows, cols))
    return [(Int(r) + 1, Int(c) + 1) for (r, c) in zip(rows, cols)]     # the ABI is 0-based
end

# ---- beyond the evaluator: the same NLP solved on the GPU in place of `solve(Z0, nlp)` (src/moi.jl:46-103) -------------
# qln_solve_host: augmented-Lagrangian iLQR, one wavefront per problem (DESIGN.md 4.6); objective, constraint bounds and the
# variable bounds of solve() (quirk Q6 included) are the reference's.  Field order = include/qln_evaluator.h.
struct QlnSolveOptions
    max_outer::Int32; max_inner::Int32
    tol_violation::Cdouble; inner_tol::Cdouble
    rho0::Cdouble; rho_factor::Cdouble; rho_max::Cdouble
    h_min::Cdouble; h_max::Cdouble; theta_min::Cdouble; theta_max::Cdouble
    q6_bounds::Int32; exact_h_gradient::Int32
    h_prox::Cdouble
    rescue_outer::Int32
end

function solve_hip(x0, prob::HybridNLPHIP; c_tol=1.0e-6)
    opt = Ref{QlnSolveOptions}()
    qln_check(ccall((:qln_solve_default_options, LIBQLN), Cint, (Ref{QlnSolveOptions},), opt))
    o = opt[]
    opt[] = QlnSolveOptions(o.max_outer, o.max_inner, c_tol, o.inner_tol, o.rho0, o.rho_factor, o.rho_max, o.h_min, o.h_max,
                            o.theta_min, o.theta_max, o.q6_bounds, o.exact_h_gradient, o.h_prox, o.rescue_outer)
    Z = collect(Float64, x0)             # in: initial guess (its controls are used); out: the solution
    info = zeros(16)                     # {outer, iLQR iterations, f, violation, rho, status, ...}
    qln_check(ccall((:qln_solve_host, LIBQLN), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{QlnSolveOptions}, Ptr{Cdouble}),
                    prob.handle, Z, opt, info))
    return Z, info
end

# ---- the reference's Ipopt solve, for this evaluator type: a method of `solve` (src/moi.jl:46-103) ------------------------
# Same generic function, same keyword arguments and defaults, same five things handed to Ipopt.  What differs from the
# HybridNLP method, on purpose:
#   * the variable bounds come from the library (qln_variable_bounds: theta, h and the two lower bounds of quirk Q6 at the
#     reference's own indices 22+20(k-1), 24+20(k-1)) instead of being restated here;
#   * the evaluator Ipopt calls back is `prob` itself.  The reference's MOI.eval_constraint_jacobian passes the GLOBAL
#     `nlp` to jac_c! (src/moi.jl:22) and jac_c! reads the GLOBAL `lb` (src/constraints.jl:270,272) -- quirk Q4: both are
#     sidestepped, the veneer's callbacks use their argument's handle, whose model carries lb.
# Requires `using Ipopt` in the session, as the reference's notebook has (MathOptInterface 0.9: MOI.SingleVariable).
function solve(x0, prob::HybridNLPHIP; tol=1.0e-6, c_tol=1.0e-6, max_iter=2000)
    n_nlp = num_primals(prob)
    length(x0) == n_nlp || error("solve: x0 has $(length(x0)) entries, the problem has $n_nlp variables")
    x_l = Vector{Float64}(undef, n_nlp); x_u = Vector{Float64}(undef, n_nlp)
    qln_check(ccall((:qln_variable_bounds, LIBQLN), Cint, (Int32, Ptr{QlnSolveOptions}, Ptr{Cdouble}, Ptr{Cdouble}),
                    prob.N, C_NULL, x_l, x_u))
    block = MOI.NLPBlockData(MOI.NLPBoundsPair.(prob.lb, prob.ub), prob, true)       # true: has an objective
    solver = Ipopt.Optimizer()
    for (name, value) in ("max_iter" => max_iter, "tol" => tol, "constr_viol_tol" => c_tol)
        solver.options[name] = value
    end
    x = MOI.add_variables(solver, n_nlp)
    for (xi, lo, hi, start) in zip(x, x_l, x_u, x0)
        v = MOI.SingleVariable(xi)
        MOI.add_constraint(solver, v, MOI.LessThan(hi))        # +-Inf bounds included, as the reference adds them
        MOI.add_constraint(solver, v, MOI.GreaterThan(lo))
        MOI.set(solver, MOI.VariablePrimalStart(), xi, start)
    end
    MOI.set(solver, MOI.NLPBlock(), block)
    MOI.set(solver, MOI.ObjectiveSense(), MOI.MIN_SENSE)
    MOI.optimize!(solver)
    return MOI.get(solver, MOI.VariablePrimal(), x), solver
end
