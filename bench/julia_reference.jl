# julia_reference.jl -- times the reference ALGORITHM (constraint + ForwardDiff Jacobian per knot point) in plain
# Julia, for the CPU-baseline row of SURVEY.md 8d.  It includes nothing from /root/reference: the dynamics below are
# this build's own compact restatement (one function, mode flags).  NOT EXECUTED in this pipeline (no Julia on
# either box); bench/run_julia_reference.sh prints "SKIPPED: julia not found" there.
#   julia bench/julia_reference.jl [problems=256] [N=40] [k_trans=14]
using ForwardDiff, Random

const G, MB, MF, LB = -9.81, 10.0, 0.1, 0.5
const IB = MB * LB^2 / 12

# s: 14 states, u: 4 forces + h; free1/free2: which foot is not pinned (mode 1 -> free2, mode 2 -> free1)
function sdot(s, u, free1::Bool, free2::Bool)
    tau = -u[1] * (s[5] - s[2]) + u[2] * (s[4] - s[1]) - u[3] * (s[7] - s[2]) + u[4] * (s[6] - s[1])
    z = zero(eltype(s)) * zero(eltype(u))
    [s[8], s[9], s[10],
     free1 ? s[11] : z, free1 ? s[12] : z, free2 ? s[13] : z, free2 ? s[14] : z,
     (u[1] + u[3]) / MB, (u[2] + u[4]) / MB + G, tau / IB,
     free1 ? -u[1] / MF : z, free1 ? -u[2] / MF + G : z, free2 ? -u[3] / MF : z, free2 ? -u[4] / MF + G : z]
end

function step(z, free1, free2)
    s, u, h = z[1:14], z[16:20], z[20]
    k1 = sdot(s, u, free1, free2)
    k2 = sdot(s + 0.5 * h * k1, u, free1, free2)
    k3 = sdot(s + 0.5 * h * k2, u, free1, free2)
    k4 = sdot(s + h * k3, u, free1, free2)
    [s + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4); z[15] + h]
end

function eval_problem!(c, J, Z, N, k_trans)
    for k in 1:N-1
        z = Z[20*(k-1)+1:20*k]
        free2 = k <= k_trans - 1            # init_mode 1: foot 2 free until the transition
        f = w -> step(w, false, free2)
        c[:, k] = f(z) - Z[20*k+1:20*k+15]
        J[:, :, k] = ForwardDiff.jacobian(f, z)
    end
end

function main()
    B = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 256
    N = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 40
    kt = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 14
    Random.seed!(0)
    Z = randn(20N - 5); Z[20:20:end] .= 0.01
    c = zeros(15, N - 1); J = zeros(15, 20, N - 1)
    eval_problem!(c, J, Z, N, kt)            # compile
    t = @elapsed for _ in 1:B
        eval_problem!(c, J, Z, N, kt)
    end
    println("{\"kind\": \"julia\", \"knot_evals_per_s\": ", B * N / t, ", \"threads\": 1, \"problems\": ", B, ", \"N\": ", N, "}")
end
main()
