/* sanitize_driver.c -- runs the oracle once over a small random batch under -fsanitize=address,undefined
 * (tests/test_oracle_sanitizers.py).  TEST INFRASTRUCTURE ONLY. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qln_oracle.h"

static double rnd(void) { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

int main(void) {
    const int N = 23;
    srand(7);
    for (int kt = 1; kt <= N + 1; kt += 3) {
        for (int im = 1; im <= 2; ++im) {
            orc_problem p;
            p.N = N;
            p.k_trans = kt;
            p.init_mode = im;
            orc_default_model(&p.model);
            double* cost = malloc(sizeof(double) * N * ORC_COST_STRIDE);
            for (int i = 0; i < N * ORC_COST_STRIDE; ++i) cost[i] = rnd();
            p.cost = cost;
            for (int i = 0; i < 15; ++i) {
                p.x0[i] = rnd();
                p.xf[i] = rnd();
            }
            const int n = orc_num_primals(N), m = orc_num_duals(N, kt), nnz = orc_jac_nnz(N, kt);
            double* Z = malloc(sizeof(double) * n);
            for (int i = 0; i < n; ++i) Z[i] = rnd();
            for (int k = 0; k < N - 1; ++k) Z[20 * k + 19] = 0.001 + 0.019 * fabs(rnd());
            double* c = malloc(sizeof(double) * m);
            double* g = malloc(sizeof(double) * n);
            double* v = malloc(sizeof(double) * nnz);
            int32_t* rows = malloc(sizeof(int32_t) * nnz);
            int32_t* cols = malloc(sizeof(int32_t) * nnz);
            double* dense = malloc(sizeof(double) * (size_t)m * n);
            double* lb = malloc(sizeof(double) * m);
            double* ub = malloc(sizeof(double) * m);
            orc_eval_c(&p, c, Z);
            orc_grad_f(&p, g, Z);
            const double f = orc_eval_f(&p, Z);
            orc_jac_c_coo(&p, v, Z);
            orc_jac_structure(&p, rows, cols);
            orc_jac_c_dense(&p, dense, Z);
            orc_constraint_bounds(N, kt, lb, ub);
            for (int e = 0; e < nnz; ++e)
                if (rows[e] < 0 || rows[e] >= m || cols[e] < 0 || cols[e] >= n) return 2;
            if (!(f == f)) return 3;
            free(cost); free(Z); free(c); free(g); free(v); free(rows); free(cols); free(dense); free(lb); free(ub);
        }
    }
    puts("sanitize_driver ok");
    return 0;
}
