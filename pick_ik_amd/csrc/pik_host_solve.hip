// pik_host_solve.hip -- the host solver for queries with a host cost function (pik_host_solve.hpp), compiled with
// the flags of an EXACT flavour: -DPIK_STRICT -DPIK_EXACT_FMA for the product library (symbol
// pik_exact_host_solve: the arithmetic of its exact kernels, the oracle's math mode "fma"), -DPIK_STRICT for the
// verification library (pik_strict_host_solve: plain IEEE, math mode "portable").  No kernels in here.
#if !defined(PIK_STRICT)
#error "compile with the flags of an exact flavour"
#endif
#include "pik_host_solve.hpp"

#if PIK_XF
#define PIK_HOST_SOLVE pik_exact_host_solve
#else
#define PIK_HOST_SOLVE pik_strict_host_solve
#endif

extern "C" int PIK_HOST_SOLVE(const pikamd_solver* s, const pikamd_params* p, long long B, const double* goal,
                              const double* seed, const double* guess, unsigned long long rng_seed,
                              long long problem_offset, pikamd_cost_fn cb, void* user, double* solution, int32_t* status,
                              double* final_cost, pikamd_stats* stats) {
    pik::ParamsK pk;
    if (const char* msg = pik::make_params_k(p, pk)) return pik::fail(PIKAMD_EINVAL, "%s", msg);
    switch (s->chain.dof) {
#define PIK_HOST_CASE(N)                                                                                              \
    case N:                                                                                                           \
        return pik::host_solve_batch<N>(s, p, pk, B, goal, seed, guess, rng_seed, problem_offset, cb, user, solution, \
                                        status, final_cost, stats);
        PIK_HOST_CASE(1) PIK_HOST_CASE(2) PIK_HOST_CASE(3) PIK_HOST_CASE(4) PIK_HOST_CASE(5) PIK_HOST_CASE(6)
        PIK_HOST_CASE(7) PIK_HOST_CASE(8) PIK_HOST_CASE(9) PIK_HOST_CASE(10) PIK_HOST_CASE(11) PIK_HOST_CASE(12)
        PIK_HOST_CASE(13) PIK_HOST_CASE(14) PIK_HOST_CASE(15) PIK_HOST_CASE(16)
#undef PIK_HOST_CASE
        default: return pik::fail(PIKAMD_EUNSUPPORTED, "dof %d: 1..16", s->chain.dof);
    }
}
