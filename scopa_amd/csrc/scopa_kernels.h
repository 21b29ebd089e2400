// scopa_kernels.h -- prototypes of kernels that more than one translation unit launches (the multi-deal mode reuses
// the single-deal kernels with one workgroup per deal).
#pragma once
#include "scopa_ctx.h"

__global__ void k_tree_build(const uint8_t *__restrict__ perm16, scopa_state *__restrict__ states, uint16_t *__restrict__ infoset_of,
                             int8_t *__restrict__ payoff, uint64_t *__restrict__ key_of_infoset, int32_t *__restrict__ meta);
__global__ void k_tables_reset(double *regret, double *strat, double *local, const uint64_t *key_of_infoset, const int32_t *meta);
__global__ void k_cfr_exact(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, double *__restrict__ g_regret,
                            double *__restrict__ g_strat, double *__restrict__ g_local, int n_infosets, int n_traversals,
                            int first_traverser, double *__restrict__ root_values, unsigned long long *__restrict__ g_counters,
                            int use_lds, uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta, int start_depth, int start_idx,
                            double start_r0, double start_r1);
__global__ void k_cfr_sync(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff, const uint64_t *__restrict__ g_key,
                           double *__restrict__ g_regret, double *__restrict__ g_strat, int n_infosets, int n_iters,
                           unsigned long long *__restrict__ g_counters, uint32_t *__restrict__ g_visit, int32_t *__restrict__ g_meta);
__global__ void k_exploitability(const uint16_t *__restrict__ g_infoset, const int8_t *__restrict__ g_payoff,
                                 const uint64_t *__restrict__ g_key, const double *__restrict__ g_strat,
                                 const double *__restrict__ g_policy_in, int n_infosets, double *__restrict__ out4,
                                 double *__restrict__ g_policy_out, const int32_t *__restrict__ multi_meta);

namespace scopa {
int32_t launch_mccfr_multi(scopa_ctx *ctx, int n_deals, int max_infosets, const uint16_t *d_infoset, const int8_t *d_payoff,
                           const uint64_t *d_key, double *d_regret, double *d_strat, const int32_t *d_meta, uint32_t *d_visit,
                           unsigned long long *d_counters, uint64_t seed, uint32_t iter0, uint32_t n_iters, uint32_t batch);
}
