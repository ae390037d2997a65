// K1/K2 placeholder while the greedy clustering kernels are being written.
#include "pgx_internal.h"

extern "C" int pgx_cluster_greedy(pgx_ctx *ctx, const uint8_t *residues, const uint64_t *offsets,
                                  uint32_t n, const pgx_cluster_params *params, int32_t *out_cluster,
                                  int32_t *out_member, float *out_identity, uint8_t *out_strand,
                                  uint32_t *out_n_clusters, pgx_cluster_stats *stats) {
    pgx_set_error("pgx_cluster_greedy: not implemented yet");
    return PGX_ERR_INTERNAL;
}
