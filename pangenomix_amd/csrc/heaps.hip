// Heaps-law fits of the pan-genome curves on gfx950.
//
// Replaces the loop of pangenome_analysis.py:24-48: for every iteration (row of the pan table that
// estimate_pan_core_size produced) fit  pan(j) = kappa * j^alpha,  j = 1..S, by non-linear least squares
// from the start point the reference uses (alpha = 0.5, kappa = min of the row). The reference calls
// scipy.optimize.curve_fit (MINPACK's Levenberg-Marquardt); here one wave per iteration runs
// Levenberg-Marquardt on the 2 x 2 normal equations in double precision. Both converge to the same
// least-squares minimum; floating point, so parity is to a tolerance: rtol 1e-5 on alpha and kappa (scipy stops at
// its default ftol = xtol = 1e-8, within ~3e-6 of the minimum; this kernel iterates to the minimum itself).
// The table can be the int32 device output of pgx_pan_core_dev directly (no host round trip).
#include "pgx_internal.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void heaps_fit_kernel(const T *__restrict__ pan, uint32_t n_iter, uint32_t S,
                                                       double *__restrict__ alpha, double *__restrict__ kappa,
                                                       int32_t *__restrict__ steps) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= n_iter) return;
    const T *y = pan + (size_t)it * S;
    double ymin = 1e300;
    for (uint32_t j = lane; j < S; j += 64) ymin = fmin(ymin, (double)y[j]);
    for (int d = 32; d > 0; d >>= 1) ymin = fmin(ymin, __shfl_xor(ymin, d));
    double a = 0.5, k = ymin;                      // the reference's p0 (pangenome_analysis.py:44)
    auto cost_of = [&](double aa, double kk) {
        double c = 0.0;
        for (uint32_t j = lane; j < S; j += 64) { const double r = kk * pow((double)(j + 1), aa) - (double)y[j]; c += r * r; }
        return wave_sum(c);
    };
    double cost = cost_of(a, k), lambda = 1e-3;
    int n = 0;
    for (; n < 200; ++n) {
        double saa = 0, sak = 0, skk = 0, ga = 0, gk = 0;
        for (uint32_t j = lane; j < S; j += 64) {
            const double x = (double)(j + 1), p = pow(x, a), r = k * p - (double)y[j];
            const double da = k * p * log(x), dk = p;
            saa += da * da; sak += da * dk; skk += dk * dk; ga += da * r; gk += dk * r;
        }
        saa = wave_sum(saa); sak = wave_sum(sak); skk = wave_sum(skk); ga = wave_sum(ga); gk = wave_sum(gk);
        bool done = false;
        for (int tries = 0; tries < 40; ++tries) {
            const double m00 = saa * (1.0 + lambda), m11 = skk * (1.0 + lambda), det = m00 * m11 - sak * sak;
            if (!(fabs(det) > 0.0)) { lambda *= 10.0; continue; }
            const double d_a = -(m11 * ga - sak * gk) / det, d_k = -(m00 * gk - sak * ga) / det;
            const double c2 = cost_of(a + d_a, k + d_k);
            if (c2 <= cost) {
                const bool small = fabs(d_a) <= 1e-14 * (fabs(a) + 1e-14) && fabs(d_k) <= 1e-14 * (fabs(k) + 1e-14);
                const bool flat = cost - c2 <= 1e-16 * cost;
                a += d_a; k += d_k; cost = c2;
                lambda = fmax(lambda * 0.1, 1e-12);
                done = small || flat;
                break;
            }
            lambda *= 10.0;
            if (tries == 39) done = true;          // no downhill step left at any damping: converged
        }
        if (done) break;
    }
    if (lane == 0) { alpha[it] = a; kappa[it] = k; if (steps) steps[it] = n; }
}

enum { HP_SLOT_TAB = 90, HP_SLOT_A, HP_SLOT_K };

}  // namespace

extern "C" {

int pgx_heaps_fit_dev(pgx_ctx *ctx, const int32_t *d_pan, uint32_t n_iter, uint32_t n_genomes, double *d_alpha,
                      double *d_kappa, void *stream_) {
    PGX_REQUIRE(ctx, "NULL context");
    if (n_iter == 0) return PGX_OK;
    PGX_REQUIRE(d_pan && d_alpha && d_kappa && n_genomes > 0, "NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    {
        ProfScope prof(ctx, "heaps_fit_kernel", stream);
        heaps_fit_kernel<int32_t><<<(n_iter + 3) / 4, 256, 0, stream>>>(d_pan, n_iter, n_genomes, d_alpha, d_kappa, nullptr);
    }
    PGX_HIP(hipGetLastError());
    return PGX_OK;
}

int pgx_heaps_fit(pgx_ctx *ctx, const double *pan, uint32_t n_iter, uint32_t n_genomes, double *out_alpha,
                  double *out_kappa) {
    PGX_REQUIRE(ctx, "NULL context");
    if (n_iter == 0) return PGX_OK;
    PGX_REQUIRE(pan && out_alpha && out_kappa && n_genomes > 0, "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    DevBuf d_tab, d_a, d_k;
    d_tab.ctx = d_a.ctx = d_k.ctx = ctx;
    d_tab.slot = HP_SLOT_TAB; d_a.slot = HP_SLOT_A; d_k.slot = HP_SLOT_K;
    const size_t nb = (size_t)n_iter * n_genomes * 8;
    PGX_HIP(d_tab.alloc(nb));
    PGX_HIP(d_a.alloc((size_t)n_iter * 8));
    PGX_HIP(d_k.alloc((size_t)n_iter * 8));
    PGX_HIP(hipMemcpyAsync(d_tab.p, pan, nb, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope prof(ctx, "heaps_fit_kernel", ctx->stream);
        heaps_fit_kernel<double><<<(n_iter + 3) / 4, 256, 0, ctx->stream>>>(d_tab.as<double>(), n_iter, n_genomes,
                                                                            d_a.as<double>(), d_k.as<double>(), nullptr);
    }
    PGX_HIP(hipGetLastError());
    PGX_HIP(hipMemcpyAsync(out_alpha, d_a.p, (size_t)n_iter * 8, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipMemcpyAsync(out_kappa, d_k.p, (size_t)n_iter * 8, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

}  // extern "C"
