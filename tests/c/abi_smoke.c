/* C-level consumer of libemx.so: proves the boundary is a plain C ABI (no Python, no torch types).
 * Built and run by tests/test_c_abi.py:  gcc abi_smoke.c -I include -ldl  (the library is dlopen'ed so
 * the same binary serves the CPU box -- host-only entry points -- and the GPU box -- a short run).
 * Usage: abi_smoke <path/to/libemx.so> [gpu]                                                      */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emx.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    SYM(emx_version) SYM(emx_device_count) SYM(emx_last_error)
    SYM(emx_mt_create) SYM(emx_mt_random_sample) SYM(emx_mt_destroy) SYM(emx_host_plan_philox)
    SYM(emx_create) SYM(emx_destroy) SYM(emx_set_target) SYM(emx_set_moves) SYM(emx_set_rng_mode)
    SYM(emx_rng_set_philox) SYM(emx_set_state) SYM(emx_eval_state_log_prob) SYM(emx_chain_config)
    SYM(emx_run) SYM(emx_chain_read) SYM(emx_accepted_counts) SYM(emx_get_state) SYM(emx_status)
    SYM(emx_autocorr) SYM(emx_walkers_independent)
    SYM(emx_snapshot_save) SYM(emx_snapshot_read) SYM(emx_snapshot_restore) SYM(emx_snapshot_free) SYM(emx_comm_count) SYM(emx_pipeline_stats) SYM(emx_persist_info) SYM(emx_host_persist_shape)
    printf("version: %s\n", p_emx_version());
    int32_t pw = -1, pg = -1;
    if (p_emx_host_persist_shape(65536, 2, 256, &pw, &pg) != 0 || pw != 8 || pg != 256) return 17;      /* host only */

    /* host-only: MT19937 seeded with init_genrand(5489)-style key is not needed; use a fixed key */
    uint32_t key[624];
    for (int i = 0; i < 624; ++i) key[i] = 1812433253u * (uint32_t)(i + 1) + 12345u;
    emx_mt* m = p_emx_mt_create(key, 624, 0, 0.0);
    double u[4];
    p_emx_mt_random_sample(m, 4, u);
    p_emx_mt_destroy(m);
    for (int i = 0; i < 4; ++i)
        if (!(u[i] >= 0.0 && u[i] < 1.0)) { fprintf(stderr, "bad uniform\n"); return 3; }

    /* host twin of the native plan: a permutation of the walkers with balanced halves */
    enum { N = 64, D = 4 };
    emx_move_desc mv = {EMX_MOVE_STRETCH, 2, 1, 0, 2.0, 1e-5, 0.5, 1.7};
    int32_t off[3], order[N], p0[N], p1[N], p2[N];
    double s0[N], ua[N];
    if (p_emx_host_plan_philox(7, 0, N, &mv, off, order, p0, p1, p2, s0, ua) != 0) return 4;
    int seen[N] = {0};
    for (int i = 0; i < N; ++i) seen[order[i]]++;
    for (int i = 0; i < N; ++i)
        if (seen[i] != 1) { fprintf(stderr, "plan order is not a permutation\n"); return 4; }
    if (off[0] != 0 || off[1] != N / 2 || off[2] != N) return 4;
    printf("host entry points ok\n");

    int32_t ndev = 0;
    p_emx_device_count(&ndev);
    if (argc < 3 || strcmp(argv[2], "gpu") != 0) {
        emx_ctx* ctx = NULL;
        if (ndev == 0 && p_emx_create(0, N, D, &ctx) == 0) { fprintf(stderr, "emx_create succeeded without a GPU\n"); return 5; }
        if (ndev == 0) printf("no GPU: emx_create refused (%s)\n", p_emx_last_error(NULL));
        return 0;
    }

    /* GPU: 200 stretch steps on an isotropic Gaussian, chain stored on the device */
    emx_ctx* ctx = NULL;
    if (p_emx_create(0, N, D, &ctx) != 0) { fprintf(stderr, "emx_create: %s\n", p_emx_last_error(NULL)); return 6; }
    double cdf = 1.0, x0[N * D];
    for (int i = 0; i < N * D; ++i) x0[i] = sin(0.37 * i) + 0.01 * i / (N * D);
    int rc = 0;
    rc |= p_emx_set_target(ctx, EMX_TARGET_ISO_GAUSS, NULL, NULL, 0.0);
    rc |= p_emx_set_moves(ctx, 1, &mv, &cdf);
    rc |= p_emx_set_rng_mode(ctx, EMX_RNG_PHILOX);
    rc |= p_emx_rng_set_philox(ctx, 2024, 0);
    rc |= p_emx_set_state(ctx, x0, NULL);
    rc |= p_emx_eval_state_log_prob(ctx);
    rc |= p_emx_chain_config(ctx, 200);
    rc |= p_emx_run(ctx, 200, 1, 1);
    if (rc) { fprintf(stderr, "run failed: %s\n", p_emx_last_error(ctx)); return 7; }
    static double chain[200 * N * D], lp[N], acc[N], xf[N * D];
    rc |= p_emx_chain_read(ctx, 0, 0, 200, 1, chain);
    rc |= p_emx_accepted_counts(ctx, acc);
    rc |= p_emx_get_state(ctx, xf, lp);
    uint32_t bits = 1;
    rc |= p_emx_status(ctx, &bits);
    if (rc || bits) { fprintf(stderr, "readback failed\n"); return 8; }
    if (memcmp(chain + 199 * N * D, xf, sizeof(xf)) != 0) { fprintf(stderr, "last chain row != state\n"); return 9; }
    double a = 0, m2 = 0;
    for (int i = 0; i < N; ++i) a += acc[i] / 200.0 / N;
    for (int t = 100; t < 200; ++t)
        for (int i = 0; i < N * D; ++i) m2 += chain[t * N * D + i] * chain[t * N * D + i];
    m2 /= 100.0 * N * D;
    for (int i = 0; i < N; ++i) {                      /* stored log-prob == -0.5 |x|^2 */
        double q = 0;
        for (int d = 0; d < D; ++d) q += xf[i * D + d] * xf[i * D + d];
        if (fabs(lp[i] + 0.5 * q) > 1e-12 * (1 + q)) { fprintf(stderr, "log-prob mismatch\n"); return 10; }
    }
    printf("gpu run ok: acceptance %.3f, <x^2> %.3f\n", a, m2);
    if (!(a > 0.2 && a < 0.9 && m2 > 0.6 && m2 < 1.5)) return 11;
    /* the two operations around the loop, from C: autocorrelation time of the device chain and the initial-state check */
    double tau[D];
    int32_t win[D];
    int64_t nsel = 0;
    if (p_emx_autocorr(ctx, 20, 1, 5.0, tau, win, &nsel) != 0) { fprintf(stderr, "emx_autocorr: %s\n", p_emx_last_error(ctx)); return 12; }
    printf("autocorrelation time over %lld samples: %.2f %.2f %.2f %.2f steps\n", (long long)nsel, tau[0], tau[1], tau[2], tau[3]);
    for (int d = 0; d < D; ++d)
        if (!(nsel == 180 && tau[d] > 1.0 && tau[d] < 60.0 && win[d] > 0)) return 12;
    int32_t indep = -1;
    double cond = 0.0;
    uint32_t lcg = 12345u;                                 /* the sinusoids above span two dimensions only: fresh pseudo-random rows */
    for (int i = 0; i < N * D; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        x0[i] = (double)(lcg >> 8) / 16777216.0 - 0.5;
    }
    if (p_emx_walkers_independent(0, x0, N, D, &indep, &cond) != 0 || indep != 1 || !(cond >= 1.0 && cond < 1e3)) {
        fprintf(stderr, "emx_walkers_independent: verdict %d, cond %g (%s)\n", indep, cond, p_emx_last_error(NULL));
        return 13;
    }
    for (int i = 0; i < N; ++i) x0[i * D + 3] = 2.0 * x0[i * D + 1];          /* a linearly dependent coordinate */
    if (p_emx_walkers_independent(0, x0, N, D, &indep, &cond) != 0 || indep != 0) { fprintf(stderr, "dependent walkers accepted\n"); return 13; }
    printf("initial-state check ok\n");
    /* a State handed out earlier stays on the device: snapshot, move on, read it back, make it current again (ensemble.py:441-447) */
    static double xs[N * D], ls[N], xr[N * D], lr[N];
    rc = p_emx_snapshot_save(ctx, 3);
    rc |= p_emx_run(ctx, 20, 1, 0);
    rc |= p_emx_snapshot_read(ctx, 3, xs, ls);
    if (rc || memcmp(xs, xf, sizeof(xf)) != 0 || memcmp(ls, lp, sizeof(lp)) != 0) { fprintf(stderr, "snapshot differs from the state it was taken of\n"); return 14; }
    rc |= p_emx_get_state(ctx, xr, lr);
    if (rc || memcmp(xr, xf, sizeof(xf)) == 0) { fprintf(stderr, "the ensemble did not move on\n"); return 14; }
    rc |= p_emx_snapshot_restore(ctx, 3);
    rc |= p_emx_get_state(ctx, xr, lr);
    rc |= p_emx_snapshot_free(ctx, 3);
    if (rc || memcmp(xr, xf, sizeof(xf)) != 0 || memcmp(lr, lp, sizeof(lp)) != 0) { fprintf(stderr, "snapshot restore failed\n"); return 14; }
    int32_t ranks = -1, fin = -1;
    int64_t produced = -1;
    double stage[6];
    if (p_emx_comm_count(ctx, &ranks) != 0 || ranks != 0 || p_emx_pipeline_stats(ctx, stage, &produced, &fin) != 0 || produced != 0) return 15;
    int64_t pinfo[4] = {-1, -1, -1, -1};
    if (p_emx_persist_info(ctx, pinfo) != 0 || pinfo[0] != 0 || pinfo[1] != 0 || pinfo[2] != 0) return 16;      /* a tiny ensemble never qualifies */
    printf("snapshots ok\n");
    p_emx_destroy(ctx);
    return 0;
}
