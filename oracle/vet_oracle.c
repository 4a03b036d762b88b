/*
 * vet_oracle.c — plain-C restatement of the reference's per-frame path (CPU, scalar, FP64).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (checked against the golden vectors generated from
 * the real reference) and by bench.py's cpu_baseline leg ("port").  The product never links or
 * calls it.  Parity status: PINNED by tests/test_oracle_c.py against tests/golden/.
 *
 * It follows the reference operation by operation (paths relative to
 * /root/reference/src/viewport_entropy_toolkit/):
 *   angle()            utilities/entropy_utils.py:41-67   normalise both, dot, clip, arccos
 *   spatial frame      utilities/entropy_utils.py:108-144, 147-211
 *   transition frame   utilities/entropy_utils.py:213-332 (literal bucket walk, incl. the
 *                      int-key first bucket and the stale transition_weight)
 *   series drivers     analyzers/spatial_entropy.py:107-164, analyzers/transition_entropy.py:107-175
 * One deviation that does not change any value: the reference sweeps the tile distances twice
 * per sample (calculate_tile_weights, then find_nearest_tile); here one sweep feeds both.
 * The pixel -> Vector grid (trigonometry + decimal rounding) is supplied by the Python oracle.
 *
 * Build:  gcc -O2 -fopenmp -fPIC -shared -o _build/libvet_oracle.so vet_oracle.c -lm   (no -ffast-math)
 *         Threads follow OMP_NUM_THREADS (bench.py times 1 thread and the box's core share).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double angle(const double *a, const double *b) {
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double c = (a[0] / na) * (b[0] / nb) + (a[1] / na) * (b[1] / nb) + (a[2] / na) * (b[2] / nb);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    return acos(c);
}

static double max_entropy(double count) {
    const double p = 1.0 / count;
    return -count * p * log2(p);
}

/* sample -> direction id on the (H+1)x(W+1) grid, -1 absent, -2 out of range */
static long dir_id(double m, double v, int W, int H) {
    if (m != m || v != v) return -1;
    if (!(m >= 0.0 && m <= 1.0 && v >= 0.0 && v <= 1.0)) return -2;
    return (long)((int)(v * H)) * (W + 1) + (int)(m * W);
}

/* returns 0, or -3 on an out-of-range sample, -4 on a frame without users */
int oracle_spatial(const double *mu, const double *mv, int T, int U, int W, int H, const double *grid,
                   int K, const int *n_tiles, const double *const *tiles, double fov_deg, double power,
                   int weighted, double *out_entropy, int32_t *out_assign, double *out_weights0) {
    const double max_ang = (fov_deg / 2.0) * (M_PI / 180.0);
    int nmax = 0;
    for (int k = 0; k < K; ++k) if (n_tiles[k] > nmax) nmax = n_tiles[k];
    int rc = 0;
    /* frames are independent: with -fopenmp (and OMP_NUM_THREADS > 1) they are spread over the host
     * cores, each thread with its own histogram; without OpenMP this is the plain serial loop */
#pragma omp parallel
    {
        double *hist = (double *)malloc(sizeof(double) * nmax);
        char *touched = (char *)malloc(nmax);
#pragma omp for schedule(static)
        for (int t = 0; t < T; ++t) {
            int frc = 0;
            double total_entropy = 0.0;
            for (int k = 0; k < K && !frc; ++k) {
                const int n = n_tiles[k];
                const double *tl = tiles[k];
                memset(hist, 0, sizeof(double) * n);
                memset(touched, 0, n);
                double total_weight = 0.0;
                int present = 0;
                for (int u = 0; u < U; ++u) {
                    const long id = dir_id(mu[(long)t * U + u], mv[(long)t * U + u], W, H);
                    if (id == -2) { frc = -3; break; }
                    if (id < 0) { if (k == 0 && out_assign) out_assign[(long)t * U + u] = -1; continue; }
                    ++present;
                    const double *d = grid + 3 * id;
                    double best = 1e300;
                    int bi = 0;
                    for (int j = 0; j < n; ++j) {
                        const double a = angle(d, tl + 3 * j);
                        if (a < best) { best = a; bi = j; }
                        if (weighted && a < max_ang) {
                            const double w = pow((max_ang - a) / max_ang, power);
                            hist[j] += w;
                            touched[j] = 1;
                            total_weight += w;
                        }
                    }
                    if (!weighted) { hist[bi] += 1.0; touched[bi] = 1; total_weight += 1.0; }
                    if (k == 0 && out_assign) out_assign[(long)t * U + u] = bi;
                }
                if (frc) break;
                if (!present) { frc = -4; break; }
                double ent = 0.0;
                for (int j = 0; j < n; ++j)
                    if (touched[j]) {
                        const double p = hist[j] / total_weight;
                        ent -= p * log2(p);
                    }
                const double mx = (weighted || total_weight > n) ? max_entropy((double)n) : max_entropy(total_weight);
                total_entropy += ent / mx;
                /* dense convention of the C-ABI: a dict key with the value 0.0 carries -0.0, no key +0.0 */
                if (k == 0 && out_weights0)
                    for (int j = 0; j < n; ++j)
                        out_weights0[(long)t * n + j] = (touched[j] && hist[j] == 0.0) ? -0.0 : hist[j];
            }
            out_entropy[t] = total_entropy / K;
            if (frc) {
#pragma omp critical
                if (!rc || frc > rc) rc = frc;
            }
        }
        free(hist);
        free(touched);
    }
    return rc;
}

/* One frame pair, literal walk of entropy_utils.py:258-330 over the common users in order.
 * Buckets of a source tile: slot 0 = the int-keyed bucket of its first user, then Vector-keyed
 * buckets in insertion order. */
static double transition_frame(const int *p, const int *c, int N, int n) {
    /* weight_per_tile in insertion order */
    int *src = (int *)malloc(sizeof(int) * (N + 1)), nsrc = 0;
    int *cnt = (int *)calloc(n, sizeof(int));
    int *seen = (int *)calloc(n, sizeof(int));
    /* buckets: for each source tile a list of (dest, weight); flat arrays sized N */
    int *b_src = (int *)malloc(sizeof(int) * (N + 1));
    int *b_dst = (int *)malloc(sizeof(int) * (N + 1));
    int *b_w = (int *)malloc(sizeof(int) * (N + 1));
    char *b_first = (char *)malloc(N + 1);
    int nb = 0;
    for (int i = 0; i < N; ++i) {
        if (!seen[p[i]]) {
            seen[p[i]] = 1;
            src[nsrc++] = p[i];
            b_src[nb] = p[i]; b_dst[nb] = c[i]; b_w[nb] = 1; b_first[nb] = 1; ++nb;
        } else {
            int f = -1;
            for (int b = 0; b < nb; ++b)
                if (b_src[b] == p[i] && !b_first[b] && b_dst[b] == c[i]) { f = b; break; }
            if (f < 0) { b_src[nb] = p[i]; b_dst[nb] = c[i]; b_w[nb] = 1; b_first[nb] = 0; ++nb; }
            else b_w[f] += 1;
        }
        cnt[p[i]] += 1;
    }
    double ent = 0.0;
    for (int s = 0; s < nsrc; ++s) {
        const int tile = src[s];
        const double prop = (double)cnt[tile] / (double)N;
        int tsum = 0, last = 0, nbk = 0;
        for (int b = 0; b < nb; ++b)
            if (b_src[b] == tile) { last = b_w[b]; tsum += last; ++nbk; }
        double cell = 0.0;
        for (int b = 0; b < nbk; ++b) {
            const double q = (double)last / (double)tsum;
            cell += q * log2(q);
        }
        ent += -prop * cell;
    }
    double mx;
    if (N > n) { const double tp = 1.0 / n; mx = n * -tp * log2(tp); }
    else { const double tp = 1.0 / N; mx = N * -tp * log2(tp); }
    free(src); free(cnt); free(seen); free(b_src); free(b_dst); free(b_w); free(b_first);
    return ent / mx;
}

/* nearest tile of every grid direction actually used is computed on the fly (no LUT).
 * With -fopenmp the samples of the nearest-tile sweep and then the rows are spread over the host cores
 * (every row is the same literal walk, with its own p / c lists); one thread = the plain serial loops.
 * Error precedence as in the serial walk: an out-of-range sample (-3) is met in the first lattice's
 * sweep, before any row is looked at; otherwise a row without a common user gives -4. */
int oracle_transition(const double *mu, const double *mv, int T, int U, int W, int H, const double *grid,
                      int K, const int *n_tiles, const double *const *tiles, double *out_entropy,
                      int32_t *out_pairs) {
    int *tile = (int *)malloc(sizeof(int) * (size_t)T * U);
    int rc = 0;
    for (int r = 0; r + 1 < T; ++r) out_entropy[r] = 0.0;
    for (int k = 0; k < K && !rc; ++k) {
        const int n = n_tiles[k];
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (long i = 0; i < (long)T * U; ++i) {
            const long id = dir_id(mu[i], mv[i], W, H);
            if (id == -2) { bad |= 1; tile[i] = -1; continue; }
            if (id < 0) { tile[i] = -1; continue; }
            double best = 1e300;
            int bi = 0;
            for (int j = 0; j < n; ++j) {
                const double a = angle(grid + 3 * id, tiles[k] + 3 * j);
                if (a < best) { best = a; bi = j; }
            }
            tile[i] = bi;
        }
        if (bad) { rc = -3; break; }
        int empty = 0;
#pragma omp parallel reduction(| : empty)
        {
            int *p = (int *)malloc(sizeof(int) * U), *c = (int *)malloc(sizeof(int) * U);
#pragma omp for schedule(static)
            for (int r = 0; r < T - 1; ++r) {
                int N = 0;
                for (int u = 0; u < U; ++u) {
                    const int a = tile[(long)r * U + u], b = tile[(long)(r + 1) * U + u];
                    if (k == 0 && out_pairs) {
                        out_pairs[((long)r * U + u) * 2] = (a >= 0 && b >= 0) ? a : -1;
                        out_pairs[((long)r * U + u) * 2 + 1] = (a >= 0 && b >= 0) ? b : -1;
                    }
                    if (a >= 0 && b >= 0) { p[N] = a; c[N] = b; ++N; }
                }
                if (N == 0) { empty |= 1; continue; }
                out_entropy[r] += transition_frame(p, c, N, n);
            }
            free(p); free(c);
        }
        if (empty) rc = -4;
    }
    for (int r = 0; r + 1 < T; ++r) out_entropy[r] /= K;
    free(tile);
    return rc;
}
