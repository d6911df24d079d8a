/*
 * nf_oracle.c — CPU restatement (TEST INFRASTRUCTURE, never on the product path) of the two
 * third-party neighbour ops the NeuroFluid hot path calls.  Plain C, compiled without FP
 * contraction (-ffp-contract=off) so that the fp32 "d2 < r2" test is a mul + add chain in
 * d = 0,1,2 order.
 *
 *  - nfo_ball_query_firstk : pytorch3d v0.6.1 ops.ball_query semantics, call site
 *        /root/reference/models/renderer.py:116-118  (SURVEY §8c "third-party arithmetic #1").
 *        For each query scan p2 in index order, accept when sum_d (p1_d - p2_d)^2 <  r*r,
 *        store index / squared distance / xyz in the next free slot, stop at K.
 *        Padding: idx = -1, dist2 = 0, nn = 0.
 *  - nfo_radius_count / nfo_radius_csr : Open3D 0.15.2 FixedRadiusSearch semantics (L2 metric,
 *        d2 <= r*r, neighbours at the *identical position* skipped when ignore_query_point),
 *        call sites /root/reference/models/transmodel.py:86-95,116-118,125,135-138
 *        (SURVEY §8c "third-party arithmetic #2").  Row order = ascending point index
 *        (Open3D's order is hash-bucket order; the set per row is what is specified).
 *
 * PARITY: unpinned against the real libraries (neither is importable here) unless the fixtures of
 * tools/gen_goldens_{pytorch3d,open3d}.py are present (tests/test_oracle_thirdparty.py).
 *
 * Queries are independent, so the outer loops are OpenMP-parallel (identical results for any thread count); the
 * thread count follows omp_set_num_threads / OMP_NUM_THREADS — bench.py's cpu_baseline sets it explicitly and
 * reports it.
 */
#include <stdint.h>
#include <stddef.h>

static inline float d2f(const float* a, const float* b)
{
    float dx = a[0] - b[0];
    float dy = a[1] - b[1];
    float dz = a[2] - b[2];
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

void nfo_ball_query_firstk(const float* q, int64_t nq, const float* p, int64_t np,
                           float radius, int K, float* dists2, int64_t* idx, float* nn)
{
    const float r2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < nq; ++i) {
        int cnt = 0;
        float* dd = dists2 + i * K;
        int64_t* ii = idx + i * K;
        float* nnp = nn + i * K * 3;
        for (int k = 0; k < K; ++k) { dd[k] = 0.f; ii[k] = -1; nnp[3*k] = nnp[3*k+1] = nnp[3*k+2] = 0.f; }
        for (int64_t j = 0; j < np && cnt < K; ++j) {
            float s = d2f(q + 3 * i, p + 3 * j);
            if (s < r2) {
                dd[cnt] = s; ii[cnt] = j;
                nnp[3*cnt] = p[3*j]; nnp[3*cnt+1] = p[3*j+1]; nnp[3*cnt+2] = p[3*j+2];
                ++cnt;
            }
        }
    }
}

/* counts per query -> counts[nq]; returns total */
int64_t nfo_radius_count(const float* q, int64_t nq, const float* p, int64_t np,
                         float radius, int ignore_query_point, int64_t* counts)
{
    const float r2 = radius * radius;
    int64_t tot = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+:tot)
    for (int64_t i = 0; i < nq; ++i) {
        int64_t c = 0;
        const float* qi = q + 3 * i;
        for (int64_t j = 0; j < np; ++j) {
            const float* pj = p + 3 * j;
            if (ignore_query_point && qi[0] == pj[0] && qi[1] == pj[1] && qi[2] == pj[2]) continue;
            if (d2f(qi, pj) <= r2) ++c;
        }
        counts[i] = c; tot += c;
    }
    return tot;
}

/* row_splits[nq+1] already holds the exclusive prefix sum of counts */
void nfo_radius_csr(const float* q, int64_t nq, const float* p, int64_t np,
                    float radius, int ignore_query_point, const int64_t* row_splits,
                    int32_t* idx, float* dist2)
{
    const float r2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < nq; ++i) {
        int64_t o = row_splits[i];
        const float* qi = q + 3 * i;
        for (int64_t j = 0; j < np; ++j) {
            const float* pj = p + 3 * j;
            if (ignore_query_point && qi[0] == pj[0] && qi[1] == pj[1] && qi[2] == pj[2]) continue;
            float s = d2f(qi, pj);
            if (s <= r2) { idx[o] = (int32_t)j; dist2[o] = s; ++o; }
        }
    }
}

/* How much of the ball query depends on FP contraction ("parity unpinned" caveat): pytorch3d's CUDA kernel accumulates
 * `dist2 += diff * diff` over d = 0, 1, 2 and nvcc contracts that to FMAs by default, the restatement above evaluates mul + add.  For every
 * (query, point) pair both sums are formed (the contracted one with fmaf; only pairs within 1e-5 relative of r*r can differ and are evaluated
 * twice) and compared against r*r with the library's strict < (inclusive = 0: pytorch3d ball_query) or <= (inclusive = 1: Open3D
 * FixedRadiusSearch; K = a bound above any count then).
 * out[0] = pairs tested, out[1] = pairs whose in / out decision differs, out[2] = queries whose first-K index list differs,
 * out[3] = pairs whose decision agrees but whose fp32 d2 differs by an ulp or more (reported in dists, not in the set). */
#include <math.h>
void nfo_ball_query_contraction_sensitivity(const float* q, int64_t nq, const float* p, int64_t np, float radius, int K, int inclusive,
                                            int64_t* out)
{
    const float r2 = radius * radius, band = r2 * 1e-5f;
    int64_t flips = 0, qdiff = 0, ulps = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+:flips, qdiff, ulps)
    for (int64_t i = 0; i < nq; ++i) {
        int ca = 0, cb = 0, differs = 0;
        for (int64_t j = 0; j < np && (ca < K || cb < K); ++j) {
            const float* a = q + 3 * i; const float* b = p + 3 * j;
            const float s = d2f(a, b);
            int ina = inclusive ? s <= r2 : s < r2, inb = ina;
            if (fabsf(s - r2) <= band || ina) {
                const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
                float t = dx * dx;
                t = fmaf(dy, dy, t);
                t = fmaf(dz, dz, t);
                inb = inclusive ? t <= r2 : t < r2;
                if (ina && inb && t != s) ++ulps;
            }
            if (ina != inb) { ++flips; if (ca < K && cb < K) differs = 1; }
            else if (ina && (ca < K) != (cb < K)) differs = 1;
            ca += ina && ca < K; cb += inb && cb < K;
        }
        qdiff += differs;
    }
    out[0] = nq * np; out[1] = flips; out[2] = qdiff; out[3] = ulps;
}

/* thread control for the timed CPU baseline (bench.py): n <= 0 leaves the OpenMP default */
#ifdef _OPENMP
#include <omp.h>
void nfo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int nfo_max_threads(void) { return omp_get_max_threads(); }
#else
void nfo_set_threads(int n) { (void)n; }
int nfo_max_threads(void) { return 1; }
#endif
