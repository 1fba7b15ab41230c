/*
 * TEST INFRASTRUCTURE ONLY -- see tvdn_oracle.c for the header that governs this file.
 *
 * Type-generic body, included twice by tvdn_oracle.c with
 *   T      = float | double
 *   SUF    = f32   | f64
 *
 * Canonical layout: every array is a C-contiguous 4-D block shape[0..3]; a 3-D
 * reference array (N0,N1,N2) is passed as (1,N0,N1,N2) and its axis `ax` as ax+1.
 * That keeps the reference's row-major visiting order (and therefore its
 * dtype-width running sums) unchanged.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* clipval: reference cyTVDN/anisotropic.pyx:11-12, generated C anisotropic.c:2409-2451.
 * Two ternaries, so that a NaN input propagates (fmin/fmax would swallow it). */
static inline T FN(orc_clip)(T a, T val)
{
    T lo = -val;
    T t = (lo > a) ? lo : a;
    return (val < t) ? val : t;
}

/*
 * Accumulator update, plain (d == NULL) or FISTA (d != NULL).
 * Reference: accumulator_update_4D        cyTVDN/anisotropic.pyx:17-84
 *            accumulator_update_4D_FISTA  cyTVDN/anisotropic.pyx:89-164
 *            accumulator_update_3D        cyTVDN/anisotropic.pyx:169-237
 *            accumulator_update_3D_FISTA  cyTVDN/anisotropic.pyx:243-317
 *
 * Visiting order is the reference's: main block (index along `ax` >= 1) in row-major
 * order, then the ax-index-0 hyperslab in row-major order.  `*norm_T` is the running
 * sum kept in T exactly as the reference keeps it with one OpenMP thread;
 * `*norm_f64` is the same sum kept in double (the order-independent yardstick).
 * With nthreads > 1 the main block is split statically over the fused (i,j) index as
 * the reference's prange does; norm_T then depends on the thread count (as upstream).
 */
void FN(orc_accumulator_update)(const T *a, T *b, T *d, T tk, int ax, T clip, int bc_mode,
                                const int64_t shape[4], int nthreads,
                                double *norm_T, double *norm_f64)
{
    const int64_t N0 = shape[0], N1 = shape[1], N2 = shape[2], N3 = shape[3];
    const int64_t st[4] = {N1 * N2 * N3, N2 * N3, N3, 1};
    int64_t start[4] = {0, 0, 0, 0};
    start[ax] = 1;
    const int64_t back = st[ax];
    const int64_t outer1 = N1 - start[1];
    const int64_t outer = (N0 - start[0]) * outer1;
    T norm = (T)0;
    double norm64 = 0.0;

    const T *restrict ar = a;
    T *restrict br = b;
    T *restrict dr = d;
    if (d) {
#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : norm, norm64) schedule(static) num_threads(nthreads)
#endif
        for (int64_t ij = 0; ij < outer; ++ij) {
            const int64_t i = start[0] + ij / outer1;
            const int64_t j = start[1] + ij % outer1;
            for (int64_t k = start[2]; k < N2; ++k) {
                const int64_t x0 = i * st[0] + j * st[1] + k * st[2];
                for (int64_t l = start[3]; l < N3; ++l) {
                    const int64_t x = x0 + l;
                    const T v = (ar[x] - ar[x - back]) + br[x];
                    const T dn = FN(orc_clip)(v, clip);
                    const T bn = dn + tk * (dn - dr[x]);
                    dr[x] = dn;
                    br[x] = bn;
                    norm += (T)fabs((double)bn);
                    norm64 += fabs((double)bn);
                }
            }
        }
    } else {
#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : norm, norm64) schedule(static) num_threads(nthreads)
#endif
        for (int64_t ij = 0; ij < outer; ++ij) {
            const int64_t i = start[0] + ij / outer1;
            const int64_t j = start[1] + ij % outer1;
            for (int64_t k = start[2]; k < N2; ++k) {
                const int64_t x0 = i * st[0] + j * st[1] + k * st[2];
                for (int64_t l = start[3]; l < N3; ++l) {
                    const int64_t x = x0 + l;
                    const T v = (ar[x] - ar[x - back]) + br[x];
                    const T bn = FN(orc_clip)(v, clip);
                    br[x] = bn;
                    norm += (T)fabs((double)bn);
                    norm64 += fabs((double)bn);
                }
            }
        }
    }

    /* boundary hyperslab: anisotropic.pyx:56-82 (4-D), :209-235 (3-D) */
    int64_t stop[4] = {N0, N1, N2, N3};
    stop[ax] = 1;
    int64_t delta = 0;
    if (bc_mode == 0)
        delta = (shape[ax] - 1) * st[ax];
    else if (bc_mode == 1)
        delta = st[ax];
    for (int64_t m = 0; m < stop[0]; ++m)
        for (int64_t n = 0; n < stop[1]; ++n)
            for (int64_t o = 0; o < stop[2]; ++o)
                for (int64_t p = 0; p < stop[3]; ++p) {
                    const int64_t x = m * st[0] + n * st[1] + o * st[2] + p;
                    T v = (a[x] - a[x + delta]) + b[x];
                    T dn = FN(orc_clip)(v, clip);
                    T bn;
                    if (d) {
                        bn = dn + tk * (dn - d[x]);
                        d[x] = dn;
                    } else {
                        bn = dn;
                    }
                    b[x] = bn;
                    norm += (T)fabs((double)bn);
                    norm64 += fabs((double)bn);
                }
    *norm_T = (double)norm;
    *norm_f64 = norm64;
}

/*
 * Reconstruction update (BC 0 and BC 2 share the periodic-wrap branch).
 * Reference: datacube_update_4D cyTVDN/utils.pyx:54-125 (association of the sum taken
 *            from the generated C, utils.c:5641), datacube_update_3D utils.pyx:131-199.
 * `nax` = 3 or 4 regularised axes = the LAST nax axes of the canonical 4-D block;
 * b[q] and lm[q] belong to canonical axis (4 - nax + q).
 * out[0] = delta/rnorm in T (reference return value, one thread),
 * out[1] = sum|new-old| in f64, out[2] = sum|old| in f64.
 */
void FN(orc_datacube_update)(const T *orig, T *recon, const T *const *b, const T *lm, int nax,
                             const int64_t shape[4], int nthreads, double out[3])
{
    const int64_t N0 = shape[0], N1 = shape[1], N2 = shape[2], N3 = shape[3];
    const int64_t st[4] = {N1 * N2 * N3, N2 * N3, N3, 1};
    const int a0 = 4 - nax;
    T delta = (T)0, rnorm = (T)0;
    double delta64 = 0.0, rnorm64 = 0.0;
    const int64_t outer = N0 * N1;

#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : delta, rnorm, delta64, rnorm64) schedule(static) num_threads(nthreads)
#endif
    for (int64_t ij = 0; ij < outer; ++ij) {
        const int64_t i = ij / N1, j = ij % N1;
        for (int64_t k = 0; k < N2; ++k)
            for (int64_t l = 0; l < N3; ++l) {
                const int64_t idx[4] = {i, j, k, l};
                const int64_t x = i * st[0] + j * st[1] + k * st[2] + l;
                T s = (T)0;
                for (int q = 0; q < nax; ++q) {
                    const int axq = a0 + q;
                    const int64_t nx = x + (((idx[axq] + 1) % shape[axq]) - idx[axq]) * st[axq];
                    T term = lm[q] * (b[q][x] - b[q][nx]);
                    s = (q == 0) ? term : (s + term);
                }
                T old = recon[x];
                T nw = orig[x] - s;
                recon[x] = nw;
                delta += (T)fabs((double)(T)(nw - old));
                rnorm += (T)fabs((double)old);
                delta64 += fabs((double)(T)(nw - old));
                rnorm64 += fabs((double)old);
            }
    }
    out[0] = (double)(T)(delta / rnorm);
    out[1] = delta64;
    out[2] = rnorm64;
}

/* Sum of squared differences: sum_square_error_4D/3D cyTVDN/utils.pyx:14-30, :35-49.
 * out[0] in T (reference order, one thread), out[1] in f64. */
void FN(orc_sum_square_error)(const T *a, const T *b, const int64_t shape[4], int nthreads,
                              double out[2])
{
    const int64_t inner = shape[2] * shape[3];
    const int64_t outer = shape[0] * shape[1];
    T acc = (T)0;
    double acc64 = 0.0;
#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : acc, acc64) schedule(static) num_threads(nthreads)
#endif
    for (int64_t ij = 0; ij < outer; ++ij)
        for (int64_t kl = 0; kl < inner; ++kl) {
            T t = a[ij * inner + kl] - b[ij * inner + kl];
            acc += t * t;
            acc64 += (double)t * (double)t;
        }
    out[0] = (double)acc;
    out[1] = acc64;
}

#undef FN
#undef CAT
#undef CAT_
