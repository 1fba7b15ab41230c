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

/* The f64 yardstick sums cost time (a convert + a double add per element).  The timed variant of this
 * library (liboracle_timed: -DORC_NO_YARDSTICK, oracle/Makefile) leaves them out, so that what bench.py's
 * cpu_baseline times is the reference's own work per voxel: same passes, same visiting order, same
 * dtype-width sums, same serial boundary hyperslab.  Arrays come out bit-identical in both variants. */
#ifdef ORC_NO_YARDSTICK
#define Y64(stmt)
#else
#define Y64(stmt) stmt
#endif

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
                    Y64(norm64 += fabs((double)bn);)
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
                    Y64(norm64 += fabs((double)bn);)
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
                    Y64(norm64 += fabs((double)bn);)
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
    T delta = (T)0, rnorm = (T)0;
    double delta64 = 0.0, rnorm64 = 0.0;
    const int64_t outer = N0 * N1;

#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : delta, rnorm, delta64, rnorm64) schedule(static) num_threads(nthreads)
#endif
    for (int64_t ij = 0; ij < outer; ++ij) {
        const int64_t i = ij / N1, j = ij % N1;
        /* (i+1)%N0, (j+1)%N1, (k+1)%N2 do not depend on l: the reference's compiler hoists them out of the
         * innermost loop too (utils.c:5574-5628); only (l+1)%N3 is evaluated per voxel */
        const int64_t di = (((i + 1) % N0) - i) * st[0], dj = (((j + 1) % N1) - j) * st[1];
        for (int64_t k = 0; k < N2; ++k) {
            const int64_t dk = (((k + 1) % N2) - k) * st[2];
            const int64_t x0 = i * st[0] + j * st[1] + k * st[2];
            for (int64_t l = 0; l < N3; ++l) {
                const int64_t x = x0 + l;
                const int64_t dl = ((l + 1) % N3) - l;
                T s;
                if (nax == 4) /* utils.c:5641: ((t0 + t1) + t2) + t3 */
                    s = ((lm[0] * (b[0][x] - b[0][x + di]) + lm[1] * (b[1][x] - b[1][x + dj])) +
                         lm[2] * (b[2][x] - b[2][x + dk])) + lm[3] * (b[3][x] - b[3][x + dl]);
                else /* 3-D: canonical axes 1,2,3 (utils.pyx:176-178) */
                    s = (lm[0] * (b[0][x] - b[0][x + dj]) + lm[1] * (b[1][x] - b[1][x + dk])) +
                        lm[2] * (b[2][x] - b[2][x + dl]);
                T old = recon[x];
                T nw = orig[x] - s;
                recon[x] = nw;
                delta += (T)fabs((double)(T)(nw - old));
                rnorm += (T)fabs((double)old);
                Y64(delta64 += fabs((double)(T)(nw - old));)
                Y64(rnorm64 += fabs((double)old);)
            }
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
            Y64(acc64 += (double)t * (double)t;)
        }
    out[0] = (double)acc;
    out[1] = acc64;
}

#undef Y64
#undef FN
#undef CAT
#undef CAT_
