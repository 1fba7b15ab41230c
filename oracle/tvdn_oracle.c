/*
 * tvdn_oracle.c -- CPU restatement of the cyTVDN anisotropic TV hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load or
 * call anything under oracle/.  The product path (cytvdn_amd/) never falls back to
 * it: when the HIP library is missing the product raises.
 *
 * What it restates (reference file:line, /root/reference):
 *   clipval                          cyTVDN/anisotropic.pyx:11-12  (C: anisotropic.c:2409-2451)
 *   accumulator_update_{3D,4D}       cyTVDN/anisotropic.pyx:169-237, :17-84
 *   accumulator_update_{3D,4D}_FISTA cyTVDN/anisotropic.pyx:243-317, :89-164
 *   datacube_update_{3D,4D}          cyTVDN/utils.pyx:131-199, :54-125   (BC 0 / BC 2 branch)
 *   sum_square_error_{3D,4D}         cyTVDN/utils.pyx:35-49, :14-30
 * The iteration loop that strings these together (cyTVDN/cyTVDN.py:147-242, :368-430)
 * is restated in oracle/oracle.py.
 *
 * Parity pin: PINNED.  oracle/make_golden.py runs the reference's own compiled kernels
 * (oracle/_ref, built from the C files the reference ships by oracle/Makefile) and the
 * reference's own Python driver on seeded inputs and commits the outputs as
 * the tests/golden .npz fixtures; tests/test_oracle_golden.py checks this restatement against every
 * one of them bit-for-bit (recon, accumulators) and to a stated tolerance (scalars).
 *
 * Arithmetic contract: all operations in the array dtype,
 * round-to-nearest, left-to-right as written, NO fused multiply-add, denormals kept.
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile does).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define T float
#define SUF f32
#include "tvdn_oracle_impl.h"
#undef T
#undef SUF

#define T double
#define SUF f64
#include "tvdn_oracle_impl.h"
#undef T
#undef SUF

int orc_abi_version(void) { return 1; }

int orc_openmp_enabled(void)
{
#if defined(_OPENMP)
    return 1;
#else
    return 0;
#endif
}
