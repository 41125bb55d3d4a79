/*
 * dcx_oracle.c — CPU oracle for the DiffCo score(+grad) hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Plain-C (gcc, OpenMP) restatement of the reference
 * algorithm (ucsdarclab/diffco: diffco/kernel.py, kernel_perceptrons.py, model.py, utils.py),
 * in fp32 (`*_f32`) and fp64 (`*_f64`, the tolerance referee).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle.so; the shipped
 * path (diffco_amd + libdcx.so) never links, imports or calls it.
 *
 * Pinning: the reference holds no golden vector / known-answer test for this path
 * (SURVEY.md §4, §8c), and it is pure Python, so there is no oracle/_ref build.  The oracle
 * is pinned against outputs of the reference itself generated in the build container by
 * tools/make_golden.py (tests/golden npz files) — see tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/dcx.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(x) CAT(x, _f32)
#define MATH(fn) CAT(fn, f)
#include "dcx_oracle_impl.h"
#undef REAL
#undef FN
#undef MATH

#define REAL double
#define FN(x) CAT(x, _f64)
#define MATH(fn) fn
#include "dcx_oracle_impl.h"
#undef REAL
#undef FN
#undef MATH

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
