/*
 * lrt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement ("oracle") of the reference tracer; see the header of
 * lrt_oracle_impl.inc for scope, citations and the parity-pinning status.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Test switch: feed the any-hit program in ascending order of t instead of BVH traversal order.  The order only matters for rays
 * with a stale K-buffer slot (a hit closer than 0.2 m, forward.cu:214): there the reference's `cnt` -- hence which hits a chunk looks
 * at -- depends on the order in which OptiX happens to report the hits; ascending order is the realisation the HIP path replays. */
static int g_sorted_anyhit = 0;
void orc_set_sorted_anyhit(int on) { g_sorted_anyhit = on; }

#define REAL float
#define SFX(n) n##_f32
#define R_SQRT sqrtf
#define R_EXP expf
#define R_LOG logf
#define R_FABS fabsf
#include "lrt_oracle_impl.inc"
#undef REAL
#undef SFX
#undef R_SQRT
#undef R_EXP
#undef R_LOG
#undef R_FABS

#define REAL double
#define SFX(n) n##_f64
#define R_SQRT sqrt
#define R_EXP exp
#define R_LOG log
#define R_FABS fabs
#include "lrt_oracle_impl.inc"
#undef REAL
#undef SFX

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif
