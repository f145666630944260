/* cfnmpc_dev.h -- entry points of DEVELOPMENT builds only (make -C crazyflie_nmpc_amd/csrc DEV=1, -DCFN_DEV): experiments whose
 * results are recorded in DESIGN.md / profiles/ and that the shipped library neither contains nor exports.  Not part of the
 * boundary (include/): nothing in tests/ or bench.py uses them; the tools that do (tools/chunked_pair.py,
 * tools/sub_fleet_emul.py) say so. */
#ifndef CFNMPC_DEV_H
#define CFNMPC_DEV_H
#include "../../include/cfnmpc.h"
#ifdef __cplusplus
extern "C" {
#endif
/* linearisation + start-solve factorisation alternating in chunks of `chunk` stages (0: the product's
 * two kernels); *ms = average duration of one pair over `reps` repetitions */
int cfnmpc_debug_chunked_pair(cfnmpc_solver *s, int chunk, int reps, double *ms, void *stream);
/* sums of gains / feed-forward terms / checkpoints (bitwise comparison of two start solves) */
int cfnmpc_debug_checksum(cfnmpc_solver *s, double *out3);
/* the parts of one fused RTI step on separate streams (1: k_linfactor, 2: everything behind it; flags bit 0: k_forward_half) */
int cfnmpc_debug_solve_part(cfnmpc_solver *s, int part, int flags, void *stream);
#ifdef __cplusplus
}
#endif
#endif
