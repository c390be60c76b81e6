// al_kernels_align.hip -- (stub, replaced below) regions + extension stage
#include <hip/hip_runtime.h>
#include "al_internal.h"
#include "al_device.h"
#include "al_runtime.h"
int al_run_align_stage(al_ctx_t *c) { AL_HIP_CHECK(hipEventRecord(c->ev[ST_REGS + 1], c->stream)); return 0; }
int al_fetch_align(al_ctx_t *c, int *n_regs, al_reg1_t **regs, int *rep_len) { (void)c; (void)n_regs; (void)regs; (void)rep_len; return -1; }
