// MFMA path -- placeholder until the fused kernels land: reports "unsupported" so that
// AUTO resolves to the generic path and an explicit MFMA request fails loudly.
#include "cnf_mfma.h"

void mfma_plan_init(MfmaPlan& p, const NetDesc&) { p.variant = 0; }
void mfma_plan_free(MfmaPlan& p) { if (p.d_packed) (void)hipFree(p.d_packed); p.d_packed = nullptr; }
cnf_status mfma_plan_pack(MfmaPlan&, const NetDesc&, const float*, hipStream_t) { return CNF_OK; }
bool mfma_supported(const MfmaPlan&, const NetDesc&, bool, int) { return false; }
cnf_status mfma_rhs(const MfmaPlan&, const NetDesc&, bool, const float*, const float*, float*, int, hipStream_t) { return CNF_ERR_UNSUPPORTED; }
cnf_status mfma_rhs_stage(const MfmaPlan&, const NetDesc&, bool, const StepState*, float* const[2], float* const[2], float* const[5], const float*, int, int, hipStream_t) { return CNF_ERR_UNSUPPORTED; }
cnf_status mfma_step(const MfmaPlan&, const NetDesc&, bool, StepState*, float* const[2], float* const[2], float* const[5], const float*, float*, int, hipStream_t) { return CNF_ERR_UNSUPPORTED; }
int mfma_step_launches() { return 1; }
