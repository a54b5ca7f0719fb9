// MFMA (v_mfma_f32_*_f32) path of libcnfhip: fused RHS and fused Tsit5 step kernels.
#pragma once
#include "../../include/cnfhip.h"
#include "cnf_dev.h"

struct MfmaPlan {
    int variant = 0;            // 0: unsupported shape
    float* d_packed = nullptr;  // weights re-laid for the kernel (padded, LDS image order)
    size_t packed_floats = 0;
    int pdims[CNF_MAX_LAYERS + 1] = {0};
};

void mfma_plan_init(MfmaPlan& p, const NetDesc& nd);
void mfma_plan_free(MfmaPlan& p);
cnf_status mfma_plan_pack(MfmaPlan& p, const NetDesc& nd, const float* d_params, hipStream_t s);
bool mfma_supported(const MfmaPlan& p, const NetDesc& nd, bool train, int B);
cnf_status mfma_rhs(const MfmaPlan& p, const NetDesc& nd, bool train, const float* u,
                    const float* eps, float* du, int B, hipStream_t s);
// f(u + h*sum_j a_j k_j) with nk = 1 (initial-dt probe) -> Ks[0]
cnf_status mfma_rhs_stage(const MfmaPlan& p, const NetDesc& nd, bool train, const StepState* st,
                          float* const U[2], float* const K1[2], float* const Ks[5],
                          const float* eps, int nk, int B, hipStream_t s);
// one full Tsit5 step attempt (6 RHS evaluations + error partials + controller)
cnf_status mfma_step(const MfmaPlan& p, const NetDesc& nd, bool train, StepState* st,
                     float* const U[2], float* const K1[2], float* const Ks[5],
                     const float* eps, float* partials, int B, hipStream_t s);
int mfma_step_launches();
