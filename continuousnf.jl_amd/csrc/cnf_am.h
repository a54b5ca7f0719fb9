// Device-side building blocks shared by the MFMA kernels outside the fused step kernel
// (cnf_grad.hip: pullback; cnf_trace.hip: exact trace): workgroup geometry, the LDS-only barrier, the
// weight-fragment stream and the tile GEMM on v_mfma_f32_16x16x4_f32.  Conventions as in cnf_mfma.hip:
// A = 16x16 fragment of a padded row-major weight image (b128 per lane from global/L2), B = the
// operand in LDS as [column][feature] (b128 per lane), accumulator lane = (column l & 15, rows
// 4 (l >> 4) .. +3).
#pragma once
#include "cnf_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- second derivative of the activations ----------------------------------------------------
__device__ __forceinline__ void cnf_act2(int kind, float a, float& h, float& d1, float& d2) {
    switch (kind) {
        case 0: h = a; d1 = 1.0f; d2 = 0.0f; break;
        case 1: h = cnf_tanh(a); d1 = fmaf(-h, h, 1.0f); d2 = -2.0f * h * d1; break;
        case 2: { float s = cnf_sigmoid(a); h = s; d1 = s * (1.0f - s); d2 = d1 * (1.0f - 2.0f * s); } break;
        case 3: { float s = cnf_sigmoid(a); h = (a > 15.0f) ? a : log1pf(__expf(a)); d1 = s; d2 = s * (1.0f - s); } break;
        case 4: h = fmaxf(a, 0.0f); d1 = (a > 0.0f) ? 1.0f : 0.0f; d2 = 0.0f; break;
        case 5: {
            float s = cnf_sigmoid(a), ds = s * (1.0f - s);
            h = a * s; d1 = s * (1.0f + a * (1.0f - s)); d2 = 2.0f * ds + a * ds * (1.0f - 2.0f * s);
        } break;
        default: {
            float e = __expf(fminf(a, 0.0f));
            h = (a > 0.0f) ? a : e - 1.0f; d1 = (a > 0.0f) ? 1.0f : e; d2 = (a > 0.0f) ? 0.0f : e;
        } break;
    }
}


#define AM_NS 16
#ifndef AM_WAVES
#define AM_WAVES 8            // measured at config 3, B = 8192 (round 3, row-major images): (waves, chunk) = (8, 2) 11.1 ms
#endif                        // per gradient, (8, 4) 11.6, (4, 8) 12.3

#define AM_EC (AM_WAVES * 4)          // feature lanes of an elementwise pass: AM_THREADS / 16 samples
#define AM_THREADS (AM_WAVES * 64)


// Workgroup barrier that waits for LDS traffic only: __syncthreads() also drains the vector-memory
// counter, which would stall on the weight fragments prefetched for the next sweep (and on the
// stores of the factor arrays, which nobody reads in this kernel).
__device__ __forceinline__ void am_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A-operand stream.  The weight fragments come from L2 (~1-2k cycles away), far longer than the few
// MFMAs of one k-block, so a wave keeps a whole CHUNK of fragments (AM_CH k-blocks of one output tile)
// in registers and has the NEXT chunk in flight while it multiplies: the next chunk of the same tile,
// else the first chunk of its next tile, else the first chunk of the NEXT sweep's image -- that one is
// issued before the barrier and the elementwise pass between the sweeps.
// The images are in FRAGMENT order (AdjMfmaLayout::ff_off / fr_off: 1 KB per tile and k-block, lane L's four values at
// 16 L), so a wave's load is one contiguous KB.  Round 5, config 5's pullback (k_adj_mfma, B = 2048, phase stamps of
// -DAM_STAMPS): with the row-major images (a wave's b128 load = 16 rows x 64 bytes, four lanes 16 apart per row) the loads
// cost 42 k of a stage's 142 k cycles whatever the chunk depth -- the address unit, not the latency; in fragment order 24 k at
// AM_CH = 2, 17 k at 4 (loss_and_grad 12.2 -> 11.3 -> 10.8 ms), 8 no better.
#ifndef AM_CH
#define AM_CH 4
#endif
#ifdef AM_ROWPAD               // A/B: row-major images with a stride of k_p + 4 floats (k_mfma's streamed layout): 11.1 against
                               // 10.8 ms at config 5.  (The reverse experiment, k_mfma's stream in fragment order: 25 against
                               // 17 us per evaluation -- its rows are re-used from L1 by the next k-block, here they are not.)
#define AM_IMG_PAD 4
#define AM_KSTEP 16
#else
#define AM_IMG_PAD 0
#define AM_KSTEP 256
#endif
struct AFrag { f32x4 a[AM_CH]; };

// Straight-line loads only: a branch around a load makes the compiler copy the loaded registers into
// the loop-carried ones right away (waiting for every load in turn).  So the ADDRESS is selected, the
// loads are unconditional; k-blocks past the end re-read the last valid fragment and are never used.
__device__ __forceinline__ void am_load(AFrag& f, const float* __restrict__ p, int last) {
#pragma unroll
    for (int i = 0; i < AM_CH; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(p + AM_KSTEP * min(i, last));
}
__device__ __forceinline__ const float* am_addr(const float* __restrict__ img, int k_p, int tile, int u0) {
    const int lane = threadIdx.x & 63;
#ifdef AM_ROWPAD
    return img + (size_t)(16 * tile + (lane & 15)) * (k_p + AM_IMG_PAD) + 4 * (lane >> 4) + 16 * u0;
#else
    return img + ((size_t)(tile * (k_p >> 4) + u0) * 64 + lane) * 4;      // fragment order: AdjMfmaLayout::ff_off / fr_off
#endif
}
__device__ __forceinline__ void am_first(AFrag& f, const float* __restrict__ img, int rows_p, int k_p) {
    const int wave = threadIdx.x >> 6;
    const int tile = min(wave, (rows_p >> 4) - 1);       // waves without a tile fetch a valid one (unused)
    am_load(f, am_addr(img, k_p, tile, 0), (k_p >> 4) - 1);
}

// Out tile(s) of one sweep: rows_p x k_p image against the [sample][feature] operand X in LDS.
// `pf` holds am_first() of this image on entry and am_first() of (nimg, nrows_p, nk_p) on exit (nimg
// may be null).  `pre`: optional per-row vector (bias) fetched before the MFMAs.
template <class Epi>
__device__ __forceinline__ void am_gemm(const float* __restrict__ img, int rows_p, int k_p, const float* X, int PS,
                                        AFrag& pf, const float* __restrict__ nimg, int nrows_p, int nk_p,
                                        const float* __restrict__ pre, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const float* xrow = X + s * PS + 4 * q;
    const int nu = k_p >> 4, nt = rows_p >> 4;
    // where the next sweep's first chunk lives (a valid address even when there is no next sweep)
    const float* nfirst = nimg ? am_addr(nimg, nk_p, min(wave, (nrows_p >> 4) - 1), 0) : am_addr(img, k_p, 0, 0);
    const int nfirst_last = nimg ? (nk_p >> 4) - 1 : 0;
    if (wave >= nt) {                                    // no tile in this sweep: only hand the prefetch on
        if (nimg) am_load(pf, nfirst, nfirst_last);
        return;
    }
    for (int t = wave; t < nt; t += AM_WAVES) {
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (pre) pv = *reinterpret_cast<const f32x4*>(pre + 16 * t + 4 * q);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int u0 = 0; u0 < nu; u0 += AM_CH) {
            const AFrag cur = pf;
            const float* np;
            int nl;
            if (u0 + AM_CH < nu) { np = am_addr(img, k_p, t, u0 + AM_CH); nl = nu - u0 - AM_CH - 1; }
            else if (t + AM_WAVES < nt) { np = am_addr(img, k_p, t + AM_WAVES, 0); nl = nu - 1; }
            else { np = nfirst; nl = nfirst_last; }
#ifndef AM_ABL_NOALOAD
            am_load(pf, np, nl);
#endif
            const int n = min(AM_CH, nu - u0);
            // all B fragments of the chunk first (LDS latency paid once, not per k-block), then the
            // MFMAs; k-blocks i and i+1 use different accumulators so their chains interleave
            f32x4 b[AM_CH];
#pragma unroll
            for (int i = 0; i < AM_CH; ++i) b[i] = *reinterpret_cast<const f32x4*>(xrow + 16 * (u0 + min(i, n - 1)));
            if (n == AM_CH) {
#pragma unroll
                for (int i = 0; i < AM_CH; i += 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[i][c], b[i][c], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[i + 1][c], b[i + 1][c], acc1, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < AM_CH; ++i) {
                    if (i < n) {
                        if (i & 1) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[i][c], b[i][c], acc1, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[i][c], b[i][c], acc0, 0, 0, 0);
                        }
                    }
                }
            }
        }
        acc0 += acc1;
        epi(16 * t + 4 * q, s, acc0, pv);      // acc0[j] = Out[row 16t + 4q + j][sample s]
    }
}

// per-sample sum of squares of X[s][0..n): every thread returns the value of sample (threadIdx.x >> 4)
__device__ __forceinline__ float am_colnorm2(const float* X, int PS, int n, float* red) {
    const int s = (threadIdx.x >> 4) & 15, part = (threadIdx.x & 15) | ((threadIdx.x >> 8) << 4);   // AM_EC parts
    float v = 0.f;
    for (int r = part; r < n; r += AM_EC) { const float x = X[s * PS + r]; v = fmaf(x, x, v); }
    red[part * AM_NS + s] = v;
    am_barrier();
    float t = 0.f;
    for (int p = 0; p < AM_EC; ++p) t += red[p * AM_NS + s];
    am_barrier();
    return t;
}

// 4 values of a global [sample][feature] row: one 16-byte store when the row layout allows it
__device__ __forceinline__ void am_store4(float* g, f32x4 v, int r0, int n_valid, bool vec) {
#ifndef AM_ABL_NOSTORE
    // (non-temporal: the factor arrays are written once and read back a step group later by the contraction, 0.9 GB on -- with
    // ordinary stores the contraction's reads ran 13 % slower, 239 against 209 us; A/B: -DAM_PLAIN_STORE)
#ifdef AM_PLAIN_STORE
    if (vec && r0 + 3 < n_valid) *reinterpret_cast<f32x4*>(g) = v;
#else
    if (vec && r0 + 3 < n_valid) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(g));
#endif
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j < n_valid) g[j] = v[j];
    }
#endif
}


// Same stream, NCT column tiles per A fragment (compile-time, so the accumulators stay in registers):
// Out[rows_p x 16 NCT] = img[rows_p x k_p] * X, X given as [column][feature] with stride PX; column tile c
// covers columns 16c .. 16c+15.  epi(r0, s, acc) gets acc[c][j] = Out[r0 + j][16 c + s].
template <int NCT, class Epi>
__device__ __forceinline__ void am_gemm_multi(const float* __restrict__ img, int rows_p, int k_p, const float* X, int PX,
                                              AFrag& pf, const float* __restrict__ nimg, int nrows_p, int nk_p,
                                              Epi&& epi, const float* __restrict__ pre = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const float* xrow = X + s * PX + 4 * q;
    const int nu = k_p >> 4, nt = rows_p >> 4;
    const float* nfirst = nimg ? am_addr(nimg, nk_p, min(wave, (nrows_p >> 4) - 1), 0) : am_addr(img, k_p, 0, 0);
    const int nfirst_last = nimg ? (nk_p >> 4) - 1 : 0;
    if (wave >= nt) {
        if (nimg) am_load(pf, nfirst, nfirst_last);
        return;
    }
    for (int t = wave; t < nt; t += AM_WAVES) {
        f32x4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pre) acc[0] = *reinterpret_cast<const f32x4*>(pre + 16 * t + 4 * q);     // column tile 0 starts from the bias
        for (int u0 = 0; u0 < nu; u0 += AM_CH) {
            const AFrag cur = pf;
            const float* np;
            int nl;
            if (u0 + AM_CH < nu) { np = am_addr(img, k_p, t, u0 + AM_CH); nl = nu - u0 - AM_CH - 1; }
            else if (t + AM_WAVES < nt) { np = am_addr(img, k_p, t + AM_WAVES, 0); nl = nu - 1; }
            else { np = nfirst; nl = nfirst_last; }
            am_load(pf, np, nl);
            const int n = min(AM_CH, nu - u0);
#pragma unroll
            for (int i = 0; i < AM_CH; ++i) {
                if (i < n) {
                    f32x4 b[NCT];
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        b[c] = *reinterpret_cast<const f32x4*>(xrow + (size_t)16 * c * PX + 16 * (u0 + i));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int c = 0; c < NCT; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[i][k], b[c][k], acc[c], 0, 0, 0);
                    }
                }
            }
        }
        epi(16 * t + 4 * q, s, acc);
    }
}
