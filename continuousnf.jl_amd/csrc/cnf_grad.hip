// Gradient of the training loss w.r.t. the flat parameter vector.
//
// Reference: MLJModelInterface.fit (src/exts/mlj_ext/core_icnf.jl:59-73) differentiates
// loss(icnf, TrainMode(), xs, ps, st) (src/icnf.jl:481-490) with Enzyme through the ODE solve
// (SciMLSensitivity; third party).  Here: the discrete adjoint of the Tsit5 steps actually taken,
// i.e. the exact gradient of the number `loss` returns (oracle/cnf_grad_oracle.py has the algebra).
//
// Kernels in this file:
//   k_adj_mfma    the pullback on MFMA (default; second half of this file)
//   k_adj<TS>     VALU fallback: pullback of ONE augmented_f evaluation (src/icnf.jl:318-350, :384-420) at a stage
//                 state: a workgroup owns TS samples, a thread owns one feature of every layer; the
//                 four sweeps (forward, reverse of eps, forward tangent, reverse of the cotangent)
//                 are matrix-vector products against the weights (read through L2, coalesced: W for
//                 the forward sweeps, a transposed copy for the reverse ones) with the activations of
//                 the TS samples in LDS.  Emits zbar and, per layer, the four factors of the weight
//                 gradient in [sample][feature] arrays.
//   k_wgrad[_mfma] Wbar_l += ABAR_l^T H_{l-1} + PBAR_l^T T_{l-1}, bbar_l += sum_b ABAR_l: a batch-
//                 contraction GEMM, 64x64 output tiles x K-splits; every (tile, split) workgroup owns
//                 its slice of a partial buffer (no atomics: bit-reproducible), summed at the end.
//   small elementwise kernels: stage combination, lambda update, final cotangent, partial reduce,
//   weight transpose.
// DESIGN.md section 4.4 has the measurements.
#include "cnf_grad.h"
#include <atomic>
#include <type_traits>

#include "cnf_am.h"
#include "cnf_split.h"
#include <cstdlib>


// deterministic per-sample block sums: wave shuffle tree, then a fixed-order sum over the waves
template <int TS>
__device__ __forceinline__ void block_sum_ts(float (&v)[TS], float* scratch /* [nwaves][TS] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        float x = v[s];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) scratch[wave * TS + s] = x;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        float x = 0.f;
        for (int w = 0; w < nw; ++w) x += scratch[w * TS + s];
        v[s] = x;
    }
    __syncthreads();
}

// out[s] = sum_i M[row + i*ld] * x[s][i]   (row = this thread's feature; M column-major with leading dim ld)
template <int TS>
__device__ __forceinline__ void gemv(const float* __restrict__ M, int ld, int n, const float* x, int xs,
                                     float (&acc)[TS]) {
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const float w0 = M[(size_t)i * ld], w1 = M[(size_t)(i + 1) * ld], w2 = M[(size_t)(i + 2) * ld],
                    w3 = M[(size_t)(i + 3) * ld];
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const float* xv = x + s * xs + i;
            acc[s] = fmaf(w0, xv[0], acc[s]);
            acc[s] = fmaf(w1, xv[1], acc[s]);
            acc[s] = fmaf(w2, xv[2], acc[s]);
            acc[s] = fmaf(w3, xv[3], acc[s]);
        }
    }
    for (; i < n; ++i) {
        const float w = M[(size_t)i * ld];
#pragma unroll
        for (int s = 0; s < TS; ++s) acc[s] = fmaf(w, x[s * xs + i], acc[s]);
    }
}

// LDS carve-up per sample (floats): H | T | D1 | D2Q | TB | HB0 | HB1 | EPS
struct AdjLds {
    int H, T, D1, D2, TB, HB0, HB1, EPS, per_sample;
    __host__ __device__ static AdjLds make(const GradLayout& g, int n_in) {
        AdjLds l;
        int p = 0;
        l.H = p; p += g.sum_in + g.out_last;
        l.T = p; p += g.sum_in + g.out_last;
        l.D1 = p; p += g.sum_out;
        l.D2 = p; p += g.sum_out;
        l.TB = p; p += g.sum_out;
        l.HB0 = p; p += g.max_dim;
        l.HB1 = p; p += g.max_dim;
        l.EPS = p; p += n_in;
        l.per_sample = p;
        return l;
    }
};

template <int TS>
__global__ void k_adj(NetDesc nd, GradLayout gl, AdjArgs a) {
    extern __shared__ float lds[];
    const AdjLds L = AdjLds::make(gl, nd.n_in);
    const int PS = L.per_sample;
    float* red = lds + (size_t)TS * PS;                  // [nwaves][TS]
    const int f = threadIdx.x;
    const int b0 = blockIdx.x * TS;
    const int n_in = nd.n_in, D = n_in + 3, NL = nd.n_layers;
    const int in0 = gl.in0;

    // ---- stage inputs: z (and ys), eps, cotangent of zdot -------------------------------------
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int b = b0 + s;
        float* S = lds + (size_t)s * PS;
        if (f < in0) {
            float v = 0.f;
            if (b < a.B) v = f < n_in ? a.ustage[(size_t)b * D + f] : a.ys[(size_t)b * nd.n_cond + (f - n_in)];
            S[L.H + f] = v;
        }
        if (f < n_in) S[L.EPS + f] = b < a.B ? a.eps[(size_t)b * n_in + f] : 0.f;
    }
    __syncthreads();

    // ---- forward sweep: h_l, sigma', sigma'' ---------------------------------------------------
    for (int l = 0; l < NL; ++l) {
        const int in = l == 0 ? in0 : nd.dims[l], out = nd.dims[l + 1];
        const int hin = gl.in_off[l], hout = l + 1 < NL ? gl.in_off[l + 1] : gl.sum_in;
        if (f < out) {
            float acc[TS];
            const float bias = a.P[nd.b_off[l] + f];
#pragma unroll
            for (int s = 0; s < TS; ++s) acc[s] = bias;
            gemv<TS>(a.P + nd.w_off[l] + f, out, in, lds + L.H + hin, PS, acc);
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                float h, d1, d2;
                cnf_act2(nd.acts[l], acc[s], h, d1, d2);
                float* S = lds + (size_t)s * PS;
                S[L.H + hout + f] = h;
                S[L.D1 + gl.out_off[l] + f] = d1;
                S[L.D2 + gl.out_off[l] + f] = d2;
            }
        }
        __syncthreads();
    }
    const int hL = gl.sum_in;                               // offset of h_L (= zdot) in H
    const int outL = nd.dims[NL];                           // == n_in

    // ---- ahat = kbar_z + c_E zdot/|zdot| -> HB0 ------------------------------------------------
    float nz[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const float v = f < outL ? lds[(size_t)s * PS + L.H + hL + f] : 0.f;
        nz[s] = v * v;
    }
    if (nd.norm_z) block_sum_ts<TS>(nz, red);
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int b = b0 + s;
        if (f < outL) {
            float kb = 0.f;
            if (b < a.B) {
                kb = a.cb * a.lam[(size_t)b * n_in + f];
                for (int m = 0; m < a.nw; ++m) kb = fmaf(a.wc[m], a.w[m][(size_t)b * n_in + f], kb);
                kb *= a.hstep;
            }
            float v = kb;
            if (nd.norm_z) {
                const float nrm = sqrtf(nz[s]);
                if (nrm > 0.f) v = fmaf(a.c_E, lds[(size_t)s * PS + L.H + hL + f] / nrm, v);
            }
            lds[(size_t)s * PS + L.HB0 + f] = v;
        }
    }
    __syncthreads();

    // two chains run over the layers: the tangent chain (forward) and the tbar chain (reverse).
    // VJP mode: omega = eps  -> tbar chain first (gives eJ), then tau, then the tangent chain.
    // JVP mode: tau = eps    -> tangent chain first (gives J eps), then omega, then the tbar chain.
    auto tangent_chain = [&]() {
        for (int l = 0; l < NL; ++l) {
            const int in = l == 0 ? in0 : nd.dims[l], out = nd.dims[l + 1];
            const int hin = gl.in_off[l], hout = l + 1 < NL ? gl.in_off[l + 1] : gl.sum_in;
            if (f < out) {
                float acc[TS];
#pragma unroll
                for (int s = 0; s < TS; ++s) acc[s] = 0.f;
                gemv<TS>(a.P + nd.w_off[l] + f, out, in, lds + L.T + hin, PS, acc);
#pragma unroll
                for (int s = 0; s < TS; ++s) {
                    float* S = lds + (size_t)s * PS;
                    const int o = gl.out_off[l] + f;
                    S[L.T + hout + f] = S[L.D1 + o] * acc[s];      // t_l = sigma' .* p_l
                    S[L.D2 + o] = S[L.D2 + o] * acc[s];            // q_l = sigma'' .* p_l
                }
            }
            __syncthreads();
        }
    };
    auto tbar_chain = [&]() {                                    // TB[l] holds tbar_{l+1} (layer l's output side)
        for (int l = NL - 1; l >= 0; --l) {
            const int in = l == 0 ? in0 : nd.dims[l], out = nd.dims[l + 1];
            // pbar_l = tbar_l .* sigma'_l   (kept in HB1, written out for the weight gradient)
            if (f < out) {
#pragma unroll
                for (int s = 0; s < TS; ++s) {
                    float* S = lds + (size_t)s * PS;
                    const int o = gl.out_off[l] + f;
                    const float pb = S[L.TB + o] * S[L.D1 + o];
                    S[L.HB1 + f] = pb;
                    if (b0 + s < a.B) a.PB[(size_t)(b0 + s) * gl.sum_out + o] = pb;
                }
            }
            __syncthreads();
            if (l > 0) {
                if (f < in) {
                    float acc[TS];
#pragma unroll
                    for (int s = 0; s < TS; ++s) acc[s] = 0.f;
                    gemv<TS>(a.PT + nd.w_off[l] + f, in, out, lds + L.HB1, PS, acc);
#pragma unroll
                    for (int s = 0; s < TS; ++s) lds[(size_t)s * PS + L.TB + gl.out_off[l - 1] + f] = acc[s];
                }
            } else if (f < n_in) {                               // tbar_0 rows of z: eJ (VJP mode) -> HB1 is busy, use T's h_0 slot later
                float acc[TS];
#pragma unroll
                for (int s = 0; s < TS; ++s) acc[s] = 0.f;
                gemv<TS>(a.PT + nd.w_off[0] + f, in0, out, lds + L.HB1, PS, acc);
#pragma unroll
                for (int s = 0; s < TS; ++s) lds[(size_t)s * PS + L.T + f] = acc[s];   // parked in T_0
            }
            __syncthreads();
        }
    };

    const int oL = gl.out_off[NL - 1];
    if (!nd.jvp) {
        // tbar_L = eps
        if (f < outL) {
#pragma unroll
            for (int s = 0; s < TS; ++s) lds[(size_t)s * PS + L.TB + oL + f] = lds[(size_t)s * PS + L.EPS + f];
        }
        __syncthreads();
        tbar_chain();                                            // T_0[0..n_in) = eJ
        float nj[TS];
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const float v = f < n_in ? lds[(size_t)s * PS + L.T + f] : 0.f;
            nj[s] = v * v;
        }
        if (nd.norm_j) block_sum_ts<TS>(nj, red);
        // tau = -c_l eps + c_n eJ/|eJ|   (rows of ys carry a zero tangent)
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            if (f < in0) {
                float* S = lds + (size_t)s * PS;
                float v = 0.f;
                if (f < n_in) {
                    v = -a.c_l * S[L.EPS + f];
                    if (nd.norm_j) {
                        const float nrm = sqrtf(nj[s]);
                        if (nrm > 0.f) v = fmaf(a.c_n, S[L.T + f] / nrm, v);
                    }
                }
                S[L.T + f] = v;
            }
        }
        __syncthreads();
        tangent_chain();
    } else {
        // tau = eps
#pragma unroll
        for (int s = 0; s < TS; ++s)
            if (f < in0) lds[(size_t)s * PS + L.T + f] = f < n_in ? lds[(size_t)s * PS + L.EPS + f] : 0.f;
        __syncthreads();
        tangent_chain();                                         // T_L = J eps
        float nj[TS];
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const float v = f < outL ? lds[(size_t)s * PS + L.T + hL + f] : 0.f;
            nj[s] = v * v;
        }
        if (nd.norm_j) block_sum_ts<TS>(nj, red);
        // omega = -c_l eps + c_n Je/|Je|
        if (f < outL) {
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                float* S = lds + (size_t)s * PS;
                float v = -a.c_l * S[L.EPS + f];
                if (nd.norm_j) {
                    const float nrm = sqrtf(nj[s]);
                    if (nrm > 0.f) v = fmaf(a.c_n, S[L.T + hL + f] / nrm, v);
                }
                S[L.TB + oL + f] = v;
            }
        }
        __syncthreads();
        // the tbar chain parks tbar_0 in T_0, which the weight gradient still needs as t_0 = eps:
        // save and restore it around the chain
        float keep[TS];
#pragma unroll
        for (int s = 0; s < TS; ++s) keep[s] = f < n_in ? lds[(size_t)s * PS + L.T + f] : 0.f;
        tbar_chain();
#pragma unroll
        for (int s = 0; s < TS; ++s)
            if (f < n_in) lds[(size_t)s * PS + L.T + f] = keep[s];
        __syncthreads();
    }

    // ---- hbar chain: abar_l = hbar_l sigma' + tbar_l q_l ; hbar_{l-1} = W_l' abar_l --------------
    int cur = L.HB0, nxt = L.HB1;
    for (int l = NL - 1; l >= 0; --l) {
        const int in = l == 0 ? in0 : nd.dims[l], out = nd.dims[l + 1];
        if (f < out) {
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                float* S = lds + (size_t)s * PS;
                const int o = gl.out_off[l] + f;
                const float ab = fmaf(S[cur + f], S[L.D1 + o], S[L.TB + o] * S[L.D2 + o]);
                S[cur + f] = ab;
                if (b0 + s < a.B) a.AB[(size_t)(b0 + s) * gl.sum_out + o] = ab;
            }
        }
        __syncthreads();
        const int rows = l == 0 ? n_in : in;
        if (f < rows) {
            float acc[TS];
#pragma unroll
            for (int s = 0; s < TS; ++s) acc[s] = 0.f;
            gemv<TS>(a.PT + nd.w_off[l] + f, in, out, lds + cur, PS, acc);
#pragma unroll
            for (int s = 0; s < TS; ++s) lds[(size_t)s * PS + nxt + f] = acc[s];
        }
        __syncthreads();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    // ---- outputs: zbar and the layer inputs / tangents for the weight gradient ------------------
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int b = b0 + s;
        if (b >= a.B) continue;
        const float* S = lds + (size_t)s * PS;
        if (f < n_in) a.w_out[(size_t)b * n_in + f] = S[cur + f];
        for (int i = f; i < gl.sum_in; i += blockDim.x) {
            a.HS[(size_t)b * gl.sum_in + i] = S[L.H + i];
            a.TS[(size_t)b * gl.sum_in + i] = S[L.T + i];
        }
    }
}

// ---- weight gradient: 64x64 tiles x K-splits ---------------------------------------------------
#define WG_T 64
#define WG_K 16
__global__ void __launch_bounds__(256)
k_wgrad(NetDesc nd, GradLayout gl, const float* __restrict__ AB, const float* __restrict__ PB,
        const float* __restrict__ HS, const float* __restrict__ TSb, float* __restrict__ gpart,
        int n_params, int B, int chunk) {
    __shared__ float sA[WG_K][WG_T + 4], sP[WG_K][WG_T + 4], sH[WG_K][WG_T + 4], sT[WG_K][WG_T + 4];
    // locate this workgroup's tile
    int tile = blockIdx.x, l = 0, to = 0, ti = 0;
    for (; l < nd.n_layers; ++l) {
        const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
        const int no = (out + WG_T - 1) / WG_T, ni = (in + WG_T - 1) / WG_T;
        if (tile < no * ni) { to = tile / ni; ti = tile % ni; break; }
        tile -= no * ni;
    }
    if (l == nd.n_layers) return;
    const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
    const int o0 = to * WG_T, i0 = ti * WG_T;
    const int oo = gl.out_off[l], io = gl.in_off[l];
    const int k0 = blockIdx.y * chunk, k1 = min(B, k0 + chunk);
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    float acc[4][4] = {};
    float bsum[4] = {};
    for (int kb = k0; kb < k1; kb += WG_K) {
        for (int e = t; e < WG_K * WG_T; e += 256) {
            const int r = e / WG_T, c = e % WG_T, b = kb + r;
            const bool vb = b < k1;
            const bool vo = vb && o0 + c < out, vi = vb && i0 + c < in;
            sA[r][c] = vo ? AB[(size_t)b * gl.sum_out + oo + o0 + c] : 0.f;
            sP[r][c] = vo ? PB[(size_t)b * gl.sum_out + oo + o0 + c] : 0.f;
            sH[r][c] = vi ? HS[(size_t)b * gl.sum_in + io + i0 + c] : 0.f;
            sT[r][c] = vi ? TSb[(size_t)b * gl.sum_in + io + i0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WG_K; ++r) {
            const float4 a4 = *reinterpret_cast<const float4*>(&sA[r][4 * ty]);
            const float4 p4 = *reinterpret_cast<const float4*>(&sP[r][4 * ty]);
            const float4 h4 = *reinterpret_cast<const float4*>(&sH[r][4 * tx]);
            const float4 t4 = *reinterpret_cast<const float4*>(&sT[r][4 * tx]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w};
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                bsum[x] += av[x];
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fmaf(av[x], hv[y], fmaf(pv[x], tv[y], acc[x][y]));
            }
        }
        __syncthreads();
    }
    float* g = gpart + (size_t)blockIdx.y * n_params;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int o = o0 + 4 * ty + x;
        if (o >= out) continue;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = i0 + 4 * tx + y;
            if (i < in) g[nd.w_off[l] + o + (size_t)i * out] += acc[x][y];     // Lux layout: out x in, column-major
        }
        if (ti == 0 && tx == 0) g[nd.b_off[l] + o] += bsum[x];
    }
}

// The same contraction on v_mfma_f32_16x16x4_f32: the K index of the MFMA is the sample.  A workgroup
// (4 waves) owns a 64x64 output tile for one K-split; wave w the 16-column strip w (4 accumulator
// tiles).  Per 16 samples the four operand tiles go through LDS; an MFMA k-step takes samples
// 4c..4c+3: lane (x = l & 15, q = l >> 4) feeds A = abar[4c+q][o0+16to+x], B = h[4c+q][i0+16w+x].
#define WM_LD (WG_T + 16)          // row stride: 16 q + x covers all 64 banks
__global__ void __launch_bounds__(256)
k_wgrad_mfma(NetDesc nd, GradLayout gl, const float* __restrict__ AB, const float* __restrict__ PB,
             const float* __restrict__ HS, const float* __restrict__ TSb, float* __restrict__ gpart,
             int n_params, int B, int chunk) {
    __shared__ float sA[WG_K][WM_LD], sP[WG_K][WM_LD], sH[WG_K][WM_LD], sT[WG_K][WM_LD];
    int tile = blockIdx.x, l = 0, to_ = 0, ti = 0;
    for (; l < nd.n_layers; ++l) {
        const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
        const int no = (out + WG_T - 1) / WG_T, ni = (in + WG_T - 1) / WG_T;
        if (tile < no * ni) { to_ = tile / ni; ti = tile % ni; break; }
        tile -= no * ni;
    }
    if (l == nd.n_layers) return;
    const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
    const int o0 = to_ * WG_T, i0 = ti * WG_T;
    const int oo = gl.out_off[l], io = gl.in_off[l];
    const int k0 = blockIdx.y * chunk, k1 = min(B, k0 + chunk);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, x = lane & 15, q = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // software pipeline: the next 16 samples travel global -> registers while the MFMAs run on the
    // current ones in LDS (a K-split has only a handful of chunks; without this the loop is latency-bound)
    const int lc = t & 63, lr = t >> 6;                     // this thread loads column lc of rows lr, lr+4, lr+8, lr+12
    const bool co = o0 + lc < out, ci = i0 + lc < in;
    float ra[4], rp[4], rh[4], rt[4];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = kb + lr + 4 * j;
            const bool vb = b < k1;
            ra[j] = (vb && co) ? AB[(size_t)b * gl.sum_out + oo + o0 + lc] : 0.f;
            rp[j] = (vb && co) ? PB[(size_t)b * gl.sum_out + oo + o0 + lc] : 0.f;
            rh[j] = (vb && ci) ? HS[(size_t)b * gl.sum_in + io + i0 + lc] : 0.f;
            rt[j] = (vb && ci) ? TSb[(size_t)b * gl.sum_in + io + i0 + lc] : 0.f;
        }
    };
    fetch(k0);
    for (int kb = k0; kb < k1; kb += WG_K) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sA[lr + 4 * j][lc] = ra[j]; sP[lr + 4 * j][lc] = rp[j];
            sH[lr + 4 * j][lc] = rh[j]; sT[lr + 4 * j][lc] = rt[j];
        }
        __syncthreads();
        if (kb + WG_K < k1) fetch(kb + WG_K);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float bh = sH[4 * c + q][16 * w + x], bt = sT[4 * c + q][16 * w + x];
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[4 * c + q][16 * to + x], bh, acc[to], 0, 0, 0);
                acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(sP[4 * c + q][16 * to + x], bt, acc[to], 0, 0, 0);
            }
        }
        if (ti == 0 && t < WG_T) {
#pragma unroll
            for (int r = 0; r < WG_K; ++r) bsum += sA[r][t];
        }
        __syncthreads();
    }
    float* g = gpart + (size_t)blockIdx.y * n_params;
    const int i = i0 + 16 * w + x;
#pragma unroll
    for (int to = 0; to < 4; ++to) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + 16 * to + 4 * q + j;
            if (o < out && i < in) g[nd.w_off[l] + o + (size_t)i * out] += acc[to][j];
        }
    }
    if (ti == 0 && t < WG_T && o0 + t < out) g[nd.b_off[l] + o0 + t] += bsum;
}

// The same contraction with every fp32 product formed from six bf16 MFMA terms on exactly split operands
// (v_mfma_f32_16x16x32_bf16: 32 samples per instruction; accuracy of the fp32 MFMA, cnf_split.h).  64 x 64 output tile per
// workgroup, wave w the 16-column strip w of the input side.  Per 32 samples:
//   * output side (abar, pbar; every wave needs all four 16-row groups): thread (lc, lr) fetches 8 consecutive samples of
//     column lc, splits them and stores the three pieces as 16-byte chunks of [piece][column][32 samples] images in LDS
//     (chunks XOR-swizzled by the column: the operand reads -- lane (q, x): column 16 tile + x, chunk q -- are conflict-free);
//   * input side (h, t): lane (x, q) of wave w fetches samples 8q .. 8q + 7 of column 16 w + x -- exactly ITS operand of
//     the MFMAs -- and splits it in registers: that half of the factors never touches LDS (the kernel is bound by LDS
//     traffic: ablations in DESIGN 7.0).
// Two chunks are in flight in registers while a third is multiplied.  K-splits and the reduction over them as k_wgrad_mfma.
#define WB_K 32
#ifdef WB_STAMPS      // per-phase cycle totals of wave 0 of one full-width tile's first K-split (tools/wgrad_stamps.py)
__device__ unsigned long long wb_stamps[8];
extern "C" int cnf_debug_wgrad_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wb_stamps), sizeof(unsigned long long) * 8);
}
#define WB_T(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); wbt[i] += n_ - wb_last; wb_last = n_; } while (0)
#else
#define WB_T(i)
#endif
__global__ void __launch_bounds__(256)
k_wgrad_mfma_b(NetDesc nd, GradLayout gl, const float* __restrict__ AB, const float* __restrict__ PB,
               const float* __restrict__ HS, const float* __restrict__ TSb, float* __restrict__ gpart,
               int n_params, int B, int chunk) {
    constexpr int T = 64, PIECE = T * 64;                              // tile side; one piece of one array: T columns x 64 bytes
    extern __shared__ __attribute__((aligned(16))) char simg[];        // A | P, three pieces each; then the bias sums
    float* sbias = reinterpret_cast<float*>(simg + 2 * 3 * PIECE);     // [4][T]
    // Workgroups go to the 8 XCDs round-robin in launch order, and each XCD has its own L2: all tiles of a K-split -- which
    // read the SAME sample rows -- run next to each other on ONE XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = bx + gridDim.x * by, r = L >> 3;
        bx = r % gridDim.x; by = (L & 7) + 8 * (r / gridDim.x);
    }
    int tile = bx, l = 0, to_ = 0, ti = 0;
    for (; l < nd.n_layers; ++l) {
        const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
        const int no = (out + T - 1) / T, ni = (in + T - 1) / T;
        if (tile < no * ni) { to_ = tile / ni; ti = tile % ni; break; }
        tile -= no * ni;
    }
    if (l == nd.n_layers) return;
    const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
    const int o0 = to_ * T, i0 = ti * T;
    const int oo = gl.out_off[l], io = gl.in_off[l];
    const int k0 = by * chunk, k1 = min(B, k0 + chunk);
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), x = lane & 15, q = lane >> 4;
    const int nto = min(4, (out - o0 + 15) / 16);                      // 16-row groups of the tile that exist
    const bool bstrip = i0 + 16 * w < in;                              // this wave's strip of the input side exists (wave-uniform)
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int lc = lane, lr = w;                            // output side: column lc, samples 8 lr .. 8 lr + 7 of a chunk
    struct Buf { float a[8], p[8], h[8], t[8]; };
    Buf bufA, bufB;
    // Row bases are scalar arithmetic, a request is scalar base + 32-bit lane offset (BYTES).  Columns beyond the layer's
    // widths re-read its last column (their products land in rows / columns of the tile that are never stored); rows beyond
    // the K-split (its last, ragged chunk only) re-read its last row and are zeroed on arrival.
    const unsigned ca = 4u * (unsigned)min(lc, out - o0 - 1);
    const unsigned colb = (unsigned)min(16 * w + x, in - i0 - 1);
    const unsigned cb = 4u * (colb + (unsigned)(8 * q) * (unsigned)gl.sum_in);
#ifdef WB_ABL_NOLOAD
    auto ld = [](const float* base, unsigned off) { float v; asm volatile("v_mov_b32 %0, 1.0" : "=v"(v) : "v"(off), "s"(base)); return v; };
#else
    auto ld = [](const float* base, unsigned off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); };
#endif
    // per-lane byte offsets of the 8 rows of a chunk (loop-invariant registers): a request is then ONE instruction with a
    // scalar chunk base -- no address arithmetic on either unit inside the loop (the scalar adds per request, ~400 per 32
    // samples, were what a wave's instruction stream mostly consisted of)
    unsigned offA[8], offB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        offA[j] = ca + 4u * (unsigned)(8 * lr + j) * (unsigned)gl.sum_out;
        offB[j] = cb + 4u * (unsigned)j * (unsigned)gl.sum_in;
    }
    auto fetch = [&](Buf& r, int kb) {
        const float* bA = AB + (size_t)kb * gl.sum_out + oo + o0;
        const float* bP = PB + (size_t)kb * gl.sum_out + oo + o0;
        const float* bH = HS + (size_t)kb * gl.sum_in + io + i0;
        const float* bT = TSb + (size_t)kb * gl.sum_in + io + i0;
        if (kb + WB_K <= k1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { r.a[j] = ld(bA, offA[j]); r.p[j] = ld(bP, offA[j]); r.h[j] = ld(bH, offB[j]); r.t[j] = ld(bT, offB[j]); }
        } else {                                                       // the last, ragged chunk: rows clamped to the split's last row
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned oa = ca + 4u * (unsigned)(min(kb + 8 * lr + j, k1 - 1) - kb) * (unsigned)gl.sum_out;
                const unsigned ob = 4u * (colb + (unsigned)(min(kb + 8 * q + j, k1 - 1) - kb) * (unsigned)gl.sum_in);
                r.a[j] = ld(bA, oa); r.p[j] = ld(bP, oa); r.h[j] = ld(bH, ob); r.t[j] = ld(bT, ob);
            }
        }
    };
    auto arrived = [&](Buf& r, int kb) {
        if (kb + WB_K > k1) {                                          // the last, ragged chunk of the K-split
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (kb + 8 * lr + j >= k1) { r.a[j] = 0.f; r.p[j] = 0.f; }
                if (kb + 8 * q + j >= k1) { r.h[j] = 0.f; r.t[j] = 0.f; }
            }
        }
    };
    struct Op { bf16x8 h, m, l; };
    auto split8 = [&](const float (&v)[8]) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ hp, mp, lp;
#ifdef WB_ABL_NOSPLIT
        hp.x = __float_as_uint(v[0]); hp.y = __float_as_uint(v[2]); hp.z = __float_as_uint(v[4]); hp.w = __float_as_uint(v[6]);
        mp.x = __float_as_uint(v[1]); mp.y = __float_as_uint(v[3]); mp.z = __float_as_uint(v[5]); mp.w = __float_as_uint(v[7]); lp = hp;
#else
        { unsigned h_, m_, l_; s3b_split2(v[0], v[1], h_, m_, l_); hp.x = h_; mp.x = m_; lp.x = l_; } { unsigned h_, m_, l_; s3b_split2(v[2], v[3], h_, m_, l_); hp.y = h_; mp.y = m_; lp.y = l_; }
        { unsigned h_, m_, l_; s3b_split2(v[4], v[5], h_, m_, l_); hp.z = h_; mp.z = m_; lp.z = l_; } { unsigned h_, m_, l_; s3b_split2(v[6], v[7], h_, m_, l_); hp.w = h_; mp.w = m_; lp.w = l_; }
#endif
        Op o; o.h = __builtin_bit_cast(bf16x8, hp); o.m = __builtin_bit_cast(bf16x8, mp); o.l = __builtin_bit_cast(bf16x8, lp);
        return o;
    };
    // chunk c (8 samples) of column r at chunk position c ^ ((-(r >> 2)) & 3) of its 64-byte row
    auto put = [&](int arr, int col, const Op& o) {
        char* d = simg + arr * 3 * PIECE + col * 64 + 16 * (lr ^ ((-(col >> 2)) & 3));
        *(bf16x8*)d = o.h; *(bf16x8*)(d + PIECE) = o.m; *(bf16x8*)(d + 2 * PIECE) = o.l;
    };
    auto get = [&](int arr, int col) {                      // operand of this lane: column `col`, samples 8q .. 8q + 7
        const char* d = simg + arr * 3 * PIECE + col * 64 + 16 * (q ^ ((-(col >> 2)) & 3));
        Op o; o.h = *(const bf16x8*)d; o.m = *(const bf16x8*)(d + PIECE); o.l = *(const bf16x8*)(d + 2 * PIECE);
        return o;
    };
    auto mm6 = [&](const Op& a, const Op& b, f32x4 c) -> f32x4 {     // smallest terms first
#ifdef WB_ABL_NOMFMA
        asm volatile("" : "+v"(c) : "v"(a.h), "v"(a.m), "v"(a.l), "v"(b.h), "v"(b.m), "v"(b.l)); return c;
#endif
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    };
#ifdef WB_STAMPS
    unsigned long long wbt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb_last = __builtin_amdgcn_s_memtime();
#endif
    auto body = [&](Buf& r, int kb) {
        WB_T(5);
        arrived(r, kb);
        WB_T(0);
        if (ti == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += r.a[j];
        }
        put(0, lc, split8(r.a)); put(1, lc, split8(r.p));
        const Op bh = split8(r.h), bt = split8(r.t);                   // this lane's input-side operands, from its own registers
        WB_T(1);
        __syncthreads();
        WB_T(2);
        if (kb + 2 * WB_K < k1) fetch(r, kb + 2 * WB_K);
        WB_T(6);
        if (bstrip) {
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                if (to < nto) {
                    const Op aa = get(0, 16 * to + x), ap = get(1, 16 * to + x);
                    acc[to] = mm6(aa, bh, acc[to]);
                    acc[to] = mm6(ap, bt, acc[to]);
                }
            }
        }
        WB_T(3);
        __syncthreads();
        WB_T(4);
    };
    if (k0 < k1) fetch(bufA, k0);
    if (k0 + WB_K < k1) fetch(bufB, k0 + WB_K);
    for (int kb = k0; kb < k1; kb += 2 * WB_K) {
        body(bufA, kb);
        if (kb + WB_K < k1) body(bufB, kb + WB_K);
    }
#ifdef WB_STAMPS
    if (l == 1 && to_ == 0 && ti == 0 && by == 0 && t == 0) { for (int i_ = 0; i_ < 8; ++i_) wb_stamps[i_] = wbt[i_]; wb_stamps[7] = (unsigned long long)((k1 - k0 + WB_K - 1) / WB_K); }   // (slot 6: issuing the requests)
#endif
    float* g = gpart + (size_t)by * n_params;
    const int i = i0 + 16 * w + x;
#pragma unroll
    for (int to = 0; to < 4; ++to) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + 16 * to + 4 * q + j;
            if (o < out && i < in) g[nd.w_off[l] + o + (size_t)i * out] += acc[to][j];
        }
    }
    if (ti == 0) {                                          // bias: the column sums of abar, four sample groups per column
        sbias[lr * T + lc] = bsum;
        __syncthreads();
        for (int c = t; c < T; c += 256)
            if (o0 + c < out) g[nd.b_off[l] + o0 + c] += (sbias[c] + sbias[T + c]) + (sbias[2 * T + c] + sbias[3 * T + c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The contraction, wave-local (the default where the row layouts are 16-byte aligned).  What bounded k_wgrad_mfma_b was not
// arithmetic, LDS or HBM but the NUMBER of vector-memory instructions: a 4-byte request per lane costs the address unit as
// much as a 16-byte one (34 cycles per request with two workgroups per CU).  Here a lane fetches 4 consecutive
// columns of a row with ONE 16-byte request, 8 rows of its sample group: lane (cg, sg) = columns 4 cg .. 4 cg + 3, samples
// 8 sg .. 8 sg + 7 of the 32-sample chunk -- and that IS an MFMA operand layout (lane (x, q): 8 consecutive k) for the tile
// made of the columns {4 x + c}: output "tile c" is a comb of every fourth column, which costs nothing (the rows and columns
// of the result are just stored where they belong).  One wave holds the whole 64 x 32-sample chunk of all four factor
// arrays in registers and multiplies the full 64 x 64 output tile (16 accumulator tiles): no LDS, no barriers, a quarter of
// the memory instructions.  A workgroup is ONE wave (512 registers); K-splits x tiles waves fill the chip one per SIMD.
// The next chunk is requested as soon as the current one has been split (its raw registers are free then) and travels
// under the 192 MFMAs.
__global__ void __launch_bounds__(64)
k_wgrad_wave(NetDesc nd, GradLayout gl, const float* __restrict__ AB, const float* __restrict__ PB,
             const float* __restrict__ HS, const float* __restrict__ TSb, float* __restrict__ gpart,
             int n_params, int B, int chunk) {
    constexpr int T = 64;
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {                            // all tiles of a K-split on one XCD (they read the same rows)
        const int L = bx + gridDim.x * by, r = L >> 3;
        bx = r % gridDim.x; by = (L & 7) + 8 * (r / gridDim.x);
    }
    int tile = bx, l = 0, to_ = 0, ti = 0;
    for (; l < nd.n_layers; ++l) {
        const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
        const int no = (out + T - 1) / T, ni = (in + T - 1) / T;
        if (tile < no * ni) { to_ = tile / ni; ti = tile % ni; break; }
        tile -= no * ni;
    }
    if (l == nd.n_layers) return;
    const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
    const int o0 = to_ * T, i0 = ti * T;
    const int oo = gl.out_off[l], io = gl.in_off[l];
    const int k0 = by * chunk, k1 = min(B, k0 + chunk);
    if (k0 >= k1) return;
    const int lane = threadIdx.x, cg = lane & 15, sg = lane >> 4;
    // column groups beyond the layer's width re-read its last group (their products land in rows / columns of the tile
    // that are never stored)
    const int ga = min(4 * cg, ((out - o0 - 1) >> 2) << 2), gb = min(4 * cg, ((in - i0 - 1) >> 2) << 2);
    const unsigned offA = 4u * ((unsigned)ga + (unsigned)(8 * sg) * (unsigned)gl.sum_out);     // BYTES from the chunk's first row
    const unsigned offB = 4u * ((unsigned)gb + (unsigned)(8 * sg) * (unsigned)gl.sum_in);
    struct Op { bf16x8 h, m, l; };
    // raw chunk: [row j of the sample group][4 columns].  (A second chunk in flight -- 512 registers -- was measured: 256 against
    // 240 us; the kernel runs at the same speed WITHOUT its MFMAs and splits: it is bound by the memory system's
    // throughput on this access pattern, 0.91 GB from HBM + 0.7 GB of L2 hits per four-step launch.)
    struct Raw { f32x4 a[8], p[8], h[8], t[8]; };
    Raw r0_;
    // Buffer loads: descriptor base = the K-split's first row at the tile's first column, record count = up to the split's
    // last row, so rows past the split (its ragged last chunk) come back as zeros by the range check; a request is
    // descriptor + the lane's byte offset, which carries the row offset (one vector add per request).  (Plain loads cost 112 64-bit vector
    // adds per chunk here, and hand-written requests are unsafe: the compiler copies their destination registers.)
    auto rsrc = [&](const float* X, int sum, int col0) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (size_t)k0 * sum + col0), 0, (int)(((size_t)(k1 - k0) * sum - col0) * 4), 0x00020000);
    };
    const auto dA = rsrc(AB, gl.sum_out, oo + o0), dP = rsrc(PB, gl.sum_out, oo + o0);
    const auto dH = rsrc(HS, gl.sum_in, io + i0), dT = rsrc(TSb, gl.sum_in, io + i0);
    auto fetch = [&](f32x4 (&x)[8], decltype(dA) d, int sum, unsigned off, int kb) {
#pragma unroll
#ifndef WGW_AUX
#define WGW_AUX 0
#endif
        // the row offset is part of the CHECKED offset (voffset): soffset takes no part in the range check on gfx9 raw buffers,
        // so a row past the split would otherwise come back as whatever an earlier, larger contraction left there
        for (int j = 0; j < 8; ++j) x[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d, (int)(off + (unsigned)((kb - k0 + j) * sum * 4)), 0, WGW_AUX));
    };
    auto split8 = [&](const f32x4 (&r)[8], int c) {         // column c of the lane's four: its 8 samples -> three bf16x8 pieces
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ hp, mp, lp;
        { unsigned h_, m_, l_; s3b_split2(r[0][c], r[1][c], h_, m_, l_); hp.x = h_; mp.x = m_; lp.x = l_; }
        { unsigned h_, m_, l_; s3b_split2(r[2][c], r[3][c], h_, m_, l_); hp.y = h_; mp.y = m_; lp.y = l_; }
        { unsigned h_, m_, l_; s3b_split2(r[4][c], r[5][c], h_, m_, l_); hp.z = h_; mp.z = m_; lp.z = l_; }
        { unsigned h_, m_, l_; s3b_split2(r[6][c], r[7][c], h_, m_, l_); hp.w = h_; mp.w = m_; lp.w = l_; }
        Op o; o.h = __builtin_bit_cast(bf16x8, hp); o.m = __builtin_bit_cast(bf16x8, mp); o.l = __builtin_bit_cast(bf16x8, lp);
        return o;
    };
    f32x4 acc[4][4];                                        // [output comb ca][input comb cb]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};                        // bias: column sums of abar over this lane's samples
    auto fetch_all = [&](Raw& r, int kb) {
        fetch(r.a, dA, gl.sum_out, offA, kb); fetch(r.h, dH, gl.sum_in, offB, kb);
        fetch(r.p, dP, gl.sum_out, offA, kb); fetch(r.t, dT, gl.sum_in, offB, kb);
    };
    fetch_all(r0_, k0);
    // One factor pair of a chunk, X (output side) x Y (input side): the two pairs (abar x h, pbar x t) go one after the other so
    // that only one pair's split operands are alive (both at once pushed half of them through AGPR copies).  A wave issues in
    // order, so vector work and MFMAs overlap only where independent instructions of both kinds sit next to each other:
    // the output side is split first (exposed) and its next chunk requested at once (those raw registers are free), then
    // input-side comb cb + 1 is split BESIDE the 24 MFMAs of comb cb, and the input side's next chunk is requested as soon
    // as its last comb has been split.
    auto pair = [&](f32x4 (&rx)[8], f32x4 (&ry)[8], decltype(dA) X, decltype(dA) Y, int kb, bool bias) {
        const bool more = kb + WB_K < k1;
        if (bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs += rx[j];
        }
        Op O[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) O[c] = split8(rx, c);
        if (more) fetch(rx, X, gl.sum_out, offA, kb + WB_K);
        Op Ic = split8(ry, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            Op In = Ic;
            if (cb < 3) In = split8(ry, cb + 1);
            if (cb == 2 && more) fetch(ry, Y, gl.sum_in, offB, kb + WB_K);   // (comb 3 has just been split: ry is free)
            // 4 accumulator tiles x 6 terms; consecutive MFMAs go to different accumulators (no back-to-back dependence)
#ifdef WB_ABL_NOMFMA
#define WGW_TERM(xa, yb) _Pragma("unroll") for (int ca_ = 0; ca_ < 4; ++ca_) asm volatile("" : "+v"(acc[ca_][cb]) : "v"(O[ca_].xa), "v"(Ic.yb));
#else
#define WGW_TERM(xa, yb) _Pragma("unroll") for (int ca_ = 0; ca_ < 4; ++ca_) \
            acc[ca_][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(O[ca_].xa, Ic.yb, acc[ca_][cb], 0, 0, 0);
#endif
            WGW_TERM(l, h) WGW_TERM(h, l) WGW_TERM(m, m) WGW_TERM(m, h) WGW_TERM(h, m) WGW_TERM(h, h)      // smallest terms first
#undef WGW_TERM
            Ic = In;
        }
    };
    for (int kb = k0; kb < k1; kb += WB_K) {
        pair(r0_.a, r0_.h, dA, dH, kb, ti == 0);
        pair(r0_.p, r0_.t, dP, dT, kb, false);
    }
    // acc[ca][cb][j] = sum over the samples of abar[.][o] * h[.][i] + pbar * t  with  o = o0 + 4 (4 sg + j) + ca,  i = i0 + 4 cg + cb
    float* g = gpart + (size_t)by * n_params;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int i = i0 + 4 * cg + cb;
#pragma unroll
        for (int ca_ = 0; ca_ < 4; ++ca_) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = o0 + 4 * (4 * sg + j) + ca_;
                if (o < out && i < in) g[nd.w_off[l] + o + (size_t)i * out] += acc[ca_][cb][j];
            }
        }
    }
    if (ti == 0) {                                          // bias: sum the four sample groups of a column group
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = bs[c];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            const int o = o0 + 4 * cg + c;
            if (sg == 0 && 4 * cg == ga && o < out) g[nd.b_off[l] + o] += v;
        }
    }
}

static bool wgrad_wave_ok(const NetDesc& nd, const GradLayout& g) {      // 16-byte aligned rows and layer blocks
    if ((g.sum_in & 3) || (g.sum_out & 3)) return false;
    for (int l = 0; l < nd.n_layers; ++l) if ((g.in_off[l] & 3) || (g.out_off[l] & 3)) return false;
    return true;
}

// grad[p] = sum over the K-splits, in a fixed order: 64 parameters per workgroup, four threads per parameter take every
// fourth split (up to 128 splits: one thread per parameter walked them in 31 us), their partial sums are added in order
__global__ void __launch_bounds__(256)
k_grad_reduce(const float* __restrict__ gpart, float* __restrict__ grad, int n_params, int ksplit) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + c;
    float s = 0.f;
    if (p < n_params)
        for (int k = q; k < ksplit; k += 4) s += gpart[(size_t)k * n_params + p];
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && p < n_params) grad[p] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// The end of a SUBMITTED gradient (cnf_loss_grad_submit): the sum of the partials as k_grad_reduce forms it -- or zeros when the
// launch that wrote them gave up (its final state says so: nothing downstream of an unwaited launch may consume garbage) --
// and the loss from the five sums (cnf_loss_from_sums' arithmetic), both left in device memory.
__global__ void __launch_bounds__(256)
k_grad_finish(const float* __restrict__ gpart, float* __restrict__ grad, int n_params, int ksplit, const StepState* __restrict__ state,
              const float* __restrict__ sums5, float l1, float l2, float l3, int train, float* __restrict__ loss_dev) {
    __shared__ float part[4][64];
    const bool ok = state->n_partials >= 0 && state->done && !state->nonfinite;
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + c;
    float s = 0.f;
    if (ok && p < n_params)
        for (int k = q; k < ksplit; k += 4) s += gpart[(size_t)k * n_params + p];
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && p < n_params) grad[p] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_dev) {
        const double cnt = sums5[4];
        const double v = train ? (-(double)sums5[0] + (double)l1 * sums5[1] + (double)l2 * sums5[2] + (double)l3 * sums5[3]) / cnt
                               : -(double)sums5[0] / cnt;
        *loss_dev = ok ? (float)v : __builtin_nanf("");
    }
}

// WT_l[i + o*in] = W_l[o + i*out]  (same offsets as the flat vector; biases are not copied)
__global__ void k_transpose_params(NetDesc nd, int in0, const float* __restrict__ P, float* __restrict__ PT) {
    const int l = blockIdx.y;
    const int in = l == 0 ? in0 : nd.dims[l], out = nd.dims[l + 1];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= in * out) return;
    const int i = e % in, o = e / in;
    PT[nd.w_off[l] + e] = P[nd.w_off[l] + o + (size_t)i * out];
}

// Uout = u + h * sum_j coef[j] * k_j    over all D*B entries
__global__ void k_stage_combine(const float* __restrict__ u, StageK ks, float h, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int j = 0; j < ks.nk; ++j) acc = fmaf(ks.coef[j], ks.k[j][i], acc);
    out[i] = fmaf(h, acc, u[i]);
}

// lam_z += sum_i w_i
__global__ void k_lambda_update(float* __restrict__ lam, StageK ws, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = lam[i];
    for (int j = 0; j < ws.nk; ++j) acc += ws.k[j][i];
    lam[i] = acc;
}

// d loss / d z(t1) for loss = mean_b(-logpx + l1 E + l2 n + l3 A): z/B (+ l3 unit(z_aug)/B on the
// augmented rows)    src/icnf.jl:481-490, src/base_icnf.jl:167-189
__global__ void k_final_cotangent(NetDesc nd, float lambda3, const float* __restrict__ fsol,
                                  float* __restrict__ lam, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n_in = nd.n_in, D = n_in + 3;
    const float* c = fsol + (size_t)b * D;
    const float inv = 1.0f / (float)B;
    float sa = 0.f;
    const bool aug = nd.norm_z_aug && nd.naugs > 0;
    if (aug) for (int i = nd.nvars; i < n_in; ++i) sa = fmaf(c[i], c[i], sa);
    const float nrm = sqrtf(sa);
    for (int i = 0; i < n_in; ++i) {
        float v = c[i];
        if (aug && i >= nd.nvars && nrm > 0.f) v = fmaf(lambda3, c[i] / nrm, v);
        lam[(size_t)b * n_in + i] = v * inv;
    }
}

// ---- launchers ---------------------------------------------------------------------------------
GradLayout grad_layout(const NetDesc& nd) {
    GradLayout g{};
    g.in0 = nd.n_in + nd.n_cond;
    int io = 0, oo = 0, mx = g.in0;
    for (int l = 0; l < nd.n_layers; ++l) {
        g.in_off[l] = io;
        io += l == 0 ? g.in0 : nd.dims[l];
        g.out_off[l] = oo;
        oo += nd.dims[l + 1];
        if (nd.dims[l + 1] > mx) mx = nd.dims[l + 1];
    }
    g.sum_in = io; g.sum_out = oo; g.out_last = nd.dims[nd.n_layers]; g.max_dim = mx;
    return g;
}

int grad_adj_threads(const GradLayout& g) { return (g.max_dim + 63) & ~63; }

size_t grad_adj_lds_bytes(const NetDesc& nd, const GradLayout& g, int ts) {
    const AdjLds L = AdjLds::make(g, nd.n_in);
    const int nw = grad_adj_threads(g) / 64;
    return ((size_t)ts * L.per_sample + (size_t)nw * ts) * sizeof(float);
}

bool grad_supported(const NetDesc& nd, const GradLayout& g) {
    return grad_adj_threads(g) <= 1024 && grad_adj_lds_bytes(nd, g, 1) <= 160 * 1024;
}

template <int TS>
static hipError_t launch_adj_ts(const NetDesc& nd, const GradLayout& g, const AdjArgs& a, hipStream_t s) {
    const size_t lds = grad_adj_lds_bytes(nd, g, TS);
    hipError_t e = hipFuncSetAttribute((const void*)k_adj<TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_adj<TS>, dim3((a.B + TS - 1) / TS), dim3(grad_adj_threads(g)), lds, s, nd, g, a);
    return hipGetLastError();
}

hipError_t launch_adj(const NetDesc& nd, const GradLayout& g, const AdjArgs& a, hipStream_t s) {
    // 4 samples per workgroup amortise the weight reads once there are enough workgroups to fill
    // the chip; small batches keep one sample per workgroup for parallelism
    if (a.B >= 2048 && grad_adj_lds_bytes(nd, g, 4) <= 64 * 1024) return launch_adj_ts<4>(nd, g, a, s);
    if (a.B >= 512 && grad_adj_lds_bytes(nd, g, 2) <= 64 * 1024) return launch_adj_ts<2>(nd, g, a, s);
    return launch_adj_ts<1>(nd, g, a, s);
}

int grad_wgrad_tiles(const NetDesc& nd, const GradLayout& g) {
    int n = 0;
    for (int l = 0; l < nd.n_layers; ++l) {
        const int in = l == 0 ? g.in0 : nd.dims[l], out = nd.dims[l + 1];
        n += ((out + WG_T - 1) / WG_T) * ((in + WG_T - 1) / WG_T);
    }
    return n;
}

static bool wgrad_wave_ok(const NetDesc& nd, const GradLayout& g);
static bool wgrad_lds_form() { static const bool v = getenv("CNF_WGRAD_LDS") != nullptr; return v; }   // A/B: k_wgrad_mfma_b
void grad_ksplit(const NetDesc& nd, const GradLayout& g, int B, int* ksplit, int* chunk) {
    const int tiles = grad_wgrad_tiles(nd, g);
    int ks = (1024 + tiles - 1) / tiles;                 // aim at ~1024 workgroups: one wave per SIMD of k_wgrad_wave ...
    const int maxks = (B + 63) / 64;                     // at least 64 samples per split
    if (ks > maxks) ks = maxks;
    const int cap = (wgrad_wave_ok(nd, g) && !wgrad_lds_form()) ? GRAD_MAX_KSPLIT : 64;   // ... or two 256-thread workgroups per CU
    if (ks > cap) ks = cap;
    { static const int force = [] { const char* e = getenv("CNF_WGRAD_KS"); return e ? atoi(e) : 0; }(); if (force > 0 && force <= GRAD_MAX_KSPLIT) ks = force; }
    if (ks < 1) ks = 1;
    int ch = (B + ks - 1) / ks;
    ch = (ch + WB_K - 1) / WB_K * WB_K;                  // whole 32-sample groups (a multiple of the fp32 kernels' 16 too)
    ks = (B + ch - 1) / ch;
    *ksplit = ks; *chunk = ch;
}

hipError_t launch_wgrad(const NetDesc& nd, const GradLayout& g, const float* AB, const float* PB, const float* HS,
                        const float* TS, float* gpart, int n_params, int B, int ksplit, int chunk, hipStream_t s) {
    static const bool valu = getenv("CNF_WGRAD_VALU") != nullptr;     // A/B switches; default: split-bf16 MFMA
    static const bool fp32 = getenv("CNF_WGRAD_FP32") != nullptr;
    if (!valu && !fp32) {
        // 64 x 64 output tiles.  (128 x 128 -- every factor column read by one workgroup per K-split, 1.11x instead of
        // 1.56x the minimum traffic -- was measured: 96 KB of LDS leave one workgroup per CU and the launch 1.5 rounds of
        // them; the gradient went from 4.76 to 5.59 ms.)
        constexpr int TB = 64;
        constexpr size_t shm = (size_t)2 * 3 * TB * 64 + 4 * TB * sizeof(float);
        int tiles = 0;
        for (int l = 0; l < nd.n_layers; ++l) {
            const int in = l == 0 ? g.in0 : nd.dims[l], out = nd.dims[l + 1];
            tiles += ((out + TB - 1) / TB) * ((in + TB - 1) / TB);
        }
        const bool lds_form = wgrad_lds_form();
        const bool aligned = (((uintptr_t)AB | (uintptr_t)PB | (uintptr_t)HS | (uintptr_t)TS) & 15) == 0;
        if (!lds_form && aligned && wgrad_wave_ok(nd, g))
            hipLaunchKernelGGL(k_wgrad_wave, dim3(tiles, ksplit), dim3(64), 0, s, nd, g, AB, PB, HS, TS, gpart, n_params, B, chunk);
        else
        hipLaunchKernelGGL(k_wgrad_mfma_b, dim3(tiles, ksplit), dim3(256), shm, s, nd, g, AB, PB, HS, TS, gpart, n_params,
                           B, chunk);
    }
    else if (valu)
        hipLaunchKernelGGL(k_wgrad, dim3(grad_wgrad_tiles(nd, g), ksplit), dim3(256), 0, s, nd, g, AB, PB, HS, TS, gpart,
                           n_params, B, chunk);
    else
        hipLaunchKernelGGL(k_wgrad_mfma, dim3(grad_wgrad_tiles(nd, g), ksplit), dim3(256), 0, s, nd, g, AB, PB, HS, TS,
                           gpart, n_params, B, chunk);
    return hipGetLastError();
}

hipError_t launch_grad_reduce(const float* gpart, float* grad, int n_params, int ksplit, hipStream_t s) {
    hipLaunchKernelGGL(k_grad_reduce, dim3((n_params + 63) / 64), dim3(256), 0, s, gpart, grad, n_params, ksplit);
    return hipGetLastError();
}

hipError_t launch_grad_finish(const float* gpart, float* grad, int n_params, int ksplit, const StepState* state, const float* sums5,
                              float l1, float l2, float l3, int train, float* loss_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_grad_finish, dim3((n_params + 63) / 64), dim3(256), 0, s, gpart, grad, n_params, ksplit, state, sums5, l1, l2, l3,
                       train, loss_dev);
    return hipGetLastError();
}

hipError_t launch_transpose_params(const NetDesc& nd, const float* P, float* PT, hipStream_t s) {
    const GradLayout g = grad_layout(nd);
    int mx = 0;
    for (int l = 0; l < nd.n_layers; ++l) {
        const int in = l == 0 ? g.in0 : nd.dims[l];
        if (in * nd.dims[l + 1] > mx) mx = in * nd.dims[l + 1];
    }
    hipLaunchKernelGGL(k_transpose_params, dim3((mx + 255) / 256, nd.n_layers), dim3(256), 0, s, nd, g.in0, P, PT);
    return hipGetLastError();
}

hipError_t launch_stage_combine(const float* u, const StageK& ks, float h, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, ks, h, out, n);
    return hipGetLastError();
}

hipError_t launch_lambda_update(float* lam, const StageK& ws, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_lambda_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lam, ws, n);
    return hipGetLastError();
}

hipError_t launch_final_cotangent(const NetDesc& nd, float lambda3, const float* fsol, float* lam, int B,
                                  hipStream_t s) {
    hipLaunchKernelGGL(k_final_cotangent, dim3((B + 255) / 256), dim3(256), 0, s, nd, lambda3, fsol, lam, B);
    return hipGetLastError();
}

// =================================================================================================
// MFMA pullback kernel.  Same algebra as k_adj; every sweep is a GEMM on v_mfma_f32_16x16x4_f32:
//   a workgroup = AM_WAVES waves owns 16 samples (one MFMA column tile); wave w takes the 16-row output
//   tiles w, w+AM_WAVES, ...; A = a 16x16 fragment of the padded row-major weight image (forward sweeps:
//   W, reverse sweeps: W^T), read from global/L2 as one b128 per lane; B = the samples' activations
//   in LDS, [sample][feature], one ds_read_b128 per lane; the accumulator (lane = sample, 4
//   consecutive rows) goes back to LDS as one ds_write_b128 -- the conventions of cnf_mfma.hip.
// VJP compute mode only (JVP handles run k_adj).
// =================================================================================================
static inline int pad16(int x) { return (x + 15) & ~15; }

AdjMfmaLayout adj_mfma_layout(const NetDesc& nd, const GradLayout& g) {
    AdjMfmaLayout m{};
    m.L = nd.n_layers;
    m.dp[0] = pad16(g.in0);
    for (int l = 1; l <= m.L; ++l) m.dp[l] = pad16(nd.dims[l]);
    int off = 0, oo = 0, mx = m.dp[0];
    for (int l = 0; l < m.L; ++l) {
        m.f_off[l] = off; off += m.dp[l + 1] * m.dp[l];
        m.r_off[l] = off; off += m.dp[l] * m.dp[l + 1];
        m.b_off[l] = off; off += m.dp[l + 1];
        m.o_off[l] = oo; oo += m.dp[l + 1];
        if (m.dp[l + 1] > mx) mx = m.dp[l + 1];
    }
    for (int l = 0; l < m.L; ++l) {
        m.ff_off[l] = off; off += m.dp[l + 1] * (m.dp[l] + AM_IMG_PAD);
        m.fr_off[l] = off; off += m.dp[l] * (m.dp[l + 1] + AM_IMG_PAD);
    }
    m.img_floats = off;
    m.sum_o = oo; m.maxd = mx; m.nin_p = pad16(nd.n_in);
    int p = 0;
    m.D1 = p; p += oo;
    m.D2 = p; p += oo;
    m.TB = p; p += oo;
    m.S0 = p; p += mx;
    m.S1 = p; p += mx;
    m.E = p; p += m.nin_p;
    m.AH = m.TB + m.o_off[m.L - 1];       // ahat reuses the tbar_L slot (free after the first reverse GEMM)
    m.PS = ((p + 15) & ~15) + 8;          // stride = 8 mod 16 floats: conflict-free b128 columns
    m.SR = 3 * oo + 2 * m.nin_p + 16;     // sigma', q, tbar | zdot | |zdot|^2 (padded to a 64-byte row) | omega (JVP compute mode)
    m.vec4 = (g.sum_in & 3) == 0;         // rows of HS/TS start 16-byte aligned ...
    for (int l = 0; l < m.L; ++l) if (g.in_off[l] & 3) m.vec4 = 0;   // ... and so does every layer's block
    m.vec4o = (g.sum_out & 3) == 0;
    for (int l = 0; l < m.L; ++l) if (g.out_off[l] & 3) m.vec4o = 0;
    return m;
}

static size_t adj_mfma_lds_bytes(const AdjMfmaLayout& m) {
    return ((size_t)AM_NS * m.PS + (size_t)AM_EC * AM_NS) * sizeof(float);
}

bool adj_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m) {
    return nd.dims[nd.n_layers] == nd.n_in && adj_mfma_lds_bytes(m) <= 160 * 1024;      // (both compute modes: k_adj_mfma<.., JM>)
}

__global__ void k_pack_adj_images(NetDesc nd, GradLayout gl, AdjMfmaLayout m, const float* __restrict__ P,
                                  float* __restrict__ img) {
    const int l = blockIdx.y;
    const int in = l == 0 ? gl.in0 : nd.dims[l], out = nd.dims[l + 1];
    const int inp = m.dp[l], outp = m.dp[l + 1];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < outp * inp) {
        // (row r, column k) of a [rows][kp] image in fragment order: tile r / 16, k-block k / 16, lane 16 (k % 16 / 4) + r % 16
#ifdef AM_ROWPAD
        auto frag = [](int r, int k, int kp) { return r * (kp + AM_IMG_PAD) + k; };
#else
        auto frag = [](int r, int k, int kp) { return (((r >> 4) * (kp >> 4) + (k >> 4)) * 64 + 16 * ((k & 15) >> 2) + (r & 15)) * 4 + (k & 3); };
#endif
        {   // forward image [o][k]
            const int o = e / inp, k = e % inp;
            const float v = (o < out && k < in) ? P[nd.w_off[l] + o + (size_t)k * out] : 0.f;
            img[m.f_off[l] + e] = v;
            img[m.ff_off[l] + frag(o, k, inp)] = v;
        }
        {   // reverse image [i][o]
            const int i = e / outp, o = e % outp;
            const float v = (o < out && i < in) ? P[nd.w_off[l] + o + (size_t)i * out] : 0.f;
            img[m.r_off[l] + e] = v;
            img[m.fr_off[l] + frag(i, o, outp)] = v;
        }
    }
    if (e < outp) img[m.b_off[l] + e] = e < out ? P[nd.b_off[l] + e] : 0.f;
}

// kbar_z of four rows of one sample: b_i lambda + sum over the later stages of a_{m,i} zbar_m.  All five zbar slots are read (the
// unused ones point at a readable array with weight 0) and the four rows are requested before the first is used.
__device__ __forceinline__ void am_kbar4(const AdjArgs& a, bool ev, int eb, int n_in, int rb, float (&kb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rb + i * AM_EC;
        kb[i] = 0.f;
        if (ev && r < n_in) {
            const size_t at = (size_t)eb * n_in + r;
            float k = a.cb * a.lam[at];
#pragma unroll
            for (int w = 0; w < 5; ++w) k = fmaf(a.wc[w], a.w[w][at], k);
            kb[i] = k;
        }
    }
}

#ifdef AM_STAMPS
__device__ unsigned long long am_stamps[64];
#define AM_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) am_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cnf_debug_adj_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(am_stamps), sizeof(unsigned long long) * (n < 64 ? n : 64));
}
#else
#define AM_STAMP(i)
#endif
// PHASE 0: the whole pullback of the stages S.first .. S.last of one step, one after the other (a workgroup = 16 samples).
// Three of its four sweeps do not depend on the adjoint state at all -- the forward sweep, the tbar chain (omega = eps) and the
// tangent chain (tau is made of eps, eJ and the constant cotangents of the scalar rows): only ahat = kbar_z + c_E zdot/|zdot|
// and the hbar chain behind it carry lambda and the zbar of the later stages.  So when the batch leaves CUs idle
// (launch_adj_mfma_step decides), a step runs as TWO launches:
//   PHASE 1, grid (tiles, stages): sweeps 1-3 of ALL stages side by side; HS, PB, TS filed as before; sigma', q = sigma'' .* p,
//            tbar and zdot of every sample parked in the scratch rows SC [stage][B][m.SR];
//   PHASE 2, grid (tiles): per stage the scratch rows back into LDS, abar_L = ahat sigma'_L + eps q_L, the hbar chain (AB, zbar),
//            then lambda <- lambda + sum zbar -- a quarter of the sequential work.
// The two-launch form takes a RUN of steps: their AdjStepArgs in a device array (last step first), scratch slot 6 j + stage for
// step j; PHASE 1 grid (tiles, 6 steps), PHASE 2 grid (tiles) walks the steps in order (k_adj_mfma_run below).
// JM: the JVP compute mode (src/icnf.jl:384-456; ldot = -eps . J eps, ndot = |J eps|): sweep 2 is the TANGENT chain from
// tau_0 = eps (q_l = s''_l .* (W_l tau_{l-1}), tau_l = s'_l .* (W_l tau_{l-1})), then omega = -c_l eps + c_n J eps / |J eps| takes
// eps's place in E and sweep 3 is the tbar chain from it, layers L .. 2 (tbar_0 is nobody's); the hbar chain and the parked rows are
// those of the VJP mode with tbar_L = omega where that has eps.  Factor rows: HS = h_{l-1}, TS = tau_{l-1}, AB = abar_l,
// PB = s'_l .* tbar_l, so the contraction kernels do not know the difference.
template <bool ALL_TANH, int PHASE, bool JM = false>
__device__ __forceinline__ void adj_mfma_body(const NetDesc& nd, const GradLayout& gl, const AdjMfmaLayout& m, const float* __restrict__ img,
                                              const AdjStepArgs& S, float* __restrict__ SC, int ystage) {
    extern __shared__ float lds[];
    const int PS = m.PS, NL = m.L;
    float* red = lds + (size_t)AM_NS * PS;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * AM_NS;
    const int n_in = nd.n_in, D = n_in + 3, in0 = gl.in0;
    // elementwise passes: sample es, features ec, ec + AM_EC, ... (16 consecutive lanes = 64 contiguous bytes)
    const int es = (tid >> 4) & 15, ec = (tid & 15) | ((tid >> 8) << 4);
    const int eb = b0 + es;
    const bool ev = eb < S.B;
    const int oL = m.o_off[NL - 1];

  const int stg_hi = PHASE == 1 ? S.first - ystage : S.first, stg_lo = PHASE == 1 ? stg_hi : S.last;
  for (int stg = stg_hi; stg >= stg_lo; --stg) {       // the stages of one Runge-Kutta step, last to first
    const AdjArgs& a = S.st[stg];
    float* scrow = SC + ((size_t)stg * S.B + eb) * m.SR;  // PHASE 1 / 2: this thread's sample in the scratch rows (if ev)
    AM_STAMP(0);
    AFrag pf;
    int cur = m.S0, nxt = m.S1;
   if (PHASE != 2) {
    am_first(pf, img + m.ff_off[0], m.dp[1], m.dp[0]);
    // ---- inputs: [z; ys; 0] -> S0, eps -> E; h_0 also goes out for the weight gradient ---------
    for (int r = ec; r < m.dp[0]; r += AM_EC) {
        float v = 0.f;
        if (ev && r < in0) v = r < n_in ? a.ustage[(size_t)eb * D + r] : a.ys[(size_t)eb * nd.n_cond + (r - n_in)];
        lds[es * PS + m.S0 + r] = v;
        if (ev && r < in0) a.HS[(size_t)eb * gl.sum_in + r] = v;
    }
    for (int r = ec; r < m.nin_p; r += AM_EC) lds[es * PS + m.E + r] = (ev && r < n_in) ? a.eps[(size_t)eb * n_in + r] : 0.f;
    am_barrier();
    AM_STAMP(1);

   if (JM) {
    // ======== JVP compute mode ========
    // ---- sweep 1: forward ----
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], act = nd.acts[l];
        const int oo = m.o_off[l];
        const bool last = l + 1 == NL;
        const int hs_off = last ? -1 : gl.in_off[l + 1];
        am_gemm(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PS, pf,
                img + (last ? m.ff_off[0] : m.ff_off[l + 1]), last ? m.dp[1] : m.dp[l + 2], last ? m.dp[0] : m.dp[l + 1], img + m.b_off[l],
                [&](int r0, int s, f32x4 acc, f32x4 bias) {
            f32x4 h, d1, d2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float hh, dd1, dd2;
                if (ALL_TANH) { hh = cnf_tanh(acc[j] + bias[j]); dd1 = fmaf(-hh, hh, 1.0f); dd2 = -2.0f * hh * dd1; }
                else cnf_act2(act, acc[j] + bias[j], hh, dd1, dd2);
                const bool live = r0 + j < out;            // padded rows stay exactly zero
                h[j] = live ? hh : 0.f; d1[j] = live ? dd1 : 0.f; d2[j] = live ? dd2 : 0.f;
            }
            float* S = lds + s * PS;
            *reinterpret_cast<f32x4*>(S + nxt + r0) = h;
            *reinterpret_cast<f32x4*>(S + m.D1 + oo + r0) = d1;
            *reinterpret_cast<f32x4*>(S + m.D2 + oo + r0) = d2;
            if (!last && b0 + s < a.B) am_store4(a.HS + (size_t)(b0 + s) * gl.sum_in + hs_off + r0, h, r0, out, m.vec4);
        });
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    {   // zdot sits in S[cur]: ahat = kbar_z + c_E zdot/|zdot| -> AH (PHASE 1: zdot and |zdot|^2 -> scratch instead)
        const int zd = cur;
        float nz = 0.f;
        if (nd.norm_z) nz = am_colnorm2(lds + zd, PS, n_in, red);
        if (PHASE == 1) {
            for (int r = ec; r < m.nin_p; r += AM_EC) if (ev) scrow[3 * m.sum_o + r] = lds[es * PS + zd + r];
            if (ev && ec == 0) scrow[3 * m.sum_o + m.nin_p] = nz;
        } else {
            const float inv = (nd.norm_z && nz > 0.f) ? a.c_E * __builtin_amdgcn_rsqf(nz) : 0.f;
            for (int rb = ec; rb < m.nin_p; rb += 4 * AM_EC) {
                float kb[4];
                am_kbar4(a, ev, eb, n_in, rb, kb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rb + i * AM_EC;
                    if (r < m.nin_p) lds[es * PS + m.AH + r] = (ev && r < n_in) ? fmaf(inv, lds[es * PS + zd + r], kb[i] * a.hstep) : 0.f;
                }
            }
        }
        // tau_0 = [eps; 0] -> the other buffer, TS
        for (int r = ec; r < m.dp[0]; r += AM_EC) {
            const float v = r < n_in ? lds[es * PS + m.E + r] : 0.f;
            lds[es * PS + nxt + r] = v;
            if (ev && r < in0) a.TS[(size_t)eb * gl.sum_in + r] = v;
        }
        am_barrier();                                      // (zdot's buffer is the tangent sweep's first target)
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    // ---- sweep 2: the tangent chain ----
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], oo = m.o_off[l];
        const bool last = l + 1 == NL;
        const int ts_off = last ? -1 : gl.in_off[l + 1];
        am_gemm(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PS, pf,
                img + (last ? m.fr_off[NL - 1] : m.ff_off[l + 1]), last ? m.dp[NL - 1] : m.dp[l + 2], last ? m.dp[NL] : m.dp[l + 1], nullptr,
                [&](int r0, int s, f32x4 acc, f32x4) {
            float* S = lds + s * PS;
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(S + m.D1 + oo + r0);
            *reinterpret_cast<f32x4*>(S + m.D2 + oo + r0) = *reinterpret_cast<const f32x4*>(S + m.D2 + oo + r0) * acc;   // q_l
            const f32x4 t = d1 * acc;
            *reinterpret_cast<f32x4*>(S + nxt + r0) = t;
            if (!last && b0 + s < a.B) am_store4(a.TS + (size_t)(b0 + s) * gl.sum_in + ts_off + r0, t, r0, out, m.vec4);
        });
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    {   // J eps in S[cur]: omega = -c_l eps + c_n J eps / |J eps| -> E (over eps); pbar_L = s'_L .* omega -> S[nxt], PB
        float nj = 0.f;
        if (nd.norm_j) nj = am_colnorm2(lds + cur, PS, n_in, red);
        const float inv = (nd.norm_j && nj > 0.f) ? a.c_n * __builtin_amdgcn_rsqf(nj) : 0.f;
        for (int r = ec; r < m.dp[NL]; r += AM_EC) {
            float* Sr = lds + es * PS;
            const float om = r < n_in ? fmaf(inv, Sr[cur + r], -a.c_l * Sr[m.E + r]) : 0.f;
            const float pb = om * Sr[m.D1 + oL + r];
            if (r < m.nin_p) Sr[m.E + r] = om;
            Sr[nxt + r] = pb;
            if (ev && r < n_in) a.PB[(size_t)eb * gl.sum_out + gl.out_off[NL - 1] + r] = pb;
        }
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    // ---- sweep 3: the tbar chain, layers L .. 2: layer l's product turns pbar_l into tbar_{l-1} ----
    for (int l = NL - 1; l >= 1; --l) {
        const int oprev = m.o_off[l - 1], outp = nd.dims[l], gprev = gl.out_off[l - 1];
        am_gemm(img + m.fr_off[l], m.dp[l], m.dp[l + 1], lds + cur, PS, pf,
                img + (l > 1 ? m.fr_off[l - 1] : m.fr_off[NL - 1]), l > 1 ? m.dp[l - 1] : m.dp[NL - 1], l > 1 ? m.dp[l] : m.dp[NL], nullptr,
                [&](int r0, int s, f32x4 acc, f32x4) {
            float* S = lds + s * PS;
            *reinterpret_cast<f32x4*>(S + m.TB + oprev + r0) = acc;
            const f32x4 pb = acc * *reinterpret_cast<const f32x4*>(S + m.D1 + oprev + r0);
            *reinterpret_cast<f32x4*>(S + nxt + r0) = pb;
            if (b0 + s < a.B) am_store4(a.PB + (size_t)(b0 + s) * gl.sum_out + gprev + r0, pb, r0, outp, m.vec4o);
        });
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    if (PHASE == 1) {                                       // sigma', q, tbar and omega of this stage -> scratch
        for (int r = 4 * ec; r < 3 * m.sum_o; r += 4 * AM_EC)
            if (ev) *reinterpret_cast<f32x4*>(scrow + r) = *reinterpret_cast<const f32x4*>(lds + es * PS + m.D1 + r);
        for (int r = ec; r < m.nin_p; r += AM_EC) if (ev) scrow[3 * m.sum_o + m.nin_p + 16 + r] = lds[es * PS + m.E + r];
        break;
    }
    // abar_L = ahat s'_L + omega q_L -> S[nxt], AB
    for (int r = ec; r < m.dp[NL]; r += AM_EC) {
        float* Sr = lds + es * PS;
        const float ab = r < m.nin_p ? fmaf(Sr[m.AH + r], Sr[m.D1 + oL + r], Sr[m.E + r] * Sr[m.D2 + oL + r]) : 0.f;
        Sr[nxt + r] = ab;
        if (ev && r < n_in) a.AB[(size_t)eb * gl.sum_out + gl.out_off[NL - 1] + r] = ab;
    }
    am_barrier();
    { const int t_ = cur; cur = nxt; nxt = t_; }
   } else {
    // ---- sweep 1: forward.  The last layer's epilogue also forms pbar_L = eps .* sigma'_L (the first
    //      operand of the tbar chain, tbar_L = omega = eps), parked in the tbar_L slot of TB ------------
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], act = nd.acts[l];
        const int oo = m.o_off[l];
        const bool last = l + 1 == NL;
        const int hs_off = last ? -1 : gl.in_off[l + 1];
        am_gemm(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PS, pf,
                img + (last ? m.fr_off[NL - 1] : m.ff_off[l + 1]), last ? m.dp[NL - 1] : m.dp[l + 2],
                last ? m.dp[NL] : m.dp[l + 1], img + m.b_off[l],
                [&](int r0, int s, f32x4 acc, f32x4 bias) {
            f32x4 h, d1, d2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float hh, dd1, dd2;
                if (ALL_TANH) { hh = cnf_tanh(acc[j] + bias[j]); dd1 = fmaf(-hh, hh, 1.0f); dd2 = -2.0f * hh * dd1; }
                else cnf_act2(act, acc[j] + bias[j], hh, dd1, dd2);
                const bool live = r0 + j < out;            // padded rows stay exactly zero
                h[j] = live ? hh : 0.f; d1[j] = live ? dd1 : 0.f; d2[j] = live ? dd2 : 0.f;
            }
            float* S = lds + s * PS;
            *reinterpret_cast<f32x4*>(S + nxt + r0) = h;
            *reinterpret_cast<f32x4*>(S + m.D1 + oo + r0) = d1;
            *reinterpret_cast<f32x4*>(S + m.D2 + oo + r0) = d2;
            const bool sv = b0 + s < a.B;
            if (!last) {
                if (sv) am_store4(a.HS + (size_t)(b0 + s) * gl.sum_in + hs_off + r0, h, r0, out, m.vec4);
            } else {
                const f32x4 pb = *reinterpret_cast<const f32x4*>(S + m.E + r0) * d1;      // out_L == n_in
                *reinterpret_cast<f32x4*>(S + m.TB + oL + r0) = pb;
                if (sv) am_store4(a.PB + (size_t)(b0 + s) * gl.sum_out + gl.out_off[l] + r0, pb, r0, out, m.vec4o);
            }
        });
        am_barrier();
        AM_STAMP(2 + l);
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    const int zd = cur;                                     // zdot stays in S[zd] until AH is formed (inside sweep 2)

    // ---- sweep 2: tbar chain (omega = eps).  Layer l's GEMM turns pbar_l into tbar_{l-1}; its epilogue
    //      keeps tbar_{l-1} (sweep 4 needs it) and forms pbar_{l-1} = tbar_{l-1} .* sigma'_{l-1} -----------
    for (int l = NL - 1; l >= 0; --l) {
        const float* X = l == NL - 1 ? lds + m.TB + oL : lds + cur;
        const int oprev = l > 0 ? m.o_off[l - 1] : 0, outp = l > 0 ? nd.dims[l] : 0;
        const int gprev = l > 0 ? gl.out_off[l - 1] : 0;
        am_gemm(img + m.fr_off[l], m.dp[l], m.dp[l + 1], X, PS, pf,
                img + (l > 0 ? m.fr_off[l - 1] : m.ff_off[0]), l > 0 ? m.dp[l - 1] : m.dp[1], l > 0 ? m.dp[l] : m.dp[0],
                nullptr, [&](int r0, int s, f32x4 acc, f32x4) {
            float* S = lds + s * PS;
            if (l > 0) {
                *reinterpret_cast<f32x4*>(S + m.TB + oprev + r0) = acc;
                const f32x4 pb = acc * *reinterpret_cast<const f32x4*>(S + m.D1 + oprev + r0);
                *reinterpret_cast<f32x4*>(S + nxt + r0) = pb;
                if (b0 + s < a.B) am_store4(a.PB + (size_t)(b0 + s) * gl.sum_out + gprev + r0, pb, r0, outp, m.vec4o);
            } else {
                *reinterpret_cast<f32x4*>(S + nxt + r0) = acc;                 // tbar_0 = eJ
            }
        });
        am_barrier();
        AM_STAMP(7 + (NL - 1 - l));
        const int t_ = cur; cur = nxt; nxt = t_;
        if (l == NL - 1) {
            if (PHASE == 1) {                                // zdot and |zdot|^2 -> scratch (ahat is formed in PHASE 2)
                float nz = 0.f;
                if (nd.norm_z) nz = am_colnorm2(lds + zd, PS, n_in, red);
                for (int r = ec; r < m.nin_p; r += AM_EC) if (ev) scrow[3 * m.sum_o + r] = lds[es * PS + zd + r];
                if (ev && ec == 0) scrow[3 * m.sum_o + m.nin_p] = nz;
            } else {
            // zdot still sits in S[zd]: ahat = kbar_z + c_E zdot/|zdot| -> AH.  AH shares the tbar_L slot of TB,
            // free now that pbar_L (parked there by sweep 1) has been consumed by this GEMM.
            float nz = 0.f;
            if (nd.norm_z) nz = am_colnorm2(lds + zd, PS, n_in, red);
            const float inv = (nd.norm_z && nz > 0.f) ? a.c_E * __builtin_amdgcn_rsqf(nz) : 0.f;
            // (all five zbar slots are read -- the unused ones point at a readable array with weight 0 -- and four rows per thread
            // are requested before the first is used: one round trip to memory per four rows instead of up to six per row)
            for (int rb = ec; rb < m.nin_p; rb += 4 * AM_EC) {
                float kb[4];
                am_kbar4(a, ev, eb, n_in, rb, kb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rb + i * AM_EC;
                    if (r < m.nin_p) lds[es * PS + m.AH + r] = (ev && r < n_in) ? fmaf(inv, lds[es * PS + zd + r], kb[i] * a.hstep) : 0.f;
                }
            }
            }
            am_barrier();                                   // zdot's buffer is the next epilogue's target
        }
    }
    // eJ in S[cur].  tau = -c_l eps + c_n eJ/|eJ|  -> S[nxt] as t_0 (rows of ys and padding: 0)
    {
        float nj = 0.f;
        if (nd.norm_j) nj = am_colnorm2(lds + cur, PS, n_in, red);
        const float inv = (nd.norm_j && nj > 0.f) ? a.c_n * __builtin_amdgcn_rsqf(nj) : 0.f;
        for (int r = ec; r < m.dp[0]; r += AM_EC) {
            float v = 0.f;
            if (r < n_in) v = fmaf(inv, lds[es * PS + cur + r], -a.c_l * lds[es * PS + m.E + r]);
            lds[es * PS + nxt + r] = v;
            if (ev && r < in0) a.TS[(size_t)eb * gl.sum_in + r] = v;
        }
    }
    am_barrier();
    AM_STAMP(13);
    { const int t_ = cur; cur = nxt; nxt = t_; }

    // ---- sweep 3: tangent chain.  The last layer's epilogue forms abar_L = ahat sigma' + eps q_L instead
    //      of t_L (nobody reads t_L) -------------------------------------------------------------------------
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], oo = m.o_off[l];
        const bool last = l + 1 == NL;
        const int ts_off = last ? -1 : gl.in_off[l + 1];
        am_gemm(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PS, pf,
                img + (last ? m.fr_off[NL - 1] : m.ff_off[l + 1]), last ? m.dp[NL - 1] : m.dp[l + 2],
                last ? m.dp[NL] : m.dp[l + 1], nullptr,
                [&](int r0, int s, f32x4 acc, f32x4) {
            float* S = lds + s * PS;
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(S + m.D1 + oo + r0);
            const f32x4 q = *reinterpret_cast<const f32x4*>(S + m.D2 + oo + r0) * acc;   // q_l = sigma'' .* p_l
            *reinterpret_cast<f32x4*>(S + m.D2 + oo + r0) = q;
            const bool sv = b0 + s < a.B;
            if (!last) {
                const f32x4 t = d1 * acc;
                *reinterpret_cast<f32x4*>(S + nxt + r0) = t;
                if (sv) am_store4(a.TS + (size_t)(b0 + s) * gl.sum_in + ts_off + r0, t, r0, out, m.vec4);
            } else if (PHASE == 0) {
                const f32x4 ab = *reinterpret_cast<const f32x4*>(S + m.AH + r0) * d1 +
                                 *reinterpret_cast<const f32x4*>(S + m.E + r0) * q;
                *reinterpret_cast<f32x4*>(S + nxt + r0) = ab;
                if (sv) am_store4(a.AB + (size_t)(b0 + s) * gl.sum_out + gl.out_off[l] + r0, ab, r0, out, m.vec4o);
            }
        });
        am_barrier();
        AM_STAMP(14 + l);
        const int t_ = cur; cur = nxt; nxt = t_;
    }

    if (PHASE == 1) {                                       // sigma', q, tbar of this stage -> scratch; the stage is PHASE 2's from here
        for (int r = 4 * ec; r < 3 * m.sum_o; r += 4 * AM_EC)
            if (ev) *reinterpret_cast<f32x4*>(scrow + r) = *reinterpret_cast<const f32x4*>(lds + es * PS + m.D1 + r);
        break;
    }
   }
   } else {
    // ---- PHASE 2: the scratch rows back into LDS (sigma', q, tbar: one contiguous block of the sample row; zdot -> S0), eps -> E;
    //      abar_L = ahat sigma'_L + eps q_L -> S1 and AB, elementwise (ahat never leaves the registers) ----
    am_first(pf, img + m.fr_off[NL - 1], m.dp[NL - 1], m.dp[NL]);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // (kbar_z of this thread's first four rows and |zdot|^2 are requested before anything waits: their trip to memory runs under
    // the copy of the scratch rows)
    float kb0[4];
    am_kbar4(a, ev, eb, n_in, ec, kb0);
    const float nz = (nd.norm_z && ev) ? scrow[3 * m.sum_o + m.nin_p] : 0.f;
    for (int r = 4 * ec; r < 3 * m.sum_o; r += 4 * AM_EC)
        *reinterpret_cast<f32x4*>(lds + es * PS + m.D1 + r) = ev ? *reinterpret_cast<const f32x4*>(scrow + r) : z4;
    for (int r = ec; r < m.nin_p; r += AM_EC) {
        lds[es * PS + m.S0 + r] = ev ? scrow[3 * m.sum_o + r] : 0.f;
        lds[es * PS + m.E + r] = JM ? (ev ? scrow[3 * m.sum_o + m.nin_p + 16 + r] : 0.f)        // (omega, parked)
                                    : ((ev && r < n_in) ? a.eps[(size_t)eb * n_in + r] : 0.f);
    }
    am_barrier();
    const float inv = (nd.norm_z && nz > 0.f) ? a.c_E * __builtin_amdgcn_rsqf(nz) : 0.f;
    for (int rb = ec; rb < m.nin_p; rb += 4 * AM_EC) {
        float kb[4];
        if (rb == ec) { kb[0] = kb0[0]; kb[1] = kb0[1]; kb[2] = kb0[2]; kb[3] = kb0[3]; }
        else am_kbar4(a, ev, eb, n_in, rb, kb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb + i * AM_EC;
            if (r < m.nin_p) {
                const float* Sr = lds + es * PS;
                const float ah = (ev && r < n_in) ? fmaf(inv, Sr[m.S0 + r], kb[i] * a.hstep) : 0.f;
                const float ab = fmaf(ah, Sr[m.D1 + oL + r], Sr[m.E + r] * Sr[m.D2 + oL + r]);
                lds[es * PS + m.S1 + r] = ab;
                if (ev && r < n_in) a.AB[(size_t)eb * gl.sum_out + gl.out_off[NL - 1] + r] = ab;
            }
        }
    }
    am_barrier();
    cur = m.S1; nxt = m.S0;
   }

    // ---- sweep 4: hbar chain.  Layer l's GEMM turns abar_l into hbar_{l-1}; its epilogue forms
    //      abar_{l-1} = hbar_{l-1} sigma' + tbar_{l-1} q_{l-1}; the last one is zbar -------------------------
    for (int l = NL - 1; l >= 0; --l) {
        const int oprev = l > 0 ? m.o_off[l - 1] : 0, outp = l > 0 ? nd.dims[l] : 0;
        const int gprev = l > 0 ? gl.out_off[l - 1] : 0;
        am_gemm(img + m.fr_off[l], m.dp[l], m.dp[l + 1], lds + cur, PS, pf,
                l > 0 ? img + m.fr_off[l - 1] : nullptr, l > 0 ? m.dp[l - 1] : 0, l > 0 ? m.dp[l] : 0,
                nullptr, [&](int r0, int s, f32x4 acc, f32x4) {
            float* S = lds + s * PS;
            const bool sv = b0 + s < a.B;
            if (l > 0) {
                const f32x4 ab = acc * *reinterpret_cast<const f32x4*>(S + m.D1 + oprev + r0) +
                                 *reinterpret_cast<const f32x4*>(S + m.TB + oprev + r0) *
                                 *reinterpret_cast<const f32x4*>(S + m.D2 + oprev + r0);
                *reinterpret_cast<f32x4*>(S + nxt + r0) = ab;
                if (sv) am_store4(a.AB + (size_t)(b0 + s) * gl.sum_out + gprev + r0, ab, r0, outp, m.vec4o);
            } else if (sv) {
                am_store4(a.w_out + (size_t)(b0 + s) * n_in + r0, acc, r0, n_in, (n_in & 3) == 0);   // zbar
            }
        });
        if (l > 0) am_barrier();
        AM_STAMP(18 + (NL - 1 - l));
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    AM_STAMP(24);
    // zbar of this stage is input to the earlier stages (and to the lambda update): stores first, then everyone
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    am_barrier();
  }
  if (PHASE != 1 && S.lam_update) {        // lambda <- lambda + sum over the stages of zbar   (rows of this workgroup's samples)
    for (int r = ec; r < n_in; r += AM_EC) {
        if (ev) {
            float acc = S.st[0].lam[(size_t)eb * n_in + r];
            for (int k = S.last; k <= S.first; ++k) acc += S.st[k].w_out[(size_t)eb * n_in + r];
            S.lam_out[(size_t)eb * n_in + r] = acc;
        }
    }
    if (PHASE == 2) {                      // (a run of steps: the next step's kbar_z reads this lambda)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        am_barrier();
    }
  }
}

template <bool ALL_TANH, bool JM>
__global__ void __launch_bounds__(AM_THREADS)
k_adj_mfma(NetDesc nd, GradLayout gl, AdjMfmaLayout m, const float* __restrict__ img, AdjStepArgs S) {
    adj_mfma_body<ALL_TANH, 0, JM>(nd, gl, m, img, S, nullptr, 0);
}

// (one step in two launches with its arguments by value: 5 % faster than through the device array -- 128 + 124 against
// 137 + 129 us per step at config 5, B = 2048 -- so a sub-run of ONE step takes this form)
template <bool ALL_TANH, int PHASE, bool JM>
__global__ void __launch_bounds__(AM_THREADS)
k_adj_mfma_split(NetDesc nd, GradLayout gl, AdjMfmaLayout m, const float* __restrict__ img, AdjStepArgs S, float* __restrict__ SC) {
    adj_mfma_body<ALL_TANH, PHASE, JM>(nd, gl, m, img, S, SC, blockIdx.y);
}

template <bool ALL_TANH, int PHASE, bool JM>
__global__ void __launch_bounds__(AM_THREADS)
k_adj_mfma_run(NetDesc nd, GradLayout gl, AdjMfmaLayout m, const float* __restrict__ img, const AdjStepArgs* __restrict__ SA, int nsteps,
               float* __restrict__ SC) {
    if (PHASE == 1) {
        const int j = blockIdx.y / 6;
        adj_mfma_body<ALL_TANH, 1, JM>(nd, gl, m, img, SA[j], SC + (size_t)6 * j * SA[j].B * m.SR, blockIdx.y % 6);
    } else {
        for (int j = 0; j < nsteps; ++j) adj_mfma_body<ALL_TANH, 2, JM>(nd, gl, m, img, SA[j], SC + (size_t)6 * j * SA[j].B * m.SR, 0);
    }
}

// ---------------------------------------------------------------------------------------------------
// The same pullback for the headline shape 32-128-128-32 (BASELINE configs 3/4): 32 samples per workgroup (two
// MFMA column tiles that share every A fragment) and every weight fragment resident in registers for all six
// stages of the launch -- wave w keeps its 16-row tile of W1, W2, W3^T and W2^T, waves 0..3 also a 16-row tile of
// W3 and W1^T (out tile w & 1, column tile w >> 1: the 32-row sweeps run on one wave per SIMD).  A sweep is then
// B operand from LDS -> MFMAs -> epilogue -> LDS with no weight traffic at all; k_adj_mfma re-streams 200 KB of
// fragments from L2 per stage and 16 samples, and waits for them in each of its 12 sweeps.  Epilogues, LDS rows
// and factor arrays are those of k_adj_mfma.
// ---------------------------------------------------------------------------------------------------
#define A3_NS 32
#ifdef A3_ABL_NOMFMA      // ablation: no MFMAs, operands kept alive
__device__ __forceinline__ f32x4 a3_mfma(float a, float b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
__device__ __forceinline__ f32x4 a3_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif
// two column tiles against one A tile; B operands run two k-blocks ahead of the MFMAs (ring of 3)
template <int KB>
__device__ __forceinline__ void adj3_tile2(const f32x4 (&A)[KB], const float* X, int PS, int s, int q, const f32x4& init,
                                           f32x4& out0, f32x4& out1) {
    const float* x0 = X + s * PS + 4 * q;
    const float* x1 = x0 + 16 * PS;
    f32x4 b0[2], b1[2];                    // B operands one k-block (8 MFMAs) ahead
    b0[0] = *reinterpret_cast<const f32x4*>(x0);
    b1[0] = *reinterpret_cast<const f32x4*>(x1);
    f32x4 a00 = init, a10 = init, a01 = {0.f, 0.f, 0.f, 0.f}, a11 = a01;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
        if (u + 1 < KB) {
            b0[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(x0 + 16 * (u + 1));
            b1[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(x1 + 16 * (u + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (u & 1) {
                a01 = a3_mfma(A[u][k], b0[u & 1][k], a01);
                a11 = a3_mfma(A[u][k], b1[u & 1][k], a11);
            } else {
                a00 = a3_mfma(A[u][k], b0[u & 1][k], a00);
                a10 = a3_mfma(A[u][k], b1[u & 1][k], a10);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out0 = a00 + a01; out1 = a10 + a11;
}
// one column tile (rows of X start at the tile's first sample)
template <int KB>
__device__ __forceinline__ f32x4 adj3_tile1(const f32x4 (&A)[KB], const float* X, int PS, int s, int q, const f32x4& init) {
    const float* x0 = X + s * PS + 4 * q;
    f32x4 b[3];
#pragma unroll
    for (int u = 0; u < 2 && u < KB; ++u) b[u] = *reinterpret_cast<const f32x4*>(x0 + 16 * u);
    f32x4 a0 = init, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KB; ++u) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {       // two chains, alternating: a dependent MFMA would wait for its predecessor
            if (k & 1) a1 = a3_mfma(A[u][k], b[u % 3][k], a1);
            else a0 = a3_mfma(A[u][k], b[u % 3][k], a0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < KB) b[(u + 2) % 3] = *reinterpret_cast<const f32x4*>(x0 + 16 * (u + 2));
    }
    return a0 + a1;
}
// The state row (or conditioning input) of one (sample, row) element of a stage: the only thing a stage reads from global
// memory (eps and lambda are the same for every stage of the launch, the zbar of the later stages never leave the CU).
__device__ __forceinline__ float adj3_fetch(const AdjArgs& a, const NetDesc& nd, int eb, bool ev, int r, int n_in, int in0, int D) {
    // both base pointers as scalars first: a per-lane choice between two pointer FIELDS makes the compiler fetch the
    // chosen field with a vector load and wait for it (and for every request in front of it)
    const float* pu = a.ustage;
    const float* py = a.ys ? a.ys : a.ustage;
    asm volatile("" ::"s"(pu), "s"(py));
    float x = 0.f;
    if (ev && r < in0) x = *(r < n_in ? pu + (size_t)eb * D + r : py + (size_t)eb * nd.n_cond + (r - n_in));
    return x;
}

static size_t adj3_lds_bytes(const AdjMfmaLayout& m) { return ((size_t)A3_NS * m.PS + (size_t)AM_EC * A3_NS) * sizeof(float); }

template <bool ALL_TANH>
__global__ void __launch_bounds__(AM_THREADS)
k_adj3(NetDesc nd, GradLayout gl, AdjMfmaLayout m, const float* __restrict__ img, AdjStepArgs S) {
    extern __shared__ float lds[];
    constexpr int NL = 3, H = 128, NI = 32;               // dp = {32, 128, 128, 32}
    static_assert(AM_WAVES == 8, "one 16-row tile of the 128-wide layers per wave");
    const int PS = m.PS;
    float* red = lds + (size_t)A3_NS * PS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * A3_NS;
    const int n_in = nd.n_in, D = n_in + 3, in0 = gl.in0;
    const int es = (tid >> 4) & 15, ec = (tid & 15) | ((tid >> 8) << 4);
    const int oL = m.o_off[NL - 1];
    const bool narrow = wave < 4;                          // 32-row sweeps: out tile nt, column tile nc, one wave per SIMD
    const int nt = wave & 1, nc = (wave >> 1) & 1;

    // resident fragments: A operand (rows 16 tile + s, k = 16u + 4q ..) of each image
    // (the two tiles of the 32-row sweeps, W3 and W1^T, share ONE register buffer `na`: each is fetched from L2 a whole
    // sweep before its use -- keeping both resident as well would spill)
    f32x4 f1[NI / 16], f2[H / 16], r3[NI / 16], r2[H / 16], na[H / 16], bias1, bias2, bias3;
    const float* F3 = img + m.f_off[2] + (size_t)(16 * nt + s) * H + 4 * q;             // W3   [32][128]
    const float* R1 = img + m.r_off[0] + (size_t)(16 * nt + s) * H + 4 * q;             // W1^T [32][128]
    auto fetch_na = [&](const float* base) {
        if (narrow) {
#pragma unroll
            for (int u = 0; u < H / 16; ++u) na[u] = *reinterpret_cast<const f32x4*>(base + 16 * u);
        }
    };
    {
        const float* F1 = img + m.f_off[0] + (size_t)(16 * wave + s) * NI + 4 * q;      // W1   [128][32]
        const float* F2 = img + m.f_off[1] + (size_t)(16 * wave + s) * H + 4 * q;       // W2   [128][128]
        const float* R3 = img + m.r_off[2] + (size_t)(16 * wave + s) * NI + 4 * q;      // W3^T [128][32]
        const float* R2 = img + m.r_off[1] + (size_t)(16 * wave + s) * H + 4 * q;       // W2^T [128][128]
#pragma unroll
        for (int u = 0; u < NI / 16; ++u) { f1[u] = *reinterpret_cast<const f32x4*>(F1 + 16 * u); r3[u] = *reinterpret_cast<const f32x4*>(R3 + 16 * u); }
#pragma unroll
        for (int u = 0; u < H / 16; ++u) {
            f2[u] = *reinterpret_cast<const f32x4*>(F2 + 16 * u); r2[u] = *reinterpret_cast<const f32x4*>(R2 + 16 * u);
        }
        bias1 = *reinterpret_cast<const f32x4*>(img + m.b_off[0] + 16 * wave + 4 * q);
        bias2 = *reinterpret_cast<const f32x4*>(img + m.b_off[1] + 16 * wave + 4 * q);
        bias3 = *reinterpret_cast<const f32x4*>(img + m.b_off[2] + 16 * nt + 4 * q);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int r0w = 16 * wave + 4 * q, r0n = 16 * nt + 4 * q;       // first row of this lane in a wide / narrow tile

    // Per thread: row ec of samples es and es + 16 in the elementwise passes.  lambda and eps are the same for every stage
    // of the launch (registers / the E rows of LDS, written once); the zbar of a finished stage goes through LDS to these
    // threads, which keep  K[d] = sum_{m done} a_{m, j} zbar_m  for the d-th stage still to come (a shift register: the
    // stages run 5, 4, .., 0) and the running sum of all zbar for the lambda update -- nothing of it touches global memory,
    // so no stage waits for its stores.
    static_assert(AM_EC == NI, "one row of the 32 per thread in the elementwise passes");
    float xpf[2], lamv[2], lsum[2] = {0.f, 0.f}, K[2][5];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        const int e2 = es + 16 * hs, eb = b0 + e2;
        const bool evl = eb < S.B && ec < n_in;
        lamv[hs] = evl ? S.st[S.first].lam[(size_t)eb * n_in + ec] : 0.f;
        lds[e2 * PS + m.E + ec] = evl ? S.st[S.first].eps[(size_t)eb * n_in + ec] : 0.f;
#pragma unroll
        for (int d = 0; d < 5; ++d) K[hs][d] = 0.f;
    }

  for (int stg = S.first; stg >= S.last; --stg) {      // the stages of one Runge-Kutta step, last to first
    // opaque zero: keeps the compiler from hoisting the per-sample global addresses of all six array families out of
    // the stage loop into 64-bit register pairs (they pushed the kernel into scratch)
    int zopq = 0;
    asm volatile("" : "+v"(zopq));
    const int b0v = b0 + zopq;
    const AdjArgs& a = S.st[stg];
    AM_STAMP(14 + (stg == S.last ? 0 : 1));
    // the argument block of the NEXT stage is pulled into the scalar cache together with this stage's (one miss per
    // stage instead of two: the prefetch below would otherwise stall on it in the middle of the stage)
    {
        const AdjArgs& an = S.st[stg > S.last ? stg - 1 : stg];
        const int t0 = __float_as_int(an.cb), t1 = an.nw, t2 = an.B;
        const float* t3 = an.ustage;
        const float* t4 = an.w[2];
        asm volatile("" ::"s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4));
    }
    // ---- inputs: [z; ys; 0] -> S0; h_0 also goes out for the weight gradient ---------
    // The state row was requested during the previous stage (`xpf`, after its second sweep); the zbar of the stage just
    // finished sits in the S1 rows (its last epilogue put it there in front of the end-of-stage barrier).
    float kbv[2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        const int e2 = es + 16 * hs, eb = b0v + e2;
        const bool ev = eb < S.B;
        if (stg == S.first) xpf[hs] = adj3_fetch(a, nd, eb, ev, ec, n_in, in0, D);
        else {
            const float w = lds[e2 * PS + m.S1 + ec];                 // zbar of stage stg + 1 (zero rows beyond the batch)
            lsum[hs] += w;
#pragma unroll
            for (int d = 0; d < 5; ++d) K[hs][d] = fmaf(S.kc[stg + 1][d], w, K[hs][d]);
        }
        kbv[hs] = (ev && ec < n_in) ? fmaf(a.cb, lamv[hs], K[hs][0]) * a.hstep : 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) K[hs][d] = K[hs][d + 1];          // the next stage's sum moves to the front
        K[hs][4] = 0.f;
        lds[e2 * PS + m.S0 + ec] = xpf[hs];
        if (ev && ec < in0) __builtin_nontemporal_store(xpf[hs], a.HS + (size_t)eb * gl.sum_in + ec);
    }
    am_barrier();
    AM_STAMP(0);

    int cur = m.S0, nxt = m.S1;
    // ---- sweep 1: forward (h, sigma', sigma''); the last layer also forms pbar_L = eps .* sigma'_L ----
    auto fwd_epi = [&](int l, int r0, int ct, const f32x4& x) {
        float* Sw = lds + (16 * ct + s) * PS;
        const size_t gb = (size_t)(b0v + 16 * ct + s);
        const bool sv = b0v + 16 * ct + s < a.B;
        const int out = nd.dims[l + 1], act = nd.acts[l], oo = m.o_off[l];
        const bool last = l + 1 == NL;
        f32x4 h, d1, d2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float hh, dd1, dd2;
            if (ALL_TANH) { hh = tanh_fast(x[j]); dd1 = fmaf(-hh, hh, 1.0f); dd2 = -2.0f * hh * dd1; }   // as the step kernels
            else cnf_act2(act, x[j], hh, dd1, dd2);
            const bool live = r0 + j < out;            // padded rows stay exactly zero
            h[j] = live ? hh : 0.f; d1[j] = live ? dd1 : 0.f; d2[j] = live ? dd2 : 0.f;
        }
        *reinterpret_cast<f32x4*>(Sw + nxt + r0) = h;
        *reinterpret_cast<f32x4*>(Sw + m.D1 + oo + r0) = d1;
        *reinterpret_cast<f32x4*>(Sw + m.D2 + oo + r0) = d2;
        if (!last) {
            if (sv) am_store4(a.HS + gb * gl.sum_in + gl.in_off[l + 1] + r0, h, r0, out, m.vec4);
        } else {
            const f32x4 pb = *reinterpret_cast<const f32x4*>(Sw + m.E + r0) * d1;      // out_L == n_in
            *reinterpret_cast<f32x4*>(Sw + m.TB + oL + r0) = pb;
            if (sv) am_store4(a.PB + gb * gl.sum_out + gl.out_off[l] + r0, pb, r0, out, m.vec4o);
            // |zdot|^2 of this lane's 4 rows: one of the sample's 8 partials (2 tiles x 4 row groups)
            red[(16 * ct + s) * 8 + 4 * nt + q] = (h[0] * h[0] + h[1] * h[1]) + (h[2] * h[2] + h[3] * h[3]);
        }
    };
    f32x4 o0, o1;
    adj3_tile2(f1, lds + cur, PS, s, q, bias1, o0, o1);
    fwd_epi(0, r0w, 0, o0); fwd_epi(0, r0w, 1, o1);
    am_barrier();
    AM_STAMP(1);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    fetch_na(F3);
    adj3_tile2(f2, lds + cur, PS, s, q, bias2, o0, o1);
    fwd_epi(1, r0w, 0, o0); fwd_epi(1, r0w, 1, o1);
    am_barrier();
    AM_STAMP(2);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    if (narrow) fwd_epi(2, r0n, nc, adj3_tile1(na, lds + cur + 16 * nc * PS, PS, s, q, bias3));
    am_barrier();
    AM_STAMP(3);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    const int zd = cur;                                     // zdot stays in S[zd] until AH is formed (inside sweep 2)

    // ---- sweep 2: tbar chain (omega = eps): layer l's product turns pbar_l into tbar_{l-1} ----
    auto tb_epi = [&](int l, int r0, int ct, const f32x4& acc) {      // l = the layer whose transposed image was applied
        float* Sw = lds + (16 * ct + s) * PS;
        if (l > 0) {
            const int oprev = m.o_off[l - 1];
            *reinterpret_cast<f32x4*>(Sw + m.TB + oprev + r0) = acc;
            const f32x4 pb = acc * *reinterpret_cast<const f32x4*>(Sw + m.D1 + oprev + r0);
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = pb;
            if (b0v + 16 * ct + s < a.B)
                am_store4(a.PB + (size_t)(b0v + 16 * ct + s) * gl.sum_out + gl.out_off[l - 1] + r0, pb, r0, nd.dims[l], m.vec4o);
        } else {
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = acc;                 // tbar_0 = eJ
            float p2 = 0.f;                                                 // |eJ|^2 partial (rows of z only)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (r0 + j < n_in) p2 = fmaf(acc[j], acc[j], p2);
            red[256 + (16 * ct + s) * 8 + 4 * nt + q] = p2;
        }
    };
    fetch_na(R1);
    adj3_tile2(r3, lds + m.TB + oL, PS, s, q, zero4, o0, o1);
    tb_epi(2, r0w, 0, o0); tb_epi(2, r0w, 1, o1);
    am_barrier();
    AM_STAMP(4);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    {
        // zdot still sits in S[zd]: ahat = kbar_z + c_E zdot/|zdot| -> AH (shares the tbar_L slot of TB, free now)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int e2 = es + 16 * hs, eb = b0v + e2;
            const bool ev = eb < S.B;
            float nzs = 0.f;
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) nzs += red[e2 * 8 + p8];
            const float nz[2] = {nzs, nzs};
            const float inv = (nd.norm_z && nz[hs] > 0.f) ? a.c_E * __builtin_amdgcn_rsqf(nz[hs]) : 0.f;
            lds[e2 * PS + m.AH + ec] = (ev && ec < n_in) ? fmaf(inv, lds[e2 * PS + zd + ec], kbv[hs]) : 0.f;
        }
        am_barrier();                                   // zdot's buffer is the next epilogue's target
    }
    adj3_tile2(r2, lds + cur, PS, s, q, zero4, o0, o1);
    tb_epi(1, r0w, 0, o0); tb_epi(1, r0w, 1, o1);
    am_barrier();
    AM_STAMP(5);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    if (narrow) tb_epi(0, r0n, nc, adj3_tile1(na, lds + cur + 16 * nc * PS, PS, s, q, zero4));
    am_barrier();
    AM_STAMP(6);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    // eJ in S[cur].  tau = -c_l eps + c_n eJ/|eJ|  -> S[nxt] as t_0 (rows of ys and padding: 0)
    {
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int e2 = es + 16 * hs, eb = b0v + e2;
            const bool ev = eb < S.B;
            float njs = 0.f;
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) njs += red[256 + e2 * 8 + p8];
            const float nj[2] = {njs, njs};
            const float inv = (nd.norm_j && nj[hs] > 0.f) ? a.c_n * __builtin_amdgcn_rsqf(nj[hs]) : 0.f;
            for (int r = ec; r < NI; r += AM_EC) {
                float v = 0.f;
                if (r < n_in) v = fmaf(inv, lds[e2 * PS + cur + r], -a.c_l * lds[e2 * PS + m.E + r]);
                lds[e2 * PS + nxt + r] = v;
                if (ev && r < in0) __builtin_nontemporal_store(v, a.TS + (size_t)eb * gl.sum_in + r);
            }
        }
    }
    am_barrier();
    AM_STAMP(7);
    { const int t_ = cur; cur = nxt; nxt = t_; }

    if (stg > S.last) {                                   // inputs of the next stage: in flight during sweeps 3 and 4
        const AdjArgs& an = S.st[stg - 1];
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int eb = b0v + es + 16 * hs;
            xpf[hs] = adj3_fetch(an, nd, eb, eb < S.B, ec, n_in, in0, D);
        }
    }
    // ---- sweep 3: tangent chain; the last layer forms abar_L = ahat sigma' + eps q_L instead of t_L ----
    auto tan_epi = [&](int l, int r0, int ct, const f32x4& acc) {
        float* Sw = lds + (16 * ct + s) * PS;
        const size_t gb = (size_t)(b0v + 16 * ct + s);
        const bool sv = b0v + 16 * ct + s < a.B;
        const int out = nd.dims[l + 1], oo = m.o_off[l];
        const bool last = l + 1 == NL;
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(Sw + m.D1 + oo + r0);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(Sw + m.D2 + oo + r0) * acc;   // q_l = sigma'' .* p_l
        *reinterpret_cast<f32x4*>(Sw + m.D2 + oo + r0) = qv;
        if (!last) {
            const f32x4 t = d1 * acc;
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = t;
            if (sv) am_store4(a.TS + gb * gl.sum_in + gl.in_off[l + 1] + r0, t, r0, out, m.vec4);
        } else {
            const f32x4 ab = *reinterpret_cast<const f32x4*>(Sw + m.AH + r0) * d1 +
                             *reinterpret_cast<const f32x4*>(Sw + m.E + r0) * qv;
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = ab;
            if (sv) am_store4(a.AB + gb * gl.sum_out + gl.out_off[l] + r0, ab, r0, out, m.vec4o);
        }
    };
    fetch_na(F3);
    adj3_tile2(f1, lds + cur, PS, s, q, zero4, o0, o1);
    tan_epi(0, r0w, 0, o0); tan_epi(0, r0w, 1, o1);
    am_barrier();
    AM_STAMP(8);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    adj3_tile2(f2, lds + cur, PS, s, q, zero4, o0, o1);
    tan_epi(1, r0w, 0, o0); tan_epi(1, r0w, 1, o1);
    am_barrier();
    AM_STAMP(9);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    if (narrow) tan_epi(2, r0n, nc, adj3_tile1(na, lds + cur + 16 * nc * PS, PS, s, q, zero4));
    am_barrier();
    AM_STAMP(10);
    { const int t_ = cur; cur = nxt; nxt = t_; }

    // ---- sweep 4: hbar chain; abar_{l-1} = hbar_{l-1} sigma' + tbar_{l-1} q_{l-1}; the last one is zbar ----
    auto hb_epi = [&](int l, int r0, int ct, const f32x4& acc) {
        float* Sw = lds + (16 * ct + s) * PS;
        const size_t gb = (size_t)(b0v + 16 * ct + s);
        const bool sv = b0v + 16 * ct + s < a.B;
        if (l > 0) {
            const int oprev = m.o_off[l - 1];
            const f32x4 ab = acc * *reinterpret_cast<const f32x4*>(Sw + m.D1 + oprev + r0) +
                             *reinterpret_cast<const f32x4*>(Sw + m.TB + oprev + r0) *
                             *reinterpret_cast<const f32x4*>(Sw + m.D2 + oprev + r0);
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = ab;
            if (sv) am_store4(a.AB + gb * gl.sum_out + gl.out_off[l - 1] + r0, ab, r0, nd.dims[l], m.vec4o);
        } else {
            *reinterpret_cast<f32x4*>(Sw + nxt + r0) = sv ? acc : zero4;             // zbar: to the elementwise threads through LDS
        }
    };
    fetch_na(R1);
    adj3_tile2(r3, lds + cur, PS, s, q, zero4, o0, o1);
    hb_epi(2, r0w, 0, o0); hb_epi(2, r0w, 1, o1);
    am_barrier();
    AM_STAMP(11);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    adj3_tile2(r2, lds + cur, PS, s, q, zero4, o0, o1);
    hb_epi(1, r0w, 0, o0); hb_epi(1, r0w, 1, o1);
    am_barrier();
    AM_STAMP(12);
    { const int t_ = cur; cur = nxt; nxt = t_; }
    if (narrow) hb_epi(0, r0n, nc, adj3_tile1(na, lds + cur + 16 * nc * PS, PS, s, q, zero4));   // (nxt == S1 here: twelve swaps)
    am_barrier();                                         // zbar of this stage visible to the elementwise threads
    AM_STAMP(13);
  }
  // lambda <- lambda + sum over the stages of zbar   (rows of this workgroup's samples)
#pragma unroll
  for (int hs = 0; hs < 2; ++hs) {
    const int e2 = es + 16 * hs, eb = b0 + e2;
    if (eb < S.B && ec < n_in) S.lam_out[(size_t)eb * n_in + ec] = lamv[hs] + (lsum[hs] + lds[e2 * PS + m.S1 + ec]);
  }
}

static bool adj3_shape(const NetDesc& nd, const AdjMfmaLayout& m) {
    return !nd.jvp && nd.n_layers == 3 && m.dp[0] == 32 && m.dp[1] == 128 && m.dp[2] == 128 && m.dp[3] == 32 && m.nin_p == 32 &&
           AM_WAVES == 8 && adj3_lds_bytes(m) <= 160 * 1024;       // (k_adj3 is written for the VJP compute mode)
}

hipError_t launch_pack_adj_images(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* P,
                                  float* img, hipStream_t s) {
    int mx = 0;
    for (int l = 0; l < m.L; ++l) if (m.dp[l] * m.dp[l + 1] > mx) mx = m.dp[l] * m.dp[l + 1];
    hipLaunchKernelGGL(k_pack_adj_images, dim3((mx + 255) / 256, m.L), dim3(256), 0, s, nd, g, m, P, img);
    return hipGetLastError();
}

hipError_t launch_adj_mfma_step(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                                const AdjStepArgs& S, hipStream_t s) {
    const size_t lds = adj_mfma_lds_bytes(m);
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    static const bool generic_only = [] { const char* e = getenv("CNF_ADJ_GENERIC"); return e && e[0] == '1'; }();
    if (adj3_shape(nd, m) && !generic_only && S.first == 5 && S.last == 0 && S.lam_update && S.lam_out) {   // (whole steps: the zbar stay in the kernel)          // resident-fragment pullback (A/B switch: CNF_ADJ_GENERIC=1)
        const void* f3 = all_tanh ? (const void*)k_adj3<true> : (const void*)k_adj3<false>;
        const size_t lds3 = adj3_lds_bytes(m);
        hipError_t e3 = hipFuncSetAttribute(f3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        if (e3 != hipSuccess) return e3;
        const dim3 grid3((S.B + A3_NS - 1) / A3_NS);
        if (all_tanh) hipLaunchKernelGGL(k_adj3<true>, grid3, dim3(AM_THREADS), lds3, s, nd, g, m, img, S);
        else hipLaunchKernelGGL(k_adj3<false>, grid3, dim3(AM_THREADS), lds3, s, nd, g, m, img, S);
        return hipGetLastError();
    }
    const int tiles = (S.B + AM_NS - 1) / AM_NS;
    auto go = [&](auto tanh_c, auto jm_c) -> hipError_t {
        constexpr bool T = decltype(tanh_c)::value, J = decltype(jm_c)::value;
        hipError_t e = hipFuncSetAttribute((const void*)k_adj_mfma<T, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_adj_mfma<T, J>), dim3(tiles), dim3(AM_THREADS), lds, s, nd, g, m, img, S);
        return hipGetLastError();
    };
    using B1 = std::integral_constant<bool, true>;
    using B0 = std::integral_constant<bool, false>;
    if (nd.jvp) return all_tanh ? go(B1{}, B1{}) : go(B0{}, B1{});
    return all_tanh ? go(B1{}, B0{}) : go(B0{}, B0{});
}

// Two launches for a run of whole steps when that is less sequential work: the stage-parallel launch costs ~0.7 of a stage per
// round of CUs workgroups (1 per CU: 156 KB of LDS), the sequential one ~0.3 of a stage per stage (phase stamps at config 5,
// DESIGN 4.4).
bool adj_mfma_run_split(const NetDesc& nd, const AdjMfmaLayout& m, int B, int nsteps) {
    const int mode = adj_split_mode();
    static const bool generic_only = [] { const char* e = getenv("CNF_ADJ_GENERIC"); return e && e[0] == '1'; }();
    if (mode == 0 || nsteps < 1 || B < 1 || (adj3_shape(nd, m) && !generic_only)) return false;      // (k_adj3 has one form)
    if (mode == 1) return true;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus < 1) cus = 1;
    }
    const long nst = 6L * nsteps, rounds = (nst * ((B + AM_NS - 1) / AM_NS) + cus - 1) / cus;
    return 0.7 * rounds + 0.3 * nst < 0.95 * nst;
}

hipError_t launch_adj_mfma_run(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                               const AdjStepArgs* d_steps, const AdjStepArgs* h_steps, int nsteps, int B, float* scratch, hipStream_t s) {
    if (!d_steps || !h_steps || !scratch || nsteps < 1 || 6 * nsteps > 65535) return hipErrorInvalidValue;
    const size_t lds = adj_mfma_lds_bytes(m);
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    const int tiles = (B + AM_NS - 1) / AM_NS;
    auto pair = [&](auto tanh_c, auto jm_c) -> hipError_t {
        constexpr bool T = decltype(tanh_c)::value, J = decltype(jm_c)::value;
        if (nsteps == 1) {                                 // one step: its arguments by value
            hipError_t e = hipFuncSetAttribute((const void*)k_adj_mfma_split<T, 1, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_adj_mfma_split<T, 2, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_adj_mfma_split<T, 1, J>), dim3(tiles, 6), dim3(AM_THREADS), lds, s, nd, g, m, img, h_steps[0], scratch);
            hipLaunchKernelGGL((k_adj_mfma_split<T, 2, J>), dim3(tiles), dim3(AM_THREADS), lds, s, nd, g, m, img, h_steps[0], scratch);
            return hipGetLastError();
        }
        hipError_t e = hipFuncSetAttribute((const void*)k_adj_mfma_run<T, 1, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_adj_mfma_run<T, 2, J>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_adj_mfma_run<T, 1, J>), dim3(tiles, 6 * nsteps), dim3(AM_THREADS), lds, s, nd, g, m, img, d_steps, nsteps, scratch);
        hipLaunchKernelGGL((k_adj_mfma_run<T, 2, J>), dim3(tiles), dim3(AM_THREADS), lds, s, nd, g, m, img, d_steps, nsteps, scratch);
        return hipGetLastError();
    };
    using B1 = std::integral_constant<bool, true>;
    using B0 = std::integral_constant<bool, false>;
    if (nd.jvp) return all_tanh ? pair(B1{}, B1{}) : pair(B0{}, B1{});
    return all_tanh ? pair(B1{}, B0{}) : pair(B0{}, B0{});
}

size_t adj_mfma_scratch_floats(const AdjMfmaLayout& m, size_t B, int nsteps) { return (size_t)6 * nsteps * B * (size_t)m.SR; }

static std::atomic<int> g_adj_split{[] { const char* e = getenv("CNF_ADJ_SPLIT"); return e ? (e[0] == '0' ? 0 : 1) : -1; }()};
int adj_split_mode() { return g_adj_split.load(std::memory_order_relaxed); }
void set_adj_split_mode(int mode) { g_adj_split.store(mode < 0 ? -1 : (mode > 0 ? 1 : 0), std::memory_order_relaxed); }
