// k_adj_test -- the adjoint of the EXACT-TRACE (TestMode) solve for ANY Dense chain: the gradient of
// loss(icnf, TestMode(), xs, ps, st) = -mean(logpx) (src/base_icnf.jl:489-497) w.r.t. the flat parameters, which the reference
// takes with Enzyme through jacobian_batched (src/utils.jl:1-36) and the solve in its call tests (test/call_tests.jl:239-252,
// `diff_loss` with omode = TestMode()) and its benchmark suite (benchmark/benchmarks.jl:60-99).  k_solve_wave<TEST, GRAD>
// (cnf_wave.hip) covers small two-layer networks inside the launch of the solve; this kernel is the route for everything else --
// deeper networks (the headline 32-128-128-32), wider ones, any activation, conditional models.  A generic VALU formulation:
// correctness and coverage first (one sample per workgroup at a time, ~3 MFLOP per stage pullback at the headline shape).
//
// Discrete adjoint of the accepted Tsit5 steps (step sizes are constants), restated from oracle/cnf_grad_oracle.py
// (loss_and_grad_test, rhs_vjp_test).  Per step, in reverse: the stage states U_1 = u_n, U_{j+1} = u_n + h sum_m a_{j+1,m} k_m are
// formed again from the recorded u_n (only the z rows matter: the right-hand side is autonomous and its z part does not see the
// dlogp row), then for i = 6..1 the pullback of f(z) = (nn(z), -tr J(z)) at U_i with the cotangents
//     kbar = h (b_i lambda + sum_{m>i} a_{m,i} w_m)   of zdot        c = h b_i / B   of ldot = -tr J,
// and lambda += sum_i w_i.  With M_l = D_l W_l, P_l = M_{l-1} ... M_1 [I; 0] (P_1 = [I; 0]), Q_l = M_L ... M_{l+1} (Q_L = I) and
// G_l = (P_l Q_l)':
//     d tr / d W_l = D_l G_l         d tr / d a_l (direct) = s''(a_l) .* rowsum(W_l .* G_l)
//     abar_l = hbar_l .* s'_l - c s''_l .* rowsum(W_l .* G_l)      Wbar_l += abar_l h_{l-1}' - c D_l G_l      hbar_{l-1} = W_l' abar_l.
// Every workgroup owns a partial of the flat gradient (its samples, its threads' own elements: no atomics, bit-reproducible);
// k_grad_reduce adds the partials in workgroup order.
#include "cnf_gradt.h"
#include "cnf_am.h"

namespace {

constexpr int GT_THREADS = 256;

struct GtLayout {              // LDS carve-up (floats)
    int U, K, W, LAM, KB, X0, H, D1, D2, AB, HB, UP, PQ, total;
    int pst;                   // row stride of the P / Q' rows: n_in + 1 (odd: rows of consecutive j fall into different banks)
    int p_off[CNF_MAX_LAYERS], q_off[CNF_MAX_LAYERS], pq_floats;
    int in0, sum_out, out_off[CNF_MAX_LAYERS];
};

__host__ __device__ inline GtLayout gt_layout(const NetDesc& nd, bool pq_lds) {
    GtLayout g{};
    const int L = nd.n_layers, n_in = nd.n_in;
    g.in0 = n_in + nd.n_cond;
    int so = 0, md = g.in0;
    for (int l = 0; l < L; ++l) { g.out_off[l] = so; so += nd.dims[l + 1]; if (nd.dims[l + 1] > md) md = nd.dims[l + 1]; }
    g.sum_out = so;
    g.pst = n_in + 1;
    // P[l], l = 1..L-1: dims[l] x n_in;  Q'[l], l = 0..L-2: dims[l + 1] x n_in   (row-major, stride pst)
    int pq = 0;
    for (int l = 1; l < L; ++l) { g.p_off[l] = pq; pq += nd.dims[l] * g.pst; }
    for (int l = 0; l + 1 < L; ++l) { g.q_off[l] = pq; pq += nd.dims[l + 1] * g.pst; }
    g.pq_floats = pq;
    int p = 0;
    g.U = p; p += 6 * n_in;
    g.K = p; p += 6 * n_in;
    g.W = p; p += 6 * n_in;
    g.LAM = p; p += n_in;
    g.KB = p; p += n_in;
    g.X0 = p; p += g.in0;
    g.H = p; p += so;
    g.D1 = p; p += so;
    g.D2 = p; p += so;
    g.AB = p; p += md;
    g.HB = p; p += md;
    g.UP = p; p += (md > GT_THREADS ? md : GT_THREADS);
    g.PQ = p; if (pq_lds) p += pq;
    g.total = p;
    return g;
}

struct GtTab { float a[6][6]; float b[6]; };
static const GtTab kGtTab = {{{0, 0, 0, 0, 0, 0},
                              {TS_A21, 0, 0, 0, 0, 0},
                              {TS_A31, TS_A32, 0, 0, 0, 0},
                              {TS_A41, TS_A42, TS_A43, 0, 0, 0},
                              {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0},
                              {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0}},
                             {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76}};

__global__ void __launch_bounds__(GT_THREADS)
k_adj_test(NetDesc nd, AdjTestArgs a, const GtTab tab, int pq_lds) {
    extern __shared__ float lds[];
    const GtLayout g = gt_layout(nd, pq_lds != 0);
    const int tid = threadIdx.x, L = nd.n_layers, n_in = nd.n_in, D = n_in + 1, in0 = g.in0, pst = g.pst;
    float* const U = lds + g.U; float* const Kk = lds + g.K; float* const Ws = lds + g.W;
    float* const lam = lds + g.LAM; float* const kb = lds + g.KB; float* const x0 = lds + g.X0;
    float* const H = lds + g.H; float* const D1 = lds + g.D1; float* const D2 = lds + g.D2;
    float* const ab = lds + g.AB; float* const hb = lds + g.HB; float* const up = lds + g.UP;
    float* const scr = a.scratch + (size_t)blockIdx.x * a.scratch_per_wg;
    float* const PQ = pq_lds ? lds + g.PQ : scr;                       // P / Q' matrices
    float* const Gs = scr + (pq_lds ? 0 : g.pq_floats);                // G_l of the layer being pulled back: [j + k out]
    float* const gp = a.gpart + (size_t)blockIdx.x * a.n_params;
    const float* P = a.P;
    for (int e = tid; e < a.n_params; e += GT_THREADS) gp[e] = 0.f;
    auto Wm = [&](int l, int j, int k) -> float { return P[nd.w_off[l] + j + (size_t)k * nd.dims[l + 1]]; };   // W_l[j][k] (Lux: out x in, column-major)
    auto in_of = [&](int l) -> const float* { return l == 0 ? x0 : H + g.out_off[l - 1]; };
    auto in_dim = [&](int l) { return l == 0 ? in0 : nd.dims[l]; };

    // forward through the chain from x0 (z rows already there); with `derivs` also s' and s''
    auto forward = [&](bool derivs) {
        for (int l = 0; l < L; ++l) {
            const int out = nd.dims[l + 1], in = in_dim(l);
            const float* x = in_of(l);
            for (int j = tid; j < out; j += GT_THREADS) {
                float acc = P[nd.b_off[l] + j];
                for (int k = 0; k < in; ++k) acc = fmaf(Wm(l, j, k), x[k], acc);
                float h, d1, d2;
                cnf_act2(nd.acts[l], acc, h, d1, d2);
                H[g.out_off[l] + j] = h;
                if (derivs) { D1[g.out_off[l] + j] = d1; D2[g.out_off[l] + j] = d2; }
            }
            __syncthreads();
        }
    };

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        // conditioning rows of x0 (constant over the solve) and d loss / d z(t1) = z(t1) / B
        for (int k = tid; k < nd.n_cond; k += GT_THREADS) x0[n_in + k] = a.ys[(size_t)b * nd.n_cond + k];
        {
            const float* uf = a.traj + (size_t)a.nsteps * a.slot_stride + (size_t)b * D;
            for (int i = tid; i < n_in; i += GT_THREADS) lam[i] = uf[i] * a.lam_l;
        }
        __syncthreads();
        for (int step = a.nsteps - 1; step >= 0; --step) {
            const float hstep = a.hs[step];
            const float* un = a.traj + (size_t)step * a.slot_stride + (size_t)b * D;
            // ---- the stage states again: U_1 = u_n, k_j = nn(U_j), U_{j+1} = u_n + h sum a_{j+1,m} k_m ----
            for (int i = tid; i < n_in; i += GT_THREADS) U[i] = un[i];
            __syncthreads();
            for (int j = 0; j < 5; ++j) {
                for (int i = tid; i < n_in; i += GT_THREADS) x0[i] = U[j * n_in + i];
                __syncthreads();
                forward(false);
                for (int i = tid; i < n_in; i += GT_THREADS) {
                    Kk[j * n_in + i] = H[g.out_off[L - 1] + i];
                }
                __syncthreads();
                for (int i = tid; i < n_in; i += GT_THREADS) {
                    float acc = 0.f;
                    for (int m = 0; m <= j; ++m) acc = fmaf(tab.a[j + 1][m], Kk[m * n_in + i], acc);
                    U[(j + 1) * n_in + i] = fmaf(hstep, acc, U[i]);
                }
                __syncthreads();
            }
            // ---- the six pullbacks, last stage first ----
            for (int st = 5; st >= 0; --st) {
                const float c = hstep * tab.b[st] * a.lam_l;                 // cotangent of ldot = -tr J
                for (int i = tid; i < n_in; i += GT_THREADS) {
                    float acc = tab.b[st] * lam[i];
                    for (int m = st + 1; m < 6; ++m) acc = fmaf(tab.a[m][st], Ws[m * n_in + i], acc);
                    kb[i] = hstep * acc;
                    x0[i] = U[st * n_in + i];
                }
                __syncthreads();
                forward(true);
                // P[l] = M_{l-1} P[l-1]  (P[0] = [I; 0]):  P[1][j][i] = s'_0[j] W_0[j][i]
                for (int l = 1; l < L; ++l) {
                    const int rows = nd.dims[l], kin = in_dim(l - 1);
                    float* Pl = PQ + g.p_off[l];
                    const float* Pp = l > 1 ? PQ + g.p_off[l - 1] : nullptr;
                    const float* d1 = D1 + g.out_off[l - 1];
                    for (int e = tid; e < rows * n_in; e += GT_THREADS) {
                        const int j = e % rows, i = e / rows;
                        float acc;
                        if (l == 1) acc = Wm(0, j, i);
                        else { acc = 0.f; for (int k = 0; k < kin; ++k) acc = fmaf(Wm(l - 1, j, k), Pp[k * pst + i], acc); }
                        Pl[j * pst + i] = d1[j] * acc;
                    }
                    __syncthreads();
                }
                // Q'[l][k][i] = Q[l][i][k],  Q[l] = Q[l+1] M_{l+1}  (Q[L-1] = I):  Q'[L-2][k][i] = s'_{L-1}[i] W_{L-1}[i][k]
                for (int l = L - 2; l >= 0; --l) {
                    const int rows = nd.dims[l + 1], jn = nd.dims[l + 2];
                    float* Ql = PQ + g.q_off[l];
                    const float* Qn = l + 2 < L ? PQ + g.q_off[l + 1] : nullptr;
                    const float* d1 = D1 + g.out_off[l + 1];
                    for (int e = tid; e < rows * n_in; e += GT_THREADS) {
                        const int i = e % n_in, k = e / n_in;
                        float acc;
                        if (l == L - 2) acc = d1[i] * Wm(L - 1, i, k);
                        else { acc = 0.f; for (int j = 0; j < jn; ++j) acc = fmaf(Qn[j * pst + i] * d1[j], Wm(l + 1, j, k), acc); }
                        Ql[k * pst + i] = acc;
                    }
                    __syncthreads();
                }
                // ---- back through the layers ----
                for (int i = tid; i < n_in; i += GT_THREADS) hb[i] = kb[i];
                __syncthreads();
                for (int l = L - 1; l >= 0; --l) {
                    const int out = nd.dims[l + 1], in = in_dim(l);
                    const float* Pl = l > 0 ? PQ + g.p_off[l] : nullptr;
                    const float* Ql = l + 1 < L ? PQ + g.q_off[l] : nullptr;
                    const float* d1 = D1 + g.out_off[l]; const float* d2 = D2 + g.out_off[l];
                    // G_l[j][k] = sum_i P[l][k][i] Q[l][i][j] into the scratch; rowsum(W_l .* G_l) in slices of k
                    const int jt = out < GT_THREADS ? out : GT_THREADS;       // threads along j
                    const int S = GT_THREADS / jt;                              // slices of k
                    {
                        const int s = tid / jt;
                        for (int j = tid % jt; j < out && s < S; j += jt) {
                            float us = 0.f;
                            for (int k = s; k < in; k += S) {
                                float gv;
                                if (!Pl && !Ql) gv = (k < n_in && k == j) ? 1.f : 0.f;            // one layer: P = [I; 0], Q = I
                                else if (!Pl) gv = k < n_in ? Ql[j * pst + k] : 0.f;              // l = 0: G[j][k] = Q[0][k][j]
                                else if (!Ql) gv = Pl[k * pst + j];                               // l = L-1: G[j][k] = P[L-1][k][j]
                                else { gv = 0.f; for (int i = 0; i < n_in; ++i) gv = fmaf(Pl[k * pst + i], Ql[j * pst + i], gv); }
                                Gs[j + (size_t)k * out] = gv;
                                us = fmaf(Wm(l, j, k), gv, us);
                            }
                            up[s * jt + (j % jt)] = us;                         // (out > 256: a thread's j's share its slot in turn)
                            if (out > GT_THREADS) {                             // one j per pass: finish it here
                                ab[j] = hb[j] * d1[j] - c * d2[j] * us;
                            }
                        }
                    }
                    __syncthreads();
                    if (out <= GT_THREADS) {
                        for (int j = tid; j < out; j += GT_THREADS) {
                            float us = 0.f;
                            for (int s = 0; s < S; ++s) us += up[s * jt + j];
                            ab[j] = hb[j] * d1[j] - c * d2[j] * us;
                        }
                        __syncthreads();
                    }
                    // Wbar_l += abar_l h_{l-1}' - c D_l G_l ;  bbar_l += abar_l   (each element by its own thread: no atomics)
                    const float* x = in_of(l);
                    for (int e = tid; e < out * in; e += GT_THREADS) {
                        const int j = e % out, k = e / out;
                        float* w = gp + nd.w_off[l] + j + (size_t)k * out;
                        *w += ab[j] * x[k] - c * d1[j] * Gs[j + (size_t)k * out];
                    }
                    for (int j = tid; j < out; j += GT_THREADS) gp[nd.b_off[l] + j] += ab[j];
                    // hbar_{l-1} = W_l' abar_l   (hbar_l itself was last read when abar_l was formed)
                    for (int k = tid; k < in; k += GT_THREADS) {
                        float acc = 0.f;
                        for (int j = 0; j < out; ++j) acc = fmaf(Wm(l, j, k), ab[j], acc);
                        hb[k] = acc;
                    }
                    __syncthreads();
                }
                for (int i = tid; i < n_in; i += GT_THREADS) Ws[st * n_in + i] = hb[i];
                __syncthreads();
            }
            for (int i = tid; i < n_in; i += GT_THREADS) {
                float s = lam[i];
                for (int m = 0; m < 6; ++m) s += Ws[m * n_in + i];
                lam[i] = s;
            }
            __syncthreads();
        }
        for (int i = tid; i < n_in; i += GT_THREADS) a.lam_out[(size_t)b * n_in + i] = lam[i];
        __syncthreads();
    }
}

}  // namespace

size_t adj_test_scratch_floats(const NetDesc& nd) {
    const GtLayout g = gt_layout(nd, false);
    size_t mx = 0;
    for (int l = 0; l < nd.n_layers; ++l) {
        const size_t e = (size_t)nd.dims[l + 1] * (l == 0 ? g.in0 : nd.dims[l]);
        if (e > mx) mx = e;
    }
    return (size_t)g.pq_floats + mx + 64;
}

int adj_test_workgroups(int B) { return B < 256 ? B : 256; }

hipError_t launch_adj_test(const NetDesc& nd, const AdjTestArgs& a_, hipStream_t s) {
    AdjTestArgs a = a_;
    const int G = adj_test_workgroups(a.B);
    // the P / Q matrices in LDS when they fit beside the vectors (the headline shape: 64 KB of them), else in the scratch
    GtLayout g = gt_layout(nd, true);
    int pq_lds = 1;
    if ((size_t)g.total * sizeof(float) > 150 * 1024) { pq_lds = 0; g = gt_layout(nd, false); }
    const size_t lds = (size_t)g.total * sizeof(float);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void*)k_adj_test, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    GtTab tab = kGtTab;
    hipLaunchKernelGGL(k_adj_test, dim3(G), dim3(GT_THREADS), lds, s, nd, a, tab, pq_lds);
    return hipGetLastError();
}
