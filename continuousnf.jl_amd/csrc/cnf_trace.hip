// Exact trace on MFMA for networks with three or more layers (TestMode: src/icnf.jl:148-184 with
// jacobian_batched src/utils.jl:1-36).  The reference runs n_in AD sweeps and materialises an
// n_in x n_in x B tensor; here   tr J = sum_i [D_L W_L T_{L-1}]_ii,   T_l = D_l W_l T_{l-1},  T_0 = I,
// with D_l = diag(sigma'_l) per sample: the first layer is a row scaling of W_1, the last one only
// needs its diagonal, and the middle layers are GEMMs  W_l (out x in) x T_{l-1} (in x n_in) per sample.
// A workgroup (AM_WAVES waves) owns 16 samples: one forward pass for all of them (sigma' of every
// layer stays in LDS), then groups of `gs` samples whose tangent columns (gs * n_in of them, up to 8
// MFMA column tiles) share every weight fragment fetched from L2.  The diagonal of the last layer is
// folded into the epilogue of the last middle GEMM; J is never formed.  Two-layer networks use the
// closed form in cnf_mfma.hip instead.
#include <cstdlib>
#include "cnf_trace.h"
#include "cnf_mfma.h"
#include "cnf_am.h"

#define TR_NCMAX 8

// Tsit5 rows a_{s+1,1..s} (s = 1..6) for the kernels that run the stages of an attempt themselves
__device__ static const float kTsit5Row[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {TS_A21, 0, 0, 0, 0, 0},
    {TS_A31, TS_A32, 0, 0, 0, 0},
    {TS_A41, TS_A42, TS_A43, 0, 0, 0},
    {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0},
    {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0},
    {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76}};

// entry `idx` of the state the evaluation runs at: given (plain evaluation) or formed from the Runge-Kutta stages.
// Explicit selects: run-time indexing of kernel-argument arrays would go through memory.
__device__ __forceinline__ float trace_in(const TraceArgs& a, int cur, float hstep, size_t idx) {
    if (a.nk == 0) return a.u[idx];
    float acc = a.coef[0] * (cur ? a.K1[1] : a.K1[0])[idx];
    if (a.nk > 1) acc = fmaf(a.coef[1], a.Ks[0][idx], acc);
    if (a.nk > 2) acc = fmaf(a.coef[2], a.Ks[1][idx], acc);
    if (a.nk > 3) acc = fmaf(a.coef[3], a.Ks[2][idx], acc);
    if (a.nk > 4) acc = fmaf(a.coef[4], a.Ks[3][idx], acc);
    if (a.nk > 5) acc = fmaf(a.coef[5], a.Ks[4][idx], acc);
    return fmaf(hstep, acc, (cur ? a.U[1] : a.U[0])[idx]);
}
// stage 6: rows n_in .. D-1 of the new solution (the network never reads them); one thread per sample
__device__ __forceinline__ void trace_unew_tail(const TraceArgs& a, int cur, float hstep, int b, int n_in, int D) {
    float* un = cur ? a.U[0] : a.U[1];
    for (int r = n_in; r < D; ++r) un[(size_t)b * D + r] = trace_in(a, cur, hstep, (size_t)b * D + r);
}

static inline int pad8m16(int x) { return ((x + 15) & ~15) + 8; }
__device__ __forceinline__ int pad8m16_dev(int x) { return ((x + 15) & ~15) + 8; }

TraceLayout trace_layout(const NetDesc& nd, const AdjMfmaLayout& m) {
    TraceLayout t{};
    const int L = m.L;
    int midmax = 16;                                   // widest operand of a middle GEMM
    for (int l = 1; l <= L - 1; ++l) if (m.dp[l] > midmax) midmax = m.dp[l];
    t.PD = pad8m16(m.sum_o);
    t.PT = pad8m16(midmax);
    const int PSf = pad8m16(m.maxd);
    const int nbuf = L >= 4 ? 2 : 1;
    // as many column tiles per group as fit next to a second workgroup on the CU (<= 80 KB), at least one sample
    const int per_sample_tiles = m.nin_p / 16;                          // 1 .. 8 (trace_mfma_supported)
    // samples per group: a divisor of 16 (groups tile the workgroup's samples), as many as fit 8 column tiles and
    // leave room for a second workgroup on the CU (<= 80 KB), at least one
    int gs = 1;
    for (int g = 16; g >= 1; g /= 2) {
        const int c = g * per_sample_tiles;
        const size_t fl = (size_t)AM_NS * t.PD + (size_t)nbuf * 16 * c * t.PT + (size_t)AM_NS * AM_WAVES;
        if (c <= TR_NCMAX && (fl * 4 <= 80 * 1024 || g == 1)) { gs = g; break; }
    }
    int nct = gs * per_sample_tiles;
    t.nct = nct;
    t.gs = 16 * nct / m.nin_p;
    t.off_T0 = AM_NS * t.PD;
    size_t region = (size_t)nbuf * 16 * nct * t.PT;
    if (region < (size_t)2 * AM_NS * PSf) region = (size_t)2 * AM_NS * PSf;     // the forward ping-pong aliases it
    t.off_T1 = t.off_T0 + 16 * nct * t.PT;
    t.off_red = t.off_T0 + (int)region;
    t.total_floats = t.off_red + AM_NS * AM_WAVES;
    return t;
}

bool trace_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m) {
    if (nd.n_layers < 3 || nd.dims[nd.n_layers] != nd.n_in) return false;
    if (m.nin_p / 16 > TR_NCMAX) return false;
    const TraceLayout t = trace_layout(nd, m);
    return (size_t)t.total_floats * 4 <= 160 * 1024;
}

template <bool ALL_TANH, int NCT>
__global__ void __launch_bounds__(AM_THREADS)
k_trace_mfma(NetDesc nd, GradLayout gl, AdjMfmaLayout m, TraceLayout tl, const float* __restrict__ img, TraceArgs a) {
    if (a.st && a.st->done) return;
    extern __shared__ float lds[];
    const int st_cur = a.st ? a.st->cur : 0;
    const float st_h = a.st ? a.st->h : 0.f;
    const int NL = m.L, PD = tl.PD, PT = tl.PT;
    const int PSf = pad8m16_dev(m.maxd);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * AM_NS;
    const int n_in = nd.n_in, D = n_in + 1, in0 = gl.in0;
    const int es = (tid >> 4) & 15, ec = (tid & 15) | ((tid >> 8) << 4);
    const int eb = b0 + es;
    const bool ev = eb < a.B;
    float* du = a.du;
    if (a.st && a.du_is_k7) du = (a.st->cur ? a.K1[0] : a.K1[1]);
    float* red = lds + tl.off_red;

    AFrag pf;
    am_first(pf, img + m.ff_off[0], m.dp[1], m.dp[0]);
    int cur = tl.off_T0, nxt = tl.off_T0 + AM_NS * PSf;
    for (int r = ec; r < m.dp[0]; r += AM_EC) {
        float v = 0.f;
        if (ev && r < in0) {
            v = r < n_in ? trace_in(a, st_cur, st_h, (size_t)eb * D + r) : a.ys[(size_t)eb * nd.n_cond + (r - n_in)];
            if (a.also_unew && r < n_in) (st_cur ? a.U[0] : a.U[1])[(size_t)eb * D + r] = v;
        }
        lds[cur + es * PSf + r] = v;
    }
    if (a.also_unew && ev && ec == 0) trace_unew_tail(a, st_cur, st_h, eb, n_in, D);
    am_barrier();

    // ---- forward: zdot out, sigma' of every layer kept ------------------------------------------------
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], act = nd.acts[l], oo = m.o_off[l];
        const bool last = l + 1 == NL;
        am_gemm(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PSf, pf,
                img + (last ? m.ff_off[1] : m.ff_off[l + 1]), last ? m.dp[2] : m.dp[l + 2], last ? m.dp[1] : m.dp[l + 1],
                img + m.b_off[l], [&](int r0, int s, f32x4 acc, f32x4 bias) {
            f32x4 h, d1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float hh, dd1, dd2;
                if (ALL_TANH) { hh = cnf_tanh(acc[j] + bias[j]); dd1 = fmaf(-hh, hh, 1.0f); }
                else cnf_act2(act, acc[j] + bias[j], hh, dd1, dd2);
                const bool live = r0 + j < out;
                h[j] = live ? hh : 0.f; d1[j] = live ? dd1 : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + s * PD + oo + r0) = d1;
            if (!last) *reinterpret_cast<f32x4*>(lds + nxt + s * PSf + r0) = h;
            else if (b0 + s < a.B) {
                float* g = du + (size_t)(b0 + s) * D + r0;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (r0 + j < out) g[j] = h[j];                 // zdot rows
            }
        });
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }

    // ---- trace: groups of gs samples -------------------------------------------------------------------
    constexpr int nct = NCT;
    const int gs = tl.gs, tps = m.nin_p >> 4;                         // column tiles per sample
    const int d1w = m.dp[1];                                          // features of T_1
    const float* R0 = img + m.r_off[0];                               // W_1^T image [i][k]
    const float* WL = img + m.f_off[NL - 1];                          // W_L image [i][j], k_p = dp[NL-1]
    const int kL = m.dp[NL - 1];
    const int oLast = m.o_off[NL - 1];
    for (int g0 = 0; g0 < AM_NS; g0 += gs) {
        if (b0 + g0 >= a.B) break;                                    // uniform: no samples left in this workgroup
        // T_1[col][k] = sigma'_1[sample][k] * W_1[k][i]   (columns i >= n_in: zero)
        const int ncols = 16 * nct, nq = d1w >> 2;
        for (int e = tid; e < ncols * nq; e += AM_THREADS) {
            const int col = e / nq, k4 = (e % nq) << 2;
            const int sl = col / m.nin_p, i = col % m.nin_p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n_in)
                v = *reinterpret_cast<const f32x4*>(lds + (g0 + sl) * PD + m.o_off[0] + k4) *
                    *reinterpret_cast<const f32x4*>(R0 + (size_t)i * d1w + k4);
            *reinterpret_cast<f32x4*>(lds + tl.off_T0 + col * PT + k4) = v;
        }
        am_barrier();
        int tc = tl.off_T0, tn = tl.off_T1;
        float p[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) p[c] = 0.f;
        for (int l = 1; l <= NL - 2; ++l) {
            const bool lastmid = l == NL - 2;
            const int oo = m.o_off[l];
            const float* nimg = lastmid ? img + m.ff_off[1] : img + m.ff_off[l + 1];
            const int nr = lastmid ? m.dp[2] : m.dp[l + 2], nk = lastmid ? m.dp[1] : m.dp[l + 1];
            am_gemm_multi<NCT>(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + tc, PT, pf, nimg, nr, nk,
                               [&](int r0, int s, f32x4 (&acc)[NCT]) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    {
                        const int sample = g0 + c / tps, i = 16 * (c % tps) + s;
                        const f32x4 t = acc[c] * *reinterpret_cast<const f32x4*>(lds + sample * PD + oo + r0);
                        if (!lastmid) {
                            *reinterpret_cast<f32x4*>(lds + tn + (16 * c + s) * PT + r0) = t;
                        } else {
                            const f32x4 w = *reinterpret_cast<const f32x4*>(WL + (size_t)i * kL + r0);
                            p[c] = fmaf(t[0], w[0], fmaf(t[1], w[1], fmaf(t[2], w[2], fmaf(t[3], w[3], p[c]))));
                        }
                    }
                }
            });
            if (!lastmid) { am_barrier(); const int t_ = tc; tc = tn; tn = t_; }
        }
        // tr_sample = sum_i sigma'_L[i] * (sum over rows) ; lanes hold (column i, a quarter of the rows)
#pragma unroll
        for (int sl = 0; sl < NCT; ++sl) {
            if (sl < gs) {
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    if (c / tps == sl)
                        v = fmaf(p[c], lds[(g0 + sl) * PD + oLast + 16 * (c % tps) + (lane & 15)], v);
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[(g0 + sl) * AM_WAVES + wave] = v;
            }
        }
        am_barrier();                                                  // T buffers free for the next group
    }
    am_barrier();
    if (tid < AM_NS && b0 + tid < a.B) {
        float tr = 0.f;
        for (int w = 0; w < AM_WAVES; ++w) tr += red[tid * AM_WAVES + w];
        du[(size_t)(b0 + tid) * D + n_in] = -tr;                       // src/icnf.jl:162
    }
}

// ---------------------------------------------------------------------------------------------------
// Three-layer networks whose middle weight matrix fits the register file (BASELINE configs 3/4: 32-128-128-32):
//   tr J = sum_i d3_i sum_j W3[i][j] d2_j X[j][i],     X = (W2 diag(d1)) W1   per sample,
// as MFMA products with EVERY operand fragment resident in registers for the whole kernel: wave w keeps its 16-row
// tile of W2 (A), all of W1^T (B) and its slice of W3.  The per-sample part is one v_mul per two MFMAs (the A
// fragment scaled by sigma'_1, fetched from LDS as a broadcast); no tangent image is built in LDS, no weight is
// re-read, and after the forward pass the waves do not meet again until the final sum.  Two samples advance together:
// four independent accumulator chains.
// ---------------------------------------------------------------------------------------------------
template <bool ALL_TANH, int NIP, int H1, int H2>
__global__ void __launch_bounds__(AM_THREADS)
k_trace3(NetDesc nd, GradLayout gl, AdjMfmaLayout m, TraceLayout tl, const float* __restrict__ img, TraceArgs a) {
    extern __shared__ float lds[];
    // no branch before the loads: the integrator state is read through a pointer that is always valid and every
    // request of the prologue (state words, the 16 samples, the resident fragments) is in flight before `done` is tested
    const StepState* stp = a.st ? a.st : reinterpret_cast<const StepState*>(img);
    const int st_done = stp->done, st_cur = stp->cur;
    const float st_h = stp->h;
    constexpr int TI = NIP / 16, KB = H1 / 16, K0 = NIP / 16;
    static_assert(H2 / 16 == AM_WAVES && H1 / 16 == AM_WAVES, "one row tile of W1 and of W2 per wave");
#ifdef TR_STAMPS
    const unsigned long long te = __builtin_amdgcn_s_memtime();
#endif
    const int PD = tl.PD;
    const int PSf = pad8m16_dev(m.maxd);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * AM_NS;
    const int n_in = nd.n_in, D = n_in + 1, in0 = gl.in0;
    float* du = (a.st && a.du_is_k7) ? (st_cur ? a.K1[0] : a.K1[1]) : a.du;

    // state of the 16 samples first (its consumer, the LDS image, must not queue behind the weight fragments)
    const int es = tid >> 5, er = tid & 31;                            // (sample, input row): NIP == 32 rows
    float xin = 0.f;
    if (b0 + es < a.B && er < in0)
        xin = er < n_in ? trace_in(a, st_cur, st_h, (size_t)(b0 + es) * D + er)
                        : a.ys[(size_t)(b0 + es) * nd.n_cond + (er - n_in)];
    // resident fragments: W1 row tile (forward), W2 row tile (forward AND trace), all of W1^T (trace), the W3 slice
    // (trace epilogue AND this wave's k-block of the last forward layer)
    f32x4 w1a[K0], wa[KB], wb[TI][KB], w3[TI], bias1, bias2;
    {
        const float* W1 = img + m.f_off[0] + (size_t)(16 * wave + s) * NIP + 4 * q;       // [j][i]
        const float* W2 = img + m.f_off[1] + (size_t)(16 * wave + s) * H1 + 4 * q;        // [j][k]
        const float* WL = img + m.f_off[2] + 16 * wave + 4 * q;                            // W3 [i][j]
#pragma unroll
        for (int u = 0; u < K0; ++u) w1a[u] = *reinterpret_cast<const f32x4*>(W1 + 16 * u);
        bias1 = *reinterpret_cast<const f32x4*>(img + m.b_off[0] + 16 * wave + 4 * q);
#pragma unroll
        for (int u = 0; u < KB; ++u) wa[u] = *reinterpret_cast<const f32x4*>(W2 + 16 * u);
        bias2 = *reinterpret_cast<const f32x4*>(img + m.b_off[1] + 16 * wave + 4 * q);
#pragma unroll
        for (int c = 0; c < TI; ++c) w3[c] = *reinterpret_cast<const f32x4*>(WL + (size_t)(16 * c + s) * H2);
    }
    // W1^T is the same for all eight waves: one coalesced copy through LDS instead of eight scattered ones from L2
    // (the prologue is bound by the bytes the waves pull through the L1, not by latency)
    constexpr int WTS = H1 + 8, WT_PER = NIP * H1 / 4 / AM_THREADS;   // row stride of the staging image; float4s per thread
    f32x4 wt[WT_PER];
#pragma unroll
    for (int j = 0; j < WT_PER; ++j) {
        const int idx = tid + AM_THREADS * j, row = idx / (H1 / 4), c4 = idx % (H1 / 4);
        wt[j] = *reinterpret_cast<const f32x4*>(img + m.r_off[0] + (size_t)row * H1 + 4 * c4);
        if (row >= n_in) wt[j] = f32x4{0.f, 0.f, 0.f, 0.f};       // columns n_in.. of W1 are the conditioning inputs
    }
    const float bias3 = er < nd.dims[3] ? img[m.b_off[2] + er] : 0.f;

    float* S0 = lds + tl.off_T0;                    // [sample][feature], stride PSf: state, then h2
    float* S1 = S0 + AM_NS * PSf;                   // h1
    float* Z = S1 + AM_NS * PSf;                    // per-wave partial sums of the last layer [wave][sample][ZS]
    constexpr int ZS = NIP + 4;
    float* WT = Z + AM_WAVES * AM_NS * ZS;          // staging image of W1^T [i][k], stride WTS (never aliased)
    if (a.st && st_done) return;
    if (a.also_unew && b0 + es < a.B) {              // stage 6: the state this evaluation runs at is the new solution
        if (er < n_in) (st_cur ? a.U[0] : a.U[1])[(size_t)(b0 + es) * D + er] = xin;
        if (er == 0) trace_unew_tail(a, st_cur, st_h, b0 + es, n_in, D);
    }
#pragma unroll
    for (int j = 0; j < WT_PER; ++j) {
        const int idx = tid + AM_THREADS * j, row = idx / (H1 / 4), c4 = idx % (H1 / 4);
        *reinterpret_cast<f32x4*>(WT + row * WTS + 4 * c4) = wt[j];
    }
    S0[es * PSf + er] = xin;
    am_barrier();
#ifdef TR_STAMPS
    const unsigned long long tf = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int c = 0; c < TI; ++c)
#pragma unroll
        for (int u = 0; u < KB; ++u) wb[c][u] = *reinterpret_cast<const f32x4*>(WT + (16 * c + s) * WTS + 16 * u + 4 * q);
    // ---- forward, every weight already in registers: sigma' of every layer to LDS, zdot out ----
    auto act4v = [&](int l, const f32x4& x, f32x4& h, f32x4& d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float hh, dd1, dd2;
            if (ALL_TANH) { hh = cnf_tanh(x[j]); dd1 = fmaf(-hh, hh, 1.0f); }
            else cnf_act2(nd.acts[l], x[j], hh, dd1, dd2);
            h[j] = hh; d[j] = dd1;
        }
    };
    {   // layer 1: rows 16 wave .. +15 of H1
        f32x4 acc = bias1;
#pragma unroll
        for (int u = 0; u < K0; ++u) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(S0 + s * PSf + 16 * u + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[u][k], b[k], acc, 0, 0, 0);
        }
        f32x4 h, d;
        act4v(0, acc, h, d);
        const int r0 = 16 * wave + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j >= nd.dims[1]) { h[j] = 0.f; d[j] = 0.f; }
        *reinterpret_cast<f32x4*>(S1 + s * PSf + r0) = h;
        *reinterpret_cast<f32x4*>(lds + s * PD + m.o_off[0] + r0) = d;
    }
    am_barrier();
    {   // layer 2: rows 16 wave .. +15 of H2, A = the resident W2 tile
        f32x4 acc0 = bias2, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KB; u += 2) {
            const f32x4 b0v = *reinterpret_cast<const f32x4*>(S1 + s * PSf + 16 * u + 4 * q);
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(S1 + s * PSf + 16 * (u + 1) + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u][k], b0v[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u + 1][k], b1v[k], acc1, 0, 0, 0);
            }
        }
        f32x4 h, d;
        act4v(1, acc0 + acc1, h, d);
        const int r0 = 16 * wave + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j >= nd.dims[2]) { h[j] = 0.f; d[j] = 0.f; }
        *reinterpret_cast<f32x4*>(S0 + s * PSf + r0) = h;             // the state image was last read before the barrier
        *reinterpret_cast<f32x4*>(lds + s * PD + m.o_off[1] + r0) = d;
    }
    am_barrier();
    {   // layer 3: this wave's k-block (16 wave .. +15) of both output tiles; A = the resident W3 slice
        const f32x4 b = *reinterpret_cast<const f32x4*>(S0 + s * PSf + 16 * wave + 4 * q);
#pragma unroll
        for (int c = 0; c < TI; ++c) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[c][k], b[k], acc, 0, 0, 0);
            *reinterpret_cast<f32x4*>(Z + (wave * AM_NS + s) * ZS + 16 * c + 4 * q) = acc;
        }
    }
    am_barrier();
    {   // sum over the k-blocks, bias, activation: zdot out, sigma'_3 kept
        float z = bias3;
#pragma unroll
        for (int w = 0; w < AM_WAVES; ++w) z += Z[(w * AM_NS + es) * ZS + er];
        float hh, dd1, dd2;
        if (ALL_TANH) { hh = cnf_tanh(z); dd1 = fmaf(-hh, hh, 1.0f); }
        else cnf_act2(nd.acts[2], z, hh, dd1, dd2);
        const bool live = er < nd.dims[3];
        lds[es * PD + m.o_off[2] + er] = live ? dd1 : 0.f;
        if (live && b0 + es < a.B) du[(size_t)(b0 + es) * D + er] = hh;                    // zdot rows
    }
    am_barrier();

    // ---- trace: pairs of samples, no workgroup barrier ----
    // per-lane partials go to LDS as they are (the forward ping-pong buffers are free now); one reduction at the end
    float* part = lds + tl.off_T0;                                   // [sample][512 lanes]
#ifdef TR_STAMPS
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int o1 = m.o_off[0] + 4 * q, o2 = m.o_off[1] + 16 * wave + 4 * q, o3 = m.o_off[2] + s;
    f32x4 d1a = *reinterpret_cast<const f32x4*>(lds + o1), d1b = *reinterpret_cast<const f32x4*>(lds + PD + o1);
#ifdef TR_ABL_NOTRACE
    for (int s0 = 0; s0 < 0; s0 += 2) {
#else
    for (int s0 = 0; s0 < AM_NS; s0 += 2) {
#endif
        const float* da = lds + s0 * PD;
        const float* db = da + PD;
        const float* dnext = lds + (s0 + 2 < AM_NS ? s0 + 2 : 0) * PD;      // the next pair's first k-block
        f32x4 acc[2][TI];
#pragma unroll
        for (int c = 0; c < TI; ++c) { acc[0][c] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][c] = acc[0][c]; }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const f32x4 xa = wa[u] * d1a, xb = wa[u] * d1b;           // rows of W2 diag(sigma'_1), this k-block
            // sigma'_1 of the next k-block (of the next pair at the end) is requested before this block's MFMAs
            if (u + 1 < KB) {
                d1a = *reinterpret_cast<const f32x4*>(da + o1 + 16 * (u + 1));
                d1b = *reinterpret_cast<const f32x4*>(db + o1 + 16 * (u + 1));
            } else {
                d1a = *reinterpret_cast<const f32x4*>(dnext + o1);
                d1b = *reinterpret_cast<const f32x4*>(dnext + PD + o1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < TI; ++c) {
                    acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[k], wb[c][u][k], acc[0][c], 0, 0, 0);
                    acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[k], wb[c][u][k], acc[1][c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // acc[.][c][j] = X[16 wave + 4q + j][16c + s]
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* dn = n ? db : da;
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(dn + o2);
            float p = 0.f;
#pragma unroll
            for (int c = 0; c < TI; ++c) {
                const f32x4 t = acc[n][c] * d2 * w3[c];
                p = fmaf((t[0] + t[1]) + (t[2] + t[3]), dn[o3 + 16 * c], p);
            }
            part[(s0 + n) * AM_THREADS + tid] = p;
        }
    }
#ifdef TR_STAMPS
    {
        const unsigned long long tc1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
        if ((blockIdx.x == 3 || blockIdx.x == 300) && lane == 0 && (wave == 0 || wave == 5))
            printf("blk %d wave %d: prologue %llu forward %llu trace loop %llu cycles, %llu ticks of 100 MHz -> %.0f MHz\n", blockIdx.x, wave,
                   tf - te, tc0 - tf, tc1 - tc0, tr1 - tr0, (double)(tc1 - tc0) / (double)(tr1 - tr0) * 100.0);
    }
#endif
    am_barrier();
    {   // 32 lanes per sample: 16 partials each, then a 5-step butterfly
        const int smp = tid >> 5, j = tid & 31;
        float tr = 0.f;
#pragma unroll
        for (int i = 0; i < AM_THREADS / 32; ++i) tr += part[smp * AM_THREADS + j + 32 * i];
        for (int off = 16; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
        if (j == 0 && b0 + smp < a.B) du[(size_t)(b0 + smp) * D + n_in] = -tr;      // src/icnf.jl:162
    }
}

// ---------------------------------------------------------------------------------------------------
// k_trace3s -- the same exact trace with the per-sample product on SIX-TERM bf16 MFMAs (cnf_split.h), re-associated so
// that the operand a sample rescales is the SMALL one and never leaves the wave that builds it:
//   tr J = sum_{k,j} d1_k W2[j][k] (d2_j Z[k][j]),      Z = W1 (D3 W3)     per sample   (128 x 128, contraction over n_in),
// where k_trace3 forms X = (W2 D1) W1 (contraction over 128: a 128 x 128 operand rescaled per sample).  Wave w owns the
// 16 columns j = 16w .. 16w+15 of Z: its B operand is ITS column tile of D3 W3 D2 -- 8 values per lane, scaled and split in
// registers (16 products, 4 pair splits per sample) -- against all eight row tiles of W1, resident as split fragments (96
// VGPRs, split once per launch): 48 v_mfma_f32_16x16x32_bf16 per wave and sample where k_trace3 issues 64
// v_mfma_f32_16x16x4_f32 at a sixth of the rate, with no LDS traffic for the operands and NO barrier inside the sample loop
// (the waves drift freely).  The diagonal contraction runs on the accumulators against the wave's forward fragment of W2
// (the very registers of the forward pass) and d1 (68 VALU operations per sample).  Forward pass, stage-state assembly and
// outputs are k_trace3's.
//   STEP: the six stage evaluations of one attempt in this launch -- stage s forms its state from k_1..k_s, the result goes
//   to Ks[s-1], for s = 6 to K1[1 - cur] with the state stored as the new solution: what six launches did --, then the
//   error norm of the attempt and, in the workgroup that finishes last, the controller (what k_norm_partials did): one
//   launch per attempt.  norm_kind 0 / 1 fuse the two norms of the automatic initial dt into the plain launches likewise.
// ---------------------------------------------------------------------------------------------------
#include "cnf_split.h"
//   SOLVE: the WHOLE solve in this launch (the state in a.U[0] on entry, in a.U[cur] on exit): k1 = f(u0), the two
//   evaluations and norms of the automatic initial dt, then the step attempts -- STEP's six stages each -- with the
//   workgroups MEETING once per norm through the tagged words of k_solve3b (cnf_step3.hip) instead of a ticket and a launch
//   boundary: every workgroup adds the same partials in the same order and runs the same controller.  Needs every
//   workgroup resident (the launcher bounds the grid by the CU count); every wait is bounded (sv.wait_ticks / spin_limit),
//   a run-out ends the launch with the abort word and the host streams the launches instead.
template <bool ALL_TANH, bool STEP, int NS, int NIP, int H1, int H2, bool SOLVE = false>
__global__ void __launch_bounds__(AM_THREADS, 2)
k_trace3s(NetDesc nd, GradLayout gl, AdjMfmaLayout m, TraceLayout tl, const float* __restrict__ img, TraceArgs a, Solve3Args sv) {
    extern __shared__ float lds[];
    static_assert(!(SOLVE && STEP), "SOLVE contains STEP's attempt");
    const StepState* stp = (a.st && !SOLVE) ? a.st : reinterpret_cast<const StepState*>(img);
    const int st_done = SOLVE ? 0 : stp->done;
    int st_cur = SOLVE ? 0 : stp->cur;
    float st_h = SOLVE ? sv.init.h : stp->h;
    constexpr int TI = NIP / 16, KB = H1 / 16, K0 = NIP / 16;
    static_assert(NIP == 32 && H1 == 128 && H2 == 128 && AM_WAVES == 8, "one row tile of W1 and of W2 per wave, K = n_in = one k-block");
    const int PD = tl.PD;
    const int PSf = pad8m16_dev(m.maxd);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, q = lane >> 4;
    static_assert(NS == 16 || NS == 32, "samples per workgroup: one or two MFMA column tiles in the forward pass");
    constexpr int NH = NS / 16;
    const int b0 = blockIdx.x * NS;
    const int n_in = nd.n_in, D = n_in + 1, in0 = gl.in0;
#ifdef TR3S_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
    unsigned long long ts1 = 0, ts2 = 0, ts3 = 0;
#endif
    const int es = tid >> 5, er = tid & 31;                            // (sample, input row): NIP == 32 rows
    if (!SOLVE && a.st && st_done) {                 // queued past the end of the solve: keep the host's view current and leave
        if (a.ticket && blockIdx.x == 0 && tid == 0) mirror_store(a.mirror, a.seq, *a.st);
        return;
    }
    // ---- resident for the whole launch ----
    // forward fragments (fp32): W1 row tile, W2 row tile (ALSO the weights of the diagonal contraction), the W3 slice of
    // this wave's k-block of the last layer
    f32x4 w1a[K0], wa[KB], w3[TI], bias1, bias2;
    {
        const float* W1 = img + m.f_off[0] + (size_t)(16 * wave + s) * NIP + 4 * q;       // [j][i]
        const float* W2 = img + m.f_off[1] + (size_t)(16 * wave + s) * H1 + 4 * q;        // [j][k]
        const float* WL = img + m.f_off[2] + 16 * wave + 4 * q;                            // W3 [i][j]
#pragma unroll
        for (int u = 0; u < K0; ++u) w1a[u] = *reinterpret_cast<const f32x4*>(W1 + 16 * u);
        bias1 = *reinterpret_cast<const f32x4*>(img + m.b_off[0] + 16 * wave + 4 * q);
#pragma unroll
        for (int u = 0; u < KB; ++u) wa[u] = *reinterpret_cast<const f32x4*>(W2 + 16 * u);
        bias2 = *reinterpret_cast<const f32x4*>(img + m.b_off[1] + 16 * wave + 4 * q);
#pragma unroll
        for (int c = 0; c < TI; ++c) w3[c] = *reinterpret_cast<const f32x4*>(WL + (size_t)(16 * c + s) * H2);
    }
    const float bias3 = er < nd.dims[3] ? img[m.b_off[2] + er] : 0.f;
    // trace operands: the eight row tiles of W1 as split fragments (A: row 16 t + s, inputs 8q .. 8q+7; conditioning columns
    // zero), and W3[8q .. 8q+7][16 wave + s], the column of D3 W3 D2 this lane rescales per sample
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    bf16x8 ah[8], am_[8], al[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float* W1r = img + m.f_off[0] + (size_t)(16 * t + s) * NIP + 8 * q;
        f32x4 lo = *reinterpret_cast<const f32x4*>(W1r), hi = *reinterpret_cast<const f32x4*>(W1r + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (8 * q + j >= n_in) lo[j] = 0.f; if (8 * q + 4 + j >= n_in) hi[j] = 0.f; }
        u32x4_ h, mm, l;
        { unsigned x, y, z; s3b_split2(lo[0], lo[1], x, y, z); h.x = x; mm.x = y; l.x = z; }
        { unsigned x, y, z; s3b_split2(lo[2], lo[3], x, y, z); h.y = x; mm.y = y; l.y = z; }
        { unsigned x, y, z; s3b_split2(hi[0], hi[1], x, y, z); h.z = x; mm.z = y; l.z = z; }
        { unsigned x, y, z; s3b_split2(hi[2], hi[3], x, y, z); h.w = x; mm.w = y; l.w = z; }
        ah[t] = __builtin_bit_cast(bf16x8, h); am_[t] = __builtin_bit_cast(bf16x8, mm); al[t] = __builtin_bit_cast(bf16x8, l);
    }
    f32x4 w3lo, w3hi;
    {
        const float* R3 = img + m.r_off[2] + (size_t)(16 * wave + s) * NIP + 8 * q;       // reverse image [j][i] = W3[i][j]
        w3lo = *reinterpret_cast<const f32x4*>(R3); w3hi = *reinterpret_cast<const f32x4*>(R3 + 4);
    }
    float* S0 = lds + NS * PD;                      // (sigma' rows first: [sample][PD]) [sample][feature], stride PSf: state, then h2
    float* S1 = S0 + NS * PSf;                      // h1
    float* Z = S1 + NS * PSf;                       // per-wave partial sums of the last layer [wave][sample][ZS]
    constexpr int ZS = NIP + 4;
    float* red = Z + AM_WAVES * NS * ZS;            // trace partials [sample][wave]; then scratch of the norms

#ifdef TR3S_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts1 = __builtin_amdgcn_s_memtime();
#endif
    // SOLVE: the passes of the solve -- 0: k1 = f(u0) (+ the first norms of the automatic initial dt), 1: f(u0 + h0 k1) and its
    // norm, 2..: step attempts; the integrator state lives in LDS (thread 0 runs the controller on it after every meeting)
    float* const sol = red + NS * AM_WAVES;              // SOLVE scratch: [0] h, [1] cur, [2] done, [3] alive; [8..] the StepState
    StepState* const ns = reinterpret_cast<StepState*>(sol + 8);
    static_assert(sizeof(StepState) <= 24 * sizeof(float), "fits the scratch words");
    int pass = 0, nsync = 0, it_ = 0;
    unsigned mbase = 0;
    if (SOLVE) {
        if (tid == 0) *ns = sv.init;
        mbase = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv.base_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (sv.t_out && blockIdx.x == 0 && tid == 0) sv.t_out[0] = __builtin_amdgcn_s_memrealtime();
    }
    bool sol_alive = true;
  for (;;) {
    const bool attempt = SOLVE ? pass >= 2 : STEP;
  for (int stage = 1; stage <= (attempt ? 6 : 1); ++stage) {
    // this evaluation's stage state and destination (an attempt: of stage `stage`; else as the arguments / the pass say)
    const int nk = attempt ? stage : (SOLVE ? pass : a.nk);
    float cf[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) cf[j] = attempt ? kTsit5Row[stage][j] : (SOLVE ? (j == 0 ? 1.f : 0.f) : a.coef[j]);
    const bool is_k7 = attempt ? stage == 6 : (!SOLVE && a.st && a.du_is_k7), also_unew = attempt ? stage == 6 : (!SOLVE && a.also_unew != 0);
    float* du = is_k7 ? (st_cur ? a.K1[0] : a.K1[1])
                      : (attempt ? const_cast<float*>(stage == 1 ? a.Ks[0] : stage == 2 ? a.Ks[1] : stage == 3 ? a.Ks[2] : stage == 4 ? a.Ks[3] : a.Ks[4])
                                 : (SOLVE ? (pass == 0 ? (st_cur ? a.K1[1] : a.K1[0]) : const_cast<float*>(a.Ks[0])) : a.du));
    // entry `idx` of the state the evaluation runs at (trace_in with this stage's coefficients).  An opaque zero in the
    // index, new in every stage, keeps the compiler from hoisting the 64-bit addresses of the seven arrays (two sample halves,
    // two rows each) out of the stage loop -- 56 registers it then has to spill around the trace loop.
    int oz = 0;
    if (STEP || SOLVE) asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    auto state_in = [&](size_t idx) {
        idx += (size_t)oz;
        if (nk == 0) return SOLVE ? (st_cur ? a.U[1] : a.U[0])[idx] : a.u[idx];
        float acc = cf[0] * (st_cur ? a.K1[1] : a.K1[0])[idx];
        if (nk > 1) acc = fmaf(cf[1], a.Ks[0][idx], acc);
        if (nk > 2) acc = fmaf(cf[2], a.Ks[1][idx], acc);
        if (nk > 3) acc = fmaf(cf[3], a.Ks[2][idx], acc);
        if (nk > 4) acc = fmaf(cf[4], a.Ks[3][idx], acc);
        if (nk > 5) acc = fmaf(cf[5], a.Ks[4][idx], acc);
        return fmaf(st_h, acc, (st_cur ? a.U[1] : a.U[0])[idx]);
    };

#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {                 // thread (es, er): row er of samples es and es + 16
        const int sm = es + 16 * hh;
        float xin = 0.f;
        if (b0 + sm < a.B && er < in0)
            xin = er < n_in ? state_in((size_t)(b0 + sm) * D + er)
                            : a.ys[(size_t)(b0 + sm) * nd.n_cond + (er - n_in)];
        if (also_unew && b0 + sm < a.B) {            // stage 6: the state this evaluation runs at is the new solution
            float* un = st_cur ? a.U[0] : a.U[1];
            if (er < n_in) un[(size_t)(b0 + sm) * D + er] = xin;
            if (er == 0) un[(size_t)(b0 + sm) * D + n_in] = state_in((size_t)(b0 + sm) * D + n_in);     // the dlogp row
        }
        S0[sm * PSf + er] = xin;
    }
    am_barrier();
    // ---- forward (k_trace3's): sigma' of every layer to LDS, zdot out ----
    auto act4v = [&](int l, const f32x4& x, f32x4& h, f32x4& d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float hh, dd1, dd2;
            if (ALL_TANH) { hh = cnf_tanh(x[j]); dd1 = fmaf(-hh, hh, 1.0f); }
            else cnf_act2(nd.acts[l], x[j], hh, dd1, dd2);
            h[j] = hh; d[j] = dd1;
        }
    };
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {   // layer 1: rows 16 wave .. +15 of H1, samples 16 hh + s
        const int sm = s + 16 * hh;
        f32x4 acc = bias1;
#pragma unroll
        for (int u = 0; u < K0; ++u) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(S0 + sm * PSf + 16 * u + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[u][k], b[k], acc, 0, 0, 0);
        }
        f32x4 h, d;
        act4v(0, acc, h, d);
        const int r0 = 16 * wave + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j >= nd.dims[1]) { h[j] = 0.f; d[j] = 0.f; }
        *reinterpret_cast<f32x4*>(S1 + sm * PSf + r0) = h;
        *reinterpret_cast<f32x4*>(lds + sm * PD + m.o_off[0] + r0) = d;
    }
    am_barrier();
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {   // layer 2: rows 16 wave .. +15 of H2
        const int sm = s + 16 * hh;
        f32x4 acc0 = bias2, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KB; u += 2) {
            const f32x4 b0v = *reinterpret_cast<const f32x4*>(S1 + sm * PSf + 16 * u + 4 * q);
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(S1 + sm * PSf + 16 * (u + 1) + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u][k], b0v[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u + 1][k], b1v[k], acc1, 0, 0, 0);
            }
        }
        f32x4 h, d;
        act4v(1, acc0 + acc1, h, d);
        const int r0 = 16 * wave + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (r0 + j >= nd.dims[2]) { h[j] = 0.f; d[j] = 0.f; }
        *reinterpret_cast<f32x4*>(S0 + sm * PSf + r0) = h;            // the state image was last read before the barrier
        *reinterpret_cast<f32x4*>(lds + sm * PD + m.o_off[1] + r0) = d;
    }
    am_barrier();
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {   // layer 3: this wave's k-block (16 wave .. +15) of both output tiles
        const int sm = s + 16 * hh;
        const f32x4 b = *reinterpret_cast<const f32x4*>(S0 + sm * PSf + 16 * wave + 4 * q);
#pragma unroll
        for (int c = 0; c < TI; ++c) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[c][k], b[k], acc, 0, 0, 0);
            *reinterpret_cast<f32x4*>(Z + (wave * NS + sm) * ZS + 16 * c + 4 * q) = acc;
        }
    }
    am_barrier();
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {   // sum over the k-blocks, bias, activation: zdot out, sigma'_3 kept
        const int sm = es + 16 * hf;
        float z = bias3;
#pragma unroll
        for (int w = 0; w < AM_WAVES; ++w) z += Z[(w * NS + sm) * ZS + er];
        float hh, dd1, dd2;
        if (ALL_TANH) { hh = cnf_tanh(z); dd1 = fmaf(-hh, hh, 1.0f); }
        else cnf_act2(nd.acts[2], z, hh, dd1, dd2);
        const bool live = er < nd.dims[3];
        lds[sm * PD + m.o_off[2] + er] = live ? dd1 : 0.f;
        if (live && b0 + sm < a.B) du[(size_t)(b0 + sm) * D + er] = hh;                    // zdot rows
    }
    am_barrier();                                   // sigma' rows complete
#ifdef TR3S_STAMPS
    ts2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- trace: every wave on its own columns of Z, sample after sample, no barrier ----
    // Software pipeline inside the wave (its instruction stream is in-order; the matrix pipe runs beside it): behind the six
    // products of tile t come one quarter of the NEXT sample's operand split and the diagonal contraction of tile t - 1,
    // whose accumulator has had a tile's time to arrive.
    struct Bop { bf16x8 h, m, l; };
    auto scaled = [&](int b, f32x4& vlo, f32x4& vhi) {          // this lane's column of D3 W3 D2: inputs 8q .. 8q+7 of column 16 wave + s
        const float* db = lds + b * PD;
        const f32x4 dlo = *reinterpret_cast<const f32x4*>(db + m.o_off[2] + 8 * q), dhi = *reinterpret_cast<const f32x4*>(db + m.o_off[2] + 8 * q + 4);
        const float d2 = db[m.o_off[1] + 16 * wave + s];
        vlo = (w3lo * dlo) * d2; vhi = (w3hi * dhi) * d2;
    };
    u32x4_ nh, nm, nl;                               // the next sample's operand, filled a quarter at a time
    auto split_q = [&](int qq, const f32x4& vlo, const f32x4& vhi) {
        unsigned x, y, z;
        if (qq == 0) { s3b_split2(vlo[0], vlo[1], x, y, z); nh.x = x; nm.x = y; nl.x = z; }
        else if (qq == 1) { s3b_split2(vlo[2], vlo[3], x, y, z); nh.y = x; nm.y = y; nl.y = z; }
        else if (qq == 2) { s3b_split2(vhi[0], vhi[1], x, y, z); nh.z = x; nm.z = y; nl.z = z; }
        else { s3b_split2(vhi[2], vhi[3], x, y, z); nh.w = x; nm.w = y; nl.w = z; }
    };
    Bop bc;
    {
        f32x4 vlo, vhi;
        scaled(0, vlo, vhi);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) split_q(qq, vlo, vhi);
        bc.h = __builtin_bit_cast(bf16x8, nh); bc.m = __builtin_bit_cast(bf16x8, nm); bc.l = __builtin_bit_cast(bf16x8, nl);
    }
    for (int b = 0; b < NS; ++b) {
        const float* db = lds + b * PD;
        f32x4 vlo, vhi;
        scaled(b + 1 < NS ? b + 1 : b, vlo, vhi);          // (the last iteration prepares a copy that nobody uses)
        f32x4 sum = {0.f, 0.f, 0.f, 0.f}, cprev = sum;
        f32x4 d1n = *reinterpret_cast<const f32x4*>(db + m.o_off[0] + 4 * q);          // d1 of tile 0
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bc.h, c, 0, 0, 0);       // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bc.l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_[t], bc.m, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am_[t], bc.h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bc.m, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bc.h, c, 0, 0, 0);
#ifndef TR3S_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (t < 4) split_q(t, vlo, vhi);
            if (t > 0) {
                // cprev[r] = d2[j] Z[k = 16 (t-1) + 4q + r][j = 16 wave + s]; W2[j][k] = wa[t-1][r], d1[k] = d1n[r]
                const f32x4 wd = wa[t - 1] * d1n;
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = fmaf(cprev[r], wd[r], sum[r]);
            }
            d1n = *reinterpret_cast<const f32x4*>(db + m.o_off[0] + 16 * t + 4 * q);
#ifdef TR3S_INTERLEAVE
            // -DTR3S_INTERLEAVE=<n> (A/B, round 5): the tile's vector work BETWEEN its six dependent MFMAs, n VALU instructions per gap,
            // instead of behind them.  Measured on the TestMode solve of the headline network (B = 8192, 74 evaluations): n = 3:
            // 3.76-3.81 ms, n = 5: 3.60-3.70 ms, as shipped: 3.59-3.63 ms -- no gain: not the default.
#pragma unroll
            for (int g_ = 0; g_ < 6; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, TR3S_INTERLEAVE, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            cprev = c;
        }
        {
            const f32x4 wd = wa[7] * d1n;
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] = fmaf(cprev[r], wd[r], sum[r]);
        }
        bc.h = __builtin_bit_cast(bf16x8, nh); bc.m = __builtin_bit_cast(bf16x8, nm); bc.l = __builtin_bit_cast(bf16x8, nl);
        float p = (sum[0] + sum[1]) + (sum[2] + sum[3]);
        // fixed tree over the 64 lanes (quad swaps, half-row, row mirrors, then the four row totals)
        p += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(p), 0xB1, 0xF, 0xF, true));
        p += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(p), 0x4E, 0xF, 0xF, true));
        p += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(p), 0x141, 0xF, 0xF, true));
        p += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(p), 0x140, 0xF, 0xF, true));
        const int pi = __float_as_int(p);
        const float tot = (__int_as_float(__builtin_amdgcn_readlane(pi, 0)) + __int_as_float(__builtin_amdgcn_readlane(pi, 16))) +
                          (__int_as_float(__builtin_amdgcn_readlane(pi, 32)) + __int_as_float(__builtin_amdgcn_readlane(pi, 48)));
        if (lane == 0) red[b * AM_WAVES + wave] = tot;
    }
#ifdef TR3S_STAMPS
    ts3 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x == 3 || blockIdx.x == 200) && lane == 0 && (wave == 0 || wave == 5))
        printf("k_trace3s blk %d wave %d: resident operands %llu, state + forward %llu, trace loop %llu cycles\n", blockIdx.x, wave,
               ts1 - ts0, ts2 - ts1, ts3 - ts2);
#endif
    am_barrier();
    if (tid < NS && b0 + tid < a.B) {
        float tr = 0.f;
        for (int w = 0; w < AM_WAVES; ++w) tr += red[tid * AM_WAVES + w];
        du[(size_t)(b0 + tid) * D + n_in] = -tr;                       // src/icnf.jl:162
    }
    // The stage derivative just written is read back by OTHER threads of this workgroup (the next stage state, the norms):
    // the stores have to be complete before anybody loads those lines (nothing of them is in this CU's L1 yet).
    if (STEP || SOLVE || a.norm_kind >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); am_barrier(); }
  }
    if (SOLVE) {
        // ---- the norm of this pass over this workgroup's entries, the meeting, the controller ----
        const int nkind = pass >= 2 ? 2 : pass;
        const bool hairer = sv.hairer != 0;
        if (pass == 0 && !hairer) { pass = 2; continue; }
        const float abstol = ns->abstol, reltol = ns->reltol;
        const float* U = st_cur ? a.U[1] : a.U[0];
        const float* K1c = st_cur ? a.K1[1] : a.K1[0];
        float p0 = 0.f, p1 = 0.f;
        auto entry = [&](size_t i) {
            if (nkind == 0) {
                const float uv = U[i], sk = fmaf(fabsf(uv), reltol, abstol);
                const float x = uv / sk, y = K1c[i] / sk;
                p0 = fmaf(x, x, p0); p1 = fmaf(y, y, p1);
            } else if (nkind == 1) {
                const float uv = U[i], sk = fmaf(fabsf(uv), reltol, abstol);
                const float x = (a.Ks[0][i] - K1c[i]) / sk;
                p0 = fmaf(x, x, p0);
            } else {
                const float* un = st_cur ? a.U[0] : a.U[1];
                const float* k7 = st_cur ? a.K1[0] : a.K1[1];
                float e = TS_BT1 * K1c[i];
                e = fmaf(TS_BT2, a.Ks[0][i], e); e = fmaf(TS_BT3, a.Ks[1][i], e); e = fmaf(TS_BT4, a.Ks[2][i], e);
                e = fmaf(TS_BT5, a.Ks[3][i], e); e = fmaf(TS_BT6, a.Ks[4][i], e); e = fmaf(TS_BT7, k7[i], e);
                e *= st_h;
                const float uv = U[i], nv = un[i];
                const float sc = fmaf(fmaxf(fabsf(uv), fabsf(nv)), reltol, abstol);
                const float x = e / sc;
                p0 = fmaf(x, x, p0);
                if (!(fabsf(nv) <= 3.0e38f)) p1 += 1.f;
            }
        };
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int sm = es + 16 * hf;
            if (b0 + sm < a.B) {
                if (er < n_in) entry((size_t)(b0 + sm) * D + er);
                if (er == 0) entry((size_t)(b0 + sm) * D + n_in);
            }
        }
        auto wsum = [&](float v) {
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
            const int i = __float_as_int(v);
            return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
                   (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
        };
        float* nr = red;
        p0 = wsum(p0); p1 = wsum(p1);
        if (lane == 0) { nr[wave] = p0; nr[8 + wave] = p1; }
        am_barrier();
        unsigned long long* pb = reinterpret_cast<unsigned long long*>(sv.part) + (nsync & 1) * 1024;
        const unsigned tag = mbase + (unsigned)nsync + 1u;
        if (tid == 0) {
            float s0 = 0.f, s1 = 0.f;
            for (int w = 0; w < AM_WAVES; ++w) { s0 += nr[w]; s1 += nr[8 + w]; }
            __hip_atomic_store(pb + 2 * blockIdx.x, ((unsigned long long)tag << 32) | __float_as_uint(s0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb + 2 * blockIdx.x + 1, ((unsigned long long)tag << 32) | __float_as_uint(s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float q0 = 0.f, q1 = 0.f;
        int ok = 1;
        if (tid < (int)gridDim.x) {                      // thread i polls workgroup i's two words (grid <= AM_THREADS)
            typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
            const auto prs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, 16 * 512, 0x00020000);
            ok = 0;
            const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
            for (int spin = 0; spin < sv.spin_limit; ++spin) {
                const u32x4p wq = __builtin_bit_cast(u32x4p, __builtin_amdgcn_raw_buffer_load_b128(prs, 16 * tid, 0, 0x11));
                if (wq.y == tag && wq.w == tag) { q0 = __uint_as_float(wq.x); q1 = __uint_as_float(wq.z); ok = 1; break; }
                if ((spin & 63) == 63 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        q0 = wsum(q0); q1 = wsum(q1);
        const float bad = wsum(ok ? 0.f : 1.f);
        am_barrier();                                    // (thread 0 has read nr[])
        if (lane == 0) { nr[wave] = q0; nr[8 + wave] = q1; nr[16 + wave] = bad; }
        am_barrier();
        ++nsync;
        if (tid == 0) {
            float s0 = 0.f, s1 = 0.f, sb = 0.f;
            for (int w = 0; w < AM_WAVES; ++w) { s0 += nr[w]; s1 += nr[8 + w]; sb += nr[16 + w]; }
            int accepted = 0;
            if (sb == 0.f) {
                if (nkind < 2) ctrl_phase(ns, nkind, s0, s1, a.n_total);
                else {
                    const int acc0 = ns->naccept;
                    const float t_att = ns->t, h_att = ns->h;
                    ctrl_after_step(ns, s0, s1, a.n_total);
                    accepted = ns->naccept != acc0;
                    if (sv.trace && blockIdx.x == 0 && it_ < sv.trace_cap) {
                        float* tr = sv.trace + 4 * it_;
                        tr[0] = t_att; tr[1] = h_att; tr[2] = ns->eest; tr[3] = accepted ? 1.f : 0.f;
                    }
                }
            }
            sol[0] = ns->h; sol[1] = __int_as_float(ns->cur); sol[2] = __int_as_float(ns->done); sol[3] = sb == 0.f ? 1.f : 0.f;
        }
        am_barrier();
        st_h = sol[0]; st_cur = __float_as_int(sol[1]);
        const int done = __float_as_int(sol[2]);
        sol_alive = sol[3] != 0.f;
        am_barrier();
        if (pass >= 2) ++it_;
        pass = pass < 2 ? pass + 1 : 2;
        if (!sol_alive || done || it_ >= sv.maxiters) break;
        continue;
    }
    break;
  }
    if (SOLVE) {
        if (blockIdx.x == 0 && tid == 0) {
            if (!sol_alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns->done = 0; ns->n_partials = -1; }
            __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.st_mut = *ns;
            if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
            mirror_store(a.mirror, a.seq, *ns);
        }
        return;
    }
    if (a.norm_kind < 0) return;
    // ---- fused norm (k_norm_partials: kind 0 / 1 = the two norms of the automatic initial dt, 2 = the error estimate of
    // the attempt) over this workgroup's 32 x D entries, then -- in the workgroup that draws the last ticket -- the controller
    {
        const StepState* sp = a.st_mut;
        const float abstol = sp->abstol, reltol = sp->reltol;
        const float* U = a.nk == 0 && !STEP && a.u ? a.u : (st_cur ? a.U[1] : a.U[0]);
        const float* K1c = st_cur ? a.K1[1] : a.K1[0];
        float p0 = 0.f, p1 = 0.f;
        auto entry = [&](size_t i) {
            if (a.norm_kind == 0) {                  // f0 = the evaluation just written (a.du)
                const float uv = U[i], sk = fmaf(fabsf(uv), reltol, abstol);
                const float x = uv / sk, y = a.du[i] / sk;
                p0 = fmaf(x, x, p0); p1 = fmaf(y, y, p1);
            } else if (a.norm_kind == 1) {           // f1 = the evaluation just written (Ks[0]), f0 = k1
                const float uv = U[i], sk = fmaf(fabsf(uv), reltol, abstol);
                const float x = (a.du[i] - K1c[i]) / sk;
                p0 = fmaf(x, x, p0);
            } else {
                const float* un = st_cur ? a.U[0] : a.U[1];
                const float* k7 = st_cur ? a.K1[0] : a.K1[1];
                float e = TS_BT1 * K1c[i];
                e = fmaf(TS_BT2, a.Ks[0][i], e); e = fmaf(TS_BT3, a.Ks[1][i], e); e = fmaf(TS_BT4, a.Ks[2][i], e);
                e = fmaf(TS_BT5, a.Ks[3][i], e); e = fmaf(TS_BT6, a.Ks[4][i], e); e = fmaf(TS_BT7, k7[i], e);
                e *= st_h;
                const float uv = U[i], nv = un[i];
                const float sc = fmaf(fmaxf(fabsf(uv), fabsf(nv)), reltol, abstol);
                const float x = e / sc;
                p0 = fmaf(x, x, p0);
                if (!(fabsf(nv) <= 3.0e38f)) p1 += 1.f;
            }
        };
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int sm = es + 16 * hf;
            if (b0 + sm < a.B) {
                if (er < n_in) entry((size_t)(b0 + sm) * D + er);
                if (er == 0) entry((size_t)(b0 + sm) * D + n_in);
            }
        }
        // fixed trees: the 64 lanes of a wave, then the eight waves
        auto wsum = [&](float v) {
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
            v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
            const int i = __float_as_int(v);
            return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
                   (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
        };
        float* nr = red;
        p0 = wsum(p0); p1 = wsum(p1);
        if (lane == 0) { nr[wave] = p0; nr[8 + wave] = p1; }
        am_barrier();
        if (tid == 0) {
            float s0 = 0.f, s1 = 0.f;
            for (int w = 0; w < AM_WAVES; ++w) { s0 += nr[w]; s1 += nr[8 + w]; }
            __hip_atomic_store(a.partials + 2 * blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = tk == gridDim.x - 1;
            if (last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nr[16] = last ? 1.f : 0.f;
        }
        am_barrier();
        if (nr[16] == 0.f) return;
        float q0 = 0.f, q1 = 0.f;                    // the last workgroup: all partials in the fixed order of k_controller
        for (int i = tid; i < (int)gridDim.x; i += AM_THREADS) {
            q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        q0 = wsum(q0); q1 = wsum(q1);
        am_barrier();
        if (lane == 0) { nr[wave] = q0; nr[8 + wave] = q1; }
        am_barrier();
        if (tid == 0) {
            float s0 = 0.f, s1 = 0.f;
            for (int w = 0; w < AM_WAVES; ++w) { s0 += nr[w]; s1 += nr[8 + w]; }
            ctrl_phase(a.st_mut, a.norm_kind, s0, s1, a.n_total);
            mirror_store(a.mirror, a.seq, *a.st_mut);
        }
    }
}

static bool trace3_shape(const NetDesc& nd, const AdjMfmaLayout& m) {
    return nd.n_layers == 3 && m.nin_p == 32 && m.dp[0] == 32 && m.dp[1] == 128 && m.dp[2] == 128 && m.dp[3] == 32 &&
           AM_WAVES == 8;
}

// ---------------------------------------------------------------------------------------------------
// TrainMode, JVP compute mode (src/icnf.jl:384-420) for networks whose weights + tangent images do not
// fit the fused step kernel's LDS plan: (zdot, J eps) by one forward sweep in which the activations and
// the tangents of the 16 samples are two column tiles sharing every weight fragment;
//   ldot = -eps' (J eps),  Edot = |zdot|,  ndot = |J eps|.
// ---------------------------------------------------------------------------------------------------
struct JvpLayout { int PX, off_E, off_red, total_floats; };

static JvpLayout jvp_layout(const NetDesc&, const AdjMfmaLayout& m) {
    JvpLayout j{};
    j.PX = pad8m16(m.maxd);
    j.off_E = 2 * 32 * j.PX;                       // two ping-pong buffers of 32 columns (16 h + 16 t)
    j.off_red = j.off_E + AM_NS * (m.nin_p + 8);
    j.total_floats = j.off_red + 3 * AM_EC * AM_NS;
    return j;
}

bool jvp_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m) {
    if (!nd.jvp || nd.dims[nd.n_layers] != nd.n_in) return false;
    return (size_t)jvp_layout(nd, m).total_floats * 4 <= 160 * 1024;
}

template <bool ALL_TANH>
__global__ void __launch_bounds__(AM_THREADS)
k_jvp_mfma(NetDesc nd, GradLayout gl, AdjMfmaLayout m, JvpLayout jl, const float* __restrict__ img, TraceArgs a,
           const float* __restrict__ eps) {
    if (a.st && a.st->done) return;
    extern __shared__ float lds[];
    const int st_cur = a.st ? a.st->cur : 0;
    const float st_h = a.st ? a.st->h : 0.f;
    const int NL = m.L, PX = jl.PX, PE = m.nin_p + 8;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * AM_NS;
    const int n_in = nd.n_in, D = n_in + 3, in0 = gl.in0;
    const int es = (tid >> 4) & 15, ec = (tid & 15) | ((tid >> 8) << 4);
    const int eb = b0 + es;
    const bool ev = eb < a.B;
    float* du = a.du;
    if (a.st && a.du_is_k7) du = (a.st->cur ? a.K1[0] : a.K1[1]);
    float* red = lds + jl.off_red;

    AFrag pf;
    am_first(pf, img + m.ff_off[0], m.dp[1], m.dp[0]);
    int cur = 0, nxt = 32 * PX;
    for (int r = ec; r < m.dp[0]; r += AM_EC) {
        float v = 0.f, e = 0.f;
        if (ev && r < in0) {
            v = r < n_in ? trace_in(a, st_cur, st_h, (size_t)eb * D + r) : a.ys[(size_t)eb * nd.n_cond + (r - n_in)];
            if (a.also_unew && r < n_in) (st_cur ? a.U[0] : a.U[1])[(size_t)eb * D + r] = v;
        }
        if (ev && r < n_in) e = eps[(size_t)eb * n_in + r];
        lds[cur + es * PX + r] = v;                       // h_0 = [z; ys]
        lds[cur + (16 + es) * PX + r] = e;                // t_0 = [eps; 0]
        if (r < m.nin_p) lds[jl.off_E + es * PE + r] = e;
    }
    if (a.also_unew && ev && ec == 0) trace_unew_tail(a, st_cur, st_h, eb, n_in, D);
    am_barrier();
    for (int l = 0; l < NL; ++l) {
        const int out = nd.dims[l + 1], act = nd.acts[l];
        const bool last = l + 1 == NL;
        am_gemm_multi<2>(img + m.ff_off[l], m.dp[l + 1], m.dp[l], lds + cur, PX, pf,
                         last ? nullptr : img + m.ff_off[l + 1], last ? 0 : m.dp[l + 2], last ? 0 : m.dp[l + 1],
                         [&](int r0, int s, f32x4 (&acc)[2]) {
            f32x4 h, t;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float hh, dd1, dd2;
                if (ALL_TANH) { hh = cnf_tanh(acc[0][j]); dd1 = fmaf(-hh, hh, 1.0f); }
                else cnf_act2(act, acc[0][j], hh, dd1, dd2);
                const bool live = r0 + j < out;
                h[j] = live ? hh : 0.f; t[j] = live ? dd1 * acc[1][j] : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + nxt + s * PX + r0) = h;
            *reinterpret_cast<f32x4*>(lds + nxt + (16 + s) * PX + r0) = t;
            if (last && b0 + s < a.B) {
                float* g = du + (size_t)(b0 + s) * D + r0;
#pragma unroll
                for (int j = 0; j < 4; ++j) if (r0 + j < out) g[j] = h[j];             // zdot rows
            }
        }, img + m.b_off[l]);
        am_barrier();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    // scalar rows: zdot in columns 0..15, J eps in columns 16..31 of S[cur]
    {
        float e2 = 0.f, n2 = 0.f, dot = 0.f;
        for (int r = ec; r < n_in; r += AM_EC) {
            const float z = lds[cur + es * PX + r], je = lds[cur + (16 + es) * PX + r];
            e2 = fmaf(z, z, e2); n2 = fmaf(je, je, n2); dot = fmaf(je, lds[jl.off_E + es * PE + r], dot);
        }
        red[(0 * AM_EC + ec) * AM_NS + es] = e2;
        red[(1 * AM_EC + ec) * AM_NS + es] = n2;
        red[(2 * AM_EC + ec) * AM_NS + es] = dot;
    }
    am_barrier();
    if (tid < AM_NS && b0 + tid < a.B) {
        float e2 = 0.f, n2 = 0.f, dot = 0.f;
        for (int p = 0; p < AM_EC; ++p) {
            e2 += red[(0 * AM_EC + p) * AM_NS + tid]; n2 += red[(1 * AM_EC + p) * AM_NS + tid];
            dot += red[(2 * AM_EC + p) * AM_NS + tid];
        }
        float* g = du + (size_t)(b0 + tid) * D + n_in;
        g[0] = -dot;                                            // src/icnf.jl:404
        g[1] = nd.norm_z ? sqrtf(e2) : 0.f;                     // :405-411
        g[2] = nd.norm_j ? sqrtf(n2) : 0.f;                     // :412-413
    }
}

hipError_t launch_jvp_mfma(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                           const TraceArgs& a, const float* eps, hipStream_t s) {
    const JvpLayout jl = jvp_layout(nd, m);
    const size_t lds = (size_t)jl.total_floats * sizeof(float);
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    const void* fn = all_tanh ? (const void*)k_jvp_mfma<true> : (const void*)k_jvp_mfma<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid((a.B + AM_NS - 1) / AM_NS), block(AM_THREADS);
    if (all_tanh) hipLaunchKernelGGL(k_jvp_mfma<true>, grid, block, lds, s, nd, g, m, jl, img, a, eps);
    else hipLaunchKernelGGL(k_jvp_mfma<false>, grid, block, lds, s, nd, g, m, jl, img, a, eps);
    return hipGetLastError();
}

// fused norms / the fused six-stage step exist in the split kernel of the 32-128-128-32 shape only
bool trace_fused_supported(const NetDesc& nd, const AdjMfmaLayout& m, int B) {
    static const bool generic_only = [] { const char* e = getenv("CNF_TRACE_GENERIC"); return e && e[0] == '1'; }();
    static const bool trace_fp32 = [] { const char* e = getenv("CNF_TRACE_FP32"); return e && e[0] == '1'; }();
    static const bool unfused = [] { const char* e = getenv("CNF_TRACE_UNFUSED"); return e && e[0] == '1'; }();
    return trace3_shape(nd, m) && !generic_only && !trace_fp32 && !unfused && trace_fused_grid(B) <= 1024;
}
// k_trace3s: 32 samples per workgroup once that still gives every CU of an MI355X a workgroup, 16 below (measured, config 3:
// B = 8192: 59 against 67 us per evaluation; B = 4096: 52 against 35)
static int trace3s_ns(int B) { return B >= 32 * 256 ? 32 : 16; }
int trace_fused_grid(int B) { const int ns = trace3s_ns(B); return (B + ns - 1) / ns; }

// The whole TestMode solve of the 32-128-128-32 network in one launch (k_trace3s<SOLVE>): the state in a.U[0], the final state
// in a.U[cur]; a.st_mut receives the final integrator state, sv carries the initial state and the meeting words.  The grid
// must be resident at once: one workgroup per CU.
bool trace_solve_supported(const NetDesc& nd, const AdjMfmaLayout& m, int B, int device) {
    if (!trace_fused_supported(nd, m, B)) return false;
    static const bool off = [] { const char* e = getenv("CNF_PERSISTENT"); const char* w = getenv("CNF_TRACE_SOLVE"); return (e && e[0] == '0') || (w && w[0] == '0'); }();
    if (off) return false;
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return false; }
    return trace_fused_grid(B) <= n_cu && trace_fused_grid(B) <= AM_THREADS;
}
hipError_t launch_trace_solve(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                              const TraceArgs& a, const Solve3Args& sv, hipStream_t s) {
    const TraceLayout tl = trace_layout(nd, m);
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    const int PSf = pad8m16(m.maxd);
    const int ns = trace3s_ns(a.B);
    const size_t lds = (size_t)(ns * tl.PD + 2 * ns * PSf + AM_WAVES * ns * (32 + 4) + ns * AM_WAVES + 32) * sizeof(float);
    const dim3 grid((a.B + ns - 1) / ns), block(AM_THREADS);
    const void* fn;
    if (ns == 32) fn = all_tanh ? (const void*)k_trace3s<true, false, 32, 32, 128, 128, true> : (const void*)k_trace3s<false, false, 32, 32, 128, 128, true>;
    else fn = all_tanh ? (const void*)k_trace3s<true, false, 16, 32, 128, 128, true> : (const void*)k_trace3s<false, false, 16, 32, 128, 128, true>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    NetDesc nd_ = nd; GradLayout g_ = g; AdjMfmaLayout m_ = m; TraceLayout tl_ = tl; TraceArgs a_ = a; Solve3Args sv_ = sv;
    void* args[] = {&nd_, &g_, &m_, &tl_, &img, &a_, &sv_};
    return hipLaunchKernel(fn, grid, block, args, lds, s);
}

hipError_t launch_trace_mfma(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                             const TraceArgs& a, hipStream_t s) {
    const TraceLayout tl = trace_layout(nd, m);
    if ((a.norm_kind >= 0 || a.fused_step) && !trace_fused_supported(nd, m, a.B)) return hipErrorInvalidValue;
    const size_t lds_generic = (size_t)tl.total_floats * sizeof(float), lds = lds_generic;
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    const dim3 grid((a.B + AM_NS - 1) / AM_NS), block(AM_THREADS);
    static const bool generic_only = [] { const char* e = getenv("CNF_TRACE_GENERIC"); return e && e[0] == '1'; }();
    static const bool trace_fp32 = [] { const char* e = getenv("CNF_TRACE_FP32"); return e && e[0] == '1'; }();
    if (trace3_shape(nd, m) && !generic_only && !trace_fp32) {      // split-bf16 products (A/B switch: CNF_TRACE_FP32=1 -> k_trace3)
        const int PSf = pad8m16(m.maxd);
        const int ns = trace3s_ns(a.B);                  // samples per workgroup
        const size_t lds = (size_t)(ns * tl.PD + 2 * ns * PSf + AM_WAVES * ns * (32 + 4) + ns * AM_WAVES + 32) * sizeof(float);
        const dim3 grid((a.B + ns - 1) / ns);
        const bool step = a.fused_step != 0;
        const void* fn;
        if (ns == 32) fn = all_tanh ? (step ? (const void*)k_trace3s<true, true, 32, 32, 128, 128> : (const void*)k_trace3s<true, false, 32, 32, 128, 128>)
                                    : (step ? (const void*)k_trace3s<false, true, 32, 32, 128, 128> : (const void*)k_trace3s<false, false, 32, 32, 128, 128>);
        else fn = all_tanh ? (step ? (const void*)k_trace3s<true, true, 16, 32, 128, 128> : (const void*)k_trace3s<true, false, 16, 32, 128, 128>)
                           : (step ? (const void*)k_trace3s<false, true, 16, 32, 128, 128> : (const void*)k_trace3s<false, false, 16, 32, 128, 128>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        NetDesc nd_ = nd; GradLayout g_ = g; AdjMfmaLayout m_ = m; TraceLayout tl_ = tl; TraceArgs a_ = a;
        Solve3Args sv_{};
        void* args[] = {&nd_, &g_, &m_, &tl_, &img, &a_, &sv_};
        return hipLaunchKernel(fn, grid, block, args, lds, s);
    }
    if (trace3_shape(nd, m) && !generic_only) {          // resident-fragment kernel (A/B switch: CNF_TRACE_GENERIC=1)
        // forward images + per-wave partials of the last layer; later the per-lane trace partials (same area)
        const int PSf = pad8m16(m.maxd), fwd = 2 * AM_NS * PSf + AM_WAVES * AM_NS * (32 + 4) + 32 * (128 + 8),
                  tr = AM_NS * AM_THREADS;
        const size_t need = (size_t)(tl.off_T0 + (fwd > tr ? fwd : tr)) * sizeof(float);
        const size_t lds = lds_generic > need ? lds_generic : need;
        hipError_t e = hipFuncSetAttribute(all_tanh ? (const void*)k_trace3<true, 32, 128, 128>
                                                    : (const void*)k_trace3<false, 32, 128, 128>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (all_tanh) hipLaunchKernelGGL((k_trace3<true, 32, 128, 128>), grid, block, lds, s, nd, g, m, tl, img, a);
        else hipLaunchKernelGGL((k_trace3<false, 32, 128, 128>), grid, block, lds, s, nd, g, m, tl, img, a);
        return hipGetLastError();
    }
#define TR_LAUNCH(T, N)                                                                                             \
    do {                                                                                                            \
        hipError_t e = hipFuncSetAttribute((const void*)k_trace_mfma<T, N>,                                         \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((k_trace_mfma<T, N>), grid, block, lds, s, nd, g, m, tl, img, a);                        \
    } while (0)
    switch (tl.nct) {
        case 1: if (all_tanh) TR_LAUNCH(true, 1); else TR_LAUNCH(false, 1); break;
        case 2: if (all_tanh) TR_LAUNCH(true, 2); else TR_LAUNCH(false, 2); break;
        case 3: if (all_tanh) TR_LAUNCH(true, 3); else TR_LAUNCH(false, 3); break;
        case 4: if (all_tanh) TR_LAUNCH(true, 4); else TR_LAUNCH(false, 4); break;
        case 5: if (all_tanh) TR_LAUNCH(true, 5); else TR_LAUNCH(false, 5); break;
        case 6: if (all_tanh) TR_LAUNCH(true, 6); else TR_LAUNCH(false, 6); break;
        case 7: if (all_tanh) TR_LAUNCH(true, 7); else TR_LAUNCH(false, 7); break;
        default: if (all_tanh) TR_LAUNCH(true, 8); else TR_LAUNCH(false, 8); break;
    }
#undef TR_LAUNCH
    return hipGetLastError();
}
