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
#include "cnf_am.h"

#define TR_NCMAX 8

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
    am_first(pf, img + m.f_off[0], m.dp[1], m.dp[0]);
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
        am_gemm(img + m.f_off[l], m.dp[l + 1], m.dp[l], lds + cur, PSf, pf,
                img + (last ? m.f_off[1] : m.f_off[l + 1]), last ? m.dp[2] : m.dp[l + 2], last ? m.dp[1] : m.dp[l + 1],
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
            const float* nimg = lastmid ? img + m.f_off[1] : img + m.f_off[l + 1];
            const int nr = lastmid ? m.dp[2] : m.dp[l + 2], nk = lastmid ? m.dp[1] : m.dp[l + 1];
            am_gemm_multi<NCT>(img + m.f_off[l], m.dp[l + 1], m.dp[l], lds + tc, PT, pf, nimg, nr, nk,
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
    am_first(pf, img + m.f_off[0], m.dp[1], m.dp[0]);
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
        am_gemm_multi<2>(img + m.f_off[l], m.dp[l + 1], m.dp[l], lds + cur, PX, pf,
                         last ? nullptr : img + m.f_off[l + 1], last ? 0 : m.dp[l + 2], last ? 0 : m.dp[l + 1],
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

hipError_t launch_trace_mfma(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                             const TraceArgs& a, hipStream_t s) {
    const TraceLayout tl = trace_layout(nd, m);
    const size_t lds_generic = (size_t)tl.total_floats * sizeof(float), lds = lds_generic;
    bool all_tanh = true;
    for (int l = 0; l < nd.n_layers; ++l) all_tanh = all_tanh && nd.acts[l] == 1;
    const dim3 grid((a.B + AM_NS - 1) / AM_NS), block(AM_THREADS);
    static const bool generic_only = [] { const char* e = getenv("CNF_TRACE_GENERIC"); return e && e[0] == '1'; }();
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
