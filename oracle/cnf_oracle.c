/* C restatement (float32, OpenMP over sample blocks) of the batched augmented-ODE hot path
 * of ContinuousNormalizingFlows.jl.  TEST INFRASTRUCTURE ONLY: used as a second checker
 * and as bench.py's `cpu_baseline` (kind "port").  Never linked into libcnfhip and never
 * imported by the product path.  PARITY UNPINNED against the Julia package itself (no
 * Julia in the build container, no golden vectors in the reference): it is pinned against
 * oracle/cnf_oracle.py, which is pinned by tests/test_oracle.py.
 *
 * Follows (file:line under /root/reference): augmented_f Matrix/Train/VJP src/icnf.jl:318-350,
 * Matrix/Train/JVP :384-420, Matrix/Test :148-164 + jacobian_batched src/utils.jl:1-36,
 * inference_sol src/base_icnf.jl:167-189.  Dense/AD sweeps/Tsit5 are third-party in the
 * reference and restated from their definitions.
 * Layout: Julia column-major D x B (sample b's rows contiguous at b*D). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NB 16          /* samples per block: inner loops vectorise over this axis */
#define MAXL 8

typedef struct {
    int n_layers;
    int dims[MAXL + 1];
    int acts[MAXL];
    int nvars, naugs;
    int norm_z, norm_j, norm_z_aug, jvp;
} oc_net;

static inline float sigm(float a) { return 1.0f / (1.0f + expf(-a)); }

static inline void act(int kind, float a, float* h, float* d) {
    switch (kind) {
        case 0: *h = a; *d = 1.0f; break;
        case 1: { float t = tanhf(a); *h = t; *d = 1.0f - t * t; } break;
        case 2: { float s = sigm(a); *h = s; *d = s * (1.0f - s); } break;
        case 3: *h = (a > 15.0f) ? a : log1pf(expf(a)); *d = sigm(a); break;
        case 4: *h = a > 0 ? a : 0; *d = a > 0 ? 1.0f : 0.0f; break;
        case 5: { float s = sigm(a); *h = a * s; *d = s * (1.0f + a * (1.0f - s)); } break;
        default: { float e = expf(a < 0 ? a : 0); *h = a > 0 ? a : e - 1.0f; *d = a > 0 ? 1.0f : e; }
    }
}

/* y[o][:] = sum_k W[o + k*out] x[k][:]  (W column-major out x in).  Four output rows share every load of x[k][:] (register
   blocking: the one-row form is bound by its loads); each accumulator still adds its products in the order k = 0, 1, ..:
   results are the same bit for bit. */
static void gemm_fwd(const float* W, int out, int in, const float (*x)[NB], float (*y)[NB]) {
    int o = 0;
    for (; o + 4 <= out; o += 4) {
        float a0[NB], a1[NB], a2[NB], a3[NB];
        for (int j = 0; j < NB; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; a3[j] = 0.f; }
        for (int k = 0; k < in; ++k) {
            const float* w = W + o + (size_t)k * out;
            const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            for (int j = 0; j < NB; ++j) {
                const float xv = x[k][j];
                a0[j] += w0 * xv; a1[j] += w1 * xv; a2[j] += w2 * xv; a3[j] += w3 * xv;
            }
        }
        for (int j = 0; j < NB; ++j) { y[o][j] = a0[j]; y[o + 1][j] = a1[j]; y[o + 2][j] = a2[j]; y[o + 3][j] = a3[j]; }
    }
    for (; o < out; ++o) {
        float acc[NB];
        for (int j = 0; j < NB; ++j) acc[j] = 0.f;
        for (int k = 0; k < in; ++k) {
            const float w = W[o + (size_t)k * out];
            for (int j = 0; j < NB; ++j) acc[j] += w * x[k][j];
        }
        for (int j = 0; j < NB; ++j) y[o][j] = acc[j];
    }
}
/* y[k][:] = sum_o W[o + k*out] g[o][:]  (four input columns share every load of g[o][:]) */
static void gemm_bwd(const float* W, int out, int in, const float (*g)[NB], float (*y)[NB]) {
    int k = 0;
    for (; k + 4 <= in; k += 4) {
        float a0[NB], a1[NB], a2[NB], a3[NB];
        for (int j = 0; j < NB; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; a3[j] = 0.f; }
        const float *c0 = W + (size_t)k * out, *c1 = c0 + out, *c2 = c1 + out, *c3 = c2 + out;
        for (int o = 0; o < out; ++o) {
            const float w0 = c0[o], w1 = c1[o], w2 = c2[o], w3 = c3[o];
            for (int j = 0; j < NB; ++j) {
                const float gv = g[o][j];
                a0[j] += w0 * gv; a1[j] += w1 * gv; a2[j] += w2 * gv; a3[j] += w3 * gv;
            }
        }
        for (int j = 0; j < NB; ++j) { y[k][j] = a0[j]; y[k + 1][j] = a1[j]; y[k + 2][j] = a2[j]; y[k + 3][j] = a3[j]; }
    }
    for (; k < in; ++k) {
        float acc[NB];
        for (int j = 0; j < NB; ++j) acc[j] = 0.f;
        const float* w = W + (size_t)k * out;
        for (int o = 0; o < out; ++o)
            for (int j = 0; j < NB; ++j) acc[j] += w[o] * g[o][j];
        for (int j = 0; j < NB; ++j) y[k][j] = acc[j];
    }
}

static int maxdim(const oc_net* n) {
    int m = 0;
    for (int l = 0; l <= n->n_layers; ++l) m = n->dims[l] > m ? n->dims[l] : m;
    return m;
}
static int sumdim(const oc_net* n) {
    int s = 0;
    for (int l = 0; l <= n->n_layers; ++l) s += n->dims[l];
    return s;
}

/* one block of NB samples starting at column b0 (nb valid) */
static void rhs_block(const oc_net* net, const float* P, const float* u, const float* eps,
                      float* du, int b0, int nb, int train, float* scratch) {
    const int L = net->n_layers, n_in = net->dims[0];
    const int D = n_in + 1 + (train ? 2 : 0);
    const int S = sumdim(net), M = maxdim(net);
    float (*H)[NB] = (float (*)[NB])scratch;            /* S rows */
    float (*Dv)[NB] = H + S;                            /* S rows */
    float (*G0)[NB] = Dv + S;                           /* M rows */
    float (*G1)[NB] = G0 + M;                           /* M rows */
    for (int r = 0; r < n_in; ++r)
        for (int j = 0; j < NB; ++j) H[r][j] = j < nb ? u[(size_t)(b0 + j) * D + r] : 0.f;
    int off = 0, woff = 0;
    int woffs[MAXL], boffs[MAXL], hoffs[MAXL + 1];
    for (int l = 0; l < L; ++l) {
        const int in = net->dims[l], out = net->dims[l + 1];
        woffs[l] = woff; boffs[l] = woff + in * out; woff += in * out + out;
        hoffs[l] = off;
        gemm_fwd(P + woffs[l], out, in, (const float (*)[NB])(H + off), H + off + in);
        for (int o = 0; o < out; ++o)
            for (int j = 0; j < NB; ++j) {
                float h, d;
                act(net->acts[l], H[off + in + o][j] + P[boffs[l] + o], &h, &d);
                H[off + in + o][j] = h;
                Dv[off + in + o][j] = d;
            }
        off += in;
    }
    hoffs[L] = off;
    float ldot[NB], nsq[NB], esq[NB];
    for (int j = 0; j < NB; ++j) { ldot[j] = 0.f; nsq[j] = 0.f; esq[j] = 0.f; }
    for (int i = 0; i < n_in; ++i)
        for (int j = 0; j < NB; ++j) {
            float v = H[off + i][j];
            if (j < nb) du[(size_t)(b0 + j) * D + i] = v;
            esq[j] += v * v;
        }
    if (train) {
        float e[NB];
        if (!net->jvp) {
            for (int i = 0; i < n_in; ++i)
                for (int j = 0; j < NB; ++j)
                    G0[i][j] = (j < nb ? eps[(size_t)(b0 + j) * n_in + i] : 0.f) * Dv[off + i][j];
            float (*g)[NB] = G0, (*gn)[NB] = G1;
            for (int l = L - 1; l >= 0; --l) {
                const int in = net->dims[l], out = net->dims[l + 1];
                gemm_bwd(P + woffs[l], out, in, (const float (*)[NB])g, gn);
                if (l > 0)
                    for (int k = 0; k < in; ++k)
                        for (int j = 0; j < NB; ++j) gn[k][j] *= Dv[hoffs[l] + k][j];
                float (*t)[NB] = g; g = gn; gn = t;
            }
            for (int i = 0; i < n_in; ++i) {
                for (int j = 0; j < NB; ++j) e[j] = j < nb ? eps[(size_t)(b0 + j) * n_in + i] : 0.f;
                for (int j = 0; j < NB; ++j) { ldot[j] -= g[i][j] * e[j]; nsq[j] += g[i][j] * g[i][j]; }
            }
        } else {
            for (int i = 0; i < n_in; ++i)
                for (int j = 0; j < NB; ++j) G0[i][j] = j < nb ? eps[(size_t)(b0 + j) * n_in + i] : 0.f;
            float (*t0)[NB] = G0, (*t1)[NB] = G1;
            for (int l = 0; l < L; ++l) {
                const int in = net->dims[l], out = net->dims[l + 1];
                gemm_fwd(P + woffs[l], out, in, (const float (*)[NB])t0, t1);
                for (int o = 0; o < out; ++o)
                    for (int j = 0; j < NB; ++j) t1[o][j] *= Dv[hoffs[l] + in + o][j];
                float (*t)[NB] = t0; t0 = t1; t1 = t;
            }
            for (int i = 0; i < n_in; ++i) {
                for (int j = 0; j < NB; ++j) e[j] = j < nb ? eps[(size_t)(b0 + j) * n_in + i] : 0.f;
                for (int j = 0; j < NB; ++j) { ldot[j] -= t0[i][j] * e[j]; nsq[j] += t0[i][j] * t0[i][j]; }
            }
        }
        for (int j = 0; j < nb; ++j) {
            float* c = du + (size_t)(b0 + j) * D;
            c[n_in] = ldot[j];
            c[n_in + 1] = net->norm_z ? sqrtf(esq[j]) : 0.f;
            c[n_in + 2] = net->norm_j ? sqrtf(nsq[j]) : 0.f;
        }
    } else {
        /* exact trace: n_in one-hot tangent sweeps (src/utils.jl:19-36) */
        float tr[NB];
        for (int j = 0; j < NB; ++j) tr[j] = 0.f;
        for (int i = 0; i < n_in; ++i) {
            float (*t0)[NB] = G0, (*t1)[NB] = G1;
            for (int k = 0; k < n_in; ++k)
                for (int j = 0; j < NB; ++j) t0[k][j] = k == i ? 1.f : 0.f;
            for (int l = 0; l < L; ++l) {
                const int in = net->dims[l], out = net->dims[l + 1];
                gemm_fwd(P + woffs[l], out, in, (const float (*)[NB])t0, t1);
                for (int o = 0; o < out; ++o)
                    for (int j = 0; j < NB; ++j) t1[o][j] *= Dv[hoffs[l] + in + o][j];
                float (*t)[NB] = t0; t0 = t1; t1 = t;
            }
            for (int j = 0; j < NB; ++j) tr[j] += t0[i][j];
        }
        for (int j = 0; j < nb; ++j) du[(size_t)(b0 + j) * D + n_in] = -tr[j];
    }
}

static size_t scratch_floats(const oc_net* net) {
    return (size_t)(2 * sumdim(net) + 2 * maxdim(net)) * NB;
}

/* threads of the following parallel regions (bench.py: the cores the process may run on, not the host's count) */
void oc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int oc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oc_rhs(const oc_net* net, const float* P, const float* u, const float* eps, float* du,
            int B, int train) {
    const int nblk = (B + NB - 1) / NB;
#pragma omp parallel
    {
        float* scratch = (float*)aligned_alloc(64, ((scratch_floats(net) * sizeof(float) + 63) / 64) * 64);
#pragma omp for schedule(static)
        for (int blk = 0; blk < nblk; ++blk) {
            int b0 = blk * NB, nb = B - b0 < NB ? B - b0 : NB;
            rhs_block(net, P, u, eps, du, b0, nb, train, scratch);
        }
        free(scratch);
    }
}

/* ---- Tsit5 ------------------------------------------------------------------------- */
static const float A_[7][6] = {
    {0},
    {0.161f},
    {-0.008480655492356989f, 0.335480655492357f},
    {2.8971530571054935f, -6.359448489975075f, 4.3622954328695815f},
    {5.325864828439257f, -11.748883564062828f, 7.4955393428898365f, -0.09249506636175525f},
    {5.86145544294642f, -12.92096931784711f, 8.159367898576159f, -0.071584973281401f, -0.028269050394068383f},
    {0.09646076681806523f, 0.01f, 0.4798896504144996f, 1.379008574103742f, -3.290069515436081f, 2.324710524099774f}};
static const float BT_[7] = {-0.00178001105222577714f, -0.0008164344596567469f, 0.007880878010261995f,
                             -0.1447110071732629f, 0.5823571654525552f, -0.45808210592918697f,
                             0.015151515151515152f};

typedef struct { int nf, naccept, nreject; float t_final, dt_last; } oc_stats;

/* optional log of the step attempts of the next solves: 4 floats each, (t, signed h, EEst, accepted) */
static float* g_trace = 0;
static int g_trace_cap = 0;
void oc_set_trace(float* buf, int cap_attempts) { g_trace = buf; g_trace_cap = buf ? cap_attempts : 0; }

static double rms2(const float* x, const float* sk, size_t n) { /* sum (x/sk)^2 */
    double s = 0;
#pragma omp parallel for reduction(+ : s)
    for (size_t i = 0; i < n; ++i) { double v = x[i] / sk[i]; s += v * v; }
    return s;
}

/* returns 0 ok, 5 maxiters */
int oc_solve_tsit5(const oc_net* net, const float* P, const float* u0, const float* eps,
                   float* u_out, int B, int train, float t0, float t1, float abstol,
                   float reltol, float dt, int adaptive, int maxiters, oc_stats* st) {
    const int D = net->dims[0] + 1 + (train ? 2 : 0);
    const size_t n = (size_t)D * B;
    float* buf = (float*)malloc(sizeof(float) * n * 11);
    float *u = buf, *un = buf + n, *us = buf + 2 * n, *sk = buf + 3 * n;
    float* k[7];
    for (int i = 0; i < 7; ++i) k[i] = buf + (4 + i) * n;
    memcpy(u, u0, n * sizeof(float));
    const float tdir = t1 >= t0 ? 1.f : -1.f;
    float t = t0;
    int nf = 0, nacc = 0, nrej = 0;
    oc_rhs(net, P, u, eps, k[0], B, train); nf++;
    if (adaptive && dt == 0.f) {
        for (size_t i = 0; i < n; ++i) sk[i] = abstol + fabsf(u[i]) * reltol;
        double d0 = sqrt(rms2(u, sk, n) / n), d1 = sqrt(rms2(k[0], sk, n) / n);
        float dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6f : (float)(0.01 * d0 / d1);
        dt0 = fminf(dt0, fabsf(t1 - t0));
        for (size_t i = 0; i < n; ++i) us[i] = u[i] + tdir * dt0 * k[0][i];
        oc_rhs(net, P, us, eps, k[1], B, train); nf++;
        for (size_t i = 0; i < n; ++i) un[i] = k[1][i] - k[0][i];
        double d2 = sqrt(rms2(un, sk, n) / n) / dt0;
        double m = d1 > d2 ? d1 : d2;
        float dt1 = m <= 1e-15 ? fmaxf(1e-6f, dt0 * 1e-3f) : (float)pow(0.01 / m, 0.2);
        dt = fminf(fminf(100.f * dt0, dt1), fabsf(t1 - t0));
    }
    float qold = 1e-4f;
    int rc = 5;
    for (int it = 0; it < maxiters; ++it) {
        float rem = fabsf(t1 - t);
        float habs = dt < rem ? dt : rem;
        float h = tdir * habs;
        for (int s = 1; s <= 6; ++s) {
#pragma omp parallel for
            for (size_t i = 0; i < n; ++i) {
                float acc = 0.f;
                for (int j = 0; j < s; ++j) acc = fmaf(A_[s][j], k[j][i], acc);
                us[i] = fmaf(h, acc, u[i]);
            }
            oc_rhs(net, P, us, eps, k[s], B, train); nf++;
        }
        /* us == u_new (a7 = b) */
        int accept = 1;
        float q = 1.f, q11 = 1.f, eest = 0.f;
        if (adaptive) {
            double s2 = 0;
#pragma omp parallel for reduction(+ : s2)
            for (size_t i = 0; i < n; ++i) {
                float e = 0.f;
                for (int j = 0; j < 7; ++j) e = fmaf(BT_[j], k[j][i], e);
                e *= h;
                float sc = abstol + fmaxf(fabsf(u[i]), fabsf(us[i])) * reltol;
                double v = e / sc;
                s2 += v * v;
            }
            eest = (float)sqrt(s2 / n);
            accept = eest <= 1.0f;
            q11 = powf(fmaxf(eest, 1e-30f), 7.f / 50.f);
            q = q11 / powf(qold, 2.f / 25.f);
            q = fmaxf(0.1f, fminf(5.f, q / 0.9f));
        }
        if (g_trace && it < g_trace_cap) {
            g_trace[4 * it] = t; g_trace[4 * it + 1] = h; g_trace[4 * it + 2] = eest; g_trace[4 * it + 3] = (float)accept;
        }
        if (accept) {
            nacc++;
            t += h;
            float* tmp = u; u = us; us = tmp;
            tmp = k[0]; k[0] = k[6]; k[6] = tmp;
            if (adaptive) {
                if (q >= 1.f && q <= 1.2f) q = 1.f;
                qold = fmaxf(eest, 1e-4f);
                dt = habs / q;
            }
            if (fabsf(t1 - t) <= 100.f * 1.1920929e-7f * fmaxf(1.f, fabsf(t1))) { t = t1; rc = 0; break; }
        } else {
            nrej++;
            dt = habs / fminf(5.f, q11 / 0.9f);
        }
    }
    memcpy(u_out, u, n * sizeof(float));
    if (st) { st->nf = nf; st->naccept = nacc; st->nreject = nrej; st->t_final = t; st->dt_last = dt; }
    free(buf);
    return rc;
}

/* inference_sol (src/base_icnf.jl:167-189): regs is 3 x B row-major (E, n, A) */
void oc_post(const oc_net* net, const float* fsol, float* logpx, float* regs, int B, int train) {
    const int n_in = net->dims[0], D = n_in + 1 + (train ? 2 : 0);
    for (int b = 0; b < B; ++b) {
        const float* c = fsol + (size_t)b * D;
        float ss = 0.f, sa = 0.f;
        for (int i = 0; i < n_in; ++i) { ss += c[i] * c[i]; if (i >= net->nvars) sa += c[i] * c[i]; }
        logpx[b] = -0.5f * (n_in * 1.8378770664093453f + ss) - c[n_in];
        regs[b] = train ? c[n_in + 1] : 0.f;
        regs[B + b] = train ? c[n_in + 2] : 0.f;
        regs[2 * (size_t)B + b] = (net->norm_z_aug && net->naugs > 0) ? sqrtf(sa) : 0.f;
    }
}
