// Generic (any layer sizes, any activation) kernels of libcnfhip: one lane per sample,
// activations in an HBM/L2 workspace laid out feature-major ([feature][sample], so a
// wave's 64 lanes read 64 consecutive floats), weights read through scalar loads (they are
// wave-uniform).  This is the shape-agnostic path; the MFMA path (cnf_mfma.hip) takes over
// for the layer sizes it is built for.  Also here: the Tsit5 bookkeeping kernels shared by
// both paths (error norms, on-device step controller, post-processing, loss sums).
//
// Reference functions restated (file:line under /root/reference):
//   augmented_f Matrix/Train/VJP  src/icnf.jl:318-350     Matrix/Train/JVP  src/icnf.jl:384-420
//   augmented_f Matrix/Test       src/icnf.jl:148-164  +  jacobian_batched  src/utils.jl:1-36
//   inference_sol                 src/base_icnf.jl:167-189      loss  src/icnf.jl:481-490
#include "cnf_dev.h"
#include "cnf_kernels.h"

// ---------------------------------------------------------------------------------------
// dense layer helpers (lane = sample b; W is out x in, column-major: W[o + k*out])
// ---------------------------------------------------------------------------------------
// y[o] = sum_k W[o,k] x[k]   for o in [0,out), written via `emit(o, acc)`
template <typename Emit>
__device__ __forceinline__ void gemv_fwd(const float* __restrict__ W, int out, int in,
                                         const float* __restrict__ x, size_t xs, Emit emit) {
    int o = 0;
    for (; o + 4 <= out; o += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < in; ++k) {
            float xv = x[(size_t)k * xs];
            const float* w = W + o + (size_t)k * out;
            a0 = fmaf(w[0], xv, a0); a1 = fmaf(w[1], xv, a1);
            a2 = fmaf(w[2], xv, a2); a3 = fmaf(w[3], xv, a3);
        }
        emit(o, a0); emit(o + 1, a1); emit(o + 2, a2); emit(o + 3, a3);
    }
    for (; o < out; ++o) {
        float a = 0.f;
        for (int k = 0; k < in; ++k) a = fmaf(W[o + (size_t)k * out], x[(size_t)k * xs], a);
        emit(o, a);
    }
}
// y[k] = sum_o W[o,k] g[o]   for k in [0,in)
template <typename Emit>
__device__ __forceinline__ void gemv_bwd(const float* __restrict__ W, int out, int in,
                                         const float* __restrict__ g, size_t gs, Emit emit) {
    int k = 0;
    for (; k + 4 <= in; k += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float* w0 = W + (size_t)k * out;
        for (int o = 0; o < out; ++o) {
            float gv = g[(size_t)o * gs];
            a0 = fmaf(w0[o], gv, a0); a1 = fmaf(w0[o + out], gv, a1);
            a2 = fmaf(w0[o + 2 * out], gv, a2); a3 = fmaf(w0[o + 3 * out], gv, a3);
        }
        emit(k, a0); emit(k + 1, a1); emit(k + 2, a2); emit(k + 3, a3);
    }
    for (; k < in; ++k) {
        float a = 0.f;
        const float* w0 = W + (size_t)k * out;
        for (int o = 0; o < out; ++o) a = fmaf(w0[o], g[(size_t)o * gs], a);
        emit(k, a);
    }
}

// ---------------------------------------------------------------------------------------
// RHS, generic.  Optional fused Tsit5 stage prologue:
//   u_stage = u + h * sum_j coef[j] * k_j   (nk = 0: u_stage = u)
// ws layout (floats, S = padded sample count of the launch):
//   H  : sum_dims * S   activations h_0..h_L       (h_l at hoff[l]*S)
//   Dv : sum_dims * S   sigma'(a_l) for l = 1..L   (same offsets, slot 0 unused)
//   G  : 2*max_dim * S  ping-pong cotangent / tangent
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_rhs_generic(NetDesc nd, const float* __restrict__ P, RhsArgs a) {
    const StepState* st = a.st;
    if (st && st->done) return;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int train = a.train;
    const int n_in = nd.n_in;
    const int D = n_in + 1 + (train ? 2 : 0);
    const size_t S = a.S;
    const int L = nd.n_layers;
    float* H = a.ws;
    float* Dv = a.ws + (size_t)nd.sum_dims * S;
    float* G0 = Dv + (size_t)nd.sum_dims * S;
    float* G1 = G0 + (size_t)nd.max_dim * S;

    // ---- prologue: stage state ------------------------------------------------------
    const float* u;
    const float* kb[6];
    float hstep = 0.f;
    if (st) {
        const int cur = st->cur;
        u = (cur ? a.U[1] : a.U[0]);
        kb[0] = (cur ? a.K1[1] : a.K1[0]);
        for (int j = 1; j < 6; ++j) kb[j] = a.Ks[j - 1];
        hstep = st->h;
    } else {
        u = a.u;
    }
    float* ust = a.ustage;
    if (st && a.ustage_is_unew) ust = (st->cur ? a.U[0] : a.U[1]);
    for (int r = 0; r < D; ++r) {
        float v = u[(size_t)b * D + r];
        if (a.nk > 0) {
            float acc = 0.f;
            for (int j = 0; j < a.nk; ++j) acc = fmaf(a.coef[j], kb[j][(size_t)b * D + r], acc);
            v = fmaf(hstep, acc, v);
        }
        if (r < n_in) H[(size_t)r * S + b] = v;
        if (ust) ust[(size_t)b * D + r] = v;
    }

    // ---- forward (a5: snn(z), src/icnf.jl:329,331) -------------------------------------
    int hoff = 0;
    for (int l = 0; l < L; ++l) {
        const int in = nd.dims[l], out = nd.dims[l + 1];
        const float* W = P + nd.w_off[l];
        const float* bias = P + nd.b_off[l];
        const float* x = H + (size_t)hoff * S + b;
        float* y = H + (size_t)(hoff + in) * S + b;
        float* dv = Dv + (size_t)(hoff + in) * S + b;
        const int act = nd.acts[l];
        // conditional models: layer 0 uses the per-sample bias W1[:, n_in:] ys + b1
        const float* cb = (l == 0 && a.cond) ? a.cond + (size_t)b * a.cbs : nullptr;
        gemv_fwd(W, out, in, x, S, [&](int o, float acc) {
            float h, d;
            cnf_act(act, acc + (cb ? cb[o] : bias[o]), h, d);
            y[(size_t)o * S] = h;
            dv[(size_t)o * S] = d;
        });
        hoff += in;
    }
    const float* zdot = H + (size_t)hoff * S + b;   // h_L, n_in entries
    const int hoffL = hoff;

    float* du = a.du;
    if (st && a.du_is_k7) du = (st->cur ? a.K1[0] : a.K1[1]);
    float* dub = du + (size_t)b * D;

    float ldot = 0.f, nsq = 0.f, esq = 0.f;
    for (int i = 0; i < n_in; ++i) {
        float v = zdot[(size_t)i * S];
        dub[i] = v;
        esq = fmaf(v, v, esq);
    }

    if (train) {
        const float* eps = a.eps + (size_t)b * n_in;
        if (!nd.jvp) {
            // ---- VJP sweep: g_L = eps .* d_L; g_{l-1} = (W_l^T g_l) .* d_{l-1}; eJ = W_1^T g_1
            float* g = G0 + b;
            float* gn = G1 + b;
            {
                const float* dv = Dv + (size_t)hoffL * S + b;
                for (int i = 0; i < n_in; ++i) g[(size_t)i * S] = eps[i] * dv[(size_t)i * S];
            }
            int ho = hoffL;
            for (int l = L - 1; l >= 0; --l) {
                const int in = nd.dims[l], out = nd.dims[l + 1];
                const float* W = P + nd.w_off[l];
                ho -= in;  // offset of h_l's input = h_{l}
                const float* dprev = Dv + (size_t)ho * S + b;  // sigma' of layer l-1's output (l>0)
                if (l > 0) {
                    gemv_bwd(W, out, in, g, S, [&](int k, float acc) {
                        gn[(size_t)k * S] = acc * dprev[(size_t)k * S];
                    });
                    float* t = g; g = gn; gn = t;
                } else {
                    gemv_bwd(W, out, in, g, S, [&](int k, float acc) {
                        ldot = fmaf(-acc, eps[k], ldot);     // icnf.jl:334
                        nsq = fmaf(acc, acc, nsq);           // icnf.jl:343
                    });
                }
            }
        } else {
            // ---- JVP sweep: tau_0 = eps; tau_l = d_l .* (W_l tau_{l-1}); J eps = tau_L
            float* tg = G0 + b;
            float* tn = G1 + b;
            for (int i = 0; i < n_in; ++i) tg[(size_t)i * S] = eps[i];
            int ho = 0;
            for (int l = 0; l < L; ++l) {
                const int in = nd.dims[l], out = nd.dims[l + 1];
                const float* W = P + nd.w_off[l];
                const float* dv = Dv + (size_t)(ho + in) * S + b;
                if (l < L - 1) {
                    gemv_fwd(W, out, in, tg, S, [&](int o, float acc) {
                        tn[(size_t)o * S] = acc * dv[(size_t)o * S];
                    });
                    float* t = tg; tg = tn; tn = t;
                } else {
                    gemv_fwd(W, out, in, tg, S, [&](int o, float acc) {
                        float je = acc * dv[(size_t)o * S];
                        ldot = fmaf(-je, eps[o], ldot);      // icnf.jl:404
                        nsq = fmaf(je, je, nsq);             // icnf.jl:413
                    });
                }
                ho += in;
            }
        }
        dub[n_in] = ldot;
        dub[n_in + 1] = nd.norm_z ? sqrtf(esq) : 0.f;         // icnf.jl:335-341
        dub[n_in + 2] = nd.norm_j ? sqrtf(nsq) : 0.f;         // icnf.jl:342-348
    } else {
        // ---- exact trace (a3/a4): n_in one-hot tangent sweeps, tr J = sum_i (J e_i)_i.
        // (utils.jl:19-36 fills column i from the pushforward of e_i; the pullback form
        // utils.jl:1-17 gives the same diagonal.)  J is never materialised.
        float tr = 0.f;
        for (int i = 0; i < n_in; ++i) {
            float* tg = G0 + b;
            float* tn = G1 + b;
            int ho = 0;
            for (int l = 0; l < L; ++l) {
                const int in = nd.dims[l], out = nd.dims[l + 1];
                const float* W = P + nd.w_off[l];
                const float* dv = Dv + (size_t)(ho + in) * S + b;
                if (l == 0) {
                    // tau_0 = e_i  ->  W_1 e_i = column i of W_1
                    if (L == 1) {
                        tr = fmaf(W[i + (size_t)i * out], dv[(size_t)i * S], tr);
                    } else {
                        for (int o = 0; o < out; ++o)
                            tn[(size_t)o * S] = W[o + (size_t)i * out] * dv[(size_t)o * S];
                    }
                } else if (l < L - 1) {
                    gemv_fwd(W, out, in, tg, S, [&](int o, float acc) {
                        tn[(size_t)o * S] = acc * dv[(size_t)o * S];
                    });
                } else {
                    // only output row i is needed
                    float acc = 0.f;
                    for (int k = 0; k < in; ++k) acc = fmaf(W[i + (size_t)k * out], tg[(size_t)k * S], acc);
                    tr = fmaf(acc, dv[(size_t)i * S], tr);
                }
                float* t = tg; tg = tn; tn = t;
                ho += in;
            }
        }
        dub[n_in] = -tr;                                       // icnf.jl:162
    }
}

// ---------------------------------------------------------------------------------------
// block reduction helper (256 threads), deterministic order
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}

// ---------------------------------------------------------------------------------------
// squared-norm partials over all D*B entries.
//  kind 0 (init A):  p0 = sum (u/sk)^2, p1 = sum (f0/sk)^2, sk = abstol + |u| reltol
//  kind 1 (init B):  p0 = sum ((f1 - f0)/sk)^2
//  kind 2 (step)  :  p0 = sum (err/sc)^2, err = h*sum_j btilde_j k_j,
//                    sc = abstol + max(|u|,|u_new|) reltol;  p1 = nonfinite count
// partials[2*blockIdx.x + {0,1}]
// ---------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_norm_partials(NormArgs a) {
    __shared__ float sm[8];
    const StepState* st = a.st;
    float p0 = 0.f, p1 = 0.f;
    if (!st->done) {
        const int cur = st->cur;
        const float* u = (cur ? a.U[1] : a.U[0]);
        const float* k1 = (cur ? a.K1[1] : a.K1[0]);
        const float abstol = st->abstol, reltol = st->reltol;
        const size_t n = a.n;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (size_t)gridDim.x * blockDim.x) {
            if (a.kind == 0) {
                float uv = u[i];
                float sk = fmaf(fabsf(uv), reltol, abstol);
                float x = uv / sk, y = k1[i] / sk;
                p0 = fmaf(x, x, p0);
                p1 = fmaf(y, y, p1);
            } else if (a.kind == 1) {
                float uv = u[i];
                float sk = fmaf(fabsf(uv), reltol, abstol);
                float x = (a.Ks[0][i] - k1[i]) / sk;
                p0 = fmaf(x, x, p0);
            } else {
                const float* un = (cur ? a.U[0] : a.U[1]);
                const float* k7 = (cur ? a.K1[0] : a.K1[1]);
                float e = TS_BT1 * k1[i];
                e = fmaf(TS_BT2, a.Ks[0][i], e);
                e = fmaf(TS_BT3, a.Ks[1][i], e);
                e = fmaf(TS_BT4, a.Ks[2][i], e);
                e = fmaf(TS_BT5, a.Ks[3][i], e);
                e = fmaf(TS_BT6, a.Ks[4][i], e);
                e = fmaf(TS_BT7, k7[i], e);
                e *= st->h;
                float uv = u[i], nv = un[i];
                float sc = fmaf(fmaxf(fabsf(uv), fabsf(nv)), reltol, abstol);
                float x = e / sc;
                p0 = fmaf(x, x, p0);
                if (!(fabsf(nv) <= 3.0e38f)) p1 += 1.f;
            }
        }
    }
    float s0 = block_sum(p0, sm);
    float s1 = block_sum(p1, sm);
    if (!a.ticket) {
        if (threadIdx.x == 0) {
            a.partials[2 * blockIdx.x] = s0;
            a.partials[2 * blockIdx.x + 1] = s1;
        }
        return;
    }
    // fused controller: partials and ticket go through agent-scope atomics (no cache flushes); the block
    // that draws the last ticket sums the partials in the fixed order of k_controller and runs the phase
    __shared__ int last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.partials + 2 * blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
        if (last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    if (st->done) {                       // queued past the end: keep the host's view current
        if (threadIdx.x == 0) mirror_store(a.mirror, a.seq, *st);
        return;
    }
    float q0 = 0.f, q1 = 0.f;
    for (int i = threadIdx.x; i < st->n_partials; i += blockDim.x) {
        q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    q0 = block_sum(q0, sm);
    q1 = block_sum(q1, sm);
    if (threadIdx.x == 0) {
        ctrl_phase(a.st_mut, a.ctrl_phase, q0, q1, a.n_total);
        mirror_store(a.mirror, a.seq, *a.st_mut);
    }
}

// ---------------------------------------------------------------------------------------
// on-device controller (one block).  phase 0: initial-dt part A; 1: part B; 2: after a step.
// Control law: OrdinaryDiffEq-style PI controller for Tsit5 (third
// party in the reference, restated from the published scheme, mirrored by the oracle).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_controller(StepState* st, const float* __restrict__ partials, int phase, float n_total) {
    __shared__ float sm[8];
    if (st->done) return;
    float p0 = 0.f, p1 = 0.f;
    for (int i = threadIdx.x; i < st->n_partials; i += blockDim.x) {
        p0 += partials[2 * i];
        p1 += partials[2 * i + 1];
    }
    // deterministic: fixed thread->entry map and fixed reduction tree
    p0 = block_sum(p0, sm);
    p1 = block_sum(p1, sm);
    if (threadIdx.x != 0) return;
    ctrl_phase(st, phase, p0, p1, n_total);
}

// Lock-step sharded solves (option 2): every shard reduces its partial sums to
// (p0, p1, n_local), the host sums the three numbers over the shards (cnf_set_shard_reduce), and
// every shard runs the controller on the identical global sums -> identical accept/reject and dt.
__global__ void __launch_bounds__(256)
k_reduce_partials(const StepState* st, const float* __restrict__ partials, float* __restrict__ out3,
                  float n_local) {
    __shared__ float sm[8];
    float p0 = 0.f, p1 = 0.f;
    for (int i = threadIdx.x; i < st->n_partials; i += blockDim.x) {
        p0 += partials[2 * i];
        p1 += partials[2 * i + 1];
    }
    p0 = block_sum(p0, sm);
    p1 = block_sum(p1, sm);
    if (threadIdx.x == 0) { out3[0] = p0; out3[1] = p1; out3[2] = n_local; }
}

__global__ void k_controller_sums(StepState* st, const float* __restrict__ sums3, int phase) {
    if (st->done) return;
    const float p0 = sums3[0], p1 = sums3[1], n_total = sums3[2];
    ctrl_phase(st, phase, p0, p1, n_total);
}

// ---------------------------------------------------------------------------------------
// small elementwise kernels
// ---------------------------------------------------------------------------------------
// u0 = vcat(xs, zeros(naugs + n_aug + 1, B))      src/base_icnf.jl:275-276, 282
// (st_dst: the launch also sets the integrator's initial state, handed over by value -- one launch less at the start of a
// solve, where the stream is empty and every launch is a host round trip)
__global__ void k_build_u0(const float* __restrict__ xs, float* __restrict__ u0, int nvars,
                           int D, int B, StepState* st_dst, StepState st_val) {
    if (st_dst && blockIdx.x == 0 && threadIdx.x == 0) *st_dst = st_val;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)D * B) return;
    int r = (int)(i % D);
    size_t b = i / D;
    u0[i] = r < nvars ? xs[b * nvars + r] : 0.f;
}

__global__ void k_copy_final(const StepState* st, const float* U0, const float* U1,
                             float* __restrict__ out, size_t n) {
    const float* src = st->cur ? U1 : U0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = src[i];
}

// inference_sol (src/base_icnf.jl:167-189), one lane per column
__global__ void k_post(NetDesc nd, int train, const float* __restrict__ fsol,
                       float* __restrict__ logpx, float* __restrict__ regs, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n_in = nd.n_in;
    const int D = n_in + 1 + (train ? 2 : 0);
    const float* c = fsol + (size_t)b * D;
    float ss = 0.f, sa = 0.f;
    for (int i = 0; i < n_in; ++i) {
        float v = c[i];
        ss = fmaf(v, v, ss);
        if (i >= nd.nvars) sa = fmaf(v, v, sa);
    }
    const float log2pi = 1.8378770664093453f;
    float logpz = -0.5f * fmaf((float)n_in, log2pi, ss);      // base_icnf.jl:177
    logpx[b] = logpz - c[n_in];                                // base_icnf.jl:178
    regs[b] = train ? c[n_in + 1] : 0.f;
    regs[(size_t)B + b] = train ? c[n_in + 2] : 0.f;
    regs[2 * (size_t)B + b] = (nd.norm_z_aug && nd.naugs > 0) ? sqrtf(sa) : 0.f;  // :179-187
}

// Post-processing straight from the integrator's buffer U[st->cur], optionally with the five loss sums of src/icnf.jl:489
// in the same launch.  Four lanes share a sample (their loads of one row are neighbours; one lane per sample would touch
// 64 rows, 140 bytes apart, with every load).  Sums: fixed tree inside the block, block partials through agent-scope
// atomics, the block that draws the last ticket adds them in block order -- the result does not depend on which one it is.
// need_done: the launch was enqueued on the strength of the host's step estimate, right behind the attempt expected to
// be the last; if that attempt did not finish the solve it does nothing and the host enqueues it again.
__global__ void __launch_bounds__(256)
k_post_state(NetDesc nd, int train, const StepState* st, const float* U0, const float* U1,
             float* __restrict__ logpx, float* __restrict__ regs, int B, int need_done, float* __restrict__ sums5,
             float* part, unsigned* ticket) {
    if (need_done && !st->done) return;
    const float* fsol = st->cur ? U1 : U0;
    const int tid = threadIdx.x, p = tid & 3;
    const int b = blockIdx.x * 64 + (tid >> 2);
    const int n_in = nd.n_in;
    const int D = n_in + 1 + (train ? 2 : 0);
    const float* c = fsol + (size_t)min(b, B - 1) * D;
    float ss = 0.f, sa = 0.f;
    for (int i = p; i < n_in; i += 4) {
        const float v = c[i];
        ss = fmaf(v, v, ss);
        if (i >= nd.nvars) sa = fmaf(v, v, sa);
    }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
    sa += __shfl_xor(sa, 1, 64); sa += __shfl_xor(sa, 2, 64);
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    if (b < B && p == 0) {
        const float log2pi = 1.8378770664093453f;
        const float logpz = -0.5f * fmaf((float)n_in, log2pi, ss);      // base_icnf.jl:177
        v4[0] = logpz - c[n_in];                                        // base_icnf.jl:178
        v4[1] = train ? c[n_in + 1] : 0.f;
        v4[2] = train ? c[n_in + 2] : 0.f;
        v4[3] = (nd.norm_z_aug && nd.naugs > 0) ? sqrtf(sa) : 0.f;      // :179-187
        logpx[b] = v4[0];
        regs[b] = v4[1];
        regs[(size_t)B + b] = v4[2];
        regs[2 * (size_t)B + b] = v4[3];
    }
    if (!sums5) return;
    __shared__ float sm[4][4];
    __shared__ int last;
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float r = wave_sum(v4[j]);
        if (l == 0) sm[j][w] = r;
    }
    __syncthreads();
    if (tid < 4)
        __hip_atomic_store(part + 4 * blockIdx.x + tid, (sm[tid][0] + sm[tid][1]) + (sm[tid][2] + sm[tid][3]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {                              // (the four stores above are this wave's: one wait covers them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == gridDim.x - 1;
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!last) return;
    // wave j adds sum j: lane l the blocks l, l + 64, ... in that order, then the fixed tree over the lanes
    float r = 0.f;
    for (unsigned i = l; i < gridDim.x; i += 64) r += __hip_atomic_load(part + 4 * i + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r = wave_sum(r);
    if (l == 0) sums5[w] = r;
    if (tid == 0) sums5[4] = (float)B;
}

// the integrator's initial state by value (a 76-byte host-to-device copy is a 5 us staging kernel of the runtime)
__global__ void k_set_state(StepState* dst, StepState v) { *dst = v; }

// conditional models: cond[b][o] = b1[o] + sum_c W1[o, n_in + c] * ys[c, b]   (padded to cbs with 0)
__global__ void k_cond_bias(NetDesc nd, const float* __restrict__ P, const float* __restrict__ ys,
                            float* __restrict__ cond, int cbs, int B) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * cbs) return;
    const int o = (int)(i % cbs);
    const size_t b = i / cbs;
    const int out = nd.dims[1];
    float v = 0.f;
    if (o < out) {
        v = P[nd.b_off[0] + o];
        const float* Wy = P + nd.wy_off;
        for (int c = 0; c < nd.n_cond; ++c) v = fmaf(Wy[o + (size_t)c * out], ys[b * nd.n_cond + c], v);
    }
    cond[i] = v;
}

// loss sums (src/icnf.jl:489): one block of 1024 lanes, deterministic (fixed lane->entry map,
// fixed tree); sums5 = (S logpx, S E, S n, S A, B)
__global__ void __launch_bounds__(1024)
k_loss_sums(const float* __restrict__ logpx, const float* __restrict__ regs, int B,
            float* __restrict__ sums5) {
    __shared__ float sm[4][16];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        s[0] += logpx[b];
        s[1] += regs[b];
        s[2] += regs[(size_t)B + b];
        s[3] += regs[2 * (size_t)B + b];
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int j = 0; j < 4; ++j) {
        float v = wave_sum(s[j]);
        if (l == 0) sm[j][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float r = 0.f;
        for (int i = 0; i < 16; ++i) r += sm[threadIdx.x][i];
        sums5[threadIdx.x] = r;
    }
    if (threadIdx.x == 0) sums5[4] = (float)B;
}

// ---------------------------------------------------------------------------------------
// host-callable launch wrappers
// ---------------------------------------------------------------------------------------
void launch_rhs_generic(const NetDesc& nd, const float* P, const RhsArgs& a, hipStream_t s) {
    const int tpb = 64;
    hipLaunchKernelGGL(k_rhs_generic, dim3((a.B + tpb - 1) / tpb), dim3(tpb), 0, s, nd, P, a);
}
void launch_norm_partials(const NormArgs& a, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_norm_partials, dim3(nblocks), dim3(256), 0, s, a);
}
void launch_controller(StepState* st, const float* partials, int phase, float n_total,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_controller, dim3(1), dim3(256), 0, s, st, partials, phase, n_total);
}
void launch_reduce_partials(const StepState* st, const float* partials, float* out3, float n_local,
                            hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, st, partials, out3, n_local);
}
void launch_controller_sums(StepState* st, const float* sums3, int phase, hipStream_t s) {
    hipLaunchKernelGGL(k_controller_sums, dim3(1), dim3(1), 0, s, st, sums3, phase);
}
void launch_build_u0(const float* xs, float* u0, int nvars, int D, int B, hipStream_t s, StepState* st_dst,
                     const StepState* st_val) {
    const size_t n = (size_t)D * B;
    StepState v{};
    if (st_dst) v = *st_val;
    hipLaunchKernelGGL(k_build_u0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xs, u0, nvars, D, B, st_dst, v);
}
void launch_copy_final(const StepState* st, const float* U0, const float* U1, float* out,
                       size_t n, hipStream_t s) {
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_copy_final, dim3(nb), dim3(256), 0, s, st, U0, U1, out, n);
}
void launch_post(const NetDesc& nd, int train, const float* fsol, float* logpx, float* regs,
                 int B, hipStream_t s) {
    hipLaunchKernelGGL(k_post, dim3((B + 255) / 256), dim3(256), 0, s, nd, train, fsol, logpx,
                       regs, B);
}
void launch_post_state(const NetDesc& nd, int train, const StepState* st, const float* U0, const float* U1,
                       float* logpx, float* regs, int B, hipStream_t s, bool need_done, float* sums5, float* part,
                       unsigned* ticket) {
    hipLaunchKernelGGL(k_post_state, dim3((B + 63) / 64), dim3(256), 0, s, nd, train, st, U0, U1, logpx, regs, B,
                       need_done ? 1 : 0, sums5, part, ticket);
}
void launch_set_state(StepState* dst, const StepState& v, hipStream_t s) {
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, s, dst, v);
}
void launch_cond_bias(const NetDesc& nd, const float* P, const float* ys, float* cond, int cbs, int B,
                      hipStream_t s) {
    const size_t n = (size_t)B * cbs;
    hipLaunchKernelGGL(k_cond_bias, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nd, P, ys, cond, cbs, B);
}
void launch_loss_sums(const float* logpx, const float* regs, int B, float* sums5,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_loss_sums, dim3(1), dim3(1024), 0, s, logpx, regs, B, sums5);
}
