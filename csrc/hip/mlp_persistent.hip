// Whole-network persistent training kernel for the MLP members of a batch-32 ensemble.
//
// The reference's distributed scripts train MLP-only ensembles (DDL_Ray*.py: `conv2D_size` = 0, Dense 3..19 units,
// /root/reference/DDL_Ray.py:21,76-85) with Keras' default batch of 32 (/root/reference/DDL.py:92): one optimizer
// step is ~60 tiny kernels (~190 us of launch-latency-bound work) on a 256-CU GPU.  Here ONE workgroup owns ONE
// ensemble member and runs a whole epoch of optimizer steps in ONE launch: gather the batch by index from the
// HBM-resident uint8 dataset, Flatten -> [Dense + activation -> BatchNorm]* -> Dense -> softmax cross-entropy,
// the full backward pass and the Adam / SGD / RMSprop update, with every activation of the step in LDS and the
// fp32 master parameters, moments and BN moving statistics updated in place in the network's flat buffers.
// The grid is the ensemble: all members of a generation train side by side, one CU each, nothing but the
// epoch losses ever returns to the host.  All arithmetic is fp32 (the multi-kernel path keeps activations in bf16).
//
// Layout contract (models/network.py): Dense kernels are [N][Kp] with K contiguous; the first Dense sits behind
// Flatten of an [pixels][cpad] image (cpad = 8 physical channels, `ctrue` <= 4 real ones, all copies of the one
// grey channel: DDL.py:119 stacks MNIST x3), so its K index is pix * cpad + c; later layers have Kp = pad8(units).
#include "common.h"

#define MLP_B 32            // rows per step (Keras default batch); a step may use fewer (ragged last batch)
#define MLP_MAXN 32         // units per layer
#define MLP_MAXL 11         // Dense layers including the softmax head (DDL.py:19-20,31-32: up to 9 hidden + head)
#define MLP_T 1024          // 16 waves: the first layer's weight streams are latency-bound, occupancy hides it
#define MLP_KQ 8            // split-K groups of the first layer's forward
#define MLP_NQ (MLP_T / 32 / MLP_KQ)     // output groups per (row, K range): 4 -> 8 outputs per thread
#define MLP_LD 33           // padded leading dimension of the [32][units] LDS tiles

struct MlpLayer {
    int w_off, b_off, N, Kp, act, has_bn, g_off, be_off, mm_off, mv_off;
    float alpha, momentum, bn_eps;
    int pad_;
};

struct MlpMember {
    float *P, *M, *V, *S;
    int* step;                 // device optimizer step counter (advanced by the number of steps run)
    float* loss_acc;           // += sum over steps of (mean batch loss * rows)
    const long* idx;           // [steps][MLP_B] sample indices into the resident dataset
    int n_layers, pixels, cpad0, ctrue0;
    int opt_kind;              // 0 adam, 1 sgd, 2 rmsprop
    float lr, b1, b2, eps, rho;
    MlpLayer L[MLP_MAXL];
};

struct OptStep { int kind; float lr, lr_t, b1, b2, eps, rho; };

__device__ __forceinline__ void opt1(const OptStep& o, float& p, float& m, float& v, float g) {
    if (o.kind == 0) {
        m = o.b1 * m + (1.f - o.b1) * g;
        v = o.b2 * v + (1.f - o.b2) * g * g;
        p -= o.lr_t * m / (sqrtf(v) + o.eps);
    } else if (o.kind == 1) {
        p -= o.lr * g;
    } else {
        v = o.rho * v + (1.f - o.rho) * g * g;
        p -= o.lr * g / (sqrtf(v) + o.eps);
    }
}

__device__ __forceinline__ void opt_at(const OptStep& o, float* P, float* M, float* V, int i, float g) {
    float p = P[i], m = o.kind == 0 ? M[i] : 0.f, v = o.kind != 1 ? V[i] : 0.f;
    opt1(o, p, m, v, g);
    P[i] = p;
    if (o.kind == 0) M[i] = m;
    if (o.kind != 1) V[i] = v;
}

__global__ __launch_bounds__(MLP_T) void mlp_persistent_kernel(const MlpMember* __restrict__ members,
                                                               const uint8_t* __restrict__ images,
                                                               const long* __restrict__ labels, int steps,
                                                               int last_rows) {
    extern __shared__ __align__(16) unsigned char smem[];
    const MlpMember& mb = members[blockIdx.x];
    const int t = threadIdx.x;
    const int NL = mb.n_layers, pixels = mb.pixels, cpad = mb.cpad0, ctrue = mb.ctrue0;
    const int pixS = (pixels + 15) / 16 * 16;
    // ---- LDS carve-up
    uint8_t* Xs = smem;                                                    // [32][pixS]
    float* A = (float*)(smem + MLP_B * pixS);                              // [MAXL][32][LD] post-activation
    float* H = A + MLP_MAXL * MLP_B * MLP_LD;                              // [MAXL][32][LD] layer outputs (post-BN)
    float* D0 = H + MLP_MAXL * MLP_B * MLP_LD;                             // [32][LD] gradient ping
    float* D1 = D0 + MLP_B * MLP_LD;                                       // [32][LD] gradient pong
    float* mean_s = D1 + MLP_B * MLP_LD;                                   // [MAXL][32]
    float* istd_s = mean_s + MLP_MAXL * MLP_MAXN;                          // [MAXL][32]
    float* coef = istd_s + MLP_MAXL * MLP_MAXN;                            // [2][32] dbeta, dgamma of the current BN
    float* rowloss = coef + 2 * MLP_MAXN;                                  // [32]
    // layer-0 split-K partials [8][32][32] live on H: they are produced and consumed inside layer 0's forward, before
    // its BatchNorm writes H[0] (a barrier in between), and H[1..] is not touched until layer 1
    static_assert(MLP_MAXL * MLP_LD >= MLP_KQ * MLP_MAXN, "split-K partials must fit inside H");
    float* red = H;
    int* lab_s = (int*)(rowloss + MLP_B);                                  // [32]
    __shared__ OptStep opt;
    __shared__ float s_loss;

    float *P = mb.P, *Mo = mb.M, *Vo = mb.V, *S = mb.S;
    const int step0 = mb.step[0];
    if (t == 0) s_loss = 0.f;

    for (int st = 0; st < steps; ++st) {
        const int rows = (st == steps - 1) ? last_rows : MLP_B;
        const float inv_rows = 1.f / (float)rows;
        const long* idx = mb.idx + (size_t)st * MLP_B;
        __syncthreads();
        // ---- optimizer constants of this step, labels, gather of the batch (uint8 pixels -> LDS)
        if (t == 0) {
            opt.kind = mb.opt_kind; opt.lr = mb.lr; opt.b1 = mb.b1; opt.b2 = mb.b2; opt.eps = mb.eps; opt.rho = mb.rho;
            const float tt = (float)(step0 + st + 1);
            opt.lr_t = mb.opt_kind == 0 ? mb.lr * sqrtf(1.f - powf(mb.b2, tt)) / (1.f - powf(mb.b1, tt)) : mb.lr;
        }
        if (t < MLP_B) lab_s[t] = t < rows ? (int)labels[idx[t]] : 0;
        {
            const int vec = pixS / 16;                     // 16-byte units per row (dataset rows are `pixels` bytes)
            for (int e = t; e < MLP_B * vec; e += MLP_T) {
                const int b = e / vec, u = e - b * vec;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (b < rows) {
                    const uint8_t* src = images + (size_t)idx[b] * pixels + u * 16;
                    if (u * 16 + 16 <= pixels && (((uintptr_t)src) & 15) == 0) v = *(const uint4*)src;
                    else {
                        uint8_t tmp[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) tmp[j] = (u * 16 + j < pixels) ? src[j] : 0;
                        v = *(const uint4*)tmp;
                    }
                }
                *(uint4*)(Xs + b * pixS + u * 16) = v;
            }
        }
        __syncthreads();

        // =================================================================== forward
        for (int l = 0; l < NL; ++l) {
            const MlpLayer& L = mb.L[l];
            const int N = L.N;
            float* Al = A + l * MLP_B * MLP_LD;
            float* Hl = H + l * MLP_B * MLP_LD;
            if (l == 0) {
                // split-K over 8 pixel ranges x 4 output groups: thread (nq, kq, b) accumulates outputs nq*8 .. nq*8+7 of
                // row b over its pixel range
                const int b = t & 31, kq = (t >> 5) & (MLP_KQ - 1), nq = t >> 8;
                constexpr int NPT = MLP_MAXN / MLP_NQ;
                const int n0 = nq * NPT;
                const int ppq = (pixels + MLP_KQ - 1) / MLP_KQ;
                const int p0 = kq * ppq, p1 = min(p0 + ppq, pixels);
                float acc[NPT];
#pragma unroll
                for (int j = 0; j < NPT; ++j) acc[j] = 0.f;
                const float* W = P + L.w_off;
                if (n0 < N) {
                    for (int pix = p0; pix < p1; ++pix) {
                        const float xv = (float)Xs[b * pixS + pix];
                        const float* wp = W + (size_t)pix * cpad;
#pragma unroll
                        for (int j = 0; j < NPT; ++j) {
                            if (n0 + j < N) {
                                const float4 w4 = *(const float4*)(wp + (size_t)(n0 + j) * L.Kp);
                                float ws = w4.x;
                                if (ctrue > 1) ws += w4.y;
                                if (ctrue > 2) ws += w4.z;
                                if (ctrue > 3) ws += w4.w;
                                acc[j] = fmaf(xv, ws, acc[j]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NPT; ++j)
                    if (n0 + j < N) red[(kq * MLP_B + b) * MLP_MAXN + n0 + j] = acc[j];
                __syncthreads();
                act_switch(L.act, [&](auto KA) {        // loops compiled per activation: no branch chain per element
                    for (int e = t; e < MLP_B * N; e += MLP_T) {
                        const int b2 = e / N, n = e - b2 * N;
                        float z = P[L.b_off + n];
#pragma unroll
                        for (int q = 0; q < MLP_KQ; ++q) z += red[(q * MLP_B + b2) * MLP_MAXN + n];
                        Al[b2 * MLP_LD + n] = b2 < rows ? act_apply_k<decltype(KA)::value>(z, L.act, L.alpha) : 0.f;
                    }
                });
            } else {
                const MlpLayer& Lp = mb.L[l - 1];
                const float* Hp = H + (l - 1) * MLP_B * MLP_LD;
                const int K = Lp.N;
                const int b = t & 31;
                act_switch(L.act, [&](auto KA) {
                    for (int n = t >> 5; n < N; n += MLP_T / 32) {
                        const float* w = P + L.w_off + (size_t)n * L.Kp;
                        float z = P[L.b_off + n];
                        for (int k = 0; k < K; ++k) z = fmaf(Hp[b * MLP_LD + k], w[k], z);
                        Al[b * MLP_LD + n] = b < rows ? act_apply_k<decltype(KA)::value>(z, L.act, L.alpha) : 0.f;
                    }
                });
            }
            __syncthreads();
            if (L.has_bn) {
                if (t < N) {
                    float s0 = 0.f, s1 = 0.f;
                    for (int b = 0; b < rows; ++b) { const float a = Al[b * MLP_LD + t]; s0 += a; s1 += a * a; }
                    const float mu = s0 * inv_rows;
                    float var = s1 * inv_rows - mu * mu;
                    var = var < 0.f ? 0.f : var;
                    mean_s[l * MLP_MAXN + t] = mu;
                    istd_s[l * MLP_MAXN + t] = rsqrtf(var + L.bn_eps);
                    S[L.mm_off + t] = S[L.mm_off + t] * L.momentum + mu * (1.f - L.momentum);
                    S[L.mv_off + t] = S[L.mv_off + t] * L.momentum + var * (1.f - L.momentum);
                }
                __syncthreads();
                for (int e = t; e < MLP_B * N; e += MLP_T) {
                    const int b = e / N, n = e - b * N;
                    const float xh = (Al[b * MLP_LD + n] - mean_s[l * MLP_MAXN + n]) * istd_s[l * MLP_MAXN + n];
                    Hl[b * MLP_LD + n] = b < rows ? fmaf(P[L.g_off + n], xh, P[L.be_off + n]) : 0.f;
                }
            } else {
                for (int e = t; e < MLP_B * N; e += MLP_T) {
                    const int b = e / N, n = e - b * N;
                    Hl[b * MLP_LD + n] = Al[b * MLP_LD + n];
                }
            }
            __syncthreads();
        }

        // =================================================================== softmax cross-entropy (Keras: clipped probs)
        {
            const MlpLayer& L = mb.L[NL - 1];
            const float* Z = H + (NL - 1) * MLP_B * MLP_LD;
            if (t < MLP_B) {
                float loss = 0.f;
                if (t < rows) {
                    float mx = -3.0e38f;
                    for (int n = 0; n < L.N; ++n) mx = fmaxf(mx, Z[t * MLP_LD + n]);
                    float sum = 0.f;
                    for (int n = 0; n < L.N; ++n) sum += __expf(Z[t * MLP_LD + n] - mx);
                    const float inv = 1.f / sum;
                    const int y = lab_s[t];
                    for (int n = 0; n < L.N; ++n) {
                        const float p = __expf(Z[t * MLP_LD + n] - mx) * inv;
                        if (n == y) loss = -__logf(fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f));
                        D0[t * MLP_LD + n] = (p - (n == y ? 1.f : 0.f)) * inv_rows;
                    }
                } else {
                    for (int n = 0; n < L.N; ++n) D0[t * MLP_LD + n] = 0.f;
                }
                rowloss[t] = loss;
            }
            __syncthreads();
            if (t == 0) {
                float s = 0.f;
                for (int b = 0; b < rows; ++b) s += rowloss[b];
                s_loss += s;                         // = mean batch loss * rows
            }
        }

        // =================================================================== backward + update
        float* Dz = D0;          // gradient w.r.t. the current layer's pre-activation (head) / output (others)
        float* Dn = D1;
        for (int l = NL - 1; l >= 0; --l) {
            const MlpLayer& L = mb.L[l];
            const int N = L.N;
            const float* Al = A + l * MLP_B * MLP_LD;
            if (l != NL - 1 || L.has_bn || L.act != ACT_LINEAR) {
                // Dz holds dL/dH[l]: through BatchNorm (batch statistics) and the activation -> dL/dz
                if (L.has_bn) {
                    if (t < N) {
                        float db = 0.f, dg = 0.f;
                        const float mu = mean_s[l * MLP_MAXN + t], is = istd_s[l * MLP_MAXN + t];
                        for (int b = 0; b < rows; ++b) {
                            const float g = Dz[b * MLP_LD + t];
                            db += g;
                            dg += g * (Al[b * MLP_LD + t] - mu) * is;
                        }
                        coef[t] = db;
                        coef[MLP_MAXN + t] = dg;
                    }
                    __syncthreads();
                    act_switch(L.act, [&](auto KA) {
                        for (int e = t; e < MLP_B * N; e += MLP_T) {
                            const int b = e / N, n = e - b * N;
                            const float is = istd_s[l * MLP_MAXN + n];
                            const float a = Al[b * MLP_LD + n];
                            const float xh = (a - mean_s[l * MLP_MAXN + n]) * is;
                            float d = P[L.g_off + n] * is *
                                      (Dz[b * MLP_LD + n] - coef[n] * inv_rows - xh * coef[MLP_MAXN + n] * inv_rows);
                            d *= act_grad_y_k<decltype(KA)::value>(a, L.act, L.alpha);
                            Dz[b * MLP_LD + n] = b < rows ? d : 0.f;
                        }
                    });
                    __syncthreads();
                    if (t < N) {                     // gamma / beta: every reader of the old values is past the barrier
                        opt_at(opt, P, Mo, Vo, L.g_off + t, coef[MLP_MAXN + t]);
                        opt_at(opt, P, Mo, Vo, L.be_off + t, coef[t]);
                    }
                } else {
                    act_switch(L.act, [&](auto KA) {
                        for (int e = t; e < MLP_B * N; e += MLP_T) {
                            const int b = e / N, n = e - b * N;
                            Dz[b * MLP_LD + n] *= act_grad_y_k<decltype(KA)::value>(Al[b * MLP_LD + n], L.act, L.alpha);
                        }
                    });
                    __syncthreads();
                }
            }
            // input gradient first (needs the OLD weights), then the weight / bias update
            if (l > 0) {
                const int K = mb.L[l - 1].N;
                for (int e = t; e < MLP_B * K; e += MLP_T) {
                    const int b = e / K, k = e - b * K;
                    float s = 0.f;
                    for (int n = 0; n < N; ++n) s = fmaf(Dz[b * MLP_LD + n], P[L.w_off + (size_t)n * L.Kp + k], s);
                    Dn[b * MLP_LD + k] = s;
                }
                __syncthreads();
                const float* Hp = H + (l - 1) * MLP_B * MLP_LD;
                for (int e = t; e < N * K; e += MLP_T) {
                    const int n = e / K, k = e - n * K;
                    float g = 0.f;
                    for (int b = 0; b < rows; ++b) g = fmaf(Dz[b * MLP_LD + n], Hp[b * MLP_LD + k], g);
                    opt_at(opt, P, Mo, Vo, L.w_off + n * L.Kp + k, g);
                }
            } else {
                // first layer: dW[n][pix][c] = sum_b dz[b][n] * x[b][pix], identical for the c copies of the grey channel
                // thread (nq, pixel lane): pixels lane, lane + 256, ... and outputs nq, nq + 4, ...
                const int lane = t & 255, nq = t >> 8;
                for (int pix = lane; pix < pixels; pix += 256) {
                    float xc[MLP_B];
#pragma unroll
                    for (int b = 0; b < MLP_B; ++b) xc[b] = (float)Xs[b * pixS + pix];
                    for (int n = nq; n < N; n += MLP_T / 256) {
                        const size_t wi = (size_t)L.w_off + (size_t)n * L.Kp + (size_t)pix * cpad;
                        float4 p4 = *(float4*)(P + wi);                    // issued before the reduction below
                        float4 m4 = opt.kind == 0 ? *(float4*)(Mo + wi) : make_float4(0, 0, 0, 0);
                        float4 v4 = opt.kind != 1 ? *(float4*)(Vo + wi) : make_float4(0, 0, 0, 0);
                        float g = 0.f;
#pragma unroll
                        for (int b = 0; b < MLP_B; ++b) g = fmaf(Dz[b * MLP_LD + n], xc[b], g);
                        opt1(opt, p4.x, m4.x, v4.x, g);
                        if (ctrue > 1) opt1(opt, p4.y, m4.y, v4.y, g);
                        if (ctrue > 2) opt1(opt, p4.z, m4.z, v4.z, g);
                        if (ctrue > 3) opt1(opt, p4.w, m4.w, v4.w, g);
                        *(float4*)(P + wi) = p4;
                        if (opt.kind == 0) *(float4*)(Mo + wi) = m4;
                        if (opt.kind != 1) *(float4*)(Vo + wi) = v4;
                    }
                }
            }
            if (t < N) {
                float g = 0.f;
                for (int b = 0; b < rows; ++b) g += Dz[b * MLP_LD + t];
                opt_at(opt, P, Mo, Vo, L.b_off + t, g);
            }
            __syncthreads();
            float* tmp = Dz; Dz = Dn; Dn = tmp;
        }
    }
    __syncthreads();
    if (t == 0) {
        mb.loss_acc[0] += s_loss;
        mb.step[0] = step0 + steps;
    }
}

DDL_API long ddl_mlp_member_bytes() { return (long)sizeof(MlpMember); }
DDL_API int ddl_mlp_limits(int* max_layers, int* max_units, int* batch) {
    *max_layers = MLP_MAXL; *max_units = MLP_MAXN; *batch = MLP_B;
    return 0;
}

static size_t mlp_smem(int pixels) {
    const int pixS = (pixels + 15) / 16 * 16;
    size_t f = (size_t)2 * MLP_MAXL * MLP_B * MLP_LD + 2 * MLP_B * MLP_LD + 2 * MLP_MAXL * MLP_MAXN + 2 * MLP_MAXN +
               MLP_B + MLP_B;
    return (size_t)MLP_B * pixS + f * sizeof(float);
}

// `members`: device array of n_members MlpMember records (all with the same `pixels`); runs `steps` optimizer steps
// of every member, the last one on `last_rows` (<= 32) samples.
DDL_API int ddl_mlp_persistent(const void* members, int n_members, const void* images, const long* labels,
                               int pixels, int steps, int last_rows, hipStream_t s) {
    if (n_members < 1 || steps < 1 || last_rows < 1 || last_rows > MLP_B || pixels < 1) return -1;
    const size_t smem = mlp_smem(pixels);
    if (smem > 158 * 1024) return -2;       // the kernel also has a few static LDS words
    static size_t attr_set = 0;
    if (smem > attr_set) {
        if (hipFuncSetAttribute((const void*)mlp_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess) return -3;
        attr_set = smem;
    }
    mlp_persistent_kernel<<<n_members, MLP_T, smem, s>>>((const MlpMember*)members, (const uint8_t*)images, labels,
                                                        steps, last_rows);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
