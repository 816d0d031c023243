// One float4 of the fused multi-tensor optimizer update, shared by optim_kernel (loss_optim.hip) and the
// fused epilogue of the xGMI direct all-reduce (csrc/comm/p2p_allreduce.hip): both must produce the same bits.
// kind: 0 adam, 1 sgd, 2 rmsprop, 3 adam_fresh_double (the reference GAN's "new Adam applied twice").
#pragma once
#include "common.h"

struct OptimHyper {
    int kind;
    float lr, b1, b2, eps, rho;
    float lr_t, lr_2;       // bias-corrected step sizes, derived from the device step counter by optim_prepare
};

__device__ __forceinline__ OptimHyper optim_prepare(int kind, float lr, float b1, float b2, float eps, float rho,
                                                    const int* __restrict__ step, int t_add = 0) {
    // t_add = 1: the counter still holds the number of COMPLETED updates (the calling kernel advances it itself
    // when its last block finishes); 0: somebody advanced it before the launch
    OptimHyper h{kind, lr, b1, b2, eps, rho, lr, lr};
    if (kind == 0) {
        float t = (float)(step[0] + t_add);
        h.lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    } else if (kind == 3) {
        h.lr_t = lr * sqrtf(1.f - b2) / (1.f - b1);
        h.lr_2 = lr * sqrtf(1.f - b2 * b2) / (1.f - b1 * b1);
    }
    return h;
}

// p, g: one float4 of parameters / (already scaled) gradients; Mo / Vo: moment arrays (may be null for kinds that
// do not use them); i: float4 index.  Writes P[i] (fp32 master) and, if Pc, the bf16 working copy.
__device__ __forceinline__ void optim_update4(const OptimHyper& h, float4* __restrict__ P, float4* __restrict__ Mo,
                                              float4* __restrict__ Vo, uint2* __restrict__ Pc, long i, float4 g4,
                                              const float4* __restrict__ avg, float avg_w) {
    float4 p4 = P[i];
    float p[4] = {p4.x, p4.y, p4.z, p4.w};
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float b1 = h.b1, b2 = h.b2, eps = h.eps;
    if (h.kind == 0) {
        float4 m4 = Mo[i], v4 = Vo[i];
        float m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = b1 * m[j] + (1.f - b1) * g[j];
            v[j] = b2 * v[j] + (1.f - b2) * g[j] * g[j];
            p[j] -= h.lr_t * m[j] / (sqrtf(v[j]) + eps);
        }
        Mo[i] = make_float4(m[0], m[1], m[2], m[3]);
        Vo[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else if (h.kind == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] -= h.lr * g[j];
    } else if (h.kind == 2) {
        float4 v4 = Vo[i];
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = h.rho * v[j] + (1.f - h.rho) * g[j] * g[j];
            p[j] -= h.lr * g[j] / (sqrtf(v[j]) + eps);
        }
        Vo[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m1 = (1.f - b1) * g[j], v1 = (1.f - b2) * g[j] * g[j];
            p[j] -= h.lr_t * m1 / (sqrtf(v1) + eps);
            float m2 = b1 * m1 + (1.f - b1) * g[j], v2 = b2 * v1 + (1.f - b2) * g[j] * g[j];
            p[j] -= h.lr_2 * m2 / (sqrtf(v2) + eps);
        }
    }
    if (avg) {
        float4 a = avg[i];
        p[0] = (1.f - avg_w) * p[0] + avg_w * a.x; p[1] = (1.f - avg_w) * p[1] + avg_w * a.y;
        p[2] = (1.f - avg_w) * p[2] + avg_w * a.z; p[3] = (1.f - avg_w) * p[3] + avg_w * a.w;
    }
    P[i] = make_float4(p[0], p[1], p[2], p[3]);
    if (Pc) Pc[i] = make_uint2(pack2bf(p[0], p[1]), pack2bf(p[2], p[3]));
}
