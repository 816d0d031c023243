// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef uint16_t bf16_t;   // raw bfloat16 bits; activations/weights are bf16, accumulation fp32

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DDL_API extern "C" __attribute__((visibility("default")))

enum Act { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SOFTPLUS = 3, ACT_TANH = 4,
           ACT_SOFTSIGN = 5, ACT_LEAKY = 6, ACT_TANH01 = 7 };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// Activation transcendentals straight on the hardware units: v_exp_f32 (2^x), v_log_f32 (log2 x), v_rcp_f32, one
// instruction each.  `__expf` / `__logf` wrap them in denormal-range scaling (compare, select, ldexp: ~7 more
// instructions per call) and `__frcp_rn` is a correctly rounded division (div_scale / div_fmas / div_fixup: ~10);
// an 8-channel softplus epilogue was 14 VALU instructions per element beside its two transcendentals and the
// streaming kernels that carry one were instruction-bound.  Every result here is rounded to bf16 (8 mantissa bits),
// the units are good to 1 ulp of fp32, and the arguments never need the denormal path: exp underflows to the 0 it
// should round to, and the logarithm only ever sees 1 + e^x >= 1.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh(x) = 1 - 2 / (1 + e^{2x}); saturates cleanly for |x| large (e^{2x} -> 0 or inf)
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * fast_rcp(1.f + fast_exp(2.f * x)); }

__device__ __forceinline__ float act_apply(float x, int kind, float alpha) {
    switch (kind) {
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_SIGMOID: return fast_rcp(1.f + fast_exp(-x));
        case ACT_SOFTPLUS: return x > 15.f ? x : fast_log(1.f + fast_exp(x));
        case ACT_TANH: return fast_tanh(x);
        case ACT_SOFTSIGN: return x * fast_rcp(1.f + fabsf(x));
        case ACT_LEAKY: return x > 0.f ? x : alpha * x;
        case ACT_TANH01: return fast_rcp(1.f + fast_exp(-2.f * x));       // (tanh(x) + 1) / 2 == sigmoid(2x)
        default: return x;
    }
}

// d act(x)/dx written through the OUTPUT y = act(x) (no pre-activation is ever stored).
__device__ __forceinline__ float act_grad_y(float y, int kind, float alpha) {
    switch (kind) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_SIGMOID: return y * (1.f - y);
        case ACT_SOFTPLUS: return 1.f - fast_exp(-y);
        case ACT_TANH: return 1.f - y * y;
        case ACT_SOFTSIGN: { float t = 1.f - fabsf(y); return t * t; }
        case ACT_LEAKY: return y > 0.f ? 1.f : alpha;
        case ACT_TANH01: { float t = 2.f * y - 1.f; return 0.5f * (1.f - t * t); }
        default: return 1.f;
    }
}

// linear / relu / leaky as ONE branch-free form: x > 0 ? x : slope * x (+ 0 so that relu gives +0, as fmaxf does)
__device__ __forceinline__ float act_slope(int kind, float alpha) {
    return kind == ACT_RELU ? 0.f : (kind == ACT_LEAKY ? alpha : 1.f);
}
__device__ __forceinline__ float act_apply_slope(float x, float slope) { return x > 0.f ? x : slope * x + 0.f; }

// The switches above cost a chain of scalar compares and branches PER ELEMENT when `kind` is a run-time value (131
// s_cbranch in bn_apply_cs for 8 elements per lane: the streaming kernels were branch-bound, not bandwidth-bound).
// Kernels therefore take their activation codes as template parameters -- act_apply_k<K> folds the switch at compile
// time; K = -1 keeps the run-time code -- and the host picks the instantiation (ACT1_DISPATCH / ACT2_DISPATCH).
template <int K>
__device__ __forceinline__ float act_apply_k(float x, int kind, float alpha) {
    return act_apply(x, K >= 0 ? K : kind, alpha);
}
template <int K>
__device__ __forceinline__ float act_grad_y_k(float y, int kind, float alpha) {
    return act_grad_y(y, K >= 0 ? K : kind, alpha);
}
// In-kernel form for epilogues: f(std::integral_constant<int, kind>) -- one branch chain per call instead of one per
// element inside it (the body is compiled once per activation).
template <typename F>
__device__ __forceinline__ void act_switch(int kind, F&& f) {
    switch (kind) {
        case ACT_RELU: f(std::integral_constant<int, ACT_RELU>{}); break;
        case ACT_SIGMOID: f(std::integral_constant<int, ACT_SIGMOID>{}); break;
        case ACT_SOFTPLUS: f(std::integral_constant<int, ACT_SOFTPLUS>{}); break;
        case ACT_TANH: f(std::integral_constant<int, ACT_TANH>{}); break;
        case ACT_SOFTSIGN: f(std::integral_constant<int, ACT_SOFTSIGN>{}); break;
        case ACT_LEAKY: f(std::integral_constant<int, ACT_LEAKY>{}); break;
        case ACT_TANH01: f(std::integral_constant<int, ACT_TANH01>{}); break;
        default: f(std::integral_constant<int, ACT_LINEAR>{}); break;
    }
}
// CALL(A): one activation code.
#define ACT1_DISPATCH(act, CALL)                                                                                   \
    do {                                                                                                           \
        switch (act) {                                                                                             \
            case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;        \
            case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break;        \
            default: CALL(-1); break;                                                                              \
        }                                                                                                          \
    } while (0)
// CALL(A, P): two codes; every network here has at least one of them linear (conv(act) -> BN, or conv -> BN ->
// act), so those 15 combinations are compiled and anything else runs the run-time instantiation.
#define ACT2_DISPATCH(act, prev, CALL)                                                                             \
    do {                                                                                                           \
        if ((prev) == 0) {                                                                                         \
            switch (act) {                                                                                         \
                case 0: CALL(0, 0); break; case 1: CALL(1, 0); break; case 2: CALL(2, 0); break;                   \
                case 3: CALL(3, 0); break; case 4: CALL(4, 0); break; case 5: CALL(5, 0); break;                   \
                case 6: CALL(6, 0); break; case 7: CALL(7, 0); break; default: CALL(-1, -1); break;                \
            }                                                                                                      \
        } else if ((act) == 0) {                                                                                   \
            switch (prev) {                                                                                        \
                case 1: CALL(0, 1); break; case 2: CALL(0, 2); break; case 3: CALL(0, 3); break;                   \
                case 4: CALL(0, 4); break; case 5: CALL(0, 5); break; case 6: CALL(0, 6); break;                   \
                case 7: CALL(0, 7); break; default: CALL(-1, -1); break;                                           \
            }                                                                                                      \
        } else {                                                                                                   \
            CALL(-1, -1);                                                                                          \
        }                                                                                                          \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
