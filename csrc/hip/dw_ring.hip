// Depthwise convolution forward over an LDS ring of input rows (unpadded 'valid' layers, K <= 3, stride <= 3: every
// DepthwiseConv2D of the DDL_Normal4 search space, /root/reference/DDL_Normal4.py:36-37).
//
// The per-pixel kernel (dw_fprop_fast, wgrad.hip) gives every (output pixel, 8 channels) item to a thread that issues
// its K*K input loads and K*K weight loads itself: 18 vector-memory instructions per 16-byte store, each input byte
// requested 2.25 (stride 2) to 9 (stride 1) times from L1 / L2, and model0's 57x57x128 layer at batch 512 ran at
// 2.1 TB/s of its compulsory traffic.  Here a block owns (image, channel chunk) and walks the image top to bottom:
// the input rows live in an LDS ring of K + stride rows, every input byte is loaded from global memory exactly once
// (coalesced 16-byte units, requested one output row ahead into registers, written to the ring after the row's
// arithmetic), the stencil reads come from LDS, and the weights sit in LDS as fp32.  One barrier per step of R output
// rows (R > 1 where the ring has room: a 28-pixel-wide stride-1 layer has 208 items and one new input row per output
// row, and one row per barrier left it bound by the latency of that row's load: 97 us for 191 MB).
//
// Optional BatchNorm on load (in_scale / in_shift: a BatchNorm in front of the layer that is never materialised):
// with every tap inside the image the affine goes on the tap SUM, s * sum_t w_t x_t + h * sum_t w_t, in fp32 -- the
// same arithmetic as dw_fprop_fast's AFF = 1 path.
#include "common.h"
#include <cstdlib>

struct DwRingArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    const float* in_scale; const float* in_shift; int in_ctrue;
    int B, H, W, C, OH, OW, sh, sw, c_true, act;
    float alpha;
    int Cc, RR, R;            // channels per block (multiple of 8), ring rows, output rows per step (barrier)
    float* stats;             // BatchNorm partials of the stored output, part[2][C][B] (column = image), or null
};

__device__ __forceinline__ int dwr_fdiv(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

// staging: unit u of a group of `nr` rows starting at input row r0 = (row r0 + u / upr, pixel-major 16-byte unit)
typedef unsigned int dwr_u32x4 __attribute__((ext_vector_type(4)));

struct DwStage { const bf16_t* xb; char* ring; int upr, cgc, W, C, Cc, RR, rowb; float inv_upr, inv_cgc; };

template <int U>
__device__ __forceinline__ void dwr_stage_load(const DwStage& g, int r0, int nr, dwr_u32x4 (&v)[U]) {
    const int n = nr * g.upr, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int u = min(tid + k * 256, n - 1);                // clamped: always a valid address
        const int r = dwr_fdiv(u, g.inv_upr), q = u - r * g.upr;
        const int px = dwr_fdiv(q, g.inv_cgc), g8 = q - px * g.cgc;
        v[k] = *(const dwr_u32x4*)(g.xb + ((size_t)(r0 + r) * g.W + px) * g.C + g8 * 8);
    }
}

template <int U>
__device__ __forceinline__ void dwr_stage_store(const DwStage& g, int r0, int nr, const dwr_u32x4 (&v)[U]) {
    const int n = nr * g.upr, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int u = tid + k * 256;
        if (u < n) {
            const int r = dwr_fdiv(u, g.inv_upr), q = u - r * g.upr;
            const int px = dwr_fdiv(q, g.inv_cgc), g8 = q - px * g.cgc;
            const int slot = (r0 + r) % g.RR;
            *(dwr_u32x4*)(g.ring + (size_t)slot * g.rowb + (px * g.Cc + g8 * 8) * 2) = v[k];
        }
    }
}

template <int K, int U>      // U: 16-byte units a thread stages per output row (sh * W * Cc / 8 <= 256 * U)
__global__ __launch_bounds__(256) void dw_fprop_ring_kernel(const DwRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, c0 = blockIdx.y * a.Cc;
    const int Cc = min(a.Cc, a.C - c0), cgc = Cc >> 3;          // this block's channels / channel groups
    const int rowb = a.W * a.Cc * 2;                            // ring row pitch in bytes
    char* ring = smem;                                          // [RR][W][a.Cc] bf16
    float* wsm = (float*)(smem + (size_t)a.RR * rowb);          // [K*K][a.Cc] fp32
    float* aff = wsm + K * K * a.Cc;                            // [2][a.Cc]: scale, shift * sum_t w_t + bias
    const bool has_aff = a.in_scale != nullptr;

    for (int i = tid; i < K * K * Cc; i += 256) {
        const int t = i / Cc, c = i - t * Cc;
        wsm[t * a.Cc + c] = bf2f(a.w[(size_t)t * a.C + c0 + c]);
    }
    __syncthreads();
    for (int c = tid; c < Cc; c += 256) {
        float ws = 0.f;
#pragma unroll
        for (int t = 0; t < K * K; ++t) ws += wsm[t * a.Cc + c];
        const bool live = has_aff && c0 + c < a.in_ctrue;
        const float bv = (a.bias && c0 + c < a.c_true) ? a.bias[c0 + c] : 0.f;
        aff[c] = has_aff ? (live ? a.in_scale[c0 + c] : 0.f) : 1.f;
        aff[a.Cc + c] = (live ? a.in_shift[c0 + c] * ws : 0.f) + bv;
    }

    const int upr = a.W * cgc;                                  // units per input row
    const float inv_cgc = 1.0f / (float)cgc;
    const DwStage sg{a.x + (size_t)b * a.H * a.W * a.C + c0, ring, upr, cgc, a.W, a.C, a.Cc, a.RR, rowb,
                     1.0f / (float)upr, inv_cgc};
    // prologue: the window of the first step, in passes of at most R * sh rows (what the staging registers hold)
    const int R = a.R, step_rows = R * a.sh;
    const int win0 = min(a.H, (min(R, a.OH) - 1) * a.sh + K);
    for (int r = 0; r < win0; r += step_rows) {
        dwr_u32x4 v[U];
        const int nr = min(step_rows, win0 - r);
        dwr_stage_load<U>(sg, r, nr, v);
        dwr_stage_store<U>(sg, r, nr, v);
    }
    __syncthreads();

    const int rowitems = a.OW * cgc;
    const float inv_rowitems = 1.0f / (float)rowitems;
    float st_s[8], st_q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st_s[j] = st_q[j] = 0.f;
    const int act = a.act;
    const float alpha = a.alpha;
    bf16_t* yb = a.y + (size_t)b * a.OH * a.OW * a.C + c0;
    for (int oh0 = 0; oh0 < a.OH; oh0 += R) {
        // the rows the NEXT step adds to the window, requested now, written to the ring after this step's math
        // (after the last step: an unconditional load of row 0 that is never stored -- a conditional one, or HIP's
        // uint4 struct instead of a vector type, put the staging registers into scratch memory)
        const int ohn = oh0 + R;                                // first output row of the next step
        const bool more = ohn < a.OH;
        const int have = (ohn - 1) * a.sh + K;                  // input rows [0, have) are in the ring
        const int need = more ? (min(ohn + R, a.OH) - 1) * a.sh + K : 0;
        const int nr0 = more ? have : 0, nrn = more ? need - have : 1;
        dwr_u32x4 nv[U];
        dwr_stage_load<U>(sg, nr0, nrn, nv);
        const int items = min(R, a.OH - oh0) * rowitems;
        for (int it = tid; it < items; it += 256) {
            const int rl = dwr_fdiv(it, inv_rowitems), q = it - rl * rowitems;
            const int ow = dwr_fdiv(q, inv_cgc), g8 = q - ow * cgc;
            const int oh = oh0 + rl;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const char* row = ring + (size_t)((oh * a.sh + kh) % a.RR) * rowb + g8 * 16;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    float xf[8];
                    unpack8(*(const uint4*)(row + (ow * a.sw + kw) * a.Cc * 2), xf);
                    const float4 w0 = *(const float4*)(wsm + (kh * K + kw) * a.Cc + g8 * 8);
                    const float4 w1 = *(const float4*)(wsm + (kh * K + kw) * a.Cc + g8 * 8 + 4);
                    acc[0] = fmaf(xf[0], w0.x, acc[0]); acc[1] = fmaf(xf[1], w0.y, acc[1]);
                    acc[2] = fmaf(xf[2], w0.z, acc[2]); acc[3] = fmaf(xf[3], w0.w, acc[3]);
                    acc[4] = fmaf(xf[4], w1.x, acc[4]); acc[5] = fmaf(xf[5], w1.y, acc[5]);
                    acc[6] = fmaf(xf[6], w1.z, acc[6]); acc[7] = fmaf(xf[7], w1.w, acc[7]);
                }
            }
            const float4 s0 = *(const float4*)(aff + g8 * 8), s1 = *(const float4*)(aff + g8 * 8 + 4);
            const float4 h0 = *(const float4*)(aff + a.Cc + g8 * 8), h1 = *(const float4*)(aff + a.Cc + g8 * 8 + 4);
            acc[0] = fmaf(s0.x, acc[0], h0.x); acc[1] = fmaf(s0.y, acc[1], h0.y);
            acc[2] = fmaf(s0.z, acc[2], h0.z); acc[3] = fmaf(s0.w, acc[3], h0.w);
            acc[4] = fmaf(s1.x, acc[4], h1.x); acc[5] = fmaf(s1.y, acc[5], h1.y);
            acc[6] = fmaf(s1.z, acc[6], h1.z); acc[7] = fmaf(s1.w, acc[7], h1.w);
            act_switch(act, [&](auto KA) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[j] = c0 + g8 * 8 + j < a.c_true ? act_apply_k<decltype(KA)::value>(acc[j], act, alpha) : 0.f;
            });
            const uint4 o = pack8(acc);
            *(uint4*)(yb + ((size_t)oh * a.OW + ow) * a.C + g8 * 8) = o;
            if (a.stats) {                  // (host-checked: 256 % cgc == 0, a thread's channel group never changes)
                float r[8];
                unpack8(o, r);              // what the stored tensor holds
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    st_s[j] += r[j];
                    st_q[j] = fmaf(r[j], r[j], st_q[j]);
                }
            }
        }
        if (more) dwr_stage_store<U>(sg, nr0, nrn, nv);
        __syncthreads();
    }
    if (a.stats) {
        // lanes l, l + cgc, ... of a wave hold different pixels of one channel group: fold them, then the four waves
        const int lane = tid & 63, wave = tid >> 6;
        for (int off = cgc; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                st_s[j] += __shfl_xor(st_s[j], off, 64);
                st_q[j] += __shfl_xor(st_q[j], off, 64);
            }
        float* red = (float*)smem;          // [4 waves][2][a.Cc]  (the ring is no longer read: barrier above)
        if (lane < cgc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(wave * 2 + 0) * a.Cc + lane * 8 + j] = st_s[j];
                red[(wave * 2 + 1) * a.Cc + lane * 8 + j] = st_q[j];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * Cc; i += 256) {
            const int k = i / Cc, c = i - k * Cc;
            const float v = ((red[k * a.Cc + c] + red[(2 + k) * a.Cc + c]) + red[(4 + k) * a.Cc + c]) + red[(6 + k) * a.Cc + c];
            a.stats[((size_t)k * a.C + c0 + c) * a.B + b] = v;
        }
    }
}

// Returns 0 on success, 1 if the layer is outside this kernel's range (nothing launched).
extern "C" int ddl_dw_fprop_ring(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C,
                                 int OH, int OW, int K, int sh, int sw, int c_true, int act, float alpha,
                                 const float* in_scale, const float* in_shift, int in_ctrue, float* stats, hipStream_t s) {
    static const int on = getenv("DDL_DW_RING") ? atoi(getenv("DDL_DW_RING")) : 1;
    if (!on || K < 1 || K > 3 || sh < 1 || sh > 3 || sw < 1 || sw > 3 || C % 8) return 1;
    if ((OH - 1) * sh + K > H || (OW - 1) * sw + K > W || OH < 1 || OW < 1) return 1;      // every tap inside the image
    if ((long)H * W * C >= (1L << 24)) return 1;                    // per-image index math in 32 bits / dwr_fdiv's range
    static const long budget = getenv("DDL_DW_RING_LDS_KB") ? atol(getenv("DDL_DW_RING_LDS_KB")) * 1024 : 36 * 1024;
    static const int rmax = getenv("DDL_DW_RING_ROWS") ? atoi(getenv("DDL_DW_RING_ROWS")) : 4;
    // channel chunk: the whole layer if a one-row-per-step ring fits the budget, else the largest multiple of 8 that does
    int Cc = C;
    while (Cc > 8 && (long)(K + sh) * W * Cc * 2 > budget) Cc = ((Cc / 2) + 7) / 8 * 8;
    // ... and enough blocks for four per CU while a pixel's chunk stays at least 64 bytes
    while (Cc > 32 && (long)B * ((C + Cc - 1) / Cc) < 1024) Cc = ((Cc / 2) + 7) / 8 * 8;
    // output rows per step: as many as the ring has room for (window (R-1) sh + K rows + R sh incoming rows) and the
    // staging registers hold (R sh W Cc / 8 units over 256 threads, at most 8 each)
    int R = 1;
    while (R < rmax && R < OH && (long)((2 * (R + 1) - 1) * sh + K) * W * Cc * 2 <= budget &&
           (long)(R + 1) * sh * W * (Cc / 8) <= 256 * 8)
        ++R;
    const int RR = (2 * R - 1) * sh + K;
    size_t smem = (size_t)RR * W * Cc * 2 + (size_t)(K * K + 2) * Cc * 4;
    if (smem < (size_t)32 * Cc) smem = (size_t)32 * Cc;             // the statistics epilogue's [4 waves][2][Cc] scratch
    if (smem > 64 * 1024) return 1;
    const int units = R * sh * W * (Cc / 8);                        // staged per step
    const int U = (units + 255) / 256;
    if (U > 8) return 1;
    DwRingArgs a{(const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, in_scale, in_shift, in_ctrue,
                 B, H, W, C, OH, OW, sh, sw, c_true, act, alpha, Cc, RR, R, stats};
    if (stats && (256 % (Cc / 8) || C % Cc)) return 1;      // a thread keeps one channel group; full chunks only
    const dim3 grid(B, (C + Cc - 1) / Cc);
#define DWR_CASE(K_, U_) if (K == K_ && U <= U_) { dw_fprop_ring_kernel<K_, U_><<<grid, 256, smem, s>>>(a); return 0; }
    DWR_CASE(1, 1) DWR_CASE(1, 2) DWR_CASE(1, 4) DWR_CASE(1, 8)
    DWR_CASE(2, 1) DWR_CASE(2, 2) DWR_CASE(2, 4) DWR_CASE(2, 8)
    DWR_CASE(3, 1) DWR_CASE(3, 2) DWR_CASE(3, 4) DWR_CASE(3, 8)
#undef DWR_CASE
    return 1;
}
