// Transposed convolution FROM a single-channel image (a Conv2DTranspose as the first layer of a random_layer
// classifier, /root/reference/DDL_Normal4.py:39-43; model0 of the shipped replicas: 28x28x1 -> 57x57x64, k3 s2):
//
//   y[b][h][w][n] = act(bias[n] + sum_{kh,kw} [sh | h+pt-kh*dh][sw | w+pl-kw*dw] x[b][(h+pt-kh*dh)/sh][(w+pl-kw*dw)/sw] * wgt[n][kh][kw])
//
// and its weight gradient dW[n][kh][kw] = sum_{b,h,w} dY[b][h][w][n] * x[...same pixel...].
//
// With one input channel the layer has K = taps <= 9 multiply-adds per output value: there is nothing for the matrix
// cores to do (the implicit-GEMM kernels pad K to 32 and ran this layer at 1.4 TB/s of stores, the streaming weight
// gradient needed an im2col tensor and read 388 MB for a 213 MB dY).  Both passes are pure streaming problems, so
// they are written as such: a lane owns 8 output channels of one pixel (one 16-byte store / load, the 8 lanes of a
// pixel cover a whole 128-byte line at N = 64), the image sits in LDS as a zero-haloed fp32 plane, the lane's
// 8 x taps weights (forward) or 8 x taps accumulators (gradient) live in registers.  A block owns ONE stride class
// (h+pt mod sh, w+pl mod sw) of one image, so which taps are live is block-uniform: dead taps are skipped by scalar
// branches (a stride-2 3x3 layer has 2.25 live taps per pixel on average, not 9) and tap offsets into the plane are
// scalars.  The forward pass also leaves the BatchNorm statistics' partial sums of its (bf16-rounded) output, one
// column per block, laid out like pconv's STATS epilogue (part[2][Np][columns]).
//
// The same two kernels serve the FORWARD direction (`fwdconv`: a Conv2D from the one-channel image with a kernel of at
// most 3 x 3 -- the first layer of four of the five shipped replicas): one "class" holding every output pixel, every
// tap live, the plane indexed at (h * sh + kh * dh, w * sw + kw * dw) with the padding as its halo.
#include "common.h"
#include <cstdlib>

struct UpArgs {
    const bf16_t* x;          // [B][IH][IW][Cs], channel 0 real
    const bf16_t* w;          // [N][KH*KW][Cw], channel 0 real (canonical [layer_out][KH][KW][pad(layer_in)])
    const float* bias;        // [N] or null
    bf16_t* y;                // forward: output [B][H][W][Np]; gradient: dY (read only)
    float* part;              // forward: statistics partials (or null); gradient: slab [blocks][T * Np]
    int stats_stride, stats_off;
    int B, IH, IW, Cs, Cw, H, W, Np, N, KH, KW, sh, sw, dh, dw, pt, pl, act;
    float alpha;
    int PH, PW, halo_h, halo_w;   // plane[(ih + halo_h) * PW + iw + halo_w]
    int bands, rpb;               // bands per class, class rows per band
    int ipb;                      // images per block (weight gradient: fewer, fatter slabs)
    int lps;                      // log2(Np / 8): lanes per pixel
    int fwdconv;                  // 0: transposed convolution (stride classes); 1: forward convolution
    // weight gradient only (BNB instantiation): the layer is followed by a batch-statistics BatchNorm and `y` is the
    // gradient at that BatchNorm's OUTPUT; the gradient at its input -- dx = act'(bx) * (k0 g + k1 bx + k2), what
    // bn_bwd_apply would write -- is formed on load from g and the BatchNorm's input bx (= this layer's own output)
    // and never stored: the first layer is its only consumer.
    const bf16_t* bx; const float *bmean, *binvstd, *bgamma, *bc1, *bc2;
    int bact; float balpha;
    // stream1_wgrad_kernel only: per-channel affine applied to the streamed operand on load (a BatchNorm folded into
    // its consumer), and whether the block also sums its image plane (the scalar bias gradient of a C -> 1 layer)
    const float *as, *ah;
    int actrue, planesum;
};

__device__ __forceinline__ int up_fdiv(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }
__device__ __forceinline__ int up_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// Block-uniform description of the class / band this block owns.
struct UpBlock {
    int h_first, w_first, ncols, i0, i1, m_first, n_first;
    int oh_step, ow_step, ih_step, iw_step;   // output pixel (h_first + oh_step * i, ..) reads plane row m_first + ih_step * i (+ halo)
    int toff[25];
    unsigned live;            // bit t: tap t contributes to this class
};

template <int K>
__device__ __forceinline__ UpBlock up_block(const UpArgs& a) {
    UpBlock u;
    if (a.fwdconv) {
        const int band = blockIdx.y;
        u.h_first = u.w_first = 0;
        u.ncols = a.W;
        u.i0 = min(a.H, band * a.rpb);
        u.i1 = min(a.H, u.i0 + a.rpb);
        u.m_first = -a.halo_h;              // plane row of output row i, tap row 0: i * sh (the halo IS the padding)
        u.n_first = -a.halo_w;
        u.oh_step = u.ow_step = 1;
        u.ih_step = a.sh;
        u.iw_step = a.sw;
        u.live = (1u << (K * K)) - 1u;
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) u.toff[kh * K + kw] = kh * a.dh * a.PW + kw * a.dw;
        return u;
    }
    const int cls = blockIdx.y / a.bands, band = blockIdx.y - cls * a.bands;
    const int ph = cls / a.sw, pw = cls - ph * a.sw;
    u.h_first = up_mod(ph - a.pt, a.sh);
    u.w_first = up_mod(pw - a.pl, a.sw);
    const int nrows = u.h_first < a.H ? (a.H - u.h_first + a.sh - 1) / a.sh : 0;
    u.ncols = u.w_first < a.W ? (a.W - u.w_first + a.sw - 1) / a.sw : 0;
    u.i0 = min(nrows, band * a.rpb);
    u.i1 = min(nrows, u.i0 + a.rpb);
    u.m_first = (u.h_first + a.pt - ph) / a.sh;
    u.n_first = (u.w_first + a.pl - pw) / a.sw;
    u.oh_step = a.sh;
    u.ow_step = a.sw;
    u.ih_step = u.iw_step = 1;
    u.live = 0;
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int rh = ph - kh * a.dh, rw = pw - kw * a.dw;
            const bool ok = (rh % a.sh == 0) && (rw % a.sw == 0);
            u.toff[kh * K + kw] = ok ? (rh / a.sh) * a.PW + (rw / a.sw) : 0;
            if (ok) u.live |= 1u << (kh * K + kw);
        }
    return u;
}

// zero-haloed fp32 plane of image b
__device__ __forceinline__ void up_stage(const UpArgs& a, float* plane, int b) {
    const int tid = threadIdx.x;
    for (int i = tid; i < a.PH * a.PW; i += 256) plane[i] = 0.f;
    __syncthreads();
    const bf16_t* src = a.x + (size_t)b * a.IH * a.IW * a.Cs;
    const float invW = 1.0f / (float)a.IW;
    const int n = a.IH * a.IW;
    for (int i0 = tid; i0 < n; i0 += 1024) {
        bf16_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = min(i0 + k * 256, n - 1);
            v[k] = src[(size_t)i * a.Cs];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * 256;
            if (i < n) {
                const int r = up_fdiv(i, invW), c = i - r * a.IW;
                plane[(r + a.halo_h) * a.PW + c + a.halo_w] = bf2f(v[k]);
            }
        }
    }
    __syncthreads();
}

template <int K>
__global__ __launch_bounds__(256) void upconv_1ch_fwd_kernel(const UpArgs a) {
    constexpr int T = K * K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* plane = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const UpBlock u = up_block<K>(a);
    const int LP = 1 << a.lps;
    const int cg = tid & (LP - 1);
    // this lane's 8 channels: bias in registers, the weights of every tap in LDS as fp32 [T][Np] (72 registers of
    // weights left the kernel at two waves per SIMD; a live tap now costs two 16-byte LDS reads per lane)
    float* wsm = plane + ((a.PH * a.PW + 3) & ~3);
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = cg * 8 + j;
        bv[j] = (n < a.N && a.bias) ? a.bias[n] : 0.f;
    }
    for (int i = tid; i < T * a.Np; i += 256) {
        const int t = i / a.Np, n = i - t * a.Np;
        wsm[i] = n < a.N ? bf2f(a.w[((size_t)n * T + t) * a.Cw]) : 0.f;
    }
    up_stage(a, plane, b);

    float st_s[8], st_q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st_s[j] = st_q[j] = 0.f;
    const int total = ((u.i1 - u.i0) * u.ncols) << a.lps;
    const float invC = 1.0f / (float)max(u.ncols, 1);
    const int act = a.act;
    const float alpha = a.alpha;
    bf16_t* yb = a.y + (size_t)b * a.H * a.W * a.Np;         // (an image is far below 2^31 elements: up_fill)
    const bool padded = a.N < a.Np;
    auto body = [&](auto KA) {
        for (int idx = tid; idx < total; idx += 256) {
            const int q = idx >> a.lps;
            const int il = up_fdiv(q, invC), j = q - il * u.ncols;
            const int i = u.i0 + il;
            const int h = u.h_first + u.oh_step * i, w = u.w_first + u.ow_step * j;
            const int pb = (u.m_first + u.ih_step * i + a.halo_h) * a.PW + u.n_first + u.iw_step * j + a.halo_w;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = bv[e];
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (u.live & (1u << t)) {
                    const float xv = plane[pb + u.toff[t]];
                    const float4 w0 = *(const float4*)(wsm + t * a.Np + cg * 8);
                    const float4 w1 = *(const float4*)(wsm + t * a.Np + cg * 8 + 4);
                    acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]);
                    acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
                    acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]);
                    acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
                }
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = act_apply_k<decltype(KA)::value>(acc[e], act, alpha);
                v[e] = (!padded || cg * 8 + e < a.N) ? z : 0.f;
            }
            const uint4 o = pack8(v);
            *(uint4*)(yb + (unsigned)((h * a.W + w) * a.Np + cg * 8)) = o;
            if (a.part) {
                float r[8];
                unpack8(o, r);               // what the stored tensor holds
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    st_s[e] += r[e];
                    st_q[e] = fmaf(r[e], r[e], st_q[e]);
                }
            }
        }
    };
    act_switch(act, body);

    if (a.part) {
        // lanes with the same channel group hold different pixels: fold them, then the four waves
        for (int o = LP; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                st_s[e] += __shfl_xor(st_s[e], o, 64);
                st_q[e] += __shfl_xor(st_q[e], o, 64);
            }
        __syncthreads();                     // the plane is no longer read
        float* red = (float*)smem;           // [4 waves][2][Np]
        if (lane < LP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * a.Np + cg * 8 + e] = st_s[e];
                red[(wave * 2 + 1) * a.Np + cg * 8 + e] = st_q[e];
            }
        }
        __syncthreads();
        const size_t col = (size_t)a.stats_off + (size_t)blockIdx.x * gridDim.y + blockIdx.y;
        for (int i = tid; i < 2 * a.Np; i += 256) {
            const int k = i / a.Np, c = i - k * a.Np;
            const float s = ((red[(0 * 2 + k) * a.Np + c] + red[(1 * 2 + k) * a.Np + c]) + red[(2 * 2 + k) * a.Np + c]) +
                            red[(3 * 2 + k) * a.Np + c];
            a.part[((size_t)k * a.Np + c) * a.stats_stride + col] = s;
        }
    }
}

// dW partials: slab[block][t * Np + n] = sum over the block's pixels of dY[.][n] * x[tap t of the pixel]
template <int K, bool BNB>
__global__ __launch_bounds__(256) void upconv_1ch_wgrad_kernel(const UpArgs a) {
    constexpr int T = K * K, U = BNB ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* plane = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const UpBlock u = up_block<K>(a);
    const int LP = 1 << a.lps;
    const int cg = tid & (LP - 1);
    // BNB: k0 = gamma * is, k1 = -gamma * is * is * c2, k2 = -gamma * is * (c1 - mean * is * c2) per channel, in LDS
    // behind the plane (24 registers of coefficients would cost a wave of occupancy)
    float* kco = plane + ((a.PH * a.PW + 3) & ~3);          // [3][Np]
    if (BNB) {
        for (int c = tid; c < a.Np; c += 256) {
            const bool real = c < a.N;
            const float is = real ? a.binvstd[c] : 0.f, gi = real ? a.bgamma[c] * is : 0.f;
            const float c2 = real ? a.bc2[c] : 0.f, c1 = real ? a.bc1[c] : 0.f, mu = real ? a.bmean[c] : 0.f;
            kco[c] = gi;
            kco[a.Np + c] = -gi * is * c2;
            kco[2 * a.Np + c] = -gi * (c1 - mu * is * c2);
        }
    }

    float acc[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const int total = ((u.i1 - u.i0) * u.ncols) << a.lps;
    const float invC = 1.0f / (float)max(u.ncols, 1);
    for (int b = blockIdx.x * a.ipb; b < min(a.B, (int)(blockIdx.x + 1) * a.ipb); ++b) {
    __syncthreads();                         // every lane is done with the previous image's plane
    up_stage(a, plane, b);
    const bf16_t* yb = a.y + (size_t)b * a.H * a.W * a.Np;
    const bf16_t* xbb = BNB ? a.bx + (size_t)b * a.H * a.W * a.Np : nullptr;
    for (int base = tid; base < total; base += 256 * U) {
        uint4 v[U], vx[BNB ? U : 1];
        int pb[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int idx = base + k * 256;
            ok[k] = idx < total;
            const int q = min(idx, total - 1) >> a.lps;
            const int il = up_fdiv(q, invC), j = q - il * u.ncols;
            const int i = u.i0 + il;
            const int h = u.h_first + u.oh_step * i, w = u.w_first + u.ow_step * j;
            pb[k] = (u.m_first + u.ih_step * i + a.halo_h) * a.PW + u.n_first + u.iw_step * j + a.halo_w;
            v[k] = *(const uint4*)(yb + (unsigned)((h * a.W + w) * a.Np + cg * 8));
            if (BNB) vx[k] = *(const uint4*)(xbb + (unsigned)((h * a.W + w) * a.Np + cg * 8));
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            float d[8];
            unpack8(v[k], d);
            if (BNB) {
                float xf[8], k0[8], k1[8], k2[8];
                unpack8(vx[k], xf);
                *(float4*)k0 = *(const float4*)(kco + cg * 8);  *(float4*)(k0 + 4) = *(const float4*)(kco + cg * 8 + 4);
                *(float4*)k1 = *(const float4*)(kco + a.Np + cg * 8);
                *(float4*)(k1 + 4) = *(const float4*)(kco + a.Np + cg * 8 + 4);
                *(float4*)k2 = *(const float4*)(kco + 2 * a.Np + cg * 8);
                *(float4*)(k2 + 4) = *(const float4*)(kco + 2 * a.Np + cg * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = fmaf(k0[e], d[e], fmaf(k1[e], xf[e], k2[e]));
                act_switch(a.bact, [&](auto KA) {
                    if (decltype(KA)::value != ACT_LINEAR) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) d[e] *= act_grad_y_k<decltype(KA)::value>(xf[e], a.bact, a.balpha);
                    }
                });
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = bf2f(f2bf(d[e]));        // the rounding of the tensor it replaces
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (u.live & (1u << t)) {
                    const float xv = ok[k] ? plane[pb[k] + u.toff[t]] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(xv, d[e], acc[t][e]);
                }
        }
    }
    }
    for (int o = LP; o < 64; o <<= 1)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[t][e] += __shfl_xor(acc[t][e], o, 64);
    __syncthreads();                         // the plane is no longer read
    float* red = (float*)smem;               // [4 waves][T][Np]
    if (lane < LP) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(wave * T + t) * a.Np + cg * 8 + e] = acc[t][e];
    }
    __syncthreads();
    const int E = T * a.Np;
    float* dst = a.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * E;
    for (int i = tid; i < E; i += 256) dst[i] = ((red[i] + red[E + i]) + red[2 * E + i]) + red[3 * E + i];
}

// out[(n * T + t) * Cw + 0] = sum_k slab[k][t * Np + n] (fixed order: deterministic), channels 1 .. Cw-1 = 0.
// A block owns 32 slab columns, 8 lanes split the slabs of a column and combine through LDS.
__global__ __launch_bounds__(256) void upconv_1ch_wgrad_finish_kernel(const float* __restrict__ slab,
                                                                      float* __restrict__ out, int E, int Np, int N,
                                                                      int T, int Cw, int nslab) {
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + col;
    float s = 0.f;
    if (e < E) {
        int k = kl;
        for (; k + 56 < nslab; k += 64) {           // eight independent loads per trip, summed in a fixed order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(size_t)(k + 8 * q) * E + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; k < nslab; k += 8) s += slab[(size_t)k * E + e];
    }
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && e < E) {
#pragma unroll
        for (int q = 1; q < 8; ++q) s += red[q][col];
        const int t = e / Np, n = e - t * Np;
        if (n < N) {
            float* o = out + ((size_t)n * T + t) * Cw;
            o[0] = s;
            for (int c = 1; c < Cw; ++c) o[c] = 0.f;
        }
    }
}

static long up_env(const char* name, long dflt) {
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
}

// `target`: blocks to aim for (bands split a class's rows until there are that many; the weight gradient also folds
// images into one block while there would be twice as many).  Measured on model0's layer (B = 512, 4 classes): forward
// 2048 / 4096 / 8192 / 16384 blocks -> 99 / 104 / 121 / 156 us (every block stages its image and its weights);
// weight gradient 512 / 1024 / 2048 blocks -> 63 / 53 / 68 us (1024 = one round at its register count, half the slabs).
static int up_fill(UpArgs& a, const void* x, const void* w, const float* bias, void* y, float* part, int B, int IH,
                   int IW, int Cs, int Cw, int H, int W, int Np, int N, int KH, int KW, int sh, int sw, int dh, int dw,
                   int pt, int pl, int act, float alpha, long target, int fwdconv, int ch = 8, int kmax = 3) {
    if (B <= 0 || IH <= 0 || IW <= 0 || H <= 0 || W <= 0 || Cs <= 0 || Cw <= 0 || N <= 0 || N > Np || Np % 8) return -1;
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || pt < 0 || pl < 0) return -1;
    if (KH != KW || KH < 1 || KH > kmax) return 1;
    if (Np % ch) return 1;
    const int LP = Np / ch;
    if (LP > (ch == 8 ? 16 : 32) || (LP & (LP - 1))) return 1;
    if ((long)B * H * W * Np >= (1L << 31) || (long)B * IH * IW * Cs >= (1L << 31)) return 1;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.part = part;
    a.stats_stride = 0; a.stats_off = 0;
    a.B = B; a.IH = IH; a.IW = IW; a.Cs = Cs; a.Cw = Cw; a.H = H; a.W = W; a.Np = Np; a.N = N; a.KH = KH; a.KW = KW;
    a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl; a.act = act; a.alpha = alpha;
    a.fwdconv = fwdconv;
    a.as = a.ah = nullptr; a.actrue = 0; a.planesum = 0;
    if (fwdconv) {          // plane rows 0 .. (H-1) sh + (KH-1) dh, the image at row pt
        a.halo_h = pt;
        a.halo_w = pl;
        const int ext_h = (H - 1) * sh + (KH - 1) * dh + 1, ext_w = (W - 1) * sw + (KW - 1) * dw + 1;
        a.PH = pt + IH > ext_h ? pt + IH : ext_h;
        a.PW = pl + IW > ext_w ? pl + IW : ext_w;
    } else {
        a.halo_h = ((KH - 1) * dh + sh - 1) / sh;
        a.halo_w = ((KW - 1) * dw + sw - 1) / sw;
        const int top_h = (H - 1 + pt) / sh + 1, top_w = (W - 1 + pl) / sw + 1;
        a.PH = a.halo_h + (IH > top_h ? IH : top_h);
        a.PW = a.halo_w + (IW > top_w ? IW : top_w);
    }
    int lps = 0;
    while ((1 << lps) < LP) ++lps;
    a.lps = lps;
    // enough blocks to fill the chip: bands split a class's rows when batch x classes alone would not
    const int classes = fwdconv ? 1 : sh * sw, rows = fwdconv ? H : (H + sh - 1) / sh;
    int bands = 1, ipb = 1;
    while ((long)B * classes * bands < target && bands * 2 <= rows) bands *= 2;
    if (bands == 1 && (long)B * classes > target) ipb = (int)(((long)B * classes + target - 1) / target);
    a.bands = bands;
    a.rpb = (rows + bands - 1) / bands;
    a.ipb = ipb;
    if ((long)rows * (fwdconv ? W : (W + sw - 1) / sw) * LP >= (1L << 21)) return 1;     // up_fdiv's exact range
    return 0;
}

static long up_target(int wgrad) {     // 0: forward, 1: weight gradient, 2: weight gradient with the BatchNorm backward on load
    static const long tf = up_env("DDL_UP_FWD_BLOCKS", 2048), tw = up_env("DDL_UP_WG_BLOCKS", 1024),
                      tb = up_env("DDL_UP_WGBN_BLOCKS", 768);      // (149 registers: three blocks per CU, one round)
    return wgrad == 2 ? tb : (wgrad ? tw : tf);
}

// Statistics columns (forward, wgrad = 0) / slabs (weight gradient, wgrad = 1) of a launch.  0: outside the kernels' range.
DDL_API int ddl_upconv_1ch_blocks(int wgrad, int B, int H, int W, int Np, int KH, int KW, int sh, int sw, int fwdconv) {
    UpArgs a;
    if (up_fill(a, nullptr, nullptr, nullptr, nullptr, nullptr, B, 1, 1, 8, 8, H, W, Np, Np, KH, KW, sh, sw, 1, 1, 0, 0, 0,
                0.f, up_target(wgrad), fwdconv))
        return 0;
    return ((B + a.ipb - 1) / a.ipb) * (fwdconv ? 1 : sh * sw) * a.bands;
}

// Returns 0 on success, 1 if the shape is outside this kernel's range (nothing launched), < 0 on bad arguments.
DDL_API int ddl_upconv_1ch(const void* x, const void* w, const float* bias, void* y, float* stats, int stats_stride,
                           int stats_off, int B, int IH, int IW, int Cs, int Cw, int H, int W, int Np, int N, int KH,
                           int KW, int sh, int sw, int dh, int dw, int pt, int pl, int act, float alpha, int fwdconv,
                           hipStream_t s) {
    UpArgs a;
    const int rc = up_fill(a, x, w, bias, y, stats, B, IH, IW, Cs, Cw, H, W, Np, N, KH, KW, sh, sw, dh, dw, pt, pl, act,
                           alpha, up_target(0), fwdconv);
    if (rc) return rc;
    a.ipb = 1;
    const int gy = (fwdconv ? 1 : sh * sw) * a.bands;
    a.stats_stride = stats_stride > 0 ? stats_stride : B * gy;
    a.stats_off = stats_stride > 0 ? stats_off : 0;
    if (stats && (a.stats_off < 0 || a.stats_off + B * gy > a.stats_stride)) return -1;
    size_t smem = (size_t)((a.PH * a.PW + 3) & ~3) * 4 + (size_t)KH * KW * Np * 4;      // plane + weights
    const size_t red = (size_t)4 * 2 * Np * 4;
    if (red > smem) smem = red;
    if (smem > 64 * 1024) return 1;
    const dim3 grid(B, gy);
    if (KH == 1) upconv_1ch_fwd_kernel<1><<<grid, 256, smem, s>>>(a);
    else if (KH == 2) upconv_1ch_fwd_kernel<2><<<grid, 256, smem, s>>>(a);
    else upconv_1ch_fwd_kernel<3><<<grid, 256, smem, s>>>(a);
    return 0;
}

// dW into the canonical fp32 gradient [N][KH*KW][Cw] (channel 0; the padding channels are written as zeros).
// `slab`: ddl_upconv_1ch_blocks(1, ...) * KH*KW*Np floats of scratch.
// bn_x (optional, with bn_mean .. bn_c2 fp32 [>= N]): dy is the gradient at the OUTPUT of the batch-statistics BatchNorm
// behind the layer, bn_x that BatchNorm's input; its backward apply step (coefficients c1, c2 from ddl_bn_bwd's finalize,
// bn_act = the layer's own activation, whose derivative is taken through bn_x) is performed on load.
DDL_API int ddl_upconv_1ch_wgrad(const void* x, const void* dy, float* slab, float* out, int B, int IH, int IW, int Cs,
                                 int Cw, int H, int W, int Np, int N, int KH, int KW, int sh, int sw, int dh, int dw,
                                 int pt, int pl, const void* bn_x, const float* bn_mean, const float* bn_invstd,
                                 const float* bn_gamma, const float* bn_c1, const float* bn_c2, int bn_act,
                                 float bn_alpha, int fwdconv, hipStream_t s) {
    UpArgs a;
    const int rc = up_fill(a, x, nullptr, nullptr, const_cast<void*>(dy), slab, B, IH, IW, Cs, Cw, H, W, Np, N, KH, KW,
                           sh, sw, dh, dw, pt, pl, 0, 0.f, up_target(bn_x ? 2 : 1), fwdconv);
    if (rc) return rc;
    if (!slab || !out) return -1;
    if (bn_x && !(bn_mean && bn_invstd && bn_gamma && bn_c1 && bn_c2)) return -1;
    a.bx = (const bf16_t*)bn_x; a.bmean = bn_mean; a.binvstd = bn_invstd; a.bgamma = bn_gamma; a.bc1 = bn_c1; a.bc2 = bn_c2;
    a.bact = bn_act; a.balpha = bn_alpha;
    const int gy = (fwdconv ? 1 : sh * sw) * a.bands, T = KH * KW;
    size_t smem = (size_t)((a.PH * a.PW + 3) & ~3) * 4 + (bn_x ? (size_t)3 * Np * 4 : 0);
    const size_t red = (size_t)4 * T * Np * 4;
    if (red > smem) smem = red;
    if (smem > 64 * 1024) return 1;
    const int gx = (B + a.ipb - 1) / a.ipb;
    const dim3 grid(gx, gy);
    if (bn_x) {
        if (KH == 1) upconv_1ch_wgrad_kernel<1, true><<<grid, 256, smem, s>>>(a);
        else if (KH == 2) upconv_1ch_wgrad_kernel<2, true><<<grid, 256, smem, s>>>(a);
        else upconv_1ch_wgrad_kernel<3, true><<<grid, 256, smem, s>>>(a);
    } else {
        if (KH == 1) upconv_1ch_wgrad_kernel<1, false><<<grid, 256, smem, s>>>(a);
        else if (KH == 2) upconv_1ch_wgrad_kernel<2, false><<<grid, 256, smem, s>>>(a);
        else upconv_1ch_wgrad_kernel<3, false><<<grid, 256, smem, s>>>(a);
    }
    const int E = T * Np;
    upconv_1ch_wgrad_finish_kernel<<<(E + 31) / 32, 256, 0, s>>>(slab, out, E, Np, N, T, Cw, gx * gy);
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a LARGER kernel (k = 4, 5) between a one-channel image and an N-channel tensor:
//
//   dW[n][kh][kw] = sum_{b,oh,ow} t[b][oh][ow][n] * img[b][oh*sh + kh*dh - pt][ow*sw + kw*dw - pl]
//
// This is the weight gradient of BOTH ends of the DCGAN pair (/root/reference/DCGAN.py:22-66): the discriminator's
// first Conv2D (img = the input image, t = dY) and the generator's last Conv2DTranspose (img = the gradient at its
// one-channel output, t = the layer's input, optionally with the BatchNorm in front of it applied on load).  The
// matrix-core path needed a taps tensor of the image (im2col), a split-K GEMM with 25 of 32 K columns live and a slab
// reduce -- 29 + 44 us of kernel time in the 1.14 ms headline step for 12 MB operands.  Here the image sits in LDS, a
// lane owns CH channels of one pixel of t and 25 x CH accumulators, and t is read exactly once: 31 + 18 us at first
// (index arithmetic and per-tap address adds cost more than the multiply-adds), then tap offsets as immediates of
// paired LDS reads, a division-free pixel walk and all of a block's images staged behind one barrier pair.  Both
// layers' kernels take as long inside the step as the path they replace (see s1_ch), so the executor uses this kernel
// from 2^27 operand elements up (network.S1_WGRAD_MIN_ELEMS): there it removes an 8.6 GB taps tensor and 1 ms per step.
template <int CH> struct S1Vec;
template <> struct S1Vec<2> { typedef unsigned type; };
template <> struct S1Vec<4> { typedef unsigned type __attribute__((ext_vector_type(2))); };

template <int CH>
__device__ __forceinline__ void s1_unpack(const typename S1Vec<CH>::type& v, float* d) {
    if constexpr (CH == 2) {
        d[0] = __uint_as_float(v << 16);
        d[1] = __uint_as_float(v & 0xffff0000u);
    } else {
#pragma unroll
        for (int i = 0; i < CH / 2; ++i) {
            const unsigned w = v[i];
            d[2 * i] = __uint_as_float(w << 16);
            d[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
        }
    }
}

// zero-haloed fp32 planes of images b0 .. b0 + nimg - 1, `ps` floats apart: ONE barrier pair for all of them (a block
// that stages image by image spends its time in the load latency, not in the 25 taps)
__device__ __forceinline__ void s1_stage(const UpArgs& a, float* plane, int b0, int nimg, int ps) {
    const int tid = threadIdx.x;
    for (int i = tid; i < nimg * ps; i += 256) plane[i] = 0.f;
    __syncthreads();
    const bf16_t* src = a.x + (size_t)b0 * a.IH * a.IW * a.Cs;
    const float invW = 1.0f / (float)a.IW, invI = 1.0f / (float)(a.IH * a.IW);
    const int per = a.IH * a.IW, n = nimg * per;
    for (int i0 = tid; i0 < n; i0 += 2048) {
        bf16_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = min(i0 + k * 256, n - 1);
            v[k] = src[(size_t)i * a.Cs];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 256;
            if (i < n) {
                const int im = up_fdiv(i, invI), p = i - im * per;
                const int r = up_fdiv(p, invW), c = p - r * a.IW;
                plane[im * ps + (r + a.halo_h) * a.PW + c + a.halo_w] = bf2f(v[k]);
            }
        }
    }
    __syncthreads();
}

// slab[block][(T + 1) * Np + 1]: rows 0 .. T-1 the taps, row T the column sums of t, the last element the plane sum
// PWC: the plane's row pitch as a compile-time constant (0: a.PW, any dilation) -- tap offsets become the immediate
// offsets of the LDS reads (paired into two-address reads) instead of one address add per tap.
template <int K, int CH, int PWC>
__global__ __launch_bounds__(256) void stream1_wgrad_kernel(const UpArgs a) {
    constexpr int T = K * K, U = 4;
    typedef typename S1Vec<CH>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* plane = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const UpBlock u = up_block<K>(a);
    const int LP = 1 << a.lps;
    const int cg = tid & (LP - 1);
    const bool affine = a.as != nullptr;
    float sc[CH], sf[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int c = cg * CH + e;
        const bool real = affine && c < a.actrue;
        sc[e] = real ? a.as[c] : 0.f;
        sf[e] = real ? a.ah[c] : 0.f;
    }

    float acc[T][CH], accb[CH], psum = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int e = 0; e < CH; ++e) accb[e] = 0.f;
    const int per = (u.i1 - u.i0) * u.ncols;                // pixels of one image's band
    const float invC = 1.0f / (float)max(u.ncols, 1), invP = 1.0f / (float)max(per, 1);
    const int ps = (a.PH * a.PW + 3) & ~3;
    // a block takes image groups blockIdx.x, + gridDim.x, ...: the slab count stays at one round of blocks however
    // large the batch (64 x 64 images at B = 32768: two images fit the LDS budget, 16384 groups)
    const int ngrp = (a.B + a.ipb - 1) / a.ipb;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * a.ipb, nimg = min(a.B, b0 + a.ipb) - b0;
        __syncthreads();                     // every lane is done with the previous group's planes
        s1_stage(a, plane, b0, nimg, ps);
        if (a.planesum && blockIdx.y == 0)   // (the halos are zero: the planes' sum is the images')
            for (int i = tid; i < nimg * ps; i += 256) psum += plane[i];
        const bf16_t* yb = a.y + (size_t)b0 * a.H * a.W * a.Np;
        const int total = (nimg * per) << a.lps;
        // A lane walks pixels tid / LP, + D, + 2 D, ... (D = 256 / LP) of the block's images: (image, row, column) are
        // carried along and advanced by (D / ncols, D % ncols) -- no division and no 32-bit multiply per pixel (the
        // first version's index arithmetic cost as many cycles as its 25 multiply-adds).  s1_fill guarantees
        // D / ncols + 1 <= rows, so one wrap per step suffices.
        const int D = 256 >> a.lps, rows = u.i1 - u.i0;
        const int dR = D / max(u.ncols, 1), dC = D - dR * u.ncols;
        int wim, wil, wj;
        {
            const int q0 = tid >> a.lps;
            wim = up_fdiv(q0, invP);
            const int r = q0 - wim * per;
            wil = up_fdiv(r, invC);
            wj = r - wil * u.ncols;
        }
        const int pw = PWC ? PWC : a.PW;
        const int pconst = (u.m_first + u.ih_step * u.i0 + a.halo_h) * pw + u.n_first + a.halo_w;
        const int gconst = u.i0 * a.W * a.Np + cg * CH, gimg = a.H * a.W * a.Np, grow = a.W * a.Np;
        const int prow = u.ih_step * pw;
        // the loads of trip n+1 are in flight while trip n is multiplied (two or three waves per SIMD do not hide a
        // global load behind each other)
        vec_t vn[U];
        int pbn[U];
        bool okn[U];
        auto issue = [&]() {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const bool ok = wim < nimg;             // (past the end: plane offset 0 and element 0, multiplied by 0)
                okn[k] = ok;
                const int po = __mul24(wim, ps) + __mul24(wil, prow) + __mul24(wj, u.iw_step) + pconst;
                const int go = __mul24(wim, gimg) + __mul24(wil, grow) + __mul24(wj, a.Np) + gconst;
                pbn[k] = ok ? po : 0;
                vn[k] = *(const vec_t*)(yb + (unsigned)(ok ? go : 0));
                wj += dC;
                wil += dR;
                if (wj >= u.ncols) { wj -= u.ncols; ++wil; }
                if (wil >= rows) { wil -= rows; ++wim; }
            }
        };
        if (total > 0) issue();
        for (int base = tid; base < total; base += 256 * U) {
            vec_t v[U];
            int pb[U];
            bool okc[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                v[k] = vn[k];
                pb[k] = pbn[k];
                okc[k] = okn[k];
            }
            issue();
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const bool ok = okc[k];
                float d[CH];
                s1_unpack<CH>(v[k], d);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    if (affine) d[e] = bf2f(f2bf(fmaf(d[e], sc[e], sf[e])));     // the rounding of the tensor it replaces
                    d[e] = ok ? d[e] : 0.f;
                    accb[e] += d[e];
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float xv = plane[pb[k] + (PWC ? (t / K) * PWC + (t % K) : u.toff[t])];
#pragma unroll
                    for (int e = 0; e < CH; ++e) acc[t][e] = fmaf(xv, d[e], acc[t][e]);
                }
                if (!PWC) __builtin_amdgcn_sched_barrier(0);      // (run-time tap offsets: hoisted address adds cost a wave)
            }
        }
    }
    for (int o = LP; o < 64; o <<= 1) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[t][e] += __shfl_xor(acc[t][e], o, 64);
#pragma unroll
        for (int e = 0; e < CH; ++e) accb[e] += __shfl_xor(accb[e], o, 64);
    }
    for (int o = 1; o < 64; o <<= 1) psum += __shfl_xor(psum, o, 64);
    __syncthreads();                         // the plane is no longer read
    float* red = (float*)smem;               // [4 waves][(T + 1) * Np] + [4]
    const int E = (T + 1) * a.Np;
    if (lane < LP) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < CH; ++e) red[wave * E + t * a.Np + cg * CH + e] = acc[t][e];
#pragma unroll
        for (int e = 0; e < CH; ++e) red[wave * E + T * a.Np + cg * CH + e] = accb[e];
    }
    if (lane == 0) red[4 * E + wave] = psum;
    __syncthreads();
    float* dst = a.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (E + 1);
    for (int i = tid; i < E; i += 256) dst[i] = ((red[i] + red[E + i]) + red[2 * E + i]) + red[3 * E + i];
    if (tid == 0) dst[E] = ((red[4 * E] + red[4 * E + 1]) + red[4 * E + 2]) + red[4 * E + 3];
}

// out[n * sn + t * st] = sum_k slab[k][t * Np + n] (+ zfill - 1 zeros behind it), the column sums / the plane sum into
// `bias` (bias_mode 1: bias[n]; 2: bias[0]).  Fixed summation order: deterministic.
__global__ __launch_bounds__(256) void stream1_wgrad_finish_kernel(const float* __restrict__ slab,
                                                                   float* __restrict__ out, float* __restrict__ bias,
                                                                   int E1, int Np, int N, int T, long sn, long st,
                                                                   int zfill, int bias_mode, int nslab) {
    __shared__ float red[32][9];
    const int col = threadIdx.x & 7, kl = threadIdx.x >> 3;       // 8 columns, 32 lanes over the slabs
    const int e = blockIdx.x * 8 + col;
    float s = 0.f;
    if (e < E1) {
        int k = kl;
        for (; k + 96 < nslab; k += 128) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = slab[(size_t)(k + 32 * q) * E1 + e];
#pragma unroll
            for (int q = 0; q < 4; ++q) s += v[q];
        }
        for (; k < nslab; k += 32) s += slab[(size_t)k * E1 + e];
    }
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && e < E1) {
#pragma unroll
        for (int q = 1; q < 32; ++q) s += red[q][col];
        if (e < T * Np) {
            const int t = e / Np, n = e - t * Np;
            if (n < N) {
                float* o = out + (size_t)n * sn + (size_t)t * st;
                o[0] = s;
                for (int c = 1; c < zfill; ++c) o[c] = 0.f;
            }
        } else if (e < (T + 1) * Np) {
            const int n = e - T * Np;
            if (bias_mode == 1 && n < N) bias[n] = s;
        } else if (bias_mode == 2) {
            bias[0] = s;
        }
    }
}

// Channels per lane.  Measured (DCGAN step, both layers on the side stream): 2 -> 1.154 ms, 4 -> 1.148 ms, the
// matrix-core path with its taps tensor 1.146 .. 1.150 ms; the 64 x 64 pair 149.5 -> 148.0 .. 149.0 ms.
static int s1_ch() {
    static const int ch = (int)up_env("DDL_S1_CH", 4);
    return ch == 2 ? 2 : 4;
}

static int s1_fill(UpArgs& a, const void* img, const void* t, float* slab, int B, int IH, int IW, int Cs, int H, int W,
                   int Np, int N, int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl) {
    // 512 = one round at two blocks per CU; a block stages ALL its images at once (LDS budget 48 KB)
    static const long target = up_env("DDL_S1_WG_BLOCKS", 512);
    if (KH < 4) return 1;                    // (k <= 3 is upconv_1ch_wgrad's)
    const int rc = up_fill(a, img, nullptr, nullptr, const_cast<void*>(t), slab, B, IH, IW, Cs, 8, H, W, Np, N, KH, KW, sh,
                           sw, dh, dw, pt, pl, 0, 0.f, target, 1, s1_ch(), 5);
    if (rc) return rc;
    if (dh == 1 && dw == 1) a.PW = a.PW <= 32 ? 32 : (a.PW <= 72 ? 72 : a.PW);     // the kernels' compile-time pitches
    const long ps = ((long)a.PH * a.PW + 3) & ~3L;
    long cap = (48 * 1024) / (ps * 4);
    if (cap < 1) cap = 1;
    if (a.ipb > cap) a.ipb = (int)cap;
    // up_fdiv's exact range over (image, pixel) of a block, and 32-bit offsets into the block's slice of t
    if ((long)a.ipb * a.rpb * W * (Np / s1_ch()) >= (1L << 21) || (long)a.ipb * IH * IW >= (1L << 21)) return 1;
    if ((long)a.ipb * H * W * Np >= (1L << 31)) return 1;
    // the kernel's pixel walk wraps at most one row and one image per step, and multiplies 24-bit operands
    const int Dpix = 256 / (Np / s1_ch());
    auto short_band = [&]() {
        const int last_rows = H - (a.bands - 1) * a.rpb;
        return last_rows < 1 || Dpix / W + 1 > (last_rows < a.rpb ? last_rows : a.rpb);
    };
    while (a.bands > 1 && short_band()) {        // fewer, taller bands
        a.bands /= 2;
        a.rpb = (H + a.bands - 1) / a.bands;
    }
    if (short_band()) return 1;
    if ((long)H * W * Np >= (1L << 24) || ps * a.ipb >= (1L << 24)) return 1;
    return 0;
}

// Slabs of a launch (each (KH*KW + 1) * Np + 1 floats); 0: outside the kernel's range.
static int s1_gx(const UpArgs& a, int B) {
    static const long cap = up_env("DDL_S1_WG_MAXBLOCKS", 512);
    const long ngrp = (B + a.ipb - 1) / a.ipb;
    return (int)(ngrp < cap ? ngrp : cap);
}

DDL_API int ddl_stream1_wgrad_blocks(int B, int IH, int IW, int Cs, int H, int W, int Np, int KH, int KW, int sh, int sw,
                                     int dh, int dw, int pt, int pl) {
    UpArgs a;
    if (s1_fill(a, nullptr, nullptr, nullptr, B, IH, IW, Cs, H, W, Np, Np, KH, KW, sh, sw, dh, dw, pt, pl)) return 0;
    return s1_gx(a, B) * a.bands;
}

// img [B][IH][IW][Cs] (channel 0), t [B][H][W][Np]; out[n * sn + t * st] (n < N) fp32.  Returns 0 on success, 1 when the
// shape is outside the kernel's range (nothing launched), < 0 on bad arguments.
DDL_API int ddl_stream1_wgrad(const void* img, const void* t, float* slab, float* out, float* bias, int B, int IH, int IW,
                              int Cs, int H, int W, int Np, int N, int KH, int KW, int sh, int sw, int dh, int dw, int pt,
                              int pl, long sn, long st, int zfill, int bias_mode, const float* aff_scale,
                              const float* aff_shift, int aff_ctrue, int slab_first, int finish_slabs, hipStream_t s) {
    UpArgs a;
    const int rc = s1_fill(a, img, t, slab, B, IH, IW, Cs, H, W, Np, N, KH, KW, sh, sw, dh, dw, pt, pl);
    if (rc) return rc;
    if (!slab || !out || sn < 0 || st < 0 || zfill < 1 || bias_mode < 0 || bias_mode > 2) return -1;
    if (slab_first < 0 || finish_slabs < 0) return -1;
    if (bias_mode && !bias) return -1;
    if ((aff_scale == nullptr) != (aff_shift == nullptr) || (aff_scale && (aff_ctrue < 0 || aff_ctrue > Np))) return -1;
    // the image must cover what the taps read: rows 0 .. (H-1) sh + (KH-1) dh - pt may exceed IH only into the halo,
    // which up_fill sized from both
    a.as = aff_scale; a.ah = aff_shift; a.actrue = aff_ctrue; a.planesum = bias_mode == 2;
    const int T = KH * KW, E = (T + 1) * Np;
    size_t smem = (size_t)((a.PH * a.PW + 3) & ~3) * 4 * a.ipb;
    const size_t red = (size_t)(4 * E + 4) * 4;
    if (red > smem) smem = red;
    if (smem > 64 * 1024) return 1;
    // batch slices of one reduction (tensors of 2^31 elements and more): every slice writes its slabs behind the
    // previous one's (`slab_first`), the last call sums all of them (`finish_slabs`, 0 = not yet)
    const int gx = s1_gx(a, B);
    a.part = slab + (size_t)slab_first * (E + 1);
    const dim3 grid(gx, a.bands);
    const int ch = s1_ch();
    const int pwc = (dh == 1 && dw == 1) ? a.PW : 0;           // s1_fill rounded PW up to 32 / 72 where it could
#define S1_LAUNCH(K_, CH_, P_) stream1_wgrad_kernel<K_, CH_, P_><<<grid, 256, smem, s>>>(a)
#define S1_PICK(K_, CH_) (pwc == 32 ? S1_LAUNCH(K_, CH_, 32) : pwc == 72 ? S1_LAUNCH(K_, CH_, 72) : S1_LAUNCH(K_, CH_, 0))
    if (KH == 4) {
        if (ch == 4) S1_PICK(4, 4);
        else S1_PICK(4, 2);
    } else {
        if (ch == 4) S1_PICK(5, 4);
        else S1_PICK(5, 2);
    }
#undef S1_PICK
#undef S1_LAUNCH
    if (finish_slabs > 0) {
        if (finish_slabs < slab_first + gx * a.bands) return -1;
        stream1_wgrad_finish_kernel<<<(E + 1 + 7) / 8, 256, 0, s>>>(slab, out, bias, E + 1, Np, N, T, sn, st, zfill,
                                                                      bias_mode, finish_slabs);
    }
    return 0;
}
