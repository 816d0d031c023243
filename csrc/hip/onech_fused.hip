// Fused "taps GEMM + fold" for transposed convolutions with ONE output channel (DCGAN's last generator
// layer) and for the input gradient of convolutions with ONE input channel (the discriminator's first
// layer):
//
//   z[p][t]      = sum_c x[b][p][c] * w[t][c]                          (1x1 GEMM: C channels -> T taps, MFMA)
//   img[b][h][w] = act(bias + sum_t z[(h + pt - kh*dh)/sh][(w + pl - kw*dw)/sw][t])     (col2im fold)
//
// The unfused path (gemm_sk -> taps-as-channels tensor in HBM -> col2im_1ch) writes and re-reads a
// [B][OH][OW][32] tensor (50 MB at the reference batch) that only ever lives between the two kernels.
// Here a block owns (image, band of output rows): it computes the band's z rows straight from x with
// v_mfma_f32_16x16x32_bf16 -- a lane's B fragment (8 consecutive channels of one pixel) is one 16-byte
// global load, no LDS staging of x -- drops them into LDS as bf16 (the same rounding the HBM round trip
// applied, so results are bit-identical to the unfused path), and folds from LDS exactly like
// col2im_1ch.  T <= 32 taps (two 16-row MFMA tiles), C <= 128.
#include "common.h"
#include <cstdlib>

struct TapFoldArgs {
    const bf16_t* x;          // [B][OH][OW][C]
    const bf16_t* w;          // [T][C]
    const float* bias;        // 1 value or null
    uint4* img;               // [B][H][W][8 bf16] (channel 0 real)
    const bf16_t* mul_y;      // optional: img *= act'(mul_y) (activation backward of the layer below), laid out like img
    int mul_act; float mul_alpha;
    int B, H, W, OH, OW, C, KH, KW, sh, sw, dh, dw, pt, pl, T, act;
    float alpha;
    int band_rows, span_rows, ring_rows;     // rows per fold band, rows a block walks, z rows in the LDS ring
    // optional per-channel affine applied to x as it is loaded (the BatchNorm in front of this layer, never
    // materialised): x' = bf16(x * in_scale[grp][c] + in_shift[grp][c]), grp = (in_b_off + b) / in_group_rows
    const float* in_scale; const float* in_shift;
    int in_group_rows, in_b_off, in_ctrue;
};

#define TF_ZP 68                               // bytes per z pixel in LDS: 32 taps * 2 + 4 (spreads the 2-byte gathers)
#define TF_WP(KP) ((KP) * 2 + 32)

__device__ __forceinline__ int tf_fdiv(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

// A block owns (image, span of output rows) and WALKS the span in bands of `band_rows` rows: the z rows live in an
// LDS ring of `ring_rows` rows (row oh in slot oh % ring_rows), each band computes only the z rows the previous one
// did not leave behind, so every x pixel is read and multiplied ONCE per span.  (The first version gave every band
// its own block: with a 5 x 5 stride-1 kernel and the 4-row bands a 64-wide image allows, each block recomputed a
// 4-row halo -- x was read twice and the layer ran at 1.3 TB/s of useful traffic.)  The x fragments of the next
// band are requested before the fold of the current one.
template <int KS, int SH, int SW>
__global__ __launch_bounds__(256) void tapfold_kernel(const TapFoldArgs a) {
    constexpr int KP = KS * 32, WP = TF_WP(KP);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                            // [32][WP]
    char* sZ = smem + 32 * WP;                  // [ring_rows][OW][TF_ZP]
    const int sh = SH ? SH : a.sh, sw = SW ? SW : a.sw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    const int H0 = blockIdx.y * a.span_rows, H1 = min(a.H, H0 + a.span_rows);
    const int RING = a.ring_rows;
    const float invRING = 1.0f / (float)RING, invOW = 1.0f / (float)a.OW;
    auto slot_of = [&](int oh) { return oh - RING * tf_fdiv(oh, invRING); };     // oh >= 0

    // ---- weights -> LDS (zero rows beyond T, zero k beyond C)
    for (int u = tid; u < 32 * (KP / 8); u += 256) {
        int t = u / (KP / 8), k8 = u - t * (KP / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < a.T && k8 * 8 < a.C) v = *(const uint4*)(a.w + (long)t * a.C + k8 * 8);
        *(uint4*)(sW + t * WP + k8 * 16) = v;
    }
    __syncthreads();
    bf16x8_t wf[2][KS];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[tt][ks] = *(const bf16x8_t*)(sW + (16 * tt + fr) * WP + ks * 64 + fq * 16);

    // ---- input affine of this lane's channels (k = ks * 32 + fq * 8 .. + 7)
    const bool aff = a.in_scale != nullptr;
    float isc[KS][8], ish[KS][8];
    if (aff) {
        const int grp = (a.in_b_off + b) / a.in_group_rows;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = ks * 32 + fq * 8 + j;
                const bool real = c < a.in_ctrue;
                isc[ks][j] = real ? a.in_scale[grp * a.C + c] : 0.f;
                ish[ks][j] = real ? a.in_shift[grp * a.C + c] : 0.f;
            }
    }

    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.OH * a.OW * a.C * 2u), 0x00020000);
    // NF fragments (16 pixels x KS 16-byte loads each) in flight per wave: the kernel streams x once and is bound
    // by how many loads it keeps outstanding (two fragments: 41 us for 100 MB; see profiles/r2_dcgan28_*)
    constexpr int NF = KS == 1 ? 4 : 2;

    // z rows [zs, ze] of the image, 16 pixels per MFMA column block; pixel p of the range sits in z row zs + p / OW
    auto z_load = [&](int zs, int npix, int f0, u32x4_t (&xr)[NF][KS]) {
        const unsigned xb = (unsigned)((b * a.OH + zs) * a.OW * a.C);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int p = (f0 + j) * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 32 + fq * 8;
                const bool ok = (p < npix) & (k < a.C);
                const unsigned off = ok ? (xb + (unsigned)(p * a.C + k)) * 2u : 0xFFFFFFF0u;
                xr[j][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 2);     // nt: read once
            }
        }
    };
    auto z_mma = [&](int zs, int npix, int f0, u32x4_t (&xr)[NF][KS]) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int p = (f0 + j) * 16 + fr;
            const bool okp = p < npix;
            bf16x8_t xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4_t v = xr[j][ks];
                if (aff) {
                    const bool ok = okp & (ks * 32 + fq * 8 < a.C);
                    float f[8];
                    unpack8(make_uint4(v[0], v[1], v[2], v[3]), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = ok ? fmaf(f[e], isc[ks][e], ish[ks][e]) : 0.f;
                    const uint4 r = pack8(f);
                    v = (u32x4_t){r.x, r.y, r.z, r.w};
                }
                xf[ks] = __builtin_bit_cast(bf16x8_t, v);
            }
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tt][ks], xf[ks], acc[tt], 0, 0, 0);
            if (okp) {
                // a lane holds taps 16tt + 4fq .. +3 of pixel p
                const int zr = tf_fdiv(p, invOW), zc = p - zr * a.OW;
                char* zp = sZ + (slot_of(zs + zr) * a.OW + zc) * TF_ZP;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    uint32_t* d = (uint32_t*)(zp + (16 * tt + 4 * fq) * 2);
                    d[0] = pack2bf(acc[tt][0], acc[tt][1]);
                    d[1] = pack2bf(acc[tt][2], acc[tt][3]);
                }
            }
        }
    };
    // the z rows the band of output rows [h0, h1) folds from
    auto z_range = [&](int h0, int h1, int& lo_, int& hi_) {
        const int lo = h0 + a.pt - (a.KH - 1) * a.dh;
        lo_ = lo <= 0 ? 0 : (lo + sh - 1) / sh;
        hi_ = min(a.OH - 1, (h1 - 1 + a.pt) / sh);
    };

    const float b0 = a.bias ? a.bias[0] : 0.f;
    const float invW = 1.0f / (float)a.W;
    // Branch-free fold for the compile-time strides with kernels up to 5x5 (stride 1) / 6x6 (stride 2): every candidate
    // tap is read (clamped address) and masked, rows / columns validated once per pixel -- the generic loop below is a
    // divergent branch per tap with a dependent 2-byte LDS read behind it (the fold was 40 us of the step's two
    // tapfold launches).  Same taps in the same order: same bits.
    constexpr bool FASTS = (SH == 1 && SW == 1) || (SH == 2 && SW == 2);
    const bool fast_fold = FASTS && a.dh == 1 && a.dw == 1 && a.KH <= (SH == 1 ? 5 : 6) && a.KW <= (SW == 1 ? 5 : 6);

    auto walk = [&](auto KA) {
    int z_next = 0;                             // first z row not yet in the ring
    u32x4_t xr[NF][KS];
    int pre_zs = 0, pre_np = -1;                // the range whose first pass of loads is already in flight
    {
        int lo_, hi_;
        z_range(H0, min(H1, H0 + a.band_rows), lo_, hi_);
        z_next = lo_;
        pre_zs = lo_;
        pre_np = max(0, hi_ - lo_ + 1) * a.OW;
        z_load(pre_zs, pre_np, wave * NF, xr);
    }
    for (int h0 = H0; h0 < H1; h0 += a.band_rows) {
        const int h1 = min(H1, h0 + a.band_rows);
        int oh_lo, oh_hi;
        z_range(h0, h1, oh_lo, oh_hi);
        // ---- new z rows [zs, oh_hi] (the loads of the first pass were requested before the previous fold)
        const int zs = pre_zs, npix = pre_np;
        for (int f0 = wave * NF; f0 * 16 < npix; f0 += 4 * NF) {
            if (f0 != wave * NF) z_load(zs, npix, f0, xr);
            z_mma(zs, npix, f0, xr);
        }
        z_next = max(z_next, oh_hi + 1);
        __syncthreads();
        if (h1 < H1) {                          // next band's first pass of loads rides under this band's fold
            int lo_, hi_;
            z_range(h1, min(H1, h1 + a.band_rows), lo_, hi_);
            pre_zs = max(z_next, lo_);
            pre_np = max(0, hi_ - pre_zs + 1) * a.OW;
            z_load(pre_zs, pre_np, wave * NF, xr);
        }
        // ---- fold (identical to col2im_1ch_kernel)
        for (int i = tid; i < (h1 - h0) * a.W; i += 256) {
            int hl = tf_fdiv(i, invW), w = i - hl * a.W, h = h0 + hl;
            float acc = 0.f;
            if (FASTS && fast_fold) {
                constexpr int NI = SH == 1 ? 5 : 3, ST = SH == 1 ? 1 : 2;
                const int hh = h + a.pt, ww = w + a.pl;
                const int ph = ST == 1 ? 0 : (hh & 1), pw = ST == 1 ? 0 : (ww & 1);
                const int ohb = ST == 1 ? hh : (hh >> 1), owb = ST == 1 ? ww : (ww >> 1);
                int roff[NI], coff[NI];
                bool rv[NI], cv[NI];
#pragma unroll
                for (int q = 0; q < NI; ++q) {
                    const int kh = ph + ST * q, ol = ohb - q;
                    rv[q] = (kh < a.KH) & (ol >= oh_lo) & (ol <= oh_hi);
                    roff[q] = slot_of(max(ol, 0)) * a.OW * TF_ZP + kh * a.KW * 2;
                    const int kw = pw + ST * q, oc = owb - q;
                    cv[q] = (kw < a.KW) & ((unsigned)oc < (unsigned)a.OW);
                    coff[q] = oc * TF_ZP + kw * 2;
                }
                float tv[NI * NI];
#pragma unroll
                for (int q = 0; q < NI; ++q)
#pragma unroll
                    for (int r = 0; r < NI; ++r) {
                        const bool ok = rv[q] & cv[r];
                        const bf16_t z = *(const bf16_t*)(sZ + (ok ? roff[q] + coff[r] : 0));
                        tv[q * NI + r] = ok ? bf2f(z) : 0.f;
                    }
#pragma unroll
                for (int t = 0; t < NI * NI; ++t) acc += tv[t];
            } else
            for (int kh = 0; kh < a.KH; ++kh) {
                int th = h + a.pt - kh * a.dh;
                if (th < 0 || th % sh) continue;
                int oh = th / sh;
                if (oh > oh_hi) continue;
                const char* row = sZ + slot_of(oh) * a.OW * TF_ZP + kh * a.KW * 2;
                for (int kw = 0; kw < a.KW; ++kw) {
                    int tw = w + a.pl - kw * a.dw;
                    if (tw < 0 || tw % sw) continue;
                    int ow = tw / sw;
                    if (ow >= a.OW) continue;
                    acc += bf2f(*(const bf16_t*)(row + ow * TF_ZP + kw * 2));
                }
            }
            float v = act_apply_k<decltype(KA)::value>(acc + b0, a.act, a.alpha);
            const long pix = ((long)b * a.H + h) * a.W + w;
            if (a.mul_y) v = bf2f(f2bf(v)) * act_grad_y(bf2f(a.mul_y[pix * 8]), a.mul_act, a.mul_alpha);
            a.img[pix] = make_uint4((uint32_t)f2bf(v), 0, 0, 0);
        }
        __syncthreads();                        // the next band overwrites the ring rows this fold no longer needs
    }
    };
    act_switch(a.act, walk);                    // the loops compiled per activation: no branch chain per pixel
}

template <int KS, int SH, int SW>
static void tapfold_launch(const TapFoldArgs& a, int bands, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipFuncSetAttribute((const void*)tapfold_kernel<KS, SH, SW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
            done = true;
        }
    }
    tapfold_kernel<KS, SH, SW><<<dim3(a.B, bands), 256, smem, s>>>(a);
}

// Returns 0 on success, 1 if the shape is outside this kernel's range (nothing launched), < 0 on bad arguments.
DDL_API int ddl_tapfold_1ch(const void* x, const void* w, const float* bias, void* img, int B, int H, int W, int OH,
                            int OW, int C, int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int act,
                            float alpha, const void* mul_y, int mul_act, float mul_alpha, const float* in_scale,
                            const float* in_shift, int in_group_rows, int in_b_off, int in_ctrue, hipStream_t s) {
    if (C % 8 || C <= 0 || B <= 0 || KH * KW <= 0) return -1;
    if (in_scale && (!in_shift || in_group_rows <= 0 || in_b_off < 0)) return -1;
    if (KH * KW > 32 || C > 128) return 1;
    if ((long)B * OH * OW * C >= (1L << 31) || (long)B * H * W * 8 >= (1L << 31)) return 1;
    TapFoldArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.img = (uint4*)img;
    a.mul_y = (const bf16_t*)mul_y; a.mul_act = mul_act; a.mul_alpha = mul_alpha;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_group_rows = in_group_rows; a.in_b_off = in_b_off;
    a.in_ctrue = in_ctrue;
    a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.C = C; a.KH = KH; a.KW = KW; a.sh = sh; a.sw = sw;
    a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl; a.T = KH * KW; a.act = act; a.alpha = alpha;
    const int KS = (C + 31) / 32;
    // band height: the band's z rows ((rows-1)/sh + halo) stay under ~40 KiB of LDS
    const size_t row_bytes = (size_t)OW * TF_ZP;
    const int halo = ((KH - 1) * dh) / sh + 2;
    static const size_t band_cap = getenv("DDL_TAPFOLD_LDS_KB") ? (size_t)atol(getenv("DDL_TAPFOLD_LDS_KB")) * 1024 : 40 * 1024;
    int band = H;
    while (band > 1 && ((size_t)((band - 1) / sh + halo)) * row_bytes > band_cap) band = (band + 1) / 2;
    const size_t zrows = (size_t)((band - 1) / sh + halo);
    const size_t zbytes = (zrows * OW + 15) / 16 * 16 * TF_ZP;
    const size_t smem = 32 * (size_t)TF_WP(KS * 32) + zbytes;
    if (smem > 160 * 1024 || (long)band * W >= (1L << 21)) return 1;
    a.band_rows = band;
    a.ring_rows = (int)zrows;
    // a block walks `span` rows band by band (ring of z rows: no halo recomputed inside a span); spans only split the
    // image when the batch alone would leave CUs idle
    int spans = 1;
    const int nb = (H + band - 1) / band;
    while ((long)B * spans < 1024 && spans < nb) spans *= 2;
    a.span_rows = ((nb + spans - 1) / spans) * band;
    const int bands = (H + a.span_rows - 1) / a.span_rows;
    const int sv = (sh == 1 && sw == 1) ? 1 : ((sh == 2 && sw == 2) ? 2 : 0);
#define TF_CASE(KS_) \
    if (KS == KS_) { \
        if (sv == 1) tapfold_launch<KS_, 1, 1>(a, bands, smem, s); \
        else if (sv == 2) tapfold_launch<KS_, 2, 2>(a, bands, smem, s); \
        else tapfold_launch<KS_, 0, 0>(a, bands, smem, s); \
        return 0; \
    }
    TF_CASE(1) TF_CASE(2) TF_CASE(3) TF_CASE(4)
#undef TF_CASE
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Direct convolution FROM a single-channel image (the discriminator's first layer; the input gradient of the
// generator's last layer): y[b][oh][ow][n] = act(bias[n] + sum_t img[b][oh*sh + kh*dh - pt][ow*sw + kw*dw - pl] * w[n][t]).
// The unfused path materialises the taps-as-channels tensor (im2col_1ch, [B][OH][OW][32]) and runs a 1x1 GEMM over
// it: 25-50 MB written and read back per call.  Here a block stages its image once into LDS as a zero-haloed plane
// of bf16 scalars (2 KB for 28x28) and every lane assembles its MFMA B fragment -- 8 consecutive taps of one output
// pixel -- from eight 2-byte LDS reads at per-lane constant offsets.  Same MFMA operand order and rounding as
// gemm_sk on the im2col tensor, so results are bit-identical to the unfused path.  T <= 32 taps, N <= 64.
struct TapConvArgs {
    const bf16_t* img;        // [B][H][W][8] (channel 0 real)
    const bf16_t* w;          // [N][Tp] rows (Tp = taps padded to 8)
    const float* bias;        // [>= n_lim] or null
    bf16_t* y;                // [B][OH][OW][Np]
    int B, H, W, OH, OW, KH, KW, sh, sw, dh, dw, pt, pl, T, Tp, Np, n_lim, act;
    float alpha;
    int PH, PW;
    // RED instantiations: y is the gradient arriving at a BatchNorm's output (no activation of its own) whose INPUT
    // is rx [B][OH][OW][Np]; the block also emits that BatchNorm's backward column sums over its image, laid out
    // r_part[RED][Np][gridDim.x] exactly like colreduce_kernel<1> (RED 2: sum g, sum g * xhat) / <3> (RED 5: plus,
    // with a = prev_act'(rx), sum a * g, sum a * rx, sum a) -- the reduction pass over (y, rx) is not needed.
    const bf16_t* rx; const float* r_mean; const float* r_invstd; float* r_part;
    int r_prev_act; float r_prev_alpha; int red_off;
};

// PACT (RED 5): the activation in front of the reduced BatchNorm as a compile-time code (-1: a.r_prev_act at run time)
template <int NT, bool SLOW, int RED, int PACT = -1>
__global__ __launch_bounds__(256) void tapconv_1ch_kernel(const TapConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sI = (bf16_t*)smem;                 // [PH][PW] zero-haloed plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    const int plane = a.PH * a.PW;
    for (int i = tid; i < plane; i += 256) sI[i] = 0;
    __syncthreads();
    {
        // channel 0 of every pixel: 4-byte loads at a 16-byte stride, four in flight per thread before any LDS store
        const uint32_t* src = (const uint32_t*)a.img + (size_t)b * a.H * a.W * 4;
        const float invW = 1.0f / (float)a.W;
        const int HW = a.H * a.W;
        for (int i0 = tid; i0 < HW; i0 += 1024) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 256;
                v[k] = i < HW ? src[(size_t)i * 4] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 256;
                if (i < HW) {
                    const int r = tf_fdiv(i, invW), c = i - r * a.W;
                    sI[(r + a.pt) * a.PW + c + a.pl] = (bf16_t)(v[k] & 0xffffu);
                }
            }
        }
    }
    // weights: lane (row n = 16t + fr, taps fq*8 .. +7), zero beyond Tp / N
    bf16x8_t wf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + fr;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < a.n_lim && fq * 8 < a.Tp) v = *(const uint4*)(a.w + (size_t)n * a.Tp + fq * 8);
        wf[t] = __builtin_bit_cast(bf16x8_t, v);
    }
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = fq * 8 + j;
        const int kh = t / a.KW, kw = t - kh * a.KW;
        toff[j] = t < a.T ? (kh * a.dh) * a.PW + kw * a.dw : 0;
    }
    float bv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = 16 * t + 4 * fq + e;
            bv[t][e] = (a.bias && n < a.n_lim) ? a.bias[n] : 0.f;
        }
    __syncthreads();

    const int npix = a.OH * a.OW;
    const float invOW = 1.0f / (float)a.OW;
    const int act = a.act;
    const float alpha = a.alpha;
    const float slope = act_slope(act, alpha);
    float rs[RED ? RED : 1][NT][4], rmu[RED ? NT : 1][4], ris[RED ? NT : 1][4];
    if (RED) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 16 * t + 4 * fq + e;
                rmu[t][e] = c < a.Np ? a.r_mean[c] : 0.f;
                ris[t][e] = c < a.Np ? a.r_invstd[c] : 0.f;
#pragma unroll
                for (int k = 0; k < RED; ++k) rs[k][t][e] = 0.f;
            }
    }
    // RED: the BatchNorm-input values of a fragment are fetched one fragment ahead (a load issued where it is
    // consumed would expose a full memory latency per fragment: 12 of them per wave)
    uint2 xq[RED ? NT : 1];
    auto fetch_x = [&](int f_, uint2 (&dst_)[RED ? NT : 1]) {
        const int p_ = f_ * 16 + fr;
#pragma unroll
        for (int t = 0; t < (RED ? NT : 0); ++t) {
            const bool ok = (p_ < npix) & (16 * t + 4 * fq < a.Np);
            dst_[t] = ok ? *(const uint2*)(a.rx + ((size_t)b * npix + p_) * a.Np + 16 * t + 4 * fq) : make_uint2(0, 0);
        }
    };
    if (RED) fetch_x(wave, xq);
    for (int f = wave; f * 16 < npix; f += 4) {
        const int p = f * 16 + fr;
        const int pp = p < npix ? p : 0;
        uint2 xcur[RED ? NT : 1];
        if (RED) {
#pragma unroll
            for (int t = 0; t < NT; ++t) xcur[t] = xq[t];
            fetch_x(f + 4, xq);
        }
        const int oh = tf_fdiv(pp, invOW), ow = pp - oh * a.OW;
        const int base = (oh * a.sh) * a.PW + ow * a.sw;
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[j] = (uint32_t)sI[base + toff[2 * j]] | ((uint32_t)sI[base + toff[2 * j + 1]] << 16);
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, (u32x4_t){q[0], q[1], q[2], q[3]});
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // one activation dispatch per fragment for the slow epilogue (common.h: a run-time switch per element is a
        // branch chain per element); the RED 5 activation is a template parameter (a dispatch in here cost 38 registers)
        auto tile = [&](auto KA) {
            bf16_t* dst = a.y + ((size_t)b * npix + p) * a.Np;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[t][e] + bv[t][e];
                    if (SLOW) x = act_apply_k<decltype(KA)::value>(x, act, alpha);
                    else x = act_apply_slope(x, slope);
                    v[e] = (16 * t + 4 * fq + e < a.n_lim) ? x : 0.f;
                }
                if (16 * t + 4 * fq < a.Np) {
                    *(uint2*)(dst + 16 * t + 4 * fq) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    if (RED) {
                        const uint2 xr = xcur[t];
                        const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u),
                                             __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = bf2f(f2bf(v[e]));           // what the stored tensor holds
                            rs[0][t][e] += g;
                            rs[1][t][e] += g * (xv[e] - rmu[t][e]) * ris[t][e];
                            if (RED == 5) {
                                const float ad = act_grad_y_k<PACT>(xv[e], a.r_prev_act, a.r_prev_alpha);
                                rs[RED - 3][t][e] += ad * g;
                                rs[RED - 2][t][e] += ad * xv[e];
                                rs[RED - 1][t][e] += ad;
                            }
                        }
                    }
                }
            }
        };
        if (p < npix) {
            if (SLOW) act_switch(act, tile);
            else tile(std::integral_constant<int, -1>{});
        }
    }
    if (RED) {
        // the 16 lanes sharing fq hold the same channels for 16 different pixels: fold them, then the four waves
#pragma unroll
        for (int k = 0; k < RED; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) rs[k][t][e] += __shfl_xor(rs[k][t][e], o, 64);
        float* red = (float*)(smem + a.red_off);       // [4 waves][RED][NT * 16 channels]
        if (fr == 0) {
#pragma unroll
            for (int k = 0; k < RED; ++k)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[(wave * RED + k) * (NT * 16) + 16 * t + 4 * fq + e] = rs[k][t][e];
        }
        __syncthreads();
        for (int i = tid; i < RED * NT * 16; i += 256) {
            const int k = i / (NT * 16), c = i - k * (NT * 16);
            if (c >= a.Np) continue;
            const float v = ((red[i] + red[RED * NT * 16 + i]) + red[2 * RED * NT * 16 + i]) + red[3 * RED * NT * 16 + i];
            a.r_part[((size_t)k * a.Np + c) * gridDim.x + b] = v;
        }
    }
}

// Returns 0 on success, 1 if the shape is outside this kernel's range (nothing launched), < 0 on bad arguments.
DDL_API int ddl_tapconv_1ch(const void* img, const void* w, const float* bias, void* y, int B, int H, int W, int OH,
                            int OW, int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int Tp, int Np,
                            int n_lim, int act, float alpha, const void* rx, const float* r_mean,
                            const float* r_invstd, float* r_part, int r_sums, int r_prev_act, float r_prev_alpha,
                            hipStream_t s) {
    if (B <= 0 || KH * KW <= 0 || Np % 8 || Tp % 8 || n_lim <= 0 || n_lim > Np) return -1;
    if (rx && (!r_mean || !r_invstd || !r_part || (r_sums != 2 && r_sums != 5))) return -1;
    if (KH * KW > 32 || Tp > 32 || Tp < KH * KW || Np > 64) return 1;
    if ((long)B * OH * OW * Np >= (1L << 31) || (long)B * H * W * 8 >= (1L << 31)) return 1;
    TapConvArgs a;
    a.img = (const bf16_t*)img; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
    a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.KH = KH; a.KW = KW; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw;
    a.pt = pt; a.pl = pl; a.T = KH * KW; a.Tp = Tp; a.Np = Np; a.n_lim = n_lim; a.act = act; a.alpha = alpha;
    // plane: every tap of every output pixel in range, and the whole image (pt / pl may exceed what outputs reach)
    int ph = (OH - 1) * sh + (KH - 1) * dh + 1, pw = (OW - 1) * sw + (KW - 1) * dw + 1;
    if (ph < H + pt) ph = H + pt;
    if (pw < W + pl) pw = W + pl;
    a.PH = ph; a.PW = pw;
    const int NT = (Np + 15) / 16;
    a.rx = (const bf16_t*)rx; a.r_mean = r_mean; a.r_invstd = r_invstd; a.r_part = r_part;
    a.r_prev_act = r_prev_act; a.r_prev_alpha = r_prev_alpha;
    a.red_off = (int)(((size_t)ph * pw * 2 + 15) / 16 * 16);
    const size_t smem = (size_t)a.red_off + (rx ? (size_t)4 * 5 * NT * 16 * 4 : 0);
    if (smem > 64 * 1024 || (long)OH * OW >= (1L << 21) || (long)H * W >= (1L << 21)) return 1;
    const bool fast = act == ACT_LINEAR || act == ACT_RELU || act == ACT_LEAKY;
    if (rx && !fast) return 1;
#define TC_CALL5(A) tapconv_1ch_kernel<TC_NT, false, 5, A><<<B, 256, smem, s>>>(a)
#define TC_RED5(NT_) { constexpr int TC_NT = NT_; ACT1_DISPATCH(r_prev_act, TC_CALL5); }
#define TC_CASE(NT_) \
    if (NT == NT_) { \
        if (rx && r_sums == 5) { TC_RED5(NT_) } \
        else if (rx) tapconv_1ch_kernel<NT_, false, 2><<<B, 256, smem, s>>>(a); \
        else if (fast) tapconv_1ch_kernel<NT_, false, 0><<<B, 256, smem, s>>>(a); \
        else tapconv_1ch_kernel<NT_, true, 0><<<B, 256, smem, s>>>(a); \
        return 0; \
    }
    TC_CASE(1) TC_CASE(2) TC_CASE(3) TC_CASE(4)
#undef TC_CASE
#undef TC_RED5
#undef TC_CALL5
    return 1;
}
