// Weight gradient of the core correlation on the matrix cores:
//
//   dW[n][t][c] = sum_{m=(b,oh,ow)} dY[m][n] * X[b, oh*sh + kh*dh - pt, ow*sw + kw*dw - pl, c]
//
// The reduction index is the PIXEL, but both operands are stored pixel-major / channel-minor
// (NHWC), i.e. k-strided for an MFMA fragment.  gfx950's ds_read_b64_tr_b16 does the transpose
// on the LDS read: tiles are staged exactly as they sit in memory ([pixel][16 channels] sub-tiles
// with a 32-byte row pitch, so the 8 rows a half-wave reads are one contiguous 256-B bank row ->
// conflict-free) and read back column-major.  The MFMA k index of lane group g, element j is
// pixel 4g+j (j<4) / 16+4g+(j-4) (j>=4) for BOTH operands, which is all a dot product needs.
//
// v0 decomposition: block = (k-split, tap, [TN x TC] output tile); the block's 4 waves each own a
// private 32-pixel slice per iteration (no block barrier in the main loop), partial tiles are
// reduced across waves through LDS and written to a per-split slab; a second kernel sums the
// slabs in a fixed order (deterministic, no float atomics).
#include "common.h"
#include <cstdlib>

struct WgradArgs {
    const bf16_t* dy; const bf16_t* x; float* slab;
    int B, OH, OW, Ny;        // dY [B][OH][OW][Ny]
    int H, W, Cx;             // X  [B][H][W][Cx]
    int KH, KW, sh, sw, dh, dw, pt, pl;
    int n_lim, c_lim;         // valid output rows / cols
    long sn, st, sc;          // output strides (elements): index = n*sn + t*st + c*sc
    long slab_stride;         // elements per k-split slab
    int M, chunk;             // total pixels, pixels per k-split (multiple of 128)
    int tiles_c;              // number of c tiles (blockIdx.z = nt * tiles_c + ct)
    unsigned mg_ohw, mg_ow;   // magic multipliers for m / (OH*OW) and rem / OW (0 = power of two: shift only)
    int sf_ohw, sf_ow;
    // optional per-channel affine on X as it is staged (a BatchNorm in front of the layer that is never
    // materialised): X' = bf16(X * x_scale[c] + x_shift[c]) on real pixels, padding stays zero
    const float* x_scale; const float* x_shift; int x_ctrue;
    // optional (1x1 problems): column `bias_col` of X (>= c_lim, a column of ones written by im2col_1ch) lands at
    // out[bias_off + n] -- the bias gradient sum_m dY[m][n] rides in the same MFMAs; -1 = off
    int bias_col; long bias_off;
    // optional: row `bias_row` of the dY operand summed over all pixels (one more MFMA per sub-tile against a fragment
    // of ones) lands at out[bias_off] -- the scalar bias gradient of a one-channel transposed conv, whose dY operand
    // is the taps tensor of the image gradient (centre tap = the image gradient itself); -1 = off;
    // -2 = every row n < n_lim, at out[bias_off + n] (the bias gradient of a Dense / 1x1 layer)
    int bias_row;
};

// floor(x / d) for 0 <= x < 2^31 with d's magic pair (host: magic_div)
__device__ __forceinline__ int fastdiv(int x, unsigned mg, int sf) {
    return mg ? (int)(__umulhi((unsigned)x, mg) >> sf) : (x >> sf);
}
static void magic_div(int d, unsigned* mg, int* sf) {
    int s = 0;
    while ((2L << s) <= d) ++s;                 // 2^s <= d < 2^(s+1)
    *sf = s;
    *mg = ((1L << s) == d) ? 0u : (unsigned)(((1UL << (32 + s)) / (unsigned long)d) + 1UL);
}

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <int TN, int TC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int SN = TN / 16, SC = TC / 16;           // 16-channel sub-tiles
    // sub-tile strides: 1 KiB of data + a skew, so the 16-byte staging stores of one pixel row (its 2*S units go
    // to S different sub-tiles) and of the next rows land in different banks (unskewed, sub-tiles alias: 4-way
    // conflicts for 64 channels, measured 66 % of LDS cycles)
    constexpr int SYS = 1024 + (SN >= 4 ? 32 : (SN == 2 ? 64 : 0));
    constexpr int SXS = 1024 + (SC >= 4 ? 32 : (SC == 2 ? 64 : 0));
    constexpr int STAGE = SN * SYS + SC * SXS;          // bytes per wave: [sub][32 px][16 ch] bf16
    constexpr int RED = TN * TC * 4;                    // bytes per wave partial tile
    constexpr int PER_WAVE = STAGE > RED ? STAGE : RED;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* sY = smem + wave * PER_WAVE;
    char* sX = sY + SN * SYS;

    const int ks = blockIdx.x, tap = blockIdx.y;
    const int nt = blockIdx.z / a.tiles_c, ct = blockIdx.z - nt * a.tiles_c;
    const int n0 = nt * TN, c0 = ct * TC;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int offh = kh * a.dh - a.pt, offw = kw * a.dw - a.pl;
    const int m_begin = ks * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int OHW = a.OH * a.OW;

    f32x4 acc[SN][SC];
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SC; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[SN];                                     // row sums of the dY operand (bias_row)
#pragma unroll
    for (int i = 0; i < SN; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool rowsum = a.bias_row != -1 && blockIdx.y == 0 && ct == 0;
    const s16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    // staging assignment: 16-B units (8 channels of one pixel)
    constexpr int UY = TN / 8, UX = TC / 8;              // units per pixel row
    constexpr int RPY = 64 / UY, RPX = 64 / UX;          // rows per pass
    constexpr int PY = 32 / RPY, PX = 32 / RPX;          // passes
    const int yrow = lane / UY, ycol = lane % UY;
    const int xrow = lane / UX, xcol = lane % UX;
    const bool ycol_ok = n0 + ycol * 8 < a.Ny;
    const bool xcol_ok = c0 + xcol * 8 < a.Cx;

    // transposed-read addressing
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int tr_off = (4 * g + tq) * 32 + tp * 8;

    // Branch-free bounds-checked buffer loads (an out-of-range unit gets an offset past the descriptor and
    // reads as zero), software-pipelined one 32-pixel slice ahead: the gathers of slice s+1 are in flight
    // while slice s goes LDS -> transposed fragments -> MFMA.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.M * a.Ny * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.H * a.W * a.Cx * 2u), 0x00020000);
    const bool xaff = a.x_scale != nullptr;
    float xsc[8], xsh[8];
    if (xaff) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + xcol * 8 + j;
            const bool real = c < a.x_ctrue;
            xsc[j] = real ? a.x_scale[c] : 0.f;
            xsh[j] = real ? a.x_shift[c] : 0.f;
        }
    }
    auto gload = [&](int mb, u32x4 (&ry)[PY], u32x4 (&rx)[PX], unsigned& okx) {
        okx = 0;
#pragma unroll
        for (int p = 0; p < PY; ++p) {
            int m = mb + yrow + p * RPY;
            bool ok = ycol_ok & (m < m_end);
            unsigned off = ok ? (unsigned)(m * a.Ny + n0 + ycol * 8) * 2u : 0xFFFFFFF0u;
            ry[p] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            int m = mb + xrow + p * RPX;
            bool ok = xcol_ok & (m < m_end);
            int mm = ok ? m : 0;
            int b = fastdiv(mm, a.mg_ohw, a.sf_ohw), rem = mm - b * OHW;
            int oh = fastdiv(rem, a.mg_ow, a.sf_ow), ow = rem - oh * a.OW;
            int ih = oh * a.sh + offh, iw = ow * a.sw + offw;
            ok = ok & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            unsigned off = ok ? (unsigned)(((b * a.H + ih) * a.W + iw) * a.Cx + c0 + xcol * 8) * 2u : 0xFFFFFFF0u;
            rx[p] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
            okx |= (ok ? 1u : 0u) << p;
        }
    };
    auto consume = [&](const u32x4 (&ry)[PY], u32x4 (&rx)[PX], unsigned okx) {
        if (xaff) {
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                float f[8];
                unpack8(make_uint4(rx[p][0], rx[p][1], rx[p][2], rx[p][3]), f);
                const bool ok = (okx >> p) & 1u;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = ok ? fmaf(f[e], xsc[e], xsh[e]) : 0.f;
                const uint4 r = pack8(f);
                rx[p] = (u32x4){r.x, r.y, r.z, r.w};
            }
        }
#pragma unroll
        for (int p = 0; p < PY; ++p) {
            int row = yrow + p * RPY;
            *(u32x4*)(sY + (ycol >> 1) * SYS + row * 32 + (ycol & 1) * 16) = ry[p];
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            int row = xrow + p * RPX;
            *(u32x4*)(sX + (xcol >> 1) * SXS + row * 32 + (xcol & 1) * 16) = rx[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        bf16x8_t fy[SN], fx[SC];
#pragma unroll
        for (int i = 0; i < SN; ++i) {
            s16x4 lo = tr_read(sY + i * SYS + tr_off), hi = tr_read(sY + i * SYS + tr_off + 512);
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fy[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < SC; ++j) {
            s16x4 lo = tr_read(sX + j * SXS + tr_off), hi = tr_read(sX + j * SXS + tr_off + 512);
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fx[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < SN; ++i)
#pragma unroll
            for (int j = 0; j < SC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
        if (rowsum) {
#pragma unroll
            for (int i = 0; i < SN; ++i)
                accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], __builtin_bit_cast(bf16x8_t, ones8), accb[i],
                                                                  0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    {
        u32x4 ryA[PY], rxA[PX], ryB[PY], rxB[PX];
        unsigned okA, okB;
        const int mb0 = m_begin + wave * 32;
        gload(mb0, ryA, rxA, okA);
        for (int mb = mb0; mb < m_end; mb += 256) {       // two slices per trip; a slice past m_end is all zeros
            gload(mb + 128, ryB, rxB, okB);
            consume(ryA, rxA, okA);
            gload(mb + 256, ryA, rxA, okA);
            consume(ryB, rxB, okB);
        }
    }

    // ---- cross-wave reduction through LDS, then one slab write per block
    __syncthreads();
    float* red = (float*)(smem + wave * PER_WAVE);       // [TN][TC] row-major per wave
#pragma unroll
    for (int i = 0; i < SN; ++i)
#pragma unroll
        for (int j = 0; j < SC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                red[(16 * i + 4 * g + e) * TC + 16 * j + li] = acc[i][j][e];
    __syncthreads();
    float* out = a.slab + (long)ks * a.slab_stride;
    for (int idx = tid; idx < TN * TC; idx += 256) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += ((const float*)(smem + w * PER_WAVE))[idx];
        int n = n0 + idx / TC, c = c0 + idx % TC;
        if (n < a.n_lim) {
            if (c < a.c_lim) out[n * a.sn + tap * a.st + c * a.sc] = s;
            else if (c == a.bias_col) out[a.bias_off + n] = s;
        }
    }
    if (rowsum && (a.bias_row == -2 || (a.bias_row >= n0 && a.bias_row < n0 + TN))) {      // block-uniform
        // every column of the ones product holds the row sum: take column 0 (li == 0) of each wave, fold the 4 waves
        __syncthreads();
        float* rb = (float*)smem;                                   // [4 waves][TN]
        if (li == 0) {
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[wave * TN + 16 * i + 4 * g + e] = accb[i][e];
        }
        __syncthreads();
        if (a.bias_row == -2) {
            if (tid < TN && n0 + tid < a.n_lim)
                out[a.bias_off + n0 + tid] = ((rb[tid] + rb[TN + tid]) + rb[2 * TN + tid]) + rb[3 * TN + tid];
        } else if (tid == 0) {
            const int r = a.bias_row - n0;
            out[a.bias_off] = ((rb[r] + rb[TN + r]) + rb[2 * TN + r]) + rb[3 * TN + r];
        }
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long n,
                                   long slab_stride, int ksplit) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 7 < ksplit; k += 8) {              // eight independent loads per trip, summed in slab order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(k + u) * slab_stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < ksplit; ++k) s += slab[k * slab_stride + i];
        out[i] = s;
    }
}

// Many-slab variant (persistent wgrad2 workers: up to 256 slabs).  A block owns 32 float4 columns and
// splits the slabs over 8 lanes per column (independent, coalesced 512-byte rows), then combines the
// 8 lane sums through LDS in a fixed order -> deterministic, and ~8x more loads in flight than the
// one-thread-per-element walk above.
__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float4* __restrict__ slab, float4* __restrict__ out,
                                                               long n4, long stride4, int ksplit) {
    __shared__ float4 red[8][33];
    const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int k = kl;
        for (; k + 120 < ksplit; k += 128) {          // sixteen loads per trip where there are that many slabs (same order)
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slab[(long)(k + 8 * u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k + 24 < ksplit; k += 32) {            // four independent 16-byte loads per trip, fixed order
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = slab[(long)(k + 8 * u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < ksplit; k += 8) {
            float4 v = slab[(long)k * stride4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && i < n4) {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            float4 v = red[q][col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[i] = s;
    }
}

DDL_API void ddl_slab_reduce(const float* slab, float* out, long n, long stride, int k, hipStream_t s) {
    if (k >= 16 && n % 4 == 0 && stride % 4 == 0) {
        long n4 = n / 4;
        slab_reduce_wide_kernel<<<(int)((n4 + 31) / 32), 256, 0, s>>>((const float4*)slab, (float4*)out, n4,
                                                                     stride / 4, k);
        return;
    }
    long g = (n + 255) / 256;
    slab_reduce_kernel<<<(int)(g > 2048 ? 2048 : g), 256, 0, s>>>(slab, out, n, stride, k);
}

template <int TN, int TC>
static void launch_wgrad(const WgradArgs& a, int ksplit, int taps, int tiles, hipStream_t s) {
    constexpr int SN = TN / 16, SC = TC / 16;           // must mirror the kernel's skewed sub-tile strides
    constexpr int STAGE = SN * (1024 + (SN >= 4 ? 32 : (SN == 2 ? 64 : 0))) + SC * (1024 + (SC >= 4 ? 32 : (SC == 2 ? 64 : 0)));
    constexpr int RED = TN * TC * 4;
    constexpr int PER_WAVE = STAGE > RED ? STAGE : RED;
    wgrad_kernel<TN, TC><<<dim3(ksplit, taps, tiles), 256, 4 * PER_WAVE, s>>>(a);
}

DDL_API int ddl_wgrad_ksplit(int M, int taps, int n_lim, int c_lim) {
    int tn = n_lim > 32 ? 64 : (n_lim > 16 ? 32 : 16);
    int tc = c_lim > 32 ? 64 : (c_lim > 16 ? 32 : 16);
    long tiles = (long)((n_lim + tn - 1) / tn) * ((c_lim + tc - 1) / tc) * taps;
    static const long target = getenv("DDL_WGRAD_BLOCKS") ? atol(getenv("DDL_WGRAD_BLOCKS")) : 1024;
    long want = (target + tiles - 1) / tiles;
    // At least 1024 rows of the reduction per split (was 128).  The DCGAN generator's Dense layer (M = 1000 rows,
    // 196 tiles) ran as six splits of 167 rows -- 19 MB of slab traffic for a 3 MB gradient, plus the reduce launch;
    // as one split the headline step is 10-14 us shorter.  Measured 128 / 512 / 1024 / 2048 / 4096: dcgan28 1.134 /
    // 1.126 / 1.120 / 1.124 / 1.124 ms, classifier 3.287 / 3.274 / 3.273 / 3.297 / 3.307 ms, batch-32 ensemble
    // 25.85 / - / 25.83 / - / 26.9 ms.
    static const long minrows = getenv("DDL_WGRAD_MINROWS") ? atol(getenv("DDL_WGRAD_MINROWS")) : 1024;
    long maxk = (M + minrows - 1) / minrows;
    long k = want < 1 ? 1 : want;
    if (k > maxk) k = maxk;
    if (k > 256) k = 256;
    return (int)k;
}

// out: fp32 [numel] gradient view (contiguous); slab: fp32 scratch of ksplit*numel floats (unused
// when ksplit == 1).  Every element of `out` inside the (n_lim, T, c_lim) box is overwritten;
// the caller zero-fills `out` beforehand if the box does not cover it.
DDL_API int ddl_conv_wgrad(const void* dy, const void* x, float* out, float* slab, long numel, int ksplit, int B,
                           int OH, int OW, int Ny, int H, int W, int Cx, int KH, int KW, int sh, int sw, int dh,
                           int dw, int pt, int pl, int n_lim, int c_lim, long sn, long st, long sc,
                           const float* x_scale, const float* x_shift, int x_ctrue, int bias_col, long bias_off,
                           int bias_row, hipStream_t s) {
    if (Ny % 8 || Cx % 8 || ksplit < 1) return -1;
    if (bias_col >= 0 && (KH * KW != 1 || bias_col < c_lim || bias_col >= Cx || bias_off < 0 ||
                          bias_off + n_lim > numel || x_scale))
        return -4;
    if (bias_row >= 0 && (KH * KW != 1 || bias_col >= 0 || bias_row >= n_lim || bias_off < 0 || bias_off >= numel))
        return -4;
    if (bias_row < -1 && (bias_row != -2 || KH * KW != 1 || bias_col >= 0 || bias_off < 0 || bias_off + n_lim > numel))
        return -4;
    if (x_scale && !x_shift) return -1;
    long M = (long)B * OH * OW;
    if (M >= (1L << 31) || M * Ny >= (1L << 31) || (long)B * H * W * Cx >= (1L << 31)) return -3;
    WgradArgs a;
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x;
    a.slab = ksplit == 1 ? out : slab;
    a.x_scale = x_scale; a.x_shift = x_shift; a.x_ctrue = x_ctrue;
    a.bias_col = bias_col >= 0 ? bias_col : -1; a.bias_off = bias_off;
    a.bias_row = bias_row;
    a.B = B; a.OH = OH; a.OW = OW; a.Ny = Ny; a.H = H; a.W = W; a.Cx = Cx;
    a.KH = KH; a.KW = KW; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl;
    a.n_lim = n_lim; a.c_lim = c_lim; a.sn = sn; a.st = st; a.sc = sc;
    a.slab_stride = numel; a.M = (int)M;
    magic_div(OH * OW, &a.mg_ohw, &a.sf_ohw);
    magic_div(OW, &a.mg_ow, &a.sf_ow);
    long chunk = (M + ksplit - 1) / ksplit;
    a.chunk = (int)((chunk + 127) / 128 * 128);
    int tn = n_lim > 32 ? 64 : (n_lim > 16 ? 32 : 16);
    int tc = c_lim > 32 ? 64 : (c_lim > 16 ? 32 : 16);
    int tiles_n = (n_lim + tn - 1) / tn;
    a.tiles_c = (c_lim + tc - 1) / tc;
    long tiles = (long)tiles_n * a.tiles_c;
    if (tiles > 65535 || KH * KW > 65535) return -2;
    int taps = KH * KW;
#define WG_CASE(TN_, TC_) if (tn == TN_ && tc == TC_) launch_wgrad<TN_, TC_>(a, ksplit, taps, (int)tiles, s);
    WG_CASE(64, 64) WG_CASE(64, 32) WG_CASE(64, 16) WG_CASE(32, 64) WG_CASE(32, 32) WG_CASE(32, 16)
    WG_CASE(16, 64) WG_CASE(16, 32) WG_CASE(16, 16)
#undef WG_CASE
    if (ksplit > 1) ddl_slab_reduce(slab, out, numel, numel, ksplit, s);
    return 0;
}

// ------------------------------------------------------------------------------ depthwise (direct)
// Depthwise correlation has no channel reduction, so it is a bandwidth-bound stencil: one lane
// handles 8 channels (16 B) of one output pixel.  Weights are [KH][KW][C] fp32-from-bf16.
__global__ void dw_fprop_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                const float* __restrict__ bias, bf16_t* __restrict__ y, int B, int H, int W,
                                int C, int OH, int OW, int KH, int KW, int sh, int sw, int dh, int dw, int pt,
                                int pl, int c_true, int act, float alpha) {
    const int cg = C / 8;
    const long total = (long)B * OH * OW * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int gch = (int)(i % cg);
        long p = i / cg;
        int ow = (int)(p % OW);
        long q = p / OW;
        int oh = (int)(q % OH);
        int b = (int)(q / OH);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            int ih = oh * sh + kh * dh - pt;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < KW; ++kw) {
                int iw = ow * sw + kw * dw - pl;
                if ((unsigned)iw >= (unsigned)W) continue;
                float xv[8], wv[8];
                unpack8(*(const uint4*)(x + (((long)b * H + ih) * W + iw) * C + gch * 8), xv);
                unpack8(*(const uint4*)(w + ((long)kh * KW + kw) * C + gch * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = gch * 8 + j;
            acc[j] = acc[j] + ((bias && c < c_true) ? bias[c] : 0.f);
        }
        act_switch(act, [&](auto KA) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = gch * 8 + j < c_true ? act_apply_k<decltype(KA)::value>(acc[j], act, alpha) : 0.f;
        });
        *(uint4*)(y + i * 8) = pack8(acc);
    }
}

__global__ void dw_dgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w,
                                bf16_t* __restrict__ dx, int B, int H, int W, int C, int OH, int OW, int KH,
                                int KW, int sh, int sw, int dh, int dw, int pt, int pl) {
    const int cg = C / 8;
    const long total = (long)B * H * W * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int gch = (int)(i % cg);
        long p = i / cg;
        int iw = (int)(p % W);
        long q = p / W;
        int ih = (int)(q % H);
        int b = (int)(q / H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            int th = ih + pt - kh * dh;
            if (th < 0 || th % sh) continue;
            int oh = th / sh;
            if (oh >= OH) continue;
            for (int kw = 0; kw < KW; ++kw) {
                int tw = iw + pl - kw * dw;
                if (tw < 0 || tw % sw) continue;
                int ow = tw / sw;
                if (ow >= OW) continue;
                float gv[8], wv[8];
                unpack8(*(const uint4*)(dy + (((long)b * OH + oh) * OW + ow) * C + gch * 8), gv);
                unpack8(*(const uint4*)(w + ((long)kh * KW + kw) * C + gch * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(gv[j], wv[j], acc[j]);
            }
        }
        *(uint4*)(dx + i * 8) = pack8(acc);
    }
}

// dW[kh][kw][c]: block = (pixel chunk, tap); lanes sweep pixels x channel groups, reduce in LDS.
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                       float* __restrict__ part, int B, int H, int W, int C, int OH,
                                                       int OW, int KH, int KW, int sh, int sw, int dh, int dw,
                                                       int pt, int pl) {
    __shared__ float red[256][9];
    const int cg = C / 8;
    const int cgl = cg < 256 ? cg : 256;
    const int rpi = 256 / cgl;
    const int gch = blockIdx.z * 256 + threadIdx.x % cgl, rl = threadIdx.x / cgl;
    const int tap = blockIdx.y, kh = tap / KW, kw = tap - kh * KW;
    const long M = (long)B * OH * OW;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (rl < rpi && gch < cg) {
        for (long m = blockIdx.x * (long)rpi + rl; m < M; m += (long)gridDim.x * rpi) {
            int ow = (int)(m % OW);
            long q = m / OW;
            int oh = (int)(q % OH);
            int b = (int)(q / OH);
            int ih = oh * sh + kh * dh - pt, iw = ow * sw + kw * dw - pl;
            if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
            float gv[8], xv[8];
            unpack8(*(const uint4*)(dy + m * C + gch * 8), gv);
            unpack8(*(const uint4*)(x + (((long)b * H + ih) * W + iw) * C + gch * 8), xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(gv[j], xv[j], acc[j]);
        }
    }
    if (cgl <= 64 && (cgl & (cgl - 1)) == 0) {          // shuffle fold + four waves through LDS (see the fast kernel)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int off = cgl; off < 64; off <<= 1) acc[j] += __shfl_xor(acc[j], off, 64);
        float* fr = &red[0][0];
        if (lane < cgl) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fr[(wave * cgl + lane) * 8 + j] = acc[j];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < cgl * 8; t += 256) {
            const int gg = blockIdx.z * 256 + (t >> 3);
            if (gg >= cg) continue;
            float v = fr[t];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += fr[w * cgl * 8 + t];
            part[((long)blockIdx.x * KH * KW + tap) * C + gg * 8 + (t & 7)] = v;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (rl == 0 && gch < cg) {
        for (int r = 1; r < rpi; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[threadIdx.x][j] += red[threadIdx.x + r * cgl][j];
        float* o = part + ((long)blockIdx.x * KH * KW + tap) * C + gch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = red[threadIdx.x][j];
    }
}

// ---- fast paths: square kernels up to 3x3 (the whole DDL_Normal4 search space), index space < 2^31.
// Compile-time tap loops (all of a pixel's loads are issued together, out-of-image taps load a clamped
// address and are masked), 32-bit index arithmetic with magic-number division (the generic kernels above
// spend more on 64-bit div/mod than on memory), and a weight-gradient kernel that reads dY once per
// pixel for ALL taps instead of once per tap.
struct FastDiv { unsigned mg; int sf; };
static FastDiv make_fastdiv(int d) {
    FastDiv f;
    int s = 0;
    while ((2L << s) <= d) ++s;                 // 2^s <= d < 2^(s+1)
    f.sf = s;
    f.mg = ((1L << s) == d) ? 0u : (unsigned)(((1UL << (32 + s)) / (unsigned long)d) + 1UL);
    return f;
}
// floor(x / d), exact for 0 <= x < 2^31
__device__ __forceinline__ unsigned fdivu(unsigned x, FastDiv d) {
    return d.mg ? (__umulhi(x, d.mg) >> d.sf) : (x >> d.sf);
}

struct DwArgs {
    int B, H, W, C, OH, OW, sh, sw, dh, dw, pt, pl, c_true, act;
    float alpha;
    unsigned total;
    FastDiv dcg, dOW, dOH, dW_, dH_, dsh, dsw;
    FastDiv dsh_hc[3], dsw_wc[3];     // dgrad: dividers by the class-grid height / width of parity ph / pw (strides <= 3)
    // fprop / wgrad: the layer's input is x * in_scale[c] + in_shift[c] on real pixels and channels < in_ctrue -- a
    // BatchNorm in front of the layer whose output tensor is never written (null = plain x).  A depthwise tap sum is
    // linear per channel, so the affine is applied to the SUMS, not to every loaded pixel (a 3x3 stride-1 layer
    // would redo the BatchNorm nine times per pixel and turn ALU-bound):
    //   forward          sum_t w_t (s x_t + h) v_t  =  s * sum_t w_t x_t v_t  +  h * sum_t w_t v_t
    //   weight gradient  sum_m g_m (s x_mt + h) v_mt =  s * sum_m g_m x_mt v_mt + h * sum_m g_m v_mt
    // (v = tap inside the image: the zero padding applies to the BatchNorm's OUTPUT.)  fp32 throughout: the
    // BatchNorm output is not rounded to bf16 on the way, unlike the materialised path.
    const float* in_scale; const float* in_shift; int in_ctrue;
    int all_valid;                    // no tap of any output pixel falls outside the image
};
__device__ __forceinline__ void dw_load_affine(const DwArgs& a, int gch, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = gch * 8 + j;
        const bool live = c < a.in_ctrue;
        sc[j] = live ? a.in_scale[c] : 0.f;
        sh[j] = live ? a.in_shift[c] : 0.f;
    }
}

// AFF: 0 = plain x; 1 = affine, every tap inside the image (scale and shift * sum_t w_t of each channel group are
// computed once per block, in front of the loop); 2 = affine with per-pixel live-tap sums (padded layers, > 512
// channels).  Separate instantiations: the plain path keeps its 84 registers.
template <int K, int AFF>
__global__ __launch_bounds__(256, (AFF == 1 && K == 3) ? 5 : 1) void dw_fprop_fast_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                            const DwArgs a) {
    const int cg = a.C / 8;
    float sc[8], sf[8];
    // AFF 1: (scale, shift * sum_t w_t) of every channel group in an LDS table, read back at the end of a trip -- held
    // in registers across the loop they cost one wave of occupancy (101 vs 84 registers)
    __shared__ __attribute__((aligned(16))) float saff[AFF == 1 ? 64 * 16 : 4];
    if (AFF == 1) {
        if ((int)threadIdx.x < cg) {
            const int g0 = threadIdx.x;
            float ws[8];
            dw_load_affine(a, g0, sc, sf);
#pragma unroll
            for (int j = 0; j < 8; ++j) ws[j] = 0.f;
#pragma unroll
            for (int t = 0; t < K * K; ++t) {
                float wf[8];
                unpack8(*(const uint4*)(w + t * a.C + g0 * 8), wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) ws[j] += wf[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { saff[g0 * 16 + j] = sc[j]; saff[g0 * 16 + 8 + j] = sf[j] * ws[j]; }
        }
        __syncthreads();
    }
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < a.total; i += gridDim.x * 256u) {
        const unsigned p = fdivu(i, a.dcg);
        const int gch = (int)(i - p * cg);
        const unsigned q = fdivu(p, a.dOW);
        const int ow = (int)(p - q * a.OW);
        const unsigned b = fdivu(q, a.dOH);
        const int oh = (int)(q - b * a.OH);
        const bf16_t* xb = x + (size_t)b * a.H * a.W * a.C + gch * 8;
        uint4 xv[K * K], wv[K * K];
        unsigned vm = 0;
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int ih = oh * a.sh + kh * a.dh - a.pt, iw = ow * a.sw + kw * a.dw - a.pl;
                const bool v = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                const uint4 ld = *(const uint4*)(xb + (v ? (ih * a.W + iw) * a.C : 0));
                xv[kh * K + kw] = v ? ld : make_uint4(0, 0, 0, 0);
                wv[kh * K + kw] = *(const uint4*)(w + (kh * K + kw) * a.C + gch * 8);
                vm |= v ? (1u << (kh * K + kw)) : 0u;
            }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        float ws[8];
        if (AFF == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ws[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            float xf[8], wf[8];
            unpack8(xv[t], xf);
            unpack8(wv[t], wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xf[j], wf[j], acc[j]);
            if (AFF == 2) {
                const bool live = (vm >> t) & 1u;
#pragma unroll
                for (int j = 0; j < 8; ++j) ws[j] += live ? wf[j] : 0.f;
            }
        }
        if (AFF == 1) {
            const float4* r = (const float4*)&saff[gch * 16];
            const float4 s0 = r[0], s1 = r[1], h0 = r[2], h1 = r[3];
            const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(sv[j], acc[j], hv[j]);
        } else if (AFF == 2) {
            dw_load_affine(a, gch, sc, sf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(sc[j], acc[j], sf[j] * ws[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (bias && gch * 8 + j < a.c_true) ? bias[gch * 8 + j] : 0.f;
        // one activation dispatch per pixel (8 channels).  Wrapping the whole loop, or compiling the activation in as
        // a template parameter, both cost registers -- spills under the five-wave bound: 4.63 -> 4.74 / 5.15 ms per
        // classifier step
        act_switch(a.act, [&](auto KA) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = gch * 8 + j < a.c_true ? act_apply_k<decltype(KA)::value>(acc[j], a.act, a.alpha) : 0.f;
        });
        *(uint4*)(y + (size_t)i * 8) = pack8(acc);
    }
}

// Input gradient by output-parity class (blockIdx.y = ph * sw + pw): dx pixels (ph + sh*i, pw + sw*j) of one
// class are reached by the SAME taps -- those with (ph + pt - kh*dh) % sh == 0 -- so the tap test is uniform
// per block and the other taps cost nothing (a per-pixel test made a stride-2 3x3 layer issue nine masked
// loads and 72 FMAs for 2.25 live taps on average: ALU-bound at 300 us for a 426 MB tensor).
template <int K>
__global__ __launch_bounds__(256) void dw_dgrad_fast_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w,
                                                            bf16_t* __restrict__ dx, const DwArgs a) {
    const int cg = a.C / 8;
    const int ph = blockIdx.y / a.sw, pw = blockIdx.y - ph * a.sw;
    const int Hc = (a.H - ph + a.sh - 1) / a.sh, Wc = (a.W - pw + a.sw - 1) / a.sw;     // class grid
    if (Hc <= 0 || Wc <= 0) return;
    // live taps of this class and their dY offsets: th = ph + pt - kh*dh + sh*i  ->  oh = oh0[kh] + i
    int oh0[K], ow0[K];
    bool lh[K], lw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int th0 = ph + a.pt - k * a.dh, tw0 = pw + a.pl - k * a.dw;
        const int qh = th0 >= 0 ? th0 / a.sh : -((-th0 + a.sh - 1) / a.sh);            // floor division
        const int qw = tw0 >= 0 ? tw0 / a.sw : -((-tw0 + a.sw - 1) / a.sw);
        lh[k] = qh * a.sh == th0;
        lw[k] = qw * a.sw == tw0;
        oh0[k] = qh;
        ow0[k] = qw;
    }
    const FastDiv dWc = a.dsw_wc[pw], dHc = a.dsh_hc[ph];
    const unsigned total = (unsigned)a.B * Hc * Wc * cg;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = fdivu(i, a.dcg);
        const int gch = (int)(i - p * cg);
        const unsigned q = fdivu(p, dWc);
        const int jc = (int)(p - q * Wc);
        const unsigned b = fdivu(q, dHc);
        const int ic = (int)(q - b * Hc);
        const bf16_t* gb = dy + (size_t)b * a.OH * a.OW * a.C + gch * 8;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            if (!lh[kh]) continue;                                   // block-uniform
            const int oh = oh0[kh] + ic;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                if (!lw[kw]) continue;                               // block-uniform
                const int ow = ow0[kw] + jc;
                const bool v = ((unsigned)oh < (unsigned)a.OH) & ((unsigned)ow < (unsigned)a.OW);
                const uint4 ld = *(const uint4*)(gb + (v ? (oh * a.OW + ow) * a.C : 0));
                const uint4 g4 = v ? ld : make_uint4(0, 0, 0, 0);
                float gf[8], wf[8];
                unpack8(g4, gf);
                unpack8(*(const uint4*)(w + (kh * K + kw) * a.C + gch * 8), wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(gf[j], wf[j], acc[j]);
            }
        }
        const int ih = ph + a.sh * ic, iw = pw + a.sw * jc;
        *(uint4*)(dx + ((size_t)(b * a.H + ih) * a.W + iw) * a.C + gch * 8) = pack8(acc);
    }
}

// ---- depthwise input gradient fused with the backward of the BatchNorm IN FRONT of the layer -----------------
// The classifier replicas (DDL_Normal4.random_layer) follow every conv-like layer by a BatchNorm, and a BatchNorm in
// front of a depthwise layer is applied on load, so in the backward pass the depthwise input gradient g is exactly
// that BatchNorm's dY.  Unfused that is three passes over the LARGE tensor (57 x 57 x 128 x batch in model0): write g,
// read (g, x) for the column sums, read (g, x) again and write dx.  g is cheap to recompute from the small dY
// (2.25 live taps per pixel for a 3x3 stride-2 layer), so it is never stored:
//   REDUCE pass: g from dY on the fly, x loaded -> the BatchNorm's column sums (colreduce MODE 1 / MODE 3 semantics,
//                partials part[NS][C][blocks], finalized by the ordinary bn_bwd finalize kernel);
//   APPLY  pass: g recomputed, dx = prev_act'(x) * (k0 * g + k1 * x + k2)  (bn_bwd_apply_cs semantics) stored.
// g is rounded to bf16 before use, as the materialised tensor was.  Traffic: 2 reads of x + 1 write of dx instead of
// 1 write of g + 2 reads of g + 2 reads of x + 1 write of dx.
struct DwBnArgs {
    const bf16_t* x;               // the BatchNorm's input (same shape as dx)
    const float *mean, *invstd, *gamma, *c1, *c2;
    float* part;                   // [NS][C][gridDim.x * gridDim.y]
    int prev_act; float prev_alpha;
};

template <int K, int NS, int P>
__global__ __launch_bounds__(256) void dw_dgrad_bn_reduce_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w,
                                                                 const DwArgs a, const DwBnArgs n) {
    __shared__ float red[4 * 64 * NS * 8];
    const int cg = a.C / 8;                    // a power of two <= 64 (host-checked): a thread's channel group is fixed
    const int ph = blockIdx.y / a.sw, pw = blockIdx.y - ph * a.sw;
    const int Hc = (a.H - ph + a.sh - 1) / a.sh, Wc = (a.W - pw + a.sw - 1) / a.sw;
    float s[NS][8];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
    const int gch = threadIdx.x & (cg - 1);
    if (Hc > 0 && Wc > 0) {
        int oh0[K], ow0[K];
        bool lh[K], lw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int th0 = ph + a.pt - k * a.dh, tw0 = pw + a.pl - k * a.dw;
            const int qh = th0 >= 0 ? th0 / a.sh : -((-th0 + a.sh - 1) / a.sh);
            const int qw = tw0 >= 0 ? tw0 / a.sw : -((-tw0 + a.sw - 1) / a.sw);
            lh[k] = qh * a.sh == th0;
            lw[k] = qw * a.sw == tw0;
            oh0[k] = qh;
            ow0[k] = qw;
        }
        float mu[8], is[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { mu[j] = n.mean[gch * 8 + j]; is[j] = n.invstd[gch * 8 + j]; }
        const FastDiv dWc = a.dsw_wc[pw], dHc = a.dsh_hc[ph];
        const unsigned total = (unsigned)a.B * Hc * Wc * cg;
        // x is the HBM stream of this pass (dY is small and cache-resident): the load of the NEXT pixel's x is issued
        // before this pixel's taps are gathered, unconditionally (clamped index), so the compiler can count it
        auto x_at = [&](unsigned i) {
            const unsigned p = fdivu(i, a.dcg);
            const unsigned q = fdivu(p, dWc);
            const int jc = (int)(p - q * Wc);
            const unsigned b = fdivu(q, dHc);
            const int ic = (int)(q - b * Hc);
            return *(const uint4*)(n.x + ((size_t)(b * a.H + ph + a.sh * ic) * a.W + pw + a.sw * jc) * a.C + gch * 8);
        };
        const unsigned stride = gridDim.x * 256u;
        unsigned i = blockIdx.x * 256u + threadIdx.x;
        uint4 xn = x_at(i < total ? i : total - 1);
        for (; i < total; i += stride) {
            const uint4 x4 = xn;
            xn = x_at(i + stride < total ? i + stride : total - 1);
            const unsigned p = fdivu(i, a.dcg);
            const unsigned q = fdivu(p, dWc);
            const int jc = (int)(p - q * Wc);
            const unsigned b = fdivu(q, dHc);
            const int ic = (int)(q - b * Hc);
            const bf16_t* gb = dy + (size_t)b * a.OH * a.OW * a.C + gch * 8;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                if (!lh[kh]) continue;
                const int oh = oh0[kh] + ic;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    if (!lw[kw]) continue;
                    const int ow = ow0[kw] + jc;
                    const bool v = ((unsigned)oh < (unsigned)a.OH) & ((unsigned)ow < (unsigned)a.OW);
                    const uint4 ld = *(const uint4*)(gb + (v ? (oh * a.OW + ow) * a.C : 0));
                    const uint4 g4 = v ? ld : make_uint4(0, 0, 0, 0);
                    float gf[8], wf[8];
                    unpack8(g4, gf);
                    unpack8(*(const uint4*)(w + (kh * K + kw) * a.C + gch * 8), wf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(gf[j], wf[j], acc[j]);
                }
            }
            float gv[8], xv[8];
            unpack8(pack8(acc), gv);                    // the bf16 rounding of the materialised g
            unpack8(x4, xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[0][j] += gv[j]; s[1][j] += gv[j] * (xv[j] - mu[j]) * is[j]; }
            if (NS == 5) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ad = act_grad_y_k<P>(xv[j], n.prev_act, n.prev_alpha);
                    s[2][j] += ad * gv[j];
                    s[3][j] += ad * xv[j];
                    s[4][j] += ad;
                }
            }
        }
    }
    // lanes l, l + cg, ... of a wave hold different pixels of one channel group: shuffle fold, four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = cg; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[k][j] += __shfl_xor(s[k][j], off, 64);
    }
    if (lane < cg) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(wave * cg + lane) * (NS * 8) + k * 8 + j] = s[k][j];
    }
    __syncthreads();
    const size_t nb = (size_t)gridDim.x * gridDim.y;
    const size_t bi = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int t = threadIdx.x; t < cg * NS * 8; t += 256) {
        const int gl = t / (NS * 8), e = t - gl * (NS * 8);
        float v = red[t];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) v += red[wv * cg * (NS * 8) + t];
        const int k = e >> 3, j = e & 7;
        n.part[((size_t)k * a.C + gl * 8 + j) * nb + bi] = v;
    }
}

template <int K, int P>
__global__ __launch_bounds__(256) void dw_dgrad_bn_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w,
                                                                bf16_t* __restrict__ dx, const DwArgs a,
                                                                const DwBnArgs n) {
    const int cg = a.C / 8;
    const int ph = blockIdx.y / a.sw, pw = blockIdx.y - ph * a.sw;
    const int Hc = (a.H - ph + a.sh - 1) / a.sh, Wc = (a.W - pw + a.sw - 1) / a.sw;
    if (Hc <= 0 || Wc <= 0) return;
    int oh0[K], ow0[K];
    bool lh[K], lw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int th0 = ph + a.pt - k * a.dh, tw0 = pw + a.pl - k * a.dw;
        const int qh = th0 >= 0 ? th0 / a.sh : -((-th0 + a.sh - 1) / a.sh);
        const int qw = tw0 >= 0 ? tw0 / a.sw : -((-tw0 + a.sw - 1) / a.sw);
        lh[k] = qh * a.sh == th0;
        lw[k] = qw * a.sw == tw0;
        oh0[k] = qh;
        ow0[k] = qw;
    }
    const int gch = threadIdx.x & (cg - 1);
    // dx = prev_act'(x) * (k0*g + k1*x + k2),  k0 = gamma*is, k1 = -gamma*is*is*c2, k2 = -gamma*is*(c1 - mean*is*c2)
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = gch * 8 + j;
        const float is = n.invstd[c], gi = n.gamma[c] * is;
        k0[j] = gi;
        k1[j] = -gi * is * n.c2[c];
        k2[j] = -gi * (n.c1[c] - n.mean[c] * is * n.c2[c]);
    }
    const FastDiv dWc = a.dsw_wc[pw], dHc = a.dsh_hc[ph];
    const unsigned total = (unsigned)a.B * Hc * Wc * cg;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = fdivu(i, a.dcg);
        const unsigned q = fdivu(p, dWc);
        const int jc = (int)(p - q * Wc);
        const unsigned b = fdivu(q, dHc);
        const int ic = (int)(q - b * Hc);
        const bf16_t* gb = dy + (size_t)b * a.OH * a.OW * a.C + gch * 8;
        const int ih = ph + a.sh * ic, iw = pw + a.sw * jc;
        const size_t xo = ((size_t)(b * a.H + ih) * a.W + iw) * a.C + gch * 8;
        const uint4 x4 = *(const uint4*)(n.x + xo);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            if (!lh[kh]) continue;
            const int oh = oh0[kh] + ic;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                if (!lw[kw]) continue;
                const int ow = ow0[kw] + jc;
                const bool v = ((unsigned)oh < (unsigned)a.OH) & ((unsigned)ow < (unsigned)a.OW);
                const uint4 ld = *(const uint4*)(gb + (v ? (oh * a.OW + ow) * a.C : 0));
                const uint4 g4 = v ? ld : make_uint4(0, 0, 0, 0);
                float gf[8], wf[8];
                unpack8(g4, gf);
                unpack8(*(const uint4*)(w + (kh * K + kw) * a.C + gch * 8), wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(gf[j], wf[j], acc[j]);
            }
        }
        float gv[8], xv[8];
        unpack8(pack8(acc), gv);
        unpack8(x4, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = fmaf(k0[j], gv[j], fmaf(k1[j], xv[j], k2[j]));
            if (P != ACT_LINEAR) d *= act_grad_y_k<P>(xv[j], n.prev_act, n.prev_alpha);
            gv[j] = d;
        }
        *(uint4*)(dx + xo) = pack8(gv);
    }
}

// phase 0: REDUCE (writes n.part, returns the number of partial blocks per channel), phase 1: APPLY (writes dx).
// `ns`: 2 (dgamma / dbeta sums) or 5 (plus the bias gradient of the conv in front of the BatchNorm).
// Returns < 0 when the shape is outside the fused kernels' range (the caller runs the three-pass form).
DDL_API int ddl_dw_dgrad_bn(int phase, const void* dy, const void* w, const void* x, void* dx, float* part, int B,
                            int H, int W, int C, int OH, int OW, int K, int sh, int sw, int dh, int dw, int pt, int pl,
                            const float* mean, const float* invstd, const float* gamma, const float* c1,
                            const float* c2, int prev_act, float prev_alpha, int ns, hipStream_t s) {
    const int cg = C / 8;
    if (C % 8 || cg > 64 || (cg & (cg - 1)) || K < 1 || K > 3 || sh > 3 || sw > 3 || (ns != 2 && ns != 5)) return -1;
    const long in_elems = (long)B * H * W * C, out_elems = (long)B * OH * OW * C;
    if (in_elems >= (1L << 31) || out_elems >= (1L << 31)) return -1;
    DwArgs d;
    d.B = B; d.H = H; d.W = W; d.C = C; d.OH = OH; d.OW = OW; d.sh = sh; d.sw = sw; d.dh = dh; d.dw = dw;
    d.pt = pt; d.pl = pl; d.c_true = C; d.act = 0; d.alpha = 0.f;
    d.in_scale = nullptr; d.in_shift = nullptr; d.in_ctrue = 0; d.all_valid = 0;
    d.total = (unsigned)((long)B * H * W * cg);
    d.dcg = make_fastdiv(cg); d.dOW = make_fastdiv(OW); d.dOH = make_fastdiv(OH);
    d.dW_ = make_fastdiv(W); d.dH_ = make_fastdiv(H); d.dsh = make_fastdiv(sh); d.dsw = make_fastdiv(sw);
    for (int p_ = 0; p_ < 3; ++p_) {
        const int hc = p_ < sh ? (H - p_ + sh - 1) / sh : 1, wc = p_ < sw ? (W - p_ + sw - 1) / sw : 1;
        d.dsh_hc[p_] = make_fastdiv(hc > 0 ? hc : 1);
        d.dsw_wc[p_] = make_fastdiv(wc > 0 ? wc : 1);
    }
    DwBnArgs n;
    n.x = (const bf16_t*)x; n.mean = mean; n.invstd = invstd; n.gamma = gamma; n.c1 = c1; n.c2 = c2; n.part = part;
    n.prev_act = prev_act; n.prev_alpha = prev_alpha;
    const int classes = sh * sw;
    unsigned g = (d.total + 255u) / 256u;
    // REDUCE: few, long blocks (every block leaves NS * C partials); APPLY: the plain dgrad's grid
    unsigned gx = phase == 0 ? (2048u + classes - 1) / classes : ((g > 8192u ? 8192u : g) + classes - 1) / classes;
    if (phase == 0 && gx * 256u * classes > d.total) gx = (d.total / (256u * classes)) ? d.total / (256u * classes) : 1u;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, classes);
#define DWBN_K(KK)                                                                                                    \
    do {                                                                                                              \
        if (phase == 0) {                                                                                             \
            if (ns == 5) {                                                                                            \
                ACT1_DISPATCH(prev_act, DWBN_R5_##KK);                                                                \
            } else                                                                                                    \
                dw_dgrad_bn_reduce_kernel<KK, 2, 0><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, d, n);  \
        } else {                                                                                                      \
            ACT1_DISPATCH(prev_act, DWBN_A_##KK);                                                                     \
        }                                                                                                             \
    } while (0)
#define DWBN_R5_1(A) dw_dgrad_bn_reduce_kernel<1, 5, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, d, n)
#define DWBN_R5_2(A) dw_dgrad_bn_reduce_kernel<2, 5, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, d, n)
#define DWBN_R5_3(A) dw_dgrad_bn_reduce_kernel<3, 5, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, d, n)
#define DWBN_A_1(A) dw_dgrad_bn_apply_kernel<1, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx, d, n)
#define DWBN_A_2(A) dw_dgrad_bn_apply_kernel<2, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx, d, n)
#define DWBN_A_3(A) dw_dgrad_bn_apply_kernel<3, A><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx, d, n)
    if (K == 1) DWBN_K(1);
    else if (K == 2) DWBN_K(2);
    else DWBN_K(3);
#undef DWBN_K
    return phase == 0 ? (int)(gx * classes) : 0;
}

// dW[t][c] partials: block = (pixel chunk, kernel ROW): the K taps of one kernel row share the dY load and
// their X pixels are neighbours.  One block per tap (nine dY reads) and one block for all nine taps (one
// dY read, 72 accumulators: 771 vs 544 us per classifier step) were both measured slower: K x the
// blocks with a small register footprint hide the gather latency, K x fewer dY reads cut the traffic.
template <int K, bool AFF>
__global__ __launch_bounds__(256) void dw_wgrad_fast_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            float* __restrict__ part, const DwArgs a) {
    __shared__ float red[256][9];
    const int cg = a.C / 8;
    const int cgl = cg < 256 ? cg : 256;
    const int rpi = 256 / cgl;
    const int gch = blockIdx.z * 256 + threadIdx.x % cgl, rl = threadIdx.x / cgl;
    const int kh = blockIdx.y;
    const int offh = kh * a.dh - a.pt;
    const unsigned M = a.total;                  // B * OH * OW
    float acc[K][8], gs[AFF ? K : 1][8];        // gs: sums of dY over the pixels whose tap is live (on-load affine only)
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[t][j] = 0.f; if (AFF) gs[t][j] = 0.f; }
    constexpr bool aff = AFF;
    if (rl < rpi && gch < cg) {
        const unsigned stride = gridDim.x * (unsigned)rpi;
        auto one = [&](unsigned m, uint4& g4, uint4 (&x4)[K], unsigned& vm) {
            vm = 0;
            const unsigned q = fdivu(m, a.dOW);
            const int ow = (int)(m - q * a.OW);
            const unsigned b = fdivu(q, a.dOH);
            const int oh = (int)(q - b * a.OH);
            const int ih = oh * a.sh + offh;
            const bool vh = (unsigned)ih < (unsigned)a.H;
            const bf16_t* xb = x + (size_t)b * a.H * a.W * a.C + gch * 8;
            g4 = *(const uint4*)(dy + (size_t)m * a.C + gch * 8);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int iw = ow * a.sw + kw * a.dw - a.pl;
                const bool v = vh & ((unsigned)iw < (unsigned)a.W);
                const uint4 ld = *(const uint4*)(xb + (v ? (ih * a.W + iw) * a.C : 0));
                x4[kw] = v ? ld : make_uint4(0, 0, 0, 0);
                vm |= v ? (1u << kw) : 0u;
            }
        };
        auto fma = [&](const uint4& g4, const uint4 (&x4)[K], unsigned vm) {
            float gf[8];
            unpack8(g4, gf);
            if (aff && a.all_valid) {           // every tap live: one dY sum serves all taps
#pragma unroll
                for (int j = 0; j < 8; ++j) gs[0][j] += gf[j];
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                float xf[8];
                unpack8(x4[kw], xf);
                if (aff && !a.all_valid) {
                    const bool live = (vm >> kw) & 1u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) gs[kw][j] += live ? gf[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[kw][j] = fmaf(gf[j], xf[j], acc[kw][j]);
            }
        };
        unsigned m = blockIdx.x * (unsigned)rpi + rl;
        for (; m + stride < M; m += 2 * stride) {
            uint4 gA, gB, xA[K], xB[K];
            unsigned vA, vB;
            one(m, gA, xA, vA);
            one(m + stride, gB, xB, vB);
            fma(gA, xA, vA);
            fma(gB, xB, vB);
        }
        for (; m < M; m += stride) {
            uint4 gA, xA[K];
            unsigned vA;
            one(m, gA, xA, vA);
            fma(gA, xA, vA);
        }
        if (aff) {                      // linear in the partial sums: applied per block, before the slab reduce
            float sc[8], sf[8];
            dw_load_affine(a, gch, sc, sf);
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[kw][j] = fmaf(sc[j], acc[kw][j], sf[j] * (a.all_valid ? gs[0][j] : gs[kw][j]));
        }
    }
    if (cgl <= 64 && (cgl & (cgl - 1)) == 0) {
        // Lanes l, l + cgl, ... of a wave hold different pixels of one channel group: fold them with shuffles, then the
        // four waves through LDS, one (group, channel) element per thread.  (The serial form below let `cgl` threads
        // walk 256 / cgl rows of LDS each: with 16 channels that was TWO threads doing 3000 dependent LDS updates --
        // 100 us of a 107 us kernel on an 18x18x16 layer whose forward takes 15 us.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                for (int off = cgl; off < 64; off <<= 1) acc[kw][j] += __shfl_xor(acc[kw][j], off, 64);
        float* fr = &red[0][0];                               // [4 waves][cgl][8] per kernel column
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            __syncthreads();
            if (lane < cgl) {
#pragma unroll
                for (int j = 0; j < 8; ++j) fr[(wave * cgl + lane) * 8 + j] = acc[kw][j];
            }
            __syncthreads();
            for (int t = threadIdx.x; t < cgl * 8; t += 256) {
                const int gl = t >> 3, gg = blockIdx.z * 256 + gl;
                if (gg >= cg) continue;
                float v = fr[t];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += fr[w * cgl * 8 + t];
                part[((size_t)blockIdx.x * K * K + kh * K + kw) * a.C + gg * 8 + (t & 7)] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[kw][j];
        __syncthreads();
        if (rl == 0 && gch < cg) {
            for (int r = 1; r < rpi; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[threadIdx.x][j] += red[threadIdx.x + r * cgl][j];
            float* o = part + ((size_t)blockIdx.x * K * K + kh * K + kw) * a.C + gch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = red[threadIdx.x][j];
        }
    }
}

DDL_API int ddl_dw_wgrad_blocks(long M, int C) {
    int cg = C / 8, cgl = cg < 256 ? cg : 256, rpi = 256 / cgl;
    long nb = (M + (long)rpi * 8 - 1) / ((long)rpi * 8);       // >= 8 pixels per thread; K row-blocks per pixel chunk
    return (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
}

template <int K>
static void dw_fast_launch(int mode, const void* a_, const void* w, const float* bias, void* out, float* part,
                           const DwArgs& a, int nb, hipStream_t s) {
    const int cg = a.C / 8;
    if (mode == 0 || mode == 1) {
        unsigned g = (a.total + 255u) / 256u;
        if (g > 8192u) g = 8192u;
        if (mode == 0) {
            if (!a.in_scale) {
                dw_fprop_fast_kernel<K, 0><<<g, 256, 0, s>>>((const bf16_t*)a_, (const bf16_t*)w, bias, (bf16_t*)out, a);
            } else {
                if (a.all_valid && cg <= 64)
                    dw_fprop_fast_kernel<K, 1><<<g, 256, 0, s>>>((const bf16_t*)a_, (const bf16_t*)w, bias, (bf16_t*)out, a);
                else
                    dw_fprop_fast_kernel<K, 2><<<g, 256, 0, s>>>((const bf16_t*)a_, (const bf16_t*)w, bias, (bf16_t*)out, a);
            }
        } else {
            const int classes = a.sh * a.sw;
            unsigned gx = (g + classes - 1) / classes;
            if (gx < 1) gx = 1;
            dw_dgrad_fast_kernel<K><<<dim3(gx, classes), 256, 0, s>>>((const bf16_t*)a_, (const bf16_t*)w,
                                                                      (bf16_t*)out, a);
        }
    } else {
        if (a.in_scale)
            dw_wgrad_fast_kernel<K, true><<<dim3(nb, K, (cg + 255) / 256), 256, 0, s>>>((const bf16_t*)a_,
                                                                                        (const bf16_t*)w, part, a);
        else
            dw_wgrad_fast_kernel<K, false><<<dim3(nb, K, (cg + 255) / 256), 256, 0, s>>>((const bf16_t*)a_,
                                                                                         (const bf16_t*)w, part, a);
    }
}

extern "C" int ddl_dw_fprop_ring(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C,
                                 int OH, int OW, int K, int sh, int sw, int c_true, int act, float alpha,
                                 const float* in_scale, const float* in_shift, int in_ctrue, float* stats, hipStream_t s);

// Forward with the BatchNorm statistics partials of the output (part[2][C][B], one column per image): only the ring
// kernel has that epilogue.  Returns 0 on success, 1 when the layer is outside its range (nothing launched: the caller
// runs the plain forward and a reduction pass).
DDL_API int ddl_depthwise_fwd_stats(const void* x, const void* w, const float* bias, void* y, float* stats, int B, int H,
                                    int W, int C, int OH, int OW, int K, int sh, int sw, int c_true, int act, float alpha,
                                    const float* in_scale, const float* in_shift, int in_ctrue, hipStream_t s) {
    if (!stats) return 1;
    return ddl_dw_fprop_ring(x, w, bias, y, B, H, W, C, OH, OW, K, sh, sw, c_true, act, alpha, in_scale, in_shift,
                             in_ctrue, stats, s);
}

DDL_API int ddl_depthwise(int mode, const void* a, const void* w, const float* bias, void* out, float* part,
                          int B, int H, int W, int C, int OH, int OW, int KH, int KW, int sh, int sw, int dh,
                          int dw, int pt, int pl, int c_true, int act, float alpha, const float* in_scale,
                          const float* in_shift, int in_ctrue, hipStream_t s) {
    if (C % 8) return -1;
    if (in_scale && mode == 1) return -1;
    const int cg = C / 8;
    const long in_elems = (long)B * H * W * C, out_elems = (long)B * OH * OW * C;
    // forward of an unpadded, undilated layer: the LDS-ring kernel (dw_ring.hip) reads every input byte once
    if (mode == 0 && KH == KW && dh == 1 && dw == 1 && pt == 0 && pl == 0 &&
        ddl_dw_fprop_ring(a, w, bias, out, B, H, W, C, OH, OW, KH, sh, sw, c_true, act, alpha, in_scale, in_shift,
                          in_ctrue, nullptr, s) == 0)
        return 0;
    if (KH == KW && KH <= 3 && in_elems < (1L << 31) && out_elems < (1L << 31) && sh <= 3 && sw <= 3) {
        DwArgs d;
        d.B = B; d.H = H; d.W = W; d.C = C; d.OH = OH; d.OW = OW; d.sh = sh; d.sw = sw; d.dh = dh; d.dw = dw;
        d.pt = pt; d.pl = pl; d.c_true = c_true; d.act = act; d.alpha = alpha;
        d.in_scale = in_scale; d.in_shift = in_shift; d.in_ctrue = in_ctrue;
        d.all_valid = pt == 0 && pl == 0 && (OH - 1) * sh + (KH - 1) * dh < H && (OW - 1) * sw + (KW - 1) * dw < W;
        d.total = (unsigned)(mode == 0 ? (long)B * OH * OW * cg : (mode == 1 ? (long)B * H * W * cg : (long)B * OH * OW));
        d.dcg = make_fastdiv(cg); d.dOW = make_fastdiv(OW); d.dOH = make_fastdiv(OH);
        d.dW_ = make_fastdiv(W); d.dH_ = make_fastdiv(H); d.dsh = make_fastdiv(sh); d.dsw = make_fastdiv(sw);
        for (int p_ = 0; p_ < 3; ++p_) {
            const int hc = p_ < sh ? (H - p_ + sh - 1) / sh : 1, wc = p_ < sw ? (W - p_ + sw - 1) / sw : 1;
            d.dsh_hc[p_] = make_fastdiv(hc > 0 ? hc : 1);
            d.dsw_wc[p_] = make_fastdiv(wc > 0 ? wc : 1);
        }
        const int nb = mode == 2 ? ddl_dw_wgrad_blocks((long)B * OH * OW, C) : 0;
        if (KH == 1) dw_fast_launch<1>(mode, a, w, bias, out, part, d, nb, s);
        else if (KH == 2) dw_fast_launch<2>(mode, a, w, bias, out, part, d, nb, s);
        else dw_fast_launch<3>(mode, a, w, bias, out, part, d, nb, s);
        if (mode == 2) ddl_slab_reduce(part, (float*)out, (long)KH * KW * C, (long)KH * KW * C, nb, s);
        return 0;
    }
    if (in_scale) return -5;  // the generic kernels have no on-load affine: the caller materialises
    if (mode == 0) {          // fprop: a = x, out = y
        long total = (long)B * OH * OW * cg;
        long g = (total + 255) / 256;
        dw_fprop_kernel<<<(int)(g > 4096 ? 4096 : g), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)w, bias,
                                                                    (bf16_t*)out, B, H, W, C, OH, OW, KH, KW, sh,
                                                                    sw, dh, dw, pt, pl, c_true, act, alpha);
    } else if (mode == 1) {   // dgrad: a = dy, out = dx
        long total = (long)B * H * W * cg;
        long g = (total + 255) / 256;
        dw_dgrad_kernel<<<(int)(g > 4096 ? 4096 : g), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)w,
                                                                    (bf16_t*)out, B, H, W, C, OH, OW, KH, KW, sh,
                                                                    sw, dh, dw, pt, pl);
    } else {                  // wgrad: a = dy, w = x, out = dW fp32 [KH][KW][C], part = scratch
        int nb = ddl_dw_wgrad_blocks((long)B * OH * OW, C);
        dw_wgrad_kernel<<<dim3(nb, KH * KW, (cg + 255) / 256), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)w,
                                                                             part, B, H, W, C, OH, OW, KH, KW,
                                                                             sh, sw, dh, dw, pt, pl);
        long n = (long)KH * KW * C;
        ddl_slab_reduce(part, (float*)out, n, n, nb, s);
    }
    return 0;
}
