// Skinny-K row-major GEMM for 1x1 convolutions (gfx950, v_mfma_f32_16x16x32_bf16):
//
//     Y[m][n] = act(bias[n] + sum_k X[m][k] * W[n][k])      X [M][K], W [Nw][K], Y [M][Nd], bf16
//
// with K <= 128; Nd <= 128 in one tile, wider outputs (the generator's Dense 128 -> 6272) as 128-wide
// column tiles over blockIdx.y (X is re-read per tile from cache): the taps-as-channels form of the single-channel convolutions, 1x1
// layers of the random search space, and their data gradients.  With so little K the kernel is a
// pure streaming problem (read X once, write Y once), so there is no LDS staging of the pixel
// operand at all: a lane's MFMA fragment -- 8 consecutive k of one row -- is exactly one 16-byte
// global load of the row-major X.  W (<= 32 KiB) is staged into LDS once per block.  A wave owns 64
// rows; results leave through the same LDS-transposed 16-byte-store epilogue as pconv.
#include "common.h"
#include <cstdlib>

struct GemmSkArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    long M;
    int K, Nd, Nw, act; float alpha;
    // EXT instantiation only.  in_scale / in_shift: per-channel affine on X as it is loaded (a BatchNorm in front of
    // the layer that is never materialised): X' = bf16(X * in_scale[k] + in_shift[k]), k < in_ctrue, else 0.
    // stats: per-block partial sums of the stored (bf16) output and its square, part[2][Nd][stats_stride] at column
    // stats_off + blockIdx.x like pconv's STATS epilogue -- the BatchNorm behind the layer needs no pass over Y.
    const float* in_scale; const float* in_shift; int in_ctrue;
    float* stats; int stats_stride, stats_off;
};

#define SK_WPITCH(K) ((K) * 2 + 32)     // K*2 bytes + 32: (pitch/16) == 2 (mod 4) for K = 32, 64, 96, 128

template <int NT, int KS, bool SLOW, bool EXT = false>
__global__ __launch_bounds__(256) void gemm_sk_kernel(const GemmSkArgs a) {
    constexpr int BN = NT * 16, KP = KS * 32;
    constexpr int FRG = NT >= 8 ? 1 : (NT >= 4 ? 2 : 4);      // 16-row fragments per wave: keep acc <= 32-64 regs
    constexpr int WROWS = FRG * 16;
    constexpr int WP = SK_WPITCH(KP);
    constexpr int EROW = BN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;                                           // [BN][WP]
    char* sE = smem + BN * WP + (threadIdx.x >> 6) * (16 * EROW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = a.K;
    const int n0 = blockIdx.y * BN;             // column tile

    // ---- weights -> LDS (zero rows beyond Nw, zero k beyond K)
    for (int u = tid; u < BN * (KP / 8); u += 256) {
        int n = u / (KP / 8), k8 = u - n * (KP / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n0 + n < a.Nw && k8 * 8 < K) v = *(const uint4*)(a.w + (long)(n0 + n) * K + k8 * 8);
        *(uint4*)(sW + n * WP + k8 * 16) = v;
    }
    __syncthreads();

    float bv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int n = n0 + 16 * t + 4 * fq + e;
            bv[t][e] = (a.bias && n < a.Nw) ? a.bias[n] : 0.f;
        }

    // branch-free bounds-checked loads of X (rows >= M / k >= K read as zero): predicated global loads
    // become exec-masked branches and the FRG x KS loads of an iteration no longer overlap
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)(a.M * K * 2)), 0x00020000);

    // EXT: the affine of the k channels in LDS as fp32 [2][KP] (in registers it cost 16 * KS of them and the widest
    // tile fell to two waves per SIMD); statistics of the 8 output channels this lane stores (c8 = lane % (BN / 8) is
    // the same in every pass of the store loop)
    const bool aff = EXT && a.in_scale != nullptr, stats = EXT && a.stats != nullptr;
    float* sAff = (float*)(smem + BN * WP + 4 * 16 * EROW);
    float st_s[EXT ? 8 : 1], st_q[EXT ? 8 : 1];
    if (EXT) {
        if (aff) {
            for (int k = tid; k < KP; k += 256) {
                const bool real = k < a.in_ctrue;
                sAff[k] = real ? a.in_scale[k] : 0.f;
                sAff[KP + k] = real ? a.in_shift[k] : 0.f;
            }
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
    }

    // the X fragments of a wave's rows: FRG x KS independent 16-byte loads, requested one iteration ahead (a wave's
    // iteration is load -> MFMA -> epilogue -> LDS transpose -> store on 16-64 rows: with the load issued where it is
    // consumed every iteration paid a full memory latency, and the 1.66 M-row layer of model0 is 104 k iterations)
    auto xload = [&](long mb, bf16x8_t (&dst)[FRG][KS]) {
#pragma unroll
        for (int j = 0; j < FRG; ++j) {
            const long m = mb + 16 * j + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 32 + fq * 8;
                const bool ok = (m < a.M) & (k < K);
                const unsigned off = ok ? (unsigned)(m * K + k) * 2u : 0xFFFFFFF0u;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
                dst[j][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    };
    const long mstep = (long)gridDim.x * 4 * WROWS;
    bf16x8_t xn[FRG][KS];
    xload(((long)blockIdx.x * 4 + wave) * WROWS, xn);

    for (long m0 = ((long)blockIdx.x * 4 + wave) * WROWS; m0 < a.M; m0 += mstep) {
        bf16x8_t xf[FRG][KS];
#pragma unroll
        for (int j = 0; j < FRG; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[j][ks] = xn[j][ks];
        xload(m0 + mstep, xn);                  // (past the end: offsets beyond the descriptor, zeros)
        if (EXT && aff) {
#pragma unroll
            for (int j = 0; j < FRG; ++j) {
                const bool okm = m0 + 16 * j + fr < a.M;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4_t v = __builtin_bit_cast(u32x4_t, xf[j][ks]);
                    float f[8], sc8[8], sh8[8];
                    unpack8(make_uint4(v[0], v[1], v[2], v[3]), f);
                    *(float4*)(sc8) = *(const float4*)(sAff + ks * 32 + fq * 8);
                    *(float4*)(sc8 + 4) = *(const float4*)(sAff + ks * 32 + fq * 8 + 4);
                    *(float4*)(sh8) = *(const float4*)(sAff + KP + ks * 32 + fq * 8);
                    *(float4*)(sh8 + 4) = *(const float4*)(sAff + KP + ks * 32 + fq * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = okm ? fmaf(f[e], sc8[e], sh8[e]) : 0.f;
                    const uint4 r = pack8(f);
                    xf[j][ks] = __builtin_bit_cast(bf16x8_t, (u32x4_t){r.x, r.y, r.z, r.w});
                }
            }
        }
        f32x4 acc[NT][FRG];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < FRG; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const bf16x8_t*)(sW + (16 * t + fr) * WP + ks * 64 + fq * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < FRG; ++j)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[j][ks], acc[t][j], 0, 0, 0);
        }
        // ---- bias + activation (switch hoisted), padding channels forced to zero
        const int act = a.act;
        const float alpha = a.alpha;
        const float slope = act_slope(act, alpha);
        auto epi = [&](auto KA) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < FRG; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[t][j][e] + bv[t][e];
                        if (SLOW) v = act_apply_k<decltype(KA)::value>(v, act, alpha);
                        else v = act_apply_slope(v, slope);
                        acc[t][j][e] = (n0 + 16 * t + 4 * fq + e < a.Nw) ? v : 0.f;
                    }
        };
        if (SLOW) act_switch(act, epi);             // one branch chain per 64-row slab, not one per element
        else epi(std::integral_constant<int, -1>{});
        // ---- transposed store: 16 rows at a time through the wave's LDS scratch
#pragma unroll
        for (int j = 0; j < FRG; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *(uint2*)(sE + fr * EROW + (16 * t + 4 * fq) * 2) =
                    make_uint2(pack2bf(acc[t][j][0], acc[t][j][1]), pack2bf(acc[t][j][2], acc[t][j][3]));
            // The scratch is private to this wave and a wave's LDS operations execute in order, so ordering the
            // compiler (wavefront-scope fences) is all the transpose needs.  Workgroup-scope fences also drain the
            // global stores of the previous fragment (vmcnt(0)) and serialised the store latency per fragment.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = lane; u < 16 * (BN / 8); u += 64) {
                int px = u / (BN / 8), c8 = u - px * (BN / 8);
                long m = m0 + 16 * j + px;
                if (m < a.M && n0 + c8 * 8 < a.Nd) {
                    const uint4 o = *(const uint4*)(sE + px * EROW + c8 * 16);
                    *(uint4*)(a.y + m * a.Nd + n0 + c8 * 8) = o;
                    if (EXT && stats) {
                        float r[8];
                        unpack8(o, r);           // what the stored tensor holds
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            st_s[e] += r[e];
                            st_q[e] = fmaf(r[e], r[e], st_q[e]);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (EXT && stats) {
        // lanes sharing c8 = lane % (BN / 8) stored different rows of the same 8 channels: fold them, then the waves
#pragma unroll
        for (int o = BN / 8; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                st_s[e] += __shfl_xor(st_s[e], o, 64);
                st_q[e] += __shfl_xor(st_q[e], o, 64);
            }
        __syncthreads();                        // every wave is done with its transpose scratch
        float* red = (float*)(smem + BN * WP);  // [4 waves][2][BN]
        if (lane < BN / 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * BN + lane * 8 + e] = st_s[e];
                red[(wave * 2 + 1) * BN + lane * 8 + e] = st_q[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += 256) {
            const int k = i / BN, c = i - k * BN;
            if (n0 + c < a.Nd) {
                const float v = ((red[k * BN + c] + red[(2 + k) * BN + c]) + red[(4 + k) * BN + c]) + red[(6 + k) * BN + c];
                a.stats[((size_t)k * a.Nd + n0 + c) * a.stats_stride + a.stats_off + blockIdx.x] = v;
            }
        }
    }
}

// Rows per block-iteration and the grid of a launch (the statistics variant keeps the partial columns few: its
// waves walk several 64-row slabs instead)
template <int NT>
static long sk_blocks(long M, long ntiles, bool stats) {
    constexpr int FRG = NT >= 8 ? 1 : (NT >= 4 ? 2 : 4);
    long blocks = (M + 64 * FRG - 1) / (64 * FRG);
    if (ntiles > 1) {
        // Wide outputs: every block stages its 128-column weight tile (32 KiB) before it computes anything, so a block
        // that then owns only 64 rows spends its life staging (the generator's Dense 128 -> 6272 at 2000 rows: 1568
        // blocks, 50 MB of weight staging for 25 MB of output).  Cap the grid near `target` blocks instead; the row
        // loop strides by gridDim.x, so each block walks several 64-row slabs under one staged tile.
        static const long target = getenv("DDL_SK_WIDE_BLOCKS") ? atol(getenv("DDL_SK_WIDE_BLOCKS")) : 512;
        long cap = target / ntiles;
        if (cap < 1) cap = 1;
        if (blocks > cap) blocks = cap;
    }
    static const long scap = getenv("DDL_SK_STATS_BLOCKS") ? atol(getenv("DDL_SK_STATS_BLOCKS")) : 2048;
    const long lim = stats ? scap : 8192;
    return blocks > lim ? lim : blocks;
}

template <int NT, int KS>
static int launch_sk(const GemmSkArgs& a, hipStream_t s) {
    constexpr int BN = NT * 16, KP = KS * 32;
    size_t smem = (size_t)BN * SK_WPITCH(KP) + 4 * 16 * (BN * 2 + 16) + 2 * KP * 4;
    const long ntiles = (a.Nd + BN - 1) / BN;
    const bool ext = a.in_scale != nullptr || a.stats != nullptr;
    const long blocks = sk_blocks<NT>(a.M, ntiles, a.stats != nullptr);
    if (a.stats && (a.stats_off < 0 || a.stats_off + blocks > a.stats_stride)) return -1;
    dim3 grid((unsigned)blocks, (unsigned)ntiles);
    const bool fast = a.act == ACT_LINEAR || a.act == ACT_RELU || a.act == ACT_LEAKY;
    if (ext) {
        if (fast) gemm_sk_kernel<NT, KS, false, true><<<grid, 256, smem, s>>>(a);
        else gemm_sk_kernel<NT, KS, true, true><<<grid, 256, smem, s>>>(a);
    } else {
        if (fast) gemm_sk_kernel<NT, KS, false><<<grid, 256, smem, s>>>(a);
        else gemm_sk_kernel<NT, KS, true><<<grid, 256, smem, s>>>(a);
    }
    return 0;
}

static bool sk_in_range(long M, int K, int Nd) {
    if (K > 128 || M * (long)K >= (1L << 31)) return false;
    // wide outputs: 128-wide column tiles, only while X (re-read once per tile) stays small next to Y
    if (Nd > 128 && (M * (long)K > (8L << 20) || Nd > 65535 * 128L)) return false;
    return true;
}

static int sk_nt(int Nd) {
    const int NT = (Nd + 15) / 16;
    return NT <= 1 ? 1 : (NT <= 2 ? 2 : (NT <= 4 ? 4 : 8));
}

// Partial columns per channel that a statistics launch of this problem writes (0: outside the kernel's range).
DDL_API int ddl_gemm_sk_stats_cols(long M, int K, int Nd) {
    if (K % 8 || Nd % 8 || K <= 0 || Nd <= 0 || M <= 0 || !sk_in_range(M, K, Nd)) return 0;
    const int nt = sk_nt(Nd);
    const long ntiles = (Nd + nt * 16 - 1) / (nt * 16);
    return (int)(nt == 1 ? sk_blocks<1>(M, ntiles, true) : nt == 2 ? sk_blocks<2>(M, ntiles, true)
                 : nt == 4 ? sk_blocks<4>(M, ntiles, true) : sk_blocks<8>(M, ntiles, true));
}

// Returns 0 on success, 1 if the shape is outside this kernel's range (nothing launched).
// in_scale / in_shift (fp32 [>= K], both or neither): affine applied to X on load, channels >= in_ctrue read as zero.
// stats (fp32 [2][Nd][stats_stride], or null): BatchNorm partials of Y at columns stats_off .. + ddl_gemm_sk_stats_cols.
DDL_API int ddl_gemm_sk(const void* x, const void* w, const float* bias, void* y, long M, int K, int Nd, int Nw,
                        int act, float alpha, const float* in_scale, const float* in_shift, int in_ctrue,
                        float* stats, int stats_stride, int stats_off, hipStream_t s) {
    if (K % 8 || Nd % 8 || K <= 0 || Nd <= 0 || M <= 0) return -1;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return -1;
    if (!sk_in_range(M, K, Nd)) return 1;
    GemmSkArgs a{(const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, M, K, Nd, Nw, act, alpha,
                 in_scale, in_shift, in_ctrue, stats, stats_stride, stats_off};
    const int KS = (K + 31) / 32;
    const int nt = sk_nt(Nd);
#define SK_CASE(NT_, KS_) if (nt == NT_ && KS == KS_) return launch_sk<NT_, KS_>(a, s);
    SK_CASE(1, 1) SK_CASE(1, 2) SK_CASE(1, 3) SK_CASE(1, 4)
    SK_CASE(2, 1) SK_CASE(2, 2) SK_CASE(2, 3) SK_CASE(2, 4)
    SK_CASE(4, 1) SK_CASE(4, 2) SK_CASE(4, 3) SK_CASE(4, 4)
    SK_CASE(8, 1) SK_CASE(8, 2) SK_CASE(8, 3) SK_CASE(8, 4)
#undef SK_CASE
    return 1;
}
