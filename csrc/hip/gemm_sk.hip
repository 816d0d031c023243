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
};

#define SK_WPITCH(K) ((K) * 2 + 32)     // K*2 bytes + 32: (pitch/16) == 2 (mod 4) for K = 32, 64, 96, 128

template <int NT, int KS, bool SLOW>
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

    for (long m0 = ((long)blockIdx.x * 4 + wave) * WROWS; m0 < a.M; m0 += (long)gridDim.x * 4 * WROWS) {
        // ---- all X fragments of the wave's 64 rows: FR x KS independent 16-byte loads in flight
        bf16x8_t xf[FRG][KS];
#pragma unroll
        for (int j = 0; j < FRG; ++j) {
            long m = m0 + 16 * j + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int k = ks * 32 + fq * 8;
                const bool ok = (m < a.M) & (k < K);
                const unsigned off = ok ? (unsigned)(m * K + k) * 2u : 0xFFFFFFF0u;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
                xf[j][ks] = __builtin_bit_cast(bf16x8_t, v);
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
                if (m < a.M && n0 + c8 * 8 < a.Nd)
                    *(uint4*)(a.y + m * a.Nd + n0 + c8 * 8) = *(const uint4*)(sE + px * EROW + c8 * 16);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

template <int NT, int KS>
static int launch_sk(const GemmSkArgs& a, hipStream_t s) {
    constexpr int BN = NT * 16, KP = KS * 32;
    size_t smem = (size_t)BN * SK_WPITCH(KP) + 4 * 16 * (BN * 2 + 16);
    constexpr int FRG = NT >= 8 ? 1 : (NT >= 4 ? 2 : 4);
    long blocks = (a.M + 64 * FRG - 1) / (64 * FRG);
    const long ntiles = (a.Nd + BN - 1) / BN;
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
    dim3 grid((unsigned)(blocks > 8192 ? 8192 : blocks), (unsigned)ntiles);
    const bool fast = a.act == ACT_LINEAR || a.act == ACT_RELU || a.act == ACT_LEAKY;
    if (fast) gemm_sk_kernel<NT, KS, false><<<grid, 256, smem, s>>>(a);
    else gemm_sk_kernel<NT, KS, true><<<grid, 256, smem, s>>>(a);
    return 0;
}

// Returns 0 on success, 1 if the shape is outside this kernel's range (nothing launched).
DDL_API int ddl_gemm_sk(const void* x, const void* w, const float* bias, void* y, long M, int K, int Nd, int Nw,
                        int act, float alpha, hipStream_t s) {
    if (K % 8 || Nd % 8 || K <= 0 || Nd <= 0 || M <= 0) return -1;
    if (K > 128 || M * (long)K >= (1L << 31)) return 1;
    // wide outputs: 128-wide column tiles, only while X (re-read once per tile) stays small next to Y
    if (Nd > 128 && (M * (long)K > (8L << 20) || Nd > 65535 * 128L)) return 1;
    GemmSkArgs a{(const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, M, K, Nd, Nw, act, alpha};
    const int NT = (Nd + 15) / 16, KS = (K + 31) / 32;
    const int nt = NT <= 1 ? 1 : (NT <= 2 ? 2 : (NT <= 4 ? 4 : 8));
#define SK_CASE(NT_, KS_) if (nt == NT_ && KS == KS_) return launch_sk<NT_, KS_>(a, s);
    SK_CASE(1, 1) SK_CASE(1, 2) SK_CASE(1, 3) SK_CASE(1, 4)
    SK_CASE(2, 1) SK_CASE(2, 2) SK_CASE(2, 3) SK_CASE(2, 4)
    SK_CASE(4, 1) SK_CASE(4, 2) SK_CASE(4, 3) SK_CASE(4, 4)
    SK_CASE(8, 1) SK_CASE(8, 2) SK_CASE(8, 3) SK_CASE(8, 4)
#undef SK_CASE
    return 1;
}
