// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// One kernel serves Conv2D forward (fprop), Conv2DTranspose forward / Conv2D backward-data
// (dgrad) and Dense: everything is phrased as a *class grid* problem
//
//   dst[b, ph + sdh*i, pw + sdw*j, n] = act( bias[n] +
//        sum_{t < ntaps} sum_{c < Cs} src[b, i*ssh + offh[t], j*ssw + offw[t], c] * wgt[n][wtap[t]][c] )
//
// fprop : one class, (ssh,ssw) = conv stride, offsets kh*dh - pt.
// dgrad : one class per output parity (ph,pw) in [0,sh)x[0,sw); only taps with
//         (ph + pt - kh*dh) % sh == 0 contribute, with offset (ph + pt - kh*dh)/sh  (sub-pixel
//         decomposition: no zero-insertion, no wasted MACs).
// Out-of-range source pixels read as zero, so TF's asymmetric `same` padding, `valid` transpose
// output padding and kernels larger than the image need no special cases.
//
// Tiling (wave64, 256 threads = 4 waves): block tile BM pixels x BN channels, K step 64.
// MFMA operands are swapped -- weights are the A operand (rows = channel n), pixels the B operand
// (cols) -- so a lane's 4 accumulator registers are 4 consecutive channels of one pixel and the
// epilogue stores 8-byte bf16x4 vectors.  LDS rows are 128 B of data, unpadded, with the 16-byte slot XOR-ed by the
// row's low three bits (slot' = chunk ^ (row & 7)): the four 16-lane groups of a ds_read_b128 fragment read hit 16
// distinct 16-B slots of the 256-B bank row and a ds_write_b128 group of 8 lanes writes one whole row
// (conflict-free both ways).  The earlier 160-B padded pitch was conflict-free too but made a 128 x 128 tile 80 KB of
// LDS -- ONE block per CU, four waves, every latency exposed; at 64 KB two blocks co-reside.  Global->LDS staging goes through a ring of three register stages: the gathers
// of K-step s+3 are issued before the MFMAs of step s, so three steps of loads are in flight per
// wave (the gather latency, not the math, bounds these small-N problems); one barrier per step.
#include "common.h"

struct ClassDesc { int ph, pw, Hc, Wc, ntaps, tap_off, M, pad; };

#define BK 64                 // K elements per LDS stage
#define PITCH 128             // bytes per LDS row (64 bf16 K-elements, XOR-swizzled 16-B slots)

struct ConvArgs {
    const bf16_t* src; const bf16_t* wgt; const float* bias; bf16_t* dst;
    const ClassDesc* classes; const int4* taps;     // tap = {offh, offw, wtap, 0}
    int B, Hs, Ws, Cs;        // source tensor [B][Hs][Ws][Cs]
    int Hd, Wd, Nd;           // destination tensor [B][Hd][Wd][Nd] (Nd = physical channels)
    int Nw;                   // weight rows (true output channels) ; n >= Nw produces 0
    int ssh, ssw, sdh, sdw;
    int Ttot;                 // taps per weight row
    int act; float alpha;
    int max_taps;
    int ksplit; float* slab;   // split-K (single-class problems, Dense with huge K): fp32 partials [split][M][Nd]
    // optional: dst = bf16(bf16(conv) * (y > 0 ? 1 : mul_slope)), y laid out like dst -- act'(y) of the layer below
    // (linear / relu / leaky) fused into an input gradient; the same two roundings as the separate act_bwd pass
    const bf16_t* mul_y; float mul_slope;
    // optional: per-block column sums / sums of squares of the STORED (bf16-rounded) output for the BatchNorm behind
    // the layer: stats[(s * Nd + n) * stats_stride + stats_off + col], s = 0 | 1, col = pixel tile * classes + class
    // (pixel-tile-major = batch-major); blocks past a class's rows write zeros
    float* stats; int stats_stride, stats_off;
};

#define PITCH1 64
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void igemm_stats_zero(const ConvArgs& a, int n0, int BN, int col) {
    for (int c = threadIdx.x; c < BN; c += 256)
        if (n0 + c < a.Nd) {
            a.stats[(size_t)(n0 + c) * a.stats_stride + a.stats_off + col] = 0.f;
            a.stats[(size_t)(a.Nd + n0 + c) * a.stats_stride + a.stats_off + col] = 0.f;
        }
}

// Column sums / sums of squares of the block's stored tile: one 16-channel MFMA tile at a time (eight live sums),
// the 16 pixel lanes by shuffles, the `waves_m` waves that share a channel range through LDS in a fixed order.
// RAW: `acc` still holds the plain sums (bias and activation are applied here as the store applies them); else the
// activated values.  `red` ([waves_m][BN][2] floats) aliases the staging buffers: the leading barrier waits for
// every wave's last fragment read.
template <int NT, int MT, bool RAW>
__device__ __forceinline__ void igemm_stats_pass(const ConvArgs& a, const ClassDesc& cd, const f32x4 (&acc)[NT][MT], int mb,
                                                 int wch0, int n0, int BN, int wm, int waves_m, int lane, float* red,
                                                 int col) {
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();
    act_switch(RAW ? a.act : ACT_LINEAR, [&](auto KA) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s_[4] = {0.f, 0.f, 0.f, 0.f}, q_[4] = {0.f, 0.f, 0.f, 0.f}, bias_[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wch0 + 16 * t + 4 * fq + e;
                bias_[e] = (RAW && n < a.Nw && a.bias) ? a.bias[n] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const bool ok = mb + 16 * j + fr < cd.M;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[t][j][e];
                    if (RAW) v = n0 + wch0 + 16 * t + 4 * fq + e < a.Nw
                                     ? act_apply_k<decltype(KA)::value>(v + bias_[e], a.act, a.alpha) : 0.f;
                    const float r = ok ? bf2f(f2bf(v)) : 0.f;           // what the stored tensor holds
                    s_[e] += r;
                    q_[e] = fmaf(r, r, q_[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s_[e] += __shfl_xor(s_[e], o, 64);
                    q_[e] += __shfl_xor(q_[e], o, 64);
                }
            if (fr == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = wch0 + 16 * t + 4 * fq + e;
                    red[(wm * BN + c) * 2 + 0] = s_[e];
                    red[(wm * BN + c) * 2 + 1] = q_[e];
                }
            }
        }
    });
    __syncthreads();
    for (int c = threadIdx.x; c < BN; c += 256)
        if (n0 + c < a.Nd) {
            float s0 = 0.f, s1 = 0.f;
            for (int w = 0; w < waves_m; ++w) { s0 += red[(w * BN + c) * 2]; s1 += red[(w * BN + c) * 2 + 1]; }
            a.stats[(size_t)(n0 + c) * a.stats_stride + a.stats_off + col] = s0;
            a.stats[(size_t)(a.Nd + n0 + c) * a.stats_stride + a.stats_off + col] = s1;
        }
}

#define DEPTH 3               // global loads are issued DEPTH K-steps ahead of their use

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_conv_kernel(const ConvArgs a) {
    constexpr int WAVES_M = BM / 64;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WN = BN / WAVES_N;          // channels per wave
    constexpr int NT = WN / 16;               // 16-channel MFMA tiles per wave
    constexpr int RA = BM / 32;               // pixel rows staged per thread
    constexpr int RW = (BN + 31) / 32;        // weight rows staged per thread
    static_assert(WN >= 16 && WN % 16 == 0, "bad tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                  // [2][BM][PITCH]
    char* sW = smem + 2 * BM * PITCH;                 // [2][BN][PITCH]
    int4* sT = (int4*)(smem + 2 * (BM + BN) * PITCH); // [max_taps]

    const int split = a.ksplit > 1 ? blockIdx.y : 0;
    const ClassDesc cd = a.classes[a.ksplit > 1 ? 0 : blockIdx.y];
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.z * BN;
    if (m0 >= cd.M) {
        if (a.stats) igemm_stats_zero(a, n0, BN, blockIdx.x * gridDim.y + blockIdx.y);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    for (int i = tid; i < cd.ntaps; i += 256) sT[i] = a.taps[cd.tap_off + i];

    // ---- per-thread staging assignment: kgroup q (8 k-elements of the 64-wide stage), rows r0 + 32*r
    const int q = tid & 7, r0 = tid >> 3;
    const int qsw = (q ^ (r0 & 7)) * 16;      // swizzled store slot (rows r0 + 32 r share r0's low bits)
    const int HW = cd.Hc * cd.Wc;
    const int Cs = a.Cs;
    int abase[RA], aih[RA], aiw[RA];          // element offsets fit 31 bits (checked on the host)
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        int m = m0 + r0 + 32 * r;
        bool v = m < cd.M;
        int mm = v ? m : 0;
        int b = mm / HW, rem = mm - b * HW;
        int i = rem / cd.Wc, j = rem - i * cd.Wc;
        abase[r] = b * a.Hs * a.Ws * Cs;
        aih[r] = v ? i * a.ssh : -(1 << 24);  // invalid rows fail the bounds test below
        aiw[r] = j * a.ssw;
    }
    int wbase[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int nl = r0 + 32 * r, n = n0 + nl;
        wbase[r] = (nl < BN && n < a.Nw) ? n * a.Ttot * Cs : -1;
    }
    const int nsteps_all = (cd.ntaps * Cs + BK - 1) / BK;
    const int per_split = (nsteps_all + a.ksplit - 1) / a.ksplit;
    const int s_begin = split * per_split;
    const int nsteps = max(0, min(nsteps_all, s_begin + per_split) - s_begin);
    int tap = (s_begin * BK + q * 8) / Cs, c = (s_begin * BK + q * 8) - tap * Cs;
    int lstep = 0;                            // K-steps loaded so far

    // bounds-checked buffer loads: an out-of-image / out-of-range element gets an offset beyond the
    // descriptor's size and reads as zero, so the gathers are branch-free and the compiler can count
    // them (a predicated global_load becomes an exec-masked branch and forces vmcnt(0) drains)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((unsigned)a.B * a.Hs * a.Ws * Cs * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wgt, 0, (int)((unsigned)a.Nw * a.Ttot * Cs * 2u), 0x00020000);
    const int dq = BK / Cs, dr = BK - dq * Cs;
    const int last_tap = max(cd.ntaps - 1, 0);

    u32x4 regA[DEPTH][RA], regW[DEPTH][RW];   // register ring of in-flight stages (statically indexed)
    auto gload = [&](u32x4 (&rA)[RA], u32x4 (&rW)[RW]) {
        const bool tv = (tap < cd.ntaps) & (lstep < nsteps);
        const int4 te = sT[min(tap, last_tap)];
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            int ih = aih[r] + te.x, iw = aiw[r] + te.y;
            bool ok = tv & ((unsigned)ih < (unsigned)a.Hs) & ((unsigned)iw < (unsigned)a.Ws);   // no short-circuit
            unsigned off = ok ? (unsigned)(abase[r] + (ih * a.Ws + iw) * Cs + c) * 2u : 0xFFFFFFF0u;
            rA[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            bool ok = tv & (wbase[r] >= 0);
            unsigned off = ok ? (unsigned)(wbase[r] + te.z * Cs + c) * 2u : 0xFFFFFFF0u;
            rW[r] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
        }
        ++lstep;
        tap += dq;
        c += dr;
        if (c >= Cs) { c -= Cs; ++tap; }
    };
    auto lstore = [&](int buf, const u32x4 (&rA)[RA], const u32x4 (&rW)[RW]) {
#pragma unroll
        for (int r = 0; r < RA; ++r)
            *(u32x4*)(sA + (buf * BM + r0 + 32 * r) * PITCH + qsw) = rA[r];
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (BN % 32 == 0 || r0 + 32 * r < BN) *(u32x4*)(sW + (buf * BN + r0 + 32 * r) * PITCH + qsw) = rW[r];
    };

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();                 // tap table visible
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) gload(regA[u], regW[u]);
    lstore(0, regA[0], regW[0]);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    auto mma = [&](int cur) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8_t wf[NT], xf[4];
#pragma unroll
            for (int i = 0; i < NT; ++i)
                wf[i] = *(const bf16x8_t*)(sW + (cur * BN + wn * WN + 16 * i + fr) * PITCH + (((h * 4 + fq) ^ (fr & 7)) * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xf[j] = *(const bf16x8_t*)(sA + (cur * BM + wm * 64 + 16 * j + fr) * PITCH + (((h * 4 + fq) ^ (fr & 7)) * 16));
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    };
    // main loop: whole groups of DEPTH steps, no guards (a simple loop keeps the compiler's vmcnt
    // counting exact: two stages of gathers stay in flight across every barrier)
    int s = 0;
    for (; s + DEPTH <= nsteps; s += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int cur = (s + u) & 1;
            gload(regA[u], regW[u]);                    // K-step s+u+DEPTH (slot u was stored one step ago)
            mma(cur);
            lstore(cur ^ 1, regA[(u + 1) % DEPTH], regW[(u + 1) % DEPTH]);     // K-step s+u+1
            __syncthreads();
        }
    }
    // tail: nsteps % DEPTH steps whose gathers are already in the ring
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) {
        if (s + u < nsteps) {                           // block-uniform
            const int cur = (s + u) & 1;
            mma(cur);
            if (s + u + 1 < nsteps) {
                lstore(cur ^ 1, regA[(u + 1) % DEPTH], regW[(u + 1) % DEPTH]);
                __syncthreads();
            }
        }
    }

    if (a.ksplit > 1) {      // raw fp32 partials; bias / activation / bf16 happen in splitk_finish_kernel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + wm * 64 + 16 * j + fr;
            if (m >= cd.M) continue;
            float* prow = a.slab + ((long)split * cd.M + m) * a.Nd;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int nb = n0 + wn * WN + 16 * t + 4 * fq;
                if (nb + 3 < a.Nd) *(float4*)(prow + nb) = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
            }
        }
        return;
    }
    // ---- epilogue: lane holds channels nb..nb+3 (nb = 4*fq) of pixel fr in each 16x16 tile
    // bias + activation in place first, under ONE activation dispatch (a run-time switch per element is a chain of
    // scalar branches per element, common.h); channels beyond the weight rows are zero
    act_switch(a.act, [&](auto KA) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn * WN + 16 * t + 4 * fq + e;
                const bool live = n < a.Nw;
                const float bn_ = (live && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t][j][e] = live ? act_apply_k<decltype(KA)::value>(acc[t][j][e] + bn_, a.act, a.alpha) : 0.f;
            }
    });
    if (a.stats)                         // uniform; never together with the fused multiply (host-checked)
        igemm_stats_pass<NT, 4, false>(a, cd, acc, m0 + wm * 64, wn * WN, n0, BN, wm, WAVES_M, lane, (float*)smem,
                                       blockIdx.x * gridDim.y + blockIdx.y);
    const bool vec_ok = (a.Nd & 3) == 0;
    if (a.mul_y && vec_ok) {            // uniform.  All y loads first (one round trip), then the multiplies
        uint2 yv[NT][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + 16 * j + fr;
            const int mm = m < cd.M ? m : 0;
            const int b = mm / HW, rem = mm - b * HW;
            const int i = rem / cd.Wc, jj = rem - i * cd.Wc;
            const long pix = ((long)b * a.Hd + cd.ph + (long)a.sdh * i) * a.Wd + cd.pw + (long)a.sdw * jj;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nb = n0 + wn * WN + 16 * t + 4 * fq;
                yv[t][j] = (m < cd.M && nb < a.Nd) ? *(const uint2*)(a.mul_y + pix * a.Nd + nb) : make_uint2(0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float y4[4] = {__uint_as_float(yv[t][j].x << 16), __uint_as_float(yv[t][j].x & 0xffff0000u),
                                     __uint_as_float(yv[t][j].y << 16), __uint_as_float(yv[t][j].y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[t][j][e] = bf2f(f2bf(acc[t][j][e])) * (y4[e] > 0.f ? 1.f : a.mul_slope);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + wm * 64 + 16 * j + fr;
        if (m >= cd.M) continue;
        int b = m / HW, rem = m - b * HW;
        int i = rem / cd.Wc, jj = rem - i * cd.Wc;
        long pix = ((long)b * a.Hd + cd.ph + (long)a.sdh * i) * a.Wd + cd.pw + (long)a.sdw * jj;
        bf16_t* drow = a.dst + pix * a.Nd;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int nb = n0 + wn * WN + 16 * t + 4 * fq;
            if (nb >= a.Nd) continue;
            const float v[4] = {acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]};
            if (vec_ok) {
                *(uint2*)(drow + nb) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (nb + e < a.Nd) drow[nb + e] = f2bf(v[e]);
            }
        }
    }
}

// ---- the shallow variant: 32-wide K stages, loads one step ahead.  Less LDS and fewer registers per
// block (more resident blocks), and no pipeline ramp: it wins on short reductions and on grids
// that already fill the CUs several times over; the autotuner times both.
// LDS rows are 64 B (one 32-wide K stage), unpadded, with the 16-byte slot XOR-swizzled by the row's group of
// four: slot' = slot ^ ((-(row >> 2)) & 3).  ds_write_b128 is served in groups of 8 contiguous lanes over 32
// banks (two whole rows = one 128-byte bank row, any in-row permutation is conflict-free); ds_read_b128 in the
// four 16-lane groups {0-3,12-15,20-27},... over 64 banks, and this XOR sends the four lanes of a group that
// share (row mod 4) to four different slots.  The padded layout (96 B) conflicted 2-way on every store
// (30 % of LDS cycles, profiles/r1_dcgan64_pmc_counters.md) and used 1.5x the LDS.

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_conv_kernel_k32(const ConvArgs a) {
    constexpr int WAVES_M = BM / 64;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WN = BN / WAVES_N;          // channels per wave
    constexpr int NT = WN / 16;               // 16-channel MFMA tiles per wave
    constexpr int RA = BM / 64;               // pixel rows staged per thread
    constexpr int RW = (BN + 63) / 64;        // weight rows staged per thread
    static_assert(WN >= 16 && WN % 16 == 0, "bad tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                  // [2][BM][PITCH1]
    char* sW = smem + 2 * BM * PITCH1;                 // [2][BN][PITCH1]
    int4* sT = (int4*)(smem + 2 * (BM + BN) * PITCH1); // [max_taps]

    const int split = a.ksplit > 1 ? blockIdx.y : 0;
    const ClassDesc cd = a.classes[a.ksplit > 1 ? 0 : blockIdx.y];
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.z * BN;
    if (m0 >= cd.M) {
        if (a.stats) igemm_stats_zero(a, n0, BN, blockIdx.x * gridDim.y + blockIdx.y);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    for (int i = tid; i < cd.ntaps; i += 256) sT[i] = a.taps[cd.tap_off + i];

    // ---- per-thread staging assignment: kgroup q (8 k-elements), rows r0 + 64*r
    const int q = tid & 3, r0 = tid >> 2;
    const int qsw = (q ^ ((-(r0 >> 2)) & 3)) * 16;      // swizzled store slot (rows r0 + 64 r share it)
    const int HW = cd.Hc * cd.Wc;
    int abase[RA];                            // element offsets fit 31 bits (checked on the host)
    int aih[RA], aiw[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        int m = m0 + r0 + 64 * r;
        bool v = m < cd.M;
        int mm = v ? m : 0;
        int b = mm / HW, rem = mm - b * HW;
        int i = rem / cd.Wc, j = rem - i * cd.Wc;
        abase[r] = b * a.Hs * a.Ws * a.Cs;
        aih[r] = v ? i * a.ssh : -(1 << 24);  // invalid rows fail the bounds test below
        aiw[r] = j * a.ssw;
    }
    const int Cs = a.Cs;
    const int nsteps_all = (cd.ntaps * Cs + 31) / 32;
    const int per_split = (nsteps_all + a.ksplit - 1) / a.ksplit;
    const int s_begin = split * per_split;
    const int nsteps = max(0, min(nsteps_all, s_begin + per_split) - s_begin);
    int tap = (s_begin * 32 + q * 8) / Cs, c = (s_begin * 32 + q * 8) - tap * Cs;

    // branch-free bounds-checked buffer loads (see the deep variant): a predicated global_load is an exec-masked
    // branch and serialises the gathers of a stage
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((unsigned)a.B * a.Hs * a.Ws * Cs * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wgt, 0, (int)((unsigned)a.Nw * a.Ttot * Cs * 2u), 0x00020000);
    const int last_tap = max(cd.ntaps - 1, 0);
    int wbase[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int nl = r0 + 64 * r, n = n0 + nl;
        wbase[r] = (nl < BN && n < a.Nw) ? n * a.Ttot * Cs : -1;
    }

    // ---- uniform-tap gather (UT).  When a K stage never straddles two taps (Cs % 32 == 0) the tap of a stage is the
    // same for every thread of the block: its table entry, the tap's byte offset and the position along the channel
    // axis are SCALAR values (SALU, s_load), and what is left per gathered row is one add and a validity bit out of a
    // per-row mask built once (bit = position of the tap's (offh, offw) inside the class's offset box, host-packed
    // into ClassDesc::pad and the tap's fourth word).  The generic gather below spends ~12 vector instructions per
    // row and stage -- two of them 64-bit multiply-adds -- against 8-16 MFMAs per wave: these kernels were bound by
    // address arithmetic, not by the matrix pipe or by memory.
    const bool ut = cd.pad != 0 && (Cs & 31) == 0;
    unsigned vmask[RA], rowoff[RA], wrow[RW];
    int ut_tap = 0, ut_c = 0;
    if (ut) {
        const int minh = ((cd.pad >> 1) & 127) - 64, minw = ((cd.pad >> 8) & 127) - 64;
        const int rw = (cd.pad >> 15) & 63, rh = (cd.pad >> 21) & 63;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            unsigned wm = 0, m2 = 0;
            for (int wb = 0; wb < rw; ++wb) wm |= ((unsigned)(aiw[r] + minw + wb) < (unsigned)a.Ws ? 1u : 0u) << wb;
            for (int hb = 0; hb < rh; ++hb)
                m2 |= ((unsigned)(aih[r] + minh + hb) < (unsigned)a.Hs ? wm : 0u) << (hb * rw);
            vmask[r] = m2;                        // rows beyond M have aih = -2^24: no bit set
            rowoff[r] = (unsigned)(abase[r] + (aih[r] * a.Ws + aiw[r]) * Cs + q * 8) * 2u;
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) wrow[r] = wbase[r] >= 0 ? (unsigned)(wbase[r] + q * 8) * 2u : 0xF0000000u;
        ut_tap = (s_begin * 32) / Cs;
        ut_c = s_begin * 32 - ut_tap * Cs;
    }
    const int4* __restrict__ gtaps = a.taps + cd.tap_off;

    u32x4 regA[RA], regW[RW];
    auto gload = [&]() {
        if (ut) {
            const bool tv = ut_tap < cd.ntaps;
            const int4 te = gtaps[min(ut_tap, last_tap)];                 // uniform address: a scalar load
            const unsigned soffA = (unsigned)(((te.x * a.Ws + te.y) * Cs + ut_c) * 2);
            const unsigned soffW = (unsigned)((te.z * Cs + ut_c) * 2);
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const bool ok = tv & (((vmask[r] >> te.w) & 1u) != 0u);
                regA[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? rowoff[r] + soffA : 0xFFFFFFF0u, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r)
                regW[r] = __builtin_amdgcn_raw_buffer_load_b128(rsW, tv ? wrow[r] + soffW : 0xFFFFFFF0u, 0, 0);
            return;
        }
        const bool tv = tap < cd.ntaps;
        const int4 te = sT[min(tap, last_tap)];
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            int ih = aih[r] + te.x, iw = aiw[r] + te.y;
            bool ok = tv & ((unsigned)ih < (unsigned)a.Hs) & ((unsigned)iw < (unsigned)a.Ws);
            unsigned off = ok ? (unsigned)(abase[r] + (ih * a.Ws + iw) * Cs + c) * 2u : 0xFFFFFFF0u;
            regA[r] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            bool ok = tv & (wbase[r] >= 0);
            unsigned off = ok ? (unsigned)(wbase[r] + te.z * Cs + c) * 2u : 0xFFFFFFF0u;
            regW[r] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RA; ++r)
            *(u32x4*)(sA + (buf * BM + r0 + 64 * r) * PITCH1 + qsw) = regA[r];
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (r0 + 64 * r < BN) *(u32x4*)(sW + (buf * BN + r0 + 64 * r) * PITCH1 + qsw) = regW[r];
    };
    auto advance = [&]() {
        if (ut) {
            ut_c += 32;
            if (ut_c >= Cs) { ut_c = 0; ++ut_tap; }
            return;
        }
        c += 32;
        while (c >= Cs) { c -= Cs; ++tap; }
    };

    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();                 // tap table visible
    if (nsteps > 0) {
        gload();
        lstore(0);
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    const int fqsw = (fq ^ ((-(fr >> 2)) & 3)) * 16;   // swizzled read slot (tile bases are multiples of 16 rows)
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) { advance(); gload(); }
        bf16x8_t wf[NT], xf[4];
#pragma unroll
        for (int i = 0; i < NT; ++i)
            wf[i] = *(const bf16x8_t*)(sW + (cur * BN + wn * WN + 16 * i + fr) * PITCH1 + fqsw);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            xf[j] = *(const bf16x8_t*)(sA + (cur * BM + wm * 64 + 16 * j + fr) * PITCH1 + fqsw);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    if (a.ksplit > 1) {      // raw fp32 partials; bias / activation / bf16 happen in splitk_finish_kernel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + wm * 64 + 16 * j + fr;
            if (m >= cd.M) continue;
            float* prow = a.slab + ((long)split * cd.M + m) * a.Nd;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int nb = n0 + wn * WN + 16 * t + 4 * fq;
                if (nb + 3 < a.Nd) *(float4*)(prow + nb) = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
            }
        }
        return;
    }
    // ---- epilogue: lane holds channels nb..nb+3 (nb = 4*fq) of pixel fr in each 16x16 tile
    // bias + activation in place first, under ONE activation dispatch (a run-time switch per element is a chain of
    // scalar branches per element, common.h); channels beyond the weight rows are zero
    act_switch(a.act, [&](auto KA) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn * WN + 16 * t + 4 * fq + e;
                const bool live = n < a.Nw;
                const float bn_ = (live && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t][j][e] = live ? act_apply_k<decltype(KA)::value>(acc[t][j][e] + bn_, a.act, a.alpha) : 0.f;
            }
    });
    if (a.stats)                         // uniform; never together with the fused multiply (host-checked)
        igemm_stats_pass<NT, 4, false>(a, cd, acc, m0 + wm * 64, wn * WN, n0, BN, wm, WAVES_M, lane, (float*)smem,
                                       blockIdx.x * gridDim.y + blockIdx.y);
    const bool vec_ok = (a.Nd & 3) == 0;
    if (a.mul_y && vec_ok) {            // uniform.  All y loads first (one round trip), then the multiplies
        uint2 yv[NT][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + 16 * j + fr;
            const int mm = m < cd.M ? m : 0;
            const int b = mm / HW, rem = mm - b * HW;
            const int i = rem / cd.Wc, jj = rem - i * cd.Wc;
            const long pix = ((long)b * a.Hd + cd.ph + (long)a.sdh * i) * a.Wd + cd.pw + (long)a.sdw * jj;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nb = n0 + wn * WN + 16 * t + 4 * fq;
                yv[t][j] = (m < cd.M && nb < a.Nd) ? *(const uint2*)(a.mul_y + pix * a.Nd + nb) : make_uint2(0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float y4[4] = {__uint_as_float(yv[t][j].x << 16), __uint_as_float(yv[t][j].x & 0xffff0000u),
                                     __uint_as_float(yv[t][j].y << 16), __uint_as_float(yv[t][j].y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[t][j][e] = bf2f(f2bf(acc[t][j][e])) * (y4[e] > 0.f ? 1.f : a.mul_slope);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + wm * 64 + 16 * j + fr;
        if (m >= cd.M) continue;
        int b = m / HW, rem = m - b * HW;
        int i = rem / cd.Wc, jj = rem - i * cd.Wc;
        long pix = ((long)b * a.Hd + cd.ph + (long)a.sdh * i) * a.Wd + cd.pw + (long)a.sdw * jj;
        bf16_t* drow = a.dst + pix * a.Nd;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int nb = n0 + wn * WN + 16 * t + 4 * fq;
            if (nb >= a.Nd) continue;
            const float v[4] = {acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]};
            if (vec_ok) {
                *(uint2*)(drow + nb) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (nb + e < a.Nd) drow[nb + e] = f2bf(v[e]);
            }
        }
    }
}

// ---- the LDS-DMA variant, for problems large enough to be bound by the staging path rather than by latency.
// The register-staged kernels above push every operand byte through VGPRs and a ds_write_b128, and the wide LDS
// store is the slowest LDS instruction there is (13 cycles per wave-instruction, ~79 B/clk per CU against 256 B/clk
// for ds_read_b128): for a 128 x 128 x 32 stage that is 16 stores = 208 LDS cycles beside 128 cycles of fragment
// reads and 256 MFMA cycles per SIMD -- the stores, not the matrix pipe, bounded the big layers at ~650 TFLOP/s.
// Here each wave fills its share of the stage with `buffer_load_dwordx4 ... lds`: the loaded bytes go straight into
// LDS (wave-uniform base + 16 B x lane), no VGPR and no LDS store instruction.  Consequences for the layout:
//   * one wave-instruction writes 1 KiB = 16 rows x 64 B linearly, so the XOR swizzle of the 16-byte slots is applied
//     on the SOURCE side: the lane that lands in (row, slot) fetches k-group  slot ^ ((-(row >> 2)) & 3)  of that row
//     (the same involution the fragment reads apply);
//   * padding pixels, rows beyond M and channels beyond Nw fetch an out-of-range buffer offset, which writes zeros;
//   * the tap of a K stage must be uniform over the block (uniform-tap tables, Cs % 32 == 0): the per-stage vector
//     work is one mask test, one select and one add per 16-row piece.
// A ring of STAGES LDS buffers keeps STAGES - 1 stages of DMA in flight across ONE raw s_barrier per K step, retired
// by a counted s_waitcnt vmcnt (a __syncthreads() would drain the whole queue at every step).
// Blocks are numbered so that the channel tiles of one pixel tile run back to back on the same XCD (its L2 then
// serves the second gather of the same pixels): id = blockIdx.x, xcd = id % 8, k = id / 8,
// (n tile, m tile) = (k % n_ntiles, (k / n_ntiles) * 8 + xcd).
template <int NT, int MT>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& a, const ClassDesc& cd, const f32x4 (&acc)[NT][MT],
                                               int mb, int nb0, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int HW = cd.Hc * cd.Wc;
    const bool vec_ok = (a.Nd & 3) == 0;
    const bool mul = a.mul_y && vec_ok;
    constexpr int JG = NT >= 8 ? 2 : 4;         // pixel tiles per group
    // ONE activation dispatch around the whole epilogue (a run-time switch per element is a chain of scalar branches
    // per element, common.h); inside it a few pixel tiles at a time, so that a 128-accumulator wave tile never holds
    // a second copy of itself: the y loads of the group first (one round trip), then activation, multiply, store
    act_switch(a.act, [&](auto KA) {
        float bias_[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nb0 + 16 * t + 4 * fq + e;
                bias_[t][e] = (n < a.Nw && a.bias) ? a.bias[n] : 0.f;
            }
#pragma unroll
        for (int j0 = 0; j0 < MT; j0 += JG) {
            long pixs[JG];
            uint2 yv[NT][JG];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                const int m = mb + 16 * (j0 + jj) + fr;
                const int mm = m < cd.M ? m : 0;
                const int b = mm / HW, rem = mm - b * HW;
                const int i = rem / cd.Wc, w = rem - i * cd.Wc;
                pixs[jj] = ((long)b * a.Hd + cd.ph + (long)a.sdh * i) * a.Wd + cd.pw + (long)a.sdw * w;
                if (mul) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int nb = nb0 + 16 * t + 4 * fq;
                        yv[t][jj] = (m < cd.M && nb < a.Nd) ? *(const uint2*)(a.mul_y + pixs[jj] * a.Nd + nb)
                                                            : make_uint2(0, 0);
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                const int j = j0 + jj;
                const int m = mb + 16 * j + fr;
                if (m >= cd.M) continue;
                bf16_t* drow = a.dst + pixs[jj] * a.Nd;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int nb = nb0 + 16 * t + 4 * fq;
                    if (nb >= a.Nd) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)             // channels beyond the weight rows are zero
                        v[e] = nb + e < a.Nw ? act_apply_k<decltype(KA)::value>(acc[t][j][e] + bias_[t][e], a.act, a.alpha)
                                             : 0.f;
                    if (mul) {
                        const float y4[4] = {__uint_as_float(yv[t][jj].x << 16), __uint_as_float(yv[t][jj].x & 0xffff0000u),
                                             __uint_as_float(yv[t][jj].y << 16), __uint_as_float(yv[t][jj].y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e])) * (y4[e] > 0.f ? 1.f : a.mul_slope);
                    }
                    if (vec_ok) {
                        *(uint2*)(drow + nb) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb + e < a.Nd) drow[nb + e] = f2bf(v[e]);
                    }
                }
            }
        }
    });
}

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const i32x4 const_i32x4;

// accumulator registers per lane -> resident waves per SIMD asked of the register allocator (512 / 3 = 168 VGPRs
// hold a 64 x 64 wave tile's 64 accumulators + 32 fragment registers; 128 accumulators get 256)
constexpr int dma_min_waves(int acc_regs) { return acc_regs <= 64 ? 3 : 2; }

template <int BM, int BN, int WAVES_M, int STAGES>
__global__ __launch_bounds__(256, dma_min_waves((BM / WAVES_M) * (BN / (4 / WAVES_M)) / 64))
void igemm_conv_kernel_dma(const ConvArgs a, const int n_ntiles, const int n_mtiles) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;      // pixels x channels per wave
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int BNL = BN < 64 ? 64 : BN;                   // weight rows in LDS: every wave fills whole pieces
    constexpr int PA = BM / 64, PW = BNL / 64;               // 16-row pieces per wave and stage
    constexpr int P = PA + PW;
    static_assert(WM % 64 == 0 && WN % 16 == 0 && BM % 64 == 0 && STAGES >= 2 && STAGES <= 4, "bad tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // [STAGES][BM][64 B]
    char* sW = smem + STAGES * BM * PITCH1;            // [STAGES][BNL][64 B]

    const ClassDesc cd = a.classes[blockIdx.y];
    const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3;
    const int n_tile = kblk % n_ntiles, m_tile = (kblk / n_ntiles) * 8 + xcd;
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;
    const int scol = m_tile * gridDim.y + blockIdx.y;
    if (m0 >= cd.M) {
        if (a.stats && m_tile < n_mtiles) igemm_stats_zero(a, n0, BN, scol);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    // ---- staging assignment: wave w fills pieces w*PA .. of A and w*PW .. of W; lane l lands in row l/4, slot l%4
    const int prow = lane >> 2;
    const int q = (lane & 3) ^ ((-(lane >> 4)) & 3);          // k-group fetched into this lane's slot
    const int HW = cd.Hc * cd.Wc;
    const int Cs = a.Cs;
    const int minh = ((cd.pad >> 1) & 127) - 64, minw = ((cd.pad >> 8) & 127) - 64;
    const int rw = (cd.pad >> 15) & 63, rh = (cd.pad >> 21) & 63;
    unsigned vmask[PA], rowoff[PA], wrow[PW];
#pragma unroll
    for (int r = 0; r < PA; ++r) {
        const int m = m0 + (wave * PA + r) * 16 + prow;
        const bool v = m < cd.M;
        const int mm = v ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        const int i = rem / cd.Wc, j = rem - i * cd.Wc;
        const int ih = v ? i * a.ssh : -(1 << 24), iw = j * a.ssw;
        unsigned wmk = 0, m2 = 0;
        for (int wb = 0; wb < rw; ++wb) wmk |= ((unsigned)(iw + minw + wb) < (unsigned)a.Ws ? 1u : 0u) << wb;
        for (int hb = 0; hb < rh; ++hb) m2 |= ((unsigned)(ih + minh + hb) < (unsigned)a.Hs ? wmk : 0u) << (hb * rw);
        vmask[r] = m2;
        rowoff[r] = (unsigned)(b * a.Hs * a.Ws * Cs + (ih * a.Ws + iw) * Cs + q * 8) * 2u;
    }
#pragma unroll
    for (int r = 0; r < PW; ++r) {
        const int nl = (wave * PW + r) * 16 + prow, n = n0 + nl;
        wrow[r] = (nl < BN && n < a.Nw) ? (unsigned)(n * a.Ttot * Cs + q * 8) * 2u : 0xF0000000u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((unsigned)a.B * a.Hs * a.Ws * Cs * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wgt, 0, (int)((unsigned)a.Nw * a.Ttot * Cs * 2u), 0x00020000);

    // The class's tap table sits in LDS behind the stage buffers and the current entry is kept in SGPRs.  It is
    // re-read (one broadcast ds_read + readfirstlane, once per tap) at the END of a K step: a scalar load anywhere
    // in the loop would be outstanding on the LGKM counter while the fragment reads are issued, and with scalar
    // memory's out-of-order return the compiler then waits lgkmcnt(0) before the first MFMA instead of only for the
    // fragments that MFMA consumes.  (An ordinary global load of the table is worse still: beside the LDS-DMA, which
    // the compiler treats as a store to unknown memory, it is not provably invariant, becomes a VECTOR load and
    // drags a vmcnt(0) into every K step.)
    const int4* sT = (const int4*)(smem + STAGES * (BM + BNL) * PITCH1);
    for (int i = tid; i <= cd.ntaps; i += 256) ((int4*)sT)[i] = a.taps[cd.tap_off + i];   // + the spare entry
    __syncthreads();
    const int last_tap = max(cd.ntaps - 1, 0);
    const int nsteps = cd.ntaps * (Cs >> 5);
    int ut_tap = 0, ut_c = 0;
    int te_x, te_y, te_z, te_w;
    auto load_te = [&]() {
        const int4 v = sT[min(ut_tap, last_tap)];
        te_x = __builtin_amdgcn_readfirstlane(v.x); te_y = __builtin_amdgcn_readfirstlane(v.y);
        te_z = __builtin_amdgcn_readfirstlane(v.z); te_w = __builtin_amdgcn_readfirstlane(v.w);
    };
    load_te();
    auto issue = [&](int buf) {
        const bool tv = ut_tap < cd.ntaps;
        const unsigned soffA = (unsigned)(((te_x * a.Ws + te_y) * Cs + ut_c) * 2);
        const unsigned soffW = (unsigned)((te_z * Cs + ut_c) * 2);
        const unsigned bit = tv ? 1u << te_w : 0u;
#pragma unroll
        for (int r = 0; r < PA; ++r) {
            const unsigned off = (vmask[r] & bit) ? rowoff[r] + soffA : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsA, (lds_void*)(sA + ((buf * BM + (wave * PA + r) * 16) * PITCH1)), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < PW; ++r) {
            const unsigned off = tv ? wrow[r] + soffW : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsW, (lds_void*)(sW + ((buf * BNL + (wave * PW + r) * 16) * PITCH1)), 16, off, 0, 0, 0);
        }
    };
    auto next_stage = [&]() {            // K position of the next issue
        ut_c += 32;
        if (ut_c >= Cs) {
            ut_c = 0;
            ++ut_tap;
            load_te();
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: STAGES - 1 stages in flight (stages beyond the reduction fetch zeros: the counts below stay uniform)
#pragma unroll
    for (int d = 0; d < STAGES - 1; ++d) { issue(d); next_stage(); }

    const int fr = lane & 15, fq = lane >> 4;
    const int fqsw = (fq ^ ((-(fr >> 2)) & 3)) * 16;
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
        // stage s has landed once at most (STAGES - 2) younger stages of this wave are still in flight; the barrier
        // makes every wave's share visible and retires the reads of the buffer refilled below
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 2) * P) : "memory");
        {
            int nb = cur + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            issue(nb);
        }
        bf16x8_t wf[NT], xf[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
            wf[i] = *(const bf16x8_t*)(sW + (cur * BNL + wn * WN + 16 * i + fr) * PITCH1 + fqsw);
#pragma unroll
        for (int j = 0; j < MT; ++j)
            xf[j] = *(const bf16x8_t*)(sA + (cur * BM + wm * WM + 16 * j + fr) * PITCH1 + fqsw);
        // all fragment reads in flight before the first MFMA (left alone the scheduler reuses one weight register
        // quad and serialises a read's latency with every group of MT MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);          // the MFMA cluster keeps the pipe over the other block's issue work
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        next_stage();
        if (++cur == STAGES) cur = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // trailing zero-fill stages: nothing may land after exit

    if (a.stats)                         // uniform
        igemm_stats_pass<NT, MT, true>(a, cd, acc, m0 + wm * WM, wn * WN, n0, BN, wm, WAVES_M, lane, (float*)smem, scol);
    igemm_epilogue<NT, MT>(a, cd, acc, m0 + wm * WM, n0 + wn * WN, lane);
}

// out[m][n] = act(bias[n] + sum_s slab[s][m][n]) for the split-K path (dst pixel index == m: single class, unit stride)
__global__ void splitk_finish_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                     bf16_t* __restrict__ dst, long MN, int Nd, int Nw, int ksplit, int act, float alpha) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < MN; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i % Nd);
        float s = 0.f;
        int k = 0;
        for (; k + 7 < ksplit; k += 8) {               // eight independent loads per trip, summed in slab order
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = slab[(long)(k + u) * MN + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v8[u];
        }
        for (; k < ksplit; ++k) s += slab[(long)k * MN + i];
        float v = 0.f;
        if (n < Nw) v = act_apply(s + (bias ? bias[n] : 0.f), act, alpha);
        dst[i] = f2bf(v);
    }
}

template <int BM, int BN, bool DEEP = true>
static int launch_igemm(const ConvArgs& a, int n_classes, int max_M, hipStream_t s) {
    size_t smem = 2 * (BM + BN) * (DEEP ? PITCH : PITCH1) + (size_t)a.max_taps * sizeof(int4);
    auto kern = DEEP ? igemm_conv_kernel<BM, BN> : igemm_conv_kernel_k32<BM, BN>;
    if (smem > 160 * 1024) return -2;
    if (smem > 64 * 1024) {
        static bool done = false;   // per-instantiation opt-in for > 64 KiB dynamic LDS
        if (!done) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    dim3 grid((max_M + BM - 1) / BM, a.ksplit > 1 ? a.ksplit : n_classes, (a.Nd + BN - 1) / BN);
    kern<<<grid, 256, smem, s>>>(a);
    if (a.ksplit > 1) {
        long MN = (long)max_M * a.Nd;
        long g = (MN + 255) / 256;
        splitk_finish_kernel<<<(int)(g > 2048 ? 2048 : g), 256, 0, s>>>(a.slab, a.bias, a.dst, MN, a.Nd, a.Nw, a.ksplit,
                                                                    a.act, a.alpha);
    }
    return 0;
}

template <int BM, int BN, int WAVES_M, int STAGES>
static int launch_igemm_dma(const ConvArgs& a, int n_classes, int max_M, hipStream_t s) {
    // needs the uniform-tap tables (every class descriptor carries its offset box: the caller vouches for that),
    // whole 32-channel K stages and no split-K
    if (a.ksplit > 1 || (a.Cs & 31)) return -6;
    constexpr int BNL = BN < 64 ? 64 : BN;
    const size_t smem = (size_t)STAGES * (BM + BNL) * PITCH1 + ((size_t)a.max_taps + 1) * sizeof(int4);
    auto kern = igemm_conv_kernel_dma<BM, BN, WAVES_M, STAGES>;
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    const int mt = (max_M + BM - 1) / BM, nt = (a.Nd + BN - 1) / BN;
    dim3 grid((unsigned)((mt + 7) / 8 * 8 * nt), n_classes, 1);
    kern<<<grid, 256, smem, s>>>(a, nt, mt);
    return 0;
}

// tile: 0 = auto.  Returns 0 on success, negative on a host-side shape/argument error (nothing
// is launched in that case).
DDL_API int ddl_igemm_conv(const void* src, const void* wgt, const float* bias, void* dst, const void* classes,
                           const void* taps, int n_classes, int max_M, int max_taps, int B, int Hs, int Ws,
                           int Cs, int Hd, int Wd, int Nd, int Nw, int ssh, int ssw, int sdh, int sdw, int Ttot,
                           int act, float alpha, int tile, int ksplit, float* slab, const void* mul_y,
                           float mul_slope, float* stats, int stats_stride, int stats_off, hipStream_t s) {
    if (Cs % 8 != 0 || Cs <= 0 || Nd <= 0 || n_classes <= 0 || max_M <= 0) return -1;
    if ((long)B * Hs * Ws * Cs >= (1L << 31) || (long)B * Hd * Wd * Nd >= (1L << 31) ||
        (long)Nw * Ttot * Cs >= (1L << 31)) return -3;
    ConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.dst = (bf16_t*)dst;
    a.classes = (const ClassDesc*)classes; a.taps = (const int4*)taps;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cs = Cs; a.Hd = Hd; a.Wd = Wd; a.Nd = Nd; a.Nw = Nw;
    a.ssh = ssh; a.ssw = ssw; a.sdh = sdh; a.sdw = sdw; a.Ttot = Ttot; a.act = act; a.alpha = alpha;
    a.max_taps = max_taps;
    a.mul_y = (ksplit > 1 || (Nd & 3)) ? nullptr : (const bf16_t*)mul_y; a.mul_slope = mul_slope;
    if (mul_y && !a.mul_y) return -5;       // the caller runs the separate pass
    a.stats = stats; a.stats_stride = stats_stride; a.stats_off = stats_off;
    if (stats && (mul_y || ksplit > 1 || stats_stride <= 0 || stats_off < 0)) return -7;
    a.ksplit = 1; a.slab = nullptr;
    if (ksplit > 1) {
        // split-K needs one class whose destination pixel index equals the row index (fprop, unit dst stride,
        // whole output = [max_M][Nd]) -- Dense / 1x1 fprop
        if (n_classes != 1 || sdh != 1 || sdw != 1 || !slab || (long)B * Hd * Wd != max_M || (Nd & 3)) return -5;
        a.ksplit = ksplit; a.slab = slab;
    }
    if (tile == 0) {
        // widest channel tile that is not mostly padding; shrink the pixel tile when the grid
        // would otherwise leave CUs idle (256 CUs, want >= ~2 blocks each)
        long blocks256 = (long)((max_M + 255) / 256) * n_classes;
        if (Nd <= 16) tile = 1;
        else if (Nd <= 32) tile = 3;
        else if (Nd <= 64) tile = blocks256 >= 512 ? 5 : 4;
        else tile = ((long)((max_M + 127) / 128) * n_classes * ((Nd + 127) / 128) >= 512) ? 6 : 8;
    }
    switch (tile) {
        case 1: return launch_igemm<256, 16>(a, n_classes, max_M, s);
        case 2: return launch_igemm<64, 64>(a, n_classes, max_M, s);     // WN = 16
        case 3: return launch_igemm<256, 32>(a, n_classes, max_M, s);
        case 4: return launch_igemm<128, 64>(a, n_classes, max_M, s);    // WN = 32
        case 5: return launch_igemm<256, 64>(a, n_classes, max_M, s);
        case 6: return launch_igemm<128, 128>(a, n_classes, max_M, s);   // WN = 64
        case 8: return launch_igemm<64, 128>(a, n_classes, max_M, s);    // WN = 32
        case 17: return launch_igemm<256, 16, false>(a, n_classes, max_M, s);      // 16 + t: shallow variant
        case 18: return launch_igemm<64, 64, false>(a, n_classes, max_M, s);
        case 19: return launch_igemm<256, 32, false>(a, n_classes, max_M, s);
        case 20: return launch_igemm<128, 64, false>(a, n_classes, max_M, s);
        case 21: return launch_igemm<256, 64, false>(a, n_classes, max_M, s);
        case 22: return launch_igemm<128, 128, false>(a, n_classes, max_M, s);
        case 24: return launch_igemm<64, 128, false>(a, n_classes, max_M, s);
        // 32 + : LDS-DMA staging (uniform-tap tables required)
        case 33: return launch_igemm_dma<128, 128, 2, 3>(a, n_classes, max_M, s);   // wave tile 64 x 64
        case 34: return launch_igemm_dma<256, 128, 2, 3>(a, n_classes, max_M, s);   // 128 x 64
        case 35: return launch_igemm_dma<128, 64, 2, 3>(a, n_classes, max_M, s);    // 64 x 32
        case 36: return launch_igemm_dma<256, 64, 4, 3>(a, n_classes, max_M, s);    // 64 x 64
        case 37: return launch_igemm_dma<256, 64, 2, 3>(a, n_classes, max_M, s);    // 128 x 32
        case 38: return launch_igemm_dma<128, 256, 2, 3>(a, n_classes, max_M, s);   // 64 x 128
        case 39: return launch_igemm_dma<256, 32, 4, 3>(a, n_classes, max_M, s);    // 64 x 32
        case 40: return launch_igemm_dma<128, 128, 2, 4>(a, n_classes, max_M, s);
        case 41: return launch_igemm_dma<256, 128, 2, 4>(a, n_classes, max_M, s);
        default: return -4;
    }
}
