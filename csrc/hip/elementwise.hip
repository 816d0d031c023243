// Memory-bound kernels: activations, BatchNorm (stats / apply / backward), bias gradient, casts,
// weight relayout, channel padding, dataset gather.  All tensors are [rows][C] with C a
// multiple of 8 so every access is a 16-byte vector (8 x bf16) per lane.
#include "common.h"
#include <cstdlib>

#ifndef CR_UNROLL
#define CR_UNROLL 4          // rows in flight per lane in colreduce (8 was measured 3-9 % slower per step: registers)
#endif
#define EW_BLOCK 256
#define EW_MAX_GRID 2048
#ifndef COLREDUCE_MAX_BLOCKS
#define COLREDUCE_MAX_BLOCKS 1024      // 4 blocks per CU: enough loads in flight to stream from HBM / Infinity Cache
#endif

static inline int ew_grid(long n_groups) {
    long g = (n_groups + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(g < 1 ? 1 : (g > EW_MAX_GRID ? EW_MAX_GRID : g));
}

// ------------------------------------------------------------------------------ activations
template <int A>
__global__ void act_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long n8, int cg,
                               int c_true, int kind, float alpha) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(x[i], f);
        int c0 = (int)(i % cg) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (c0 + j < c_true) ? act_apply_k<A>(f[j], kind, alpha) : 0.f;
        y[i] = pack8(f);
    }
}

template <int A>
__global__ void act_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y,
                               uint4* __restrict__ dx, long n8, int kind, float alpha) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float g[8], v[8];
        unpack8(dy[i], g);
        unpack8(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] *= act_grad_y_k<A>(v[j], kind, alpha);
        dx[i] = pack8(g);
    }
}

DDL_API void ddl_act_fwd(const void* x, void* y, long rows, int C, int c_true, int kind, float alpha,
                         hipStream_t s) {
    long n8 = rows * (C / 8);
#define CALL(A) act_fwd_kernel<A><<<ew_grid(n8), EW_BLOCK, 0, s>>>((const uint4*)x, (uint4*)y, n8, C / 8, c_true, kind, alpha)
    ACT1_DISPATCH(kind, CALL);
#undef CALL
}

DDL_API void ddl_act_bwd(const void* dy, const void* y, void* dx, long n, int kind, float alpha,
                         hipStream_t s) {
    long n8 = n / 8;
#define CALL(A) act_bwd_kernel<A><<<ew_grid(n8), EW_BLOCK, 0, s>>>((const uint4*)dy, (const uint4*)y, (uint4*)dx, n8, kind, alpha)
    ACT1_DISPATCH(kind, CALL);
#undef CALL
}

// ------------------------------------------------------------------------------ column reductions
// One kernel shape for the per-channel reductions over rows:
//   MODE 0 (stats):   s0 = sum x          s1 = sum x^2
//   MODE 1 (bn bwd):  s0 = sum g          s1 = sum g * xhat      g = dy * act'(y)
//   MODE 2 (colsum):  s0 = sum dy
//   MODE 3 (bn bwd + bias grad of the conv in front of the BN): MODE 1 plus, with a = prev_act'(x),
//                     s2 = sum a * g     s3 = sum a * x          s4 = sum a
//       bn_bwd_apply writes dx = a * (k0*g + k1*x + k2), so the conv's bias gradient sum_r dx is
//       k0*s2 + k1*s3 + k2*s4: the finalize kernel emits it and no separate pass over dx is needed.
// Block partials go to part[NS][C][block]; a finalize kernel combines them in fp64 (deterministic).
// The row loop issues four rows' loads before it consumes any of them (8-12 independent 16-byte
// loads in flight per lane), which is what a streaming reduction needs at 1-4 blocks per CU.
template <int MODE> struct CrSums { static constexpr int N = MODE == 3 ? 5 : 2; };

template <int MODE, int A = -1, int P = -1>
__global__ __launch_bounds__(EW_BLOCK) void colreduce_kernel(
    const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ yv,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ part,
    long rows, int C, int act, float alpha, int prev_act, float prev_alpha) {
    constexpr int NS = CrSums<MODE>::N;
    __shared__ float red[EW_BLOCK][NS * 8 + 1];
    // batch groups (blockIdx.z): `rows` rows each, statistics per group (a discriminator batch of
    // [real ; fake] normalised half by half); partial index = z * gridDim.x + blockIdx.x
    const int cg = C / 8;
    {
        const size_t goff = (size_t)blockIdx.z * rows * cg;
        a += goff;
        if (MODE == 1 || MODE == 3) { b += goff; if (yv) yv += goff; mean += blockIdx.z * C; invstd += blockIdx.z * C; }
    }
    const int cgl = cg < EW_BLOCK ? cg : EW_BLOCK;           // channel groups handled by this block column
    const int rpi = EW_BLOCK / cgl;                            // rows per block iteration
    const int g = blockIdx.y * EW_BLOCK + (threadIdx.x % cgl);
    const int rl = threadIdx.x / cgl;
    float s[NS][8];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
    const bool active = rl < rpi && g < cg;
    if (active) {
        float mu[8], is[8];
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { mu[j] = mean[g * 8 + j]; is[j] = invstd[g * 8 + j]; }
        }
        const bool has_y = (MODE == 1 || MODE == 3) && (A >= 0 ? A : act) != ACT_LINEAR;
        auto accum = [&](const uint4& ua, const uint4& ub, const uint4& uy) {
            float v[8];
            unpack8(ua, v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[0][j] += v[j]; s[1][j] += v[j] * v[j]; }
            } else if (MODE == 1 || MODE == 3) {
                float x[8], yy[8];
                unpack8(ub, x);
                if (has_y) {
                    unpack8(uy, yy);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= act_grad_y_k<A>(yy[j], act, alpha);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[0][j] += v[j]; s[1][j] += v[j] * (x[j] - mu[j]) * is[j]; }
                if (MODE == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float ad = act_grad_y_k<P>(x[j], prev_act, prev_alpha);
                        s[NS - 3][j] += ad * v[j];
                        s[NS - 2][j] += ad * x[j];
                        s[NS - 1][j] += ad;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) s[0][j] += v[j];
            }
        };
        const long stride = (long)gridDim.x * rpi;
        long r = blockIdx.x * (long)rpi + rl;
        constexpr int CRU = CR_UNROLL;
        for (; r + (CRU - 1) * stride < rows; r += CRU * stride) {
            uint4 ua[CRU], ub[CRU], uy[CRU];
#pragma unroll
            for (int u = 0; u < CRU; ++u) {
                const long i = (r + u * stride) * cg + g;
                ua[u] = a[i];
                if (MODE == 1 || MODE == 3) {
                    ub[u] = b[i];
                    if (has_y) uy[u] = yv[i];
                }
            }
#pragma unroll
            for (int u = 0; u < CRU; ++u) accum(ua[u], ub[u], uy[u]);
        }
        for (; r < rows; r += stride) {
            const long i = r * cg + g;
            uint4 ua = a[i], ub = ua, uy = ua;
            if (MODE == 1 || MODE == 3) {
                ub = b[i];
                if (has_y) uy = yv[i];
            }
            accum(ua, ub, uy);
        }
    }
    if (cgl <= 64 && (cgl & (cgl - 1)) == 0) {
        // Lanes l, l + cgl, l + 2 cgl, ... of a wave hold different rows of the same channel group: fold them with
        // shuffles, then the four waves through LDS, and let every thread combine and store one (group, sum, channel)
        // element.  (The generic path below -- 40 LDS stores per thread, a barrier per tree level, 8 threads doing
        // all the scattered partial stores -- was 10-25 % of these 15-30 us kernels.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int off = cgl; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[k][j] += __shfl_xor(s[k][j], off, 64);
        }
        float* fr = &red[0][0];                                   // [4 waves][cgl][NS * 8]
        if (lane < cgl) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) fr[(wave * cgl + lane) * (NS * 8) + k * 8 + j] = s[k][j];
        }
        __syncthreads();
        const size_t nb = (size_t)gridDim.x * gridDim.z;
        const size_t bi = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        for (int t = threadIdx.x; t < cgl * NS * 8; t += EW_BLOCK) {
            const int gl = t / (NS * 8), e = t - gl * (NS * 8);
            const int gg = blockIdx.y * EW_BLOCK + gl;
            if (gg >= cg) continue;
            float v = fr[t];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += fr[w * cgl * (NS * 8) + t];
            const int k = e >> 3, j = e & 7;
            part[((size_t)k * C + gg * 8 + j) * nb + bi] = v;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][k * 8 + j] = s[k][j];
    __syncthreads();
    // tree reduction over the row-lane dimension (rpi lanes share one channel group)
    int half = 1;
    while (half < rpi) half <<= 1;
    for (half >>= 1; half > 0; half >>= 1) {
        if (rl < half && rl + half < rpi) {
#pragma unroll
            for (int j = 0; j < NS * 8; ++j) red[threadIdx.x][j] += red[threadIdx.x + half * cgl][j];
        }
        __syncthreads();
    }
    if (rl == 0 && g < cg) {
        // part[k][c][block]: the block index is the fastest dimension, so the finalize kernel's lanes read
        // consecutive floats (with [block][k][c] every lane touched its own 64-byte line: 35 us for 1024 x 5 x 128)
        float* p = part + (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        const size_t nb = (size_t)gridDim.x * gridDim.z;
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) p[((size_t)k * C + g * 8 + j) * nb] = red[threadIdx.x][k * 8 + j];
    }
}

// Sum block partials part[k][c][b] for channel c.  256 threads = 8 channels x 32 partial lanes: lane
// `pl` adds partials pl, pl+32, ... (independent loads, fixed order), then the 32 lane sums are
// combined by a fixed-order shuffle tree inside the half-wave -> deterministic, and short enough
// (<= 8 loads per lane for 256 partials) that the kernel is a few microseconds instead of a
// latency-bound serial walk.  Result valid where pl == 0.
#define FIN_BLOCK 256
#define FIN_CH 8
template <int NS, int NUSE>
__device__ __forceinline__ void reduce_partials_n(const float* __restrict__ part, int nblocks, int C, int c,
                                                  double (&out)[NUSE], int nb_total = 0, int b0 = 0) {
    // partial row of (k, c): part[(k * C + c) * nb_total + b0 + b], b < nblocks (one batch group's blocks)
    if (nb_total == 0) nb_total = nblocks;
    part += b0;
    const int pl = threadIdx.x & 31;
    double acc[NUSE];
#pragma unroll
    for (int k = 0; k < NUSE; ++k) acc[k] = 0;
    if (c < C) {
        // four partial rows per trip: their loads are independent and issued together (a dependent
        // one-row-per-trip walk made this kernel 5-40 us of pure load latency); summation order is fixed
        int b = pl;
        for (; b + 96 < nblocks; b += 128) {
            float v[4][NUSE];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NUSE; ++k) v[u][k] = part[((size_t)k * C + c) * nb_total + b + 32 * u];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NUSE; ++k) acc[k] += v[u][k];
        }
        for (; b < nblocks; b += 32) {
#pragma unroll
            for (int k = 0; k < NUSE; ++k) acc[k] += part[((size_t)k * C + c) * nb_total + b];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < NUSE; ++k) acc[k] += __shfl_down(acc[k], o, 32);
    }
#pragma unroll
    for (int k = 0; k < NUSE; ++k) out[k] = acc[k];
}
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int nblocks, int C, int c,
                                                bool two, double& s0, double& s1, int nb_total = 0, int b0 = 0) {
    if (two) {
        double o[2];
        reduce_partials_n<2, 2>(part, nblocks, C, c, o, nb_total, b0);
        s0 = o[0]; s1 = o[1];
    } else {
        double o[1];
        reduce_partials_n<2, 1>(part, nblocks, C, c, o, nb_total, b0);
        s0 = o[0]; s1 = 0;
    }
}
#define FIN_CHANNEL() (blockIdx.x * FIN_CH + (threadIdx.x >> 5))
#define FIN_LEADER() ((threadIdx.x & 31) == 0)
#define FIN_GRID(C) (((C) + FIN_CH - 1) / FIN_CH)
// partial rows per channel above which the finalize kernels run one block per channel (DDL_FIN_WIDE_MIN)
static int fin_wide_min() {
    static const int v = getenv("DDL_FIN_WIDE_MIN") ? atoi(getenv("DDL_FIN_WIDE_MIN")) : 256;
    return v;
}

static inline void colreduce_grid(long rows, int C, dim3* grid) {
    int cg = C / 8;
    int cgl = cg < EW_BLOCK ? cg : EW_BLOCK;
    int rpi = EW_BLOCK / cgl;
    // cache-resident tensors (< 64 MB): two blocks per CU (measured on the DCGAN28 step: 256 -> 512 blocks +1 %,
    // 1024 -2.5 %); larger ones stream from HBM and need four blocks per CU to keep enough loads in flight
    // (measured: classifier step -6 %, dcgan64 -7 %)
    static const long small_cap = getenv("DDL_CR_SMALL_CAP") ? atol(getenv("DDL_CR_SMALL_CAP")) : 512;
    static const long small_rows = getenv("DDL_CR_SMALL_ROWS") ? atol(getenv("DDL_CR_SMALL_ROWS")) : 8;
    const bool big = rows * (long)C * 2 >= (64L << 20);
    const long per_thread = big ? 4 : small_rows, cap = big ? COLREDUCE_MAX_BLOCKS : small_cap;
    long nb = (rows + (long)rpi * per_thread - 1) / ((long)rpi * per_thread);
    if (nb < 1) nb = 1;
    if (nb > cap) nb = cap;
    *grid = dim3((unsigned)nb, (unsigned)((cg + EW_BLOCK - 1) / EW_BLOCK), 1);
}

DDL_API int ddl_colreduce_blocks(long rows, int C) {
    dim3 g;
    colreduce_grid(rows, C, &g);
    return (int)g.x;
}

// BN forward statistics -> per-channel affine (scale, shift), saved (mean, invstd), moving update.
// The channel's parameters are loaded by the caller BEFORE it walks the partials (BnChanParams): these kernels are a
// chain of dependent round trips, and the parameter fetch used to be one more at the very end.
struct BnChanParams { float gamma, beta, mm, mv; };
__device__ __forceinline__ BnChanParams bn_chan_params(int c, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mm,
                                                       const float* __restrict__ mv) {
    BnChanParams p{0.f, 0.f, 0.f, 0.f};
    if (c < C) { p.gamma = gamma[c]; p.beta = beta[c]; p.mm = mm[c]; p.mv = mv[c]; }
    return p;
}
__device__ __forceinline__ void bn_stats_emit(int c, double s0, double s1, long rows, BnChanParams& pr,
                                              float* __restrict__ mm,
                                              float* __restrict__ mv, float momentum, float eps, int bessel,
                                              int update_moving, float* __restrict__ scale, float* __restrict__ shift,
                                              float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    double mean = s0 / (double)rows;
    double var = s1 / (double)rows - mean * mean;
    if (var < 0) var = 0;
    float is = rsqrtf((float)var + eps);
    float sc = pr.gamma * is;
    scale[c] = sc;
    shift[c] = pr.beta - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = is;
    if (update_moving) {
        double v = bessel && rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var;
        pr.mm = pr.mm * momentum + (float)mean * (1.f - momentum);      // (groups: one update per group, in order)
        pr.mv = pr.mv * momentum + (float)v * (1.f - momentum);
        mm[c] = pr.mm;
        mv[c] = pr.mv;
    }
}

__global__ void bn_stats_finalize_kernel(const float* __restrict__ part, int nblocks, int C, long rows,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ mm, float* __restrict__ mv, float momentum,
                                         float eps, int bessel, int update_moving, float* __restrict__ scale,
                                         float* __restrict__ shift, float* __restrict__ mean_out,
                                         float* __restrict__ invstd_out, int groups) {
    // `nblocks` partials and `rows` rows PER batch group; outputs are [groups][C]; the moving statistics take one
    // momentum update per group, in group order (as calling the layer once per half would)
    int c = FIN_CHANNEL();
    BnChanParams pr = bn_chan_params(c, C, gamma, beta, mm, mv);
    for (int z = 0; z < groups; ++z) {
        double s0, s1;
        reduce_partials(part, nblocks, C, c, true, s0, s1, nblocks * groups, z * nblocks);
        if (c < C && FIN_LEADER())
            bn_stats_emit(c, s0, s1, rows, pr, mm, mv, momentum, eps, bessel, update_moving, scale + z * C,
                          shift + z * C, mean_out + z * C, invstd_out + z * C);
    }
}

// Many partials per channel (a producer kernel with thousands of blocks): one block per channel, 256 lanes walk
// the channel's contiguous partial row (4 independent loads per trip), fixed-order tree in fp64 -> deterministic.
__global__ __launch_bounds__(256) void bn_stats_finalize_wide_kernel(
    const float* __restrict__ part, int nblocks, int C, long rows, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
    int bessel, int update_moving, float* __restrict__ scale, float* __restrict__ shift,
    float* __restrict__ mean_out, float* __restrict__ invstd_out, int groups) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    const size_t nbt = (size_t)nblocks * groups;
    BnChanParams pr = bn_chan_params(c, C, gamma, beta, mm, mv);
    for (int z = 0; z < groups; ++z) {
        const float* p0 = part + (size_t)c * nbt + (size_t)z * nblocks;
        const float* p1 = part + ((size_t)C + c) * nbt + (size_t)z * nblocks;
        double a0 = 0, a1 = 0;
        int b = threadIdx.x;
        for (; b + 768 < nblocks; b += 1024) {
            float u[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { u[k] = p0[b + 256 * k]; v[k] = p1[b + 256 * k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0 += u[k]; a1 += v[k]; }
        }
        for (; b < nblocks; b += 256) { a0 += p0[b]; a1 += p1[b]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64); }
        if (z) __syncthreads();                      // `red` of the previous group has been consumed
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double s0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
            const double s1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
            bn_stats_emit(c, s0, s1, rows, pr, mm, mv, momentum, eps, bessel, update_moving, scale + z * C,
                          shift + z * C, mean_out + z * C, invstd_out + z * C);
        }
    }
}

// Hundreds of thousands of partials per channel (a patch-resident conv over a batch sized to HBM: one column per
// (image, row band)): one block per channel walks them in ~0.4 ms on 32 of 256 CUs.  Fold them first: block
// (row, group * S + s) sums the s-th of S ranges of one batch group's columns of one partial row (fp64 inside, fixed
// order -> deterministic) into out[row][group * S + s]; the ordinary finalize then reads `S` partials per group.
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ part, int nbg, int groups, int S,
                                                            float* __restrict__ out) {
    __shared__ double red[4];
    const int row = blockIdx.x, z = blockIdx.y / S, sidx = blockIdx.y - z * S;
    const int len = (nbg + S - 1) / S;
    const int b0 = sidx * len, b1 = min(nbg, b0 + len);
    const float* p = part + ((size_t)row * groups + z) * nbg;
    double a0 = 0;
    int b = b0 + threadIdx.x;
    for (; b + 768 < b1; b += 1024) {
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = p[b + 256 * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a0 += u[k];
    }
    for (; b < b1; b += 256) a0 += p[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a0 += __shfl_down(a0, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a0;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)row * groups * S + blockIdx.y] = (float)(((red[0] + red[1]) + red[2]) + red[3]);
}

// rows = partial rows (2 * C for forward statistics); part[row][groups][nbg] -> out[row][groups][S]
DDL_API void ddl_fold_partials(const float* part, int rows, int nbg, int groups, int S, float* out, hipStream_t s) {
    fold_partials_kernel<<<dim3(rows, groups * S), 256, 0, s>>>(part, nbg, groups, S, out);
}

// `groups` batch groups of `rows` rows (and `nblocks` partials) EACH; outputs [groups][C].
DDL_API void ddl_bn_stats_from_part(const float* part, int nblocks, long rows, int C, const float* gamma,
                                    const float* beta, float* mm, float* mv, float momentum, float eps, int bessel,
                                    int update_moving, float* scale, float* shift, float* mean, float* invstd,
                                    int groups, hipStream_t s);

DDL_API void ddl_bn_batch_stats(const void* x, long rows, int C, float* part, const float* gamma,
                                const float* beta, float* mm, float* mv, float momentum, float eps, int bessel,
                                int update_moving, float* scale, float* shift, float* mean, float* invstd,
                                int groups, hipStream_t s) {
    dim3 grid;
    colreduce_grid(rows, C, &grid);
    grid.z = groups;
    colreduce_kernel<0><<<grid, EW_BLOCK, 0, s>>>((const uint4*)x, nullptr, nullptr, nullptr, nullptr, part,
                                                  rows, C, 0, 0.f, 0, 0.f);
    ddl_bn_stats_from_part(part, (int)grid.x, rows, C, gamma, beta, mm, mv, momentum, eps, bessel, update_moving, scale,
                           shift, mean, invstd, groups, s);
}

// Batch statistics from partials a producer kernel already wrote (pconv's STATS epilogue): finalize only.
DDL_API void ddl_bn_stats_from_part(const float* part, int nblocks, long rows, int C, const float* gamma,
                                    const float* beta, float* mm, float* mv, float momentum, float eps, int bessel,
                                    int update_moving, float* scale, float* shift, float* mean, float* invstd,
                                    int groups, hipStream_t s) {
    // one block per channel once a 32-lane walk would take more than two trips of four loads: the narrow
    // kernel's time is its number of dependent round trips (8 of them for 1000 partials)
    if (nblocks > fin_wide_min())
        bn_stats_finalize_wide_kernel<<<C, 256, 0, s>>>(part, nblocks, C, rows, gamma, beta, mm, mv, momentum, eps,
                                                        bessel, update_moving, scale, shift, mean, invstd, groups);
    else
        bn_stats_finalize_kernel<<<FIN_GRID(C), FIN_BLOCK, 0, s>>>(part, nblocks, C, rows, gamma, beta, mm, mv,
                                                                 momentum, eps, bessel, update_moving, scale, shift,
                                                                 mean, invstd, groups);
}

// Inference-phase BN: scale/shift from the moving statistics (tiny, one launch).
__global__ void bn_infer_affine_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                       float eps, int C, float* scale, float* shift, float* invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float is = rsqrtf(mv[c] + eps);
    float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
    invstd[c] = is;
}

DDL_API void ddl_bn_infer_affine(const float* gamma, const float* beta, const float* mm, const float* mv,
                                 float eps, int C, float* scale, float* shift, float* invstd, hipStream_t s) {
    bn_infer_affine_kernel<<<(C + 127) / 128, 128, 0, s>>>(gamma, beta, mm, mv, eps, C, scale, shift, invstd);
}

// (scale, shift) of 8 channels.  With moving statistics given (inference-phase BN), `scale` / `shift`
// are gamma / beta and the affine is derived here: no separate bn_infer_affine launch.
__device__ __forceinline__ void bn_affine8(const float* __restrict__ scale, const float* __restrict__ shift,
                                           const float* __restrict__ mm, const float* __restrict__ mv, float eps,
                                           int c0, float* sc, float* sh) {
    const float4 sa = *(const float4*)(scale + c0), sb = *(const float4*)(scale + c0 + 4);
    const float4 ha = *(const float4*)(shift + c0), hb = *(const float4*)(shift + c0 + 4);
    const float s_[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float h_[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_[j]; sh[j] = h_[j]; }
    if (mm) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float k = sc[j] * rsqrtf(mv[c0 + j] + eps);
            sh[j] = sh[j] - mm[c0 + j] * k;
            sc[j] = k;
        }
    }
}

template <int A>
__global__ void bn_apply_kernel(const uint4* __restrict__ x, uint4* __restrict__ y,
                                const float* __restrict__ scale, const float* __restrict__ shift, long n8,
                                int cg, int c_true, int act, float alpha, const float* __restrict__ mm,
                                const float* __restrict__ mv, float eps) {
    // blockIdx.y = batch group: n8 vectors and one (scale, shift) row each
    x += (size_t)blockIdx.y * n8; y += (size_t)blockIdx.y * n8;
    scale += blockIdx.y * cg * 8; shift += blockIdx.y * cg * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * 8;
        float f[8];
        unpack8(x[i], f);
        float sc[8], sh[8];
        bn_affine8(scale, shift, mm, mv, eps, c0, sc, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            f[j] = (c0 + j < c_true) ? act_apply_k<A>(fmaf(f[j], sc[j], sh[j]), act, alpha) : 0.f;
        y[i] = pack8(f);
    }
}

// Channel-stationary variant (cg divides 256): a thread keeps ONE 8-channel group for its whole
// life, so scale/shift are loaded once and the loop body is load -> 8 fma/act -> store.
template <int A>
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_cs_kernel(const uint4* __restrict__ x, uint4* __restrict__ y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, long rows, int cg,
                                                               int c_true, int act, float alpha,
                                                               const float* __restrict__ mm,
                                                               const float* __restrict__ mv, float eps) {
    const int g = threadIdx.x % cg, rl = threadIdx.x / cg, rpi = EW_BLOCK / cg;
    const int c0 = g * 8;
    x += (size_t)blockIdx.y * rows * cg; y += (size_t)blockIdx.y * rows * cg;      // blockIdx.y = batch group
    scale += blockIdx.y * cg * 8; shift += blockIdx.y * cg * 8;
    float sc[8], sh[8];
    bn_affine8(scale, shift, mm, mv, eps, c0, sc, sh);
    for (long r = blockIdx.x * (long)rpi + rl; r < rows; r += (long)gridDim.x * rpi) {
        float f[8];
        unpack8(x[r * cg + g], f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            f[j] = (c0 + j < c_true) ? act_apply_k<A>(fmaf(f[j], sc[j], sh[j]), act, alpha) : 0.f;
        y[r * cg + g] = pack8(f);
    }
}

static inline bool cs_ok(int cg) { return cg <= EW_BLOCK && EW_BLOCK % cg == 0; }
static inline int cs_grid(long rows, int cg) {
    long rpi = EW_BLOCK / cg;
    long g = (rows + rpi * 4 - 1) / (rpi * 4);          // >= 4 rows per thread
    return (int)(g < 1 ? 1 : (g > EW_MAX_GRID ? EW_MAX_GRID : g));
}

static void bn_apply_launch(const void* x, void* y, const float* scale, const float* shift, long rows, int C,
                            int c_true, int act, float alpha, const float* mm, const float* mv, float eps,
                            int groups, hipStream_t s) {
    long n8 = rows * (C / 8);                  // `rows` per batch group
    if (cs_ok(C / 8)) {
        int gx = cs_grid(rows * groups, C / 8) / groups;
#define CALL(A) bn_apply_cs_kernel<A><<<dim3(gx < 1 ? 1 : gx, groups), EW_BLOCK, 0, s>>>( \
        (const uint4*)x, (uint4*)y, scale, shift, rows, C / 8, c_true, act, alpha, mm, mv, eps)
        ACT1_DISPATCH(act, CALL);
#undef CALL
        return;
    }
    int gx = ew_grid(n8 * groups) / groups;
#define CALL(A) bn_apply_kernel<A><<<dim3(gx < 1 ? 1 : gx, groups), EW_BLOCK, 0, s>>>( \
    (const uint4*)x, (uint4*)y, scale, shift, n8, C / 8, c_true, act, alpha, mm, mv, eps)
    ACT1_DISPATCH(act, CALL);
#undef CALL
}

// `groups` batch groups of `rows` rows each, scale / shift [groups][C]
DDL_API void ddl_bn_apply(const void* x, void* y, const float* scale, const float* shift, long rows, int C,
                          int c_true, int act, float alpha, int groups, hipStream_t s) {
    bn_apply_launch(x, y, scale, shift, rows, C, c_true, act, alpha, nullptr, nullptr, 0.f, groups, s);
}

// Inference-phase BN in one launch: y = act(gamma * (x - moving_mean) * rsqrt(moving_var + eps) + beta).
DDL_API void ddl_bn_apply_infer(const void* x, void* y, const float* gamma, const float* beta, const float* mm,
                                const float* mv, float eps, long rows, int C, int c_true, int act, float alpha,
                                hipStream_t s) {
    bn_apply_launch(x, y, gamma, beta, rows, C, c_true, act, alpha, mm, mv, eps, 1, s);
}

// BN backward: reduce -> finalize (dgamma, dbeta, c1 = dbeta/M, c2 = dgamma/M) -> apply.
// Per batch group z: c1 = dbeta_z / M, c2 = dgamma_z / M (M rows per group), rows [z][C].  dgamma / dbeta (and the
// producing conv's bias gradient for NS == 5) are the sums over the groups, accumulated in fp64 in group order.
template <int NS>
struct BnBwdAcc { double dbeta = 0, dgamma = 0, dbias = 0; };

template <int NS>
__device__ __forceinline__ void bn_bwd_group(int c, int z, int C, const double (&o)[NS], long rows, int train_stats,
                                             float* __restrict__ c1, float* __restrict__ c2,
                                             float mean_zc, float invstd_zc, float gamma_c, int c_bias,
                                             BnBwdAcc<NS>& acc) {
    acc.dbeta += o[0];
    acc.dgamma += o[1];
    const float f1 = train_stats ? (float)(o[0] / (double)rows) : 0.f;
    const float f2 = train_stats ? (float)(o[1] / (double)rows) : 0.f;
    c1[z * C + c] = f1;
    c2[z * C + c] = f2;
    if (NS == 5 && c < c_bias) {
        // same k0/k1/k2 as bn_bwd_apply (fp32 there); combined in fp64 here
        // (mean / invstd / gamma were fetched by the caller before it walked the partials)
        const double is = invstd_zc, gi = (double)gamma_c * is;
        const double k0 = gi, k1 = -gi * is * (double)f2,
                     k2 = -gi * ((double)f1 - (double)mean_zc * is * (double)f2);
        acc.dbias += k0 * o[NS - 3] + k1 * o[NS - 2] + k2 * o[NS - 1];
    }
}

template <int NS>
__device__ __forceinline__ void bn_bwd_store(int c, const BnBwdAcc<NS>& acc, float* __restrict__ dgamma,
                                             float* __restrict__ dbeta, float* __restrict__ dbias, int c_bias) {
    if (dbeta) dbeta[c] = (float)acc.dbeta;
    if (dgamma) dgamma[c] = (float)acc.dgamma;
    if (NS == 5 && c < c_bias) dbias[c] = (float)acc.dbias;
}

template <int NS>
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C, long rows,
                                       int train_stats, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ c1, float* __restrict__ c2,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       const float* __restrict__ gamma, float* __restrict__ dbias, int c_bias,
                                       int groups) {
    int c = FIN_CHANNEL();
    BnBwdAcc<NS> acc;
    const bool kb = NS == 5 && c < c_bias && c < C;
    const float gamma_c = kb ? gamma[c] : 0.f;
    for (int z = 0; z < groups; ++z) {
        const float mean_zc = kb ? mean[z * C + c] : 0.f, invstd_zc = kb ? invstd[z * C + c] : 0.f;
        double o[NS];
        reduce_partials_n<NS, NS>(part, nblocks, C, c, o, nblocks * groups, z * nblocks);
        if (c < C && FIN_LEADER())
            bn_bwd_group<NS>(c, z, C, o, rows, train_stats, c1, c2, mean_zc, invstd_zc, gamma_c, c_bias, acc);
    }
    if (c >= C || !FIN_LEADER()) return;
    bn_bwd_store<NS>(c, acc, dgamma, dbeta, dbias, c_bias);
}

// More than a few hundred partials per channel: one block per channel, 256 lanes on the channel's contiguous
// partial rows (the 32-lane walk above is a chain of dependent round trips), fixed-order fp64 tree.
template <int NS>
__global__ __launch_bounds__(256) void bn_bwd_finalize_wide_kernel(
    const float* __restrict__ part, int nblocks, int C, long rows, int train_stats, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, float* __restrict__ dbias, int c_bias,
    int groups) {
    __shared__ double red[NS][4];
    const int c = blockIdx.x;
    const size_t nbt = (size_t)nblocks * groups;
    BnBwdAcc<NS> total;
    const bool kb = NS == 5 && c < c_bias;
    const float gamma_c = kb ? gamma[c] : 0.f;
    for (int z = 0; z < groups; ++z) {
        const float mean_zc = kb ? mean[z * C + c] : 0.f, invstd_zc = kb ? invstd[z * C + c] : 0.f;
        const float* pz = part + (size_t)z * nblocks;
        double acc[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0;
        int b = threadIdx.x;
        for (; b + 256 < nblocks; b += 512) {         // two partial rows per trip, NS sums each: 2 NS loads in flight
            float u[2][NS];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < NS; ++k) u[h][k] = pz[((size_t)k * C + c) * nbt + b + 256 * h];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < NS; ++k) acc[k] += u[h][k];
        }
        for (; b < nblocks; b += 256) {
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] += pz[((size_t)k * C + c) * nbt + b];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] += __shfl_down(acc[k], o, 64);
        }
        if (z) __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) red[k][threadIdx.x >> 6] = acc[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double o[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) o[k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
            bn_bwd_group<NS>(c, z, C, o, rows, train_stats, c1, c2, mean_zc, invstd_zc, gamma_c, c_bias, total);
        }
    }
    if (threadIdx.x == 0) bn_bwd_store<NS>(c, total, dgamma, dbeta, dbias, c_bias);
}

template <int A, int P>
__global__ void bn_bwd_apply_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ x,
                                    const uint4* __restrict__ yv, uint4* __restrict__ dx,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ c1,
                                    const float* __restrict__ c2, long n8, int cg, int act, float alpha,
                                    int prev_act, float prev_alpha) {
    {   // blockIdx.y = batch group
        const size_t go = (size_t)blockIdx.y * n8;
        dy += go; x += go; dx += go; if (yv) yv += go;
        const int co = blockIdx.y * cg * 8;
        mean += co; invstd += co; c1 += co; c2 += co;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * 8;
        float g[8], xv[8], yy[8];
        unpack8(dy[i], g);
        unpack8(x[i], xv);
        if ((A >= 0 ? A : act) != ACT_LINEAR) {
            unpack8(yv[i], yy);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= act_grad_y_k<A>(yy[j], act, alpha);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = c0 + j;
            float is = invstd[c];
            float xhat = (xv[j] - mean[c]) * is;
            float d = gamma[c] * is * (g[j] - c1[c] - xhat * c2[c]);
            if ((P >= 0 ? P : prev_act) != ACT_LINEAR) d *= act_grad_y_k<P>(xv[j], prev_act, prev_alpha);
            g[j] = d;
        }
        dx[i] = pack8(g);
    }
}

template <int A, int P>
__global__ __launch_bounds__(EW_BLOCK) void bn_bwd_apply_cs_kernel(
    const uint4* __restrict__ dy, const uint4* __restrict__ x, const uint4* __restrict__ yv, uint4* __restrict__ dx,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ c1, const float* __restrict__ c2, long rows, int cg, int act, float alpha,
    int prev_act, float prev_alpha) {
    const int g = threadIdx.x % cg, rl = threadIdx.x / cg, rpi = EW_BLOCK / cg;
    {   // blockIdx.y = batch group
        const size_t go = (size_t)blockIdx.y * rows * cg;
        dy += go; x += go; dx += go; if (yv) yv += go;
        const int co = blockIdx.y * cg * 8;
        mean += co; invstd += co; c1 += co; c2 += co;
    }
    // dx = k0*g' + k1*x + k2 with  k0 = gamma*is, k1 = -gamma*is*is*c2, k2 = -gamma*is*(c1 - mean*is*c2)
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int c = g * 8 + j;
        float is = invstd[c], gi = gamma[c] * is;
        k0[j] = gi;
        k1[j] = -gi * is * c2[c];
        k2[j] = -gi * (c1[c] - mean[c] * is * c2[c]);
    }
    for (long r = blockIdx.x * (long)rpi + rl; r < rows; r += (long)gridDim.x * rpi) {
        const long i = r * cg + g;
        float gr[8], xv[8], yy[8];
        unpack8(dy[i], gr);
        unpack8(x[i], xv);
        if ((A >= 0 ? A : act) != ACT_LINEAR) {
            unpack8(yv[i], yy);
#pragma unroll
            for (int j = 0; j < 8; ++j) gr[j] *= act_grad_y_k<A>(yy[j], act, alpha);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = fmaf(k0[j], gr[j], fmaf(k1[j], xv[j], k2[j]));
            if ((P >= 0 ? P : prev_act) != ACT_LINEAR) d *= act_grad_y_k<P>(xv[j], prev_act, prev_alpha);
            gr[j] = d;
        }
        dx[i] = pack8(gr);
    }
}

// `groups` batch groups of `rows` rows each: mean / invstd / c1 / c2 are [groups][C]; dgamma, dbeta, dbias sum over groups.
DDL_API void ddl_bn_bwd(const void* dy, const void* x, const void* y, void* dx, long rows, int C,
                        const float* mean, const float* invstd, const float* gamma, int act, float alpha,
                        int train_stats, int prev_act, float prev_alpha, float* part, float* dgamma,
                        float* dbeta, float* c1, float* c2, float* dbias, int c_bias, int groups, int part_blocks,
                        hipStream_t s) {
    // `dbias` (optional, c_bias entries): bias gradient of the conv feeding this BN, = column sums of dx;
    // `part` then holds 5 (instead of 2) partial rows per block.
    // `part_blocks` > 0: `part` already holds [2 or 5][C][groups * part_blocks] column sums written by the kernel that
    // produced dy (the fused discriminator head, tapconv_1ch's RED epilogue): no reduction pass here.
    dim3 grid;
    colreduce_grid(rows, C, &grid);
    grid.z = groups;
    if (part_blocks > 0 && dbias) {
        grid.x = part_blocks;
        if ((int)grid.x > fin_wide_min())
            bn_bwd_finalize_wide_kernel<5><<<C, 256, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma, dbeta, c1, c2,
                                                              mean, invstd, gamma, dbias, c_bias, groups);
        else
            bn_bwd_finalize_kernel<5><<<FIN_GRID(C), FIN_BLOCK, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma,
                                                                      dbeta, c1, c2, mean, invstd, gamma, dbias,
                                                                      c_bias, groups);
    } else if (part_blocks > 0) {
        grid.x = part_blocks;
        if ((int)grid.x > fin_wide_min())
            bn_bwd_finalize_wide_kernel<2><<<C, 256, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma, dbeta, c1, c2,
                                                              mean, invstd, gamma, nullptr, 0, groups);
        else
            bn_bwd_finalize_kernel<2><<<FIN_GRID(C), FIN_BLOCK, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma,
                                                                      dbeta, c1, c2, mean, invstd, gamma, nullptr,
                                                                      0, groups);
    } else if (dbias) {
#define CALL(A, P) colreduce_kernel<3, A, P><<<grid, EW_BLOCK, 0, s>>>( \
        (const uint4*)dy, (const uint4*)x, (const uint4*)y, mean, invstd, part, rows, C, act, alpha, prev_act, prev_alpha)
        ACT2_DISPATCH(act, prev_act, CALL);
#undef CALL
        if ((int)grid.x > fin_wide_min())
            bn_bwd_finalize_wide_kernel<5><<<C, 256, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma, dbeta, c1, c2,
                                                              mean, invstd, gamma, dbias, c_bias, groups);
        else
            bn_bwd_finalize_kernel<5><<<FIN_GRID(C), FIN_BLOCK, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma,
                                                                      dbeta, c1, c2, mean, invstd, gamma, dbias,
                                                                      c_bias, groups);
    } else {
#define CALL(A) colreduce_kernel<1, A, 0><<<grid, EW_BLOCK, 0, s>>>( \
        (const uint4*)dy, (const uint4*)x, (const uint4*)y, mean, invstd, part, rows, C, act, alpha, 0, 0.f)
        ACT1_DISPATCH(act, CALL);
#undef CALL
        if ((int)grid.x > fin_wide_min())
            bn_bwd_finalize_wide_kernel<2><<<C, 256, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma, dbeta, c1, c2,
                                                              mean, invstd, gamma, nullptr, 0, groups);
        else
            bn_bwd_finalize_kernel<2><<<FIN_GRID(C), FIN_BLOCK, 0, s>>>(part, grid.x, C, rows, train_stats, dgamma,
                                                                      dbeta, c1, c2, mean, invstd, gamma, nullptr,
                                                                      0, groups);
    }
    if (!dx) return;            // finalize only: the caller applies the coefficients itself (ddl_dw_dgrad_bn)
    long n8 = rows * (C / 8);
    if (cs_ok(C / 8)) {
        int gx = cs_grid(rows * groups, C / 8) / groups;
#define CALL(A, P) bn_bwd_apply_cs_kernel<A, P><<<dim3(gx < 1 ? 1 : gx, groups), EW_BLOCK, 0, s>>>( \
        (const uint4*)dy, (const uint4*)x, (const uint4*)y, (uint4*)dx, mean, invstd, gamma, c1, c2, rows, C / 8, act, alpha, \
        prev_act, prev_alpha)
        ACT2_DISPATCH(act, prev_act, CALL);
#undef CALL
        return;
    }
    int gx = ew_grid(n8 * groups) / groups;
#define CALL(A, P) bn_bwd_apply_kernel<A, P><<<dim3(gx < 1 ? 1 : gx, groups), EW_BLOCK, 0, s>>>( \
    (const uint4*)dy, (const uint4*)x, (const uint4*)y, (uint4*)dx, mean, invstd, gamma, c1, c2, n8, C / 8, act, alpha, \
    prev_act, prev_alpha)
    ACT2_DISPATCH(act, prev_act, CALL);
#undef CALL
}

// bias gradient = per-channel column sum of dY
__global__ void colsum_finalize_kernel(const float* __restrict__ part, int nblocks, int C, int c_out,
                                       float* __restrict__ out) {
    int c = FIN_CHANNEL();
    double s0, s1;
    reduce_partials(part, nblocks, C, c, false, s0, s1);
    if (c >= c_out || !FIN_LEADER()) return;
    out[c] = (float)s0;
}

// Few rows, many channels (the generator's Dense 128 -> 6272 at batch 1000: [1000][6272]): the two-kernel scheme
// spreads only ~4 row blocks over 4 column groups and then walks partials.  One pass instead: a block owns 8
// channel vectors (64 channels), its 32 row lanes each sum every 32nd row (four loads in flight), a fixed-order LDS
// fold over the lanes writes the final sums -- deterministic, no partials, one launch.
__global__ __launch_bounds__(256) void colsum_fewrows_kernel(const uint4* __restrict__ dy, long rows, int cg,
                                                             int c_out, float* __restrict__ out) {
    __shared__ float red[32][8 * 8 + 1];
    const int gl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int g = blockIdx.x * 8 + gl;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g < cg) {
        long r = rl;
        for (; r + 96 < rows; r += 128) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = dy[(r + 32 * u) * cg + g];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += f[j];
            }
        }
        for (; r < rows; r += 32) {
            float f[8];
            unpack8(dy[r * cg + g], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][gl * 8 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x];
        if (c < c_out) out[c] = t;
    }
}

DDL_API void ddl_bias_grad(const void* dy, long rows, int C, int c_out, float* part, float* out,
                           hipStream_t s) {
    if (rows <= 8192 && C >= 1024) {
        const int cg = C / 8;
        colsum_fewrows_kernel<<<(cg + 7) / 8, 256, 0, s>>>((const uint4*)dy, rows, cg, c_out, out);
        return;
    }
    dim3 grid;
    colreduce_grid(rows, C, &grid);
    colreduce_kernel<2><<<grid, EW_BLOCK, 0, s>>>((const uint4*)dy, nullptr, nullptr, nullptr, nullptr, part,
                                                  rows, C, 0, 0.f, 0, 0.f);
    colsum_finalize_kernel<<<FIN_GRID(c_out), FIN_BLOCK, 0, s>>>(part, grid.x, C, c_out, out);
}

// ------------------------------------------------------------------------------ casts / layout
__global__ void cast_f32_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = src[i];
        dst[i] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
    }
}

DDL_API void ddl_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t s) {
    long n4 = n / 4;     // flat buffers are padded to multiples of 8 elements
    cast_f32_bf16_kernel<<<ew_grid(n4), EW_BLOCK, 0, s>>>((const float4*)src, (uint2*)dst, n4);
}

// Alternate-layout weights for the adjoint direction: canonical [R][T][C] (fp32 master) ->
// [Cp][T][Rp] (bf16), zero-filled where r >= R.  One launch handles every layer of a network
// through a descriptor table (blockIdx.y = layer).
struct RelayoutDesc { long src_off; long dst_off; int R, T, C, Rp; };

__global__ void relayout_alt_kernel(const float* __restrict__ P, bf16_t* __restrict__ alt,
                                    const RelayoutDesc* __restrict__ descs) {
    const RelayoutDesc d = descs[blockIdx.y];
    const long total = (long)d.C * d.T * d.Rp;
    const float* src = P + d.src_off;
    bf16_t* dst = alt + d.dst_off;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i % d.Rp);
        long q = i / d.Rp;
        int t = (int)(q % d.T);
        int c = (int)(q / d.T);
        float v = r < d.R ? src[((long)r * d.T + t) * d.C + c] : 0.f;
        dst[i] = f2bf(v);
    }
}

DDL_API void ddl_relayout_alt(const float* P, void* alt, const void* descs, int n_layers, long max_elems,
                              hipStream_t s) {
    if (n_layers <= 0) return;
    int gx = ew_grid(max_elems);
    if (gx > 256) gx = 256;
    relayout_alt_kernel<<<dim3(gx, n_layers), EW_BLOCK, 0, s>>>(P, (bf16_t*)alt, (const RelayoutDesc*)descs);
}

// [rows][c_in] (bf16 or fp32 or u8) -> [rows][Cp] bf16, channel c copied (or replicated when
// c_in == 1 and c_true > 1: the reference stacks MNIST to 3 channels, DDL.py:119), pad = 0.
template <typename T>
__device__ __forceinline__ float load_as_float(const T* p, long i);
template <> __device__ __forceinline__ float load_as_float<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float load_as_float<uint8_t>(const uint8_t* p, long i) { return (float)p[i]; }
template <> __device__ __forceinline__ float load_as_float<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }

// One lane per 8 output channels (one 16-byte store); the sample gather and the divisions are per row, not per element
// (the element-per-lane form spent its time in 64-bit div / mod: 10.8 us for a 512 x 28 x 28 batch).
template <typename T>
__global__ void pad_channels_kernel(const T* __restrict__ src, const long* __restrict__ index,
                                    bf16_t* __restrict__ dst, long rows, int pix_per_sample, int c_in,
                                    int c_true, int Cp, float scale) {
    const int cg = Cp / 8;
    const long total = rows * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = cg == 1 ? i : i / cg;
        const int g = cg == 1 ? 0 : (int)(i - r * cg);
        long sr = r;
        if (index) {
            const long smp = rows < (1L << 31) ? (long)((unsigned)r / (unsigned)pix_per_sample) : r / pix_per_sample;
            sr = index[smp] * pix_per_sample + (r - smp * pix_per_sample);
        }
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            f[j] = c < c_true ? load_as_float<T>(src, sr * c_in + (c_in == 1 ? 0 : c)) * scale : 0.f;
        }
        ((uint4*)dst)[i] = pack8(f);
    }
}

// dtype: 0 = fp32, 1 = u8, 2 = bf16.  `index` (optional, int64 per sample) gathers samples.
DDL_API void ddl_pad_channels(const void* src, int dtype, const long* index, void* dst, long rows,
                              int pix_per_sample, int c_in, int c_true, int Cp, float scale, hipStream_t s) {
    long total = rows * (Cp / 8);
    int grid = ew_grid(total);
    if (dtype == 0)
        pad_channels_kernel<float><<<grid, EW_BLOCK, 0, s>>>((const float*)src, index, (bf16_t*)dst, rows,
                                                             pix_per_sample, c_in, c_true, Cp, scale);
    else if (dtype == 1)
        pad_channels_kernel<uint8_t><<<grid, EW_BLOCK, 0, s>>>((const uint8_t*)src, index, (bf16_t*)dst, rows,
                                                               pix_per_sample, c_in, c_true, Cp, scale);
    else
        pad_channels_kernel<bf16_t><<<grid, EW_BLOCK, 0, s>>>((const bf16_t*)src, index, (bf16_t*)dst, rows,
                                                              pix_per_sample, c_in, c_true, Cp, scale);
}

// [rows][Cp] bf16 -> [rows][c_true] fp32 (strip padding; used for predictions / export of activations)
__global__ void strip_channels_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long rows,
                                      int Cp, int c_true) {
    const long total = rows * c_true;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % c_true);
        long r = i / c_true;
        dst[i] = bf2f(src[r * Cp + c]);
    }
}

DDL_API void ddl_strip_channels(const void* src, float* dst, long rows, int Cp, int c_true, hipStream_t s) {
    strip_channels_kernel<<<ew_grid(rows * c_true), EW_BLOCK, 0, s>>>((const bf16_t*)src, dst, rows, Cp, c_true);
}

// Spatial zero padding / cropping of an NHWC tensor (ZeroPadding2D and its backward).
__global__ void spatial_pad_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int Hs, int Ws,
                                   int Hd, int Wd, int cg, int off_h, int off_w) {
    const long total = (long)B * Hd * Wd * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int g = (int)(i % cg);
        long p = i / cg;
        int w = (int)(p % Wd);
        long q = p / Wd;
        int h = (int)(q % Hd);
        int b = (int)(q / Hd);
        int hs = h - off_h, ws = w - off_w;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)hs < (unsigned)Hs && (unsigned)ws < (unsigned)Ws)
            v = src[(((long)b * Hs + hs) * Ws + ws) * cg + g];
        dst[i] = v;
    }
}

// dst[b][h][w] = src[b][h - off_h][w - off_w] (zero outside); negative offsets crop.
DDL_API void ddl_spatial_pad(const void* src, void* dst, int B, int Hs, int Ws, int Hd, int Wd, int C,
                             int off_h, int off_w, hipStream_t s) {
    long total = (long)B * Hd * Wd * (C / 8);
    spatial_pad_kernel<<<ew_grid(total), EW_BLOCK, 0, s>>>((const uint4*)src, (uint4*)dst, B, Hs, Ws, Hd, Wd,
                                                           C / 8, off_h, off_w);
}

// y = x * scale + shift on a bf16 tensor (generic Lambda-affine), padding channels kept zero.
__global__ void affine_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long n8, int cg, int c_true,
                              float scale, float shift) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(x[i], f);
        int c0 = (int)(i % cg) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (c0 + j < c_true) ? fmaf(f[j], scale, shift) : 0.f;
        y[i] = pack8(f);
    }
}

DDL_API void ddl_affine(const void* x, void* y, long rows, int C, int c_true, float scale, float shift,
                        hipStream_t s) {
    long n8 = rows * (C / 8);
    affine_kernel<<<ew_grid(n8), EW_BLOCK, 0, s>>>((const uint4*)x, (uint4*)y, n8, C / 8, c_true, scale, shift);
}

// ------------------------------------------------------------------------------ one-channel convs
// A conv whose input (Conv2D 1->N) or output (Conv2DTranspose C->1) has a single channel wastes
// 7/8 of an implicit GEMM on channel padding.  Instead the single-channel image is expanded to a
// "taps as channels" tensor (im2col: [B][OH][OW][Tp], Tp = taps padded to 8), which turns the layer
// into a 1x1 convolution on the MFMA kernels, and col2im folds a taps-as-channels tensor back into
// a single-channel image (with the layer's bias + activation fused in).
//   im2col : col[b][oh][ow][t] = img[b][oh*sh + kh*dh - pt][ow*sw + kw*dw - pl][0]      (0 outside / t >= T)
//   col2im : img[b][h][w][0]   = act(bias + sum_t col[b][(h + pt - kh*dh)/sh][(w + pl - kw*dw)/sw][t])
struct OneChGeom { int H, W, OH, OW, KH, KW, sh, sw, dh, dw, pt, pl, T, Tp, Ci;
                   int ones; };   // im2col: padding tap T holds 1.0 (the weight gradient's bias column, wgrad.hip)

// Generic fallback (planes too large for LDS): one thread per 8 taps, gathers straight from global memory.
__global__ void im2col_1ch_generic_kernel(const bf16_t* __restrict__ img, uint4* __restrict__ col, long n_units,
                                          const OneChGeom g) {
    const int tg = g.Tp / 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_units; i += (long)gridDim.x * blockDim.x) {
        int t8 = (int)(i % tg);
        long p = i / tg;
        int ow = (int)(p % g.OW);
        long q = p / g.OW;
        int oh = (int)(q % g.OH);
        long b = q / g.OH;
        const bf16_t* base = img + b * g.H * g.W * g.Ci;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int t = t8 * 8 + j;
            uint32_t v = 0;
            if (t < g.T) {
                int kh = t / g.KW, kw = t - kh * g.KW;
                int ih = oh * g.sh + kh * g.dh - g.pt, iw = ow * g.sw + kw * g.dw - g.pl;
                if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) v = base[((long)ih * g.W + iw) * g.Ci];
            } else if (g.ones && t == g.T) {
                v = 0x3F80u;
            }
            w[j >> 1] |= v << ((j & 1) * 16);
        }
        col[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// floor(x / d) for 0 <= x < 2^22 through the reciprocal (exact in that range)
__device__ __forceinline__ int fdiv_small(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

// LDS-staged im2col: a block owns `ipb` images.  Their real channel is staged once as zero-haloed planes
// [PH][PW] (halo = every out-of-image tap position, so the gather needs no bounds test), a small table
// maps tap -> plane offset, and then consecutive lanes emit consecutive 16-byte groups of 8 taps: eight
// 2-byte LDS reads and one coalesced global store each, no integer division anywhere in the loop.
__global__ __launch_bounds__(256) void im2col_1ch_kernel(const bf16_t* __restrict__ img, uint4* __restrict__ col,
                                                         int B, const OneChGeom g, int ipb, int PH, int PW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* tap = (int*)smem;                                   // [Tp] plane offset of tap t (-1: padding tap)
    bf16_t* plane = (bf16_t*)(smem + g.Tp * 4);             // [ipb][PH*PW]
    const int b0 = blockIdx.x * ipb;
    const int nimg = min(ipb, B - b0);
    const int psz = PH * PW;
    const float invPW = 1.0f / (float)PW;
    for (int t = threadIdx.x; t < g.Tp; t += 256) {
        int kh = t / g.KW, kw = t - kh * g.KW;
        tap[t] = t < g.T ? kh * g.dh * PW + kw * g.dw : ((g.ones && t == g.T) ? -2 : -1);
    }
    // four plane elements per thread and trip, their loads issued together (a one-image block has four trips: as a
    // loop of guarded loads that was four global-memory latencies in a row before the first store)
    const float invPsz = 1.0f / (float)psz;
    for (int i0 = threadIdx.x; i0 < nimg * psz; i0 += 1024) {
        bf16_t v[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = min(i0 + k * 256, nimg * psz - 1);
            const int im = fdiv_small(i, invPsz), e = i - im * psz;
            const int r = fdiv_small(e, invPW), c = e - r * PW;
            const int ih = r - g.pt, iw = c - g.pl;
            in[k] = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
            const int ihc = min(max(ih, 0), g.H - 1), iwc = min(max(iw, 0), g.W - 1);
            v[k] = img[(((long)(b0 + im) * g.H + ihc) * g.W + iwc) * g.Ci];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * 256;
            if (i < nimg * psz) plane[i] = in[k] ? v[k] : (bf16_t)0;
        }
    }
    __syncthreads();
    const int tg = g.Tp / 8, npix = g.OH * g.OW, per_img = npix * tg;
    const float inv_tg = 1.0f / (float)tg, invOW = 1.0f / (float)g.OW, inv_per = 1.0f / (float)per_img;
    uint4* out = col + (long)b0 * per_img;
    for (int u = threadIdx.x; u < nimg * per_img; u += 256) {
        const int im = fdiv_small(u, inv_per), e = u - im * per_img;
        const int p = fdiv_small(e, inv_tg), t8 = e - p * tg;
        const int oh = fdiv_small(p, invOW), ow = p - oh * g.OW;
        const bf16_t* src = plane + im * psz + oh * g.sh * PW + ow * g.sw;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = tap[t8 * 8 + j];
            uint32_t v = src[off < 0 ? 0 : off];
            v = off < 0 ? (off == -2 ? 0x3F80u : 0u) : v;
            w[j >> 1] |= v << ((j & 1) * 16);
        }
        out[u] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// One block per (image, band of output rows): the band's slice of the taps-as-channels tensor is
// staged into LDS with coalesced 16-byte loads (it is read up to KH*KW times, two bytes at a time, by
// the fold), then every thread folds pixels.  LDS rows are padded by 4 B so the 2-byte gathers of
// neighbouring pixels spread over the banks.  SH / SW: compile-time strides (1 | 2; 0 = runtime): the
// fold's divisibility tests and divisions cost more than its loads when they are runtime integers
// (measured 34 us for a 50 MB tensor: ALU-bound at 1.5 TB/s).
template <int SH, int SW>
__global__ __launch_bounds__(256) void col2im_1ch_kernel(const bf16_t* __restrict__ col, uint4* __restrict__ img,
                                                         const OneChGeom g, const float* __restrict__ bias, int act,
                                                         float alpha, int band_rows, const bf16_t* __restrict__ mul_y,
                                                         int mul_act, float mul_alpha) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sh = SH ? SH : g.sh, sw = SW ? SW : g.sw;
    const int b = blockIdx.x;
    const int h0 = blockIdx.y * band_rows, h1 = min(g.H, h0 + band_rows);
    const int pitch = g.Tp * 2 + 4;
    int lo = h0 + g.pt - (g.KH - 1) * g.dh;                  // smallest th of the band
    const int oh_lo = lo <= 0 ? 0 : (lo + sh - 1) / sh;
    const int oh_hi = min(g.OH - 1, (h1 - 1 + g.pt) / sh);
    const int nrows = oh_hi - oh_lo + 1;
    const bf16_t* base = col + ((long)b * g.OH + oh_lo) * g.OW * g.Tp;
    if (nrows > 0) {
        const int u_per = g.Tp / 8;
        const float inv_u = 1.0f / (float)u_per;
        for (int u = threadIdx.x; u < nrows * g.OW * u_per; u += 256) {
            int p = fdiv_small(u, inv_u), c8 = u - p * u_per;
            uint4 v = *(const uint4*)(base + (long)p * g.Tp + c8 * 8);
            uint32_t* d = (uint32_t*)(smem + p * pitch + c8 * 16);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
    const float invW = 1.0f / (float)g.W;
    auto fold = [&](auto KA) {                  // the loop compiled per activation (common.h act_switch)
    for (int i = threadIdx.x; i < (h1 - h0) * g.W; i += 256) {
        int hl = fdiv_small(i, invW), w = i - hl * g.W, h = h0 + hl;
        float acc = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            int th = h + g.pt - kh * g.dh;
            if (th < 0 || th % sh) continue;
            int oh = th / sh;
            if (oh > oh_hi) continue;
            const char* row = smem + (oh - oh_lo) * g.OW * pitch + kh * g.KW * 2;
            for (int kw = 0; kw < g.KW; ++kw) {
                int tw = w + g.pl - kw * g.dw;
                if (tw < 0 || tw % sw) continue;
                int ow = tw / sw;
                if (ow >= g.OW) continue;
                acc += bf2f(*(const bf16_t*)(row + ow * pitch + kw * 2));
            }
        }
        float v = act_apply_k<decltype(KA)::value>(acc + b0, act, alpha);
        const long pix = ((long)b * g.H + h) * g.W + w;
        // fused activation backward of the layer below: (bf16 result) * act'(its saved output), rounded once more
        if (mul_y) v = bf2f(f2bf(v)) * act_grad_y(bf2f(mul_y[pix * 8]), mul_act, mul_alpha);
        img[pix] = make_uint4((uint32_t)f2bf(v), 0, 0, 0);   // channel 0 real, 7 lanes zero
    }
    };
    act_switch(act, fold);
}

DDL_API int ddl_im2col_1ch(const void* img, void* col, int B, int H, int W, int Ci, int OH, int OW, int KH, int KW,
                           int sh, int sw, int dh, int dw, int pt, int pl, int Tp, int ones, hipStream_t s) {
    if (Tp % 8 || Tp < KH * KW || Ci <= 0 || (ones && Tp == KH * KW)) return -1;
    OneChGeom g{H, W, OH, OW, KH, KW, sh, sw, dh, dw, pt, pl, KH * KW, Tp, Ci, ones ? 1 : 0};
    // zero-haloed plane covering every tap position: rows -pt .. (OH-1)*sh + (KH-1)*dh - pt
    const long PH = (long)(OH - 1) * sh + (long)(KH - 1) * dh + 1, PW = (long)(OW - 1) * sw + (long)(KW - 1) * dw + 1;
    const long psz = PH * PW, per_img = (long)OH * OW * (Tp / 8);
    if (B > 0 && psz * 2 + Tp * 4 <= 48 * 1024 && per_img < (1L << 20)) {
        long ipb = 4096 / per_img;                               // >= ~16 stores per thread and block
        if (ipb < 1) ipb = 1;
        while (ipb > 1 && (ipb * psz * 2 + Tp * 4 > 48 * 1024 || ipb * per_img >= (1L << 21) || ipb * psz >= (1L << 21)))
            --ipb;
        while (ipb > 1 && (B + ipb - 1) / ipb < 1024) --ipb;     // keep >= 4 blocks per CU when the batch allows
        const int blocks = (int)((B + ipb - 1) / ipb);
        const size_t smem = (size_t)(ipb * psz * 2 + Tp * 4);
        im2col_1ch_kernel<<<blocks, 256, smem, s>>>((const bf16_t*)img, (uint4*)col, B, g, (int)ipb, (int)PH, (int)PW);
        return 0;
    }
    long n = (long)B * per_img;
    im2col_1ch_generic_kernel<<<ew_grid(n), EW_BLOCK, 0, s>>>((const bf16_t*)img, (uint4*)col, n, g);
    return 0;
}

// img is written as [B][H][W][8] (one real channel).
DDL_API int ddl_col2im_1ch(const void* col, void* img, const float* bias, int B, int H, int W, int OH, int OW,
                           int KH, int KW, int sh, int sw, int dh, int dw, int pt, int pl, int Tp, int act,
                           float alpha, const void* mul_y, int mul_act, float mul_alpha, hipStream_t s) {
    if (Tp % 8 || Tp < KH * KW) return -1;
    OneChGeom g{H, W, OH, OW, KH, KW, sh, sw, dh, dw, pt, pl, KH * KW, Tp, 8};
    // band height: the staged slice ((rows-1)/sh + halo rows of the col tensor) stays under ~40 KiB
    const size_t row_bytes = (size_t)OW * (Tp * 2 + 4);
    const int halo = ((KH - 1) * dh) / sh + 2;
    static const size_t band_cap = getenv("DDL_COL2IM_LDS_KB") ? (size_t)atol(getenv("DDL_COL2IM_LDS_KB")) * 1024 : 40 * 1024;
    int band = H;
    while (band > 1 && ((size_t)((band - 1) / sh + halo)) * row_bytes > band_cap) band = (band + 1) / 2;
    size_t smem = ((size_t)((band - 1) / sh + halo)) * row_bytes;
    if (smem > 160 * 1024) return -2;
    if ((long)((band - 1) / sh + halo) * OW * (Tp / 8) >= (1L << 21) || (long)band * W >= (1L << 21)) return -2;
    auto kern = (sh == 1 && sw == 1) ? col2im_1ch_kernel<1, 1>
              : (sh == 2 && sw == 2) ? col2im_1ch_kernel<2, 2> : col2im_1ch_kernel<0, 0>;
    if (smem > 64 * 1024) {
        static bool done[3] = {false, false, false};
        const int v = (sh == 1 && sw == 1) ? 0 : ((sh == 2 && sw == 2) ? 1 : 2);
        if (!done[v]) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done[v] = true;
        }
    }
    kern<<<dim3(B, (H + band - 1) / band), 256, smem, s>>>((const bf16_t*)col, (uint4*)img, g, bias, act, alpha, band,
                                                           (const bf16_t*)mul_y, mul_act, mul_alpha);
    return 0;
}

// Generic strided weight gather fp32 master -> bf16 [R2][C2] (zero where r >= R1 or c >= C1):
// dst[r][c] = P[src_off + r*sr + c*sc].  One launch serves every descriptor (blockIdx.y).
struct GatherDesc { long src_off; long dst_off; int R1, C1, R2, C2; long sr, sc; };

__global__ void relayout_gather_kernel(const float* __restrict__ P, bf16_t* __restrict__ out,
                                       const GatherDesc* __restrict__ descs) {
    const GatherDesc d = descs[blockIdx.y];
    const long total = (long)d.R2 * d.C2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % d.C2);
        int r = (int)(i / d.C2);
        float v = (r < d.R1 && c < d.C1) ? P[d.src_off + r * d.sr + c * d.sc] : 0.f;
        out[d.dst_off + i] = f2bf(v);
    }
}

// Both kinds of weight copy of a network in ONE launch: blockIdx.y < n_alt walks an alternate-layout descriptor,
// the rest walk strided-gather descriptors (two launches per optimizer step otherwise, each a few microseconds
// of mostly launch latency).
__global__ void relayout_both_kernel(const float* __restrict__ P, bf16_t* __restrict__ alt,
                                     const RelayoutDesc* __restrict__ adescs, int n_alt, bf16_t* __restrict__ gout,
                                     const GatherDesc* __restrict__ gdescs) {
    if ((int)blockIdx.y < n_alt) {
        const RelayoutDesc d = adescs[blockIdx.y];
        const long total = (long)d.C * d.T * d.Rp;
        const float* src = P + d.src_off;
        bf16_t* dst = alt + d.dst_off;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            int r = (int)(i % d.Rp);
            long q = i / d.Rp;
            int t = (int)(q % d.T);
            int c = (int)(q / d.T);
            float v = r < d.R ? src[((long)r * d.T + t) * d.C + c] : 0.f;
            dst[i] = f2bf(v);
        }
    } else {
        const GatherDesc d = gdescs[blockIdx.y - n_alt];
        const long total = (long)d.R2 * d.C2;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            int c = (int)(i % d.C2);
            int r = (int)(i / d.C2);
            float v = (r < d.R1 && c < d.C1) ? P[d.src_off + r * d.sr + c * d.sc] : 0.f;
            gout[d.dst_off + i] = f2bf(v);
        }
    }
}

DDL_API void ddl_relayout_both(const float* P, void* alt, const void* adescs, int n_alt, void* gout,
                               const void* gdescs, int n_gather, long max_elems, hipStream_t s) {
    if (n_alt + n_gather <= 0) return;
    int gx = ew_grid(max_elems);
    if (gx > 256) gx = 256;
    relayout_both_kernel<<<dim3(gx, n_alt + n_gather), EW_BLOCK, 0, s>>>(
        P, (bf16_t*)alt, (const RelayoutDesc*)adescs, n_alt, (bf16_t*)gout, (const GatherDesc*)gdescs);
}

DDL_API void ddl_relayout_gather(const float* P, void* out, const void* descs, int n, long max_elems,
                                 hipStream_t s) {
    if (n <= 0) return;
    int gx = ew_grid(max_elems);
    if (gx > 64) gx = 64;
    relayout_gather_kernel<<<dim3(gx, n), EW_BLOCK, 0, s>>>(P, (bf16_t*)out, (const GatherDesc*)descs);
}
