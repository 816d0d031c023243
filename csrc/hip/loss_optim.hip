// Fused loss kernels (softmax-CE / softmax-MSE / sigmoid-BCE / Dense(K->1)+BCE forward+backward),
// metrics, the flat multi-tensor optimizers and the Philox noise generator.
#include "common.h"
#include <cstdlib>
#include "optim_update.h"

#define LB 256

// ------------------------------------------------------------------------------ softmax family
// logits: [B][Cs] bf16 (Cs = padded stride), N true classes (N <= 64).  One row per lane.
// mode 0: probabilities only; mode 1: categorical cross-entropy; mode 2: mean squared error.
// Row losses go to rowloss[B]; dlogits (bf16, [B][Cs], pad = 0) already carry the 1/B mean factor.
// `index` (optional): row r's target is target[index[r]] -- the label gather of a minibatch drawn by index from the
// HBM-resident dataset happens here, not in a gather pass of its own.
__global__ void mean_kernel(const float* __restrict__ v, long n, float scale, float* __restrict__ out);

// NV: compile-time bound on N (16 / 64): the row lives in registers (a run-time-indexed v[64] went to scratch memory:
// 272 bytes per lane and 12 us for 512 rows of 10 classes).  MEAN: the launch is ONE block covering every row, and it
// also leaves mean(rowloss) in loss_out (same double-precision tree as mean_kernel) -- one dispatch per loss instead
// of two.
template <int NV, bool MEAN>
__global__ void softmax_loss_kernel(const bf16_t* __restrict__ logits, const float* __restrict__ target,
                                    const long* __restrict__ index, float* __restrict__ probs,
                                    bf16_t* __restrict__ dlogits, float* __restrict__ rowloss,
                                    float* __restrict__ loss_out, int B, int Cs, int N, int mode, int want_grad) {
    __shared__ double red[MEAN ? 1024 : 1];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    if (r < B) {
        const bf16_t* l = logits + (size_t)r * Cs;
        float v[NV];
        if (NV == 16 && Cs == 16) {                 // the common head: 10 classes padded to 16 = two 16-byte loads
            unpack8(*(const uint4*)l, v);
            unpack8(*(const uint4*)(l + 8), v + 8);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = j < N ? bf2f(l[j]) : 0.f;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < NV; ++j) mx = j < N ? fmaxf(mx, v[j]) : mx;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { v[j] = j < N ? __expf(v[j] - mx) : 0.f; sum += v[j]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] *= inv;
        if (probs) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (j < N) probs[(size_t)r * N + j] = v[j];
        }
        if (mode != 0) {
            const float* tp = target + (size_t)(index ? index[r] : r) * N;
            float t[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) t[j] = j < N ? tp[j] : 0.f;
            float d[NV];
            if (mode == 1) {
                float tsum = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (j < N) {
                        const float pc = fminf(fmaxf(v[j], 1e-7f), 1.f - 1e-7f);
                        loss -= t[j] * __logf(pc);
                        tsum += t[j];
                    }
                }
                const float ib = 1.f / (float)B;
#pragma unroll
                for (int j = 0; j < NV; ++j) d[j] = j < N ? (v[j] * tsum - t[j]) * ib : 0.f;
            } else {
                float dot = 0.f;
                const float k = 2.f / ((float)B * (float)N);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (j < N) {
                        const float df = v[j] - t[j];
                        loss += df * df;
                        dot += k * df * v[j];
                    }
                }
                loss /= (float)N;
#pragma unroll
                for (int j = 0; j < NV; ++j) d[j] = j < N ? v[j] * (k * (v[j] - t[j]) - dot) : 0.f;
            }
            if (want_grad) {
                bf16_t* dp = dlogits + (size_t)r * Cs;
                if (NV == 16 && Cs == 16) {
                    *(uint4*)dp = pack8(d);
                    *(uint4*)(dp + 8) = pack8(d + 8);
                } else {
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        if (j < Cs) dp[j] = f2bf(d[j]);
                    for (int j = NV; j < Cs; ++j) dp[j] = 0;
                }
            }
            rowloss[r] = loss;
        }
    }
    if (MEAN && mode != 0) {
        // blockDim.x is a power of two >= B: the tree pairs rows exactly as mean_kernel's strided walk + tree does
        red[threadIdx.x] = r < B ? (double)loss : 0.0;
        __syncthreads();
        for (int o = blockDim.x / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_out[0] = (float)(red[0] * (1.0 / (double)B));
    }
}

// mean of n floats -> out[0] (single block, deterministic order)
__global__ void mean_kernel(const float* __restrict__ v, long n, float scale, float* __restrict__ out) {
    __shared__ double red[LB];
    double s = 0;
    for (long i = threadIdx.x; i < n; i += LB) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = LB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

DDL_API void ddl_softmax_loss(const void* logits, const float* target, const long* index, float* probs,
                              void* dlogits, float* rowloss, float* loss_out, int B, int Cs, int N, int mode,
                              int want_grad, hipStream_t s) {
    const bf16_t* lg = (const bf16_t*)logits;
    bf16_t* dl = (bf16_t*)dlogits;
    const bool small = N <= 16;
    if (B <= 1024 && mode != 0 && loss_out) {       // one block: losses, gradients and the mean loss in one launch
        int threads = 64;
        while (threads < B) threads *= 2;
        if (small) softmax_loss_kernel<16, true><<<1, threads, 0, s>>>(lg, target, index, probs, dl, rowloss, loss_out,
                                                                        B, Cs, N, mode, want_grad);
        else softmax_loss_kernel<64, true><<<1, threads, 0, s>>>(lg, target, index, probs, dl, rowloss, loss_out, B, Cs,
                                                                  N, mode, want_grad);
        return;
    }
    if (small) softmax_loss_kernel<16, false><<<(B + 127) / 128, 128, 0, s>>>(lg, target, index, probs, dl, rowloss,
                                                                               loss_out, B, Cs, N, mode, want_grad);
    else softmax_loss_kernel<64, false><<<(B + 127) / 128, 128, 0, s>>>(lg, target, index, probs, dl, rowloss, loss_out,
                                                                         B, Cs, N, mode, want_grad);
    if (mode != 0 && loss_out) mean_kernel<<<1, LB, 0, s>>>(rowloss, B, 1.0 / (double)B, loss_out);
}

// ------------------------------------------------------------------------------ sigmoid BCE
__device__ __forceinline__ float bce_logits(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(__expf(-fabsf(x)));
}

__global__ void sigmoid_bce_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                   float* __restrict__ rowloss, float* __restrict__ dl, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = logits[i], z = labels[i];
    rowloss[i] = bce_logits(x, z);
    if (dl) dl[i] = (1.f / (1.f + __expf(-x)) - z) / (float)n;
}

DDL_API void ddl_sigmoid_bce(const float* logits, const float* labels, float* rowloss, float* dl,
                             float* loss_out, int n, hipStream_t s) {
    sigmoid_bce_kernel<<<(n + 255) / 256, 256, 0, s>>>(logits, labels, rowloss, dl, n);
    mean_kernel<<<1, LB, 0, s>>>(rowloss, n, 1.0 / (double)n, loss_out);
}

// ------------------------------------------------------------------------------ Dense(K->1)+BCE
// The DCGAN discriminator head in one pass: one wave per sample row computes the logit (dot of
// the bf16 feature row with the weight vector), the BCE term and d logit, then immediately
// writes dX[row] = dlogit * w (outer product) while the row is still in registers; per-block
// partial dW goes to a slab reduced by a second tiny kernel.  K multiple of 8, K <= 64*8*KMAXV.
#define HEAD_ROWS 4      // waves (= rows in flight) per block
#define HEAD_VMAX 8      // uint4 vectors per lane  -> K <= 4096

// Optional fusion of the BatchNorm (+ activation) in FRONT of the head and of that BatchNorm's backward reduction
// (BNF instantiation; the DCGAN discriminator ends BN -> LeakyReLU -> Flatten -> Dense(1)):
//   * x is the PRE-BatchNorm tensor; a = bf16(act(x * scale[c] + shift[c])) is formed on load and never stored
//     (saves the bn_apply pass), c = (flat index) mod C;
//   * dX receives g = bf16(dlogit * w) * act'(a), the gradient with respect to the BatchNorm output BEFORE its
//     activation, and the block's column sums of g and g * xhat go to bn_part[2][C][gridDim.y * gridDim.x] --
//     exactly what colreduce_kernel<1> would produce from (dX, x, a) in another pass over the three tensors.
// Batch groups (per-half statistics) ride on blockIdx.y: a block only sees rows of its group.  All VM vectors of a
// lane must carry the same 8 channels: 512 % C == 0.
struct HeadBn {
    const float* scale; const float* shift; const float* mean; const float* invstd;   // [groups][C]
    float* part;                                                                     // [2][C][groups * nb]
    int C, rows_per_group, act; float alpha;
};

// BACT: the fused BatchNorm's activation as a compile-time code (-1: run-time bn.act), see common.h act_apply_k
template <int VM, bool BNF, int BACT = -1>
__global__ __launch_bounds__(HEAD_ROWS * 64) void dense1_bce_kernel(
    const uint4* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    float label_first, float label_second, int n_first, int M, int K, uint4* __restrict__ dx,
    float* __restrict__ dw_part, float* __restrict__ logits, float* __restrict__ rowloss,
    float* __restrict__ dlogit_out, int want_dx, int want_dw, const HeadBn bn) {
    __shared__ float sw[VM * 64 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kv = K / 8;                       // vectors per row
    for (int i = threadIdx.x; i < K; i += blockDim.x) sw[i] = bf2f(w[i]);
    __syncthreads();
    float acc_dw[VM][8];
#pragma unroll
    for (int v = 0; v < VM; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc_dw[v][j] = 0.f;
    const float b = bias[0];
    const float invM = 1.f / (float)M;
    // BatchNorm fusion: this lane's 8 channels (the same for every v) and their per-group coefficients
    float sc[8], sh[8], mu[8], is[8], s0[8], s1[8];
    const int c0 = BNF ? (lane * 8) % bn.C : 0;
    if (BNF) {
        const int go = blockIdx.y * bn.C + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = bn.scale[go + j]; sh[j] = bn.shift[go + j]; mu[j] = bn.mean[go + j]; is[j] = bn.invstd[go + j];
            s0[j] = s1[j] = 0.f;
        }
    }
    const int r_lo = BNF ? blockIdx.y * bn.rows_per_group : 0;
    const int r_hi = BNF ? min(M, r_lo + bn.rows_per_group) : M;
    for (int r = r_lo + blockIdx.x * HEAD_ROWS + wv; r < r_hi; r += gridDim.x * HEAD_ROWS) {
        float f[VM][8], xr[BNF ? VM : 1][8];
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < VM; ++v) {
            int idx = v * 64 + lane;
            if (idx < kv) {
                unpack8(x[(size_t)r * kv + idx], f[v]);
                if (BNF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xr[v][j] = f[v][j];
                        f[v][j] = bf2f(f2bf(act_apply_k<BACT>(fmaf(f[v][j], sc[j], sh[j]), bn.act, bn.alpha)));
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) dot += f[v][j] * sw[idx * 8 + j];
            }
        }
        dot = wave_sum(dot) + b;
        float z = r < n_first ? label_first : label_second;
        float dl = (1.f / (1.f + __expf(-dot)) - z) * invM;
        if (lane == 0) { logits[r] = dot; rowloss[r] = bce_logits(dot, z); dlogit_out[r] = dl; }
#pragma unroll
        for (int v = 0; v < VM; ++v) {
            int idx = v * 64 + lane;
            if (idx < kv) {
                if (want_dw) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc_dw[v][j] += dl * f[v][j];
                }
                if (want_dx) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = dl * sw[idx * 8 + j];
                    if (BNF) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            o[j] = bf2f(f2bf(bf2f(f2bf(o[j])) * act_grad_y_k<BACT>(f[v][j], bn.act, bn.alpha)));
                            s0[j] += o[j];
                            s1[j] += o[j] * (xr[v][j] - mu[j]) * is[j];
                        }
                    }
                    dx[(size_t)r * kv + idx] = pack8(o);
                }
            }
        }
    }
    const int bidx = blockIdx.y * gridDim.x + blockIdx.x;
    __syncthreads();
    if (BNF && want_dx) {
        // lanes l, l + C/8, ... hold the same channels: fold them, then the four waves through LDS
        const int P = bn.C / 8;
        for (int off = P; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += __shfl_xor(s0[j], off, 64); s1[j] += __shfl_xor(s1[j], off, 64); }
        }
        float* red = sw;                           // [4 waves][P lanes][16]; P * 64 floats <= K floats of sw
        const int pl = P < 64 ? P : 64;
        if (lane < pl) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(wv * pl + lane) * 16 + j] = s0[j]; red[(wv * pl + lane) * 16 + 8 + j] = s1[j]; }
        }
        __syncthreads();
        const size_t nbt = (size_t)gridDim.x * gridDim.y;
        for (int t = threadIdx.x; t < pl * 16; t += blockDim.x) {
            const int l = t >> 4, e = t & 15;
            float v = red[t];
#pragma unroll
            for (int q = 1; q < HEAD_ROWS; ++q) v += red[q * pl * 16 + t];
            const int k = e >> 3, c = (l * 8) % bn.C + (e & 7);
            bn.part[((size_t)k * bn.C + c) * nbt + bidx] = v;
        }
        __syncthreads();
    }
    if (!want_dw) return;
    // reduce the HEAD_ROWS waves' partial dW through LDS (reuse sw after a barrier)
    float* red = sw;    // K floats
    for (int q = 0; q < HEAD_ROWS; ++q) {
        if (wv == q) {
#pragma unroll
            for (int v = 0; v < VM; ++v) {
                int idx = v * 64 + lane;
                if (idx < kv) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (q == 0) red[idx * 8 + j] = acc_dw[v][j];
                        else red[idx * 8 + j] += acc_dw[v][j];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < K; i += blockDim.x) dw_part[(size_t)bidx * K + i] = red[i];
}

// dw[k] = sum_b dw_part[b][k]: 256 threads = 32 k-columns x 8 partial lanes; block 0 also sums db.
__global__ __launch_bounds__(256) void head_finalize_kernel(const float* __restrict__ dw_part, int nblocks, int K,
                                                            const float* __restrict__ dlogit, int M,
                                                            float* __restrict__ dw, float* __restrict__ db,
                                                            const float* __restrict__ rowloss,
                                                            float* __restrict__ loss_out) {
    __shared__ double fin[8][33];
    __shared__ float wsum[4];
    const int cl = threadIdx.x & 31, lb = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + cl;
    double s = 0;
    if (k < K) {
        int b = lb;
        // sixteen, then eight independent loads per trip, summed in a fixed order (the walk is a latency chain: 512
        // partial rows are 64 loads per thread)
        for (; b + 120 < nblocks; b += 128) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = dw_part[(size_t)(b + 8 * u) * K + k];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; b + 56 < nblocks; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dw_part[(size_t)(b + 8 * u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 8) s += dw_part[(size_t)b * K + k];
    }
    fin[lb][cl] = s;
    __syncthreads();
    if (lb == 0 && k < K) {
        double t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += fin[q][cl];
        dw[k] = (float)t;
    }
    if (blockIdx.x == 0) {
        float t = 0.f;
        for (int i = threadIdx.x; i < M; i += 256 * 8) {       // eight loads in flight (a serial walk is 1 us per load)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + 256 * u < M ? dlogit[i + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) db[0] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (blockIdx.x == gridDim.x - 1 && loss_out) {     // mean loss (saves the separate mean_kernel launch)
        __shared__ double lsum[4];
        double t = 0;
        for (int i = threadIdx.x; i < M; i += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + 256 * u < M ? rowloss[i + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63) == 0) lsum[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) loss_out[0] = (float)((lsum[0] + lsum[1] + lsum[2] + lsum[3]) / (double)M);
    }
}

DDL_API int ddl_dense1_bce_blocks(int M) {
    int nb = (M + HEAD_ROWS - 1) / HEAD_ROWS;
    return nb > 256 ? 256 : (nb < 1 ? 1 : nb);
}

// bn_*: optional fused BatchNorm in front of the head (see HeadBn); bn_part receives [2][C][groups * blocks] partials
// for ddl_bn_bwd(..., part_blocks = blocks).  `groups` = batch groups (rows split evenly), 1 without BN fusion.
DDL_API int ddl_dense1_bce(const void* x, const void* w, const float* bias, float label_first,
                           float label_second, int n_first, int M, int K, void* dx, float* dw_part,
                           float* logits, float* rowloss, float* dlogit, float* loss_out, float* dw, float* db,
                           int want_dx, int want_dw, const float* bn_scale, const float* bn_shift,
                           const float* bn_mean, const float* bn_invstd, float* bn_part, int bn_C, int groups,
                           int bn_act, float bn_alpha, hipStream_t s) {
    if (K % 8 != 0 || K > HEAD_VMAX * 64 * 8) return -1;
    HeadBn bn{bn_scale, bn_shift, bn_mean, bn_invstd, bn_part, bn_C, M, bn_act, bn_alpha};
    const bool fuse = bn_scale != nullptr;
    if (fuse) {
        if (!bn_shift || !bn_mean || !bn_invstd || !bn_part || bn_C <= 0 || bn_C % 8 || 512 % bn_C || K % bn_C ||
            K > 4 * 64 * 8 || groups < 1 || M % groups || (bn_C / 8) * 64 > K)
            return -2;
        bn.rows_per_group = M / groups;
    } else {
        groups = 1;
    }
    int nb = ddl_dense1_bce_blocks(M / groups);
    dim3 grid(nb, groups);
#define HEAD_ARGS (const uint4*)x, (const bf16_t*)w, bias, label_first, label_second, n_first, M, K, (uint4*)dx, \
                  dw_part, logits, rowloss, dlogit, want_dx, want_dw, bn
    if (fuse) {
#define CALL(A) dense1_bce_kernel<4, true, A><<<grid, HEAD_ROWS * 64, 0, s>>>(HEAD_ARGS)
        ACT1_DISPATCH(bn_act, CALL);
#undef CALL
    }
    else if (K <= 4 * 64 * 8) dense1_bce_kernel<4, false><<<grid, HEAD_ROWS * 64, 0, s>>>(HEAD_ARGS);
    else dense1_bce_kernel<HEAD_VMAX, false><<<grid, HEAD_ROWS * 64, 0, s>>>(HEAD_ARGS);
#undef HEAD_ARGS
    if (want_dw) {
        const int fb = (K + 31) / 32;
        head_finalize_kernel<<<fb, 256, 0, s>>>(dw_part, nb * groups, K, dlogit, M, dw, db, rowloss,
                                                fb > 1 ? loss_out : nullptr);
        if (fb > 1) return 0;
    }
    if (loss_out) mean_kernel<<<1, LB, 0, s>>>(rowloss, M, 1.0 / (double)M, loss_out);   // null: the caller defers it
    return 0;
}

// ------------------------------------------------------------------------------ metrics
// count rows where argmax(pred[r][:N]) == argmax(onehot[r][:N]); pred fp32 [B][N]
__global__ void argmax_match_kernel(const float* __restrict__ pred, const float* __restrict__ onehot, int B,
                                    int N, int* __restrict__ count) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    int hit = 0;
    if (r < B) {
        int ap = 0, at = 0;
        float bp = pred[(size_t)r * N], bt = onehot[(size_t)r * N];
        for (int j = 1; j < N; ++j) {
            float p = pred[(size_t)r * N + j], t = onehot[(size_t)r * N + j];
            if (p > bp) { bp = p; ap = j; }
            if (t > bt) { bt = t; at = j; }
        }
        hit = ap == at;
    }
    unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

DDL_API void ddl_argmax_match(const float* pred, const float* onehot, int B, int N, int* count, hipStream_t s) {
    hipMemsetAsync(count, 0, sizeof(int), s);
    argmax_match_kernel<<<(B + 255) / 256, 256, 0, s>>>(pred, onehot, B, N, count);
}

// ------------------------------------------------------------------------------ optimizers
// One launch updates every parameter of a network (flat fp32 master P, grads G, moments) and
// emits the bf16 working copy.  `step` lives in device memory and is incremented by the kernel
// launch wrapper's tiny pre-kernel so hipGraph replays keep advancing the bias correction.
// kind: 0 adam, 1 sgd, 2 rmsprop, 3 adam_fresh_double.
// If `avg` != nullptr the result is blended in place: p = (1-avg_w)*p_new + avg_w*avg[i]
// (periodic weight averaging folded into the update, [R-ns]).
// A one-thread launch advances the step counter.  (Folding the increment into optim_kernel through a
// "last block to finish" atomic was measured slower: ~1000 blocks x one returning atomic on a single
// word costs ~12 us at ~88 atomics/us, more than this launch.)
__global__ void step_inc_kernel(int* step) { step[0] += 1; }

__global__ void optim_kernel(float4* __restrict__ P, const float4* __restrict__ G, float4* __restrict__ Mo,
                             float4* __restrict__ Vo, uint2* __restrict__ Pc, int* __restrict__ step,
                             long n4, int kind, float lr, float b1, float b2, float eps, float rho,
                             float gscale, const float4* __restrict__ avg, float avg_w) {
    // step = {completed updates, ticket}.  Every block reads the counter before it finishes; the LAST block to
    // finish advances it (no one-thread kernel in front of every optimizer launch).
    const OptimHyper h = optim_prepare(kind, lr, b1, b2, eps, rho, step, 1);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 g4 = G[i];
        g4.x *= gscale; g4.y *= gscale; g4.z *= gscale; g4.w *= gscale;
        optim_update4(h, P, Mo, Vo, Pc, i, g4, avg, avg_w);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(step + 1, 1) == (int)gridDim.x - 1) {
            step[1] = 0;
            atomicAdd(step, 1);
        }
    }
}

// Advance a device-side step counter by itself (the fused all-reduce + update kernel of csrc/comm reads it).
DDL_API void ddl_step_inc(int* step, hipStream_t s) { step_inc_kernel<<<1, 1, 0, s>>>(step); }

DDL_API void ddl_optim_step(float* P, const float* G, float* M, float* V, void* Pc, int* step, long n, int kind,
                            float lr, float b1, float b2, float eps, float rho, float gscale, const float* avg,
                            float avg_w, hipStream_t s) {
    long n4 = n / 4;
    long g = (n4 + 255) / 256;
    // every block ends with one atomic on the ticket word (~10 ns each, serialised): keep the grid small
    static const long cap = getenv("DDL_OPT_BLOCKS") ? atol(getenv("DDL_OPT_BLOCKS")) : 256;
    int grid = (int)(g > cap ? cap : (g < 1 ? 1 : g));
    optim_kernel<<<grid, 256, 0, s>>>((float4*)P, (const float4*)G, (float4*)M, (float4*)V, (uint2*)Pc, step, n4,
                                      kind, lr, b1, b2, eps, rho, gscale, (const float4*)avg, avg_w);
}

// In-place weight average after an all-reduce(sum): P = sum / world, plus bf16 emit.
__global__ void scale_emit_kernel(float4* __restrict__ P, uint2* __restrict__ Pc, long n4, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 p = P[i];
        p.x *= scale; p.y *= scale; p.z *= scale; p.w *= scale;
        P[i] = p;
        if (Pc) Pc[i] = make_uint2(pack2bf(p.x, p.y), pack2bf(p.z, p.w));
    }
}

DDL_API void ddl_scale_emit(float* P, void* Pc, long n, float scale, hipStream_t s) {
    long n4 = n / 4;
    long g = (n4 + 255) / 256;
    int grid = (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
    scale_emit_kernel<<<grid, 256, 0, s>>>((float4*)P, (uint2*)Pc, n4, scale);
}

// ------------------------------------------------------------------------------ flat fp32 helpers
// dst += src (sum of per-slice weight gradients when a batch is split into several launches) and dst = 0: the
// framework's own launches, so no torch elementwise / fill kernel sits on the device path.
__global__ void add_f32_kernel(float4* __restrict__ dst, const float4* __restrict__ src, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = dst[i];
        const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}
__global__ void add_f32_tail_kernel(float* __restrict__ dst, const float* __restrict__ src, long lo, long n) {
    const long i = lo + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

DDL_API void ddl_add_f32(float* dst, const float* src, long n, hipStream_t s) {
    const long n4 = (((uintptr_t)dst | (uintptr_t)src) & 15) ? 0 : n / 4;
    if (n4 > 0) {
        long g = (n4 + 255) / 256;
        add_f32_kernel<<<(int)(g > 2048 ? 2048 : g), 256, 0, s>>>((float4*)dst, (const float4*)src, n4);
    }
    const long rest = n - n4 * 4;
    if (rest > 0) add_f32_tail_kernel<<<(int)((rest + 255) / 256), 256, 0, s>>>(dst, src, n4 * 4, n);
}

DDL_API int ddl_zero_bytes(void* dst, long bytes, hipStream_t s) {
    return hipMemsetAsync(dst, 0, (size_t)bytes, s) == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------ Philox noise
// Philox4x32-10, counter = (element group index, stream offset), key = seed.  The 64-bit offset
// lives in device memory (state[1]) and is advanced by a 1-thread kernel after each draw, so a
// captured graph produces fresh noise on every replay.
__device__ __forceinline__ void philox_round(uint32_t* c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// state = {seed, offset, scratch}: the last block to finish advances the offset (see last_block_increments)
__global__ void philox_uniform_kernel(bf16_t* __restrict__ out, long n, long* __restrict__ state,
                                      float lo, float hi) {
    const uint64_t seed = (uint64_t)state[0], off = (uint64_t)state[1];
    const long n4 = (n + 3) / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)off, (uint32_t)(off >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long e = i * 4 + j;
            if (e < n) {
                float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
                out[e] = f2bf(lo + (hi - lo) * u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* done = (unsigned long long*)(state + 2);
        if (atomicAdd(done, 1ULL) == (unsigned long long)gridDim.x - 1ULL) {
            *done = 0ULL;
            atomicAdd((unsigned long long*)(state + 1), 1ULL);
        }
    }
}

DDL_API void ddl_philox_uniform(void* out, long n, long* state, float lo, float hi, hipStream_t s) {
    long n4 = (n + 3) / 4;
    long g = (n4 + 255) / 256;
    int grid = (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
    philox_uniform_kernel<<<grid, 256, 0, s>>>((bf16_t*)out, n, state, lo, hi);
}

// Delayed (overlapped) periodic averaging, local-SGD style [R-ns]: while ranks kept training, an
// all-reduce summed the snapshot `snap` taken k steps ago into `sum`.  Fold the correction in:
// P += sum/world - snap, and re-emit the bf16 working copy.
__global__ void avg_delta_kernel(float4* __restrict__ P, uint2* __restrict__ Pc, const float4* __restrict__ sum,
                                 const float4* __restrict__ snap, long n4, float inv_world) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 p = P[i], s = sum[i], o = snap[i];
        p.x += s.x * inv_world - o.x; p.y += s.y * inv_world - o.y;
        p.z += s.z * inv_world - o.z; p.w += s.w * inv_world - o.w;
        P[i] = p;
        if (Pc) Pc[i] = make_uint2(pack2bf(p.x, p.y), pack2bf(p.z, p.w));
    }
}

DDL_API void ddl_avg_delta(float* P, void* Pc, const float* sum, const float* snap, long n, float inv_world,
                           hipStream_t s) {
    long n4 = n / 4;
    long g = (n4 + 255) / 256;
    int grid = (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
    avg_delta_kernel<<<grid, 256, 0, s>>>((float4*)P, (uint2*)Pc, (const float4*)sum, (const float4*)snap, n4,
                                          inv_world);
}
