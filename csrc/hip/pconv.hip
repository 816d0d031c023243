// Patch-resident implicit-GEMM convolution (gfx950, v_mfma_f32_16x16x32_bf16).
//
// Same problem statement as igemm_conv.hip (fprop, sub-pixel dgrad), different data movement.
// igemm_conv re-fetches the pixel operand from global memory for every tap, so with the small
// channel counts of these networks (N = 8..64) it is bound by L2->LDS traffic and by gather
// address arithmetic.  Here a block owns a *unit* -- IMG whole images, or a band of rows of one
// image -- and stages the source patch (with a zero halo) into LDS exactly once; the MFMA pixel
// fragments of every tap and every sub-pixel class are then read straight out of that patch
// (address = pixel base + per-k-unit offset), so global traffic drops to (input once + weights per
// pass).  Weights stream through a register-staged, double-buffered LDS stage of 64 k-elements,
// their global loads running three stages ahead of the MFMAs.
//
// All index arithmetic of the K loop is table driven: the host plan (ops/hip.py: _pconv_plan)
// expands every class into stages of eight 8-element k-units and stores, per unit, the patch byte
// offset of its tap/channel and the element offset inside a weight row.  The inner loop is then
// one table read, FR address adds, FR + NT ds_read_b128 and FR*NT MFMAs per 32-deep sub-step.
//
// LDS image of the patch: [img][prow][pcol] pixels of `pitch` bytes (Cs*2 data + pad).  A fragment
// read is one ds_read_b128 per lane at pixel(fr)*pitch + k-chunk(fq)*16; its 16-lane groups are
// conflict-free when (src stride x pitch/16) == 2 (mod 4) (the plan picks the pitch accordingly).
#include "common.h"

struct PClass {
    int ph, pw, Hc, Wc;
    int nstages, stage_off;
    float inv_Wc, pad;
};

struct PConvArgs {
    const bf16_t* src; const bf16_t* wgt; const float* bias; bf16_t* dst;
    const PClass* classes; const int2* ktab;     // ktab[(stage)*8 + unit] = {patch byte offset, weight elem offset | -1}
    int n_classes, total_stages;
    int B, Hs, Ws, Cs;
    int Hd, Wd, Nd, Nw;
    int ssh, ssw, sdh, sdw;
    int wrow;                      // elements per weight row (Ttot * Cs)
    int act; float alpha;
    int IMG, BR, bands;            // unit = IMG images x BR class-grid rows
    int omin_h, omin_w, PH, PW;    // patch origin (min tap offset) and extent for a full band
    int pitch, img_bytes;          // bytes per patch pixel / per patch image
    int w_off, k_off, e_off;       // LDS offsets: weight stages, k-unit table, epilogue scratch
    float inv_per_img, inv_cs8, inv_PW;   // reciprocals for the staging loop's index split
    // STATS instantiation: per-block partial sums of the (bf16-rounded) outputs and their squares, laid out
    // part[2][Nd][gridDim.x] like the column-reduction kernels' partials -- the BatchNorm that follows reads
    // its batch statistics from here instead of making a pass over the output tensor.
    float* stats; int stats_stride, stats_off;     // partial column = stats_off + block; row pitch stats_stride
    // MULY instantiation: dst = bf16(bf16(conv) * (y > 0 ? 1 : mul_slope)) -- linear / relu / leaky of the layer below,
    // the same two roundings as the separate act_bwd pass (bit-identical)
    const bf16_t* mul_y; float mul_slope;
    // AFFINE instantiation: an inference-phase BatchNorm right behind this layer, folded into the epilogue:
    // out = act(conv + bias) * k + (beta - moving_mean * k), k = gamma * rsqrt(moving_var + eps)
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var; float bn_eps;
    // optional per-channel affine on the SOURCE as the patch is staged (a BatchNorm in front of this layer that is
    // never materialised): src' = bf16(src * in_scale[grp][c] + in_shift[grp][c]) on real pixels, grp =
    // image / in_group_rows; halo and missing images stay zero
    const float* in_scale; const float* in_shift; int in_group_rows, in_ctrue;
};

// floor(x / d) for 0 <= x < 2^22 via a float reciprocal (the +0.5 keeps x/d away from integers)
__device__ __forceinline__ int fdiv(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define OOB 0xFFFFFFF0u            // a byte offset beyond any descriptor: the buffer load returns zeros

#define WPITCH 160                 // weight stage row: 64 k-elements (128 B) + 32 B pad  (P = 10 == 2 mod 4)

// SLOW = the activation is one of the transcendental ones (sigmoid / softplus / tanh / softsign /
// tanh01): only that instantiation carries their (fully unrolled) code.
// EPI: 0 = plain epilogue, 1 = STATS (batch-statistics partials for a following BN), 2 = AFFINE (inference BN folded in),
// 3 = MULY (result *= act'(y) of the layer below, y laid out like dst: the activation backward fused into an input gradient)
// NW = waves per block (4 | 8).  The patch in LDS limits a CU to one or two blocks, and every phase of a block (staging,
// per class: weight prologue, stage loop, epilogue) is a chain of dependent latencies that only OTHER waves can fill:
// eight waves on the same patch and the same weight stages double the waves per CU at the same LDS footprint.
template <int BN, int FR, bool SLOW, int EPI, int NW>
__global__ __launch_bounds__(NW * 64, (BN * FR * NW >= 2048 ? 1 : 2)) void pconv_kernel(const PConvArgs a) {
    constexpr bool STATS = EPI == 1, AFFINE = EPI == 2, MULY = EPI == 3;
    constexpr int NTHR = NW * 64;
    constexpr int NT = BN / 16;                        // 16-channel MFMA tiles per wave
    constexpr int RW = (BN * 8 + NTHR - 1) / NTHR;     // 16-B weight units staged per thread per stage
    constexpr int WROWS = NTHR / 8;                    // weight rows staged per round of the block's threads
    constexpr int PASS = NW * FR * 16;                 // rows per pass (NW waves x FR fragments x 16 px)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                                   // patch
    char* sW = smem + a.w_off;                         // [2][BN][WPITCH]
    const int2* sK = (const int2*)(smem + a.k_off);    // k-unit table

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = blockIdx.x;
    const int grp = unit / a.bands, band = unit - grp * a.bands;
    const int b0 = grp * a.IMG;
    const int i0 = band * a.BR;                        // first class-grid row of this band
    const int n0 = blockIdx.y * BN;
    const int Cs = a.Cs, cs8 = Cs >> 3;
    // Bounds-checked buffer loads keep every gather branch-free (a predicated global_load compiles to an
    // exec-masked branch, after which the compiler can no longer count outstanding loads and drains
    // them all with vmcnt(0) -- that would serialise the weight pipeline below).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((unsigned)a.B * a.Hs * a.Ws * Cs * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wgt, 0, (int)((unsigned)a.Nw * a.wrow * 2u), 0x00020000);

    // ---- stage the patch: every 16-B unit is either source data or zero (halo / missing image)
    {
        const int row0 = i0 * a.ssh + a.omin_h, col0 = a.omin_w;
        const int per_img = a.PH * a.PW * cs8;
        const int total = a.IMG * per_img;
        // eight independent 16-byte loads in flight per thread before the first LDS store
        for (int base = tid; base < total; base += NTHR * 8) {
            u32x4 v[8];
            int dst[8], aoff[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int u = base + k * NTHR;
                bool in = u < total;
                int uu = in ? u : 0;
                int img = fdiv(uu, a.inv_per_img), r = uu - img * per_img;
                int pix = fdiv(r, a.inv_cs8), c8 = r - pix * cs8;
                int pr = fdiv(pix, a.inv_PW), pc = pix - pr * a.PW;
                int sr = row0 + pr, sc = col0 + pc, b = b0 + img;
                bool ok = in & (b < a.B) & ((unsigned)sr < (unsigned)a.Hs) & ((unsigned)sc < (unsigned)a.Ws);
                unsigned off = ok ? (unsigned)(((b * a.Hs + sr) * a.Ws + sc) * Cs + c8 * 8) * 2u : OOB;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
                dst[k] = in ? img * a.img_bytes + pix * a.pitch + c8 * 16 : -1;
                aoff[k] = ok ? ((b << 8) | c8) : -1;            // decoded only when an input affine is present
            }
            if (a.in_scale) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (aoff[k] < 0) continue;
                    const int cb = (aoff[k] & 255) * 8;
                    const int ao = ((aoff[k] >> 8) / a.in_group_rows) * Cs + cb;
                    const float4 sa = *(const float4*)(a.in_scale + ao), sb = *(const float4*)(a.in_scale + ao + 4);
                    const float4 ha = *(const float4*)(a.in_shift + ao), hb = *(const float4*)(a.in_shift + ao + 4);
                    const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
                    const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
                    float f[8];
                    unpack8(make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (cb + e < a.in_ctrue) ? fmaf(f[e], sc[e], sh[e]) : 0.f;
                    const uint4 r = pack8(f);
                    v[k] = (u32x4){r.x, r.y, r.z, r.w};
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (dst[k] >= 0) *(u32x4*)(sP + dst[k]) = v[k];
        }
        int2* sKw = (int2*)(smem + a.k_off);
        for (int t = tid; t < a.total_stages * 8; t += NTHR) sKw[t] = a.ktab[t];
    }
    __syncthreads();

    // weight staging assignment: k-unit wq of rows wr0 + 32*r
    const int wq = tid & 7, wr0 = tid >> 3;
    unsigned wbase[RW];                                // byte offset of the weight row, OOB if there is none
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int nl = wr0 + WROWS * r, n = n0 + nl;
        wbase[r] = (nl < BN && n < a.Nw) ? (unsigned)(n * a.wrow) * 2u : OOB;
    }

    // this lane's bias values (channels n0 + 16t + 4fq .. +3); channels >= Nw and a null bias read as zero.
    // Small tiles keep them in registers for the whole kernel, the big ones reload them per epilogue.
    constexpr bool HOIST_BIAS = BN * FR < 512;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.bias, 0, a.bias ? a.Nw * 4 : 0, 0x00020000);
    f32x4 bv[NT];
    auto load_bias = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned nb = (unsigned)(n0 + 16 * t + 4 * fq) * 4u;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                bv[t][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, nb + 4u * e, 0, 0));
        }
    };
    if (HOIST_BIAS) load_bias();

    float bnk[AFFINE ? NT : 1][4], bnh[AFFINE ? NT : 1][4];
    if (AFFINE) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + 16 * t + 4 * fq + e;
                float k = 0.f, h = 0.f;
                if (n < a.Nw) {
                    k = a.bn_gamma[n] * rsqrtf(a.bn_var[n] + a.bn_eps);
                    h = a.bn_beta[n] - a.bn_mean[n] * k;
                }
                bnk[t][e] = k;
                bnh[t][e] = h;
            }
    }
    float st_s[STATS ? NT : 1][4], st_q[STATS ? NT : 1][4];
    if (STATS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) st_s[t][e] = st_q[t][e] = 0.f;
    }

    for (int ci = 0; ci < a.n_classes; ++ci) {
        const PClass cd = a.classes[ci];
        const int nrow = min(i0 + a.BR, cd.Hc) - i0;
        if (nrow <= 0) continue;
        const int per_img_rows = nrow * cd.Wc;
        const float inv_rows = 1.0f / (float)per_img_rows;
        const int Mloc = a.IMG * per_img_rows;
        const int nstages = cd.nstages;
        const int2* ck = sK + cd.stage_off * 8;

        for (int ms = 0; ms < Mloc; ms += PASS) {
            // ---- this lane's FR pixels (one per 16-row fragment)
            int pixbase[FR], pixdst[FR];           // patch byte offset / destination pixel (-1: no such row) of the lane's pixels
#pragma unroll
            for (int j = 0; j < FR; ++j) {
                int m = ms + (wave * FR + j) * 16 + fr;
                int mm = m < Mloc ? m : 0;
                int img = fdiv(mm, inv_rows), rem = mm - img * per_img_rows;
                int il = fdiv(rem, cd.inv_Wc), jl = rem - il * cd.Wc;
                pixbase[j] = img * a.img_bytes + (il * a.ssh * a.PW + jl * a.ssw) * a.pitch;
                const int b = b0 + img;
                pixdst[j] = (m < Mloc && b < a.B) ? (b * a.Hd + cd.ph + a.sdh * (i0 + il)) * a.Wd + cd.pw + a.sdw * jl : -1;
            }

            u32x4 regA[RW], regB[RW];              // stage s lives in regA (s even) / regB (s odd)
            int wst = 0;                           // next stage to load
            auto gload = [&](u32x4 (&reg)[RW]) {
                const int kv = ck[min(wst, nstages - 1) * 8 + wq].y;
                const bool live = (wst < nstages) & (kv >= 0);
                ++wst;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    unsigned off = (live & (wbase[r] != OOB)) ? wbase[r] + (unsigned)kv * 2u : OOB;
                    reg[r] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
                }
            };
            auto lstore = [&](int buf, const u32x4 (&reg)[RW]) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    int nl = wr0 + WROWS * r;
                    if (BN % WROWS == 0 || nl < BN) *(u32x4*)(sW + (buf * BN + nl) * WPITCH + wq * 16) = reg[r];
                }
            };

            f32x4 acc[NT][FR];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < FR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

            auto compute = [&](int cur, int st) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ka = ck[st * 8 + s * 4 + fq].x;
                    bf16x8_t wf[NT], xf[FR];
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        wf[i] = *(const bf16x8_t*)(sW + (cur * BN + 16 * i + fr) * WPITCH + s * 64 + fq * 16);
#pragma unroll
                    for (int j = 0; j < FR; ++j) xf[j] = *(const bf16x8_t*)(sP + pixbase[j] + ka);
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < FR; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                }
            };

            // Weight pipeline: loads run three stages ahead (two register sets in flight), the LDS stage
            // is double-buffered, one barrier per stage.  Loads past the last stage are zeros, never stored.
            gload(regA);                 // stage 0
            // the previous segment's epilogue used the weight stages as its transpose scratch: every wave must be
            // out of it before stage 0 lands there
            __syncthreads();
            lstore(0, regA);
            gload(regB);                 // stage 1
            gload(regA);                 // stage 2
            __syncthreads();
            for (int st = 0; st < nstages; st += 2) {       // nstages is even (the plan pads with a zero stage)
                lstore(1, regB);                            // stage st+1
                gload(regB);                                // stage st+3
                compute(0, st);
                __syncthreads();
                lstore(0, regA);                            // stage st+2 (zeros past the end; never read)
                gload(regA);                                // stage st+4
                compute(1, st + 1);
                __syncthreads();
            }

            // The weight pipeline's last loads (past the final stage: zeros, never stored) are still in flight.  Drain
            // them here, where they are old, so the compiler does not have to protect their destination registers
            // with a vmcnt(0) in the middle of the epilogue -- that wait would also drain the fragment stores.
            __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) only
            // ---- epilogue.  A lane holds channels 16t + 4fq .. +3 of pixel 16j + fr; written directly
            // that is 8-byte stores in 32-byte runs.  Each 16-pixel fragment is instead transposed
            // through a per-wave LDS scratch and leaves as 16-byte stores covering every pixel's whole
            // channel run.  Bias is hoisted, the activation switch is outside the element loops.
            {
                constexpr int EROW = BN * 2 + 16;
                char* sE = smem + a.e_off + wave * (16 * EROW + 64);
                int* sPix = (int*)(sE + 16 * EROW);
                if (!HOIST_BIAS) load_bias();
                const int act = a.act;
                const float alpha = a.alpha;
                if (SLOW) {
                    // fully unrolled: runtime-indexed accumulators would be demoted to scratch memory
                    act_switch(act, [&](auto KA) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int j = 0; j < FR; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    acc[t][j][e] = act_apply_k<decltype(KA)::value>(acc[t][j][e] + bv[t][e], act, alpha);
                    });
                } else if (act == ACT_LINEAR) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < FR; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[t][j][e] += bv[t][e];
                } else if (act == ACT_RELU) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < FR; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[t][j][e] = fmaxf(acc[t][j][e] + bv[t][e], 0.f);
                } else {                                    // ACT_LEAKY
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < FR; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = acc[t][j][e] + bv[t][e];
                                acc[t][j][e] = x > 0.f ? x : alpha * x;
                            }
                }
                if (AFFINE) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < FR; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[t][j][e] = fmaf(acc[t][j][e], bnk[t][e], bnh[t][e]);
                }
                // MULY: this lane's y vectors of every fragment it will store, fetched now so that their latency is
                // covered by the transposes (fetched inside the store loop they cost what the separate pass cost)
                constexpr int UPL = (16 * (BN / 8) + 63) / 64;      // 16-byte store units per lane per fragment
                uint4 ymul[MULY ? FR : 1][MULY ? UPL : 1];
                if (MULY) {
#pragma unroll
                    for (int j = 0; j < FR; ++j)
#pragma unroll
                        for (int k = 0; k < UPL; ++k) {
                            const int u = lane + 64 * k;
                            const int px = u / (BN / 8), c8 = u - px * (BN / 8);
                            const int m = ms + (wave * FR + j) * 16 + px;
                            const int mm = m < Mloc ? m : 0;
                            const int img = fdiv(mm, inv_rows), rem = mm - img * per_img_rows;
                            const int il = fdiv(rem, cd.inv_Wc), jl = rem - il * cd.Wc;
                            const int b = b0 + img;
                            const bool ok = u < 16 * (BN / 8) && m < Mloc && b < a.B && n0 + c8 * 8 < a.Nd;
                            const long pix = (long)(b * a.Hd + cd.ph + a.sdh * (i0 + il)) * a.Wd + cd.pw + a.sdw * jl;
                            ymul[j][k] = ok ? *(const uint4*)(a.mul_y + pix * a.Nd + n0 + c8 * 8) : make_uint4(0, 0, 0, 0);
                        }
                }
                // channels >= Nw (padding lanes) must be exactly zero
                const int nlim = a.Nw - n0 - 4 * fq;       // channel 16t + e is real iff 16t + e < nlim
#pragma unroll
                for (int j = 0; j < FR; ++j) {
                    const bool ok = pixdst[j] >= 0;
                    if (fq == 0) sPix[fr] = pixdst[j];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (16 * t + e < nlim) ? acc[t][j][e] : 0.f;
                        const uint32_t w0 = pack2bf(v[0], v[1]), w1 = pack2bf(v[2], v[3]);
                        *(uint2*)(sE + fr * EROW + (16 * t + 4 * fq) * 2) = make_uint2(w0, w1);
                        if (STATS) {
                            // what the stored tensor holds: the packed words, unpacked again
                            const float r[4] = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u),
                                                __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float q = ok ? r[e] : 0.f;
                                st_s[t][e] += q;
                                st_q[t][e] = fmaf(q, q, st_q[t][e]);
                            }
                        }
                    }
                    // The scratch is private to this wave and a wave's LDS operations execute in order, so ordering the
                    // compiler (wavefront-scope fences) is all the transpose needs.  Workgroup-scope fences also drain the
                    // global stores of the previous fragment (vmcnt(0)) and serialised the store latency per fragment.
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int k = 0; k < UPL; ++k) {
                        const int u = lane + 64 * k;
                        if (u >= 16 * (BN / 8)) break;
                        int px = u / (BN / 8), c8 = u - px * (BN / 8);
                        int p = sPix[px];
                        int n = n0 + c8 * 8;
                        if (p >= 0 && n < a.Nd) {
                            uint4 o = *(const uint4*)(sE + px * EROW + c8 * 16);
                            if (MULY) {
                                float f[8], yy[8];
                                unpack8(o, f);
                                unpack8(ymul[j][k], yy);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] *= yy[e] > 0.f ? 1.f : a.mul_slope;
                                o = pack8(f);
                            }
                            *(uint4*)(a.dst + (long)p * a.Nd + n) = o;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    }

    if (STATS) {
        // lanes sharing fq hold the same channels for 16 different pixels: fold them, then the four waves
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    st_s[t][e] += __shfl_xor(st_s[t][e], o, 64);
                    st_q[t][e] += __shfl_xor(st_q[t][e], o, 64);
                }
        __syncthreads();                                   // every wave is done with the weight stages
        float* red = (float*)sW;                           // [NW waves][BN][2]
        if (fr == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 16 * t + 4 * fq + e;
                    red[(wave * BN + c) * 2 + 0] = st_s[t][e];
                    red[(wave * BN + c) * 2 + 1] = st_q[t][e];
                }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Nd) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { s0 += red[(w * BN + tid) * 2]; s1 += red[(w * BN + tid) * 2 + 1]; }
            const size_t nb = a.stats_stride, col = (size_t)a.stats_off + blockIdx.x;
            a.stats[((size_t)(n0 + tid)) * nb + col] = s0;
            a.stats[((size_t)(a.Nd + n0 + tid)) * nb + col] = s1;
        }
    }
}

template <int BN, int FR, bool SLOW, int EPI, int NW>
static int launch_pconv2(PConvArgs& a, int units, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipFuncSetAttribute((const void*)pconv_kernel<BN, FR, SLOW, EPI, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
            done = true;
        }
    }
    dim3 grid(units, (a.Nd + BN - 1) / BN);
    pconv_kernel<BN, FR, SLOW, EPI, NW><<<grid, NW * 64, smem, s>>>(a);
    return 0;
}

template <int BN, int FR, int NW>
static int launch_pconv(PConvArgs& a, int units, size_t smem, hipStream_t s) {
    const bool fast = a.act == ACT_LINEAR || a.act == ACT_RELU || a.act == ACT_LEAKY;
    if (a.stats)
        return fast ? launch_pconv2<BN, FR, false, 1, NW>(a, units, smem, s) : launch_pconv2<BN, FR, true, 1, NW>(a, units, smem, s);
    if (a.mul_y)
        return fast ? launch_pconv2<BN, FR, false, 3, NW>(a, units, smem, s) : -4;
    if (a.bn_gamma)
        return fast ? launch_pconv2<BN, FR, false, 2, NW>(a, units, smem, s) : launch_pconv2<BN, FR, true, 2, NW>(a, units, smem, s);
    return fast ? launch_pconv2<BN, FR, false, 0, NW>(a, units, smem, s) : launch_pconv2<BN, FR, true, 0, NW>(a, units, smem, s);
}

// bytes of the weight-stage area: two stages, and the epilogue's per-wave transpose scratch that aliases them
static long pconv_wbytes(int BN, int NW) {
    const long w = 2L * BN * WPITCH, e = (long)NW * (16 * (BN * 2 + 16) + 64);
    return ((w > e ? w : e) + 15) / 16 * 16;
}

DDL_API int ddl_pconv_lds_bytes(int BN, int total_stages, long patch_bytes, int NW) {
    long w = pconv_wbytes(BN, NW), k = (long)total_stages * 64;
    long p = (patch_bytes + 15) / 16 * 16;
    long tot = p + w + k;
    return tot > (1L << 30) ? -1 : (int)tot;
}

// The plan (unit decomposition, patch geometry, k-unit table) is made by the caller; this only
// validates it against the LDS limit and launches.  Returns 0 on success, negative on a bad plan.
DDL_API int ddl_pconv(const void* src, const void* wgt, const float* bias, void* dst, const void* classes,
                      const void* ktab, int n_classes, int total_stages, int B, int Hs, int Ws, int Cs, int Hd,
                      int Wd, int Nd, int Nw, int ssh, int ssw, int sdh, int sdw, int Ttot, int act, float alpha,
                      int IMG, int BR, int bands, int omin_h, int omin_w, int PH, int PW, int pitch, int BN, int FR,
                      int NW, float* stats, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                      const float* bn_var, float bn_eps, const float* in_scale, const float* in_shift,
                      int in_group_rows, int in_ctrue, const void* mul_y, float mul_slope, int stats_stride,
                      int stats_off, hipStream_t s) {
    if (Cs % 8 != 0 || Cs <= 0 || Nd <= 0 || Nd % 8 != 0 || n_classes <= 0 || B <= 0 || IMG <= 0 || BR <= 0) return -1;
    if (pitch % 16 != 0 || pitch < Cs * 2) return -1;
    if ((long)B * Hs * Ws * Cs >= (1L << 31) || (long)B * Hd * Wd * Nd >= (1L << 31)) return -3;
    if ((long)Nw * Ttot * Cs >= (1L << 31)) return -3;
    PConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.dst = (bf16_t*)dst;
    a.classes = (const PClass*)classes; a.ktab = (const int2*)ktab;
    a.stats = stats;
    // partials of several launches (batch slices) side by side in one [2][Nd][stats_stride] array; 0 = this launch alone
    a.stats_stride = stats_stride > 0 ? stats_stride : ((B + IMG - 1) / IMG) * bands;
    a.stats_off = stats_stride > 0 ? stats_off : 0;
    if (stats && (a.stats_off < 0 || a.stats_off + ((B + IMG - 1) / IMG) * bands > a.stats_stride)) return -1;
    if (stats && bn_gamma) return -1;
    if (mul_y && (stats || bn_gamma)) return -1;
    a.mul_y = (const bf16_t*)mul_y; a.mul_slope = mul_slope;
    a.bn_gamma = bn_gamma; a.bn_beta = bn_beta; a.bn_mean = bn_mean; a.bn_var = bn_var; a.bn_eps = bn_eps;
    if (in_scale && (!in_shift || in_group_rows <= 0 || Cs > 2048 || B >= (1 << 23))) return -1;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_group_rows = in_scale ? in_group_rows : 1; a.in_ctrue = in_ctrue;
    a.n_classes = n_classes; a.total_stages = total_stages;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cs = Cs; a.Hd = Hd; a.Wd = Wd; a.Nd = Nd; a.Nw = Nw;
    a.ssh = ssh; a.ssw = ssw; a.sdh = sdh; a.sdw = sdw; a.wrow = Ttot * Cs; a.act = act; a.alpha = alpha;
    a.IMG = IMG; a.BR = BR; a.bands = bands; a.omin_h = omin_h; a.omin_w = omin_w; a.PH = PH; a.PW = PW;
    a.pitch = pitch; a.img_bytes = PH * PW * pitch;
    long patch = (long)IMG * a.img_bytes;
    if ((long)IMG * PH * PW * (Cs / 8) >= (1L << 22)) return -2;
    a.w_off = (int)((patch + 15) / 16 * 16);
    if (NW != 4 && NW != 8) return -4;
    a.k_off = a.w_off + (int)pconv_wbytes(BN, NW);
    a.e_off = a.w_off;                    // epilogue scratch aliases the weight stages (see the kernel)
    size_t smem = (size_t)a.k_off + (size_t)total_stages * 64;
    if (smem > 160 * 1024) return -2;
    a.inv_per_img = 1.0f / (float)(PH * PW * (Cs / 8));
    a.inv_cs8 = 1.0f / (float)(Cs / 8);
    a.inv_PW = 1.0f / (float)PW;
    int units = ((B + IMG - 1) / IMG) * bands;
#define PC_CASE(BN_, FR_, NW_) if (BN == BN_ && FR == FR_ && NW == NW_) return launch_pconv<BN_, FR_, NW_>(a, units, smem, s);
    PC_CASE(16, 4, 4) PC_CASE(16, 8, 4) PC_CASE(32, 2, 4) PC_CASE(32, 4, 4) PC_CASE(32, 8, 4) PC_CASE(64, 2, 4)
    PC_CASE(64, 4, 4) PC_CASE(64, 8, 4) PC_CASE(128, 4, 4)
    PC_CASE(32, 2, 8) PC_CASE(32, 4, 8) PC_CASE(64, 2, 8) PC_CASE(64, 4, 8)
#undef PC_CASE
    return -4;
}
