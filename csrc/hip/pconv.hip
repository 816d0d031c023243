// Patch-resident implicit-GEMM convolution (gfx950, v_mfma_f32_16x16x32_bf16).
//
// Same problem statement and class/tap tables as igemm_conv.hip (fprop, sub-pixel dgrad, Dense),
// different data movement.  igemm_conv re-fetches the pixel operand from global memory for every
// tap, so with the small channel counts of these networks (N = 8..128) it is bound by L2->LDS
// traffic (M x K x 2 B per launch: 0.6 GB for one DCGAN ConvTranspose at B = 1000).  Here a block
// owns a *unit* -- IMG whole images, or a band of rows of one image -- and stages the source patch
// (with a zero halo) into LDS exactly once; the MFMA pixel fragments of every tap and every
// sub-pixel class are then read straight out of that patch (address = pixel base + tap offset),
// so global traffic drops to (input once + weights per 256-row pass) and the kernel runs at the
// LDS/MFMA rate.  Only the weights are streamed through a register-staged, double-buffered LDS
// stage (64 k-elements per barrier).
//
// LDS image of the patch: [img][prow][pcol] pixels of `pitch` bytes (Cs*2 data + pad).  A fragment
// read is one ds_read_b128 per lane at pixel(fr)*pitch + k-chunk(fq)*16; its 16-lane groups are
// conflict-free when (src stride x pitch/16) == 2 (mod 4)  (see pick_pitch()).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

struct PClassDesc { int ph, pw, Hc, Wc, ntaps, tap_off, M, pad; };

struct PConvArgs {
    const bf16_t* src; const bf16_t* wgt; const float* bias; bf16_t* dst;
    const PClassDesc* classes; const int4* taps;     // tap = {offh, offw, wtap, 0}
    int n_classes, total_taps;
    int B, Hs, Ws, Cs;
    int Hd, Wd, Nd, Nw;
    int ssh, ssw, sdh, sdw, Ttot;
    int act; float alpha;
    int IMG, BR, bands;            // unit = IMG images x BR class-grid rows
    int omin_h, omin_w, PH, PW;    // patch origin (min tap offset) and extent for a full band
    int pitch, img_bytes;          // bytes per patch pixel / per patch image
    int w_off, t_off, z_off, e_off; // LDS offsets: weight stages, tap table, zero slot, epilogue scratch
    float inv_per_img, inv_cs8, inv_PW;   // reciprocals for the staging loop's index split
    int dbg;
};

// floor(x / d) for 0 <= x < 2^22 via a float reciprocal (exact: the +0.5 keeps x/d away from integers)
__device__ __forceinline__ int fdiv(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

#define WPITCH 160                 // weight stage row: 64 k-elements (128 B) + 32 B pad  (P = 10 == 2 mod 4)

template <int BN>
__global__ __launch_bounds__(256) void pconv_kernel(const PConvArgs a) {
    constexpr int NT = BN / 16;                        // 16-channel MFMA tiles per wave (wave = 64 px x BN ch)
    constexpr int RW = (BN * 8 + 255) / 256;           // 16-B weight units staged per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                                   // patch
    char* sW = smem + a.w_off;                         // [2][BN][WPITCH]
    int2* sT = (int2*)(smem + a.t_off);                // per tap: {patch byte offset, weight tap index}
    const int z_off = a.z_off;                         // 64 B of zeros

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = blockIdx.x;
    const int grp = unit / a.bands, band = unit - grp * a.bands;
    const int b0 = grp * a.IMG;
    const int i0 = band * a.BR;                        // first class-grid row of this band
    const int n0 = blockIdx.y * BN;
    const int Cs = a.Cs, cs8 = Cs >> 3;

    // ---- stage the patch: every 16-B unit is either source data or zero (halo / missing image)
    {
        const int row0 = i0 * a.ssh + a.omin_h, col0 = a.omin_w;
        const int per_img = a.PH * a.PW * cs8;
        const int total = a.IMG * per_img;
        for (int u = tid; u < total; u += 256) {
            int img = fdiv(u, a.inv_per_img), r = u - img * per_img;
            int pix = fdiv(r, a.inv_cs8), c8 = r - pix * cs8;
            int pr = fdiv(pix, a.inv_PW), pc = pix - pr * a.PW;
            int sr = row0 + pr, sc = col0 + pc, b = b0 + img;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (!(a.dbg & 1) && b < a.B && (unsigned)sr < (unsigned)a.Hs && (unsigned)sc < (unsigned)a.Ws)
                v = *(const uint4*)(a.src + (((long)b * a.Hs + sr) * a.Ws + sc) * Cs + c8 * 8);
            *(uint4*)(sP + img * a.img_bytes + pix * a.pitch + c8 * 16) = v;
        }
        for (int t = tid; t < a.total_taps; t += 256) {
            int4 te = a.taps[t];
            sT[t] = make_int2(((te.x - a.omin_h) * a.PW + (te.y - a.omin_w)) * a.pitch, te.z);
        }
        if (tid < 4) *(uint4*)(smem + z_off + tid * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // weight staging assignment: unit q (8 k-elements) of row wr0 + 32*r
    const int wq = tid & 7, wr0 = tid >> 3;

    for (int ci = 0; ci < a.n_classes; ++ci) {
        const PClassDesc cd = a.classes[ci];
        int nrow = min(i0 + a.BR, cd.Hc) - i0;
        if (nrow <= 0) continue;
        const int per_img_rows = nrow * cd.Wc;
        const int Mloc = a.IMG * per_img_rows;
        const int nstages = (cd.ntaps * Cs + 63) >> 6;

        for (int ms = 0; ms < Mloc; ms += 256) {
            // ---- this lane's four pixels (one per 16-row fragment)
            int pixbase[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int m = ms + wave * 64 + 16 * j + fr;
                int mm = m < Mloc ? m : 0;
                int img = mm / per_img_rows, rem = mm - img * per_img_rows;
                int il = rem / cd.Wc, jl = rem - il * cd.Wc;
                pixbase[j] = img * a.img_bytes + (il * a.ssh * a.PW + jl * a.ssw) * a.pitch;
            }
            // (tap, channel) cursors: two per lane for the fragment reads (sub-steps 0/1), one for staging
            int ftap[2], fc[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int k = s * 32 + fq * 8;
                ftap[s] = k / Cs;
                fc[s] = k - ftap[s] * Cs;
            }
            int wtap = (wq * 8) / Cs, wc = wq * 8 - wtap * Cs;

            uint4 regA[RW], regB[RW];              // stage s lives in regA (s even) / regB (s odd)
            auto gload = [&](uint4* reg) {
                const bool tv = wtap < cd.ntaps;
                const int wt = tv ? sT[cd.tap_off + wtap].y : 0;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    int nl = wr0 + 32 * r, n = n0 + nl;
                    bool ok = tv && nl < BN && n < a.Nw && !(a.dbg & 2);
                    reg[r] = ok ? *(const uint4*)(a.wgt + ((long)n * a.Ttot + wt) * Cs + wc) : make_uint4(0, 0, 0, 0);
                }
                wc += 64;
                while (wc >= Cs) { wc -= Cs; ++wtap; }
            };
            auto lstore = [&](int buf, const uint4* reg) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    int nl = wr0 + 32 * r;
                    if (nl < BN) *(uint4*)(sW + (buf * BN + nl) * WPITCH + wq * 16) = reg[r];
                }
            };

            f32x4 acc[NT][4];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

            auto compute = [&](int cur) {
                if (a.dbg & 4) return;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bool tv = ftap[s] < cd.ntaps;
                    const int koff = tv ? sT[cd.tap_off + ftap[s]].x + fc[s] * 2 : 0;
                    bf16x8_t wf[NT], xf[4];
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        wf[i] = *(const bf16x8_t*)(sW + (cur * BN + 16 * i + fr) * WPITCH + s * 64 + fq * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xf[j] = *(const bf16x8_t*)(sP + (tv ? pixbase[j] + koff : z_off));
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    fc[s] += 64;
                    while (fc[s] >= Cs) { fc[s] -= Cs; ++ftap[s]; }
                }
            };

            // Weight pipeline: global loads run three stages ahead of the MFMAs (two register sets in
            // flight), the LDS stage is double-buffered, one barrier per stage.  Loads past the last
            // stage return zeros (tap >= ntaps) and are never stored.
            gload(regA);                 // stage 0
            lstore(0, regA);
            gload(regB);                 // stage 1
            gload(regA);                 // stage 2
            __syncthreads();
            for (int st = 0; st < nstages; st += 2) {
                if (st + 1 < nstages) lstore(1, regB);      // stage st+1
                gload(regB);                                // stage st+3
                compute(0);
                __syncthreads();
                if (st + 1 >= nstages) break;
                if (st + 2 < nstages) lstore(0, regA);      // stage st+2
                gload(regA);                                // stage st+4
                compute(1);
                __syncthreads();
            }

            // ---- epilogue: lane holds channels 16t + 4fq .. +3 of pixel 16j + fr.  Writing that
            // straight out would be 8-byte stores in 32-byte runs; instead each 16-pixel fragment is
            // transposed through a per-wave LDS scratch and leaves as 16-byte stores covering every
            // pixel's whole channel run (64..256 contiguous bytes per pixel).
            {
                constexpr int EROW = BN * 2 + 16;
                char* sE = smem + a.e_off + wave * (16 * EROW + 64);
                int* sPix = (int*)(sE + 16 * EROW);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int m = ms + wave * 64 + 16 * j + fr;
                    int mm = m < Mloc ? m : 0;
                    int img = mm / per_img_rows, rem = mm - img * per_img_rows;
                    int il = rem / cd.Wc, jl = rem - il * cd.Wc;
                    int b = b0 + img;
                    bool ok = m < Mloc && b < a.B && !(a.dbg & 8);
                    int pix = (b * a.Hd + cd.ph + a.sdh * (i0 + il)) * a.Wd + cd.pw + a.sdw * jl;
                    if (fq == 0) sPix[fr] = ok ? pix : -1;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        int nb = n0 + 16 * t + 4 * fq;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int n = nb + e;
                            float x = acc[t][j][e];
                            if (n < a.Nw) {
                                if (a.bias) x += a.bias[n];
                                v[e] = act_apply(x, a.act, a.alpha);
                            } else {
                                v[e] = 0.f;
                            }
                        }
                        *(uint2*)(sE + fr * EROW + (16 * t + 4 * fq) * 2) =
                            make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                    for (int u = lane; u < 16 * (BN / 8); u += 64) {
                        int px = u / (BN / 8), c8 = u - px * (BN / 8);
                        int p = sPix[px];
                        int n = n0 + c8 * 8;
                        if (p >= 0 && n < a.Nd)
                            *(uint4*)(a.dst + (long)p * a.Nd + n) = *(const uint4*)(sE + px * EROW + c8 * 16);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
            }
        }
    }
}

// smallest pixel pitch (bytes, multiple of 16, >= Cs*2) whose 16-lane ds_read_b128 groups are
// conflict-free for consecutive pixels `stride` apart: (stride * pitch/16) == 2 (mod 4)
static int pick_pitch(int Cs, int stride) {
    int p0 = Cs / 8;
    for (int p = p0; p < p0 + 4; ++p)
        if (((long)stride * p) % 4 == 2) return p * 16;
    return (p0 | 1) * 16;
}

template <int BN>
static int launch_pconv(PConvArgs& a, int units, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipFuncSetAttribute((const void*)pconv_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    dim3 grid(units, (a.Nd + BN - 1) / BN);
    pconv_kernel<BN><<<grid, 256, smem, s>>>(a);
    return 0;
}

// Returns 0 on success; 1 if the problem does not fit the patch scheme (caller falls back to
// ddl_igemm_conv; nothing launched); negative on an argument error.
// omin/omax: min / max tap offsets over every class; maxHc / maxWc: largest class grid.
DDL_API int ddl_pconv(const void* src, const void* wgt, const float* bias, void* dst, const void* classes,
                      const void* taps, int n_classes, int total_taps, int B, int Hs, int Ws, int Cs, int Hd, int Wd,
                      int Nd, int Nw, int ssh, int ssw, int sdh, int sdw, int Ttot, int act, float alpha, int omin_h,
                      int omax_h, int omin_w, int omax_w, int maxHc, int maxWc, int lds_budget, int min_rows, hipStream_t s) {
    if (Cs % 8 != 0 || Cs <= 0 || Nd <= 0 || n_classes <= 0 || B <= 0) return -1;
    if ((long)B * Hs * Ws * Cs >= (1L << 31) || (long)B * Hd * Wd * Nd >= (1L << 31)) return -3;
    PConvArgs a;
    a.src = (const bf16_t*)src; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.dst = (bf16_t*)dst;
    a.classes = (const PClassDesc*)classes; a.taps = (const int4*)taps;
    a.n_classes = n_classes; a.total_taps = total_taps;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cs = Cs; a.Hd = Hd; a.Wd = Wd; a.Nd = Nd; a.Nw = Nw;
    a.ssh = ssh; a.ssw = ssw; a.sdh = sdh; a.sdw = sdw; a.Ttot = Ttot; a.act = act; a.alpha = alpha;
    a.omin_h = omin_h; a.omin_w = omin_w;
    const int BN = Nd <= 16 ? 16 : (Nd <= 32 ? 32 : (Nd <= 64 ? 64 : 128));
    a.pitch = pick_pitch(Cs, ssw);
    a.PW = (maxWc - 1) * ssw + (omax_w - omin_w) + 1;
    const long escratch = 4L * (16 * (BN * 2 + 16) + 64);
    const long fixed = 2L * BN * WPITCH + (long)total_taps * 8 + 64 + 64 + escratch;
    if (lds_budget <= 0) lds_budget = 80 * 1024;             // two blocks per CU
    const long hard = 160L * 1024;
    auto patch_rows = [&](int br) { return (br - 1) * ssh + (omax_h - omin_h) + 1; };
    long img_full = (long)patch_rows(maxHc) * a.PW * a.pitch;
    long avail = lds_budget - fixed;
    const int ntile = (Nd + BN - 1) / BN;
    if (img_full <= avail) {
        a.BR = maxHc; a.bands = 1;
        long rows_per_img = (long)maxHc * maxWc;
        long img = avail / img_full;
        long cap = (1024 + rows_per_img - 1) / rows_per_img;         // ~4 passes of 256 rows per class
        if (img > cap) img = cap;
        if (img > B) img = B;
        while (img > 1 && (long)((B + img - 1) / img) * ntile < 256) --img;      // keep every CU busy
        // among [img/2, img] take the count that wastes the fewest rows of the 256-row passes
        long best = img;
        double best_eff = 0;
        for (long c = img; c >= (img + 1) / 2 && c >= 1; --c) {
            long rows = c * rows_per_img;
            double eff = (double)rows / (double)((rows + 255) / 256 * 256);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
        }
        a.IMG = (int)best;
    } else {
        a.IMG = 1;
        int br = maxHc;
        while (br > 1 && (long)patch_rows(br) * a.PW * a.pitch > avail) --br;
        if ((long)patch_rows(br) * a.PW * a.pitch > avail) {          // not even one row at 2 blocks/CU
            avail = hard - fixed;
            if ((long)patch_rows(br) * a.PW * a.pitch > avail) return 1;
        }
        a.BR = br; a.bands = (maxHc + br - 1) / br;
    }
    a.PH = patch_rows(a.BR);
    a.img_bytes = a.PH * a.PW * a.pitch;
    long patch = (long)a.IMG * a.img_bytes;
    a.w_off = (int)((patch + 15) / 16 * 16);
    a.t_off = a.w_off + 2 * BN * WPITCH;
    a.z_off = (int)((a.t_off + (long)total_taps * 8 + 15) / 16 * 16);
    a.e_off = a.z_off + 64;
    size_t smem = (size_t)a.e_off + escratch;
    if (smem > (size_t)hard) return 1;
    a.inv_per_img = 1.0f / (float)(a.PH * a.PW * (Cs / 8));
    a.inv_cs8 = 1.0f / (float)(Cs / 8);
    a.inv_PW = 1.0f / (float)a.PW;
    { const char* e = getenv("DDL_PCONV_DBG"); a.dbg = e ? atoi(e) : 0; }
    if (getenv("DDL_PCONV_VERBOSE")) printf("pconv: BN=%d IMG=%d BR=%d bands=%d PH=%d PW=%d pitch=%d lds=%d\n", BN, a.IMG, a.BR, a.bands, a.PH, a.PW, a.pitch, (int)(a.e_off + escratch));
    if ((long)a.IMG * a.PH * a.PW * (Cs / 8) >= (1L << 22)) return 1;
    // The patch scheme pays when a unit offers enough rows to amortise streaming the weights; tiny
    // images with wide channels (4x4x128, Dense) are better served by the global-gather kernel.
    if ((long)a.IMG * maxHc * maxWc < min_rows) return 1;
    int units = ((B + a.IMG - 1) / a.IMG) * a.bands;
    switch (BN) {
        case 16: return launch_pconv<16>(a, units, smem, s);
        case 32: return launch_pconv<32>(a, units, smem, s);
        case 64: return launch_pconv<64>(a, units, smem, s);
        default: return launch_pconv<128>(a, units, smem, s);
    }
}
