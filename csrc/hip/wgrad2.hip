// Patch-resident, persistent weight-gradient kernel (gfx950, MFMA 16x16x32 bf16 + ds_read_b64_tr_b16).
//
//   dW[n][t][c] = sum_{b,oh,ow} dY[b,oh,ow,n] * X[b, oh*sh + kh*dh - pt, ow*sw + kw*dw - pl, c]
//
// wgrad.hip gives every tap its own blocks, each re-reading its dY and X pixels from global memory:
// taps x M x (N + C) x 2 bytes of L2 traffic (0.9 GB for one DCGAN layer at B = 1000), which is what
// bounds it.  Here a block is a persistent worker over a range of *units* (IMG images or a row band
// of one image).  Per unit it stages the dY tile and the X patch (zero halo) into LDS once; then
// EVERY tap is computed out of LDS: wave w owns taps {gw, gw + G, ...} (gw = global wave index among
// the G = NW x tap-groups waves that share this unit; NW = 4 or 8 waves per block), keeps their [TN x TC] accumulators in registers
// for its whole life, and walks all 32-pixel slices of the unit -- the dY fragments of a slice are
// read once and reused for all of the wave's taps.  Both operands are pixel-major (k-strided for an
// MFMA fragment); ds_read_b64_tr_b16 transposes them on the LDS read, each lane pointing at its own
// pixel, so the patch needs no re-layout (per-pixel patch offsets come from a small LDS table).
// At the end each wave writes its taps once to a per-worker slab; slab_reduce (wgrad.hip) sums the
// workers in a fixed order (deterministic).
#include "common.h"

struct Wg2Args {
    const bf16_t* dy; const bf16_t* x; float* slab;
    int B, OH, OW, Ny;        // dY [B][OH][OW][Ny]
    int H, W, Cx;             // X  [B][H][W][Cx]
    int KH, KW, sh, sw, dh, dw, pt, pl;
    int n_lim, c_lim;
    long sn, st, sc;          // output strides (elements)
    long slab_stride;
    int frag;                 // slab in fragment order (see the epilogue); else in the gradient's own layout
    int IMG, BR, bands, n_units;
    int PH, PW, xpitch, ximg_bytes;   // X patch: rows/cols (with halo), bytes per pixel / per image
    int ypitch;                        // dY tile bytes per pixel
    int y_off, t_off, k_off;           // LDS offsets: dY tile, pixel table, tap-offset table (X patch at 0)
    int Mmax;                          // pixels per full unit (IMG * BR * OW)
    int tiles_c, ngroups;
    int scx8, sny8;                    // 16-byte channel units staged per pixel: min(Cx, TC) / 8, min(Ny, TN) / 8
    float inv_xper_img, inv_cx8, inv_PW, inv_yrow, inv_OW, inv_rows;
    // LDS-DMA staging (DMA instantiations): two (X patch, dY tile) buffers of buf_bytes each, filled 16-byte chunk by
    // chunk INCLUDING the pitch padding (chunk u of the padded image -> LDS byte 16 u: one wave-instruction is one
    // contiguous KiB); rx / ry = chunks per thread of the X / dY section, xcpp / ycpp = chunks per pixel
    int buf_bytes, rx, ry, xcpp, ycpp;
    float inv_xcpp, inv_ycpp, inv_phpw;
    // register-staged forms only: a per-channel affine on the dY operand as it is staged -- dY' = bf16(dY * y_scale[n] +
    // y_shift[n]) on real pixels, channels >= y_ctrue zero -- i.e. a BatchNorm in front of a transposed convolution
    // (whose INPUT is this kernel's dY operand) that was applied on load in the forward pass and never materialised
    const float* y_scale; const float* y_shift; int y_ctrue, a_off;
};

__device__ __forceinline__ int fdiv2(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }

__device__ __forceinline__ s16x4 tr_read2(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

// NW = waves per block.  NW = 8 (one 512-thread block per CU, two waves per SIMD) shares one staged unit among
// twice the waves: twice the loads in flight while staging and half the taps per wave in the MFMA phase.
// DMA: the unit's X patch and dY tile are staged by `buffer_load_dwordx4 ... lds` into one of TWO LDS buffers while
// the MFMA phase of the previous unit runs out of the other.  Measured on the register-staged form (ablation, B = 65536
// 16x16x64 -> 8x8x128): staging alone 1.43 ms, MFMA phase alone 1.40 ms, together 2.49 ms -- the two phases barely
// overlapped, and the staging pushed every byte through VGPRs and a ds_write_b128.  The padded pixel pitches the
// transposing LDS reads want are kept: the DMA walks the PADDED image chunk by chunk and the padding chunks (like halo
// pixels, missing images and rows) fetch an out-of-range offset, which writes zeros and moves no data.
// The DMA is issued through inline asm: behind the builtin the compiler waits vmcnt(0) in front of the transposing LDS
// reads of the MFMA phase (it cannot tell them from readers of the buffer in flight) and the overlap is gone.  Hidden
// in asm the loads are ours to wait for -- one vmcnt(0) in front of the barrier that publishes a buffer.  The
// compiler's own vmcnt waits stay safe: loads complete in order, ours only make it wait longer, never shorter.
typedef unsigned int w2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w2_u32x4 w2_rsrc_words(const void* p, unsigned bytes) {
    const unsigned long a = (unsigned long)p;
    return (w2_u32x4){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void w2_dma16(w2_u32x4 rs, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_base) : "memory");
}

template <int TN, int TC, int TPW, int NW, bool DMA = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 || TPW * (TN / 16) * (TC / 16) * 4 > 160 ? 1 : 2)) void wgrad2_kernel(const Wg2Args a) {
    constexpr int SN = TN / 16, SC = TC / 16, NT = 64 * NW;
    // the one-block-per-CU variants (512-register budget) whose fragment sets leave room: they stage with deeper
    // rounds and keep the next unit's first round in flight during the MFMA phase
    constexpr bool PIPE = !DMA && NW == 4 && TPW * SN * SC * 4 > 160 && (SN + TPW * SC) * 8 <= 160;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;
    char* sY = smem + a.y_off;
    int* sTbl = (int*)(smem + a.t_off);          // per unit pixel: X patch byte offset
    int* sTap = (int*)(smem + a.k_off);          // per tap: patch byte offset

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int T = a.KH * a.KW;
    const int group = blockIdx.y;
    const int nt = blockIdx.z / a.tiles_c, ct = blockIdx.z - nt * a.tiles_c;
    const int n0 = nt * TN, c0 = ct * TC;
    const int G = NW * a.ngroups;                 // waves sharing a unit's taps
    const int gw = group * NW + wave;
    int ntaps_w = 0;                               // taps gw, gw + G, ... < T
    for (int t = gw; t < T; t += G) ++ntaps_w;

    // ---- once per block: pixel -> patch offset table, tap offsets (omin = -pt, -pl)
    for (int m = tid; m < a.Mmax; m += NT) {
        int img = fdiv2(m, a.inv_rows), rem = m - img * (a.BR * a.OW);
        int il = fdiv2(rem, a.inv_OW), jl = rem - il * a.OW;
        sTbl[m] = img * a.ximg_bytes + (il * a.sh * a.PW + jl * a.sw) * a.xpitch;
    }
    for (int t = tid; t < T; t += NT) {
        int kh = t / a.KW, kw = t - kh * a.KW;
        sTap[t] = (kh * a.dh * a.PW + kw * a.dw) * a.xpitch;
    }
    const float* sAff = (const float*)(smem + a.a_off);     // [2][TN]: scale, shift of this tile's dY channels
    if (!DMA && a.y_scale) {
        float* w = (float*)(smem + a.a_off);
        for (int c = tid; c < TN; c += NT) {
            const bool real = n0 + c < a.y_ctrue;
            w[c] = real ? a.y_scale[n0 + c] : 0.f;
            w[TN + c] = real ? a.y_shift[n0 + c] : 0.f;
        }
    }

    f32x4 acc[TPW][SN][SC];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < SN; ++i)
#pragma unroll
            for (int j = 0; j < SC; ++j) acc[q][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // only the tile's channels are staged (c0 .. c0+TC of X, n0 .. n0+TN of dY): with 2 x 2 output tiles the full
    // rows would be staged by four blocks that each use half of them
    const int cx8 = a.scx8, ny8 = a.sny8;
    // this wave's tap offsets into the patch (taps it does not own alias tap 0: read, never accumulated)
    int toff[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        int t = gw + q * G;
        int tt = t < T ? t : 0;
        int kh = tt / a.KW, kw = tt - kh * a.KW;
        toff[q] = (kh * a.dh * a.PW + kw * a.dw) * a.xpitch + (4 * tp) * 2;
    }
    const int chan_y = (4 * tp) * 2;

    // Branch-free bounds-checked buffer loads for the staging gathers: an out-of-range unit gets an offset beyond
    // the descriptor and reads as zero.  (A predicated global_load compiles to an exec-masked branch after which
    // the compiler can no longer count outstanding loads: the eight loads per thread were serialised.)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.B * a.H * a.W * a.Cx * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.B * a.OH * a.OW * a.Ny * 2u), 0x00020000);
    constexpr unsigned OOB_OFF = 0xFFFFFFF0u;

    // ---- staging: the X patch (zero halo / missing images) and the dY tile (pixel m = (img, il, jl) of the band;
    // rows past the band / batch are zero) of a unit.  The kernel is bound by this phase, so it runs as one stream
    // of rounds of NLD 16-byte loads per thread -- X rounds, then dY rounds -- with the loads of round r+1 issued
    // before the LDS stores of round r (no drain between X and dY), and in the one-block-per-CU variants the first
    // round of the NEXT unit is issued before the MFMA phase of the current one and waits in registers.
    const int per_img = a.PH * a.PW * cx8;
    const int nx = a.IMG * per_img;
    const int nyu = (a.Mmax + 31) / 32 * 32 * ny8;
    constexpr int NLD = PIPE ? 6 : 4, RND = NT * NLD;       // loads per thread per round, units per round
    const int rx = (nx + RND - 1) / RND, R = rx + (nyu + RND - 1) / RND;
    auto issue = [&](int unit, int r, u32x4_t (&v)[NLD], int (&dst)[NLD]) {
        const int grp = unit / a.bands, band = unit - grp * a.bands;
        const int b0 = grp * a.IMG, i0 = band * a.BR;
        if (r < rx) {                                  // block-uniform
            const int row0 = i0 * a.sh - a.pt, col0 = -a.pl;
            const int base = tid + r * RND;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                int u = base + k * NT;
                bool in = u < nx;
                int uu = in ? u : 0;
                int img = fdiv2(uu, a.inv_xper_img), rr = uu - img * per_img;
                int pix = fdiv2(rr, a.inv_cx8), c8 = rr - pix * cx8;
                int pr = fdiv2(pix, a.inv_PW), pc = pix - pr * a.PW;
                int sr = row0 + pr, sc = col0 + pc, bb = b0 + img;
                const bool ok = in & (bb < a.B) & ((unsigned)sr < (unsigned)a.H) & ((unsigned)sc < (unsigned)a.W) &
                                (c0 + c8 * 8 < a.Cx);
                const unsigned off = ok ? (unsigned)(((bb * a.H + sr) * a.W + sc) * a.Cx + c0 + c8 * 8) * 2u : OOB_OFF;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
                dst[k] = in ? img * a.ximg_bytes + pix * a.xpitch + c8 * 16 : -1;
            }
        } else {
            const int nrow = min(a.BR, a.OH - i0);
            const int base = tid + (r - rx) * RND;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                int u = base + k * NT;
                bool in = u < nyu;
                int uu = in ? u : 0;
                int m = fdiv2(uu, a.inv_yrow), c8 = uu - m * ny8;
                int img = fdiv2(m, a.inv_rows), rem = m - img * (a.BR * a.OW);
                int il = fdiv2(rem, a.inv_OW), jl = rem - il * a.OW;
                int bb = b0 + img;
                const bool ok = in & (m < a.Mmax) & (bb < a.B) & (il < nrow) & (n0 + c8 * 8 < a.Ny);
                const unsigned off = ok ? (unsigned)(((bb * a.OH + i0 + il) * a.OW + jl) * a.Ny + n0 + c8 * 8) * 2u : OOB_OFF;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
                // bit 30 + the channel unit in bits 20..23: `store` applies the affine (real pixels only: rows past
                // the band / batch must stay zero)
                dst[k] = in ? ((a.y_off + m * a.ypitch + c8 * 16) | ((ok && a.y_scale) ? ((1 << 30) | (c8 << 20)) : 0)) : -1;
            }
        }
    };
    auto store = [&](const u32x4_t (&v)[NLD], const int (&dst)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (dst[k] >= 0) {
                u32x4_t o = v[k];
                if (dst[k] & (1 << 30)) {
                    const int cb = ((dst[k] >> 20) & 15) * 8;
                    float f[8];
                    unpack8(make_uint4(o[0], o[1], o[2], o[3]), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sAff[cb + e], sAff[TN + cb + e]);
                    const uint4 r = pack8(f);
                    o = (u32x4_t){r.x, r.y, r.z, r.w};
                }
                *(u32x4_t*)(smem + (dst[k] & 0x3FFFF)) = o;
            }
    };
    // ---- every wave walks all 32-pixel slices; its taps' accumulators stay in registers.  All of a
    // slice's fragments are fetched first (one burst of transposing LDS reads), then the MFMAs run.
    auto load = [&](int m0, bf16x8_t (&fy)[SN], bf16x8_t (&fx)[TPW][SC], const char* sX, const char* sY) {
        const int plo = m0 + 4 * g + tq, phi = plo + 16;
        const int xlo = sTbl[plo < a.Mmax ? plo : 0], xhi = sTbl[phi < a.Mmax ? phi : 0];
#pragma unroll
        for (int i = 0; i < SN; ++i) {
            s16x4 lo = tr_read2(sY + plo * a.ypitch + chan_y + i * 32);
            s16x4 hi = tr_read2(sY + phi * a.ypitch + chan_y + i * 32);
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fy[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int j = 0; j < SC; ++j) {
                s16x4 lo = tr_read2(sX + xlo + toff[q] + j * 32);
                s16x4 hi = tr_read2(sX + xhi + toff[q] + j * 32);
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fx[q][j] = __builtin_bit_cast(bf16x8_t, v);
            }
    };
    // Unconditional MFMAs: a tap slot this wave does not own aliases tap 0 and accumulates values that are
    // never written.  (Guarding them by `q < ntaps_w` makes the compiler shuffle the accumulator file
    // through v_accvgpr moves at every slice.)
    auto mma = [&](const bf16x8_t (&fy)[SN], const bf16x8_t (&fx)[TPW][SC]) {
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int j = 0; j < SC; ++j)
                    acc[q][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[q][j], acc[q][i][j], 0, 0, 0);
    };

    if constexpr (DMA) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int phpw = a.PH * a.PW;
        const w2_u32x4 dX = w2_rsrc_words(a.x, (unsigned)a.B * a.H * a.W * a.Cx * 2u);
        const w2_u32x4 dY = w2_rsrc_words(a.dy, (unsigned)a.B * a.OH * a.OW * a.Ny * 2u);
        const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
        auto dma_issue = [&](int unit, int buf) {
            const int grp = unit / a.bands, band = unit - grp * a.bands;
            const int b0 = grp * a.IMG, i0 = band * a.BR;
            const int row0 = i0 * a.sh - a.pt, col0 = -a.pl;
            const unsigned xb = lds0 + buf * a.buf_bytes;
            for (int r = 0; r < a.rx; ++r) {
                const int u = r * NT + tid;                              // chunk of the padded patch image
                const int pixl = fdiv2(u, a.inv_xcpp), c16 = u - pixl * a.xcpp;
                const int img = fdiv2(pixl, a.inv_phpw), pix = pixl - img * phpw;
                const int pr = fdiv2(pix, a.inv_PW), pc = pix - pr * a.PW;
                const int sr = row0 + pr, sc = col0 + pc, bb = b0 + img;
                const bool ok = (c16 < cx8) & (img < a.IMG) & (bb < a.B) & ((unsigned)sr < (unsigned)a.H) &
                                ((unsigned)sc < (unsigned)a.W) & (c0 + c16 * 8 < a.Cx);
                const unsigned off = ok ? (unsigned)(((bb * a.H + sr) * a.W + sc) * a.Cx + c0 + c16 * 8) * 2u : OOB_OFF;
                w2_dma16(dX, off, xb + (r * NT + wv * 64) * 16);
            }
            const unsigned yb = xb + a.y_off;
            const int nrow = min(a.BR, a.OH - i0);
            for (int r = 0; r < a.ry; ++r) {
                const int u = r * NT + tid;
                const int m = fdiv2(u, a.inv_ycpp), c16 = u - m * a.ycpp;
                const int img = fdiv2(m, a.inv_rows), rem = m - img * (a.BR * a.OW);
                const int il = fdiv2(rem, a.inv_OW), jl = rem - il * a.OW;
                const int bb = b0 + img;
                const bool ok = (c16 < ny8) & (m < a.Mmax) & (bb < a.B) & (il < nrow) & (n0 + c16 * 8 < a.Ny);
                const unsigned off = ok ? (unsigned)(((bb * a.OH + i0 + il) * a.OW + jl) * a.Ny + n0 + c16 * 8) * 2u : OOB_OFF;
                w2_dma16(dY, off, yb + (r * NT + wv * 64) * 16);
            }
        };
        int cur = 0;
        if ((int)blockIdx.x < a.n_units) dma_issue(blockIdx.x, 0);
        for (int unit = blockIdx.x; unit < a.n_units; unit += gridDim.x) {
            // this unit's DMA has landed (every wave waits for its own, the barrier publishes them) and every wave is
            // done with the MFMA phase that read the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int next = unit + gridDim.x;
            if (next < a.n_units) dma_issue(next, cur ^ 1);             // lands during this unit's MFMA phase
            const char* xb = smem + cur * a.buf_bytes;
            for (int m0 = 0; m0 < a.Mmax; m0 += 32) {
                bf16x8_t fy[SN], fx[TPW][SC];
                load(m0, fy, fx, xb, xb + a.y_off);
                mma(fy, fx);
            }
            cur ^= 1;
        }
    }
    u32x4_t vP[PIPE ? NLD : 1];                    // PIPE: round 0 of the unit about to be staged
    int dP[PIPE ? NLD : 1];
    if constexpr (PIPE) {
        if (blockIdx.x < a.n_units) issue(blockIdx.x, 0, vP, dP);
    }
    for (int unit = blockIdx.x; !DMA && unit < a.n_units; unit += gridDim.x) {
        __syncthreads();                          // previous unit fully consumed (also orders the tables)
        {
            u32x4_t vA[NLD], vB[NLD];
            int dA[NLD], dB[NLD];
            if constexpr (PIPE) {
                if (R > 1) issue(unit, 1, vB, dB);
                store(vP, dP);
                for (int r = 1; r < R; r += 2) {      // vB holds round r
                    if (r + 1 < R) issue(unit, r + 1, vA, dA);
                    store(vB, dB);
                    if (r + 1 < R) {
                        if (r + 2 < R) issue(unit, r + 2, vB, dB);
                        store(vA, dA);
                    }
                }
            } else {
                issue(unit, 0, vA, dA);
                for (int r = 0; r < R; r += 2) {
                    if (r + 1 < R) issue(unit, r + 1, vB, dB);
                    store(vA, dA);
                    if (r + 1 < R) {
                        if (r + 2 < R) issue(unit, r + 2, vA, dA);
                        store(vB, dB);
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (PIPE) {
            const int next = unit + gridDim.x;
            if (next < a.n_units) issue(next, 0, vP, dP);          // in flight during the MFMA phase
        }
        for (int m0 = 0; m0 < a.Mmax; m0 += 32) {
            bf16x8_t fy[SN], fx[TPW][SC];
            load(m0, fy, fx, sX, sY);
            mma(fy, fx);
        }
    }

    // ---- one slab write per wave tap: lane holds rows 16i + 4g + e, column 16j + li of the tile.
    // Fragment order ([split][tile][group][wave][q][i][j][lane] float4): every store instruction of a wave is one
    // contiguous KiB.  Written in the gradient's own layout the same data leaves as 16- or 64-byte pieces, and with
    // 256 partial copies of the gradient (50 MB per call at the reference batch) that store burst was a third of
    // the kernel; slab_reduce_frag (below) sums the copies and scatters only the final, small gradient.
    if (a.frag) {
        float4* o4 = (float4*)a.slab +
                     ((((long)blockIdx.x * gridDim.z + blockIdx.z) * a.ngroups + group) * NW + wave) * (TPW * SN * SC * 64) + lane;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (q < ntaps_w) {
#pragma unroll
                for (int i = 0; i < SN; ++i)
#pragma unroll
                    for (int j = 0; j < SC; ++j)
                        o4[((q * SN + i) * SC + j) * 64] =
                            make_float4(acc[q][i][j][0], acc[q][i][j][1], acc[q][i][j][2], acc[q][i][j][3]);
            }
        }
        return;
    }
    float* out = a.slab + (long)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        if (q < ntaps_w) {
            const int tap = gw + q * G;
#pragma unroll
            for (int i = 0; i < SN; ++i)
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const int nb = n0 + 16 * i + 4 * g, c = c0 + 16 * j + li;
                    if (a.sn == 1 && nb + 3 < a.n_lim) {          // n-contiguous layout: one 16-byte store
                        if (c < a.c_lim)
                            *(float4*)(out + nb + tap * a.st + c * a.sc) =
                                make_float4(acc[q][i][j][0], acc[q][i][j][1], acc[q][i][j][2], acc[q][i][j][3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb + e < a.n_lim && c < a.c_lim)
                                out[(nb + e) * a.sn + tap * a.st + c * a.sc] = acc[q][i][j][e];
                    }
                }
        }
    }
}

// Sum the fragment-order slabs over the splits (fixed order: deterministic) and scatter into the gradient.
// One thread column per float4 of a slab, eight split lanes per column combined through LDS.
struct FragDecode {
    int TN, TC, TPW, SN, SC, NW, ngroups, tiles_c, T, n_lim, c_lim;
    long sn, st, sc;
};

__global__ __launch_bounds__(256) void slab_reduce_frag_kernel(const float4* __restrict__ slab, float* __restrict__ out,
                                                               long e4, int nsplit, const FragDecode d) {
    __shared__ float4 red[8][33];
    const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long f = (long)blockIdx.x * 32 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // decode first: slots of taps no wave owns are never written (uninitialised memory) and must not be read
    int n = 0, c = 0, tap = d.T;
    if (f < e4) {
        const int lane = (int)(f & 63);
        long r = f >> 6;
        const int j = (int)(r % d.SC); r /= d.SC;
        const int i = (int)(r % d.SN); r /= d.SN;
        const int q = (int)(r % d.TPW); r /= d.TPW;
        const int wave = (int)(r % d.NW); r /= d.NW;
        const int group = (int)(r % d.ngroups);
        const int z = (int)(r / d.ngroups);
        const int nt = z / d.tiles_c, ct = z - nt * d.tiles_c;
        tap = group * d.NW + wave + q * d.NW * d.ngroups;
        n = nt * d.TN + 16 * i + 4 * (lane >> 4);
        c = ct * d.TC + 16 * j + (lane & 15);
    }
    const bool live = tap < d.T;                 // uniform per 64 consecutive f (a wave of the producer)
    if (live) {
        int k = kl;
        // (a lane's ~30 slabs used to be seven dependent trips of four loads: the kernel is a chain of memory latencies,
        //  not a bandwidth problem -- sixteen loads per trip, summed in the same order as before)
        for (; k + 120 < nsplit; k += 128) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slab[(long)(k + 8 * u) * e4 + f];
#pragma unroll
            for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k + 24 < nsplit; k += 32) {       // four independent 16-byte loads per trip, fixed order
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = slab[(long)(k + 8 * u) * e4 + f];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; k += 8) {
            float4 v = slab[(long)k * e4 + f];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && live && c < d.c_lim) {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            float4 v = red[q][col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < d.n_lim) out[(n + e) * d.sn + tap * d.st + c * d.sc] = sv[e];
    }
}

// floats per split of a fragment-order slab
DDL_API long ddl_wgrad2_frag_elems(int TN, int TC, int TPW, int NW, int ngroups, int n_lim, int c_lim) {
    long tiles = (long)((n_lim + TN - 1) / TN) * ((c_lim + TC - 1) / TC);
    return tiles * ngroups * NW * TPW * (TN / 16) * (TC / 16) * 256;
}

DDL_API int ddl_slab_reduce_frag(const float* slab, float* out, int nsplit, int TN, int TC, int TPW, int NW, int ngroups,
                                 int T, int n_lim, int c_lim, long sn, long st, long sc, hipStream_t s) {
    if (nsplit < 1 || TN % 16 || TC % 16 || TPW < 1 || ngroups < 1 || (NW != 4 && NW != 8)) return -1;
    FragDecode d{TN, TC, TPW, TN / 16, TC / 16, NW, ngroups, (c_lim + TC - 1) / TC, T, n_lim, c_lim, sn, st, sc};
    const long e4 = ddl_wgrad2_frag_elems(TN, TC, TPW, NW, ngroups, n_lim, c_lim) / 4;
    slab_reduce_frag_kernel<<<(int)((e4 + 31) / 32), 256, 0, s>>>((const float4*)slab, out, e4, nsplit, d);
    return 0;
}

template <int TN, int TC, int TPW, int NW, bool DMA = false>
static int launch_wg2(const Wg2Args& a, int nsplit, int tiles, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipFuncSetAttribute((const void*)wgrad2_kernel<TN, TC, TPW, NW, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
            done = true;
        }
    }
    wgrad2_kernel<TN, TC, TPW, NW, DMA><<<dim3(nsplit, a.ngroups, tiles), 64 * NW, smem, s>>>(a);
    return 0;
}

// LDS bytes of the DMA form: two buffers whose X and dY sections are whole rounds of one 16-byte chunk per thread
static long wg2_dma_section(long chunks, int NT) { return (chunks + NT - 1) / NT * NT * 16; }

DDL_API int ddl_wgrad2_lds_dma(int IMG, int BR, int OW, int PH, int PW, int xpitch, int ypitch, int taps, int NW) {
    const int NT = 64 * NW;
    long mpad = ((long)IMG * BR * OW + 31) / 32 * 32;
    long buf = wg2_dma_section((long)IMG * PH * PW * (xpitch / 16), NT) + wg2_dma_section(mpad * (ypitch / 16), NT);
    long tot = 2 * buf + ((long)IMG * BR * OW * 4 + 15) / 16 * 16 + ((long)taps * 4 + 15) / 16 * 16;
    return tot > (1L << 30) ? -1 : (int)tot;
}

DDL_API int ddl_wgrad2_lds(int IMG, int BR, int OW, int Ny, int PH, int PW, int xpitch, int ypitch, int taps) {
    long x = (long)IMG * PH * PW * xpitch;
    long mpad = ((long)IMG * BR * OW + 31) / 32 * 32;
    long tot = (x + 15) / 16 * 16 + mpad * ypitch + (long)IMG * BR * OW * 4 + 16 + (long)taps * 4 + 16;
    return tot > (1L << 30) ? -1 : (int)tot;
}

// The caller (ops/hip.py: _wgrad2_plan) chooses the unit decomposition, tile and split; slab holds
// nsplit * numel floats.  The slab reduce into `out` is launched by the caller's existing path.
DDL_API int ddl_conv_wgrad2(const void* dy, const void* x, float* slab, long numel, int nsplit, int B, int OH,
                            int OW, int Ny, int H, int W, int Cx, int KH, int KW, int sh, int sw, int dh, int dw,
                            int pt, int pl, int n_lim, int c_lim, long sn, long st, long sc, int IMG, int BR,
                            int xpitch, int ypitch, int TN, int TC, int TPW, int NW, int ngroups, int frag,
                            const float* y_scale, const float* y_shift, int y_ctrue, hipStream_t s) {
    if (Ny % 8 || Cx % 8 || nsplit < 1 || IMG < 1 || BR < 1 || ngroups < 1) return -1;
    if ((y_scale == nullptr) != (y_shift == nullptr) || (y_scale && (y_ctrue < 0 || y_ctrue > Ny || TN > 128))) return -1;
    if (y_scale && ((frag >> 1) & 1)) return -6;          // the LDS-DMA forms move dY without touching it
    if ((long)B * OH * OW * Ny >= (1L << 31) || (long)B * H * W * Cx >= (1L << 31)) return -3;
    const bool dma = (frag >> 1) & 1;          // bit 1 of `frag`: LDS-DMA staging, double-buffered
    frag &= 1;
    Wg2Args a;
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.slab = slab;
    a.B = B; a.OH = OH; a.OW = OW; a.Ny = Ny; a.H = H; a.W = W; a.Cx = Cx;
    a.KH = KH; a.KW = KW; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl;
    a.n_lim = n_lim; a.c_lim = c_lim; a.sn = sn; a.st = st; a.sc = sc; a.slab_stride = numel; a.frag = frag;
    a.IMG = IMG; a.BR = BR; a.bands = (OH + BR - 1) / BR;
    a.n_units = ((B + IMG - 1) / IMG) * a.bands;
    a.PH = (BR - 1) * sh + (KH - 1) * dh + 1;
    a.PW = (OW - 1) * sw + (KW - 1) * dw + 1;
    a.xpitch = xpitch; a.ypitch = ypitch; a.ximg_bytes = a.PH * a.PW * xpitch;
    a.Mmax = IMG * BR * OW;
    a.scx8 = (Cx < TC ? Cx : TC) / 8; a.sny8 = (Ny < TN ? Ny : TN) / 8;
    if (xpitch % 16 || ypitch % 16 || xpitch < a.scx8 * 16 || ypitch < a.sny8 * 16) return -1;
    if ((long)IMG * a.PH * a.PW * (Cx / 8) >= (1L << 22) || (long)(a.Mmax + 32) * (Ny / 8) >= (1L << 22)) return -2;
    long xbytes = ((long)IMG * a.ximg_bytes + 15) / 16 * 16;
    long mpad = (a.Mmax + 31) / 32 * 32;
    a.y_off = (int)xbytes;
    a.t_off = (int)(xbytes + mpad * ypitch);
    a.k_off = a.t_off + ((a.Mmax * 4 + 15) / 16 * 16);
    a.buf_bytes = 0; a.rx = a.ry = 0; a.xcpp = xpitch / 16; a.ycpp = ypitch / 16;
    a.inv_xcpp = 1.0f / (float)a.xcpp; a.inv_ycpp = 1.0f / (float)a.ycpp; a.inv_phpw = 1.0f / (float)(a.PH * a.PW);
    if (dma) {
        const int NT = 64 * NW;
        const long nxc = (long)IMG * a.PH * a.PW * a.xcpp, nyc = mpad * a.ycpp;
        const long xsec = wg2_dma_section(nxc, NT), ysec = wg2_dma_section(nyc, NT);
        if (nxc >= (1L << 22) || nyc >= (1L << 22)) return -2;
        a.rx = (int)(xsec / (NT * 16)); a.ry = (int)(ysec / (NT * 16));
        a.y_off = (int)xsec;
        a.buf_bytes = (int)(xsec + ysec);
        a.t_off = 2 * a.buf_bytes;
        a.k_off = a.t_off + ((a.Mmax * 4 + 15) / 16 * 16);
    }
    size_t smem = (size_t)a.k_off + (size_t)(KH * KW * 4 + 15) / 16 * 16;
    a.y_scale = y_scale; a.y_shift = y_shift; a.y_ctrue = y_ctrue; a.a_off = (int)smem;
    if (y_scale) smem += (size_t)2 * TN * 4;
    if (smem > 160 * 1024) return y_scale ? -6 : -2;      // -6: the caller materialises the operand instead
    if (smem >= (1 << 18)) return -2;                      // (staging packs LDS offsets into 18 bits)
    int tiles_n = (n_lim + TN - 1) / TN;
    a.tiles_c = (c_lim + TC - 1) / TC;
    a.ngroups = ngroups;
    if (NW * ngroups * TPW < KH * KW) return -4;           // not every tap would be owned by a wave
    a.inv_xper_img = 1.0f / (float)(a.PH * a.PW * a.scx8);
    a.inv_cx8 = 1.0f / (float)a.scx8;
    a.inv_PW = 1.0f / (float)a.PW;
    a.inv_yrow = 1.0f / (float)a.sny8;
    a.inv_OW = 1.0f / (float)OW;
    a.inv_rows = 1.0f / (float)(BR * OW);
    int tiles = tiles_n * a.tiles_c;
    if (dma) {
#define W2_DMA(TN_, TC_, TPW_, NW_) if (NW == NW_ && TN == TN_ && TC == TC_ && TPW == TPW_) return launch_wg2<TN_, TC_, TPW_, NW_, true>(a, nsplit, tiles, smem, s);
        W2_DMA(64, 64, 2, 8) W2_DMA(64, 32, 4, 8) W2_DMA(32, 64, 4, 8) W2_DMA(32, 32, 4, 8)
        W2_DMA(64, 32, 7, 4) W2_DMA(64, 32, 4, 4) W2_DMA(32, 32, 7, 4) W2_DMA(32, 32, 4, 4)
#undef W2_DMA
        return -4;
    }
#define W2_CASE8(TN_, TC_, TPW_) if (NW == 8 && TN == TN_ && TC == TC_ && TPW == TPW_) return launch_wg2<TN_, TC_, TPW_, 8>(a, nsplit, tiles, smem, s);
    W2_CASE8(64, 64, 2) W2_CASE8(64, 32, 4) W2_CASE8(32, 64, 4) W2_CASE8(32, 32, 4)
#undef W2_CASE8
    if (NW != 4) return -4;
#define W2_CASE(TN_, TC_, TPW_) if (TN == TN_ && TC == TC_ && TPW == TPW_) return launch_wg2<TN_, TC_, TPW_, 4>(a, nsplit, tiles, smem, s);
    W2_CASE(64, 64, 3) W2_CASE(64, 64, 2) W2_CASE(64, 32, 7) W2_CASE(64, 32, 4) W2_CASE(32, 64, 7) W2_CASE(32, 64, 4)
    W2_CASE(32, 32, 7) W2_CASE(32, 32, 4) W2_CASE(64, 16, 7) W2_CASE(16, 64, 7) W2_CASE(32, 16, 7) W2_CASE(16, 32, 7)
    W2_CASE(16, 16, 7)
#undef W2_CASE
    return -4;
}
