// Fused depthwise -> pointwise blocks of the FRONT of the backbone at large batch (conv_dw, modules/conv.py:13-22; the blocks
// models/with_mobilenet.py:94-97 builds: 32 -> 64, 64 -> 128 stride 2, 128 -> 128, 128 -> 256 stride 2), f32 and bf16 storage.
//
// These blocks are bandwidth work (0.3 MFLOP per KB moved at bf16): at batch 32 the row-block kernels (dwpw_kernel /
// dwpw_bf16_kernel) fetch every input vector six times through L1 / L2 with per-thread loads and run a chain of five barriers
// and three global round trips per 64-pixel workgroup — 2.7 - 3.1 TB/s.  Here a workgroup owns a 2-D patch of output pixels
// (PH x 8) and ALL channels, like dw_tiled_kernel (net_kernels.hip):
//   A  the (PH s + 2) x (8 s + 2) x C input window is brought into LDS with coalesced 16-byte buffer loads, every request of a
//      thread issued before its first LDS store (pieces outside the image read offset 2^31: zeros);
//   B  the depthwise 3x3 (+ folded BN bias, ReLU) runs from LDS with a rolling 3 x 3 register window (stride 1: three LDS reads
//      per output vector instead of nine), f32 arithmetic in the reference's order (bias, then the taps row-major); the results
//      wait in registers until every thread has finished reading the window, then overwrite it as the [pixels][C] MFMA operand tile;
//   C  the pointwise 1x1 (+ BN, ReLU) is a barrier-free MFMA pass over that tile (v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16,
//      weights streamed from L2 in the fragment order the row-block kernels use), results stored straight from the accumulators
//      (a lane holds 4 / 8 consecutive channels of a pixel: 16-byte stores).
// LDS is max(window, tile): 12 - 51 KB, three or more workgroups per CU overlap each other's phases.
// Same depthwise arithmetic as dwpw_kernel; the pointwise sums run over the same k order per MFMA, so results agree with the
// row-block kernels to the last bits of f32 rounding order (fp32: bit-identical accumulation chains per output).
#include <type_traits>

#include "lwp_internal.h"

namespace lwp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct f32x8 { f32x4 lo, hi; };

template <int ACT>
__device__ __forceinline__ float tl_act(float v) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;      // (bf16 path only: the f32 path keeps expm1f, see launcher)
    return v;
}

template <bool BF16, int C, int COUT, int S, int PH, int ACT = ACT_RELU>
__global__ void __launch_bounds__(256, 3) dwpw_tiled_kernel(DwPwParams p, int tiles_y, int tiles_x, int ntiles) {
    constexpr int PW = 8;
    constexpr int VEC = BF16 ? 8 : 4;                  // channels per 16-byte vector
    constexpr int ESZ = BF16 ? 2 : 4;
    constexpr int QN = C / VEC;                         // channel vectors per pixel
    constexpr int GRP = 256 / (QN * PW);               // row groups of the patch handled in parallel
    static_assert(GRP >= 1 && PH % GRP == 0, "patch / thread mismatch");
    constexpr int RPT = PH / GRP;                       // patch rows per thread
    constexpr int WR = (PH - 1) * S + 3, WC = (PW - 1) * S + 3;
    constexpr int NPX = PH * PW;                        // output pixels of the patch
    constexpr int RT = NPX / 16;                        // MFMA row tiles
    constexpr int LDT = C + (BF16 ? 16 : 4);            // operand tile row stride (elements); bf16: +32 B = conflict-free ds_read_b128 (rows of +16 B are two-way conflicted; f32 does not care)
    constexpr int NU = COUT / 32;                       // 32-channel output units
    constexpr int WCOLS = NU < 4 ? NU : 4;              // waves across the output channels
    constexpr int WROWS = 4 / WCOLS;                    // waves across the row tiles
    constexpr int UPW = NU / WCOLS;                     // units per wave
    constexpr int RTW = RT / WROWS;                     // row tiles per wave
    static_assert(RT % WROWS == 0 && NU % WCOLS == 0, "tile / wave mismatch");
    constexpr int NSTEP = C / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    typedef typename std::conditional<BF16, __bf16, float>::type elt;
    elt* win = (elt*)tsm;                               // [WR][WC][C]   (phase A / B)
    elt* til = (elt*)tsm;                               // [NPX][LDT]    (phase C, aliases the window)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_img = tiles_y * tiles_x;
    // PERSISTENT workgroups: block b walks patches remap(b), remap(b) + G, ...; the window of the NEXT patch is requested (into
    // registers) while the current one is convolved and multiplied, so the longest latency of the chain — the global loads —
    // is off the critical path.  XCD-aware order of the first index: the workgroups of one XCD take neighbouring patches.
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int qq = G >> 3, rem = G & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    const int wc = wave % WCOLS, wr = wave / WCOLS;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.pw_w, 0, 0x7fffffff, 0x00020000);
    constexpr int WV = BF16 ? 2 : 4;                    // 16-byte weight vectors per lane, unit and K step
    f32x4 wreg[2][UPW][WV];                             // two-step ring of the pointwise weights
    auto load_w = [&](int st, f32x4 (*dst)[WV]) {
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const unsigned soff = (unsigned)(st * NU + wc + u * WCOLS) * (BF16 ? 2048u : 4096u);
#pragma unroll
            for (int j = 0; j < WV; ++j)
                dst[u][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + j * 1024, soff, 0));
        }
    };
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * ESZ), 0x00020000);
    constexpr int NPIECE = WR * WC * QN, PPT = (NPIECE + 255) / 256;
    // window request of one patch: piece = (window pixel, 16-byte part), LDS order [pixel][C]; pieces outside the image read
    // offset 2^31 >= num_records and come back as zeros
    f32x4 pc[PPT];
    auto request = [&](int tile) {
        const int n = tile / per_img, r0 = tile - n * per_img;
        const int y0 = (r0 / tiles_x) * PH * S - 1, x0 = (r0 % tiles_x) * PW * S - 1;      // input position of window (0, 0)
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int i = tid + u * 256;
            const int px = i / QN, part = i % QN;
            const int wy = px / WC, wx = px % WC;
            const int y = y0 + wy, x = x0 + wx;
            const bool ok = i < NPIECE && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi;
            const unsigned off = ok ? (unsigned)((((n * p.Hi + y) * p.Wi + x) * p.in_ld + part * VEC) * ESZ) : 0x80000000u;
            pc[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0));
        }
    };
    int tile = bid;
    if (tile < ntiles) request(tile);
    // depthwise weights [9][C] f32 + bias [C]: kept in LDS behind the window / tile region and read tap by tap (in registers they
    // cost 40 / 80 VGPRs for the whole life of the persistent loop: a wave of occupancy)
    constexpr size_t WIN_B = (size_t)WR * WC * C * ESZ, TIL_B = (size_t)NPX * LDT * ESZ;
    float* Wd = (float*)(tsm + (((WIN_B > TIL_B ? WIN_B : TIL_B) + 15) & ~(size_t)15));
    for (int i = tid * 4; i < 10 * C; i += 256 * 4) *(f32x4*)(Wd + i) = *(const f32x4*)(p.dw_w + i);
    const int cq = tid % QN, lx = (tid / QN) % PW, grp = tid / (QN * PW);
    const int c = cq * VEC;
    auto wtap = [&](int t) -> f32x8 {
        f32x8 w;
        w.lo = *(const f32x4*)(Wd + t * C + c);
        w.hi = BF16 ? *(const f32x4*)(Wd + t * C + c + 4) : w.lo;
        return w;
    };
    auto at = [&](int wy, int wx) -> f32x8 {
        f32x8 r;
        if (BF16) {
            const bf16x8 v = *(const bf16x8*)((const __bf16*)win + (size_t)(wy * WC + wx) * C + c);
            r.lo = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            r.hi = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        } else {
            r.lo = *(const f32x4*)((const float*)win + (size_t)(wy * WC + wx) * C + c);
            r.hi = r.lo;
        }
        return r;
    };
    auto fma = [&](f32x8& a, const f32x8& x, const f32x8& w) {
        a.lo += x.lo * w.lo;
        if (BF16) a.hi += x.hi * w.hi;
    };
    const int i16 = lane & 15, q = lane >> 4;
    const int ly0 = grp * RPT;

    for (; tile < ntiles; tile += G) {
        const int n = tile / per_img, r0 = tile - n * per_img;
        const int ty0 = (r0 / tiles_x) * PH, tx0 = (r0 % tiles_x) * PW;
        // ---- phase A: the window (requested one patch ago) lands in LDS; the next patch's is requested at once.  The request is
        // unconditional (past the end: the last patch again) so that the compiler counts the loads in flight instead of draining them
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int i = tid + u * 256;
            if (i < NPIECE) *(f32x4*)((char*)win + (size_t)i * 16) = pc[u];
        }
        __syncthreads();
        request(tile + G < ntiles ? tile + G : tile);

        // ---- phase B: depthwise from LDS
        f32x8 res[RPT];
        if (S == 1) {                                         // rolling 3 x 3 window down the column strip
            f32x8 r[3][3];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) r[k + 1][kx] = at(ly0 + k, lx + kx);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { r[0][kx] = r[1][kx]; r[1][kx] = r[2][kx]; r[2][kx] = at(ly0 + i + 2, lx + kx); }
                f32x8 acc = wtap(9);
#pragma unroll
                for (int t = 0; t < 9; ++t) fma(acc, r[t / 3][t % 3], wtap(t));
                res[i] = acc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int ly = ly0 + i;
                f32x8 acc = wtap(9);
#pragma unroll
                for (int t = 0; t < 9; ++t) fma(acc, at(ly * S + t / 3, lx * S + t % 3), wtap(t));
                res[i] = acc;
            }
        }
        load_w(0, wreg[0]);
        __syncthreads();                                      // every thread is done with the window: the operand tile takes its place
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int pix = (ly0 + i) * PW + lx;
            const f32x4 lo = {tl_act<ACT>(res[i].lo.x), tl_act<ACT>(res[i].lo.y), tl_act<ACT>(res[i].lo.z), tl_act<ACT>(res[i].lo.w)};
            if (BF16) {
                const f32x4 hi = {tl_act<ACT>(res[i].hi.x), tl_act<ACT>(res[i].hi.y), tl_act<ACT>(res[i].hi.z), tl_act<ACT>(res[i].hi.w)};
                const bf16x8 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
                *(bf16x8*)((__bf16*)til + (size_t)pix * LDT + c) = o;
            } else {
                *(f32x4*)((float*)til + (size_t)pix * LDT + c) = lo;
            }
        }
        __syncthreads();

        // ---- phase C: pointwise.  wave (wc, wr): units wc + u * WCOLS, row tiles wr + a * WROWS
        f32x4 acc[UPW][RTW][2];
#pragma unroll
        for (int u = 0; u < UPW; ++u)
#pragma unroll
            for (int a = 0; a < RTW; ++a) { acc[u][a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 1 < NSTEP) load_w(st + 1, wreg[(st + 1) & 1]);
            if (BF16) {
                // lane (pixel i16, q) reads k = 32 st + 8 q .. + 7; weights tile t: wreg[.][u][t]
#pragma unroll
                for (int a = 0; a < RTW; ++a) {
                    const bf16x8 xv = *(const bf16x8*)((const __bf16*)til + (size_t)((wr + a * WROWS) * 16 + i16) * LDT + st * 32 + 8 * q);
#pragma unroll
                    for (int u = 0; u < UPW; ++u) {
                        acc[u][a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[st & 1][u][0]), xv, acc[u][a][0], 0, 0, 0);
                        acc[u][a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[st & 1][u][1]), xv, acc[u][a][1], 0, 0, 0);
                    }
                }
            } else {
                // f32 fragment order of dwpw_kernel: packed value v = uu * 8 + j * 2 + t -> register wreg[.][.][v >> 2][v & 3]; the lane
                // reads k = 32 st + 16 uu + 4 q .. + 3 of its pixel row
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    f32x4 av[RTW];
#pragma unroll
                    for (int a = 0; a < RTW; ++a) av[a] = *(const f32x4*)((const float*)til + (size_t)((wr + a * WROWS) * 16 + i16) * LDT + st * 32 + 16 * uu + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int v = uu * 8 + j * 2 + t;
#pragma unroll
                            for (int u = 0; u < UPW; ++u) {
                                const float b = wreg[st & 1][u][v >> 2][v & 3];
#pragma unroll
                                for (int a = 0; a < RTW; ++a) acc[u][a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, av[a][j], acc[u][a][t], 0, 0, 0);
                            }
                        }
                }
            }
        }
        // ---- store: D row (channel) = 4 q + reg, col (pixel) = i16.  f32: tile t holds channels 16 t + 4 q + r of the unit; bf16 (rows
        // packed permuted): tiles 0 / 1 hold channels 8 q + r / 8 q + 4 + r -> 8 consecutive channels per lane
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int nb = (wc + u * WCOLS) * 32;
#pragma unroll
            for (int a = 0; a < RTW; ++a) {
                const int pix = (wr + a * WROWS) * 16 + i16;
                const int yo = ty0 + pix / PW, xo = tx0 + pix % PW;
                if (yo >= p.Ho || xo >= p.Wo) continue;
                const int64_t m = ((int64_t)n * p.Ho + yo) * p.Wo + xo;
                if (BF16) {
                    const int nn = nb + 8 * q;
                    const f32x4 b0 = *(const f32x4*)(p.pw_b + nn), b1 = *(const f32x4*)(p.pw_b + nn + 4);
                    f32x4 v0 = acc[u][a][0] + b0, v1 = acc[u][a][1] + b1;
                    v0 = f32x4{tl_act<ACT>(v0.x), tl_act<ACT>(v0.y), tl_act<ACT>(v0.z), tl_act<ACT>(v0.w)};
                    v1 = f32x4{tl_act<ACT>(v1.x), tl_act<ACT>(v1.y), tl_act<ACT>(v1.z), tl_act<ACT>(v1.w)};
                    if (p.res) {                              // residual (cpm: x + trunk(x)) before the single rounding to bf16
                        const bf16x8 r = *(const bf16x8*)((const __bf16*)p.res + m * p.res_ld + nn);
                        v0 += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                        v1 += f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
                    }
                    const bf16x8 o = {(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
                    *(bf16x8*)((__bf16*)p.out + m * p.out_ld + nn) = o;
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int nn = nb + t * 16 + 4 * q;
                        f32x4 v = acc[u][a][t] + *(const f32x4*)(p.pw_b + nn);
                        v = f32x4{tl_act<ACT>(v.x), tl_act<ACT>(v.y), tl_act<ACT>(v.z), tl_act<ACT>(v.w)};
                        if (p.res) v += *(const f32x4*)(p.res + m * p.res_ld + nn);
                        *(f32x4*)(p.out + m * p.out_ld + nn) = v;
                    }
                }
            }
        }
        __syncthreads();                                      // the operand tile has been read: the next window may overwrite it
    }
}

template <bool BF16, int C, int COUT, int S, int PH, int ACT = ACT_RELU>
static hipError_t launch_tiled_t(const DwPwParams& p, hipStream_t s) {
    constexpr int ESZ = BF16 ? 2 : 4;
    constexpr int WR = (PH - 1) * S + 3, WC = 7 * S + 3;
    constexpr size_t win = (size_t)WR * WC * C * ESZ, til = (size_t)PH * 8 * (C + (BF16 ? 16 : 4)) * ESZ;
    constexpr size_t lds = (((win > til ? win : til) + 15) & ~(size_t)15) + (size_t)10 * C * sizeof(float);
    const int tiles_y = (p.Ho + PH - 1) / PH, tiles_x = (p.Wo + 7) / 8;
    const int64_t tiles = (int64_t)p.N * tiles_y * tiles_x;
    if (tiles >= (1ll << 31) - 1) return hipErrorInvalidValue;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) { hipError_t e = attr.ensure((const void*)dwpw_tiled_kernel<BF16, C, COUT, S, PH, ACT>, 96 * 1024); if (e != hipSuccess) return e; }
    // persistent grid: as many workgroups as the chip holds at once (LDS- and register-limited), each walking patches b, b + G, ...
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    int per_cu = (int)((160 * 1024) / (lds + 256));
    constexpr int CAP = BF16 ? 4 : (C == 128 ? 3 : 2);         // measured at batch 32, 32 -> 64: bf16 104 / 109 / 124 us for 4 / 3 / 2 per CU, f32 244 / 221 / 218
    if (per_cu > CAP) per_cu = CAP;
    if (per_cu < 1) per_cu = 1;
    if (T.dwpw_tiled_wgs > 0) per_cu = T.dwpw_tiled_wgs;        // LWP_DWPW_TILED_WGS (experiments): workgroups per CU
    int64_t grid = (int64_t)256 * per_cu;
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) grid = (int64_t)cus * per_cu;
    }
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL((dwpw_tiled_kernel<BF16, C, COUT, S, PH, ACT>), dim3((unsigned)grid), dim3(256), lds, s, p, tiles_y, tiles_x, (int)tiles);
    return hipGetLastError();
}

// *used = false: the caller takes the row-block kernel
template <bool BF16>
static hipError_t try_tiled(const DwPwParams& p, hipStream_t s, bool* used) {
    *used = false;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dwpw_tiled == 0) return hipSuccess;                  // LWP_DWPW_TILED "0": off (A/B); "1": at every size (tests)
    constexpr int ESZ = BF16 ? 2 : 4;
    const bool relu = p.act_dw == ACT_RELU && p.act_pw == ACT_RELU, elu = p.act_dw == ACT_ELU && p.act_pw == ACT_ELU;
    if (p.dil != 1 || !(relu || (elu && BF16))) return hipSuccess;       // (ELU blocks: bf16 only — the f32 path's ELU is expm1f)
    if (p.res && (!elu || (p.res_ld % 8) || (((uintptr_t)p.res) & 15))) return hipSuccess;
    if (p.in_ld != p.C || (p.out_ld % (BF16 ? 8 : 4)) || (((uintptr_t)p.out) & 15) || (((uintptr_t)p.in) & 15)) return hipSuccess;
    if ((int64_t)p.N * p.Hi * p.Wi * p.in_ld * ESZ >= (1ll << 31)) return hipSuccess;      // 32-bit buffer offsets
    const int64_t pixels = (int64_t)p.N * p.Ho * p.Wo;
    if (T.dwpw_tiled != 1 && pixels < (int64_t)96 * 1024) return hipSuccess;                // small maps: the row-block kernels fill the chip better
    *used = true;
    LWP_VARIANT(p, "dwpw_tiled<%s,%d,%d,s=%d>", BF16 ? "bf16" : "f32", p.C, p.cout, p.stride);
#define TL_CASE(C_, CO_, S_, PH_) if (relu && p.C == C_ && p.cout == CO_ && p.stride == S_) return launch_tiled_t<BF16, C_, CO_, S_, PH_>(p, s);
    const bool forced = T.dwpw_tiled == 1;
    TL_CASE(32, 64, 1, 16)
    // the wider blocks take 4-row patches: the NEXT patch's window is 4 - 8 sixteen-byte pieces per thread (8-row patches: 7 - 19,
    // which spill at the 168-VGPR budget of three workgroups per CU: 2 - 5x slower, measured; 128 -> 256 stride 2 spills even so).
    // Measured at batch 32 against the row-block kernels (us, round 3): 32 -> 64 bf16 130 -> 109, f32 296 -> 222;
    // 64 -> 128 stride 2 bf16 95 -> 86, f32 222 -> 185;  128 -> 128 f32 239 -> 222 (three workgroups per CU), bf16 83 -> 98 (slower:
    // 1.9x halo re-reads on 32-pixel patches) and the ELU blocks of the cpm trunk bf16 28 -> 35 (slower) — those two stay with the
    // row-block kernel unless forced (tests)
    TL_CASE(64, 128, 2, 4)
    if (!BF16 || forced) { TL_CASE(128, 128, 1, 4) }
#undef TL_CASE
    if constexpr (BF16) {
        if (forced && elu && p.C == 128 && p.cout == 128 && p.stride == 1) return launch_tiled_t<true, 128, 128, 1, 4, ACT_ELU>(p, s);   // cpm.trunk
    }
    *used = false;
    return hipSuccess;
}
hipError_t try_dwpw_tiled_f32(const DwPwParams& p, hipStream_t s, bool* used) { return try_tiled<false>(p, s, used); }
hipError_t try_dwpw_tiled_bf16(const DwPwParams& p, hipStream_t s, bool* used) { return try_tiled<true>(p, s, used); }

}  // namespace lwp
