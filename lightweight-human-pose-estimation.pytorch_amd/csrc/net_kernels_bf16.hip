// Network kernels for gfx950 (CDNA4), bf16 storage / bf16 MFMA / f32 accumulate path (BASELINE config 3).
// Activations are NHWC bf16; biases, depthwise weights, accumulation, activations and residual adds are f32; the
// stage outputs handed to the post-processing / API are f32 NCHW.
//
// MFMA orientation: the WEIGHTS are the A operand (rows = output channels) and the activations the B operand
// (columns = pixels).  The accumulator then has the pixel on the lane and 4 consecutive channels in consecutive
// registers, so the epilogue converts 4 values to bf16 and stores 8 bytes per lane, and NCHW f32 head outputs are
// written with consecutive lanes on consecutive pixels.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "lwp_internal.h"

namespace lwp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    // bf16 path: the result is rounded to 8 significant bits, so the hardware exp (abs error ~1e-7 on (-1, 0]) replaces
    // expm1f, which cost ~30 us per cpm.trunk block at batch 32
    if (act == ACT_ELU) return v > 0.0f ? v : __expf(v) - 1.0f;
    return v;
}

// (the stem kernel is the <true> instantiation of stem_kernel in net_kernels.hip)

// ---------------------------------------------------------------------------------------- fused depthwise -> pointwise
// Same two-phase structure as the f32 kernel (net_kernels.hip): phase 1 computes the workgroup's depthwise row block
// [BM][C] (f32 math on bf16 inputs) into LDS as bf16; phase 2 is a barrier-free GEMM with v_mfma_f32_16x16x32_bf16.
// A lane (i = lane&15, q = lane>>4) reads 8 consecutive k (16 B) at k = 32s + 8q: exactly the operand lane map, for
// the activation tile (from LDS) and for the fragment-packed weights [k-step][wave][tile][lane][8] (from L2) alike.
template <int BM, int NW>
__global__ void __launch_bounds__(NW * 64) dwpw_bf16_kernel(DwPwParams p) {
    constexpr int NT = NW * 64;
    constexpr int RT = BM / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
    __bf16* At = (__bf16*)dsm_raw;                   // [BM][C + 8]
    const int ldA = p.C + 8;
    const __bf16* in = (const __bf16*)p.in;
    const __bf16* pw = (const __bf16*)p.pw_w;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.y * NW + (tid >> 6);
    const int nwt = gridDim.y * NW;
    const int i16 = lane & 15, q = lane >> 4;
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs): give every XCD a contiguous run of row
    // blocks, so that the 3-row input windows of neighbouring blocks are fetched into ONE L2 instead of all eight
    int bid = blockIdx.x;
    if (gridDim.y == 1) {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    const int64_t m0 = (int64_t)bid * BM;
    const int nsteps = p.C / 32;

    // weight fragments: PF statically rotated register buffers (K loop unrolled by PF, no copies): PF-1 steps of the
    // weight stream stay in flight per wave — the bf16 MFMA phase is far too short to hide a 1-step prefetch
    constexpr int PF = 2;
    bf16x8 bw[PF][2];
    auto load_b = [&](int step, bf16x8* dst) {
        const __bf16* src = pw + ((int64_t)(step * nwt + wave) * 2) * 512 + lane * 8;
        dst[0] = *(const bf16x8*)src;
        dst[1] = *(const bf16x8*)(src + 512);
    };
#pragma unroll
    for (int j = 0; j < PF - 1; ++j)
        if (j < nsteps) load_b(j, bw[j]);

    // ---- phase 1: one 4-channel chunk per thread (its 9 weight vectors + bias stay in registers), walking groups of
    // PXG consecutive output pixels; inside one image row (stride 1, dilation 1) the group shares a 3 x (PXG+2) input
    // window — the kernel is bound by the number of vector-memory instructions (TA busy ~ kernel time), not by bytes.
    constexpr int PXG = 4;
    static_assert(BM % PXG == 0, "row block must hold whole pixel groups");
    const int cg = p.C >> 2;
    {
        const int c = (tid % cg) * 4;
        f32x4 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const f32x4*)(p.dw_w + t * p.C + c);
        const f32x4 bias = *(const f32x4*)(p.dw_w + 9 * p.C + c);
        auto finish = [&](f32x4 a, int row) {
            const bf16x4 o = {(__bf16)act_f(a.x, p.act_dw), (__bf16)act_f(a.y, p.act_dw), (__bf16)act_f(a.z, p.act_dw), (__bf16)act_f(a.w, p.act_dw)};
            *(bf16x4*)(At + row * ldA + c) = o;
        };
        for (int grp = tid / cg; grp < BM / PXG; grp += NT / cg) {
            const int row0 = grp * PXG;
            const int64_t m = m0 + row0;
            const int64_t mm = m < M ? m : 0;
            const int xo = (int)(mm % p.Wo), yo = (int)((mm / p.Wo) % p.Ho);
            const int64_t img = mm / ((int64_t)p.Wo * p.Ho);
            if (p.stride == 1 && p.dil == 1 && m + PXG <= M && xo + PXG <= p.Wo) {
                const __bf16* base = in + ((img * p.Hi + yo) * p.Wi + xo) * p.in_ld + c;
                bf16x4 win[3][PXG + 2];                 // shared 3 x (PXG + 2) input window
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = yo + ky - 1;
                    const bool rok = yy >= 0 && yy < p.Hi;
#pragma unroll
                    for (int j = 0; j < PXG + 2; ++j) {
                        const int xx = xo + j - 1;
                        const __bf16* src = (rok && xx >= 0 && xx < p.Wi) ? base + ((int64_t)(ky - 1) * p.Wi + (j - 1)) * p.in_ld : (const __bf16*)p.zeros;
                        win[ky][j] = *(const bf16x4*)src;
                    }
                }
#pragma unroll
                for (int i = 0; i < PXG; ++i) {
                    f32x4 a = bias;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const bf16x4 w4 = win[t / 3][i + t % 3];
                        const f32x4 v = {(float)w4[0], (float)w4[1], (float)w4[2], (float)w4[3]};
                        a += v * wv[t];
                    }
                    finish(a, row0 + i);
                }
                continue;
            }
            // (a shared window for dilation 2 was tried: its 24 live vectors spill at the 128-VGPR cap of the 16-wave
            //  workgroups and slowed every layer by 30-40 %)
#pragma unroll 1
            for (int i = 0; i < PXG; ++i) {           // general path: borders of the row block, stride 2, dilation 2
                const int64_t mi = m0 + row0 + i;
                const bool ok = mi < M;
                const int64_t mq = ok ? mi : 0;
                const int xi = (int)(mq % p.Wo), yi = (int)((mq / p.Wo) % p.Ho);
                const int64_t im = mq / ((int64_t)p.Wo * p.Ho);
                const int yc = yi * p.stride, xc = xi * p.stride;
                const __bf16* base = in + ((im * p.Hi + yc) * p.Wi + xc) * p.in_ld + c;
                bf16x4 x[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = (t / 3 - 1) * p.dil, dx = (t % 3 - 1) * p.dil;
                    const bool inb = ok && yc + dy >= 0 && yc + dy < p.Hi && xc + dx >= 0 && xc + dx < p.Wi;
                    const __bf16* src = inb ? base + ((int64_t)dy * p.Wi + dx) * p.in_ld : (const __bf16*)p.zeros;
                    x[t] = *(const bf16x4*)src;
                }
                f32x4 a = bias;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const f32x4 v = {(float)x[t][0], (float)x[t][1], (float)x[t][2], (float)x[t][3]};
                    a += v * wv[t];
                }
                finish(a, row0 + i);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: acc[a][t] = W-tile(t) x X-tile(a):  row = channel 16t + 4q + reg,  col = pixel a*16 + i16
    f32x4 acc[RT][2];
#pragma unroll
    for (int a = 0; a < RT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const __bf16* x_lane = At + i16 * ldA + 8 * q;
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
        for (int jb = 0; jb < PF; ++jb) {
            const int step = s0 + jb;
            if (step >= nsteps) break;
            if (step + PF - 1 < nsteps) load_b(step + PF - 1, bw[(jb + PF - 1) % PF]);
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                const bf16x8 xv = *(const bf16x8*)(x_lane + a * 16 * ldA + step * 32);
                acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[jb][0], xv, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[jb][1], xv, acc[a][1], 0, 0, 0);
            }
        }
    }
    // epilogue
    __bf16* out = (__bf16*)p.out;
    const __bf16* res = (const __bf16*)p.res;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = wave * 32 + t * 16 + 4 * q;
        const f32x4 bias = *(const f32x4*)(p.pw_b + n);
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            const int64_t m = m0 + a * 16 + i16;
            if (m < M) {
                f32x4 v = acc[a][t] + bias;
                v.x = act_f(v.x, p.act_pw); v.y = act_f(v.y, p.act_pw); v.z = act_f(v.z, p.act_pw); v.w = act_f(v.w, p.act_pw);
                if (res) {
                    const bf16x4 r = *(const bf16x4*)(res + m * p.res_ld + n);
                    v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
                }
                bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *(bf16x4*)(out + m * p.out_ld + n) = o;
            }
        }
    }
}

template <int BM, int NW>
static hipError_t launch_dwpw_bf16_t(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const size_t lds = (size_t)BM * (p.C + 8) * 2;
    const int nsplit = (p.cout / 32) / NW;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) {
        hipError_t e = attr.ensure((const void*)dwpw_bf16_kernel<BM, NW>, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((dwpw_bf16_kernel<BM, NW>), dim3((unsigned)((M + BM - 1) / BM), nsplit), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_dwpw_bf16(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const int nw = p.cout / 32;
    int bm = 16;
    if (M / 16 >= 2048) bm = 32;
    if (M / 32 >= 2048) bm = 64;
    if (M / 64 >= 1024 && nw >= 8) bm = 128;         // >= 256 output channels: halve the per-workgroup weight stream (measured 3-8 %)
    static const char* env = getenv("LWP_DWPW_BM");
    if (env) bm = atoi(env);
    while (bm > 16 && (size_t)bm * (p.C + 8) * 2 > 150 * 1024) bm >>= 1;
    if (bm == 128 && nw < 4) bm = 64;
    // every thread must own a whole 8-channel chunk column: NW*64 threads must be a multiple of C/8 (always true here)
#define DPH_CASE(BM_, NW_) if (bm == BM_ && nw == NW_) return launch_dwpw_bf16_t<BM_, NW_>(p, s);
    DPH_CASE(16, 2) DPH_CASE(32, 2) DPH_CASE(64, 2)
    DPH_CASE(16, 4) DPH_CASE(32, 4) DPH_CASE(64, 4)
    DPH_CASE(16, 8) DPH_CASE(32, 8) DPH_CASE(64, 8)
    DPH_CASE(16, 16) DPH_CASE(32, 16) DPH_CASE(64, 16) DPH_CASE(128, 16) DPH_CASE(128, 8) DPH_CASE(128, 4)
#undef DPH_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------- implicit GEMM (1x1, dense 3x3)
// Workgroup tile BM pixels x BN channels, (BM/(32 RM)) x (BN/(32 RN)) waves, each wave RM x RN accumulators of
// v_mfma_f32_32x32x16_bf16 (register blocking keeps LDS reads at <= half the LDS rate).  K walked 64 channels per
// step inside one tap; tiles register-staged into LDS rows of 64 + 8 bf16 (144 B: conflict-free ds_read_b128),
// double-buffered, one barrier per step.  A lane (i = lane&31, h = lane>>5) reads k = 16s + 8h .. +7.
constexpr int HBK = 64;
constexpr int HLD = HBK + 8;

template <int BM, int BN, int RM, int RN>
__global__ void __launch_bounds__((BM / (32 * RM)) * (BN / (32 * RN)) * 64) gemm_bf16_kernel(GemmParams p) {
    constexpr int WM = BM / (32 * RM), WN = BN / (32 * RN);
    constexpr int NT = WM * WN * 64;
    constexpr int A_CH = BM * 8, B_CH = BN * 8;       // 16-byte chunks per tile
    constexpr int A_PER = (A_CH + NT - 1) / NT, B_PER = (B_CH + NT - 1) / NT;
    static_assert(A_CH % NT == 0 && B_CH % NT == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm_raw[];
    __bf16* As = (__bf16*)hsm_raw;                    // [2][BM][HLD]
    __bf16* Bs = As + 2 * BM * HLD;                   // [2][BN][HLD]
    const __bf16* in = (const __bf16*)p.in;
    const __bf16* wgt = (const __bf16*)p.w;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int i32 = lane & 31, h = lane >> 5;
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int ntn = p.cout_pad / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    const int tile_m = bid / ntn, tile_n = bid % ntn;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    int a_lds[A_PER], a_y[A_PER], a_x[A_PER];
    int64_t a_base[A_PER];
    bool a_ok[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int ch = tid + i * NT;
        const int row = ch >> 3, col = (ch & 7) * 8;
        a_lds[i] = row * HLD + col;
        const int64_t m = m0 + row;
        a_ok[i] = m < M;
        const int64_t mm = a_ok[i] ? m : 0;
        a_x[i] = (int)(mm % p.W);
        a_y[i] = (int)((mm / p.W) % p.H);
        a_base[i] = mm * p.in_ld + col;
    }
    int b_off[B_PER], b_lds[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = tid + i * NT;
        const int row = ch >> 3, col = (ch & 7) * 8;
        b_off[i] = row * p.cin_pad + col;
        b_lds[i] = row * HLD + col;
    }
    const int ksteps_per_tap = p.cin_pad / HBK;
    const int nsteps = p.ks * p.ks * ksteps_per_tap;

    bf16x8 a_reg[A_PER], b_reg[B_PER];
    auto load_step = [&](int step) {
        const int tap = step / ksteps_per_tap;
        const int c0 = (step - tap * ksteps_per_tap) * HBK;
        int dy = 0, dx = 0;
        if (p.ks == 3) { dy = (tap / 3 - 1) * p.dil; dx = (tap % 3 - 1) * p.dil; }
        const int64_t shift = ((int64_t)dy * p.W + dx) * p.in_ld + c0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const __bf16* src = ok ? in + a_base[i] + shift : (const __bf16*)p.zeros;
            a_reg[i] = *(const bf16x8*)src;
        }
        const __bf16* wt = wgt + ((int64_t)tap * p.cout_pad + n0) * p.cin_pad + c0;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) b_reg[i] = *(const bf16x8*)(wt + b_off[i]);
    };
    auto store_step = [&](int buf) {
        __bf16* a = As + buf * BM * HLD;
        __bf16* b = Bs + buf * BN * HLD;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) *(bf16x8*)(a + a_lds[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *(bf16x8*)(b + b_lds[i]) = b_reg[i];
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_step(0);
    store_step(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_step(step + 1);
        const __bf16* a = As + buf * BM * HLD + (wm * 32 * RM + i32) * HLD + 8 * h;
        const __bf16* b = Bs + buf * BN * HLD + (wn * 32 * RN + i32) * HLD + 8 * h;
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            bf16x8 xv[RM], wv[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) xv[i] = *(const bf16x8*)(a + i * 32 * HLD + 16 * s);
#pragma unroll
            for (int j = 0; j < RN; ++j) wv[j] = *(const bf16x8*)(b + j * 32 * HLD + 16 * s);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[j], xv[i], acc[i][j], 0, 0, 0);
        }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D row = channel (e&3) + 8*(e>>2) + 4*h of the 32-channel tile, col = pixel i32
    __bf16* out = (__bf16*)p.out;
    const __bf16* res = (const __bf16*)p.res;
    const int64_t HW = (int64_t)p.H * p.W;
    const bool vec_ok = ((p.out_ld & 3) == 0) && ((((uintptr_t)out) & 7) == 0) && (!res || (((p.res_ld & 3) == 0) && ((((uintptr_t)res) & 7) == 0)));
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        const int64_t m = m0 + (wm * RM + i) * 32 + i32;
        if (m >= M) continue;
        const int64_t img = m / HW, pix = m - img * HW;
        const int c0 = p.out_split > 0 ? p.out_split : p.cout;
        auto store_nchw = [&](int ch, float val) {       // stage outputs (NCHW); merged heads split at out_split
            if (ch < c0) { if (p.out_nchw) p.out_nchw[(img * c0 + ch) * HW + pix] = val; }
            else if (p.out_nchw2) p.out_nchw2[(img * (p.cout - c0) + (ch - c0)) * HW + pix] = val;
        };
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * RN + j) * 32 + 8 * g + 4 * h;
                if (n >= p.cout) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const f32x4 bias = *(const f32x4*)(p.bias + n);       // bias is padded to cout_pad
                v += bias;
                v.x = act_f(v.x, p.act); v.y = act_f(v.y, p.act); v.z = act_f(v.z, p.act); v.w = act_f(v.w, p.act);
                if (vec_ok && n + 3 < p.cout) {
                    if (res) {
                        const bf16x4 r = *(const bf16x4*)(res + m * p.res_ld + n);
                        v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
                    }
                    bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                    *(bf16x4*)(out + m * p.out_ld + n) = o;
                    if (p.out_nchw || p.out_nchw2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) store_nchw(n + e, v[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.cout) {
                            float u = v[e];
                            if (res) u += (float)res[m * p.res_ld + n + e];
                            out[m * p.out_ld + n + e] = (__bf16)u;
                            if (p.out_nchw || p.out_nchw2) store_nchw(n + e, u);
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int RM, int RN>
static hipError_t launch_gemm_bf16_t(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int64_t tiles = ((M + BM - 1) / BM) * (p.cout_pad / BN);
    constexpr int NT = (BM / (32 * RM)) * (BN / (32 * RN)) * 64;
    const size_t lds = (size_t)2 * (BM + BN) * HLD * 2;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) {
        hipError_t e = attr.ensure((const void*)gemm_bf16_kernel<BM, BN, RM, RN>, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, RM, RN>), dim3((unsigned)tiles), dim3(NT), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_gemm_bf16(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    // experiments: LWP_GEMMH = "BM,BN,RM,RN"
    static const char* env = getenv("LWP_GEMMH");
    int bm = 0, bn = 0, rm = 0, rn = 0;
    if (env && sscanf(env, "%d,%d,%d,%d", &bm, &bn, &rm, &rn) == 4 && p.cout_pad % bn == 0) {
    } else if (p.cout_pad % 128 == 0 && ((M + 127) / 128) * (p.cout_pad / 128) >= 512) {
        bm = 128; bn = 128; rm = 2; rn = 1;           // 8 waves x (64 x 32): 6-25 % faster than 4 waves x (64 x 64) on every layer at batch 32
    } else if (((M + 127) / 128) * (p.cout_pad / 64) >= 512) {
        bm = 128; bn = 64; rm = 1; rn = 1;            // 8 waves x (32 x 32): the N = 64 head GEMMs, ~10 % over 4 waves x (64 x 32)
    } else {
        bm = 64; bn = 64; rm = 1; rn = 1;             // 4 waves x (32 x 32): small problems
    }
#define GH_CASE(BM_, BN_, RM_, RN_) if (bm == BM_ && bn == BN_ && rm == RM_ && rn == RN_) return launch_gemm_bf16_t<BM_, BN_, RM_, RN_>(p, s);
    GH_CASE(128, 128, 2, 2) GH_CASE(128, 64, 2, 1) GH_CASE(64, 64, 1, 1) GH_CASE(256, 128, 2, 2) GH_CASE(128, 128, 2, 1) GH_CASE(128, 128, 1, 1) GH_CASE(128, 64, 1, 1) GH_CASE(256, 128, 2, 1)
#undef GH_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------- layout helper
__global__ void __launch_bounds__(256) nchw_from_nhwc_bf16_kernel(const __bf16* src, int src_ld, float* dst, int N, int HW, int C) {
    const int64_t total = (int64_t)N * C * HW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int pix = (int)(idx % HW);
    const int64_t t = idx / HW;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    dst[idx] = (float)src[((int64_t)n * HW + pix) * src_ld + c];
}
hipError_t launch_nchw_from_nhwc_bf16(const void* src, int src_ld, float* dst, int N, int HW, int C, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    hipLaunchKernelGGL(nchw_from_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const __bf16*)src, src_ld, dst, N, HW, C);
    return hipGetLastError();
}

}  // namespace lwp
