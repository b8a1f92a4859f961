// Network kernels for gfx950 (CDNA4), fp32 path.  Activations are NHWC float32.
//
//   stem_kernel   3x3 stride-2 conv 3->32 (+folded BN, ReLU), reads the NCHW input planes directly.
//   dw_kernel     depthwise 3x3, stride 1/2, dilation 1/2, (+folded BN) ReLU | ELU.  HBM-bound:
//                 16-byte channel vectors per lane, consecutive lanes on consecutive channels, so
//                 every wave instruction moves whole 1-KiB rows.
//   gemm_kernel   1x1 and dense 3x3 convs as an implicit GEMM on the f32-input matrix cores
//                 (v_mfma_f32_32x32x2_f32: exact f32 fmaf chains, 157 TFLOP/s peak).  Tiles are
//                 register-staged into padded LDS rows (36 floats: conflict-free ds_read_b128),
//                 double-buffered, one barrier per 32-deep K step.
#include <cstdio>
#include <cstdlib>

#include <type_traits>
#include "lwp_internal.h"

namespace lwp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// exp(v) - 1 for v <= 0 with expm1's RELATIVE accuracy at a third of its instructions (the library expm1f cost ~30 us per cpm.trunk
// block at batch 32): the degree-7 Taylor polynomial on (-0.5, 0] (truncation < v^8 / 40320: 2.5e-7 relative at -0.5, far less
// towards 0, where exp(v) - 1 would cancel), the hardware exp minus one below (the result is <= -0.39 there, so the subtraction
// loses nothing).  Max deviation from expm1f over (-inf, 0]: a few 1e-7 relative.
__device__ __forceinline__ float elu_negative(float v) {
    const float p = v * (1.f + v * (0.5f + v * (1.f / 6.f + v * (1.f / 24.f + v * (1.f / 120.f + v * (1.f / 720.f + v * (1.f / 5040.f)))))));
    return v > -0.5f ? p : __expf(v) - 1.0f;
}
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_ELU) return v > 0.0f ? v : elu_negative(v);
    return v;
}

// ---------------------------------------------------------------------------------------- stem
// conv 3x3 stride 2, 3 -> 32 (+folded BN, ReLU) from the NCHW float input to NHWC (f32 or bf16).
// One workgroup = 8 x 32 output pixels.  The (17 x 65) x 3 input region is loaded into LDS with coalesced row
// loads (the first version gathered 27 scattered floats per thread and was bound by the texture addresser:
// 62 % issue stalls in the PMC profile); one thread computes one pixel x all 32 channels with the weights
// arriving through scalar loads (uniform addresses); results are transposed through LDS so the write-out is
// 16 bytes per lane on consecutive addresses (4 KiB contiguous per tile row).
constexpr int ST_TX = 32;
// DBG (LWP_ABLATION builds): 1 no input staging, 2 no FMAs, 4 no write-out.
// Ablation of the first version at batch 32 fp32 (158 us): input staging alone 59 us (one 4-byte load per element with two
// integer divisions), the 864 FMAs per pixel 50 us, the write-out 29 us, launch floor 7.5 — and hardly any overlap between
// them (3 workgroups per CU).  Now: the region is staged with ALIGNED 16-byte row loads (18 per row and channel: columns
// 2*x0 - 4 .. 2*x0 + 67; image edges fall on multiples of 4, so a quad is either inside or zero) when W % 4 == 0, and the
// output staging tile re-uses the input tile's LDS (36.9 KB per workgroup: 4 workgroups per CU).
// PY = vertically adjacent output pixels per thread (tile = ST_TY*PY rows x 32 columns).  The 27 x 32 weights reach the FMAs
// as scalar operands: every (tap, channel) needs two s_load_dwordx16 whose latency the compiler does not hide (it waits
// lgkmcnt(0) in front of each group of 32 v_fmac), so with PY = 2 each scalar load feeds 64 FMAs instead of 32.
// PY = vertically adjacent output pixels per thread (tile = ST_TY*PY rows x 32 columns); every (tap, channel) needs two
// s_load_dwordx16 of weights, which PY = 2 amortises over 64 FMAs — measured equal to PY = 1 at batch 32, kept for experiments.
// (A persistent form that requests the next tile's quads before the FMAs was measured and dropped: 143 us against 104 —
// the parked quads cost a wave of occupancy and, inside a tile loop, the weights are only scalar-loaded when read through
// the constant address space.)
// WL: the 27 x 32 weights are staged in LDS and read back as broadcast vectors instead of arriving through scalar loads.  With
// one or two waves per SIMD (batch 1: 1012 one-wave workgroups) nothing hides the 54 dependent s_load_dwordx16 round trips
// of a pixel; with four or more (batch 32) the scalar form is the cheaper one (no extra LDS instructions).
template <bool BF16, int ST_TY, int PY = 1, int DBG = 0, bool WL = false>
__global__ void __launch_bounds__(ST_TY * ST_TX) stem_kernel(StemParams p) {
    constexpr int NT = ST_TY * ST_TX;
    constexpr int TROWS = ST_TY * PY;                              // output rows per tile
    constexpr int IR = 2 * TROWS + 1;                              // input rows per channel
    constexpr int IQ = 18, ICP = IQ * 4;                           // 18 aligned quads = 72 columns: LDS column j <-> x = 2*x0 - 4 + j
    constexpr int OLD = BF16 ? 20 : 36;                            // staged output row stride in dwords (80 B / 144 B)
    constexpr int IN_FLOATS = 3 * IR * ICP, OUT_FLOATS = NT * OLD;
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS > OUT_FLOATS ? IN_FLOATS : OUT_FLOATS];
    float* s_in = smem;                                            // [3][IR][ICP]
    float* s_out = smem;                                           // [NT][OLD]  (after the compute phase, one pixel of every thread at a time)
    __shared__ __attribute__((aligned(16))) float s_w[WL ? 27 * 32 : 4];
    const int tid = threadIdx.x;
    if (WL) {
        for (int i = tid * 4; i < 27 * 32; i += NT * 4) *(f32x4*)(s_w + i) = *(const f32x4*)(p.w + i);
    }
    const int tiles_x = (p.Wo + ST_TX - 1) / ST_TX;
    const int x0 = (blockIdx.x % tiles_x) * ST_TX, y0 = (blockIdx.x / tiles_x) * TROWS;
    const int n = blockIdx.y;
    const float* in = p.in + (int64_t)n * 3 * p.H * p.W;
    if (!(DBG & 1)) {
        if ((p.W & 3) == 0 && (((uintptr_t)p.in) & 15) == 0 && (int64_t)p.N * 3 * p.H * p.W * 4 < (1ll << 31)) {
            // every quad of the thread is requested before the first LDS store (buffer loads: one 32-bit offset per lane, a quad
            // outside the image reads offset 2^31 >= num_records and comes back as zeros)
            const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)p.N * 3 * p.H * p.W * 4), 0x00020000);
            constexpr int ITEMS = 3 * IR * IQ, PER = (ITEMS + NT - 1) / NT;
            f32x4 v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + u * NT;
                const int qd = i % IQ, r = (i / IQ) % IR, c = i / (IQ * IR);
                const int yi = 2 * y0 - 1 + r, xi = 2 * x0 - 4 + 4 * qd;
                const bool ok = i < ITEMS && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W;
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, ok ? (unsigned)((((n * 3 + c) * p.H + yi) * p.W + xi) * 4) : 0x80000000u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + u * NT;
                const int qd = i % IQ, r = (i / IQ) % IR, c = i / (IQ * IR);
                if (i < ITEMS) *(f32x4*)(s_in + (c * IR + r) * ICP + 4 * qd) = v[u];
            }
        } else if ((p.W & 3) == 0 && (((uintptr_t)p.in) & 15) == 0) {
            for (int i = tid; i < 3 * IR * IQ; i += NT) {
                const int qd = i % IQ, r = (i / IQ) % IR, c = i / (IQ * IR);
                const int yi = 2 * y0 - 1 + r, xi = 2 * x0 - 4 + 4 * qd;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (yi >= 0 && yi < p.H && xi >= 0 && xi < p.W) v = *(const f32x4*)(in + ((int64_t)c * p.H + yi) * p.W + xi);
                *(f32x4*)(s_in + (c * IR + r) * ICP + 4 * qd) = v;
            }
        } else {
            for (int i = tid; i < 3 * IR * ICP; i += NT) {
                const int col = i % ICP, r = (i / ICP) % IR, c = i / (ICP * IR);
                const int yi = 2 * y0 - 1 + r, xi = 2 * x0 - 4 + col;
                float v = 0.f;
                if (yi >= 0 && yi < p.H && xi >= 0 && xi < p.W) v = in[((int64_t)c * p.H + yi) * p.W + xi];
                s_in[i] = v;
            }
        }
    }
    __syncthreads();
    const int ty = tid / ST_TX, tx = tid % ST_TX;                  // the thread's pixels: rows ty*PY + j, column tx
    // channel PAIRS per register pair: v_pk_fma_f32 does two of the 864 FMAs of a pixel per issue slot (a plain v_fmac of a
    // 64-lane wave takes 4 cycles: 42 us of issue at batch 32); every output keeps its own fmaf chain in (ky, kx, ci) order
    f32x2 acc2[PY][16];
#pragma unroll
    for (int j = 0; j < PY; ++j)
#pragma unroll
        for (int o = 0; o < 16; ++o) acc2[j][o] = f32x2{p.bias[2 * o], p.bias[2 * o + 1]};
    if (WL) {
        // the eight weight vectors of tap t + 1 are read (LDS broadcast) while tap t's sixteen packed FMAs issue
        f32x4 wq[2][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wq[0][q] = *(const f32x4*)(s_w + 4 * q);
#pragma unroll
        for (int t = 0; t < ((DBG & 2) ? 0 : 27); ++t) {               // t = (ky * 3 + kx) * 3 + ci: the fmaf chain of every output is fixed
            const int ky = t / 9, kx = (t / 3) % 3, ci = t % 3;
            if (t + 1 < 27) {
#pragma unroll
                for (int q = 0; q < 8; ++q) wq[(t + 1) & 1][q] = *(const f32x4*)(s_w + (t + 1) * 32 + 4 * q);
            }
            f32x2 v[PY];
#pragma unroll
            for (int j = 0; j < PY; ++j) { const float x = s_in[(ci * IR + 2 * (ty * PY + j) + ky) * ICP + 2 * tx + 3 + kx]; v[j] = f32x2{x, x}; }
#pragma unroll
            for (int o = 0; o < 16; ++o)
#pragma unroll
                for (int j = 0; j < PY; ++j) acc2[j][o] = __builtin_elementwise_fma(v[j], f32x2{wq[t & 1][o >> 1][(o & 1) * 2], wq[t & 1][o >> 1][(o & 1) * 2 + 1]}, acc2[j][o]);
        }
    } else {
#pragma unroll
    for (int ky = 0; ky < ((DBG & 2) ? 0 : 3); ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {                       // (ky, kx, ci) order: the fmaf chain of every output is fixed
                const float* w = p.w + ((ky * 3 + kx) * 3 + ci) * 32;      // uniform address: scalar loads
                f32x2 v[PY];
#pragma unroll
                for (int j = 0; j < PY; ++j) { const float x = s_in[(ci * IR + 2 * (ty * PY + j) + ky) * ICP + 2 * tx + 3 + kx]; v[j] = f32x2{x, x}; }      // x = 2*(x0 + tx) - 1 + kx
#pragma unroll
                for (int o = 0; o < 16; ++o)
#pragma unroll
                    for (int j = 0; j < PY; ++j) acc2[j][o] = __builtin_elementwise_fma(v[j], f32x2{w[2 * o], w[2 * o + 1]}, acc2[j][o]);
            }
    }
    float acc[PY][32];
#pragma unroll
    for (int j = 0; j < PY; ++j)
#pragma unroll
        for (int o = 0; o < 16; ++o) { acc[j][2 * o] = acc2[j][o][0]; acc[j][2 * o + 1] = acc2[j][o][1]; }
    constexpr int CPP = BF16 ? 4 : 8;                               // 16-byte chunks per pixel
#pragma unroll
    for (int j = 0; j < PY; ++j) {
        __syncthreads();                                            // the tile (input, or the previous pass's output) is dead: the space is re-used
        if (BF16) {
            __bf16* so = (__bf16*)s_out + tid * (OLD * 2);
#pragma unroll
            for (int o = 0; o < 32; ++o) so[o] = (__bf16)fmaxf(acc[j][o], 0.f);
        } else {
            float* so = s_out + tid * OLD;
#pragma unroll
            for (int o = 0; o < 32; o += 4) *(f32x4*)(so + o) = f32x4{fmaxf(acc[j][o], 0.f), fmaxf(acc[j][o + 1], 0.f), fmaxf(acc[j][o + 2], 0.f), fmaxf(acc[j][o + 3], 0.f)};
        }
        __syncthreads();
        for (int qd = tid; qd < ((DBG & 4) ? 0 : NT * CPP); qd += NT) {
            const int pix = qd / CPP, part = qd % CPP;                  // pixel = thread `pix`'s j-th: row (pix / 32) * PY + j
            const int yo = y0 + (pix / ST_TX) * PY + j, xo = x0 + pix % ST_TX;
            if (yo < p.Ho && xo < p.Wo) {
                const f32x4 v = *(const f32x4*)(s_out + pix * OLD + part * 4);
                char* dst = (char*)p.out + ((((int64_t)n * p.Ho + yo) * p.Wo + xo) * 32) * (BF16 ? 2 : 4) + part * 16;
                *(f32x4*)dst = v;
            }
        }
    }
}

template <bool BF16>
static hipError_t launch_stem_t(const StemParams& p, hipStream_t s) {
    // tile height: 8 rows when that still gives every CU several workgroups, else 4 or 2 (batch 1: 253 -> 1012 workgroups)
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    const int tx = (p.Wo + ST_TX - 1) / ST_TX;
    int ty = ((int64_t)tx * ((p.Ho + 7) / 8) * p.N >= 2048) ? 8 : (((int64_t)tx * ((p.Ho + 3) / 4) * p.N >= 2048) ? 4 : 2);
    if (T.stem_ty) ty = T.stem_ty;                            // LWP_STEM_TY
    const bool wl = T.stem_wl != 0;                           // LWP_STEM_WL "0" | "1": weights through scalar loads | LDS in the small-tile variants (A/B)
    LWP_VARIANT(p, "stem<ty=%d,wl=%d>", ty, (ty == 4 || ty == 2) && wl ? 1 : 0);
#ifdef LWP_ABLATION
    const int d = T.stem_debug;
#define ST_DBG(D_) if (ty == 8 && d == D_) { hipLaunchKernelGGL((stem_kernel<BF16, 8, 1, D_>), dim3(tx * ((p.Ho + 7) / 8), p.N), dim3(256), 0, s, p); return hipGetLastError(); }
    ST_DBG(1) ST_DBG(2) ST_DBG(4) ST_DBG(3) ST_DBG(5) ST_DBG(6) ST_DBG(7)
#undef ST_DBG
#endif
    if (ty == 16) hipLaunchKernelGGL((stem_kernel<BF16, 8, 2>), dim3(tx * ((p.Ho + 15) / 16), p.N), dim3(256), 0, s, p);
    else if (ty == 8) hipLaunchKernelGGL((stem_kernel<BF16, 8>), dim3(tx * ((p.Ho + 7) / 8), p.N), dim3(256), 0, s, p);
    else if (ty == 4 && wl) hipLaunchKernelGGL((stem_kernel<BF16, 4, 1, 0, true>), dim3(tx * ((p.Ho + 3) / 4), p.N), dim3(128), 0, s, p);
    else if (ty == 2 && wl) hipLaunchKernelGGL((stem_kernel<BF16, 2, 1, 0, true>), dim3(tx * ((p.Ho + 1) / 2), p.N), dim3(64), 0, s, p);
    else if (ty == 4) hipLaunchKernelGGL((stem_kernel<BF16, 4>), dim3(tx * ((p.Ho + 3) / 4), p.N), dim3(128), 0, s, p);
    else if (ty == 2) hipLaunchKernelGGL((stem_kernel<BF16, 2>), dim3(tx * ((p.Ho + 1) / 2), p.N), dim3(64), 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_stem(const StemParams& p, hipStream_t s) { return launch_stem_t<false>(p, s); }
hipError_t launch_stem_bf16(const StemParams& p, hipStream_t s) { return launch_stem_t<true>(p, s); }

// ---------------------------------------------------------------------------------------- depthwise
// one thread = PX consecutive output pixels (along x) x 4 channels; the 3 x (PX*stride + 2*dil) input
// columns are loaded once and reused across the PX outputs (stride 1) — fewer L1/L2 reads per output.
template <int PX>
__global__ void __launch_bounds__(256) dw_kernel(DwParams p) {
    const int cg = p.C >> 2;
    const int wgroups = (p.Wo + PX - 1) / PX;
    const int64_t total = (int64_t)p.N * p.Ho * wgroups * cg;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % cg);
    int64_t r = idx / cg;
    const int xg = (int)(r % wgroups);
    r /= wgroups;
    const int yo = (int)(r % p.Ho);
    const int n = (int)(r / p.Ho);
    const int c = c4 * 4;
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *(const f32x4*)(p.w + t * p.C + c);
    const f32x4 b = *(const f32x4*)(p.bias + c);
    f32x4 acc[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) acc[i] = b;
    const float* in = p.in + (int64_t)n * p.Hi * p.Wi * p.in_ld + c;
    const int x0 = xg * PX;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = yo * p.stride + (ky - 1) * p.dil;
        if (yi < 0 || yi >= p.Hi) continue;
        const float* row = in + (int64_t)yi * p.Wi * p.in_ld;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xi = (x0 + i) * p.stride + (kx - 1) * p.dil;
                if (xi < 0 || xi >= p.Wi) continue;
                const f32x4 v = *(const f32x4*)(row + (int64_t)xi * p.in_ld);
                acc[i] += v * w[ky * 3 + kx];
            }
        }
    }
    float* out = p.out + (((int64_t)n * p.Ho + yo) * p.Wo) * p.out_ld + c;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int xo = x0 + i;
        if (xo >= p.Wo) break;
        f32x4 v = acc[i];
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
        *(f32x4*)(out + (int64_t)xo * p.out_ld) = v;
    }
}

// LDS-tiled depthwise kernel for large maps (batch 32): the per-thread kernel above fetches every input vector 6-9 times
// through L1/L2 (2.2 TB/s of algorithmic traffic at batch 32 on the 512-channel layers: the L2 -> L1 path is the limit, not
// HBM).  Here a workgroup owns an 8 x 8 patch of output pixels x CC channels (CC = 64, or 32 for the first block): the input
// window ((7 s + 2 d + 1)^2 pixels x CC channels, 256-byte row pieces) is loaded ONCE with 16-byte loads into LDS — every input
// byte is fetched (10/8)^2 = 1.56 times at stride 1 — and each thread walks a column strip of the patch (4 channels, RPT rows)
// with a rolling 3 x 3 register window; the results leave as 16-byte stores, 256 contiguous bytes per pixel.  25-37 KB of LDS
// per workgroup: 4-6 workgroups per CU overlap each other's load / compute / store phases.
// Same arithmetic and summation order as dw_kernel (bias, then the nine taps row-major): bit-identical results.
template <int CC, int S, int D, int PH>
__global__ void __launch_bounds__(256) dw_tiled_kernel(DwParams p, int tiles_y, int tiles_x) {
    constexpr int PW = 8;
    constexpr int QN = CC / 4;                        // channel quads per chunk
    constexpr int GRP = 256 / (QN * PW);              // row groups of the patch handled in parallel
    constexpr int RPT = PH / GRP;                     // patch rows per thread
    constexpr int WR = (PH - 1) * S + 2 * D + 1, WC = (PW - 1) * S + 2 * D + 1;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.y, c0 = chunk * CC;
    const int per_img = tiles_y * tiles_x;
    const int n = blockIdx.x / per_img, r0 = blockIdx.x - n * per_img;
    const int ty0 = (r0 / tiles_x) * PH, tx0 = (r0 % tiles_x) * PW;
    const int y0 = ty0 * S - D, x0 = tx0 * S - D;     // input position of window (0, 0)
    // ---- stage the window: piece = (window pixel, 16-byte part); all loads of a thread are issued before its stores.
    // Buffer loads: wave-uniform descriptor, one 32-bit byte offset per lane; a piece outside the image gets offset 2^31
    // (>= num_records: the hardware returns zeros) — no exec-masked 64-bit-address loads
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4), 0x00020000);
    constexpr int NPIECE = WR * WC * QN, PPT = (NPIECE + 255) / 256;
    f32x4 pc[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = tid + u * 256;
        const int px = i / QN, part = i % QN;
        const int wy = px / WC, wx = px % WC;
        const int y = y0 + wy, x = x0 + wx;
        const bool ok = i < NPIECE && y >= 0 && y < p.Hi && x >= 0 && x < p.Wi;
        const unsigned off = ok ? (unsigned)((((n * p.Hi + y) * p.Wi + x) * p.in_ld + c0 + part * 4) * 4) : 0x80000000u;
        pc[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0));
    }
    const int cq = tid % QN, lx = (tid / QN) % PW, grp = tid / (QN * PW);
    const int c = c0 + cq * 4;
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *(const f32x4*)(p.w + t * p.C + c);
    const f32x4 b = *(const f32x4*)(p.bias + c);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = tid + u * 256;
        if (i < NPIECE) *(f32x4*)(dsm + (size_t)i * 4) = pc[u];      // piece order = [pixel][CC] row-major
    }
    __syncthreads();
    const float* win = dsm + cq * 4;
    const int xo = tx0 + lx;
    auto at = [&](int wy, int wx) -> f32x4 { return *(const f32x4*)(win + (size_t)(wy * WC + wx) * CC); };
    auto put = [&](f32x4 acc, int ly) {
        const int yo = ty0 + ly;
        if (yo < p.Ho && xo < p.Wo) {
            acc.x = apply_act(acc.x, p.act); acc.y = apply_act(acc.y, p.act); acc.z = apply_act(acc.z, p.act); acc.w = apply_act(acc.w, p.act);
            *(f32x4*)(p.out + (((int64_t)n * p.Ho + yo) * p.Wo + xo) * p.out_ld + c) = acc;
        }
    };
    if (S == 1 && D == 1) {                            // rolling 3 x 3 window down the column strip: 3 LDS reads per output instead of 9
        f32x4 r[3][3];
        const int ly0 = grp * RPT;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) r[k + 1][kx] = at(ly0 + k, lx + kx);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { r[0][kx] = r[1][kx]; r[1][kx] = r[2][kx]; r[2][kx] = at(ly0 + i + 2, lx + kx); }
            f32x4 acc = b;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += r[t / 3][t % 3] * w[t];
            put(acc, ly0 + i);
        }
    } else if (S == 1 && D == 2 && RPT >= 2 && RPT % 2 == 0) {      // dilation 2: two interleaved rolling windows (even / odd rows)
        const int ly0 = grp * RPT;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            f32x4 r[3][3];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) r[k + 1][kx] = at(ly0 + par + 2 * k, lx + 2 * kx);
#pragma unroll
            for (int i = par; i < RPT; i += 2) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { r[0][kx] = r[1][kx]; r[1][kx] = r[2][kx]; r[2][kx] = at(ly0 + i + 4, lx + 2 * kx); }
                f32x4 acc = b;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc += r[t / 3][t % 3] * w[t];
                put(acc, ly0 + i);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int ly = grp * RPT + i;
            f32x4 acc = b;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += at(ly * S + (t / 3) * D, lx * S + (t % 3) * D) * w[t];
            put(acc, ly);
        }
    }
}

template <int CC, int S, int D, int PH>
static hipError_t launch_dw_tiled_t(const DwParams& p, hipStream_t s) {
    constexpr int WR = (PH - 1) * S + 2 * D + 1, WC = 7 * S + 2 * D + 1;
    constexpr size_t lds = (size_t)WR * WC * CC * sizeof(float);
    const int tiles_y = (p.Ho + PH - 1) / PH, tiles_x = (p.Wo + 7) / 8;
    const int64_t tiles = (int64_t)p.N * tiles_y * tiles_x;
    if (tiles >= (1ll << 31) - 1) return hipErrorInvalidValue;
    if ((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4 >= (1ll << 31)) return hipErrorInvalidValue;      // 32-bit buffer offsets
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) { hipError_t e = attr.ensure((const void*)dw_tiled_kernel<CC, S, D, PH>, 96 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((dw_tiled_kernel<CC, S, D, PH>), dim3((unsigned)tiles, p.C / CC), dim3(256), lds, s, p, tiles_y, tiles_x);
    return hipGetLastError();
}

static hipError_t try_dw_tiled(const DwParams& p, hipStream_t s, bool* used) {
    *used = false;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dw_tiled == 0) return hipSuccess;                   // LWP_DW_TILED "0": the per-thread kernel everywhere (A/B)
    const int64_t pixels = (int64_t)p.N * p.Ho * p.Wo;
    const bool force = T.dw_tiled == 1;                       // "1": the tiled kernel at every size (tests)
    if (!force && pixels * p.C < (int64_t)8 * 1024 * 1024) return hipSuccess;       // small maps: the per-thread kernel fills the chip better
    if ((p.stride != 1 && p.stride != 2) || (p.dil != 1 && p.dil != 2) || (p.in_ld & 3) || (p.out_ld & 3)) return hipSuccess;
    if (p.stride == 2 && p.dil == 2) return hipSuccess;
    // LWP_DW_CC: channels per workgroup, "128" | "64" everywhere it applies (experiments)
    int cc = p.C % 64 == 0 ? 64 : (p.C % 32 == 0 ? 32 : 0);
    // 128 channels per workgroup (512-byte pieces of the 2-KB pixel rows) measured at batch 32: the 512-channel blocks 4.68 -> 4.90
    // TB/s and model.3 (128 channels, 92 x 164) 4.76 -> 4.91; 256 channels 5.18 -> 4.90, dilation 2 4.16 -> 3.96 and the small
    // 128-channel maps 4.68 -> 3.85 lose with it
    const bool want128 = p.stride == 1 && p.dil == 1 && p.C % 128 == 0 && (p.C >= 512 || (p.C == 128 && pixels * p.C >= (int64_t)32 * 1024 * 1024));
    if (T.dw_cc ? (T.dw_cc == 128 && p.C % 128 == 0 && p.stride == 1) : want128) cc = 128;
    if (!cc) return hipSuccess;
    if (p.C / cc > 65535 || pixels >= (1ll << 31)) return hipSuccess;
    if ((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4 >= (1ll << 31)) return hipSuccess;       // the tiled kernel addresses its input with 32-bit buffer offsets
    // 16-row patches pay on the first block only (32 channels: 128-byte pixel rows, 30k workgroups): 4.05 -> 5.09 TB/s; the
    // 46 x 82 layers lose with them (cpm.trunk 4.6 -> 3.7 TB/s)
    const int ph = T.dw_ph ? T.dw_ph : (cc == 32 ? 16 : 8);    // LWP_DW_PH: patch rows 8 | 16 (experiments)
    *used = true;
    LWP_VARIANT(p, "dw_tiled<cc=%d,s=%d,d=%d,ph=%d>", cc, p.stride, p.dil, ph == 16 && p.stride == 1 ? 16 : 8);
#define DT_CASE(CC_, S_, D_) if (cc == CC_ && p.stride == S_ && p.dil == D_) return ph == 16 && S_ == 1 ? launch_dw_tiled_t<CC_, S_, D_, 16>(p, s) : launch_dw_tiled_t<CC_, S_, D_, 8>(p, s);
    DT_CASE(64, 1, 1) DT_CASE(64, 1, 2) DT_CASE(64, 2, 1) DT_CASE(32, 1, 1) DT_CASE(32, 1, 2) DT_CASE(32, 2, 1) DT_CASE(128, 1, 1) DT_CASE(128, 1, 2)
#undef DT_CASE
    *used = false;
    return hipSuccess;
}

hipError_t launch_dw(const DwParams& p, hipStream_t s) {
    {
        bool used = false;
        hipError_t e = try_dw_tiled(p, s, &used);
        if (e != hipSuccess || used) return e;
    }
    const int cg = p.C >> 2;
    // few pixels per thread when the map is small (keep the chip full), more when it is large
    const int64_t pixels = (int64_t)p.N * p.Ho * p.Wo;
    LWP_VARIANT(p, "dw<px=%d>", pixels * cg >= (int64_t)256 * 256 * 16 ? 2 : 1);
    if (pixels * cg >= (int64_t)256 * 256 * 16) {
        const int64_t total = (int64_t)p.N * p.Ho * ((p.Wo + 1) / 2) * cg;
        hipLaunchKernelGGL(dw_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    } else {
        const int64_t total = pixels * cg;
        hipLaunchKernelGGL(dw_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- implicit GEMM
// out[m][n] = act(sum_{tap,c} in[pix(m)+off(tap)][c] * w[tap][n][c] + bias[n]) (+ res[m][n])
//   m: output pixel (N*H*W, stride-1 convs only), n: output channel, K = taps * cin_pad.
// Workgroup tile BM x BN; (BM/32) x (BN/32) waves each own a 32x32 accumulator (v_mfma_f32_32x32x2_f32),
// replicated KS times: K-slice group g walks K steps g, g+KS, ... with its own double-buffered LDS
// tiles, and the KS partial accumulators are summed through LDS in a fixed order at the end
// (intra-workgroup split-K: at batch 1 the M x N tile count alone cannot fill 1024 SIMDs).
// K is walked in steps of 32 channels inside one tap.  LDS rows are [row][32 + 4 pad] floats: a lane
// (row r = lane&31, half h = lane>>5) reads 4 consecutive k with one ds_read_b128 at k = 8s + 4h; MFMA
// (s,t) then contracts k in {8s+t, 8s+4+t} on the A and the B side alike (any k order is a valid GEMM).
constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

// KSZ = spatial kernel size (1 or 3), a template parameter so that the tap arithmetic folds at compile time and the
// 1x1 and dense-3x3 launches carry different kernel names in rocprof traces.
template <int BM, int BN, int KS, int KSZ>
__global__ void __launch_bounds__((BM / 32) * (BN / 32) * KS * 64) gemm_kernel(GemmParams p) {
    constexpr int WM = BM / 32, WN = BN / 32;
    constexpr int GW = WM * WN;                     // waves per K-slice group
    constexpr int GT = GW * 64;                     // threads per group
    constexpr int A_CHUNKS = BM * (BK / 4);         // 16-byte chunks per A tile
    constexpr int B_CHUNKS = BN * (BK / 4);
    constexpr int A_PER = A_CHUNKS / GT;
    constexpr int B_PER = B_CHUNKS / GT;
    static_assert(A_CHUNKS % GT == 0 && B_CHUNKS % GT == 0, "tile/threads mismatch");
    constexpr int TILE_FLOATS = 2 * (BM + BN) * LDS_LD;   // one group's double-buffered A+B tiles

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int g = tid / GT;                         // K-slice group (wave-uniform)
    const int t = tid - g * GT;
    float* As = smem + g * TILE_FLOATS;             // [2][BM][LDS_LD]
    float* Bs = As + 2 * BM * LDS_LD;               // [2][BN][LDS_LD]

    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int ntn = p.cout_pad / BN;
    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    // run of logical tiles so the BN-neighbours that re-read one A tile hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int tile_m = bid / ntn, tile_n = bid % ntn;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // per-thread staging assignment: chunk id -> (row, 16-B column)
    // both operand streams are buffer loads (wave-uniform descriptor, one 32-bit byte offset per lane; a tap outside the image
    // reads offset 2^31 >= num_records and comes back as zeros)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)(M * p.in_ld * 4), 0x00020000);   // host: < 2^31 bytes
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
    int a_row[A_PER], a_col[A_PER], a_y[A_PER], a_x[A_PER];
    unsigned a_base[A_PER];
    bool a_ok[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int ch = t + i * GT;
        a_row[i] = ch / (BK / 4);
        a_col[i] = (ch % (BK / 4)) * 4;
        const int64_t m = m0 + a_row[i];
        a_ok[i] = m < M;
        const int64_t mm = a_ok[i] ? m : 0;
        a_x[i] = (int)(mm % p.W);
        a_y[i] = (int)((mm / p.W) % p.H);
        a_base[i] = (unsigned)(mm * p.in_ld + a_col[i]) * 4u;        // bytes
    }
    int b_off[B_PER], b_lds[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = t + i * GT;
        const int row = ch / (BK / 4), col = (ch % (BK / 4)) * 4;
        b_off[i] = (row * p.cin_pad + col) * 4;                      // bytes
        b_lds[i] = row * LDS_LD + col;
    }

    const int ksteps_per_tap = p.cin_pad / BK;
    constexpr int taps = KSZ * KSZ;
    const int nsteps = taps * ksteps_per_tap;

    // register staging, two sets: the loads of K step i+2 are issued while step i is multiplied and step i+1 (loaded
    // one iteration earlier) is written to the other LDS buffer — an L2 round trip under load is longer than one
    // step's MFMA time, so a single step of look-ahead left every iteration waiting on vmcnt.
    f32x4 a_reg[2][A_PER], b_reg[2][B_PER];
    auto load_step = [&](int step, f32x4* ar, f32x4* br) {
        const int tap = step / ksteps_per_tap;
        const int c0 = (step - tap * ksteps_per_tap) * BK;
        int dy = 0, dx = 0;
        if (KSZ == 3) { dy = (tap / 3 - 1) * p.dil; dx = (tap % 3 - 1) * p.dil; }
        const int shift = ((dy * p.W + dx) * p.in_ld + c0) * 4;         // bytes, may be negative
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            // the select is on the OFFSET, so the loaded value goes to LDS untouched and its s_waitcnt lands after the MFMA block
            ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? a_base[i] + (unsigned)shift : 0x80000000u, 0, 0));
        }
        const unsigned woff = (unsigned)__builtin_amdgcn_readfirstlane(((tap * p.cout_pad + n0) * p.cin_pad + c0) * 4);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) br[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], woff, 0));
    };
    auto store_step = [&](int buf, const f32x4* ar, const f32x4* br) {
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) *(f32x4*)(a + a_row[i] * LDS_LD + a_col[i]) = ar[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *(f32x4*)(b + b_lds[i]) = br[i];
    };

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int iters = (nsteps + KS - 1) / KS;
    if (g < nsteps) { load_step(g, a_reg[0], b_reg[0]); store_step(0, a_reg[0], b_reg[0]); }
    if (g + KS < nsteps) load_step(g + KS, a_reg[1], b_reg[1]);
    __syncthreads();
    // one half-iteration; P (compile-time) is the LDS buffer being multiplied = the register set being refilled
    auto half = [&](int it, auto P_) {
        constexpr int P = decltype(P_)::value;
        const int cur = g + it * KS, n1 = cur + KS, n2 = cur + 2 * KS;
        if (n2 < nsteps) load_step(n2, a_reg[P], b_reg[P]);
        if (cur < nsteps) {
            const float* a = As + P * BM * LDS_LD + (wm * 32 + r) * LDS_LD + 4 * h;
            const float* b = Bs + P * BN * LDS_LD + (wn * 32 + r) * LDS_LD + 4 * h;
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                const f32x4 av = *(const f32x4*)(a + 8 * s);
                const f32x4 bv = *(const f32x4*)(b + 8 * s);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        if (n1 < nsteps) store_step(P ^ 1, a_reg[P ^ 1], b_reg[P ^ 1]);
        __syncthreads();
    };
    for (int it = 0; it < iters; it += 2) {
        half(it, std::integral_constant<int, 0>{});
        half(it + 1, std::integral_constant<int, 1>{});
    }

    if (KS > 1) {   // fixed-order reduction of the K-slice partials through LDS (tiles are dead after the last barrier)
        float* red = smem;                          // [KS-1][GW][16][64]
        if (g > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((g - 1) * GW + wave) * 16 + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (g > 0) return;
#pragma unroll
        for (int gg = 1; gg < KS; ++gg)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += red[(((gg - 1) * GW + wave) * 16 + i) * 64 + lane];
    }

    // epilogue: lane holds column n = n0 + wn*32 + r, rows (i&3) + 8*(i>>2) + 4*h of the wave tile
    const int n = n0 + wn * 32 + r;
    if (n < p.cout) {
        const float bias = p.bias[n];
        const int64_t HW = (int64_t)p.H * p.W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t m = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (m < M) {
                float v = apply_act(acc[i] + bias, p.act);
                if (p.res) v += p.res[m * p.res_ld + n];
                p.out[m * p.out_ld + n] = v;
                if (p.out_nchw || p.out_nchw2) {         // stage outputs (NCHW); merged heads split at out_split
                    const int64_t img = m / HW, pix = m - img * HW;
                    const int c0 = p.out_split > 0 ? p.out_split : p.cout;
                    if (n < c0) { if (p.out_nchw) p.out_nchw[(img * c0 + n) * HW + pix] = v; }
                    else if (p.out_nchw2) p.out_nchw2[(img * (p.cout - c0) + (n - c0)) * HW + pix] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int KS, int KSZ>
static hipError_t launch_gemm_k(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int64_t tiles = ((M + BM - 1) / BM) * (p.cout_pad / BN);
    if (M * p.in_ld * 4 >= (1ll << 31)) return hipErrorInvalidValue;         // 32-bit buffer offsets (2 GiB of activations per launch)
    constexpr int NT = (BM / 32) * (BN / 32) * KS * 64;
    size_t lds = (size_t)KS * 2 * (BM + BN) * LDS_LD * sizeof(float);
    const size_t red = (size_t)(KS - 1) * (BM / 32) * (BN / 32) * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    static LdsAttrOnce attr;
    if (lds > 64 * 1024) {
        hipError_t e = attr.ensure((const void*)gemm_kernel<BM, BN, KS, KSZ>, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, KS, KSZ>), dim3((unsigned)tiles), dim3(NT), lds, s, p);
    return hipGetLastError();
}
template <int BM, int BN, int KS>
static hipError_t launch_gemm_t(const GemmParams& p, hipStream_t s) {
    if (p.ks == 3) return launch_gemm_k<BM, BN, KS, 3>(p, s);
    if (p.ks == 1) return launch_gemm_k<BM, BN, KS, 1>(p, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------- implicit GEMM, wave-private tiles
// The 32-row configurations used at small M (batch 1: M = 3772 pixels cannot fill the chip with 64-row tiles).
// Same mathematics and the same fragment layout as gemm_kernel, but every wave stages ITS OWN A tile (32 pixels x 32 k)
// in a private LDS slice: DS instructions of one wave execute in order, so the K loop needs no barrier at all (one
// buffer suffices: the writes of step i+1 are issued after the reads of step i).  B never touches LDS: the weights
// are packed on the host a second time in fragment order, so a K step of a wave is four fully coalesced 1-KiB loads
// straight into the registers the MFMAs read (double-buffered).  With the shared tiles every iteration ended in a
// workgroup barrier with all eight waves idle at once (in-kernel cycle stamps: 3400 cycles per K step against 2048
// of MFMA issue; 2800 here).  Ablation at batch 1 (dense 3x3 128->128, ~19 us per launch): MFMA-only K loop 8.2 us,
// memory-only 4.9 us (~21 TB/s of L2 -> CU traffic chip-wide), both 13.4 us, fixed prologue/epilogue/launch 6.1 us.
// Variants that did NOT help and were dropped: two-deep register prefetch, a three-stage LDS/fragment pipeline, a
// phase skew between the two waves of a SIMD, A loaded straight into fragment registers (32-byte pieces per lane:
// 21 us).  B through a private LDS tile as well: 19.3 us, this form 18.9 us.
// A is loaded by both N-halves of a K group (the second read hits L1/L2); LDS per workgroup: waves x 4.5 KiB.
template <int BN, int KS, int KSZ>
__global__ void __launch_bounds__((BN / 32) * KS * 64) gemm_wp_kernel(GemmParams p) {
    constexpr int WN = BN / 32;
    constexpr int WAVE_FLOATS = 32 * LDS_LD;                // the wave's private A tile (B never touches LDS)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / WN, wn = wave % WN;                // K-slice group, N half
    const int r = lane & 31, h = lane >> 5;
    float* Aw = smem + wave * WAVE_FLOATS;

    const int M = p.N * p.H * p.W;                          // host guarantees < 2^31
    const int ntn = p.cout_pad / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int tile_m = bid / ntn, tile_n = bid % ntn;
    const int m0 = tile_m * 32;
    const int n0 = tile_n * BN + wn * 32;                   // first output channel of this wave
    const int n = n0 + r;
    const float bias = n < p.cout ? p.bias[n] : 0.f;        // in flight during the K loop

    // A staging: 16-byte chunk i of a lane = (row (lane >> 3) + 8 i, columns 4 (lane & 7)..+3) of the 32 x 32 tile
    const int col = (lane & 7) * 4;
    int a_y[4], a_x[4], lds_off[4];
    int64_t a_base[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i;
        const int m = m0 + row;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int q = mm / p.W;
        a_x[i] = mm - q * p.W;
        a_y[i] = q % p.H;
        a_base[i] = (int64_t)mm * p.in_ld + col;
        lds_off[i] = row * LDS_LD + col;
    }
    const int ksteps_per_tap = p.cin_pad / BK;
    constexpr int taps = KSZ * KSZ;
    const int nsteps = taps * ksteps_per_tap;
    // B: the weights in fragment order, [step][32-channel tile][s][lane][4] -> one step of this wave = 4 KiB contiguous
    // buffer loads (wave-uniform descriptor and step offset, one 32-bit offset register per lane): see gemm_ar_kernel
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, 0x7fffffff, 0x00020000);
    const unsigned b_tile = (unsigned)__builtin_amdgcn_readfirstlane(n0 >> 5) * 4096u;       // bytes
    const unsigned b_step = (unsigned)(p.cout_pad >> 5) * 4096u;

    f32x4 a_reg[4], b_nxt[BK / 8];
    auto load_step = [&](int step) {
        const int tap = step / ksteps_per_tap;
        const int c0 = (step - tap * ksteps_per_tap) * BK;
        int dy = 0, dx = 0;
        if (KSZ == 3) { dy = (tap / 3 - 1) * p.dil; dx = (tap % 3 - 1) * p.dil; }
        const int64_t shift = ((int64_t)dy * p.W + dx) * p.in_ld + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const float* src = ok ? p.in + a_base[i] + shift : p.zeros;       // out-of-image taps read the zero page
            a_reg[i] = *(const f32x4*)src;
        }
        const unsigned soff = (unsigned)step * b_step + b_tile;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) b_nxt[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + s * 1024, soff, 0));
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(Aw + lds_off[i]) = a_reg[i];
    };

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int iters = (nsteps + KS - 1) / KS;
    f32x4 bv[BK / 8];
    if (g < nsteps) {
        load_step(g);
        store_step();
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) bv[s] = b_nxt[s];
    }
    const float* a = Aw + r * LDS_LD + 4 * h;
    for (int it = 0; it < iters; ++it) {
        const int cur = g + it * KS, nxt = cur + KS;
        if (nxt < nsteps) load_step(nxt);
        if (cur < nsteps) {
            f32x4 av[BK / 8];
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) av[s] = *(const f32x4*)(a + 8 * s);
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].x, bv[s].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].y, bv[s].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].z, bv[s].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].w, bv[s].w, acc, 0, 0, 0);
            }
        }
        if (nxt < nsteps) {
            store_step();                        // after this step's reads in program (= LDS execution) order
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) bv[s] = b_nxt[s];
        }
    }

    // residual operand of the K-group-0 waves: issued now, consumed after the reduction
    float resv[16];
    const bool has_res = p.res != nullptr && g == 0 && n < p.cout;
    if (has_res) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            resv[i] = m < M ? p.res[(int64_t)m * p.res_ld + n] : 0.f;
        }
    }
    if (KS > 1) {   // fixed-order reduction of the K-slice partials through LDS
        __syncthreads();                            // every wave is done with its tiles: the slices are reused
        float* red = smem;                          // [KS-1][WN][16][64]
        if (g > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((g - 1) * WN + wn) * 16 + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (g > 0) return;
#pragma unroll 1
        for (int gg = 1; gg < KS; ++gg)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += red[(((gg - 1) * WN + wn) * 16 + i) * 64 + lane];
    }

    if (n < p.cout) {
        const int HW = p.H * p.W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (m < M) {
                float v = apply_act(acc[i] + bias, p.act);
                if (has_res) v += resv[i];
                p.out[(int64_t)m * p.out_ld + n] = v;
                if (p.out_nchw || p.out_nchw2) {         // stage outputs (NCHW); merged heads split at out_split
                    const int img = m / HW, pix = m - img * HW;
                    const int c0 = p.out_split > 0 ? p.out_split : p.cout;
                    if (n < c0) { if (p.out_nchw) p.out_nchw[((int64_t)img * c0 + n) * HW + pix] = v; }
                    else if (p.out_nchw2) p.out_nchw2[((int64_t)img * (p.cout - c0) + (n - c0)) * HW + pix] = v;
                }
            }
        }
    }
}

template <int BN, int KS, int KSZ>
static hipError_t launch_gemm_wp_k(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    if (M >= (1ll << 31) - 64) return hipErrorInvalidValue;
    const int64_t tiles = ((M + 31) / 32) * (p.cout_pad / BN);
    constexpr int NWV = (BN / 32) * KS;
    size_t lds = (size_t)NWV * 32 * LDS_LD * sizeof(float);
    const size_t red = (size_t)(KS - 1) * (BN / 32) * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (!p.wf) return hipErrorInvalidValue;
    static LdsAttrOnce attr;
    if (lds > 64 * 1024) {
        hipError_t e = attr.ensure((const void*)gemm_wp_kernel<BN, KS, KSZ>, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_wp_kernel<BN, KS, KSZ>), dim3((unsigned)tiles), dim3(NWV * 64), lds, s, p);
    return hipGetLastError();
}
template <int BN, int KS>
static hipError_t launch_gemm_wp_t(const GemmParams& p, hipStream_t s) {
    if (p.ks == 3) return launch_gemm_wp_k<BN, KS, 3>(p, s);
    if (p.ks == 1) return launch_gemm_wp_k<BN, KS, 1>(p, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------- implicit GEMM, A resident in LDS
// 32-row tiles, third form.  K is split over the KS wave groups by 32-CHANNEL BLOCKS (group g owns blocks g, g+KS, ...),
// and a group walks all taps of a block.  For a dense 3x3 the nine A tiles of one channel block are the same
// (32 + 2 dil) x 3 pixel window at nine offsets, so the group stages that window ONCE (3 segments of 32 + 2 dil
// consecutive pixels, 2.75x fewer loads and LDS stores than nine tiles), one barrier, and the K loop is: four
// ds_read_b128 at a tap-dependent row offset (taps outside the image: the lane reads a zero row instead), four
// 1-KiB loads of fragment-packed weights (double-buffered in registers), sixteen MFMAs — no LDS store, no barrier, no
// global A load inside the loop.  1x1 convs stage their (up to four) 32 x 32 tiles the same way.
#ifndef AR_PF
#define AR_PF 2
#endif
template <int BN, int KS, int KSZ>
__global__ void __launch_bounds__((BN / 32) * KS * 64) gemm_ar_kernel(GemmParams p) {
    constexpr int WN = BN / 32;
    constexpr int GT = WN * 64;                             // threads per K group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const int M = p.N * p.H * p.W;                          // host guarantees < 2^31
    const int ntn = p.cout_pad / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int tile_m = bid / ntn, tile_n = bid % ntn;
    const int m0 = tile_m * 32;
    const int n0 = tile_n * BN + wn * 32;
    const int n = n0 + r;
    const float bias = n < p.cout ? p.bias[n] : 0.f;

    const int kpt = p.cin_pad / BK;                         // channel blocks
    const int nb = g < kpt ? (kpt - g + KS - 1) / KS : 0;   // blocks of this group
    const int nb_max = (kpt + KS - 1) / KS;
    const int seg = KSZ == 3 ? 32 + 2 * p.dil : 32;         // rows per segment
    const int R = KSZ == 3 ? 3 * seg : 32;                  // rows per block
    float* Ag = smem + (size_t)g * (nb_max * R + 1) * LDS_LD;   // [nb][R][36] + one zero row
    float* zrow = Ag + (size_t)nb_max * R * LDS_LD;

    // ---- stage the group's A data: chunk = (block j, row, 16-byte column)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)M * p.in_ld * 4), 0x00020000);   // host: < 2^31 bytes
    {
        const int t = tid - g * GT;                         // thread within the group
        const int col = (t & 7) * 4;
        const int total = nb * R;
        for (int base = 0; base < total; base += 8 * (GT / 8)) {      // 8 independent loads in flight per thread, then the stores
            f32x4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = base + (t >> 3) + u * (GT / 8);
                const int chc = ch < total ? ch : 0;
                const int j = chc / R, row = chc - j * R;
                int64_t idx;
                if (KSZ == 3) {
                    const int sd = row / seg, rw = row - sd * seg;
                    idx = (int64_t)m0 + (int64_t)(sd - 1) * p.dil * p.W - p.dil + rw;
                } else {
                    idx = (int64_t)m0 + row;
                }
                const bool ok = ch < total && idx >= 0 && idx < M;
                // buffer load: rows outside the tensor read offset 2^31 >= num_records and come back as zeros
                tmp[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? (unsigned)(((int)idx * p.in_ld + (g + j * KS) * BK + col) * 4) : 0x80000000u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = base + (t >> 3) + u * (GT / 8);
                if (ch < total) *(f32x4*)(Ag + (size_t)ch * LDS_LD + col) = tmp[u];
            }
        }
        if (t < 9) *(f32x4*)(zrow + 4 * t) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // the lane's pixel and the taps that fall inside the image
    const int mrow = m0 + r;
    const bool row_ok = mrow < M;
    const int mm = row_ok ? mrow : 0;
    const int qy = mm / p.W;
    const int x = mm - qy * p.W, y = qy % p.H;
    unsigned valid = 0;
#pragma unroll
    for (int t = 0; t < KSZ * KSZ; ++t) {
        const int yy = y + (KSZ == 3 ? (t / 3 - 1) * p.dil : 0), xx = x + (KSZ == 3 ? (t % 3 - 1) * p.dil : 0);
        if (row_ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) valid |= 1u << t;
    }
    constexpr int taps = KSZ * KSZ;
    const int my_steps = nb * taps;
    // B: fragment order [step = tap * kpt + block][32-channel tile][s][lane][4]
    // buffer loads: descriptor + step offset wave-uniform (SGPRs), one 32-bit offset register per lane — a 64-bit per-lane
    // address (global_load) costs ~3x the issue time while the matrix pipes are busy (measured on the fused dw+pw kernel)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, 0x7fffffff, 0x00020000);
    const unsigned b_tile = (unsigned)__builtin_amdgcn_readfirstlane(n0 >> 5) * 4096u;       // bytes
    const unsigned b_step = (unsigned)(p.cout_pad >> 5) * 4096u;
    auto load_b = [&](int i, f32x4* dst) {                  // i-th step of this group: block j = i / taps, tap = i % taps
        const int j = i / taps, tap = i - j * taps;
        const unsigned soff = (unsigned)(tap * kpt + g + j * KS) * b_step + b_tile;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) dst[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + s * 1024, soff, 0));
    };
    constexpr int PF = AR_PF;                               // weight steps in flight per wave (register ring, static indices)
    f32x4 bv[PF][BK / 8];
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) load_b(u < my_steps ? u : (my_steps > 0 ? my_steps - 1 : 0), bv[u]);
    __syncthreads();                                        // the windows of every group are complete

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* a_lane = Ag + (size_t)r * LDS_LD + 4 * h;
    const float* z_lane = zrow + 4 * h;
    auto one = [&](int i, auto P_) {
        constexpr int P = decltype(P_)::value;
        const int nxt = i + PF - 1 < my_steps ? i + PF - 1 : my_steps - 1;   // past the end: the last step again (no branch)
        load_b(nxt, bv[(P + PF - 1) % PF]);
        const int j = i / taps, tap = i - j * taps;
        int roff = j * R;
        if (KSZ == 3) roff += (tap / 3) * seg + (tap % 3) * p.dil;     // segment dy, shift dx (the segment starts at -dil)
        const float* a = (valid >> tap) & 1u ? a_lane + (size_t)roff * LDS_LD : z_lane;
        f32x4 av[BK / 8];
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) av[s] = *(const f32x4*)(a + 8 * s);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].x, bv[P][s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].y, bv[P][s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].z, bv[P][s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].w, bv[P][s].w, acc, 0, 0, 0);
        }
    };
    int i = 0;
    for (; i + PF <= my_steps; i += PF) {
        one(i, std::integral_constant<int, 0>{});
        if (PF > 1) one(i + 1, std::integral_constant<int, 1 % PF>{});
        if (PF > 2) one(i + 2, std::integral_constant<int, 2 % PF>{});
        if (PF > 3) one(i + 3, std::integral_constant<int, 3 % PF>{});
    }
    if (i < my_steps) { one(i, std::integral_constant<int, 0>{}); ++i; }
    if (PF > 2 && i < my_steps) { one(i, std::integral_constant<int, 1 % PF>{}); ++i; }
    if (PF > 3 && i < my_steps) { one(i, std::integral_constant<int, 2 % PF>{}); ++i; }

    // ---- reduction + epilogue, spread over ALL waves: every K group takes 16 / KS of the 16 accumulator registers
    // (= 8 / KS... rows) of its N half, sums the KS partials in the fixed order 0, 1, ..., KS-1 and stores those rows.
    constexpr int EPG = 16 / KS;                    // accumulator registers per group in the epilogue
    const int e0 = g * EPG;
    float resv[EPG];
    const bool has_res = p.res != nullptr && n < p.cout;
    if (has_res) {                                  // requested now, consumed after the reduction
#pragma unroll
        for (int u = 0; u < EPG; ++u) {
            const int e = e0 + u;
            const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            resv[u] = m < M ? p.res[(int64_t)m * p.res_ld + n] : 0.f;
        }
    }
    float outv[EPG];
    if (KS > 1) {
        __syncthreads();                            // every wave is done with the windows: the space is reused
        float* red = smem;                          // [KS][WN][16][64]
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((g * WN + wn) * 16 + e) * 64 + lane] = acc[e];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPG; ++u) {
            float v = red[((0 * WN + wn) * 16 + e0 + u) * 64 + lane];
#pragma unroll
            for (int gg = 1; gg < KS; ++gg) v += red[((gg * WN + wn) * 16 + e0 + u) * 64 + lane];
            outv[u] = v;
        }
    } else {
#pragma unroll
        for (int u = 0; u < EPG; ++u) outv[u] = acc[u];
    }

    if (n < p.cout) {
        const int HW = p.H * p.W;
#pragma unroll
        for (int u = 0; u < EPG; ++u) {
            const int e = e0 + u;
            const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M) {
                float v = apply_act(outv[u] + bias, p.act);
                if (has_res) v += resv[u];
                p.out[(int64_t)m * p.out_ld + n] = v;
                if (p.out_nchw || p.out_nchw2) {         // stage outputs (NCHW); merged heads split at out_split
                    const int img = m / HW, pix = m - img * HW;
                    const int c0 = p.out_split > 0 ? p.out_split : p.cout;
                    if (n < c0) { if (p.out_nchw) p.out_nchw[((int64_t)img * c0 + n) * HW + pix] = v; }
                    else if (p.out_nchw2) p.out_nchw2[((int64_t)img * (p.cout - c0) + (n - c0)) * HW + pix] = v;
                }
            }
        }
    }
}

template <int BN, int KS, int KSZ>
static hipError_t launch_gemm_ar_k(const GemmParams& p, hipStream_t s, bool* fits) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    *fits = false;
    if (M >= (1ll << 31) - 64 || !p.wf) return hipSuccess;
    if (M * p.in_ld * 4 >= (1ll << 31)) return hipSuccess;       // 32-bit buffer offsets: larger inputs take the shared-tile kernel's path (and its check)
    const int kpt = p.cin_pad / BK, nb_max = (kpt + KS - 1) / KS;
    const int R = KSZ == 3 ? 3 * (32 + 2 * p.dil) : 32;
    size_t lds = (size_t)KS * ((size_t)nb_max * R + 1) * LDS_LD * sizeof(float);
    const size_t red = (size_t)KS * (BN / 32) * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 150 * 1024) return hipSuccess;                 // caller falls back to the per-step staging kernel
    *fits = true;
    constexpr int NWV = (BN / 32) * KS;
    const int64_t tiles = ((M + 31) / 32) * (p.cout_pad / BN);
    static LdsAttrOnce attr;
    {
        hipError_t e = attr.ensure((const void*)gemm_ar_kernel<BN, KS, KSZ>, 150 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_ar_kernel<BN, KS, KSZ>), dim3((unsigned)tiles), dim3(NWV * 64), lds, s, p);
    return hipGetLastError();
}
template <int BN, int KS>
static hipError_t launch_gemm_ar_t(const GemmParams& p, hipStream_t s, bool* fits) {
    if (p.ks == 3) return launch_gemm_ar_k<BN, KS, 3>(p, s, fits);
    if (p.ks == 1) return launch_gemm_ar_k<BN, KS, 1>(p, s, fits);
    *fits = false;
    return hipSuccess;
}

struct GemmCfg { int bm, bn, ks; };

static hipError_t dispatch_gemm(const GemmParams& p, hipStream_t s, GemmCfg c) {
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    const bool wp_set = T.gemm_wp >= 0;                     // LWP_GEMM_WP "0": shared-tile kernel for the 32-row configurations too (A/B)
    if (c.bm == 32 && !wp_set) {                            // A-resident form when the group windows fit in LDS
        bool fits = false;
        hipError_t e = hipSuccess;
#define AR_CASE(BN_, KS_) if (c.bn == BN_ && c.ks == KS_) e = launch_gemm_ar_t<BN_, KS_>(p, s, &fits);
        AR_CASE(64, 1) AR_CASE(64, 2) AR_CASE(64, 4) AR_CASE(32, 4) AR_CASE(32, 8)
#undef AR_CASE
        if (fits) LWP_VARIANT(p, "gemm_ar<%d,%d,%d>", c.bn, c.ks, p.ks);
        if (e != hipSuccess || fits) return e;
    }
    if (c.bm == 32 && T.gemm_wp != 0) {
        LWP_VARIANT(p, "gemm_wp<%d,%d,%d>", c.bn, c.ks, p.ks);
#define WP_CASE(BN_, KS_) if (c.bn == BN_ && c.ks == KS_) return launch_gemm_wp_t<BN_, KS_>(p, s);
        WP_CASE(64, 1) WP_CASE(64, 2) WP_CASE(64, 4) WP_CASE(64, 8) WP_CASE(32, 4) WP_CASE(32, 8)
#undef WP_CASE
    }
    LWP_VARIANT(p, "gemm<%d,%d,%d,%d>", c.bm, c.bn, c.ks, p.ks);
#define GEMM_CASE(BM_, BN_, KS_) \
    if (c.bm == BM_ && c.bn == BN_ && c.ks == KS_) return launch_gemm_t<BM_, BN_, KS_>(p, s);
    GEMM_CASE(32, 64, 1) GEMM_CASE(32, 64, 2) GEMM_CASE(32, 64, 4)
    GEMM_CASE(64, 64, 1) GEMM_CASE(64, 64, 2) GEMM_CASE(64, 64, 4)
    GEMM_CASE(64, 128, 1) GEMM_CASE(64, 128, 2)
    GEMM_CASE(128, 128, 1)
    GEMM_CASE(32, 32, 4) GEMM_CASE(32, 32, 8)
#undef GEMM_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int nsteps = p.ks * p.ks * (p.cin_pad / BK);
    GemmCfg c{64, 64, 1};
    // experiments: LWP_GEMM_C3 / LWP_GEMM_PW = "BM,BN,KS" override the heuristic for dense-3x3 / 1x1 layers
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (p.ks == 3 ? T.has_c3 : T.has_pw) {
        const int* v = p.ks == 3 ? T.c3 : T.pw;
        const GemmCfg o{v[0], v[1], v[2]};
        if (o.bn > 0 && (p.cout_pad % o.bn) == 0) return dispatch_gemm(p, s, o);
    }
    // heuristic (from tools/gemm_sweep.py on the real layer shapes): with >= ~1.5 waves of 64x64 tiles per CU use
    // them (split K in two while the grid is still short); otherwise 32-row tiles and split K four ways inside
    // the workgroup so that every SIMD of every CU holds waves; narrow heads (N padded to 64) with a long K go
    // to 32x32 tiles with an 8-way split.
    const int64_t t64 = ((M + 63) / 64) * (p.cout_pad / 64);
    const int64_t t32 = ((M + 31) / 32) * (p.cout_pad / 64);
    const int64_t t128 = (p.cout_pad % 128 == 0) ? ((M + 127) / 128) * (p.cout_pad / 128) : 0;
    // 128 x 128 tiles (16 waves x (32 x 32): half the barriers and LDS bytes per FLOP of 64 x 64) when their count fills
    // whole rounds of the chip's 512 resident workgroups (2 per CU): 472 tiles (batch 16) gain 6 %, 529 tiles lose 8 %
    const int64_t rounds128 = (t128 + 511) / 512;
    if (t128 >= 400 && (t128 >= 2048 || t128 * 5 >= rounds128 * 512 * 4)) {
        c = GemmCfg{128, 128, 1};
    } else if (t64 >= 400) {
        c = GemmCfg{64, 64, (t64 < 1024 && nsteps >= 8) ? 2 : 1};
    } else if (p.cout_pad == 64 && nsteps >= 16) {
        c = GemmCfg{32, 32, 8};
    } else {
        c = GemmCfg{32, 64, nsteps >= 4 ? 4 : (nsteps >= 2 ? 2 : 1)};
        if (t32 >= 1024 && c.ks > 2) c.ks = 2;
    }
    return dispatch_gemm(p, s, c);
}

// ---------------------------------------------------------------------------------------- fused depthwise -> pointwise
// conv_dw / conv_dw_no_bn blocks (modules/conv.py:13-32) as ONE kernel: the depthwise 3x3 (+BN+ReLU | +ELU) output
// never goes to memory and one launch replaces two.
//   workgroup = BM output pixels x ALL output channels: NW waves, wave w owns channels [32w, 32w+32) as
//   (BM/16) x 2 accumulators of v_mfma_f32_16x16x4_f32 -> the depthwise result is computed exactly once per pixel.
//   Phase 1: all NW*64 threads compute the workgroup's whole depthwise row block [BM][C] into LDS (one round of
//            coalesced 16-byte loads: consecutive lanes on consecutive channels), one barrier.
//   Phase 2: barrier-free GEMM over K = C from the resident tile: a lane (row r = lane&15, q = lane>>4) reads
//            k = 16u+4q..+3 with one ds_read_b128; MFMA (u,j) contracts k in {16u+4q+j}.  B (pointwise weights)
//            never touches LDS: it is packed on the host in fragment order [k-step][wave][4][lane][4 floats], so each
//            wave streams its own 4 KiB per 32-deep step from L2 with four fully coalesced 1-KiB loads, prefetched
//            one step ahead in registers.  With 4 waves per SIMD and no barrier the matrix pipes stay fed.
// Row padding of the f32 depthwise tile [rows][C + 4].  The K loop's operand read (one ds_read_b128 per lane at row = lane & 15,
// 16 bytes x (lane >> 4)) is two-way bank-conflicted with this stride (16 B x odd) and conflict-free with C + 8 — measured: the
// f32 blocks do not care (the f32 MFMA leaves the LDS idle most of the time): 26.9 against 26.2 us at batch 1 with C + 8, the
// same at batch 32.  The bf16 kernels take the conflict-free stride (DWPW_APAD, net_kernels_bf16.hip).
constexpr int DWPW_F32_APAD = 4;
template <int BM, int NW, int DBG = 0, int ACT = -1>
__global__ void __launch_bounds__(NW * 64) dwpw_kernel(DwPwParams p) {
    // ACT >= 0: both activations known at compile time (ReLU for every conv_dw block): no chain of scalar branches per vector
    const int act_dw = ACT >= 0 ? ACT : p.act_dw, act_pw = ACT >= 0 ? ACT : p.act_pw;
    constexpr int NT = NW * 64;
    constexpr int RT = BM / 16;                      // row tiles per wave
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int ldA = p.C + DWPW_F32_APAD;             // padded row stride (floats)
    float* At = dsm;                                 // [BM][C + DWPW_F32_APAD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.y * NW + (tid >> 6);   // global column-wave index (blockIdx.y: column split)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.pw_w, 0, 0x7fffffff, 0x00020000);
    const int nwt = gridDim.y * NW;                  // column waves in total = cout / 32
    const int r16 = lane & 15, q = lane >> 4;
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs): give every XCD a contiguous run of row
    // blocks, so that the 3-row input windows of neighbouring blocks are fetched into ONE L2 instead of all eight
    int bid = blockIdx.x;
    if (gridDim.y == 1 && !(DBG & 8)) {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    const int64_t m0 = (int64_t)bid * BM;
    const int nsteps = p.C / 32;

    // weight fragments: PF statically rotated register buffers (the K loop is unrolled by PF, no register copies), so
    // PF-1 steps of the weight stream stay in flight per wave (bandwidth-delay: ~250 KB must be in flight per CU)
#ifndef DWPW_PF
#define DWPW_PF 2
#endif
    constexpr int PF = DWPW_PF;
    f32x4 bw[PF][4];
    auto load_b = [&](int step, f32x4* dst) {
        // buffer loads: the weights' descriptor and the step's offset are wave-uniform (SGPRs), a lane supplies one 32-bit offset
        // register; the global_load form needs a 64-bit address per lane
        const unsigned soff = (unsigned)(step * nwt + wave_u) * 4096u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + j * 1024, soff, 0));
    };
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) load_b(j < nsteps ? j : nsteps - 1, bw[j]);   // in flight during phase 1

    // ---- phase 1: depthwise row block.  A thread keeps ONE 4-channel chunk (its 9 weight vectors + bias are loaded
    // once; NT is a multiple of C/4 for every layer shape) and walks groups of PXG consecutive output pixels.  When the
    // group lies in one image row (stride 1, dilation 1) its pixels share a 3 x (PXG+2) input window: the kernel is
    // bound by the NUMBER of vector-memory instructions (TA busy ~ kernel time in the PMC profile), not by bytes.
    constexpr int PXG = 2;
    static_assert(BM % PXG == 0, "row block must hold whole pixel groups");
    const int cg = p.C >> 2;
    if (!(DBG & 1)) {
        const int c = (tid % cg) * 4;
        f32x4 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const f32x4*)(p.dw_w + t * p.C + c);
        const f32x4 bias = *(const f32x4*)(p.dw_w + 9 * p.C + c);
        // input loads are buffer loads (wave-uniform descriptor, one 32-bit byte offset per lane; a tap outside the image gets
        // offset 2^31 >= num_records and reads zeros): while other workgroups of the CU are in their MFMA phase a 64-bit
        // per-lane address costs ~3x the issue time
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4), 0x00020000);
        const int pix_b = p.in_ld * 4, row_b = p.Wi * pix_b;
        auto ldw = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0)); };
        // pixel coordinates are carried incrementally in 32 bits (the host keeps every tensor below 2^31 bytes): one pair of
        // divisions per thread and tile instead of three 64-bit ones per pixel group
        const int M32 = (int)M, Wo = p.Wo, Ho = p.Ho;
        const int gadv = NT / cg;                        // pixel groups a thread advances per iteration
        const int adv = gadv * PXG, adv_y = adv / Wo, adv_x = adv - adv_y * Wo;      // wave-uniform
        int grp = tid / cg;
        int m = (int)m0 + grp * PXG;
        int xo, yo, img;
        {
            const unsigned q1 = (unsigned)m / (unsigned)Wo;
            xo = m - (int)q1 * Wo;
            img = (int)(q1 / (unsigned)Ho);
            yo = (int)q1 - img * Ho;
        }
        for (; grp < BM / PXG; grp += gadv) {
            const int row0 = grp * PXG;
            if (p.stride == 1 && p.dil == 1 && m + PXG <= M32 && xo + PXG <= Wo) {
                const int base = ((img * p.Hi + yo) * p.Wi + xo) * pix_b + c * 4;
                f32x4 win[3][PXG + 2];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = yo + ky - 1;
                    const bool rok = yy >= 0 && yy < p.Hi;
#pragma unroll
                    for (int j = 0; j < PXG + 2; ++j) {
                        const int xx = xo + j - 1;
                        win[ky][j] = ldw((rok && xx >= 0 && xx < p.Wi) ? (unsigned)(base + (ky - 1) * row_b + (j - 1) * pix_b) : 0x80000000u);
                    }
                }
#pragma unroll
                for (int i = 0; i < PXG; ++i) {
                    f32x4 acc = bias;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc += win[t / 3][i + t % 3] * wv[t];
                    acc.x = apply_act(acc.x, act_dw); acc.y = apply_act(acc.y, act_dw);
                    acc.z = apply_act(acc.z, act_dw); acc.w = apply_act(acc.w, act_dw);
                    *(f32x4*)(At + (row0 + i) * ldA + c) = acc;
                }
            } else {
                int xi = xo, yi = yo, im = img;
#pragma unroll
                for (int i = 0; i < PXG; ++i) {           // general path: borders of the row block, stride 2, dilation 2
                    const bool ok = m + i < M32;
                    const int yc = yi * p.stride, xc = xi * p.stride;
                    const int base = ((im * p.Hi + yc) * p.Wi + xc) * pix_b + c * 4;
                    f32x4 x[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int dy = (t / 3 - 1) * p.dil, dx = (t % 3 - 1) * p.dil;
                        const bool in = ok && yc + dy >= 0 && yc + dy < p.Hi && xc + dx >= 0 && xc + dx < p.Wi;
                        x[t] = ldw(in ? (unsigned)(base + dy * row_b + dx * pix_b) : 0x80000000u);
                    }
                    f32x4 acc = bias;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc += x[t] * wv[t];
                    acc.x = apply_act(acc.x, act_dw); acc.y = apply_act(acc.y, act_dw);
                    acc.z = apply_act(acc.z, act_dw); acc.w = apply_act(acc.w, act_dw);
                    *(f32x4*)(At + (row0 + i) * ldA + c) = acc;
                    if (++xi == Wo) { xi = 0; if (++yi == Ho) { yi = 0; ++im; } }
                }
            }
            m += adv;
            xo += adv_x; yo += adv_y;
            if (xo >= Wo) { xo -= Wo; ++yo; }
            while (yo >= Ho) { yo -= Ho; ++img; }
        }
    }
    __syncthreads();

    // ---- phase 2: pointwise GEMM, no barriers
    f32x4 acc[RT][2];
#pragma unroll
    for (int a = 0; a < RT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float* a_lane = At + r16 * ldA + 4 * q;
    // one K step: request step + PF - 1 (past the end: the last step again — the loads are UNCONDITIONAL and the ring
    // index static, so that the compiler can count the loads in flight: with a branch around them it waits for
    // vmcnt(0), i.e. for the step it has just requested, and nothing of a wave's weight stream overlaps its own MFMAs)
    auto one = [&](int step, auto P_) {
        constexpr int P = decltype(P_)::value;
        const int nxt = step + PF - 1 < nsteps ? step + PF - 1 : nsteps - 1;
        if (!(DBG & 2)) load_b(nxt, bw[(P + PF - 1) % PF]);
        __builtin_amdgcn_sched_barrier(0);           // the scheduler otherwise sinks the requests below the MFMAs of this step
        if (DBG & 4) return;
        const f32x4* bcur = bw[P];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 av[RT];
#pragma unroll
            for (int a = 0; a < RT; ++a) av[a] = *(const f32x4*)(a_lane + a * 16 * ldA + step * 32 + 16 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // packed B value index v = u*8 + j*2 + t  ->  register bcur[v >> 2][v & 3]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int v = u * 8 + j * 2 + t;
                    const float b = bcur[v >> 2][v & 3];
#pragma unroll
                    // weights are the A operand (rows = channels), activations the B operand (columns = pixels): a lane ends
                    // up with 4 CONSECUTIVE CHANNELS of one pixel -> 16-byte stores, full 128-byte lines per pixel and wave
                    for (int a = 0; a < RT; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, av[a][j], acc[a][t], 0, 0, 0);
                }
            }
        }
    };
    int s0 = 0;
    for (; s0 + PF <= nsteps; s0 += PF) {
        one(s0, std::integral_constant<int, 0>{});
        if (PF > 1) one(s0 + 1, std::integral_constant<int, 1 % PF>{});
        if (PF > 2) one(s0 + 2, std::integral_constant<int, 2 % PF>{});
    }
    if (s0 < nsteps) { one(s0, std::integral_constant<int, 0>{}); ++s0; }
    if (PF > 2 && s0 < nsteps) { one(s0, std::integral_constant<int, 1 % PF>{}); ++s0; }
    // epilogue: D layout row (channel) = (lane>>4)*4 + reg, col (pixel) = lane&15
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = wave * 32 + t * 16 + 4 * q;
        const f32x4 bias = *(const f32x4*)(p.pw_b + n);
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            const int64_t m = m0 + a * 16 + r16;
            if (m < M) {
                f32x4 v = acc[a][t] + bias;
                v.x = apply_act(v.x, act_pw); v.y = apply_act(v.y, act_pw);
                v.z = apply_act(v.z, act_pw); v.w = apply_act(v.w, act_pw);
                if (p.res) v += *(const f32x4*)(p.res + m * p.res_ld + n);
                *(f32x4*)(p.out + m * p.out_ld + n) = v;
            }
        }
    }
}

template <int BM, int NW, int DBG = 0, int ACT = -1>
static hipError_t launch_dwpw_t(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const size_t lds = (size_t)BM * (p.C + DWPW_F32_APAD) * sizeof(float);
    if ((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4 >= (1ll << 31)) return hipErrorInvalidValue;     // 32-bit buffer offsets (2 GiB of input per launch)
    const int nsplit = (p.cout / 32) / NW;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) {
        hipError_t e = attr.ensure((const void*)dwpw_kernel<BM, NW, DBG, ACT>, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((dwpw_kernel<BM, NW, DBG, ACT>), dim3((unsigned)((M + BM - 1) / BM), nsplit), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- fused depthwise -> pointwise, software-pipelined
// The 512-output f32 blocks at large M (batch 32: 30 % of the step).  In dwpw_kernel all 16 waves of the workgroup that owns the CU
// run the depthwise phase together (matrix pipes idle), then the K loop together: 600 us per 512 -> 512 layer against an MFMA
// bound of 410.  A v_mfma_f32_16x16x4_f32 holds its SIMD's issue port a quarter of its 32 cycles, so the vector ALU has room for
// the WHOLE depthwise phase (~10 % of the K loop's issue time) — if its instructions sit between MFMAs.  Here a PERSISTENT
// workgroup keeps TWO 32-row tiles in LDS: while the K loop multiplies tile t out of one, every K step also carries one piece of
// the depthwise computation of tile t + G into the other (step 4i: the nine window loads of the thread's i-th pixel, 4i+1 / 4i+2:
// the taps, 4i+3: activation + LDS store), so four K steps later the pixel is done and, with four waves per SIMD at different
// points of their steps, one wave's VALU work runs under the others' MFMAs.  One barrier per tile; the weight ring wraps from the
// last step to step 0 (the same for every tile), so the stream never drains; stores go straight from the accumulators.
// Depthwise arithmetic in dwpw_kernel's order (bias, taps row-major), pointwise sums in the same k order: bit-identical results.
template <int C>     // depthwise channels (256 | 512); 512 output channels = 16 waves x 32
__global__ void __launch_bounds__(1024) dwpw_pipe_kernel(DwPwParams p, int ntiles) {
    constexpr int BM = 32, NT = 1024, RT = BM / 16, NSTEP = C / 32;
    constexpr int ldA = C + DWPW_F32_APAD;
    constexpr int CG = C / 4;                        // 4-channel chunks per pixel
    constexpr int PAR = NT / CG;                     // pixels in flight across the workgroup (8 | 16)
    constexpr int NIT = BM / PAR;                    // pixels per thread and tile (4 | 2)
    static_assert(NIT * 4 == NSTEP, "one pixel per four K steps");
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* Wd = dsm + 2 * BM * ldA;                  // [10][C] depthwise weights + bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int M32 = p.N * p.Ho * p.Wo;               // host: < 2^31
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int qq = G >> 3, rem = G & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    for (int i = tid * 4; i < 10 * C; i += NT * 4) *(f32x4*)(Wd + i) = *(const f32x4*)(p.dw_w + i);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.pw_w, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4), 0x00020000);
    const int pix_b = p.in_ld * 4, row_b = p.Wi * pix_b;
    const int c = (tid % CG) * 4, prow = tid / CG;
    constexpr int NU = 16;                           // 32-channel output units = waves
    f32x4 bw[2][4];
    auto load_b = [&](int step, f32x4* dst) {
        const unsigned soff = (unsigned)(step * NU + wave_u) * 4096u;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + j * 1024, soff, 0));
    };
    // ---- one piece of the depthwise computation of pixel `it` of tile `tl` into the tile buffer Adst
    f32x4 win[9], pacc;
    auto p_piece = [&](int tl, int it, int j, float* Adst) {
        if (j == 0) {
            const int m = tl * BM + it * PAR + prow;
            const bool ok = m < M32;
            const unsigned q1 = (unsigned)m / (unsigned)p.Wo;
            const int xo = m - (int)q1 * p.Wo;
            const int img = (int)(q1 / (unsigned)p.Ho);
            const int yo = (int)q1 - img * p.Ho;
            const int yc = yo * p.stride, xc = xo * p.stride;
            const int base = ((img * p.Hi + yc) * p.Wi + xc) * pix_b + c * 4;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = (t / 3 - 1) * p.dil, dx = (t % 3 - 1) * p.dil;
                const bool in = ok && yc + dy >= 0 && yc + dy < p.Hi && xc + dx >= 0 && xc + dx < p.Wi;
                win[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, in ? (unsigned)(base + dy * row_b + dx * pix_b) : 0x80000000u, 0, 0));
            }
        } else if (j == 1) {
            pacc = *(const f32x4*)(Wd + 9 * C + c);
#pragma unroll
            for (int t = 0; t < 4; ++t) pacc += win[t] * *(const f32x4*)(Wd + t * C + c);
        } else if (j == 2) {
#pragma unroll
            for (int t = 4; t < 9; ++t) pacc += win[t] * *(const f32x4*)(Wd + t * C + c);
        } else {
            f32x4 v = pacc;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *(f32x4*)(Adst + (it * PAR + prow) * ldA + c) = v;
        }
    };
    __syncthreads();                                 // depthwise weights are in LDS
    int tile = bid;
    // prologue: the first tile's depthwise block, not overlapped
    if (tile < ntiles) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) p_piece(tile, it, j, dsm);
    }
    load_b(0, bw[0]);
    __syncthreads();
    int cur = 0;
    for (; tile < ntiles; tile += G) {
        const float* Ac = dsm + cur * BM * ldA;
        float* An = dsm + (cur ^ 1) * BM * ldA;
        const int tnext = tile + G < ntiles ? tile + G : tile;       // past the end: this tile again (harmless, keeps every load unconditional)
        f32x4 acc[RT][2];
#pragma unroll
        for (int a = 0; a < RT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* a_lane = Ac + r16 * ldA + 4 * q;
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            load_b((step + 1) % NSTEP, bw[(step + 1) & 1]);          // the ring wraps: step 0 of the next tile is the same weights
            __builtin_amdgcn_sched_barrier(0);
            p_piece(tnext, step >> 2, step & 3, An);
            const f32x4* bcur = bw[step & 1];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 av[RT];
#pragma unroll
                for (int a = 0; a < RT; ++a) av[a] = *(const f32x4*)(a_lane + a * 16 * ldA + step * 32 + 16 * u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int v = u * 8 + j * 2 + t;
                        const float b = bcur[v >> 2][v & 3];
#pragma unroll
                        for (int a = 0; a < RT; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, av[a][j], acc[a][t], 0, 0, 0);
                    }
            }
        }
        __syncthreads();                             // tile t is multiplied, tile t + G is convolved: the buffers swap roles
        cur ^= 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = wave_u * 32 + t * 16 + 4 * q;
            const f32x4 bias = *(const f32x4*)(p.pw_b + n);
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                const int m = tile * BM + a * 16 + r16;
                if (m < M32) {
                    f32x4 v = acc[a][t] + bias;
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    *(f32x4*)(p.out + (int64_t)m * p.out_ld + n) = v;
                }
            }
        }
    }
}

template <int C>
static hipError_t launch_dwpw_pipe_t(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const int ntiles = (int)((M + 31) / 32);
    const size_t lds = (size_t)(2 * 32 * (C + DWPW_F32_APAD) + 10 * C) * sizeof(float);
    static LdsAttrOnce attr;
    hipError_t e = attr.ensure((const void*)dwpw_pipe_kernel<C>, 160 * 1024);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    int grid = cus * (lds <= 80 * 1024 ? 2 : 1);
    if (T.dwpw_pp_grid > 0) grid = T.dwpw_pp_grid;   // LWP_DWPW_PP_GRID (tests: several tiles per workgroup at small M)
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((dwpw_pipe_kernel<C>), dim3(grid), dim3(1024), lds, s, p, ntiles);
    return hipGetLastError();
}

static hipError_t try_dwpw_pipe(const DwPwParams& p, hipStream_t s, bool* used) {
    *used = false;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dwpw_pipe == 0) return hipSuccess;                       // LWP_DWPW_PIPE "0": off (A/B); "1": at every size (tests)
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    if (p.cout != 512 || (p.C != 256 && p.C != 512) || p.res || p.act_dw != ACT_RELU || p.act_pw != ACT_RELU) return hipSuccess;
    if ((p.stride != 1 && p.stride != 2) || (p.in_ld & 3) || (p.out_ld & 3) || (((uintptr_t)p.out) & 15) || (((uintptr_t)p.in) & 15)) return hipSuccess;
    if (M >= (1ll << 31) - 64 || (int64_t)p.N * p.Hi * p.Wi * p.in_ld * 4 >= (1ll << 31)) return hipSuccess;
    if (T.dwpw_pipe != 1 && M < 32 * 2048) return hipSuccess;      // small problems: one tile per workgroup fills the chip better
    // measured at batch 32 (us, round 3): 512 -> 512 597.5 -> 576.8, dilation 2 648.6 -> 588.4, 256 -> 512 368.2 -> 373.7 (its K loop is
    // half as long: the pieces no longer hide) — the 256-channel block keeps the two-phase kernel unless forced.  At the clock the
    // chip holds under this load (~2.0 GHz) the MFMA bound of a 512 -> 512 layer is ~480 us, not the 410 of the 2.4 GHz peak.
    if (T.dwpw_pipe != 1 && p.C != 512) return hipSuccess;
    *used = true;
    LWP_VARIANT(p, "dwpw_pipe<%d>", p.C);
    return p.C == 512 ? launch_dwpw_pipe_t<512>(p, s) : launch_dwpw_pipe_t<256>(p, s);
}

bool dwpw_supported(int C, int cout) {
    return C % 32 == 0 && (cout == 64 || cout == 128 || cout == 256 || cout == 512);
}

hipError_t launch_dwpw(const DwPwParams& p_in, hipStream_t s) {
    const DwPwParams& p = p_in;
    {
        bool used = false;
        hipError_t e = try_dwpw_tiled_f32(p, s, &used);
        if (e != hipSuccess || used) return e;
        e = try_dwpw_pipe(p, s, &used);
        if (e != hipSuccess || used) return e;
    }
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const int nw = p.cout / 32;
    // rows per workgroup: the largest of 64 / 32 / 16 that still leaves ~2-4 workgroups per CU (measured at batch 1:
    // model.1 (943 blocks of 64) 14.7 -> 12.1 us, model.2/3 (472 blocks of 32) 12.9 -> 10.1 / 14.6 -> 12.7 us; the
    // 46 x 82 layers (236 blocks of 16) lose with anything larger).  (BM*8 depthwise items must fit 2 per thread.)
    int bm = 16;
    if (M / 32 >= 450) bm = 32;
    if (M / 64 >= 900) bm = 64;
    // 512 output channels (16 waves): a 64-row tile is 132 KB of LDS, so ONE workgroup owns the CU and its depthwise phase runs
    // with the matrix pipes idle; with 32 rows (66 KB) a second workgroup's K loop covers it.  Measured at batch 32 (round 3):
    // 512 -> 512 630.6 -> 600.7 us, dilation 2 657.6 -> 644.7, 256 -> 512 360.5 -> 351.3.  (Measured and dropped: 8 waves x 64
    // channels on 32-row tiles, 128 VGPRs, so that two WHOLE workgroups share the CU — 662.8 us against 601.3: the per-wave
    // weight stream doubles and a wave's 16 MFMAs per step no longer hide it.)
    if (bm == 64 && nw >= 16) bm = 32;
    int nw_wg = nw;                                  // waves per workgroup (column split = nw / nw_wg)
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dwpw_bm) bm = T.dwpw_bm;                   // LWP_DWPW_BM
    if (T.dwpw_nw > 0 && T.dwpw_nw < nw && nw % T.dwpw_nw == 0) nw_wg = T.dwpw_nw;   // LWP_DWPW_NW
    if (bm > 64) bm = 64;
    while (bm > 16 && (size_t)bm * (p.C + DWPW_F32_APAD) * sizeof(float) > 150 * 1024) bm >>= 1;
    LWP_VARIANT(p, "dwpw<%d,%d>", bm, nw_wg);
#ifdef LWP_ABLATION
    const int d = T.dwpw_debug;
#define DP_DBG(BM_, NW_, D_) if (bm == BM_ && nw_wg == NW_ && d == D_) return launch_dwpw_t<BM_, NW_, D_>(p, s);
#define DP_DBGS(BM_, NW_) DP_DBG(BM_, NW_, 1) DP_DBG(BM_, NW_, 2) DP_DBG(BM_, NW_, 3) DP_DBG(BM_, NW_, 4) DP_DBG(BM_, NW_, 6) DP_DBG(BM_, NW_, 7) DP_DBG(BM_, NW_, 8)
    DP_DBGS(16, 16) DP_DBGS(16, 8) DP_DBGS(16, 4)
#undef DP_DBGS
#undef DP_DBG
#endif
    const bool relu = p.act_dw == ACT_RELU && p.act_pw == ACT_RELU;
#define DP_CASE(BM_, NW_) if (bm == BM_ && nw_wg == NW_) return relu ? launch_dwpw_t<BM_, NW_, 0, ACT_RELU>(p, s) : launch_dwpw_t<BM_, NW_>(p, s);
    DP_CASE(16, 2) DP_CASE(32, 2) DP_CASE(64, 2)
    DP_CASE(16, 4) DP_CASE(32, 4) DP_CASE(64, 4)
    DP_CASE(16, 8) DP_CASE(32, 8) DP_CASE(64, 8)
    DP_CASE(16, 16) DP_CASE(32, 16) DP_CASE(64, 16)
#undef DP_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------- stage heads, both 1x1 convs in one kernel
// out[m] = W1 . relu(W0 . x[m] + b0) + b1 (with_mobilenet.py:32-45, the merged heat / PAF pair) for small M (batch 1: 3772
// pixels, where each of the two GEMM launches is mostly launch / ramp / tail).  Workgroup = 16 pixels x 8 waves; the hidden
// dimension is split over the waves in tiles of 16 channels.  v_mfma_f32_16x16x4_f32 with the weights as the A operand:
//   GEMM 1: D[hidden 16][pixel 16].  The summation index is free to permute: k-slot (step s = 4 u + c, lane group q) is read as
//           input channel 32 q + 4 u + c, so a lane's operands are 128 contiguous bytes of its own row — eight 16-byte loads of
//           the PLAIN [hidden][128] weight rows (fully coalesced, no packing), and of the pixel's own NHWC row, kept for all tiles.
//   D leaves lane (pixel i, q) with hidden channels 4 q + r, r = 0..3: after bias + ReLU, register r IS the B operand of GEMM 2's
//   step r if k-slot q of that step means hidden 4 q + r — i.e. the W1 fragment of lane (out channel i, q) is the 16-byte vector
//   W1[out][16 tile + 4 q .. + 3].  The hidden values never leave the lane's registers.
//   GEMM 2 accumulates all 64 (57 used) output channels per wave over its hidden tiles; the NW partial results are summed in the
//   fixed order 0..NW-1 through LDS.  Weight loads are buffer loads with wave-uniform tile offsets, requested one tile ahead.
template <int NW>
__global__ void __launch_bounds__(NW * 64) heads_f32_kernel(HeadsParams p) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];     // [NW][4][64] f32x4 partial outputs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int M = p.N * p.H * p.W;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    const int m = bid * 16 + i16;
    const bool mok = m < M;
    const float* in = (const float*)p.in;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((int64_t)M * p.in_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t w0r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, p.hidden * 128 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 64 * p.hidden * 4, 0x00020000);
    auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };

    const int ntiles = p.hidden / 16;
    const int my = wave < ntiles ? (ntiles - wave + NW - 1) / NW : 0;            // tiles wave, wave + NW, ...
    constexpr int HPF = 3;                              // weight tiles in registers: HPF - 1 requested ahead (a tile's MFMAs take ~0.8 us, an L2 round trip 1-2)
    f32x4 w0v[HPF][8], w1v[HPF][4], b0v[HPF];
    const __amdgpu_buffer_rsrc_t b0r = __builtin_amdgcn_make_buffer_rsrc((void*)p.b0, 0, p.hidden * 4, 0x00020000);
    const unsigned w0_lane = (unsigned)(i16 * 128 + 32 * q) * 4u, w1_lane = (unsigned)(i16 * p.hidden + 4 * q) * 4u;
    auto request = [&](int k, f32x4* a, f32x4* b, f32x4* bias) {                               // k-th tile of this wave (clamped: no branch around loads)
        const int ht = wave + (k < my ? k : (my > 0 ? my - 1 : 0)) * NW;
        const int htc = ht < ntiles ? ht : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = ld(w0r, w0_lane + 16u * u, (unsigned)htc * (16u * 128u * 4u));
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = ld(w1r, w1_lane + (unsigned)(t * 16 * p.hidden) * 4u, (unsigned)htc * 64u);
        *bias = ld(b0r, 16u * q, (unsigned)htc * 64u);                               // the tile's bias rides with its weights (exact vmcnt counting)
    };
    request(0, w0v[0], w1v[0], &b0v[0]);
    if (HPF > 2) request(1, w0v[1], w1v[1], &b0v[1]);
    f32x4 xf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xf[u] = ld(xr, mok ? (unsigned)((m * p.in_ld + 32 * q + 4 * u) * 4) : 0x80000000u, 0);

    f32x4 acc2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // block-diagonal second conv (p.out_split > 0: merged heat / PAF pair): hidden channels < hsplit feed outputs < out_split only
    const int hsplit = p.out_split > 0 ? p.hidden / 2 : p.hidden;
    const int t_last_lo = p.out_split > 0 ? (p.out_split + 15) / 16 : 4;     // output tiles [0, t_last_lo) hold outputs of the first block
    const int t_first_hi = p.out_split > 0 ? p.out_split / 16 : 0;          // output tiles [t_first_hi, 4) hold outputs of the second
    auto one = [&](int k, auto P_) {
        constexpr int P = decltype(P_)::value;
        request(k + HPF - 1, w0v[(P + HPF - 1) % HPF], w1v[(P + HPF - 1) % HPF], &b0v[(P + HPF - 1) % HPF]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 b0 = b0v[P];
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0v[P][u][c], xf[u][c], acc1, 0, 0, 0);
        f32x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = fmaxf(acc1[r] + b0[r], 0.f);
        // merged heads: W1 is block-diagonal (heat outputs [0, out_split) read the first half of the hidden vector, PAF outputs the
        // second), so a hidden tile only feeds the output tiles its block touches: 2 (heat) or 3 (PAF) of the 4 at 19 + 38 channels
        const int ht = wave + (k < my ? k : 0) * NW;                     // wave-uniform
        const bool first_half = ht * 16 < hsplit;
        const int t_lo = first_half ? 0 : t_first_hi, t_hi = first_half ? t_last_lo : 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < t_lo || t >= t_hi) continue;                         // scalar branch around whole MFMA groups
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[P][t][r], hv[r], acc2[t], 0, 0, 0);
        }
    };
    int k = 0;
    for (; k + HPF <= my; k += HPF) {
        one(k, std::integral_constant<int, 0>{});
        one(k + 1, std::integral_constant<int, 1 % HPF>{});
        if (HPF > 2) one(k + 2, std::integral_constant<int, 2 % HPF>{});
    }
    if (k < my) { one(k, std::integral_constant<int, 0>{}); ++k; }
    if (HPF > 2 && k < my) { one(k, std::integral_constant<int, 1 % HPF>{}); ++k; }

    // fixed-order reduction of the NW partial tiles: [wave][t][lane] f32x4
#pragma unroll
    for (int t = 0; t < 4; ++t) *(f32x4*)(hsm + ((wave * 4 + t) * 64 + lane) * 4) = acc2[t];
    __syncthreads();
    if (tid >= 256) return;
    const int t = tid >> 6;                                   // output-channel tile of this thread (its lane keeps pixel / q)
    f32x4 v = *(const f32x4*)(hsm + ((0 * 4 + t) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *(const f32x4*)(hsm + ((w * 4 + t) * 64 + lane) * 4);
    const int n = 16 * t + 4 * q;                             // lane (pixel i16, q) holds output channels n .. n + 3
    if (!mok || n >= p.cout) return;
    v += *(const f32x4*)(p.b1 + n);
    float* out = (float*)p.out;
    if (((p.out_ld & 3) == 0) && ((((uintptr_t)out) & 15) == 0) && n + 3 < p.cout) *(f32x4*)(out + (int64_t)m * p.out_ld + n) = v;
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.cout) out[(int64_t)m * p.out_ld + n + r] = v[r];
    }
    if (p.out_nchw || p.out_nchw2) {
        const int HW = p.H * p.W;
        const int img = m / HW, pix = m - img * HW;
        const int c0 = p.out_split > 0 ? p.out_split : p.cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = n + r;
            if (ch >= p.cout) continue;
            if (ch < c0) { if (p.out_nchw) p.out_nchw[((int64_t)img * c0 + ch) * HW + pix] = v[r]; }
            else if (p.out_nchw2) p.out_nchw2[((int64_t)img * (p.cout - c0) + (ch - c0)) * HW + pix] = v[r];
        }
    }
}

// ---------------------------------------------------------------------------------------- stage heads at large M, weights through LDS
// The same pair (hidden values never leave the lane's registers, same MFMA operand conventions as heads_f32_kernel) for M > 4096:
// there every 16-pixel workgroup of the kernel above would stream the pair's whole weights (768 KB for the initial stage) from L2,
// and the two-GEMM form writes and reads the hidden tensor instead (494 MB per direction at batch 32: 335 + 160 us for the initial
// stage, 91 + 54 for a refinement stage).  Here a workgroup owns 64 PT pixels (4 waves x PT 16-pixel tiles) and walks the hidden
// dimension in chunks of 32: the chunk's W0 rows [32][128] and W1 columns [64][32] are staged ONCE per workgroup in LDS (double
// buffered, requested a chunk ahead into registers), every fragment read feeds PT MFMAs, and each wave accumulates all 64 (57 used)
// outputs of its own pixels over all chunks — no cross-wave reduction.  k-slot (step 4u + c, lane group q) of GEMM 1 is input
// channel 16u + 4q + c: a lane's eight fragments are 16-byte reads one slot apart between the lane groups, and with rows of
// 128 + 8 / 32 + 8 floats every ds_read_b128 is bank-conflict free (slot = row x stride + q, stride = 2 mod 4: see DWPW_APAD).
constexpr int HL_HC = 32, HL_LD0 = 128 + 8, HL_LD1 = HL_HC + 8;
template <int PT>
__global__ void __launch_bounds__(256) heads_f32_lds_kernel(HeadsParams p) {
    extern __shared__ __attribute__((aligned(16))) float hl_sm[];
    float* W0s = hl_sm;                                              // [2][HL_HC][HL_LD0]
    float* W1s = W0s + 2 * HL_HC * HL_LD0;                           // [2][64][HL_LD1]
    float* B0s = W1s + 2 * 64 * HL_LD1;                              // [hidden]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int M = p.N * p.H * p.W;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    int mr[PT];
    bool mok[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) { mr[i] = bid * (64 * PT) + (wave * PT + i) * 16 + i16; mok[i] = mr[i] < M; }
    const int nch = p.hidden / HL_HC;
    const float* in = (const float*)p.in;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((int64_t)M * p.in_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t w0r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, p.hidden * 128 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 64 * p.hidden * 4, 0x00020000);
    auto ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
    // weight staging: W0 chunk = 32 rows x 512 B (4 x 16 B per thread), W1 chunk = 64 rows x 128 B (2 x 16 B per thread)
    f32x4 st0[4], st1[2];
    auto request = [&](int c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = tid + u * 256, row = ch >> 5, col = (ch & 31) * 4;
            st0[u] = ld(w0r, (unsigned)(row * 128 + col) * 4u, (unsigned)c * (HL_HC * 128 * 4));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + u * 256, row = ch >> 3, col = (ch & 7) * 4;
            st1[u] = ld(w1r, (unsigned)(row * p.hidden + col) * 4u, (unsigned)c * (HL_HC * 4));
        }
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = tid + u * 256, row = ch >> 5, col = (ch & 31) * 4;
            *(f32x4*)(W0s + (buf * HL_HC + row) * HL_LD0 + col) = st0[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + u * 256, row = ch >> 3, col = (ch & 7) * 4;
            *(f32x4*)(W1s + (buf * 64 + row) * HL_LD1 + col) = st1[u];
        }
    };
    request(0);
    // the wave's activations: fragment u of pixel m = x[m][16 u + 4 q .. + 3]
    f32x4 xf[PT][8];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) xf[i][u] = ld(xr, mok[i] ? (unsigned)((mr[i] * p.in_ld + 16 * u + 4 * q) * 4) : 0x80000000u, 0);
    for (int i = tid * 4; i < p.hidden; i += 256 * 4) *(f32x4*)(B0s + i) = *(const f32x4*)(p.b0 + i);
    land(0);
    __syncthreads();

    f32x4 acc2[PT][4];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // block-diagonal second conv (p.out_split > 0: merged heat / PAF pair): hidden channels < hsplit feed outputs < out_split only
    const int hsplit = p.out_split > 0 ? p.hidden / 2 : p.hidden;
    const int t_last_lo = p.out_split > 0 ? (p.out_split + 15) / 16 : 4;
    const int t_first_hi = p.out_split > 0 ? p.out_split / 16 : 0;
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        request(c + 1 < nch ? c + 1 : c);                            // unconditional (the last chunk again): exact vmcnt counting
        __builtin_amdgcn_sched_barrier(0);                           // keep the requests ahead of the chunk's MFMAs
#pragma unroll
        for (int j = 0; j < HL_HC / 16; ++j) {                       // the chunk's two 16-channel hidden tiles
            const float* w0 = W0s + (buf * HL_HC + j * 16 + i16) * HL_LD0 + 4 * q;
            f32x4 wv[8];                                             // all eight fragment reads in flight before the first MFMA
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = *(const f32x4*)(w0 + 16 * u);
            const f32x4 b0 = *(const f32x4*)(B0s + c * HL_HC + j * 16 + 4 * q);
            const int ht16 = (c * HL_HC + j * 16);                   // first hidden channel of the tile (wave-uniform)
            const bool first_half = ht16 < hsplit;
            const int t_lo = first_half ? 0 : t_first_hi, t_hi = first_half ? t_last_lo : 4;
            f32x4 w1v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w1v[t] = *(const f32x4*)(W1s + (buf * 64 + t * 16 + i16) * HL_LD1 + j * 16 + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc1[PT];
#pragma unroll
            for (int i = 0; i < PT; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int i = 0; i < PT; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][cc], xf[i][u][cc], acc1[i], 0, 0, 0);
            f32x4 hv[PT];
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[i][r] = fmaxf(acc1[i][r] + b0[r], 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < t_lo || t >= t_hi) continue;                 // scalar branch around whole MFMA groups
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < PT; ++i) acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v[t][r], hv[i][r], acc2[i][t], 0, 0, 0);
            }
        }
        if (c + 1 < nch) land(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane (pixel i16, q) holds output channels 16 t + 4 q + r
    float* out = (float*)p.out;
    const int HW = p.H * p.W;
    const int c0 = p.out_split > 0 ? p.out_split : p.cout;
    const bool out_vec = ((p.out_ld & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        if (!mok[i]) continue;
        const int m = mr[i];
        const int img = m / HW, pix = m - img * HW;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 16 * t + 4 * q;
            if (n >= p.cout) continue;
            f32x4 v = acc2[i][t];
            v += *(const f32x4*)(p.b1 + n);
            if (out_vec && n + 3 < p.cout) *(f32x4*)(out + (int64_t)m * p.out_ld + n) = v;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.cout) out[(int64_t)m * p.out_ld + n + r] = v[r];
            }
            if (p.out_nchw || p.out_nchw2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = n + r;
                    if (ch >= p.cout) continue;
                    if (ch < c0) { if (p.out_nchw) p.out_nchw[((int64_t)img * c0 + ch) * HW + pix] = v[r]; }
                    else if (p.out_nchw2) p.out_nchw2[((int64_t)img * (p.cout - c0) + (ch - c0)) * HW + pix] = v[r];
                }
            }
        }
    }
}

bool heads_f32_supported(int cin_pad, int hidden, int cout_pad, int64_t M, const Tuning* tune) {
    const int64_t max_m = (tune && tune->heads_f32_max_m > 0) ? tune->heads_f32_max_m : 4096;     // LWP_HEADS_F32_MAXM (tests)
    // every 16-pixel workgroup streams all of W0 and W1 (768 KB for the initial stage): only while the grid is ONE round of the
    // chip (measured at batch 2, 472 workgroups: 57 us against 27 + 17 for the two GEMMs; batch 1: 29.5 against 18.6 + 9.3 with
    // two launch floors less)
    if (!(cin_pad == 128 && hidden % 16 == 0 && hidden >= 128 && hidden <= 4096 && cout_pad == 64)) return false;
    if (M <= max_m) return true;
    // larger M: the LDS-staged form (LWP_HEADS_F32_LDS=0: the two GEMMs, A/B)
    return hidden % HL_HC == 0 && !(tune && tune->heads_f32_lds == 0);
}

hipError_t launch_heads_f32(const HeadsParams& p, hipStream_t s) {
    constexpr int NW = 8;
    const int64_t M = (int64_t)p.N * p.H * p.W;
    if (M * p.in_ld * 4 >= (1ll << 31) || (p.in_ld & 3)) return hipErrorInvalidValue;
    const int64_t max_m = (p.tune && p.tune->heads_f32_max_m > 0) ? p.tune->heads_f32_max_m : 4096;
    if (M > max_m) {
        constexpr int PT = 2;
        if (M >= (1ll << 31) - 64 * PT) return hipErrorInvalidValue;
        const size_t lds2 = (size_t)(2 * HL_HC * HL_LD0 + 2 * 64 * HL_LD1 + p.hidden) * sizeof(float);
        static LdsAttrOnce attr;
        hipError_t e = attr.ensure((const void*)heads_f32_lds_kernel<PT>, 96 * 1024);
        if (e != hipSuccess) return e;
        LWP_VARIANT(p, "heads_f32_lds<%d>", PT);
        hipLaunchKernelGGL(heads_f32_lds_kernel<PT>, dim3((unsigned)((M + 64 * PT - 1) / (64 * PT))), dim3(256), lds2, s, p);
        return hipGetLastError();
    }
    const size_t lds = (size_t)NW * 4 * 64 * 4 * sizeof(float);
    LWP_VARIANT(p, "heads_f32<%d>", NW);
    hipLaunchKernelGGL(heads_f32_kernel<NW>, dim3((unsigned)((M + 15) / 16)), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- switches
const Tuning& default_tuning() { static const Tuning t; return t; }
Tuning tuning_from_env() {
    Tuning t;
    auto geti = [](const char* name, int* dst) { const char* e = getenv(name); if (e && e[0]) *dst = atoi(e); };
    auto digit = [](const char* name, int* dst) { const char* e = getenv(name); if (e && e[0]) *dst = e[0] - '0'; };
    geti("LWP_STEM_TY", &t.stem_ty); digit("LWP_STEM_WL", &t.stem_wl); geti("LWP_STEM_DEBUG", &t.stem_debug);
    digit("LWP_DW_TILED", &t.dw_tiled); geti("LWP_DW_CC", &t.dw_cc); geti("LWP_DW_PH", &t.dw_ph);
    digit("LWP_GEMM_WP", &t.gemm_wp);
    if (const char* e = getenv("LWP_GEMM_C3")) t.has_c3 = sscanf(e, "%d,%d,%d", &t.c3[0], &t.c3[1], &t.c3[2]) == 3;
    if (const char* e = getenv("LWP_GEMM_PW")) t.has_pw = sscanf(e, "%d,%d,%d", &t.pw[0], &t.pw[1], &t.pw[2]) == 3;
    geti("LWP_DWPW_BM", &t.dwpw_bm); geti("LWP_DWPW_NW", &t.dwpw_nw); geti("LWP_DWPW_DEBUG", &t.dwpw_debug); geti("LWP_DWPWH_DEBUG", &t.dwpwh_debug);
    digit("LWP_DWPW_PIPE", &t.dwpw_pipe);
    digit("LWP_DWPW_TILED", &t.dwpw_tiled); geti("LWP_DWPW_TILED_WGS", &t.dwpw_tiled_wgs);
    digit("LWP_DWPW_PP", &t.dwpw_pp); geti("LWP_DWPW_PP_GRID", &t.dwpw_pp_grid);
    geti("LWP_HEADS_RM", &t.heads_rm);
    digit("LWP_GEMMH_PERSIST", &t.gemmh_persist); digit("LWP_GEMMH_FOLD", &t.gemmh_fold);
    if (const char* e = getenv("LWP_GEMMH_AR")) {
        if (e[0] == '0' && e[1] == 0) t.gemmh_ar_off = 1;
        else t.has_gemmh_ar = sscanf(e, "%d,%d,%d,%d", &t.gemmh_ar[0], &t.gemmh_ar[1], &t.gemmh_ar[2], &t.gemmh_ar[3]) == 4;
    }
    if (const char* e = getenv("LWP_GEMMH_AR_FORCE")) t.gemmh_ar_force = e[0] == '1';
    geti("LWP_GEMMH_DEBUG", &t.gemmh_debug);
    if (const char* e = getenv("LWP_GEMMH")) t.has_gemmh = sscanf(e, "%d,%d,%d,%d", &t.gemmh[0], &t.gemmh[1], &t.gemmh[2], &t.gemmh[3]) == 4;
    digit("LWP_UPSAMPLE_TILED", &t.upsample_tiled);
    digit("LWP_PEAK_TILE", &t.peak_tile); digit("LWP_PAIR_FORM", &t.pair_form); digit("LWP_POST_NCHW", &t.post_nchw); digit("LWP_HEADS_F32_LDS", &t.heads_f32_lds); digit("LWP_MS_FUSED", &t.ms_fused); digit("LWP_MS_VEC", &t.ms_vec); geti("LWP_MS_TX", &t.ms_tx); digit("LWP_HOST_FETCH_DMA", &t.host_fetch_dma); geti("LWP_DWPW_LDS_PAD", &t.dwpw_lds_pad_kb);
    geti("LWP_HEADS_F32_MAXM", &t.heads_f32_max_m);
    geti("LWP_MAX_FRAMES_PER_PASS", &t.max_frames_per_pass);
    return t;
}

// ---------------------------------------------------------------------------------------- layout helper
__global__ void __launch_bounds__(256) nchw_from_nhwc_kernel(const float* src, int src_ld, float* dst, int N, int HW, int C) {
    const int64_t total = (int64_t)N * C * HW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int pix = (int)(idx % HW);
    const int64_t t = idx / HW;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    dst[idx] = src[((int64_t)n * HW + pix) * src_ld + c];
}
hipError_t launch_nchw_from_nhwc(const float* src, int src_ld, float* dst, int N, int HW, int C, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    hipLaunchKernelGGL(nchw_from_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, src_ld, dst, N, HW, C);
    return hipGetLastError();
}

}  // namespace lwp
