// Post-processing kernels for gfx950: bicubic up-sampling, key-point extraction, PAF pair scoring
// and pose assembly.  Every result is bit-identical to the CPU oracle (oracle/post_ref.py), which is
// pinned to the reference (modules/keypoints.py:11-201, demo.py:70-76).
//
// Exactness rules used throughout this file:
//   * no FMA contraction (pragma below + explicit __fmul_rn/__fadd_rn/__dmul_rn/__dadd_rn);
//   * the up-sampled value of a pixel is always computed "horizontal pass, then vertical pass",
//     left-to-right float32 sums, whether a map is materialised (upsample_kernel) or sampled on the
//     fly (find_peaks / score_pairs) — so both routes give the same bits;
//   * line-integral arithmetic is float64 exactly as NumPy does it (keypoints.py:11-13,104-136).
#pragma clang fp contract(off)
#include <type_traits>
#include "lwp_internal.h"

namespace lwp {

// ------------------------------------------------------------------------------------------------
// bicubic phase table: for an integer ratio r the source position of full-res index X = q*r + ph is
// q + off[ph] - 1 .. q + off[ph] + 2 with weights w[ph][0..3]  (OpenCV float path, A = -0.75).
struct CubicTable {
    int off[8];
    float w[8][4];
};
__constant__ CubicTable g_cubic[2];     // [0]: ratio 4, [1]: ratio 8

static void cubic_coeffs_host(float x, float* c) {
    const float A = -0.75f;
    volatile float t0 = x + 1.f;
    volatile float a = A * t0; a = a - 5.f * A; a = a * t0; a = a + 8.f * A; a = a * t0; a = a - 4.f * A;
    c[0] = a;
    volatile float b = (A + 2.f) * x; b = b - (A + 3.f); b = b * x; b = b * x; b = b + 1.f;
    c[1] = b;
    volatile float u = 1.f - x;
    volatile float d = (A + 2.f) * u; d = d - (A + 3.f); d = d * u; d = d * u; d = d + 1.f;
    c[2] = d;
    volatile float e = 1.f - c[0]; e = e - c[1]; e = e - c[2];
    c[3] = e;
}

hipError_t init_cubic_tables() {
    CubicTable t[2];
    const int ratios[2] = {4, 8};
    for (int k = 0; k < 2; ++k) {
        const int r = ratios[k];
        for (int ph = 0; ph < 8; ++ph) {
            t[k].off[ph] = 0;
            for (int j = 0; j < 4; ++j) t[k].w[ph][j] = 0.f;
        }
        for (int ph = 0; ph < r; ++ph) {
            const float fx = (float)((ph + 0.5) * (1.0 / r) - 0.5);
            const int sx = (int)floorf(fx);
            const float frac = fx - (float)sx;
            t[k].off[ph] = sx;
            cubic_coeffs_host(frac, t[k].w[ph]);
        }
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cubic), t, sizeof(t));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// value of the virtually up-sampled map (ratio 4 or 8; ratio 1 = direct read) at full-res (Y, X)
__device__ __forceinline__ float sample_map(const MapView& v, int n, int c, int Y, int X, int ratio) {
    const float* base = v.base + (int64_t)n * v.ns + (int64_t)c * v.cs;
    if (ratio == 1) return base[(int64_t)Y * v.ys + (int64_t)X * v.xs];
    const CubicTable& t = g_cubic[ratio == 4 ? 0 : 1];
    const int qx = X / ratio, px = X - qx * ratio;
    const int qy = Y / ratio, py = Y - qy * ratio;
    const int sx = qx + t.off[px], sy = qy + t.off[py];
    int64_t xo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xo[k] = (int64_t)clampi(sx - 1 + k, 0, v.w - 1) * v.xs;
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* row = base + (int64_t)clampi(sy - 1 + k, 0, v.h - 1) * v.ys;
        float a = __fmul_rn(row[xo[0]], t.w[px][0]);
        a = __fadd_rn(a, __fmul_rn(row[xo[1]], t.w[px][1]));
        a = __fadd_rn(a, __fmul_rn(row[xo[2]], t.w[px][2]));
        a = __fadd_rn(a, __fmul_rn(row[xo[3]], t.w[px][3]));
        rows[k] = a;
    }
    float o = __fmul_rn(rows[0], t.w[py][0]);
    o = __fadd_rn(o, __fmul_rn(rows[1], t.w[py][1]));
    o = __fadd_rn(o, __fmul_rn(rows[2], t.w[py][2]));
    o = __fadd_rn(o, __fmul_rn(rows[3], t.w[py][3]));
    return o;
}

// The same value with FOUR 16-byte loads instead of sixteen 4-byte ones when the four columns of the footprint are four
// consecutive floats (x stride 1, no clamping at the left / right border): a scattered gather costs the CU's address unit one
// cycle per lane and instruction whatever its width, and score_pairs_kernel is bound by exactly that (32 gathers per lane and
// sample).  The loads are only 4-byte aligned (global memory allows it).  Same products, same order: the same bits.
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };
__device__ __forceinline__ float sample_map_rows(const MapView& v, int n, int c, int Y, int X, int ratio) {
    if (ratio == 1 || v.xs != 1) return sample_map(v, n, c, Y, X, ratio);
    const CubicTable& t = g_cubic[ratio == 4 ? 0 : 1];
    const int qx = X / ratio, px = X - qx * ratio;
    const int sx = qx + t.off[px];
    if (sx < 1 || sx + 2 > v.w - 1) return sample_map(v, n, c, Y, X, ratio);
    const float* base = v.base + (int64_t)n * v.ns + (int64_t)c * v.cs + (sx - 1);
    const int qy = Y / ratio, py = Y - qy * ratio;
    const int sy = qy + t.off[py];
    F4U r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = *(const F4U*)(base + (int64_t)clampi(sy - 1 + k, 0, v.h - 1) * v.ys);
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = __fmul_rn(r[k].v[0], t.w[px][0]);
        a = __fadd_rn(a, __fmul_rn(r[k].v[1], t.w[px][1]));
        a = __fadd_rn(a, __fmul_rn(r[k].v[2], t.w[px][2]));
        a = __fadd_rn(a, __fmul_rn(r[k].v[3], t.w[px][3]));
        rows[k] = a;
    }
    float o = __fmul_rn(rows[0], t.w[py][0]);
    o = __fadd_rn(o, __fmul_rn(rows[1], t.w[py][1]));
    o = __fadd_rn(o, __fmul_rn(rows[2], t.w[py][2]));
    o = __fadd_rn(o, __fmul_rn(rows[3], t.w[py][3]));
    return o;
}

// ------------------------------------------------------------------------------------------------ upsample
// dst: N x (h*r) x (w*r) x C, one thread per output element, channel fastest (coalesced stores)
__global__ void __launch_bounds__(256) upsample_kernel(MapView src, int N, int C, int ratio, float* dst) {
    const int Hf = src.h * ratio, Wf = src.w * ratio;
    const int64_t total = (int64_t)N * Hf * Wf * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int X = (int)(r % Wf);
    r /= Wf;
    const int Y = (int)(r % Hf);
    const int n = (int)(r / Hf);
    dst[idx] = sample_map(src, n, c, Y, X, ratio);
}
// Tiled form for the materialised maps (infer_fast, val.infer): one workgroup = UTY x UTX full-resolution pixels x all C
// channels.  The low-resolution patch (with the 2-pixel cubic halo) is read ONCE into LDS, then the horizontal pass and
// the vertical pass run from LDS with exactly the arithmetic of sample_map (same products, same left-to-right sums: the
// horizontal value of a (source row, X) pair does not depend on Y), and the tile is written channel-fastest, i.e. as
// contiguous UTX * C float runs.  The per-element kernel above issues 16 global loads per output element and was
// bound by the texture addresser: 5.4 ms per launch at 32 x 552 x 984 x 38 against ~0.7 ms of write time.
constexpr int UTY = 8, UTX = 32;
template <int R>
__global__ void __launch_bounds__(256) upsample_tiled_kernel(MapView src, int C, float* dst) {
    constexpr int LH = UTY / R + 4 + (UTY % R ? 1 : 0), LW = UTX / R + 4 + (UTX % R ? 1 : 0);   // patch rows / cols (halo 2 each side)
    extern __shared__ float usm[];
    float* lo = usm;                                 // [LH][LW][C]
    float* hz = usm + LH * LW * C;                   // [LH][UTX][C]
    const CubicTable& ct = g_cubic[R == 4 ? 0 : 1];
    const int Hf = src.h * R, Wf = src.w * R;
    const int tiles_x = (Wf + UTX - 1) / UTX;
    const int X0 = (blockIdx.x % tiles_x) * UTX, Y0 = (blockIdx.x / tiles_x) * UTY;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const float* base = src.base + (int64_t)n * src.ns;
    const int ly0 = Y0 / R - 2, lx0 = X0 / R - 2;
    for (int i = tid; i < LH * LW * C; i += 256) {
        const int c = i % C, k = (i / C) % LW, j = i / (C * LW);
        lo[i] = base[(int64_t)clampi(ly0 + j, 0, src.h - 1) * src.ys + (int64_t)clampi(lx0 + k, 0, src.w - 1) * src.xs + (int64_t)c * src.cs];
    }
    __syncthreads();
    for (int i = tid; i < LH * UTX * C; i += 256) {          // horizontal pass
        const int c = i % C, lx = (i / C) % UTX, j = i / (C * UTX);
        const int X = X0 + lx;
        float a = 0.f;
        if (X < Wf) {
            const int qx = X / R, px = X - qx * R;
            const float* row = lo + ((j * LW) + (qx + ct.off[px] - 1 - lx0)) * C + c;      // first tap
            a = __fmul_rn(row[0], ct.w[px][0]);
            a = __fadd_rn(a, __fmul_rn(row[C], ct.w[px][1]));
            a = __fadd_rn(a, __fmul_rn(row[2 * C], ct.w[px][2]));
            a = __fadd_rn(a, __fmul_rn(row[3 * C], ct.w[px][3]));
        }
        hz[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < UTY * UTX * C; i += 256) {         // vertical pass, channel-fastest stores
        const int c = i % C, lx = (i / C) % UTX, ly = i / (C * UTX);
        const int Y = Y0 + ly, X = X0 + lx;
        if (Y < Hf && X < Wf) {
            const int qy = Y / R, py = Y - qy * R;
            const float* col = hz + (((qy + ct.off[py] - 1 - ly0) * UTX) + lx) * C + c;
            float v = __fmul_rn(col[0], ct.w[py][0]);
            v = __fadd_rn(v, __fmul_rn(col[UTX * C], ct.w[py][1]));
            v = __fadd_rn(v, __fmul_rn(col[2 * UTX * C], ct.w[py][2]));
            v = __fadd_rn(v, __fmul_rn(col[3 * UTX * C], ct.w[py][3]));
            dst[(((int64_t)n * Hf + Y) * Wf + X) * C + c] = v;
        }
    }
}
hipError_t launch_upsample(const MapView& src, int N, int C, int ratio, float* dst, hipStream_t s, const Tuning* tune) {
    const int Hf = src.h * ratio, Wf = src.w * ratio;
    const Tuning& T = tune ? *tune : default_tuning();               // upsample_tiled 0 (LWP_UPSAMPLE_TILED=0): the per-element kernel (A/B)
    if ((ratio == 4 || ratio == 8) && T.upsample_tiled != 0) {
        const int LH = UTY / ratio + 4, LW = UTX / ratio + 4;
        const size_t lds = (size_t)(LH * LW + LH * UTX) * C * sizeof(float);
        if (lds <= 64 * 1024) {
            const int tiles = ((Wf + UTX - 1) / UTX) * ((Hf + UTY - 1) / UTY);
            if (ratio == 4) hipLaunchKernelGGL(upsample_tiled_kernel<4>, dim3(tiles, N), dim3(256), lds, s, src, C, dst);
            else hipLaunchKernelGGL(upsample_tiled_kernel<8>, dim3(tiles, N), dim3(256), lds, s, src, C, dst);
            return hipGetLastError();
        }
    }
    const int64_t total = (int64_t)N * Hf * Wf * C;
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, N, C, ratio, dst);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ generic cubic resize
// cv2.resize(map, (dst_w, dst_h), INTER_CUBIC) on a cropped float32 HWC map, accumulated as avg + m / n
// (val.py:99-101,106-108).  Per-axis tables (4 clamped source indices + 4 float32 weights per destination index) are
// built on the host with the same arithmetic as the oracle; horizontal pass then vertical pass, left-to-right sums.
void build_resize_table(int n_src, int n_dst, std::vector<int>& idx, std::vector<float>& w) {
    const double inv = (double)n_dst / (double)n_src;
    const double scale = 1.0 / inv;
    idx.resize((size_t)n_dst * 4);
    w.resize((size_t)n_dst * 4);
    for (int d = 0; d < n_dst; ++d) {
        const float fx = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)floorf(fx);
        volatile float frac = fx - (float)s;
        for (int k = 0; k < 4; ++k) {
            int j = s - 1 + k;
            idx[(size_t)d * 4 + k] = j < 0 ? 0 : (j > n_src - 1 ? n_src - 1 : j);
        }
        cubic_coeffs_host(frac, &w[(size_t)d * 4]);
    }
}

__global__ void __launch_bounds__(256) resize_accum_kernel(const float* src, int64_t src_frame, int Ws, int C, int crop_top, int crop_left,
                                                            const int* xi, const float* xw, const int* yi, const float* yw,
                                                            int dst_h, int dst_w, float divisor, int init, float* accum) {
    const int64_t total = (int64_t)dst_h * dst_w * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    src += (int64_t)blockIdx.y * src_frame;          // one grid row per frame
    accum += (int64_t)blockIdx.y * total;
    const int c = (int)(idx % C);
    const int x = (int)((idx / C) % dst_w);
    const int y = (int)(idx / ((int64_t)C * dst_w));
    int64_t xo[4];
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { xo[j] = (int64_t)(crop_left + xi[x * 4 + j]) * C + c; wx[j] = xw[x * 4 + j]; }
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* row = src + (int64_t)(crop_top + yi[y * 4 + k]) * Ws * C;
        float a = __fmul_rn(row[xo[0]], wx[0]);
        a = __fadd_rn(a, __fmul_rn(row[xo[1]], wx[1]));
        a = __fadd_rn(a, __fmul_rn(row[xo[2]], wx[2]));
        a = __fadd_rn(a, __fmul_rn(row[xo[3]], wx[3]));
        const float t = __fmul_rn(a, yw[y * 4 + k]);
        o = k == 0 ? t : __fadd_rn(o, t);
    }
    // init: the accumulator starts at zero (val.py:86-87) — the add is kept so that the bits are those of 0 + m / n
    accum[idx] = __fadd_rn(init ? 0.f : accum[idx], __fdiv_rn(o, divisor));
}
hipError_t launch_resize_accum(const float* src, int N, int Hs, int Ws, int C, int crop_top, int crop_left, const int* xi, const float* xw,
                               const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum, hipStream_t s) {
    const int64_t total = (int64_t)dst_h * dst_w * C;
    hipLaunchKernelGGL(resize_accum_kernel, dim3((unsigned)((total + 255) / 256), N), dim3(256), 0, s, src, (int64_t)Hs * Ws * C, Ws, C,
                       crop_top, crop_left, xi, xw, yi, yw, dst_h, dst_w, divisor, init, accum);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fused multi-scale step
// val.py:96-108 for one scale WITHOUT the x8 map in memory: up-sample (cv2 cubic, x R) + crop + cubic resize to the frame size +
// avg += m / n in ONE kernel.  The two-kernel form writes the up-sampled map (2.6 GB for 32 x 552 x 984 x 38 floats) and gathers
// it back 16 taps per output element: 15.6 % of the config-4 step at 1.3-2.9 TB/s.  Here a workgroup owns MS_TY x MS_TX output
// pixels x CG channels and walks the four separable passes through LDS:
//   lo  [LH][LW][CG]   the low-resolution patch (replicate border), read once
//   hz  [LH][UW][CG]   up-sampling, horizontal pass  (sample_map's products and left-to-right sums)
//   up  [UH][UW][CG]   up-sampling, vertical pass    = the x R map on the rows / columns this tile's taps touch
//   h2  [UH][TX][CG]   resize, horizontal pass       (resize_accum_kernel's products and sums; aliases lo / hz)
//   out                resize, vertical pass + accumulate, channel-fastest stores
// The value of a (row, column) pair of every pass does not depend on which output pixel asks for it, so the bits are those of the
// two-kernel form (tested against it and against the oracle).  HBM traffic: the accumulator's read + write.
constexpr int MS_TY = 8;
// K1 / K2: (column, channel) items per thread and row of the x R passes / of the output passes.  The tile width MS_TX is a run-time
// value chosen per geometry so that MS_TX * CG and the tile's x R extent * CG sit just under a multiple of 256 (13 columns at scale
// 1.5: 247 and 456 items, K = 1 / 2 at 96 % / 89 % of the lanes — 16 columns gave 304 and 532: K = 2 / 3 at 59 % / 69 %).
template <int R, int CG, int K1, int K2>
__global__ void __launch_bounds__(256) multiscale_fused_kernel(MapView src, int C, int MS_TX, int crop_top, int crop_left, const int* __restrict__ xi,
                                                               const float* __restrict__ xw, const int* __restrict__ yi, const float* __restrict__ yw,
                                                               int dst_h, int dst_w, float divisor, int init, float* __restrict__ accum,
                                                               int lo_hz_floats) {
    extern __shared__ float msm[];
    const CubicTable& ct = g_cubic[R == 4 ? 0 : 1];
    const int tiles_x = (dst_w + MS_TX - 1) / MS_TX;
    const int x0 = (blockIdx.x % tiles_x) * MS_TX, y0 = (blockIdx.x / tiles_x) * MS_TY;
    const int x1 = min(x0 + MS_TX, dst_w), y1 = min(y0 + MS_TY, dst_h);
    const int c0 = blockIdx.y * CG, n = blockIdx.z;
    const int tid = threadIdx.x;
    // rows / columns of the x R map this tile's taps touch (the tables hold clamped, non-decreasing indices into the crop)
    const int ua = crop_top + yi[y0 * 4], ub = crop_top + yi[(y1 - 1) * 4 + 3];
    const int va = crop_left + xi[x0 * 4], vb = crop_left + xi[(x1 - 1) * 4 + 3];
    const int UH = ub - ua + 1, UW = vb - va + 1;
    const int la = ua / R - 2, ka = va / R - 2;                        // first low-resolution row / column (taps q + off - 1 .. + 2, off in {-1, 0})
    const int LH = ub / R + 2 - la + 1, LW = vb / R + 2 - ka + 1;
    float* lo = msm;                                                   // [LH][LW][CG]
    float* hz = lo + LH * LW * CG;                                     // [LH][UW][CG]
    float* up = msm + lo_hz_floats;                                    // [UH][UW][CG]
    float* h2 = msm;                                                   // [UH][x1 - x0][CG]  (lo / hz are dead by then)
    const float* base = src.base + (int64_t)n * src.ns + (int64_t)c0 * src.cs;
    // the accumulator values this thread will update are requested NOW: their latency hides behind the four passes (at the
    // magnifying scales a workgroup's passes are short and the read-modify-write at the end was an exposed round trip)
    const int W2 = (x1 - x0) * CG;
    int xl2[K2], c2[K2];
    float accv[MS_TY][K2];
#pragma unroll
    for (int k = 0; k < K2; ++k) {
        const int i = tid + 256 * k, ic = i < W2 ? i : 0;
        xl2[k] = ic / CG; c2[k] = ic - xl2[k] * CG;
    }
    int off2[K2][4];                                                   // taps of the resize's horizontal pass: requested now, used after three passes
    float w2t[K2][4];
#pragma unroll
    for (int k = 0; k < K2; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            off2[k][t] = xi[(x0 + xl2[k]) * 4 + t];
            w2t[k][t] = xw[(x0 + xl2[k]) * 4 + t];
        }
#pragma unroll
    for (int r = 0; r < MS_TY; ++r)
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int y = y0 + r;
            accv[r][k] = (!init && y < y1 && tid + 256 * k < W2) ? accum[(((int64_t)n * dst_h + y) * dst_w + x0 + xl2[k]) * C + c0 + c2[k]] : 0.f;
        }
    if (tid < LH * LW) {                                               // lanes along x: contiguous runs of the channel planes
        const int j = tid / LW, k = tid - j * LW;
        const float* p = base + (int64_t)clampi(la + j, 0, src.h - 1) * src.ys + (int64_t)clampi(ka + k, 0, src.w - 1) * src.xs;
#pragma unroll
        for (int c = 0; c < CG; ++c) lo[tid * CG + c] = p[(int64_t)c * src.cs];
    }
    __syncthreads();
    // A thread's items of a row — (column, channel) pairs tid, tid + 256, ... — are the same for every row of a pass, so their tap
    // offsets and weights are computed ONCE per pass (the first version redid the index arithmetic and the table loads per row and
    // item: ~300 instructions per output element; ~170 now, most of them the up-sampled values of the tile's halo).
    const int W1 = UW * CG;
    {                                                                  // up-sampling, horizontal
        int off[K1];
        float w[K1][4];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const int i = tid + 256 * k, ic = i < W1 ? i : 0;
            const int u = ic / CG, c = ic - u * CG;
            const int X = va + u, qx = X / R, px = X - qx * R;
            off[k] = (qx + ct.off[px] - 1 - ka) * CG + c;
#pragma unroll
            for (int t = 0; t < 4; ++t) w[k][t] = ct.w[px][t];
        }
        for (int j = 0; j < LH; ++j) {
            const float* lrow = lo + j * LW * CG;
#pragma unroll
            for (int k = 0; k < K1; ++k) {
                const int i = tid + 256 * k;
                if (i < W1) {
                    const float* row = lrow + off[k];
                    float a = __fmul_rn(row[0], w[k][0]);
                    a = __fadd_rn(a, __fmul_rn(row[CG], w[k][1]));
                    a = __fadd_rn(a, __fmul_rn(row[2 * CG], w[k][2]));
                    a = __fadd_rn(a, __fmul_rn(row[3 * CG], w[k][3]));
                    hz[j * W1 + i] = a;
                }
            }
        }
    }
    __syncthreads();
    for (int v = 0; v < UH; ++v) {                                     // up-sampling, vertical (row and weights are wave-uniform)
        const int Y = ua + v, qy = Y / R, py = Y - qy * R;
        const float* rows = hz + (qy + ct.off[py] - 1 - la) * W1;
        const float w0 = ct.w[py][0], w1 = ct.w[py][1], w2 = ct.w[py][2], w3 = ct.w[py][3];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const int i = tid + 256 * k;
            if (i < W1) {
                float t = __fmul_rn(rows[i], w0);
                t = __fadd_rn(t, __fmul_rn(rows[W1 + i], w1));
                t = __fadd_rn(t, __fmul_rn(rows[2 * W1 + i], w2));
                t = __fadd_rn(t, __fmul_rn(rows[3 * W1 + i], w3));
                up[v * W1 + i] = t;
            }
        }
    }
    __syncthreads();
    {                                                                  // resize, horizontal
#pragma unroll
        for (int k = 0; k < K2; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) off2[k][t] = (crop_left + off2[k][t] - va) * CG + c2[k];
        for (int v = 0; v < UH; ++v) {
            const float* row = up + v * W1;
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                const int i = tid + 256 * k;
                if (i < W2) {
                    float a = __fmul_rn(row[off2[k][0]], w2t[k][0]);
                    a = __fadd_rn(a, __fmul_rn(row[off2[k][1]], w2t[k][1]));
                    a = __fadd_rn(a, __fmul_rn(row[off2[k][2]], w2t[k][2]));
                    a = __fadd_rn(a, __fmul_rn(row[off2[k][3]], w2t[k][3]));
                    h2[v * W2 + i] = a;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MS_TY; ++r) {                                  // resize, vertical + accumulate (rows and weights are wave-uniform)
        const int y = y0 + r;
        if (y >= y1) break;
        const int r0 = (crop_top + yi[y * 4] - ua) * W2, r1 = (crop_top + yi[y * 4 + 1] - ua) * W2, r2 = (crop_top + yi[y * 4 + 2] - ua) * W2, r3 = (crop_top + yi[y * 4 + 3] - ua) * W2;
        const float w0 = yw[y * 4], w1 = yw[y * 4 + 1], w2 = yw[y * 4 + 2], w3 = yw[y * 4 + 3];
        float* drow = accum + (((int64_t)n * dst_h + y) * dst_w + x0) * C + c0;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = tid + 256 * k;
            if (i < W2) {
                float o = __fmul_rn(h2[r0 + i], w0);
                o = __fadd_rn(o, __fmul_rn(h2[r1 + i], w1));
                o = __fadd_rn(o, __fmul_rn(h2[r2 + i], w2));
                o = __fadd_rn(o, __fmul_rn(h2[r3 + i], w3));
                float* dst = drow + xl2[k] * C + c2[k];
                *dst = __fadd_rn(accv[r][k], __fdiv_rn(o, divisor));       // first scale: 0 + m / n (val.py:86-87), the same bits
            }
        }
    }
}
// Four channels per lane: the same four passes on 19-channel groups padded to 20 (five float4 per pixel in LDS; the pad lane
// computes on zeros and is never stored).  One ds_read_b128 / ds_write_b128 and one address per FOUR values: ~9 instructions per
// value instead of ~14.  A row of a pass has columns x 5 items, so 256 threads cover 256 / items ROWS per iteration (the thread's
// item — column, quad — stays fixed; its row advances); per-row taps and weights (they differ between the lanes of an iteration
// now) come from small LDS tables.  Every component goes through the same __fmul_rn / __fadd_rn sequence as the scalar kernel:
// bit-identical (tested against it, the two-kernel form and the oracle).
typedef float ms_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ms_f4 ms_mul4(ms_f4 a, float w) { return ms_f4{__fmul_rn(a.x, w), __fmul_rn(a.y, w), __fmul_rn(a.z, w), __fmul_rn(a.w, w)}; }
__device__ __forceinline__ ms_f4 ms_mad4(ms_f4 acc, ms_f4 a, float w) {
    return ms_f4{__fadd_rn(acc.x, __fmul_rn(a.x, w)), __fadd_rn(acc.y, __fmul_rn(a.y, w)), __fadd_rn(acc.z, __fmul_rn(a.z, w)), __fadd_rn(acc.w, __fmul_rn(a.w, w))};
}
template <int R>
__global__ void __launch_bounds__(256) multiscale_fused_v4_kernel(MapView src, int C, int MS_TX, int crop_top, int crop_left, const int* __restrict__ xi,
                                                                  const float* __restrict__ xw, const int* __restrict__ yi, const float* __restrict__ yw,
                                                                  int dst_h, int dst_w, float divisor, int init, float* __restrict__ accum,
                                                                  int lo_hz_floats) {
    constexpr int CG = 19, Q = 5, PP = 20;                             // channels of a group, quads and LDS floats per pixel
    extern __shared__ __attribute__((aligned(16))) float msv[];
    __shared__ __attribute__((aligned(16))) float s_cw[8][4];          // the up-sampling's phase weights
    __shared__ int s_coff[8];
    __shared__ __attribute__((aligned(16))) int s_yi[MS_TY][4];        // the tile's row taps of the resize
    __shared__ __attribute__((aligned(16))) float s_yw[MS_TY][4];
    const CubicTable& ct = g_cubic[R == 4 ? 0 : 1];
    const int tiles_x = (dst_w + MS_TX - 1) / MS_TX;
    const int x0 = (blockIdx.x % tiles_x) * MS_TX, y0 = (blockIdx.x / tiles_x) * MS_TY;
    const int x1 = min(x0 + MS_TX, dst_w), y1 = min(y0 + MS_TY, dst_h);
    const int n = blockIdx.z;
    const int tid = threadIdx.x;
    const int ua = crop_top + yi[y0 * 4], ub = crop_top + yi[(y1 - 1) * 4 + 3];
    const int va = crop_left + xi[x0 * 4], vb = crop_left + xi[(x1 - 1) * 4 + 3];
    const int UH = ub - ua + 1, UW = vb - va + 1;
    const int la = ua / R - 2, ka = va / R - 2;
    const int LH = ub / R + 2 - la + 1, LW = vb / R + 2 - ka + 1;
    float* lo = msv;                                                   // [LH][LW][PP]
    float* hz = lo + LH * LW * PP;                                     // [LH][UW][PP]
    float* up = msv + lo_hz_floats;                                    // [UH][UW][PP]
    float* h2 = msv;                                                   // [UH][x1 - x0][PP]  (lo / hz are dead by then)
    const int TXn = x1 - x0, TYn = y1 - y0;
    const int W1 = UW * Q, W2 = TXn * Q;                               // items of a row of the x R passes / of the output passes (<= 256: host)
    const int rp1 = 256 / W1, rp2 = 256 / W2;                          // rows per iteration
    const int rs1 = tid / W1, it1 = tid - rs1 * W1;                    // this thread's row slot and item in the x R passes
    const int rs2 = tid / W2, it2 = tid - rs2 * W2;                    // ... in the output passes
    const bool on1 = rs1 < rp1, on2 = rs2 < rp2;
    const int u1 = it1 / Q, q1 = it1 - u1 * Q;                         // column of the x R map, quad
    const int xl2 = it2 / Q, q2 = it2 - xl2 * Q;                       // output column, quad
    if (tid < 8) { s_coff[tid] = ct.off[tid]; }
    if (tid < 32) s_cw[tid >> 2][tid & 3] = ct.w[tid >> 2][tid & 3];
    if (tid >= 64 && tid < 64 + MS_TY * 4) {
        const int r = (tid - 64) >> 2, t = tid & 3, y = min(y0 + r, dst_h - 1);
        s_yi[r][t] = crop_top + yi[y * 4 + t] - ua;
        s_yw[r][t] = yw[y * 4 + t];
    }
    // the resize's column taps and the accumulator values of this thread's outputs: requested now, used after the passes
    int off2[4];
    float w2t[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        off2[t] = on2 ? xi[(x0 + xl2) * 4 + t] : 0;
        w2t[t] = on2 ? xw[(x0 + xl2) * 4 + t] : 0.f;
    }
    const int nq2 = q2 == Q - 1 ? CG - 4 * (Q - 1) : 4;               // channels of this thread's quad (the last one holds 3)
#pragma unroll
    for (int t = 0; t < 4; ++t) off2[t] = (crop_left + off2[t] - va) * PP + 4 * q2;
    // the workgroup walks ALL channel groups of its tile (two for the 38 PAF channels): one workgroup writes whole accumulator lines
    // (two workgroups per tile shared every line: 1006 against 2 x 394 us at scale 0.5) and the tables are set up once
    for (int c0 = 0; c0 < C; c0 += CG) {
    if (c0) __syncthreads();                                           // the previous group's passes are done with the LDS
    constexpr int MS_RPT = MS_TY / 2;                                  // output rows per thread: the tile's items fill at most half the threads (host)
    ms_f4 accv[MS_RPT];
#pragma unroll
    for (int r = 0; r < MS_RPT; ++r) {
        const int yr = rs2 + r * rp2;
        accv[r] = ms_f4{0.f, 0.f, 0.f, 0.f};
        if (!init && on2 && yr < TYn) {
            const float* a = accum + (((int64_t)n * dst_h + y0 + yr) * dst_w + x0 + xl2) * C + c0 + 4 * q2;
            if (nq2 == 4) { const F4U v = *(const F4U*)a; accv[r] = ms_f4{v.v[0], v.v[1], v.v[2], v.v[3]}; }
            else accv[r] = ms_f4{a[0], a[1], a[2], 0.f};
        }
    }
    const float* base = src.base + (int64_t)n * src.ns + (int64_t)c0 * src.cs;
    if (tid < LH * LW) {                                               // lanes along x: contiguous runs of the channel planes
        const int j = tid / LW, k = tid - j * LW;
        const float* p = base + (int64_t)clampi(la + j, 0, src.h - 1) * src.ys + (int64_t)clampi(ka + k, 0, src.w - 1) * src.xs;
        float v[PP];
#pragma unroll
        for (int c = 0; c < CG; ++c) v[c] = p[(int64_t)c * src.cs];
        v[CG] = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) *(ms_f4*)(lo + tid * PP + 4 * q) = ms_f4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
    __syncthreads();
    if (on1) {                                                         // up-sampling, horizontal
        const int X = va + u1, qx = X / R, px = X - qx * R;
        const int off = (qx + s_coff[px] - 1 - ka) * PP + 4 * q1;
        const ms_f4 w = *(const ms_f4*)s_cw[px];
        for (int j = rs1; j < LH; j += rp1) {
            const float* row = lo + j * LW * PP + off;
            ms_f4 a = ms_mul4(*(const ms_f4*)row, w.x);
            a = ms_mad4(a, *(const ms_f4*)(row + PP), w.y);
            a = ms_mad4(a, *(const ms_f4*)(row + 2 * PP), w.z);
            a = ms_mad4(a, *(const ms_f4*)(row + 3 * PP), w.w);
            *(ms_f4*)(hz + (j * UW + u1) * PP + 4 * q1) = a;
        }
    }
    __syncthreads();
    if (on1) {                                                         // up-sampling, vertical
        const int col = u1 * PP + 4 * q1;
        for (int v = rs1; v < UH; v += rp1) {
            const int Y = ua + v, qy = Y / R, py = Y - qy * R;
            const ms_f4 w = *(const ms_f4*)s_cw[py];
            const float* rows = hz + (qy + s_coff[py] - 1 - la) * UW * PP + col;
            ms_f4 t = ms_mul4(*(const ms_f4*)rows, w.x);
            t = ms_mad4(t, *(const ms_f4*)(rows + UW * PP), w.y);
            t = ms_mad4(t, *(const ms_f4*)(rows + 2 * UW * PP), w.z);
            t = ms_mad4(t, *(const ms_f4*)(rows + 3 * UW * PP), w.w);
            *(ms_f4*)(up + v * UW * PP + col) = t;
        }
    }
    __syncthreads();
    if (on2) {                                                         // resize, horizontal
        for (int v = rs2; v < UH; v += rp2) {
            const float* row = up + v * UW * PP;
            ms_f4 a = ms_mul4(*(const ms_f4*)(row + off2[0]), w2t[0]);
            a = ms_mad4(a, *(const ms_f4*)(row + off2[1]), w2t[1]);
            a = ms_mad4(a, *(const ms_f4*)(row + off2[2]), w2t[2]);
            a = ms_mad4(a, *(const ms_f4*)(row + off2[3]), w2t[3]);
            *(ms_f4*)(h2 + (v * TXn + xl2) * PP + 4 * q2) = a;
        }
    }
    __syncthreads();
    if (on2) {                                                         // resize, vertical + accumulate
        const int col = xl2 * PP + 4 * q2;
#pragma unroll
        for (int r = 0; r < MS_RPT; ++r) {
            const int yr = rs2 + r * rp2;
            if (yr >= TYn) break;
            const int4 ri = *(const int4*)s_yi[yr];
            const ms_f4 w = *(const ms_f4*)s_yw[yr];
            ms_f4 o = ms_mul4(*(const ms_f4*)(h2 + ri.x * TXn * PP + col), w.x);
            o = ms_mad4(o, *(const ms_f4*)(h2 + ri.y * TXn * PP + col), w.y);
            o = ms_mad4(o, *(const ms_f4*)(h2 + ri.z * TXn * PP + col), w.z);
            o = ms_mad4(o, *(const ms_f4*)(h2 + ri.w * TXn * PP + col), w.w);
            const ms_f4 res = ms_f4{__fadd_rn(accv[r].x, __fdiv_rn(o.x, divisor)), __fadd_rn(accv[r].y, __fdiv_rn(o.y, divisor)),
                                    __fadd_rn(accv[r].z, __fdiv_rn(o.z, divisor)), __fadd_rn(accv[r].w, __fdiv_rn(o.w, divisor))};
            float* dst = accum + (((int64_t)n * dst_h + y0 + yr) * dst_w + x0 + xl2) * C + c0 + 4 * q2;
            if (nq2 == 4) { F4U v; v.v[0] = res.x; v.v[1] = res.y; v.v[2] = res.z; v.v[3] = res.w; *(F4U*)dst = v; }
            else { dst[0] = res.x; dst[1] = res.y; dst[2] = res.z; }
        }
    }
    }   // channel groups
}

// host: x R rows / columns the largest tile of a geometry touches, for tiles of MS_TY x tx output pixels
void multiscale_fused_extent(const int* xi, const int* yi, int dst_h, int dst_w, int tx, int* uh_max, int* uw_max) {
    *uh_max = *uw_max = 0;
    for (int y0 = 0; y0 < dst_h; y0 += MS_TY) {
        const int y1 = std::min(y0 + MS_TY, dst_h), a = yi[y0 * 4], b = yi[(y1 - 1) * 4 + 3];
        *uh_max = std::max(*uh_max, b - a + 1);
    }
    for (int x0 = 0; x0 < dst_w; x0 += tx) {
        const int x1 = std::min(x0 + tx, dst_w), a = xi[x0 * 4], b = xi[(x1 - 1) * 4 + 3];
        *uw_max = std::max(*uw_max, b - a + 1);
    }
}
// host, once per geometry: the tile width with the fewest thread-iterations over the whole map (19-channel groups), its extents
void multiscale_fused_plan(const int* xi, const int* yi, int dst_h, int dst_w, int R, int* tx_best, int* uh_max, int* uw_max) {
    constexpr int CG = 19;
    double best = 1e300;
    *tx_best = 0; *uh_max = 0; *uw_max = 0;
    for (int tx = 8; tx <= 40; ++tx) {
        int uh = 0, uw = 0;
        multiscale_fused_extent(xi, yi, dst_h, dst_w, tx, &uh, &uw);
        const int k1 = (uw * CG + 255) / 256, k2 = (tx * CG + 255) / 256;
        const int lh = (uh - 1) / R + 6, lw = (uw - 1) / R + 6;
        const size_t lo_hz = (size_t)(lh * lw + lh * uw) * CG, h2 = (size_t)uh * tx * CG;
        const size_t lds = ((lo_hz > h2 ? lo_hz : h2) + (size_t)uh * uw * CG) * sizeof(float);
        if (k1 > 3 || k2 > 3 || k2 > k1 || lh * lw > 256 || lds > 64 * 1024) continue;
        // iterations of a workgroup (rows x items per thread of the four passes) + a fixed part for its barriers and fetches
        const double wg = (double)lh * k1 + (double)uh * k1 + (double)uh * k2 + (double)MS_TY * k2 + 12.0;
        // workgroups per CU by LDS: measured at scale 2, 12 columns (55 KB, two per CU) run 1835 us against 1456 for 11 columns (51 KB, three)
        const int occ = (int)((160 * 1024) / (lds + 512));
        const double cost = wg * ((dst_w + tx - 1) / tx) * (occ >= 4 ? 1.0 : occ == 3 ? 1.05 : occ == 2 ? 1.4 : 2.2);
        if (cost < best) { best = cost; *tx_best = tx; *uh_max = uh; *uw_max = uw; }
    }
}
// host, once per geometry: tile width of the four-channel kernel (five items and 20 LDS floats per pixel; a row of a pass must fit
// the 256 threads; 256 / items rows per iteration)
void multiscale_fused_plan_v4(const int* xi, const int* yi, int dst_h, int dst_w, int R, int* tx_best, int* uh_max, int* uw_max) {
    constexpr int Q = 5, PP = 20;
    double best = 1e300;
    *tx_best = 0; *uh_max = 0; *uw_max = 0;
    for (int tx = 8; tx <= 25; ++tx) {                                  // tx * 5 <= 128: two or more output rows per iteration, at most four per thread
        int uh = 0, uw = 0;
        multiscale_fused_extent(xi, yi, dst_h, dst_w, tx, &uh, &uw);
        if (uw * Q > 256) continue;
        const int lh = (uh - 1) / R + 6, lw = (uw - 1) / R + 6;
        const size_t lo_hz = (size_t)(lh * lw + lh * uw) * PP, h2 = (size_t)uh * tx * PP;
        const size_t lds = ((lo_hz > h2 ? lo_hz : h2) + (size_t)uh * uw * PP) * sizeof(float) + 1024;
        if (lh * lw > 256 || lds > 64 * 1024) continue;
        const int rp1 = 256 / (uw * Q), rp2 = 256 / (tx * Q);
        const double wg = (double)((lh + rp1 - 1) / rp1) + (double)((uh + rp1 - 1) / rp1) + (double)((uh + rp2 - 1) / rp2) + (double)((MS_TY + rp2 - 1) / rp2) + 4.0;
        const int occ = (int)((160 * 1024) / (lds + 512));
        const double cost = wg * ((dst_w + tx - 1) / tx) * (occ >= 4 ? 1.0 : occ == 3 ? 1.05 : occ == 2 ? 1.4 : 2.2);
        if (cost < best) { best = cost; *tx_best = tx; *uh_max = uh; *uw_max = uw; }
    }
}
// *used = false: the geometry does not fit; the caller takes the scalar fused kernel (or the two-kernel form)
hipError_t launch_multiscale_fused_v4(const MapView& src, int N, int C, int ratio, int crop_top, int crop_left, const int* xi, const float* xw,
                                      const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum,
                                      int tx, int uh_max, int uw_max, hipStream_t s, bool* used) {
    *used = false;
    constexpr int CG = 19, Q = 5, PP = 20;
    if ((ratio != 4 && ratio != 8) || C % CG != 0 || tx <= 0 || uh_max <= 0 || uw_max <= 0 || uw_max * Q > 256 || tx * Q > 128) return hipSuccess;
    const int lh_max = (uh_max - 1) / ratio + 6, lw_max = (uw_max - 1) / ratio + 6;
    const size_t lo_hz = (size_t)(lh_max * lw_max + lh_max * uw_max) * PP, h2 = (size_t)uh_max * tx * PP;
    const size_t a = lo_hz > h2 ? lo_hz : h2, lds = (a + (size_t)uh_max * uw_max * PP) * sizeof(float);
    if (lh_max * lw_max > 256 || lds > 96 * 1024) return hipSuccess;
    static LdsAttrOnce attr4, attr8;
    hipError_t e = ratio == 4 ? attr4.ensure((const void*)multiscale_fused_v4_kernel<4>, 96 * 1024) : attr8.ensure((const void*)multiscale_fused_v4_kernel<8>, 96 * 1024);
    if (e != hipSuccess) return e;
    *used = true;
    const dim3 grid(((dst_w + tx - 1) / tx) * ((dst_h + MS_TY - 1) / MS_TY), 1, N);       // a workgroup walks all channel groups
    if (ratio == 4) hipLaunchKernelGGL(multiscale_fused_v4_kernel<4>, grid, dim3(256), lds, s, src, C, tx, crop_top, crop_left, xi, xw, yi, yw, dst_h, dst_w, divisor, init, accum, (int)a);
    else hipLaunchKernelGGL(multiscale_fused_v4_kernel<8>, grid, dim3(256), lds, s, src, C, tx, crop_top, crop_left, xi, xw, yi, yw, dst_h, dst_w, divisor, init, accum, (int)a);
    return hipGetLastError();
}
template <int R, int CG, int K1, int K2>
static hipError_t launch_multiscale_fused_t(const MapView& src, int N, int C, int tx, int crop_top, int crop_left, const int* xi, const float* xw,
                                            const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum,
                                            int uh_max, int uw_max, hipStream_t s) {
    // low-resolution extent of a run of U rows starting anywhere: at most (U - 1) / R + 1 distinct q, + 1 for the phase, + 4 taps
    const int lh_max = (uh_max - 1) / R + 6, lw_max = (uw_max - 1) / R + 6;
    const size_t lo_hz = (size_t)(lh_max * lw_max + lh_max * uw_max) * CG, h2 = (size_t)uh_max * tx * CG;
    const size_t a = lo_hz > h2 ? lo_hz : h2, lds = (a + (size_t)uh_max * uw_max * CG) * sizeof(float);
    static LdsAttrOnce attr;
    hipError_t e = attr.ensure((const void*)multiscale_fused_kernel<R, CG, K1, K2>, 96 * 1024);
    if (e != hipSuccess) return e;
    const dim3 grid(((dst_w + tx - 1) / tx) * ((dst_h + MS_TY - 1) / MS_TY), C / CG, N);
    hipLaunchKernelGGL((multiscale_fused_kernel<R, CG, K1, K2>), grid, dim3(256), lds, s, src, C, tx, crop_top, crop_left, xi, xw, yi, yw, dst_h, dst_w, divisor, init, accum, (int)a);
    return hipGetLastError();
}
// *used = false: the geometry does not fit (or the channel count is no multiple of 19): the caller takes the two-kernel form.
// tx / uh_max / uw_max: the geometry's plan (multiscale_fused_plan, or multiscale_fused_extent for a forced width).
hipError_t launch_multiscale_fused(const MapView& src, int N, int C, int ratio, int crop_top, int crop_left, const int* xi, const float* xw,
                                   const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum,
                                   int tx, int uh_max, int uw_max, hipStream_t s, bool* used) {
    *used = false;
    constexpr int CG = 19;
    if ((ratio != 4 && ratio != 8) || C % CG != 0 || tx <= 0 || uh_max <= 0 || uw_max <= 0) return hipSuccess;
    const int k1 = (uw_max * CG + 255) / 256, k2 = (tx * CG + 255) / 256;
    const int lh_max = (uh_max - 1) / ratio + 6, lw_max = (uw_max - 1) / ratio + 6;
    const size_t lo_hz = (size_t)(lh_max * lw_max + lh_max * uw_max) * CG, h2 = (size_t)uh_max * tx * CG;
    if (k1 > 3 || k2 > k1 || lh_max * lw_max > 256 || ((lo_hz > h2 ? lo_hz : h2) + (size_t)uh_max * uw_max * CG) * sizeof(float) > 96 * 1024) return hipSuccess;
    *used = true;
#define MSF(R_, K1_, K2_) launch_multiscale_fused_t<R_, CG, K1_, K2_>(src, N, C, tx, crop_top, crop_left, xi, xw, yi, yw, dst_h, dst_w, divisor, init, accum, uh_max, uw_max, s)
#define MSF_R(K1_, K2_) if (k1 == K1_ && k2 == K2_) return ratio == 4 ? MSF(4, K1_, K2_) : MSF(8, K1_, K2_);
    MSF_R(1, 1) MSF_R(2, 1) MSF_R(2, 2) MSF_R(3, 1) MSF_R(3, 2) MSF_R(3, 3)
#undef MSF_R
#undef MSF
    *used = false;
    return hipSuccess;
}

__global__ void __launch_bounds__(256) threshold_kernel(float* m, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && m[i] < 0.1f) m[i] = 0.f;     // keypoints.py:17 (NaN stays NaN)
}
hipError_t launch_threshold_inplace(float* map, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, map, n);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) reset_ws_kernel(int N, PostWorkspace ws) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * 18) { ws.peak_count[i] = 0; ws.kpt_count[i] = 0; }
    if (i < N * 19) { ws.conn_count[i] = 0; ws.sel_count[i] = 0; }
    if (i < N) {
        ws.n_entries[i] = 0;
        ws.flags[i * 4 + 0] = 0ull; ws.flags[i * 4 + 1] = ~0ull; ws.flags[i * 4 + 2] = ~0ull; ws.flags[i * 4 + 3] = 0ull;
    }
}
hipError_t launch_reset_ws(int N, PostWorkspace& ws, hipStream_t s) {
    hipLaunchKernelGGL(reset_ws_kernel, dim3((N * 19 + 255) / 256), dim3(256), 0, s, N, ws);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ u8 pre-processing
// demo.py:59-64 on the device: cv2.resize(img, fx=fy=scale, INTER_CUBIC) of the uint8 HWC frame with OpenCV's
// fixed-point path (4 taps per axis, float32 cubic coefficients (A = -0.75) scaled by 2048 and rounded to short,
// horizontal then vertical integer sums, (v + 2^21) >> 22, saturate), then normalize (val.py:30-33: float64
// (u8 - mean) * scale) and pad_width (val.py:36-49: constant border) and HWC -> CHW float32 — one kernel, one
// thread per output pixel.  Integer arithmetic, so the summation order is immaterial.
void build_resize_table_u8(int n_src, int n_dst, double inv_scale, std::vector<int>& idx, std::vector<int>& w) {
    const double scale = 1.0 / inv_scale;
    idx.resize((size_t)n_dst * 4);
    w.resize((size_t)n_dst * 4);
    for (int d = 0; d < n_dst; ++d) {
        const float fx = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)floorf(fx);
        volatile float frac = fx - (float)s;
        float c[4];
        cubic_coeffs_host(frac, c);
        for (int k = 0; k < 4; ++k) {
            int j = s - 1 + k;
            idx[(size_t)d * 4 + k] = j < 0 ? 0 : (j > n_src - 1 ? n_src - 1 : j);
            volatile float scaled = c[k] * 2048.f;
            double r = nearbyint((double)scaled);                    // round half to even, like np.rint / cvRound
            r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
            w[(size_t)d * 4 + k] = (int)r;
        }
    }
}

__global__ void __launch_bounds__(256) preprocess_u8_kernel(PreprocParams p) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= p.Wp) return;
    const int yy = y - p.top, xx = x - p.left;
    const int64_t plane = (int64_t)p.Hp * p.Wp;
    float* o = p.out + (int64_t)y * p.Wp + x;
    if (yy < 0 || yy >= p.dh || xx < 0 || xx >= p.dw) {
        o[0] = p.pad_value[0]; o[plane] = p.pad_value[1]; o[2 * plane] = p.pad_value[2];
        return;
    }
    int xo[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xo[k] = p.xi[xx * 4 + k] * 3; wx[k] = p.xw[xx * 4 + k]; }
    long long acc[3] = {0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const unsigned char* row = p.src + (int64_t)p.yi[yy * 4 + ky] * p.Ws * 3;
        const long long wy = p.yw[yy * 4 + ky];
        int t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const unsigned char* q = row + xo[kx];
            t0 += (int)q[0] * wx[kx]; t1 += (int)q[1] * wx[kx]; t2 += (int)q[2] * wx[kx];
        }
        acc[0] += t0 * wy; acc[1] += t1 * wy; acc[2] += t2 * wy;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        long long v = (acc[c] + (1ll << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[c * plane] = (float)__dmul_rn(__dsub_rn((double)v, p.mean[c]), p.scale);
    }
}
hipError_t launch_preprocess_u8(const PreprocParams& p, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((p.Wp + 255) / 256, p.Hp), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ multi-scale image side
// val.py:84-93 on the device for N same-sized uint8 frames and one scale ratio.  The normalised image is float64 in the
// reference ((float32(u8) - mean) * scale with a Python tuple: NumPy promotes to float64), cv2.resize of a float64 image
// uses float32 cubic coefficients and float64 sums (horizontal pass, then vertical pass, left to right), pad_width fills a
// constant border and the tensor handed to the network is float32.  One thread per output pixel (3 channels); the frames
// stay in L2 (0.7 MB each), the output is written once.
void build_resize_table_ratio(int n_src, int n_dst, double ratio, std::vector<int>& idx, std::vector<float>& w) {
    const double scale = 1.0 / ratio;
    idx.resize((size_t)n_dst * 4);
    w.resize((size_t)n_dst * 4);
    for (int d = 0; d < n_dst; ++d) {
        const float fx = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)floorf(fx);
        volatile float frac = fx - (float)s;
        for (int k = 0; k < 4; ++k) {
            int j = s - 1 + k;
            idx[(size_t)d * 4 + k] = j < 0 ? 0 : (j > n_src - 1 ? n_src - 1 : j);
        }
        cubic_coeffs_host(frac, &w[(size_t)d * 4]);
    }
}

template <bool F32>      // F32: the frames are float32 (any non-uint8 image, cast like normalize's np.array(img, dtype=np.float32))
__global__ void __launch_bounds__(256) preprocess_scaled_kernel(PreScaleParams p) {
    typedef typename std::conditional<F32, float, unsigned char>::type src_t;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
    if (x >= p.Wp) return;
    const int yy = y - p.top, xx = x - p.left;
    const int64_t plane = (int64_t)p.Hp * p.Wp;
    float* o = p.out + (int64_t)n * 3 * plane + (int64_t)y * p.Wp + x;
    if (yy < 0 || yy >= p.dh || xx < 0 || xx >= p.dw) {
        o[0] = p.pad_value[0]; o[plane] = p.pad_value[1]; o[2 * plane] = p.pad_value[2];
        return;
    }
    const src_t* img = (const src_t*)p.src + (int64_t)n * p.Hs * p.Ws * 3;
    int xo[4];
    double wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xo[k] = p.xi[xx * 4 + k] * 3; wx[k] = (double)p.xw[xx * 4 + k]; }
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const src_t* row = img + (int64_t)p.yi[yy * 4 + ky] * p.Ws * 3;
        const double wy = (double)p.yw[yy * 4 + ky];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double t = 0.0;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const double v = __dmul_rn(__dsub_rn((double)row[xo[kx] + c], p.mean[c]), p.scale);   // normalize, val.py:30-33
                const double pr = __dmul_rn(v, wx[kx]);
                t = kx == 0 ? pr : __dadd_rn(t, pr);
            }
            const double q = __dmul_rn(t, wy);
            acc[c] = ky == 0 ? q : __dadd_rn(acc[c], q);
        }
    }
    o[0] = (float)acc[0]; o[plane] = (float)acc[1]; o[2 * plane] = (float)acc[2];
}
hipError_t launch_preprocess_scaled(const PreScaleParams& p, hipStream_t s) {
    if (p.src_f32) hipLaunchKernelGGL(preprocess_scaled_kernel<true>, dim3((p.Wp + 255) / 256, p.Hp, p.N), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(preprocess_scaled_kernel<false>, dim3((p.Wp + 255) / 256, p.Hp, p.N), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ result hand-over
// Writes the USED part of the result block (flags, counts, the first `total` key-point rows and the first
// n_entries pose rows of every frame) straight into pinned host memory at the same offsets, so the host parses
// one block.  This replaces a hipMemcpyAsync of the whole block: the runtime may route that through an SDMA
// queue, which stalls for milliseconds when it has to wake up, and it moves ~115 KB per frame instead of ~10 KB.
__global__ __launch_bounds__(256) void publish_results_kernel(int N, PostWorkspace ws, char* __restrict__ host) {
    const int f = blockIdx.y;
    const PostCaps& c = ws.caps;
    const size_t WN = (size_t)ws.N;
    char* q = host;
    unsigned long long* h_fl = (unsigned long long*)q; q += WN * 4 * 8;
    double2* h_k = (double2*)q; q += WN * 18 * c.max_kpts * 4 * 8;
    double2* h_e = (double2*)q; q += WN * c.max_entries * 20 * 8;
    int* h_cnt = (int*)q; q += WN * 18 * 4;
    int* h_ne = (int*)q;
    int total = 0;
    for (int t = 0; t < 18; ++t) total += ws.kpt_count[f * 18 + t];
    total = min(max(total, 0), 18 * c.max_kpts);
    const int ne = min(max(ws.n_entries[f], 0), c.max_entries);
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (tid < 4) h_fl[f * 4 + tid] = ws.flags[f * 4 + tid];
    if (tid >= 32 && tid < 32 + 18) h_cnt[f * 18 + tid - 32] = ws.kpt_count[f * 18 + tid - 32];
    if (tid == 64) h_ne[f] = ws.n_entries[f];
    const double2* sk = (const double2*)(ws.kpts_out + (size_t)f * 18 * c.max_kpts * 4);
    double2* dk = h_k + (size_t)f * 18 * c.max_kpts * 2;
    for (int i = tid; i < total * 2; i += nth) dk[i] = sk[i];
    const double2* se = (const double2*)(ws.entries + (size_t)f * c.max_entries * 20);
    double2* de = h_e + (size_t)f * c.max_entries * 10;
    for (int i = tid; i < ne * 10; i += nth) de[i] = se[i];
}
hipError_t launch_publish(int N, PostWorkspace& ws, void* host_block, hipStream_t s) {
    hipLaunchKernelGGL(publish_results_kernel, dim3(4, N), dim3(256), 0, s, N, ws, (char*)host_block);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ host frames
// Pinned (device-mapped) host memory -> device buffer, as a KERNEL: 16-byte loads, every byte crosses PCIe once.  A hipMemcpyAsync
// may be routed through an SDMA queue, which stalls for milliseconds when it has to wake up (see publish_results_kernel) — with
// three engine streams feeding frames the DMA form ran 1800-2200 frames/s from run to run.
__global__ void __launch_bounds__(256) fetch_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                        const unsigned char* src_tail, unsigned char* dst_tail, int ntail) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t k = i; k < n16; k += stride) dst[k] = src[k];
    if (i < (size_t)ntail) dst_tail[i] = src_tail[i];
}
hipError_t launch_fetch_host(const void* src_host_mapped, void* dst, size_t bytes, hipStream_t s) {
    const size_t n16 = bytes / 16;
    const int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(fetch_host_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)src_host_mapped, (uint4*)dst, n16,
                       (const unsigned char*)src_host_mapped + n16 * 16, (unsigned char*)dst + n16 * 16, ntail);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ peaks
// keypoints.py:17-30: threshold, zero border, strict > against the 4 neighbours.
// One PTH x PTW tile of full-res pixels per workgroup.  For R = 4 / 8 the up-sampled values (tile + halo 1) are
// produced separably through LDS from the (PTH/R+6) x (PTW/R+6) low-res patch: horizontal pass, then vertical
// pass — the same two-pass arithmetic as sample_map / upsample_kernel, so the bits are identical, but with ~10x
// fewer global loads and no per-pixel index arithmetic.  R = 1: the map is already full resolution.
template <int R, int PTH, int PTW>
__global__ void __launch_bounds__(256) find_peaks_kernel(MapView heat, PostWorkspace ws) {
    constexpr int LH = PTH / R + 6, LW = PTW / R + 6;
    __shared__ float lo[R == 1 ? 1 : LH][R == 1 ? 1 : LW + 1];
    __shared__ float hz[R == 1 ? 1 : LH][R == 1 ? 1 : PTW + 3];
    __shared__ float tile[PTH + 2][PTW + 3];
    const int Hf = heat.h * R, Wf = heat.w * R;
    const int tiles_x = (Wf + PTW - 1) / PTW;
    const int X0 = (blockIdx.x % tiles_x) * PTW, Y0 = (blockIdx.x / tiles_x) * PTH;
    const int t = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const float* base = heat.base + (int64_t)n * heat.ns + (int64_t)t * heat.cs;
    if (blockIdx.x == 0 && t == 0 && tid < 4)   // flags are only touched by the kernels that follow in the stream
        ws.flags[n * 4 + tid] = (tid == 1 || tid == 2) ? ~0ull : 0ull;
    if (R == 1) {
        for (int i = tid; i < (PTH + 2) * (PTW + 2); i += 256) {
            const int ly = i / (PTW + 2), lx = i % (PTW + 2);
            const int Y = Y0 + ly - 1, X = X0 + lx - 1;
            float v = 0.f;
            if (Y >= 0 && Y < Hf && X >= 0 && X < Wf) {
                v = base[(int64_t)Y * heat.ys + (int64_t)X * heat.xs];
                if (v < 0.1f) v = 0.f;
            }
            tile[ly][lx] = v;
        }
    } else {
        const CubicTable& ct = g_cubic[R == 4 ? 0 : 1];
        const int ly0 = Y0 / R - 3, lx0 = X0 / R - 3;
        for (int i = tid; i < LH * LW; i += 256) {
            const int j = i / LW, k = i % LW;
            lo[j][k] = base[(int64_t)clampi(ly0 + j, 0, heat.h - 1) * heat.ys + (int64_t)clampi(lx0 + k, 0, heat.w - 1) * heat.xs];
        }
        __syncthreads();
        for (int i = tid; i < LH * (PTW + 2); i += 256) {        // horizontal pass
            const int j = i / (PTW + 2), lx = i % (PTW + 2);
            const int X = X0 + lx - 1;
            float a = 0.f;
            if (X >= 0 && X < Wf) {
                const int qx = X / R, px = X - qx * R;
                const int c = qx + ct.off[px] - 1 - lx0;            // patch column of the first tap
                a = __fmul_rn(lo[j][c], ct.w[px][0]);
                a = __fadd_rn(a, __fmul_rn(lo[j][c + 1], ct.w[px][1]));
                a = __fadd_rn(a, __fmul_rn(lo[j][c + 2], ct.w[px][2]));
                a = __fadd_rn(a, __fmul_rn(lo[j][c + 3], ct.w[px][3]));
            }
            hz[j][lx] = a;
        }
        __syncthreads();
        for (int i = tid; i < (PTH + 2) * (PTW + 2); i += 256) {  // vertical pass + threshold
            const int ly = i / (PTW + 2), lx = i % (PTW + 2);
            const int Y = Y0 + ly - 1, X = X0 + lx - 1;
            float v = 0.f;
            if (Y >= 0 && Y < Hf && X >= 0 && X < Wf) {
                const int qy = Y / R, py = Y - qy * R;
                const int rr = qy + ct.off[py] - 1 - ly0;
                v = __fmul_rn(hz[rr][lx], ct.w[py][0]);
                v = __fadd_rn(v, __fmul_rn(hz[rr + 1][lx], ct.w[py][1]));
                v = __fadd_rn(v, __fmul_rn(hz[rr + 2][lx], ct.w[py][2]));
                v = __fadd_rn(v, __fmul_rn(hz[rr + 3][lx], ct.w[py][3]));
                if (v < 0.1f) v = 0.f;
            }
            tile[ly][lx] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < PTH * PTW; i += 256) {
        const int ly = i / PTW + 1, lx = i % PTW + 1;
        const int Y = Y0 + ly - 1, X = X0 + lx - 1;
        if (Y < Hf && X < Wf) {
            const float c = tile[ly][lx];
            if (c > tile[ly][lx + 1] && c > tile[ly][lx - 1] && c > tile[ly + 1][lx] && c > tile[ly - 1][lx]) {
                const int slot = n * gridDim.y + t;
                const int pos = atomicAdd(&ws.peak_count[slot], 1);
                if (pos < ws.caps.max_peaks) {
                    ws.peak_key[(int64_t)slot * ws.caps.max_peaks + pos] = ((uint32_t)X << 16) | (uint32_t)Y;
                    ws.peak_val[(int64_t)slot * ws.caps.max_peaks + pos] = c;
                }
            }
        }
    }
}
// Column form for the up-sampled maps at large batches (R = 4 | 8): tile = 32 rows x 62 columns, lane = one full-resolution
// COLUMN X0 - 1 + lane (lanes 0 and 63 are the halo columns), wave w = output rows 8w .. 8w + 7 of the tile.  After the
// horizontal pass (as above, through LDS) a thread reads its column's 5-6 horizontal values ONCE and keeps the ten up-sampled
// values of its rows (8 + the row above and below) in registers: up / down neighbours are registers, left / right ones come
// from the neighbouring lanes (DPP wave shifts) — no full-resolution tile in LDS, no per-pixel index arithmetic, two barriers
// instead of four.  The tile's first row is a multiple of R, so the phase of every row and its taps are compile-time constants.
// Same products and sums in the same order as sample_map: the same bits (peaks are appended in any order; nms_kernel sorts).
template <int R>
__global__ void __launch_bounds__(256) find_peaks_cols_kernel(MapView heat, PostWorkspace ws) {
    constexpr int PTH = 32, PTW = 62, RG = 8;
    constexpr int LH = PTH / R + 6, LW = 64 / R + 6;
    __shared__ float lo[LH][LW + 1];
    __shared__ float hz[LH][64];
    const int Hf = heat.h * R, Wf = heat.w * R;
    const int tiles_x = (Wf + PTW - 1) / PTW;
    const int X0 = (blockIdx.x % tiles_x) * PTW, Y0 = (blockIdx.x / tiles_x) * PTH;
    const int t = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = heat.base + (int64_t)n * heat.ns + (int64_t)t * heat.cs;
    if (blockIdx.x == 0 && t == 0 && tid < 4)   // flags are only touched by the kernels that follow in the stream
        ws.flags[n * 4 + tid] = (tid == 1 || tid == 2) ? ~0ull : 0ull;
    const CubicTable& ct = g_cubic[R == 4 ? 0 : 1];
    const int ly0 = Y0 / R - 3;
    const int lx0 = X0 == 0 ? -3 : (X0 - 1) / R - 2;              // first tap column of the left halo column
    for (int i = tid; i < LH * LW; i += 256) {
        const int j = i / LW, k = i % LW;
        lo[j][k] = base[(int64_t)clampi(ly0 + j, 0, heat.h - 1) * heat.ys + (int64_t)clampi(lx0 + k, 0, heat.w - 1) * heat.xs];
    }
    __syncthreads();
    const int X = X0 - 1 + lane;
    const bool x_ok = X >= 0 && X < Wf;
    {
        const int Xc = x_ok ? X : 0;
        const int qx = Xc / R, px = Xc - qx * R;
        const int c = qx + ct.off[px] - 1 - lx0;                   // patch column of the first tap
        const float w0 = ct.w[px][0], w1 = ct.w[px][1], w2 = ct.w[px][2], w3 = ct.w[px][3];
        for (int j = wave; j < LH; j += 4) {                       // horizontal pass
            float a = __fmul_rn(lo[j][c], w0);
            a = __fadd_rn(a, __fmul_rn(lo[j][c + 1], w1));
            a = __fadd_rn(a, __fmul_rn(lo[j][c + 2], w2));
            a = __fadd_rn(a, __fmul_rn(lo[j][c + 3], w3));
            hz[j][lane] = x_ok ? a : 0.f;
        }
    }
    __syncthreads();
    // rows Y0 + 8 wave - 1 + k, k = 0 .. 9: phase py(k) = (k - 1) mod R, low-res row (k - 1) div R, first tap row idx(k) relative to
    // the patch row of the wave's first output row
    auto fdiv = [](int a, int b) constexpr { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    auto idx = [&](int k) constexpr { const int q = fdiv(k - 1, R), py = k - 1 - q * R; return q + (py < R / 2 ? -1 : 0) - 1; };
    constexpr int IMIN = -2;                                       // idx(0): the row above the first one (phase R - 1, tap rows -2 .. 1)
    const int IMAX = idx(RG + 1);
    constexpr int NV = (R == 4 ? 0 : -1) - IMIN + 4;               // idx(9) = 0 (R = 4) | -1 (R = 8)
    (void)IMAX;
    float hv[NV];
    const int r0 = 3 + wave * (RG / R) + IMIN;
#pragma unroll
    for (int j = 0; j < NV; ++j) hv[j] = hz[r0 + j][lane];
    float val[RG + 2];
    const int Yw = Y0 + wave * RG - 1;
#pragma unroll
    for (int k = 0; k < RG + 2; ++k) {
        const int q = fdiv(k - 1, R), py = k - 1 - q * R;
        const int o = q + (py < R / 2 ? -1 : 0) - 1 - IMIN;       // compile-time
        float v = __fmul_rn(hv[o], ct.w[py][0]);
        v = __fadd_rn(v, __fmul_rn(hv[o + 1], ct.w[py][1]));
        v = __fadd_rn(v, __fmul_rn(hv[o + 2], ct.w[py][2]));
        v = __fadd_rn(v, __fmul_rn(hv[o + 3], ct.w[py][3]));
        const int Y = Yw + k;
        val[k] = (x_ok && Y >= 0 && Y < Hf && !(v < 0.1f)) ? v : 0.f;
    }
    const int slot = n * gridDim.y + t;
#pragma unroll
    for (int k = 1; k <= RG; ++k) {
        const float c = val[k];
        const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138, 0xf, 0xf, false));    // wave_shr:1
        const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x130, 0xf, 0xf, false));   // wave_shl:1
        const int Y = Yw + k;
        if (lane >= 1 && lane <= PTW && X < Wf && Y < Hf && c > right && c > left && c > val[k + 1] && c > val[k - 1]) {
            const int pos = atomicAdd(&ws.peak_count[slot], 1);
            if (pos < ws.caps.max_peaks) {
                ws.peak_key[(int64_t)slot * ws.caps.max_peaks + pos] = ((uint32_t)X << 16) | (uint32_t)Y;
                ws.peak_val[(int64_t)slot * ws.caps.max_peaks + pos] = c;
            }
        }
    }
}
template <int R>
static hipError_t launch_find_peaks_cols(const MapView& heat, int N, int ntypes, PostWorkspace& ws, hipStream_t s) {
    const int Hf = heat.h * R, Wf = heat.w * R;
    const int tiles = ((Wf + 61) / 62) * ((Hf + 31) / 32);
    hipLaunchKernelGGL((find_peaks_cols_kernel<R>), dim3(tiles, ntypes, N), dim3(256), 0, s, heat, ws);
    return hipGetLastError();
}

template <int PTH, int PTW>
static hipError_t launch_find_peaks_t(const MapView& heat, int N, int ntypes, int ratio, PostWorkspace& ws, hipStream_t s) {
    const int Hf = heat.h * ratio, Wf = heat.w * ratio;
    const int tiles = ((Wf + PTW - 1) / PTW) * ((Hf + PTH - 1) / PTH);
    const dim3 grid(tiles, ntypes, N);
    if (ratio == 1) hipLaunchKernelGGL((find_peaks_kernel<1, PTH, PTW>), grid, dim3(256), 0, s, heat, ws);
    else if (ratio == 4) hipLaunchKernelGGL((find_peaks_kernel<4, PTH, PTW>), grid, dim3(256), 0, s, heat, ws);
    else hipLaunchKernelGGL((find_peaks_kernel<8, PTH, PTW>), grid, dim3(256), 0, s, heat, ws);
    return hipGetLastError();
}
hipError_t launch_find_peaks(const MapView& heat, int N, int ntypes, int ratio, PostWorkspace& ws, hipStream_t s, const Tuning* tune) {
    // tile of full-resolution pixels per workgroup: 16 x 32 while the grid is small (one frame: 2376 workgroups at 184 x 328), 32 x 64
    // when many frames are in the batch (four barrier phases per workgroup: fewer, fatter workgroups).  Measured at batch 32
    // (round 3): 116 us (16 x 32), 84 (32 x 32), 96 (16 x 64), 81 (32 x 64); at batch 1 10.1 / 10.5 us for 16 x 32 / 32 x 64
    const Tuning& T = tune ? *tune : default_tuning();
    const int Hf = heat.h * ratio, Wf = heat.w * ratio;
    const int64_t wgs = (int64_t)((Wf + 31) / 32) * ((Hf + 15) / 16) * ntypes * N;
    int sel = ratio != 1 ? 4 : (wgs >= 16384 ? 3 : 0);            // up-sampled maps: the column form (49 / 17.8 / 8.3 us at batch 32 / 8 / 1 against 78 / 33.5 / 9.8 for the tiles)
    if (T.peak_tile >= 0) sel = T.peak_tile;                      // LWP_PEAK_TILE 0..4 (A/B; 4 = the column form)
    if (sel == 4 && ratio == 4) return launch_find_peaks_cols<4>(heat, N, ntypes, ws, s);
    if (sel == 4 && ratio == 8) return launch_find_peaks_cols<8>(heat, N, ntypes, ws, s);
    if (sel == 4) sel = 3;
    if (sel == 1) return launch_find_peaks_t<32, 32>(heat, N, ntypes, ratio, ws, s);
    if (sel == 2) return launch_find_peaks_t<16, 64>(heat, N, ntypes, ratio, ws, s);
    if (sel == 3) return launch_find_peaks_t<32, 64>(heat, N, ntypes, ratio, ws, s);
    return launch_find_peaks_t<16, 32>(heat, N, ntypes, ratio, ws, s);
}

// ------------------------------------------------------------------------------------------------ sort + NMS
// keypoints.py:30-47: candidates ordered by x then y; an unsuppressed candidate suppresses every later
// one closer than 6 px.  One wavefront per (frame, type): bitonic sort in LDS, then the greedy pass
// (sequential in i, the 64 lanes sweep the j window; sorted by x, so the window ends at x_j - x_i >= 6).
__global__ void __launch_bounds__(64) nms_kernel(int ntypes, PostWorkspace ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    const int slot = blockIdx.x;                 // n * ntypes + t
    const int lane = threadIdx.x;
    const int cap = ws.caps.max_peaks;
    int cnt = ws.peak_count[slot];
    const int frame = slot / ntypes;
    if (lane == 0 && ntypes == 18) ws.seen[frame * 37 + slot % 18] = cnt;      // debug read-out (lwp_debug_post_counts)
    if (cnt > cap) {
        if (lane == 0) atomicOr(&ws.flags[frame * 4 + 0], 1ull);
        cnt = cap;
    }
    const int kcap0 = ws.caps.max_kpts;
    if (cnt <= 64) {
        // Register form for the usual case (a few dozen candidates per type): one candidate per lane, bitonic sort by key through
        // lane exchanges, then the greedy pass as a wave-uniform loop over i — read lane i's coordinates (v_readlane), every later lane
        // tests its distance, one ballot clears the suppressed ones.  The survivors write themselves out in parallel at the rank
        // popcount(alive below me).  No LDS, no barrier: ~2 us instead of ~12 (the LDS form pays a barrier per compare-exchange pass
        // and per candidate).  Same order (keys are unique: one pixel, one candidate), same survivors, same overflow flags.
        uint32_t k = lane < cnt ? ws.peak_key[(int64_t)slot * cap + lane] : 0xFFFFFFFFu;
        float v = lane < cnt ? ws.peak_val[(int64_t)slot * cap + lane] : 0.f;
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
            for (int j = kk >> 1; j > 0; j >>= 1) {
                const uint32_t ok = (uint32_t)__shfl_xor((int)k, j);
                const float ov = __shfl_xor(v, j);
                const bool lower = (lane & j) == 0, up = (lane & kk) == 0;
                const bool take_min = lower == up;           // this lane keeps the smaller key of the pair
                if (take_min ? ok < k : ok > k) { k = ok; v = ov; }
            }
        }
        const int x = (int)(k >> 16), y = (int)(k & 0xFFFF);
        unsigned long long alive = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        for (int i = 0; i < cnt; ++i) {
            if (!((alive >> i) & 1ull)) continue;            // wave-uniform
            const int xi = __builtin_amdgcn_readlane(x, i), yi = __builtin_amdgcn_readlane(y, i);
            const int dx = x - xi, dy = y - yi;
            const unsigned long long kill = __ballot(lane > i && lane < cnt && dx * dx + dy * dy < 36);
            alive &= ~kill;
        }
        const int kept = __popcll(alive);
        const bool mine = (alive >> lane) & 1ull;
        const int rank = __popcll(alive & ((1ull << lane) - 1ull));
        if (mine && rank < kcap0) {
            ws.kpt_xy[((int64_t)slot * kcap0 + rank) * 2 + 0] = x;
            ws.kpt_xy[((int64_t)slot * kcap0 + rank) * 2 + 1] = y;
            ws.kpt_score[(int64_t)slot * kcap0 + rank] = v;
        }
        if (lane == 0) {
            if (kept > kcap0) atomicOr(&ws.flags[frame * 4 + 0], 2ull);
            ws.kpt_count[slot] = kept < kcap0 ? kept : kcap0;
        }
        return;
    }
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    uint32_t* key = (uint32_t*)nms_smem;          // [n2]
    float* val = (float*)(key + n2);              // [n2]
    int* alive = (int*)(val + n2);                // [n2]
    for (int i = lane; i < n2; i += 64) {
        const bool in = i < cnt;
        key[i] = in ? ws.peak_key[(int64_t)slot * cap + i] : 0xFFFFFFFFu;
        val[i] = in ? ws.peak_val[(int64_t)slot * cap + i] : 0.f;
        alive[i] = 1;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n2; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const uint32_t a = key[i], b = key[ixj];
                    if ((a > b) == up) {
                        key[i] = b; key[ixj] = a;
                        const float va = val[i]; val[i] = val[ixj]; val[ixj] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
    int kept = 0;
    const int kcap = ws.caps.max_kpts;
    for (int i = 0; i < cnt; ++i) {
        if (alive[i]) {                            // uniform: same LDS word for every lane
            const int xi = (int)(key[i] >> 16), yi = (int)(key[i] & 0xFFFF);
            for (int j0 = i + 1; j0 < cnt; j0 += 64) {
                if ((int)(key[j0] >> 16) - xi >= 6) break;
                const int j = j0 + lane;
                if (j < cnt) {
                    const int dx = (int)(key[j] >> 16) - xi, dy = (int)(key[j] & 0xFFFF) - yi;
                    if (dx * dx + dy * dy < 36) alive[j] = 0;
                }
            }
            if (kept < kcap) {
                if (lane == 0) {
                    ws.kpt_xy[((int64_t)slot * kcap + kept) * 2 + 0] = xi;
                    ws.kpt_xy[((int64_t)slot * kcap + kept) * 2 + 1] = yi;
                    ws.kpt_score[(int64_t)slot * kcap + kept] = val[i];
                }
            } else if (lane == 0) {
                atomicOr(&ws.flags[frame * 4 + 0], 2ull);
            }
            ++kept;
        }
        __syncthreads();
    }
    if (lane == 0) ws.kpt_count[slot] = kept < kcap ? kept : kcap;
}
hipError_t launch_nms(int N, int ntypes, int /*Hfull*/, PostWorkspace& ws, hipStream_t s) {
    int n2 = 64;
    while (n2 < ws.caps.max_peaks) n2 <<= 1;
    const size_t lds = (size_t)n2 * 12;
    hipLaunchKernelGGL(nms_kernel, dim3(N * ntypes), dim3(64), lds, s, ntypes, ws);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ pair scoring
__constant__ int c_limb_kpt[19][2] = {{1, 2}, {1, 5}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {1, 8}, {8, 9}, {9, 10}, {1, 11},
                                      {11, 12}, {12, 13}, {1, 0}, {0, 14}, {14, 16}, {0, 15}, {15, 17}, {2, 16}, {5, 17}};
__constant__ int c_limb_paf[19][2] = {{12, 13}, {20, 21}, {14, 15}, {16, 17}, {22, 23}, {24, 25}, {0, 1}, {2, 3}, {4, 5}, {6, 7},
                                      {8, 9}, {10, 11}, {28, 29}, {30, 31}, {34, 35}, {32, 33}, {36, 37}, {18, 19}, {26, 27}};

// keypoints.py:95-139 for one (frame, limb): every (i, j) candidate pair, float64 like NumPy.
// 16 lanes per pair: sub-lanes 0..9 evaluate the 10 line-integral samples, sub-lane 10 the mid-point test, in
// parallel (each is two bicubic PAF samples = 32 gathers); sub-lane 0 then adds the passed samples in k order,
// exactly the reference's sequential sum.
constexpr int SP_BLOCKS = 16;
__global__ void __launch_bounds__(256) score_pairs_kernel(MapView paf, int ratio, int demo, PostWorkspace ws) {
    const int limb = blockIdx.y, n = blockIdx.z;
    const int ta = c_limb_kpt[limb][0], tb = c_limb_kpt[limb][1];
    const int c0 = c_limb_paf[limb][0], c1 = c_limb_paf[limb][1];
    const int kcap = ws.caps.max_kpts;
    const int na = ws.kpt_count[n * 18 + ta], nb = ws.kpt_count[n * 18 + tb];
    const int npairs = na * nb;
    if (npairs == 0) return;
    const int Hf = paf.h * ratio;
    const double height_n = (double)(Hf / 2);
    const int* xa = ws.kpt_xy + (int64_t)(n * 18 + ta) * kcap * 2;
    const int* xb = ws.kpt_xy + (int64_t)(n * 18 + tb) * kcap * 2;
    unsigned long long* fl = ws.flags + n * 4;
    // 12 lanes per pair (11 used: ten samples + the mid-point), five pairs per wave, lanes 60-63 idle: 92 % of the lanes carry a
    // sample (16 lanes per pair: 69 %)
    constexpr int SP_LANES = 12, SP_PPW = 5, SP_PPB = 4 * SP_PPW;      // lanes per pair, pairs per wave, pairs per workgroup and iteration
    const int lane = threadIdx.x & 63;
    const int pw = lane / SP_LANES;                                    // pair slot within the wave (5 = the idle lanes)
    const int sub = lane - pw * SP_LANES;
    const int grp0 = (pw < SP_PPW ? pw : SP_PPW - 1) * SP_LANES;   // first lane of this pair's group (idle lanes: a valid source for the shuffles)
    // the two "first pair whose mid-point test failed / passed" orders are minima over ALL pairs of the frame: every pair used to
    // issue a 64-bit atomicMin on the frame's flag word — ~3700 same-address atomics per frame, which serialise at the memory side
    // (the dominant cost of this kernel at batch 32).  They are reduced in LDS first: one global atomic per workgroup and flag.
    __shared__ unsigned long long s_min[2];
    if (threadIdx.x < 2) s_min[threadIdx.x] = ~0ull;
    __syncthreads();
    const int npair_iters = (npairs + SP_BLOCKS * SP_PPB - 1) / (SP_BLOCKS * SP_PPB);
    for (int itp = 0; itp < npair_iters; ++itp) {                  // uniform trip count: shuffles need all lanes
        const int p = (itp * SP_BLOCKS + blockIdx.x) * SP_PPB + (threadIdx.x >> 6) * SP_PPW + pw;
        const bool valid = pw < SP_PPW && p < npairs;
        const int pp = valid ? p : 0;
        const int i = pp / nb, j = pp - i * nb;
        const int ax = xa[i * 2], ay = xa[i * 2 + 1], bx = xb[j * 2], by = xb[j * 2 + 1];
        const int dx = bx - ax, dy = by - ay;
        const double norm = sqrt((double)((long long)dx * dx + (long long)dy * dy));
        const bool live = valid && norm != 0.0;
        const double ux = (double)dx / norm, uy = (double)dy / norm;
        int px, py;
        if (sub < 10) {
            const double x = __dadd_rn(__dmul_rn(__dmul_rn(1.0 / 9.0, (double)dx), (double)sub), (double)ax);
            const double y = __dadd_rn(__dmul_rn(__dmul_rn(1.0 / 9.0, (double)dy), (double)sub), (double)ay);
            px = demo ? (int)x : (int)rint(x);
            py = demo ? (int)y : (int)rint(y);
        } else {                                                   // mid-point (keypoints.py:99-102)
            px = (int)rint(__dmul_rn((double)(ax + bx), 0.5));
            py = (int)rint(__dmul_rn((double)(ay + by), 0.5));
        }
        double sc = 0.0;
        if (live && sub <= 10)
            sc = __dadd_rn(__dmul_rn(ux, (double)sample_map_rows(paf, n, c0, py, px, ratio)),
                           __dmul_rn(uy, (double)sample_map_rows(paf, n, c1, py, px, ratio)));
        const double mid = __shfl(sc, grp0 + 10);
        double acc = 0.0;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const double s = __shfl(sc, grp0 + k);
            if (s > 0.05) { acc = __dadd_rn(acc, s); ++cnt; }
        }
        if (live && sub == 0) {
            const unsigned long long order = ((unsigned long long)limb << 32) | (unsigned)p;
            if (!(mid > -100.0)) {
                atomicMin(&s_min[0], order);
            } else {
                atomicMin(&s_min[1], order);
                double rat = cnt > 0 ? acc / (double)cnt : 0.0;
                const double pen = __dadd_rn(height_n / norm, -1.0);
                rat = __dadd_rn(rat, pen < 0.0 ? pen : 0.0);
                if (rat > 0.0 && cnt >= 9) {
                    const int slot = n * 19 + limb;
                    const int pos = atomicAdd(&ws.conn_count[slot], 1);
                    if (pos < ws.caps.max_conn) {
                        ws.conn_ij[(int64_t)slot * ws.caps.max_conn + pos] = (i << 16) | j;
                        ws.conn_ratio[(int64_t)slot * ws.caps.max_conn + pos] = rat;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_min[threadIdx.x] != ~0ull) atomicMin(&fl[1 + threadIdx.x], s_min[threadIdx.x]);
}
hipError_t launch_score_pairs(const MapView& paf, int N, int ratio, int demo, PostWorkspace& ws, hipStream_t s) {
    hipLaunchKernelGGL(score_pairs_kernel, dim3(SP_BLOCKS, 19, N), dim3(256), 0, s, paf, ratio, demo, ws);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ matching
// keypoints.py:140-155 per (frame, limb), all limbs in parallel: stable sort by descending ratio + greedy 1-1
// matching == repeatedly take the best candidate whose two end points are still free.  One wavefront per limb;
// candidates staged in LDS (global fallback beyond MATCH_LDS).  Output: the selected connections in pick order.
constexpr int MATCH_LDS = 1024;
__device__ __forceinline__ bool better(double ra, int ija, double rb, int ijb) {
    return ra > rb || (ra == rb && ija < ijb);     // descending ratio, ties in (i, j) order = stable sort
}
// 64 candidates, one per lane, into the reference's order (ratio descending, (i, j) ascending): bitonic network through lane exchanges
__device__ __forceinline__ void sort64_candidates(double& r, int& ij, int lane) {
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const double orr = __shfl_xor(r, j);
            const int oij = __shfl_xor(ij, j);
            const bool lower = (lane & j) == 0, up = (lane & kk) == 0;
            const bool want_first = lower == up;             // this lane keeps the candidate that sorts first
            const bool other_first = better(orr, oij, r, ij);
            if (want_first ? other_first : (!other_first && (orr != r || oij != ij))) { r = orr; ij = oij; }
        }
    }
}
__global__ void __launch_bounds__(64) match_kernel(PostWorkspace ws, int rounds_ok) {
    __shared__ double s_r[MATCH_LDS];
    __shared__ int s_ij[MATCH_LDS];
    extern __shared__ __attribute__((aligned(16))) int used[];     // [2 * kcap]
    const int limb = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const int slot = n * 19 + limb;
    const int kcap = ws.caps.max_kpts, ccap = ws.caps.max_conn;
    const int ta = c_limb_kpt[limb][0], tb = c_limb_kpt[limb][1];
    const int na = ws.kpt_count[n * 18 + ta], nb = ws.kpt_count[n * 18 + tb];
    int m = ws.conn_count[slot];
    if (lane == 0) ws.seen[n * 37 + 18 + limb] = m;                            // debug read-out (lwp_debug_post_counts)
    if (m > ccap) {
        if (lane == 0) atomicOr(&ws.flags[n * 4 + 0], 4ull);
        m = ccap;
    }
    const int* cij = ws.conn_ij + (int64_t)slot * ccap;
    const double* crat = ws.conn_ratio + (int64_t)slot * ccap;
    if (limb < 18) {   // this wave also writes type `limb`'s rows of all_keypoints (K,4) f64: x, y, score, id (keypoints.py:53)
        const int t = limb;
        const int c_t = ws.kpt_count[n * 18 + t];
        int below = (lane < t) ? ws.kpt_count[n * 18 + lane] : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) below += __shfl_xor(below, d);
        double* ko = ws.kpts_out + (int64_t)n * 18 * kcap * 4;
        for (int i = lane; i < c_t; i += 64) {
            const int64_t src = (int64_t)(n * 18 + t) * kcap + i;
            double* row = ko + (int64_t)(below + i) * 4;
            row[0] = (double)ws.kpt_xy[src * 2];
            row[1] = (double)ws.kpt_xy[src * 2 + 1];
            row[2] = (double)ws.kpt_score[src];
            row[3] = (double)(below + i);
        }
    }
    const int want = na < nb ? na : nb;
    int nsel = 0;
    int* sel_ij = ws.sel_ij + (int64_t)slot * kcap;
    double* sel_r = ws.sel_r + (int64_t)slot * kcap;
    if (m <= 64 && na <= 64 && nb <= 64) {
        // Register form for the usual case: one candidate per lane, bitonic sort by (ratio descending, (i, j) ascending) = the
        // reference's stable sort, then ONE wave-uniform pass over the sorted candidates with the used end points as two 64-bit
        // masks in scalar registers (the general form below finds every pick with a full scan + a 6-step f64 reduction + a barrier:
        // ~1 us per pick).  Same picks in the same order.
        int ij = lane < m ? cij[lane] : 0x7FFFFFFF;
        double r = lane < m ? crat[lane] : -1.0e300;
        sort64_candidates(r, ij, lane);
        unsigned long long used_a = 0ull, used_b = 0ull;
        int my_rank = -1;                                    // pick order of this lane's candidate
        for (int q = 0; q < m && nsel < want; ++q) {         // wave-uniform
            const int cand = __builtin_amdgcn_readlane(ij, q);
            const int ia = cand >> 16, jb = cand & 0xFFFF;
            if (((used_a >> ia) & 1ull) || ((used_b >> jb) & 1ull)) continue;
            used_a |= 1ull << ia; used_b |= 1ull << jb;
            if (lane == q) my_rank = nsel;
            ++nsel;
        }
        if (my_rank >= 0) {                                  // the picked lanes write themselves out (and gather their end points' scores) in parallel
            sel_ij[my_rank] = ij; sel_r[my_rank] = r;
            ws.sel_sa[(int64_t)slot * kcap + my_rank] = ws.kpt_score[(int64_t)(n * 18 + ta) * kcap + (ij >> 16)];
            ws.sel_sb[(int64_t)slot * kcap + my_rank] = ws.kpt_score[(int64_t)(n * 18 + tb) * kcap + (ij & 0xFFFF)];
        }
        if (lane == 0) ws.sel_count[slot] = nsel;
        return;
    }
    const bool in_lds = m <= MATCH_LDS;
    if (in_lds)
        for (int q = lane; q < m; q += 64) { s_ij[q] = cij[q]; s_r[q] = crat[q]; }
    for (int i = lane; i < 2 * kcap; i += 64) used[i] = 0;
    __syncthreads();
    if (in_lds && want <= 64 && rounds_ok) {
        // 65 .. 1024 candidates (a fifth of the limbs of the bench frames): rounds of DOMINANT candidates.  A candidate that is the
        // first in the reference's order among all live candidates at BOTH of its end points is picked by the greedy pass whatever
        // else happens (nothing ahead of it can take its end points), so every round picks all of them at once, retires the
        // candidates that share an end point with a pick, and repeats; the globally first live candidate is always dominant, so
        // the rounds end.  The picks of all rounds, sorted into the reference's order, are the greedy pass's picks in its pick
        // order.  Per round: two LDS max (ratio bits: ratios are > 0, so the IEEE bits order like the values), two LDS min on
        // (i, j) among the ties, one test — a few rounds instead of one full scan + reduction + barrier per pick.
        unsigned long long* bestA = (unsigned long long*)(used + 2 * kcap);
        unsigned long long* bestB = bestA + kcap;
        int* tieA = (int*)(bestB + kcap);
        int* tieB = tieA + kcap;
        __shared__ int s_pick[64];
        __shared__ int s_npick;
        if (lane == 0) s_npick = 0;
        unsigned alive = 0;                                      // bit t: candidate lane + 64 t
        for (int t = 0; lane + 64 * t < m; ++t) alive |= 1u << t;
        const int nt = (m + 63) >> 6;
        while (__ballot(alive != 0u) != 0ull) {
            for (int i = lane; i < kcap; i += 64) { bestA[i] = 0ull; bestB[i] = 0ull; tieA[i] = 0x7FFFFFFF; tieB[i] = 0x7FFFFFFF; }
            __syncthreads();
            for (int t = 0; t < nt; ++t)
                if ((alive >> t) & 1u) {
                    const int q = lane + 64 * t, ij = s_ij[q];
                    const unsigned long long key = (unsigned long long)__double_as_longlong(s_r[q]);
                    atomicMax(&bestA[ij >> 16], key);
                    atomicMax(&bestB[ij & 0xFFFF], key);
                }
            __syncthreads();
            for (int t = 0; t < nt; ++t)
                if ((alive >> t) & 1u) {
                    const int q = lane + 64 * t, ij = s_ij[q];
                    const unsigned long long key = (unsigned long long)__double_as_longlong(s_r[q]);
                    if (key == bestA[ij >> 16]) atomicMin(&tieA[ij >> 16], ij);
                    if (key == bestB[ij & 0xFFFF]) atomicMin(&tieB[ij & 0xFFFF], ij);
                }
            __syncthreads();
            for (int t = 0; t < nt; ++t)
                if ((alive >> t) & 1u) {
                    const int q = lane + 64 * t, ij = s_ij[q];
                    const unsigned long long key = (unsigned long long)__double_as_longlong(s_r[q]);
                    const int ia = ij >> 16, jb = ij & 0xFFFF;
                    if (key == bestA[ia] && key == bestB[jb] && ij == tieA[ia] && ij == tieB[jb]) {
                        s_pick[atomicAdd(&s_npick, 1)] = q;      // <= min(na, nb) <= 64 picks in all: one per end point
                        used[ia] = 1; used[kcap + jb] = 1;
                    }
                }
            __syncthreads();
            for (int t = 0; t < nt; ++t)
                if ((alive >> t) & 1u) {
                    const int ij = s_ij[lane + 64 * t];
                    if (used[ij >> 16] || used[kcap + (ij & 0xFFFF)]) alive &= ~(1u << t);
                }
        }
        __syncthreads();
        nsel = s_npick;
        int ij = 0x7FFFFFFF;
        double r = -1.0e300;
        if (lane < nsel) { const int q = s_pick[lane]; ij = s_ij[q]; r = s_r[q]; }
        sort64_candidates(r, ij, lane);
        if (lane < nsel) {
            sel_ij[lane] = ij; sel_r[lane] = r;
            ws.sel_sa[(int64_t)slot * kcap + lane] = ws.kpt_score[(int64_t)(n * 18 + ta) * kcap + (ij >> 16)];
            ws.sel_sb[(int64_t)slot * kcap + lane] = ws.kpt_score[(int64_t)(n * 18 + tb) * kcap + (ij & 0xFFFF)];
        }
        if (lane == 0) ws.sel_count[slot] = nsel;
        return;
    }
    while (nsel < want && m > 0) {
        double br = -1.0; int bij = 0x7FFFFFFF;
        for (int q = lane; q < m; q += 64) {
            const int ij = in_lds ? s_ij[q] : cij[q];
            if (used[ij >> 16] || used[kcap + (ij & 0xFFFF)]) continue;
            const double r = in_lds ? s_r[q] : crat[q];
            if (better(r, ij, br, bij)) { br = r; bij = ij; }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const double orr = __shfl_xor(br, d);
            const int oij = __shfl_xor(bij, d);
            if (better(orr, oij, br, bij)) { br = orr; bij = oij; }
        }
        if (bij == 0x7FFFFFFF) break;
        if (lane == 0) {
            used[bij >> 16] = 1; used[kcap + (bij & 0xFFFF)] = 1;
            sel_ij[nsel] = bij; sel_r[nsel] = br;
        }
        ++nsel;
        __syncthreads();
    }
    // scores of the two end points, gathered here (all limbs in parallel) so the sequential assembly needs no gathers
    for (int q = lane; q < nsel; q += 64) {
        const int ij = sel_ij[q];
        ws.sel_sa[(int64_t)slot * kcap + q] = ws.kpt_score[(int64_t)(n * 18 + ta) * kcap + (ij >> 16)];
        ws.sel_sb[(int64_t)slot * kcap + q] = ws.kpt_score[(int64_t)(n * 18 + tb) * kcap + (ij & 0xFFFF)];
    }
    if (lane == 0) ws.sel_count[slot] = nsel;
}
hipError_t launch_match(int N, PostWorkspace& ws, hipStream_t s) {
    // dynamic LDS: used [2][kcap] int (+ best [2][kcap] u64 + tie [2][kcap] int for the rounds form while that stays small)
    const int rounds_ok = (size_t)ws.caps.max_kpts * 32 <= 32 * 1024;
    hipLaunchKernelGGL(match_kernel, dim3(19, N), dim3(64), (size_t)ws.caps.max_kpts * (rounds_ok ? 32 : 8), s, ws, rounds_ok);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ assembly
// keypoints.py:63-92 (one-sided limbs), 159-193 (pose assembly), 195-199 (filter).  Inherently sequential over
// limbs and connections: one wavefront per frame, lanes = pose entries.  Entries live in LDS (global scratch
// when max_entries is too large for LDS).  Everything the sequential part reads — the connections picked by
// match_kernel with their scores (first ASM_STAGE per limb) and the key-point scores (first ASM_STAGE per type) —
// is staged into LDS by one round of independent loads, so no global-memory latency sits on the serial chain.
// Cost model (measured with s_memtime stamps): this kernel is ONE wavefront, so every dependent LDS round trip
// (ds_read, ds_bpermute/__shfl) costs ~120-150 cycles and nothing hides it.  The serial part therefore lives in
// registers and SGPRs only:
//   * pose entries 0..63: lane e owns entry e (18 int key-point ids, f64 score, int count); entries >= 64 spill
//     to `ext` rows in LDS / global scratch (rare: > 64 partial poses in one frame);
//   * the connections picked by match_kernel: lane q owns connection q of every limb (loaded once, up front);
//   * matching = a wave-uniform loop over the limb's connections: v_readlane broadcast + integer compare + ballot.
// The loop follows the reference's order exactly (keypoints.py:159-193): limbs in order, connections in pick order,
// every matching entry updated, a new entry appended when none matched.
template <int L> struct LimbT {
    static constexpr int kA[19] = {1, 1, 2, 3, 5, 6, 1, 8, 9, 1, 11, 12, 1, 0, 14, 0, 15, 2, 5};
    static constexpr int kB[19] = {2, 5, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 0, 14, 16, 15, 17, 16, 17};
    static constexpr int ta = kA[L], tb = kB[L];
};
struct AsmConn {           // lane q: connection q of limb l (q < 64), and key-point q's score per type
    int ij[19];
    double r[19];
    float sa[19], sb[19];
    float sc[18];
    int my_off;            // lane t < 19: first all_keypoints row of type t (lane 18: total)
    int my_nsel;           // lane l < 19: number of connections of limb l
};
struct AsmCtx {
    int id[18];
    double score;
    int count;
    int n_ent;             // uniform
    bool overflow;
    double* ext;           // [ecap - 64][20]
    int ecap, lane;
};
__device__ __forceinline__ double readlane_f64(double v, int l) {     // l must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int TA, int TB>
__device__ __forceinline__ void asm_append(AsmCtx& c, int ida, int idb, int cnt, double score) {   // uniform arguments
    if (c.n_ent >= c.ecap) { c.overflow = true; return; }
    if (c.n_ent < 64) {
        if (c.lane == c.n_ent) {
#pragma unroll
            for (int k = 0; k < 18; ++k) c.id[k] = (k == TA) ? ida : ((k == TB) ? idb : -1);
            c.score = score;
            c.count = cnt;
        }
    } else {
        if (c.lane < 20) {
            double v = -1.0;
            if (c.lane == TA) v = (double)ida;
            if (c.lane == TB) v = (double)idb;
            if (c.lane == 19) v = (double)cnt;
            if (c.lane == 18) v = score;
            c.ext[(size_t)(c.n_ent - 64) * 20 + c.lane] = v;
        }
        __syncthreads();
    }
    ++c.n_ent;
}

template <int L, bool EXT_LDS>
__device__ __forceinline__ void asm_limb(AsmCtx& c, const AsmConn& cn, const PostWorkspace& ws, int n) {
    constexpr int ta = LimbT<L>::ta, tb = LimbT<L>::tb;
    const int kcap = ws.caps.max_kpts;
    const int lane = c.lane;
    const int offa = __builtin_amdgcn_readlane(cn.my_off, ta), offb = __builtin_amdgcn_readlane(cn.my_off, tb);
    const int na = __builtin_amdgcn_readlane(cn.my_off, ta + 1) - offa, nb = __builtin_amdgcn_readlane(cn.my_off, tb + 1) - offb;
    if (na == 0 && nb == 0) return;
    const bool spilled = c.n_ent > 64;
    if (na == 0 || nb == 0) {                       // one-sided limb (keypoints.py:65-92)
        constexpr int dummy = 0; (void)dummy;
        const int cnt = na == 0 ? nb : na;
        for (int i = 0; i < cnt; ++i) {
            bool found;
            if (na == 0) {
                const int id = offb + i;
                found = lane < c.n_ent && lane < 64 && c.id[tb] == id;
                if (spilled) for (int x = 64 + lane; x < c.n_ent; x += 64) found |= c.ext[(size_t)(x - 64) * 20 + tb] == (double)id;
                if (!__any(found)) {
                    const double sc = i < 64 ? (double)readlane_f32(cn.sc[tb], i) : (double)ws.kpt_score[(int64_t)(n * 18 + tb) * kcap + i];
                    asm_append<tb, -1>(c, id, 0, 1, sc);
                }
            } else {
                const int id = offa + i;
                found = lane < c.n_ent && lane < 64 && c.id[ta] == id;
                if (spilled) for (int x = 64 + lane; x < c.n_ent; x += 64) found |= c.ext[(size_t)(x - 64) * 20 + ta] == (double)id;
                if (!__any(found)) {
                    const double sc = i < 64 ? (double)readlane_f32(cn.sc[ta], i) : (double)ws.kpt_score[(int64_t)(n * 18 + ta) * kcap + i];
                    asm_append<ta, -1>(c, id, 0, 1, sc);
                }
            }
        }
        return;
    }
    const int nsel = __builtin_amdgcn_readlane(cn.my_nsel, L);
    if (nsel == 0) return;
    const int64_t gsel = (int64_t)(n * 19 + L) * kcap;
    if (L == 0) c.n_ent = 0;                        // keypoints.py:159-165
    for (int q = 0; q < nsel; ++q) {
        int ij; double r; float sa, sb;
        if (q < 64) {
            ij = __builtin_amdgcn_readlane(cn.ij[L], q);
            r = readlane_f64(cn.r[L], q); sa = readlane_f32(cn.sa[L], q); sb = readlane_f32(cn.sb[L], q);
        } else { ij = ws.sel_ij[gsel + q]; r = ws.sel_r[gsel + q]; sa = ws.sel_sa[gsel + q]; sb = ws.sel_sb[gsel + q]; }
        const int ia = offa + (ij >> 16), ib = offb + (ij & 0xFFFF);
        const bool mine = lane < c.n_ent && lane < 64;
        if (L == 0) {
            asm_append<ta, tb>(c, ia, ib, 2, __dadd_rn(__dadd_rn((double)sa, (double)sb), r));
        } else if (L == 17 || L == 18) {            // keypoints.py:166-175: only fill a missing end point
            if (mine) {
                if (c.id[ta] == ia && c.id[tb] == -1) c.id[tb] = ib;
                else if (c.id[tb] == ib && c.id[ta] == -1) c.id[ta] = ia;
            }
            if (spilled) {
                const double iad = (double)ia, ibd = (double)ib;
                for (int x = 64 + lane; x < c.n_ent; x += 64) {
                    double* en = c.ext + (size_t)(x - 64) * 20;
                    if (en[ta] == iad && en[tb] == -1.0) en[tb] = ibd;
                    else if (en[tb] == ibd && en[ta] == -1.0) en[ta] = iad;
                }
                __syncthreads();
            }
        } else {                                    // keypoints.py:176-186
            const bool match = mine && c.id[ta] == ia;
            bool hit = __any(match);
            if (match) {
                c.id[tb] = ib;
                c.count += 1;
                c.score = __dadd_rn(c.score, __dadd_rn((double)sb, r));
            }
            if (spilled) {
                const double iad = (double)ia, ibd = (double)ib, add = __dadd_rn((double)sb, r);
                bool h2 = false;
                for (int x = 64 + lane; x < c.n_ent; x += 64) {
                    double* en = c.ext + (size_t)(x - 64) * 20;
                    if (en[ta] == iad) {
                        en[tb] = ibd;
                        en[19] = __dadd_rn(en[19], 1.0);
                        en[18] = __dadd_rn(en[18], add);
                        h2 = true;
                    }
                }
                __syncthreads();
                hit |= __any(h2);
            }
            if (!hit) asm_append<ta, tb>(c, ia, ib, 2, __dadd_rn(__dadd_rn((double)sa, (double)sb), r));   // keypoints.py:187-193
        }
    }
}

template <bool EXT_LDS>
__global__ void __launch_bounds__(64) assemble_kernel(PostWorkspace ws) {
    extern __shared__ __attribute__((aligned(16))) double asm_ext[];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int kcap = ws.caps.max_kpts, ecap = ws.caps.max_entries;
#ifdef LWP_ASM_STAMPS
    unsigned long long stamps[4];
#define STAMP(i) do { stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    STAMP(0);
    AsmConn cn;
    {   // one round of independent, unconditional loads (index clamped, never branched)
        const int li = lane < kcap ? lane : kcap - 1;
        const int my_cnt = lane < 18 ? ws.kpt_count[n * 18 + lane] : 0;
        cn.my_nsel = lane < 19 ? ws.sel_count[n * 19 + lane] : 0;
#pragma unroll
        for (int l = 0; l < 19; ++l) {
            const int64_t src = (int64_t)(n * 19 + l) * kcap + li;
            cn.ij[l] = ws.sel_ij[src]; cn.r[l] = ws.sel_r[src]; cn.sa[l] = ws.sel_sa[src]; cn.sb[l] = ws.sel_sb[src];
        }
#pragma unroll
        for (int t = 0; t < 18; ++t) cn.sc[t] = ws.kpt_score[(int64_t)(n * 18 + t) * kcap + li];
        int incl = my_cnt;                           // inclusive prefix over lanes (DPP row shifts, no LDS)
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        cn.my_off = incl - my_cnt;                   // lane 18 holds the total (its own count is 0)
    }
    STAMP(1);
    AsmCtx c;
#pragma unroll
    for (int k = 0; k < 18; ++k) c.id[k] = -1;
    c.score = 0.0; c.count = 0;
    c.n_ent = 0; c.overflow = false; c.ecap = ecap; c.lane = lane;
    c.ext = EXT_LDS ? asm_ext : ws.entries_work + (int64_t)n * ecap * 20;
#define LWP_LIMB(L) asm_limb<L, EXT_LDS>(c, cn, ws, n);
    LWP_LIMB(0) LWP_LIMB(1) LWP_LIMB(2) LWP_LIMB(3) LWP_LIMB(4) LWP_LIMB(5) LWP_LIMB(6) LWP_LIMB(7) LWP_LIMB(8) LWP_LIMB(9)
    LWP_LIMB(10) LWP_LIMB(11) LWP_LIMB(12) LWP_LIMB(13) LWP_LIMB(14) LWP_LIMB(15) LWP_LIMB(16) LWP_LIMB(17) LWP_LIMB(18)
#undef LWP_LIMB
    STAMP(2);
    // filter (keypoints.py:195-199), order preserved
    double* out = ws.entries + (int64_t)n * ecap * 20;
    int kept = 0;
    {
        const double cntd = (double)c.count;
        const bool keep = lane < c.n_ent && !(cntd < 3.0 || c.score / cntd < 0.2);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
            for (int k = 0; k < 18; ++k) out[(int64_t)pos * 20 + k] = (double)c.id[k];
            out[(int64_t)pos * 20 + 18] = c.score;
            out[(int64_t)pos * 20 + 19] = cntd;
        }
        kept = __popcll(mask);
    }
    for (int e0 = 64; e0 < c.n_ent; e0 += 64) {
        const int e = e0 + lane;
        bool keep = false;
        const double* en = c.ext + (size_t)(e - 64) * 20;
        if (e < c.n_ent) keep = !(en[19] < 3.0 || en[18] / en[19] < 0.2);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = kept + __popcll(mask & ((1ull << lane) - 1ull));
            for (int k = 0; k < 20; ++k) out[(int64_t)pos * 20 + k] = en[k];
        }
        kept += __popcll(mask);
    }
    if (lane == 0) {
        ws.n_entries[n] = kept;
        if (c.overflow) atomicOr(&ws.flags[n * 4 + 0], 4ull);
    }
    // leave the atomic append counters zeroed for the next pass (the fused pipeline has no separate reset launch)
    if (lane < 18) ws.peak_count[n * 18 + lane] = 0;
    if (lane < 19) ws.conn_count[n * 19 + lane] = 0;
    STAMP(3);
#ifdef LWP_ASM_STAMPS
    if (lane == 0) printf("assemble frame %d: stage %llu limbs %llu filter %llu cycles, n_ent %d\n", n, stamps[1] - stamps[0],
                          stamps[2] - stamps[1], stamps[3] - stamps[2], c.n_ent);
#endif
}
hipError_t launch_assemble(int N, PostWorkspace& ws, hipStream_t s) {
    const size_t ext_bytes = ws.caps.max_entries > 64 ? (size_t)(ws.caps.max_entries - 64) * 20 * 8 : 16;
    if (ext_bytes <= 60 * 1024) hipLaunchKernelGGL(assemble_kernel<true>, dim3(N), dim3(64), ext_bytes, s, ws);
    else hipLaunchKernelGGL(assemble_kernel<false>, dim3(N), dim3(64), 16, s, ws);
    return hipGetLastError();
}

}  // namespace lwp
