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

static int device_cu_count() {
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

// ---------------------------------------------------------------------------------------- fused depthwise -> pointwise
// Same two-phase structure as the f32 kernel (net_kernels.hip): phase 1 computes the workgroup's depthwise row block
// [BM][C] (f32 math on bf16 inputs) into LDS as bf16; phase 2 is a barrier-free GEMM with v_mfma_f32_16x16x32_bf16.
// A lane (i = lane&15, q = lane>>4) reads 8 consecutive k (16 B) at k = 32s + 8q: exactly the operand lane map, for
// the activation tile (from LDS) and for the fragment-packed weights [k-step][wave][tile][lane][8] (from L2) alike.
// DBG (compile time; LWP_ABLATION builds only): 1 no depthwise phase, 2 no weight stream, 4 no MFMA, 8 no epilogue
// Row padding of the depthwise tile [rows][C + DWPW_APAD] (bf16): the K loop's operand read is one ds_read_b128 per lane at
// (row = lane & 15, 16 bytes x (lane >> 4)), served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS): the 16 lanes of a group
// must land on 16 different 16-byte bank slots.  With rows of C + 8 elements (stride = 16 B x 17 mod 256) every group holds one
// pair on the same slot — a two-way conflict on EVERY operand read (rocprofv3: SQ_LDS_BANK_CONFLICT = 46-48 % of SQ_LDS_IDX_ACTIVE
// in these kernels); C + 16 (stride = 16 B x 2 mod 256) is conflict-free.
constexpr int DWPW_APAD = 16;
// (The depthwise weights' reads — two 16-byte reads per lane, lanes 32 bytes apart — are two-way conflicted too; a split
//  [low halves | high halves] layout removes that but needs a second run-time address per tap: +18 VGPRs, a wave of occupancy
//  lost on the 4-wave kernels, spills on the dilation-2 one — measured slower and dropped.)
template <int BM, int NW, int DBG = 0, int SDIL = 1, int ACT = -1>
__global__ void __launch_bounds__(NW * 64) dwpw_bf16_kernel(DwPwParams p) {
    // ACT >= 0: both activations known at compile time (ReLU for every conv_dw block) — with the run-time switch every output
    // vector walks a chain of scalar branches and the basic-block boundaries keep the loads from overlapping the arithmetic
    const int act_dw = ACT >= 0 ? ACT : p.act_dw, act_pw = ACT >= 0 ? ACT : p.act_pw;
    constexpr int NT = NW * 64;
    constexpr int RT = BM / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
    __bf16* At = (__bf16*)dsm_raw;                   // [BM][C + DWPW_APAD]
    const int ldA = p.C + DWPW_APAD;
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
    // buffer loads: descriptor and step offset are wave-uniform (SGPRs), a lane supplies one 32-bit offset register — with a
    // 64-bit per-lane address (global_load) a request costs ~3x the issue time while the matrix pipes are busy
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)pw, 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](int step, bf16x8* dst) {
        const unsigned soff = (unsigned)(step * nwt + wave_u) * 2048u;
        dst[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, soff, 0));
        dst[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + 1024, soff, 0));
    };
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) load_b(j < nsteps ? j : nsteps - 1, bw[j]);

    // ---- phase 1: a thread owns ONE 8-channel chunk (16-byte loads: the phase is bound by the NUMBER of vector-memory
    // instructions, so 8-byte loads of 4 channels cost twice as much) and walks groups of PXG consecutive output pixels; inside one
    // image row (stride 1) the group shares its 3 x (PXG + 2 dil) input window.  Loads are buffer loads: the descriptor is
    // wave-uniform, a lane supplies one 32-bit byte offset, and a tap outside the image gets offset 2^31 >= num_records, for
    // which the hardware returns zeros (no zero page, no 64-bit address arithmetic).  The depthwise weights (f32 [9][C] + bias
    // [C]) live in LDS behind the tiles: 80 registers per thread otherwise.
    constexpr int PXG = 2;                           // a 3 x 4 window of 16-byte vectors: fits the 128-VGPR cap of the 16-wave workgroups
    static_assert(BM % PXG == 0, "row block must hold whole pixel groups");
    constexpr int WC_ = NW * 32;
    const int wd_off = ((BM * (p.C + DWPW_APAD > WC_ + 8 ? p.C + DWPW_APAD : WC_ + 8) * 2) + 15) & ~15;
    float* Wd = (float*)(dsm_raw + wd_off);          // [10][C]
    for (int i = tid * 4; i < 10 * p.C; i += NT * 4) *(f32x4*)(Wd + i) = *(const f32x4*)(p.dw_w + i);
    __syncthreads();
    const int cg = p.C >> 3;
    if (!(DBG & 1)) {
        const int c = (tid % cg) * 8;
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 2), 0x00020000);
        const int pix_b = p.in_ld * 2, row_b = p.Wi * pix_b;       // bytes per input pixel / input row
        auto ldw = [&](unsigned off) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0)); };
        struct f32x8 { f32x4 lo, hi; };
        auto wtap = [&](int t) { return f32x8{*(const f32x4*)(Wd + t * p.C + c), *(const f32x4*)(Wd + t * p.C + c + 4)}; };
        auto fma8 = [&](f32x8& a, const bf16x8& x, const f32x8& w) {
            const f32x4 xl = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]}, xh = {(float)x[4], (float)x[5], (float)x[6], (float)x[7]};
            a.lo += xl * w.lo; a.hi += xh * w.hi;
        };
        auto finish = [&](const f32x8& a, int row) {
            const bf16x8 o = {(__bf16)act_f(a.lo.x, act_dw), (__bf16)act_f(a.lo.y, act_dw), (__bf16)act_f(a.lo.z, act_dw), (__bf16)act_f(a.lo.w, act_dw),
                              (__bf16)act_f(a.hi.x, act_dw), (__bf16)act_f(a.hi.y, act_dw), (__bf16)act_f(a.hi.z, act_dw), (__bf16)act_f(a.hi.w, act_dw)};
            *(bf16x8*)(At + row * ldA + c) = o;
        };
        // pixel coordinates are carried incrementally (32-bit; the host keeps every tensor below 2^31 bytes): one pair of
        // divisions per thread and tile instead of three 64-bit ones per pixel group (they were ~40 % of this phase's VALU work)
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
            if (p.stride == 1 && m + PXG <= M32 && xo + PXG <= Wo && p.dil == SDIL) {
                constexpr int NV = PXG + 2 * SDIL;
                const int base = ((img * p.Hi + yo) * p.Wi + xo) * pix_b + c * 2;
                f32x8 a[PXG];
                const f32x8 bias = wtap(9);
#pragma unroll
                for (int i = 0; i < PXG; ++i) a[i] = bias;
                {
                    bf16x8 win[3][NV];                  // the whole window in flight at once (18 loads at dilation 1), then the tap-vector FMAs
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int yy = yo + (ky - 1) * SDIL;
                        const bool rok = yy >= 0 && yy < p.Hi;
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            const int xx = xo + j - SDIL;
                            win[ky][j] = ldw((rok && xx >= 0 && xx < p.Wi) ? (unsigned)(base + (ky - 1) * SDIL * row_b + (j - SDIL) * pix_b) : 0x80000000u);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {       // tap-major: one weight vector live at a time; per pixel the order is bias, taps 0..8
                        const f32x8 w = wtap(t);
#pragma unroll
                        for (int i = 0; i < PXG; ++i) fma8(a[i], win[t / 3][i + (t % 3) * SDIL], w);
                    }
                }
#pragma unroll
                for (int i = 0; i < PXG; ++i) finish(a[i], row0 + i);
            } else {
                int xi = xo, yi = yo, im = img;
#pragma unroll 1
                for (int i = 0; i < PXG; ++i) {           // general path: groups that cross an image row or the end of the tensor, stride 2
                    const bool ok = m + i < M32;
                    const int yc = yi * p.stride, xc = xi * p.stride;
                    const int base = ((im * p.Hi + yc) * p.Wi + xc) * pix_b + c * 2;
                    bf16x8 x[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int dy = (t / 3 - 1) * p.dil, dx = (t % 3 - 1) * p.dil;
                        const bool inb = ok && yc + dy >= 0 && yc + dy < p.Hi && xc + dx >= 0 && xc + dx < p.Wi;
                        x[t] = ldw(inb ? (unsigned)(base + dy * row_b + dx * pix_b) : 0x80000000u);
                    }
                    f32x8 a = wtap(9);
#pragma unroll
                    for (int t = 0; t < 9; ++t) fma8(a, x[t], wtap(t));
                    finish(a, row0 + i);
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

    // ---- phase 2: acc[a][t] = W-tile(t) x X-tile(a):  row = channel 16t + 4q + reg,  col = pixel a*16 + i16
    f32x4 acc[RT][2];
#pragma unroll
    for (int a = 0; a < RT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const __bf16* x_lane = At + i16 * ldA + 8 * q;
    // one K step; the request for step + PF - 1 is UNCONDITIONAL (past the end: the last step again) and the ring index
    // static, so that the compiler counts the loads in flight instead of waiting for vmcnt(0)
    auto one = [&](int step, auto P_) {
        constexpr int P = decltype(P_)::value;
        const int nxt = step + PF - 1 < nsteps ? step + PF - 1 : nsteps - 1;
        if (!(DBG & 2)) load_b(nxt, bw[(P + PF - 1) % PF]);
        __builtin_amdgcn_sched_barrier(0);           // keep the requests ahead of this step's MFMAs
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            const bf16x8 xv = *(const bf16x8*)(x_lane + a * 16 * ldA + step * 32);
            if (DBG & 4) { asm volatile("" ::"v"(xv)); continue; }
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[P][0], xv, acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[P][1], xv, acc[a][1], 0, 0, 0);
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
    // epilogue
    __bf16* out = (__bf16*)p.out;
    const __bf16* res = (const __bf16*)p.res;
    if (DBG & 8) {
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < RT; ++a) sum += acc[a][0][0] + acc[a][0][1] + acc[a][0][2] + acc[a][0][3] + acc[a][1][0] + acc[a][1][1] + acc[a][1][2] + acc[a][1][3];
        if (sum == 12345.678f) out[0] = (__bf16)sum;
        return;
    }
    // The tile goes through LDS (the depthwise tile is dead once every wave has left the K loop) so that the write-out is
    // 16 bytes per lane on consecutive addresses: whole pixel rows of the workgroup's NW*32 channels.  The accumulator layout
    // itself gives 8 bytes per lane at a 2*out_ld-byte stride, i.e. partial 128-byte lines: 30 us of the 143 us of a 512 -> 512
    // block at batch 32 (ablation), 94 of the 186 us of model.1.
    constexpr int WC = NW * 32;                      // channels of this workgroup
    if (NW >= 8 ? !(p.debug & 64) : (p.debug & 64) != 0) {
        // direct form (the wide blocks; LWP_DWPWH_DEBUG bit 64 flips the choice, A/B): the permuted row packing gives a lane 8
        // consecutive channels of its pixel in its two accumulator tiles -> one 16-byte store per lane and row tile, 64 contiguous
        // bytes per pixel, no LDS round trip, no barrier.  Measured at batch 32: 512 -> 512 119.1 -> 115.5 us, 256 -> 512 76.2 -> 72.8,
        // dilation 2 129.6 -> 125.8; the narrow early blocks (64 / 128 outputs: 128-byte pixel rows) lose 1-3 % with it and keep
        // the LDS-staged form below
        const int n = blockIdx.y * WC + (tid >> 6) * 32 + 8 * q;
        const f32x4 b0 = *(const f32x4*)(p.pw_b + n), b1 = *(const f32x4*)(p.pw_b + n + 4);
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            const int64_t m = m0 + a * 16 + i16;
            f32x4 v0 = acc[a][0] + b0, v1 = acc[a][1] + b1;
            v0.x = act_f(v0.x, act_pw); v0.y = act_f(v0.y, act_pw); v0.z = act_f(v0.z, act_pw); v0.w = act_f(v0.w, act_pw);
            v1.x = act_f(v1.x, act_pw); v1.y = act_f(v1.y, act_pw); v1.z = act_f(v1.z, act_pw); v1.w = act_f(v1.w, act_pw);
            if (m < M) {
                if (res) {
                    const bf16x8 r = *(const bf16x8*)(res + m * p.res_ld + n);
                    v0.x += (float)r[0]; v0.y += (float)r[1]; v0.z += (float)r[2]; v0.w += (float)r[3];
                    v1.x += (float)r[4]; v1.y += (float)r[5]; v1.z += (float)r[6]; v1.w += (float)r[7];
                }
                const bf16x8 o = {(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
                *(bf16x8*)(out + m * p.out_ld + n) = o;
            }
        }
        return;
    }
    constexpr int OLD = WC + 8;                      // staged row stride (elements)
    __bf16* Ot = (__bf16*)dsm_raw;                   // [BM][OLD]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int nl = (tid >> 6) * 32 + 8 * q + 4 * t;              // channel within the workgroup's window (rows of a 32-channel block are packed permuted: MFMA row 4q + r of tile t <-> channel 8q + 4t + r)
        const int n = blockIdx.y * WC + nl;
        const f32x4 bias = *(const f32x4*)(p.pw_b + n);
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            const int64_t m = m0 + a * 16 + i16;
            f32x4 v = acc[a][t] + bias;
            v.x = act_f(v.x, act_pw); v.y = act_f(v.y, act_pw); v.z = act_f(v.z, act_pw); v.w = act_f(v.w, act_pw);
            if (res && m < M) {                      // residual before the (single) rounding to bf16
                const bf16x4 r = *(const bf16x4*)(res + m * p.res_ld + n);
                v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
            }
            const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *(bf16x4*)(Ot + (a * 16 + i16) * OLD + nl) = o;
        }
    }
    __syncthreads();
    constexpr int CPR = WC / 8;                      // 16-byte chunks per pixel row
    for (int ch = tid; ch < BM * CPR; ch += NT) {
        const int row = ch / CPR, col = (ch - row * CPR) * 8;
        const int64_t m = m0 + row;
        if (m < M) *(bf16x8*)(out + m * p.out_ld + blockIdx.y * WC + col) = *(const bf16x8*)(Ot + row * OLD + col);
    }
}

template <int BM, int NW, int DBG = 0, int SDIL = 1, int ACT = -1>
static hipError_t launch_dwpw_bf16_t(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    size_t lds = (size_t)BM * (p.C + DWPW_APAD) * 2;
    const size_t lds_out = (size_t)BM * (NW * 32 + 8) * 2;          // the staged output tile re-uses the space
    if (lds_out > lds) lds = lds_out;
    lds = ((lds + 15) & ~(size_t)15) + (size_t)10 * p.C * sizeof(float);   // + the depthwise weights and bias
    if (p.tune && p.tune->dwpw_lds_pad_kb > 0) lds += (size_t)p.tune->dwpw_lds_pad_kb * 1024;      // LWP_DWPW_LDS_PAD (experiments: fewer workgroups per CU)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if ((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 2 >= (1ll << 31) || (p.C & 7)) return hipErrorInvalidValue;   // 32-bit buffer offsets, 8-channel lanes
    const int nsplit = (p.cout / 32) / NW;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) {
        hipError_t e = attr.ensure((const void*)dwpw_bf16_kernel<BM, NW, DBG, SDIL, ACT>, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((dwpw_bf16_kernel<BM, NW, DBG, SDIL, ACT>), dim3((unsigned)((M + BM - 1) / BM), nsplit), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- fused depthwise -> pointwise, two half-tiles in flight
// The 256- and 512-channel blocks at large M (batch 32).  In dwpw_bf16_kernel a workgroup owns its CU (133 KB tile) and its
// phases run one after the other: depthwise (VALU + vector-memory bound, ~35 us of a 512 -> 512 layer), K loop (matrix pipe +
// LDS, ~28 us), write-out (~25 us).  Here ONE persistent 16-wave workgroup holds TWO 64-pixel half-tiles: wave group g (8 waves,
// two per SIMD) alternates between the depthwise role on its next half-tile and the K-loop + store role on its current one,
// half a period apart from the other group — so on every SIMD two waves issue VALU / vector-memory work while the other two
// issue MFMAs and LDS reads (separate pipes run concurrently across waves), and the stores of one half-tile travel while the
// other is being multiplied.  One workgroup barrier per slot; the depthwise weights are shared by both groups.
// Epilogue without LDS: the pointwise weights are packed with a row permutation inside every 32-channel block (MFMA row 4q + r
// of tile e <-> channel 8q + 4e + r), so a lane ends up with 8 CONSECUTIVE channels of a pixel in two accumulator tiles: one
// 16-byte store per lane, 64 contiguous bytes per pixel and instruction.
template <int NCT, int ACT, int SDIL>      // NCT: 16-channel column tiles per K-role wave (cout = 8 waves x 16 NCT)
__global__ void __launch_bounds__(1024) dwpw_bf16_pp_kernel(DwPwParams p, int n_half) {
    constexpr int HB = 64, RT = HB / 16, GT = 512;       // pixels per half-tile, row tiles, threads per wave group
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
    const int ldA = p.C + DWPW_APAD;
    const __bf16* in = (const __bf16*)p.in;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 3, gw = wave & 7, gt = tid & (GT - 1);
    __bf16* At = (__bf16*)dsm_raw + (size_t)g * HB * ldA;             // this group's tile [HB][C + 8]
    float* Wd = (float*)(dsm_raw + (((size_t)2 * HB * ldA * 2 + 15) & ~(size_t)15));   // [10][C], shared
    for (int i = tid * 4; i < 10 * p.C; i += 1024 * 4) *(f32x4*)(Wd + i) = *(const f32x4*)(p.dw_w + i);
    const int i16 = lane & 15, q = lane >> 4;
    const int M32 = p.N * p.Ho * p.Wo;                                // host: < 2^31
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware order: the workgroups of one XCD take neighbouring half-tiles (their input windows overlap)
        const int qq = G >> 3, rem = G & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    // half-tile of (group g, round k): 2 * (bid + k * G) + g
    const int rounds = (n_half - 2 * bid + 2 * G - 1) / (2 * G);      // rounds in which group 0 has a half-tile (>= group 1's)
    const int nslots = 2 * rounds + 1;
    const int nsteps = p.C / 32, nwt = p.cout / 32;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.pw_w, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.in_ld * 2), 0x00020000);
    const int pix_b = p.in_ld * 2, row_b = p.Wi * pix_b;
    const int cg = p.C >> 3;
    const int c = (gt % cg) * 8;
    // K role: weights of this wave's NCT column tiles; block w32 = gw * NCT / 2 + (t >> 1) of 32 channels, tile e = t & 1
    constexpr int PF = 2;
    bf16x8 bw[PF][NCT];
    auto load_b = [&](int step, bf16x8* dst) {
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const unsigned soff = (unsigned)(step * nwt + gw * (NCT / 2) + (t >> 1)) * 2048u;
            dst[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + (t & 1) * 1024, soff, 0));
        }
    };
    // run-time ablation switches (LWP_DWPWH_DEBUG, tools only; 0 in production): 1 no depthwise role, 2 no K role, 4 weights of
    // step 0 only (L1-resident), 8 no stores, 16 depthwise without global loads
    const int dbg = p.debug;
    __syncthreads();                                                  // depthwise weights are in LDS
    for (int s = 0; s < nslots; ++s) {
        const int d = s - g;                                          // wave-uniform
        if (d >= 0 && (d & 1) == 0) {
            // ---------------- depthwise role: half-tile of round d / 2 into this group's tile
            const int h = 2 * (bid + (d >> 1) * G) + g;
            if (h < n_half && !(dbg & 1)) {
                auto ldw = [&](unsigned off) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(irsrc, (dbg & 16) ? 0x80000000u : off, 0, 0)); };
                struct f32x8 { f32x4 lo, hi; };
                auto wtap = [&](int t) { return f32x8{*(const f32x4*)(Wd + t * p.C + c), *(const f32x4*)(Wd + t * p.C + c + 4)}; };
                auto fma8 = [&](f32x8& a, const bf16x8& x, const f32x8& w) {
                    const f32x4 xl = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]}, xh = {(float)x[4], (float)x[5], (float)x[6], (float)x[7]};
                    a.lo += xl * w.lo; a.hi += xh * w.hi;
                };
                auto finish = [&](const f32x8& a, int row) {
                    const bf16x8 o = {(__bf16)act_f(a.lo.x, ACT), (__bf16)act_f(a.lo.y, ACT), (__bf16)act_f(a.lo.z, ACT), (__bf16)act_f(a.lo.w, ACT),
                                      (__bf16)act_f(a.hi.x, ACT), (__bf16)act_f(a.hi.y, ACT), (__bf16)act_f(a.hi.z, ACT), (__bf16)act_f(a.hi.w, ACT)};
                    *(bf16x8*)(At + row * ldA + c) = o;
                };
                constexpr int PXG = 2;
                const int Wo = p.Wo, Ho = p.Ho;
                const int gadv = GT / cg;
                const int adv = gadv * PXG, adv_y = adv / Wo, adv_x = adv - adv_y * Wo;
                int grp = gt / cg;
                int m = h * HB + grp * PXG;
                int xo, yo, img;
                {
                    const unsigned q1 = (unsigned)m / (unsigned)Wo;
                    xo = m - (int)q1 * Wo;
                    img = (int)(q1 / (unsigned)Ho);
                    yo = (int)q1 - img * Ho;
                }
                for (; grp < HB / PXG; grp += gadv) {
                    const int row0 = grp * PXG;
                    if (m + PXG <= M32 && xo + PXG <= Wo) {
                        constexpr int NV = PXG + 2 * SDIL;
                        const int base = ((img * p.Hi + yo) * p.Wi + xo) * pix_b + c * 2;
                        f32x8 a[PXG];
                        const f32x8 bias = wtap(9);
#pragma unroll
                        for (int i = 0; i < PXG; ++i) a[i] = bias;
                        bf16x8 win[3][NV];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int yy = yo + (ky - 1) * SDIL;
                            const bool rok = yy >= 0 && yy < p.Hi;
#pragma unroll
                            for (int j = 0; j < NV; ++j) {
                                const int xx = xo + j - SDIL;
                                win[ky][j] = ldw((rok && xx >= 0 && xx < p.Wi) ? (unsigned)(base + (ky - 1) * SDIL * row_b + (j - SDIL) * pix_b) : 0x80000000u);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const f32x8 w = wtap(t);
#pragma unroll
                            for (int i = 0; i < PXG; ++i) fma8(a[i], win[t / 3][i + (t % 3) * SDIL], w);
                        }
#pragma unroll
                        for (int i = 0; i < PXG; ++i) finish(a[i], row0 + i);
                    } else {
                        int xi = xo, yi = yo, im = img;
#pragma unroll 1
                        for (int i = 0; i < PXG; ++i) {               // groups that cross an image row or the end of the tensor
                            const bool ok = m + i < M32;
                            const int base = ((im * p.Hi + yi) * p.Wi + xi) * pix_b + c * 2;
                            bf16x8 x[9];
#pragma unroll
                            for (int t = 0; t < 9; ++t) {
                                const int dy = (t / 3 - 1) * SDIL, dx = (t % 3 - 1) * SDIL;
                                const bool inb = ok && yi + dy >= 0 && yi + dy < p.Hi && xi + dx >= 0 && xi + dx < p.Wi;
                                x[t] = ldw(inb ? (unsigned)(base + dy * row_b + dx * pix_b) : 0x80000000u);
                            }
                            f32x8 a = wtap(9);
#pragma unroll
                            for (int t = 0; t < 9; ++t) fma8(a, x[t], wtap(t));
                            finish(a, row0 + i);
                            if (++xi == Wo) { xi = 0; if (++yi == Ho) { yi = 0; ++im; } }
                        }
                    }
                    m += adv;
                    xo += adv_x; yo += adv_y;
                    if (xo >= Wo) { xo -= Wo; ++yo; }
                    while (yo >= Ho) { yo -= Ho; ++img; }
                }
            }
        } else if (d >= 1) {
            // ---------------- K-loop + store role: the half-tile this group filled in the previous slot
            const int h = 2 * (bid + ((d - 1) >> 1) * G) + g;
            if (h < n_half && !(dbg & 2)) {
                load_b(0, bw[0]);
                f32x4 acc[RT][NCT];
#pragma unroll
                for (int a = 0; a < RT; ++a)
#pragma unroll
                    for (int t = 0; t < NCT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const __bf16* x_lane = At + i16 * ldA + 8 * q;
                auto one = [&](int step, auto P_) {
                    constexpr int P = decltype(P_)::value;
                    const int nxt = step + PF - 1 < nsteps ? step + PF - 1 : nsteps - 1;
                    load_b((dbg & 4) ? 0 : nxt, bw[(P + PF - 1) % PF]);   // unconditional (clamped): exact vmcnt counting
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        const bf16x8 xv = *(const bf16x8*)(x_lane + a * 16 * ldA + step * 32);
#pragma unroll
                        for (int t = 0; t < NCT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[P][t], xv, acc[a][t], 0, 0, 0);
                    }
                };
                int s0 = 0;
                for (; s0 + PF <= nsteps; s0 += PF) {
                    one(s0, std::integral_constant<int, 0>{});
                    one(s0 + 1, std::integral_constant<int, 1>{});
                }
                if (s0 < nsteps) one(s0, std::integral_constant<int, 0>{});
                // store: lane (pixel i16, q) holds channels n0 + 32 u + 8 q + 4 e + r in acc[a][2 u + e][r]
                __bf16* out = (__bf16*)p.out;
                const int n0 = gw * 16 * NCT;
#pragma unroll
                for (int u = 0; u < NCT / 2; ++u) {
                    const int n = n0 + 32 * u + 8 * q;
                    const f32x4 b0 = *(const f32x4*)(p.pw_b + n), b1 = *(const f32x4*)(p.pw_b + n + 4);
#pragma unroll
                    for (int a = 0; a < RT; ++a) {
                        const int m = h * HB + a * 16 + i16;
                        const f32x4 v0 = acc[a][2 * u] + b0, v1 = acc[a][2 * u + 1] + b1;
                        const bf16x8 o = {(__bf16)act_f(v0.x, ACT), (__bf16)act_f(v0.y, ACT), (__bf16)act_f(v0.z, ACT), (__bf16)act_f(v0.w, ACT),
                                          (__bf16)act_f(v1.x, ACT), (__bf16)act_f(v1.y, ACT), (__bf16)act_f(v1.z, ACT), (__bf16)act_f(v1.w, ACT)};
                        if (m < M32 && !(dbg & 8)) *(bf16x8*)(out + (int64_t)m * p.out_ld + n) = o;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int NCT, int ACT, int SDIL>
static hipError_t launch_dwpw_bf16_pp_t(const DwPwParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    const int n_half = (int)((M + 63) / 64);
    const size_t lds = (((size_t)2 * 64 * (p.C + DWPW_APAD) * 2 + 15) & ~(size_t)15) + (size_t)10 * p.C * sizeof(float);
    static LdsAttrOnce attr;
    hipError_t e = attr.ensure((const void*)dwpw_bf16_pp_kernel<NCT, ACT, SDIL>, 160 * 1024);
    if (e != hipSuccess) return e;
    int grid = device_cu_count();
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dwpw_pp_grid > 0) grid = T.dwpw_pp_grid;
    if (grid > (n_half + 1) / 2) grid = (n_half + 1) / 2;
    hipLaunchKernelGGL((dwpw_bf16_pp_kernel<NCT, ACT, SDIL>), dim3(grid), dim3(1024), lds, s, p, n_half);
    return hipGetLastError();
}

// two-half-tile form for the wide blocks at large M; *used = false: the caller takes the two-phase kernel
static hipError_t try_dwpw_bf16_pp(const DwPwParams& p, hipStream_t s, bool* used) {
    *used = false;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    // LWP_DWPW_PP "1": on, at every size it supports (tests, tools).  OFF by default — measured at batch 32 (round 3): 512 -> 512
    // 128.4 us against 119 for the two-phase kernel, 256 -> 512 69.9 against 76.  Run-time ablation of the 512 -> 512 layer (us):
    // depthwise role alone 56.9 (54.7 without its global loads: VALU / LDS-issue bound on 2 waves per SIMD), K role alone 79.5
    // (62 without stores: 2.5x its MFMA bound on 2 waves per SIMD; weights kept L1-resident change nothing), both 128.4 —
    // the two roles do NOT overlap: an MFMA holds its SIMD's issue port 8 of 16 cycles, every VALU instruction 2-4, and with two
    // waves per role neither hides its own latencies; four waves of one role per SIMD (the two-phase kernel) are as fast.
    if (T.dwpw_pp != 1) return hipSuccess;
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    if (p.stride != 1 || (p.dil != 1 && p.dil != 2) || p.res || p.Hi != p.Ho || p.Wi != p.Wo) return hipSuccess;
    if ((p.C != 256 && p.C != 512) || (p.cout != 256 && p.cout != 512)) return hipSuccess;
    if (!(p.act_dw == ACT_RELU && p.act_pw == ACT_RELU)) return hipSuccess;
    if ((p.out_ld & 7) || (((uintptr_t)p.out) & 15) || (p.in_ld & 7)) return hipSuccess;
    if (M >= (1ll << 31) - 128 || (int64_t)p.N * p.Hi * p.Wi * p.in_ld * 2 >= (1ll << 31)) return hipSuccess;
    if ((size_t)2 * 64 * (p.C + DWPW_APAD) * 2 + 16 + (size_t)40 * p.C > 160 * 1024) return hipSuccess;
    *used = true;
    LWP_VARIANT(p, "dwpw_bf16_pp<%d,dil=%d>", p.cout / 128, p.dil);
    DwPwParams q = p;
    q.debug = T.dwpwh_debug;
    if (p.cout == 512) return p.dil == 2 ? launch_dwpw_bf16_pp_t<4, ACT_RELU, 2>(q, s) : launch_dwpw_bf16_pp_t<4, ACT_RELU, 1>(q, s);
    return p.dil == 2 ? launch_dwpw_bf16_pp_t<2, ACT_RELU, 2>(q, s) : launch_dwpw_bf16_pp_t<2, ACT_RELU, 1>(q, s);
}

// Measured and dropped: persistent workgroups (a tile loop around the body, grid = resident workgroups): 3-10 % faster than the same
// code launched one workgroup per row block, but the loop keeps more values live across the phases (4-39 spilled VGPRs at the
// 128-VGPR cap of the 16-wave workgroups) and every layer ended 5-25 % SLOWER than the plain form below.
// Measured and dropped (round 2, batch 32, 512 -> 512, two-phase kernel 143 us): a K-streamed form — persistent workgroups over
// 8 x 16 pixel patches, the input window of every 64-channel chunk brought in by LDS-DMA (1.4x instead of 4.5x re-reads),
// depthwise conv from LDS into a double-buffered [128][64] tile, v_mfma_f32_32x32x16_bf16 with the accumulators kept across
// the chunks and a whole chunk of weights in flight.  (a) wave-specialised, 4 producer + 8 consumer waves, one barrier per
// chunk: 190 us — a single producer wave per SIMD issues one VALU instruction per 4+ cycles (2.6 us per chunk against 1.6 us
// of consumer time) and the two sides' times ADDED in every ablation; (b) symmetric, all 8 waves produce then multiply: 228 us —
// depthwise from LDS at two waves per SIMD 104 us, and the in-order vmcnt makes the wave wait for its own output stores when
// it next needs a DMA piece (63 us of "epilogue").  The depthwise conv is VALU-instruction-bound (~19 instructions per output
// element in f32 on bf16 inputs), not load-bound: the lever that remains is fewer instructions (v_dot2_f32_bf16 on tap pairs).
hipError_t launch_dwpw_bf16(const DwPwParams& p_in, hipStream_t s) {
    DwPwParams p = p_in;
    p.debug = (p.tune ? *p.tune : default_tuning()).dwpwh_debug;      // run-time A/B bits (tools): 64 = direct-store epilogue
    const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
    {
        bool used = false;
        hipError_t e = try_dwpw_tiled_bf16(p, s, &used);
        if (e != hipSuccess || used) return e;
        e = try_dwpw_bf16_pp(p, s, &used);
        if (e != hipSuccess || used) return e;
    }
    const int nw = p.cout / 32;
    int bm = 16;
    if (M / 16 >= 2048) bm = 32;
    if (M / 32 >= 2048) bm = 64;
    if (M / 64 >= 1024 && nw >= 8) bm = 128;         // >= 256 output channels: halve the per-workgroup weight stream (measured 3-8 %)
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.dwpw_bm) bm = T.dwpw_bm;                   // LWP_DWPW_BM
    while (bm > 16 && (size_t)bm * (p.C + DWPW_APAD > p.cout + 8 ? p.C + DWPW_APAD : p.cout + 8) * 2 + (size_t)40 * p.C + 16 > 160 * 1024) bm >>= 1;
    if (bm == 128 && nw < 4) bm = 64;
    // every thread must own a whole 8-channel chunk column: NW*64 threads must be a multiple of C/8 (always true here)
    LWP_VARIANT(p, "dwpw_bf16<%d,%d,dil=%d>", bm, nw, (nw == 16 && p.dil == 2 && p.stride == 1) ? 2 : 1);
#ifdef LWP_ABLATION
    const int d = T.dwpwh_debug;
#define DPH_DBG(BM_, NW_, D_) if (bm == BM_ && nw == NW_ && d == D_) return launch_dwpw_bf16_t<BM_, NW_, D_>(p, s);
#define DPH_DBGS(BM_, NW_) DPH_DBG(BM_, NW_, 1) DPH_DBG(BM_, NW_, 2) DPH_DBG(BM_, NW_, 4) DPH_DBG(BM_, NW_, 6) DPH_DBG(BM_, NW_, 7) DPH_DBG(BM_, NW_, 8) DPH_DBG(BM_, NW_, 14) DPH_DBG(BM_, NW_, 15)
    DPH_DBGS(128, 16) DPH_DBGS(64, 16) DPH_DBGS(32, 16) DPH_DBGS(64, 2) DPH_DBGS(64, 4)
#undef DPH_DBGS
#undef DPH_DBG
#endif
    // the dilation-2 layer (model.7, 512 -> 512) gets its own instantiation: carrying both shared-window forms in one kernel
    // spills 38 VGPRs at the 128-VGPR cap of the 16-wave workgroups
    const bool relu = p.act_dw == ACT_RELU && p.act_pw == ACT_RELU;
#define DPH_DIL2(BM_) if (bm == BM_ && nw == 16 && p.dil == 2 && p.stride == 1) return relu ? launch_dwpw_bf16_t<BM_, 16, 0, 2, ACT_RELU>(p, s) : launch_dwpw_bf16_t<BM_, 16, 0, 2>(p, s);
    DPH_DIL2(16) DPH_DIL2(32) DPH_DIL2(64) DPH_DIL2(128)
#undef DPH_DIL2
    const bool elu = p.act_dw == ACT_ELU && p.act_pw == ACT_ELU;       // the conv_dw_no_bn blocks of the cpm trunk
#define DPH_CASE(BM_, NW_) if (bm == BM_ && nw == NW_) return relu ? launch_dwpw_bf16_t<BM_, NW_, 0, 1, ACT_RELU>(p, s) : elu ? launch_dwpw_bf16_t<BM_, NW_, 0, 1, ACT_ELU>(p, s) : launch_dwpw_bf16_t<BM_, NW_>(p, s);
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

    // both operand streams use buffer loads (wave-uniform descriptor + step offset in SGPRs, one 32-bit offset register per
    // lane): with a 64-bit per-lane address a request costs ~3x the issue time while the matrix pipes are busy, and this kernel
    // issues one request per 16 MFMA cycles.  Pixels outside the image read offset 2^31 >= num_records -> the hardware returns 0.
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(M * p.in_ld * 2), 0x00020000);   // host: < 2^31 bytes
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, 0x7fffffff, 0x00020000);
    int a_lds[A_PER], a_y[A_PER], a_x[A_PER];
    unsigned a_base[A_PER];
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
        a_base[i] = (unsigned)(mm * p.in_ld + col) * 2u;          // bytes
    }
    int b_off[B_PER], b_lds[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = tid + i * NT;
        const int row = ch >> 3, col = (ch & 7) * 8;
        b_off[i] = (row * p.cin_pad + col) * 2;                    // bytes
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
        const int shift = (((dy * p.W + dx) * p.in_ld) + c0) * 2;                    // bytes, may be negative
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int yy = a_y[i] + dy, xx = a_x[i] + dx;
            const bool ok = a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const unsigned off = ok ? a_base[i] + (unsigned)shift : 0x80000000u;
            a_reg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
        }
        const unsigned woff = (unsigned)__builtin_amdgcn_readfirstlane(((tap * p.cout_pad + n0) * p.cin_pad + c0) * 2);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) b_reg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], woff, 0));
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

    // epilogue: D row = channel (e&3) + 8*(e>>2) + 4*h of the 32-channel tile, col = pixel i32.  The accumulator layout gives a lane
    // 8 bytes of ONE pixel row and the 32 lanes of a half-wave 32 different rows: written out directly that is 64 partial cache
    // lines per store instruction (heads.0 at batch 32: 140 us for 247 MB).  The tile therefore goes through the (now free) LDS
    // tiles: bias + activation (+ residual, f32) in registers, bf16 into [BM][BN + 8], then 16 bytes per lane on consecutive
    // addresses — whole pixel rows of the workgroup's BN channels.  The NCHW stage outputs are pixel-major per channel already.
    __bf16* out = (__bf16*)p.out;
    const __bf16* res = (const __bf16*)p.res;
    const int64_t HW = (int64_t)p.H * p.W;
    constexpr int OLD_ = BN + 8;
    static_assert((size_t)BM * OLD_ * 2 <= (size_t)2 * (BM + BN) * HLD * 2, "the staged output tile must fit the operand tiles");
    __bf16* Ot = (__bf16*)hsm_raw;                     // [BM][OLD_] (the last K step ended with a barrier: the tiles are free)
    const bool res_vec = res && ((p.res_ld & 3) == 0) && ((((uintptr_t)res) & 7) == 0);
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        const int lrow = (wm * RM + i) * 32 + i32;
        const int64_t m = m0 + lrow;
        const bool mok = m < M;
        const int64_t mc = mok ? m : 0;
        const int64_t img = mc / HW, pix = mc - img * HW;
        const int c0 = p.out_split > 0 ? p.out_split : p.cout;
        auto store_nchw = [&](int ch, float val) {       // stage outputs (NCHW); merged heads split at out_split
            if (ch < c0) { if (p.out_nchw) p.out_nchw[(img * c0 + ch) * HW + pix] = val; }
            else if (p.out_nchw2) p.out_nchw2[(img * (p.cout - c0) + (ch - c0)) * HW + pix] = val;
        };
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lcol = (wn * RN + j) * 32 + 8 * g + 4 * h;
                const int n = n0 + lcol;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const f32x4 bias = *(const f32x4*)(p.bias + n);       // bias is padded to cout_pad
                v += bias;
                v.x = act_f(v.x, p.act); v.y = act_f(v.y, p.act); v.z = act_f(v.z, p.act); v.w = act_f(v.w, p.act);
                if (res && mok) {                                     // residual before the (single) rounding to bf16
                    if (res_vec && n + 3 < p.cout) {
                        const bf16x4 r = *(const bf16x4*)(res + m * p.res_ld + n);
                        v.x += (float)r[0]; v.y += (float)r[1]; v.z += (float)r[2]; v.w += (float)r[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.cout) v[e] += (float)res[m * p.res_ld + n + e];
                    }
                }
                if ((p.out_nchw || p.out_nchw2) && mok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.cout) store_nchw(n + e, v[e]);
                }
                const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *(bf16x4*)(Ot + lrow * OLD_ + lcol) = o;
            }
        }
    }
    __syncthreads();
    const bool out_vec = ((p.out_ld & 7) == 0) && ((((uintptr_t)out) & 15) == 0);
    constexpr int CPR = BN / 8;                        // 16-byte chunks per pixel row
    for (int ch = tid; ch < BM * CPR; ch += NT) {
        const int row = ch / CPR, col = (ch - row * CPR) * 8;
        const int64_t m = m0 + row;
        const int n = n0 + col;
        if (m >= M || n >= p.cout) continue;
        const bf16x8 o = *(const bf16x8*)(Ot + row * OLD_ + col);
        if (out_vec && n + 7 < p.cout) *(bf16x8*)(out + m * p.out_ld + n) = o;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < p.cout) out[m * p.out_ld + n + e] = o[e];
        }
    }
}

template <int BM, int BN, int RM, int RN>
static hipError_t launch_gemm_bf16_t(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int64_t tiles = ((M + BM - 1) / BM) * (p.cout_pad / BN);
    if (M * p.in_ld * 2 >= (1ll << 31)) return hipErrorInvalidValue;        // 32-bit buffer offsets (2 GiB of activations per launch)
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

// Measured and dropped: the heads kernel's structure for the single 1x1 convs to 128 channels (cpm.align, refinement `initial`):
// activations straight into MFMA operand registers (32-byte pieces of 32 pixel rows per request), only the weights through
// LDS in 64-channel chunks, LDS-staged output — 36.8 / 21.7 / 21.1 us against 32.7 / 21.2 / 18.5 for the shared-tile kernel
// with its staged epilogue (batch 32, K = 512 / 192 / 128).
// ---------------------------------------------------------------------------------------- stage heads, both 1x1 convs in one kernel
// out[m][0..NH+NP) = W1 . relu(W0 . x[m] + b0) + b1 with the hidden vector (1024 wide in the initial stage: 247 MB per tensor at
// batch 32, written by one GEMM and read back by the next) kept on the CU.  Workgroup = 4 waves = 128 pixels, wave = 32 pixels:
//   - the wave's activations are loaded ONCE, straight into MFMA operand registers (8 k-slices of v_mfma_f32_32x32x16_bf16);
//   - the hidden dimension is walked in chunks of 64: W0 chunk [64][128] and W1 chunk [64 out][64] staged in LDS (double-buffered,
//     requested a chunk ahead, one barrier per chunk);
//   - GEMM 1 (16 MFMAs) leaves D[hidden channel][pixel] in the accumulators: lane (pixel i, half h) holds channels
//     32 j + 8 (e >> 2) + 4 h + (e & 3), e = 0..15.  After bias + ReLU those 16 values ARE the two B-operand k-slices of GEMM 2
//     if its k-slot (s, h, r) is read as hidden 32 j + 16 s + 8 (r >> 2) + 4 h + (r & 3): a permutation of the summation index,
//     applied to the W1 fragment instead (two ds_read_b64 at columns 16 s + 4 h and 16 s + 8 + 4 h) — the hidden tile never
//     touches LDS or another lane;
//   - GEMM 2 (8 MFMAs per chunk) accumulates the 64 (57 used) output channels of the wave's 32 pixels over all chunks.
constexpr int HD_CH = 64, HD_K = 128;
constexpr int HD_W0LD = HD_K + 8, HD_W1LD = HD_CH + 8;

template <int RM>
__global__ void __launch_bounds__(256) heads_bf16_kernel(HeadsParams p) {
    constexpr int HD_BM = 128 * RM;                                  // wave = 32 RM pixels: every weight fragment read from LDS feeds RM MFMAs
    extern __shared__ __attribute__((aligned(16))) unsigned char hd_raw[];
    __bf16* W0s = (__bf16*)hd_raw;                                   // [2][64][HD_W0LD]
    __bf16* W1s = W0s + 2 * HD_CH * HD_W0LD;                         // [2][64][HD_W1LD]
    float* B0s = (float*)(W1s + 2 * 64 * HD_W1LD);                   // [hidden]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const int M = p.N * p.H * p.W;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }
    int mr[RM];
    bool mok[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) { mr[i] = bid * HD_BM + (wave * RM + i) * 32 + i32; mok[i] = mr[i] < M; }
    const int nch = p.hidden / HD_CH;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((int64_t)M * p.in_ld * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w0r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, p.hidden * HD_K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 64 * p.hidden * 2, 0x00020000);

    // weight staging: W0 chunk = 64 rows x 256 B (4 x 16 B per thread), W1 chunk = 64 rows x 128 B (2 x 16 B per thread)
    bf16x8 st0[4], st1[2];
    auto request = [&](int c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = tid + u * 256, row = ch >> 4, col = (ch & 15) * 8;
            st0[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w0r, (row * HD_K + col) * 2, c * (HD_CH * HD_K * 2), 0));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + u * 256, row = ch >> 3, col = (ch & 7) * 8;
            st1[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1r, (row * p.hidden + col) * 2, c * (HD_CH * 2), 0));
        }
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = tid + u * 256, row = ch >> 4, col = (ch & 15) * 8;
            *(bf16x8*)(W0s + (buf * HD_CH + row) * HD_W0LD + col) = st0[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = tid + u * 256, row = ch >> 3, col = (ch & 7) * 8;
            *(bf16x8*)(W1s + (buf * 64 + row) * HD_W1LD + col) = st1[u];
        }
    };
    request(0);
    // the wave's activations: k-slice s of pixel m = x[m][16 s + 8 h .. + 7]
    bf16x8 xf[RM][HD_K / 16];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int s = 0; s < HD_K / 16; ++s)
            xf[i][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, mok[i] ? (unsigned)((mr[i] * p.in_ld + 16 * s + 8 * h) * 2) : 0x80000000u, 0, 0));
    for (int i = tid * 4; i < p.hidden; i += 256 * 4) *(f32x4*)(B0s + i) = *(const f32x4*)(p.b0 + i);
    land(0);
    __syncthreads();

    f32x16 acc2[RM][2];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][t][e] = 0.f;
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        request(c + 1 < nch ? c + 1 : c);                            // unconditional (the last chunk again): exact vmcnt counting
        __builtin_amdgcn_sched_barrier(0);                           // keep the requests ahead of the chunk's MFMAs
        const __bf16* w0 = W0s + (buf * HD_CH + i32) * HD_W0LD + 8 * h;
        const __bf16* w1 = W1s + (buf * 64 + i32) * HD_W1LD + 4 * h;
        bf16x8 hf[RM][2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 acc1[RM];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
            bf16x8 wv[HD_K / 16];                                    // all eight fragment reads in flight before the first MFMA
#pragma unroll
            for (int s = 0; s < HD_K / 16; ++s) wv[s] = *(const bf16x8*)(w0 + j * 32 * HD_W0LD + 16 * s);
            __builtin_amdgcn_sched_barrier(0);                       // (the scheduler otherwise pairs every read with its MFMA again)
#pragma unroll
            for (int s = 0; s < HD_K / 16; ++s)
#pragma unroll
                for (int i = 0; i < RM; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[s], xf[i][s], acc1[i], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(const f32x4*)(B0s + c * HD_CH + j * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hf[i][j][g >> 1][(g & 1) * 4 + r] = (__bf16)fmaxf(acc1[i][4 * g + r] + b[r], 0.f);
            }
        }
        bf16x8 w1v[2][2][2];                                         // the eight W1 fragments (hidden index permuted: see above)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const __bf16* src = w1 + t * 32 * HD_W1LD + j * 32 + 16 * s;
                    const bf16x4 lo = *(const bf16x4*)src, hi = *(const bf16x4*)(src + 8);
                    w1v[t][j][s] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < RM; ++i) acc2[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1v[t][j][s], hf[i][j][s], acc2[i][t], 0, 0, 0);
        if (c + 1 < nch) land(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane (pixel i32, half h) holds output channels 32 t + 8 g + 4 h + r
    __bf16* out = (__bf16*)p.out;
    const int HW = p.H * p.W;
    const int c0 = p.out_split > 0 ? p.out_split : p.cout;
    const bool out_vec = ((p.out_ld & 3) == 0) && ((((uintptr_t)out) & 7) == 0);
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        if (!mok[i]) continue;
        const int m = mr[i];
        const int img = m / HW, pix = m - img * HW;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * t + 8 * g + 4 * h;
                if (n >= p.cout) continue;
                const f32x4 b = *(const f32x4*)(p.b1 + n);
                const f32x4 v = {acc2[i][t][4 * g] + b[0], acc2[i][t][4 * g + 1] + b[1], acc2[i][t][4 * g + 2] + b[2], acc2[i][t][4 * g + 3] + b[3]};
                if (out_vec && n + 3 < p.cout) {                 // 4 consecutive channels: one 8-byte store
                    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *(bf16x4*)(out + (int64_t)m * p.out_ld + n) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.cout) out[(int64_t)m * p.out_ld + n + r] = (__bf16)v[r];
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

bool heads_bf16_supported(int cin_pad, int hidden, int cout_pad) {
    return cin_pad == HD_K && hidden % HD_CH == 0 && hidden >= HD_CH && hidden <= 4096 && cout_pad == 64;
}

template <int RM>
static hipError_t launch_heads_bf16_t(const HeadsParams& p, hipStream_t s) {
    constexpr int BM = 128 * RM;
    const int64_t M = (int64_t)p.N * p.H * p.W;
    if (M >= (1ll << 31) - BM || M * p.in_ld * 2 >= (1ll << 31) || (p.in_ld & 7)) return hipErrorInvalidValue;
    const size_t lds = (size_t)(2 * HD_CH * HD_W0LD + 2 * 64 * HD_W1LD) * 2 + (size_t)p.hidden * sizeof(float);
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) {
        hipError_t e = attr.ensure((const void*)heads_bf16_kernel<RM>, 96 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(heads_bf16_kernel<RM>, dim3((unsigned)((M + BM - 1) / BM)), dim3(256), lds, s, p);
    return hipGetLastError();
}
hipError_t launch_heads_bf16(const HeadsParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    const int rm = T.heads_rm ? T.heads_rm : 1;               // LWP_HEADS_RM (experiments): 64 rows per wave (198 VGPRs, 2 waves per SIMD) measured 12-23 % slower
    (void)M;
    LWP_VARIANT(p, "heads_bf16<%d>", rm == 2 ? 2 : 1);
    return rm == 2 ? launch_heads_bf16_t<2>(p, s) : launch_heads_bf16_t<1>(p, s);
}

// ---------------------------------------------------------------------------------------- implicit GEMM, activation window resident
// Dense 3x3 convs with 128 input channels at large M (batch 32: M = 120704 pixels, N = 128, K = 9 x 128).
// The nine A tiles of a 3x3 conv are ONE pixel window at nine offsets: with pixels flat over (n, y, x) the window of the tile
// [m0, m0 + BM) is the contiguous run [m0 - dil*(W+1), m0 + BM + dil*(W+1)) of NHWC rows.  The workgroup keeps ONE 64-channel
// block of that run in LDS (rows of 64 bf16 + 16 B of padding: conflict-free ds_read_b128), walks the nine taps over it, then
// overwrites it with the second block — which has been waiting in registers since the start of the tile — and walks the taps
// again.  The K loop therefore moves only the weights (16 KB per 64-deep step through a double-buffered LDS tile, one barrier
// per step, BD steps in flight in a register ring); a tap outside the image reads a zero row (select on the lane's LDS address).
// With 128-row tiles the whole working set is 79 KB, so TWO workgroups share a CU and cover each other's staging, barrier
// and epilogue phases; dilation-2 layers need 256-row tiles (1 workgroup per CU).
// Ablation of the first version (256 rows, full 128-channel window, 1 workgroup per CU, us per launch at batch 32, total 68.5):
// launch floor 3-5.7, scattered 8-byte epilogue stores 18.6, window staging 7.2, LDS fragment reads alone 23, MFMA bound 16.2;
// after fragment double-buffering, exact vmcnt counting (no data-dependent branch in the loop) and the LDS-transposed
// epilogue: 44.1 = floor 3.0 + staging 3.3 + K loop 29 + epilogue 11 (all CUs write at once: HBM-write bound, nothing overlaps).
// Epilogue: bias + activation in registers, f32 tile through the (now free) LDS, then 16-byte fully coalesced stores of whole
// 256-byte pixel rows (residual added from equally coalesced loads, rounded to bf16 once).
// DBG (compile time, 0 in production; LWP_GEMMH_DEBUG selects an ablation build of the 256-row configuration):
// 1 no epilogue, 2 no MFMA, 4 no window staging, 8 no weight stream, 16 no K loop
template <int BM, int WM, int WN, int KSZ, int BD, int WCH, int DBG = 0, bool F2 = false>     // WCH: 16-byte window chunks per thread and channel block; F2: fused second conv
__global__ void __launch_bounds__(WM * WN * 64) gemm_bf16_ar_kernel(GemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BN = 128, CIN = 128;
    constexpr int RM = BM / WM / 32, RN = BN / WN / 32;
    constexpr int B_PER = BN * 8 / NT;                 // 16-byte chunks of the weight tile per thread
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && (BN * 8) % NT == 0, "tile / wave mismatch");
    constexpr int taps = KSZ * KSZ;
    constexpr int CS = HBK + 8;                        // window row stride (elements): 144 B
    constexpr int kblocks = CIN / HBK;                 // 2
    constexpr int nsteps = kblocks * taps;
    static_assert(kblocks == 2, "one block resident, one in registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char arm_raw[];
    const int halo = KSZ == 3 ? p.dil * (p.W + 1) : 0;
    const int R = BM + 2 * halo;                       // window rows
    __bf16* Aw = (__bf16*)arm_raw;                     // [R + 1][CS]  (last row: zeros)
    __bf16* zrow = Aw + (size_t)R * CS;
    __bf16* Bs = zrow + CS;                            // [3][BN][HLD]: step k reads buffer k % 3 while step k + 2's tile is written
    const __bf16* in = (const __bf16*)p.in;
    const __bf16* wgt = (const __bf16*)p.w;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int i32 = lane & 31, h = lane >> 5;
    const int M = p.N * p.H * p.W;                     // host guarantees < 2^31
    const int ntn = p.cout_pad / BN;
    const int ntiles = ((M + BM - 1) / BM) * ntn;
    // persistent workgroups: block b walks the tiles remap(b), remap(b) + G, ...  (XCD-aware order of the first index: the
    // blocks of one XCD take neighbouring tiles, whose windows overlap by the halo)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rem = nwg & 7, xcd = bid & 7;
        bid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    }

    // ---- weight tile staging: BD steps in flight in a statically indexed register ring, double-buffered in LDS
    int b_off[B_PER], b_lds[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = tid + i * NT;
        const int row = ch >> 3, col = (ch & 7) * 8;
        b_off[i] = row * CIN + col;
        b_lds[i] = row * HLD + col;
    }
    bf16x8 b_reg[BD][B_PER];
    int n0 = 0;                                        // first output channel of the tile whose weights are being streamed
    // buffer loads: wave-uniform descriptor and tile/tap offset in SGPRs, one 32-bit offset register per lane (a 64-bit per-lane
    // address costs ~3x the issue time while the matrix pipes are busy)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](int step, bf16x8* dst) {
        const int kb = step / taps, tap = step - kb * taps;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((tap * p.cout_pad + n0) * CIN + kb * HBK) * 2);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) dst[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i] * 2, soff, 0));
    };
    auto store_b = [&](int buf, const bf16x8* src) {
        __bf16* b = Bs + buf * BN * HLD;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *(bf16x8*)(b + b_lds[i]) = src[i];
    };

    // ---- pixel window: row r <-> flat pixel m0 - halo + r.  Channel block 0 goes to LDS at the start of the tile; block 1 is
    // loaded behind it into registers (WCH 16-byte chunks per thread) and replaces block 0 in LDS after the ninth tap.
    const int wtotal = R * 8;                                    // chunks of one 64-channel block (<= WCH * NT: host check)
    bf16x8 wreg[WCH];
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((int64_t)M * p.in_ld * 2), 0x00020000);   // host: < 2^31 bytes
    auto win_load = [&](int m0_, int kb) {
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int ch = tid + u * NT;
            const int row = ch >> 3, col = (ch & 7) * 8 + kb * HBK;
            const int g = m0_ - halo + row;
            // buffer load: rows outside the tensor read offset 2^31 >= num_records and come back as zeros
            wreg[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(irsrc, (ch < wtotal && g >= 0 && g < M) ? (unsigned)((g * p.in_ld + col) * 2) : 0x80000000u, 0, 0));
        }
    };
    auto win_write = [&]() {
#pragma unroll
        for (int u = 0; u < WCH; ++u) {
            const int ch = tid + u * NT;
            if (ch < wtotal) *(bf16x8*)(Aw + (size_t)(ch >> 3) * CS + (ch & 7) * 8) = wreg[u];
        }
    };
    // everything of the NEXT tile that can be requested early: its weights (ring) and block 0 of its window (registers)
    auto prefetch_tile = [&](int tile) {
        n0 = (tile % ntn) * BN;
        if (!(DBG & 8)) {
#pragma unroll
            for (int d = 0; d < BD; ++d) load_b(d, b_reg[d]);
        }
        if (!(DBG & 4)) win_load((tile / ntn) * BM, 0);
    };

    const int zoff = R * CS + 8 * h;                   // the zero row, as an offset from Aw
    const int b_lane = (wn * RN * 32 + i32) * HLD + 8 * h;
    using std::integral_constant;
    typedef integral_constant<bool, false> no_t;
    typedef integral_constant<bool, true> yes_t;
    static_assert(BD == 3 && taps % BD == 0, "ring of three: the buffer of a step is its position in the unrolled group");

    int tile = bid;
    if (tile < ntiles) prefetch_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = (tile / ntn) * BM;
        const int n0_cur = (tile % ntn) * BN;
        // ---- start of the tile: block 0 of the window lands in LDS (its loads were issued a tile ago), block 1 is requested
        if (!(DBG & 4)) {
            win_write();
            win_load(m0, 1);                               // in flight during the first nine steps
        }
        for (int c = tid * 8; c < CS; c += NT * 8) *(bf16x8*)(zrow + c) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (!(DBG & 8)) {
            store_b(0, b_reg[0]);
            store_b(1, b_reg[1]);
            load_b(3, b_reg[0]);                           // ring: set s % 3 holds step s; steps 2, 3 (and from step 0 on: 4) are in flight
        }

        // the lane's pixels: window row of the tap (0, 0) and the taps that fall inside the image
        int a_row[RM];
        unsigned valid[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int lp = (wm * RM + i) * 32 + i32;       // pixel within the tile
            const int m = m0 + lp;
            const bool row_ok = m < M;
            const int mm = row_ok ? m : 0;
            const int qy = mm / p.W;
            const int x = mm - qy * p.W, y = qy % p.H;
            unsigned v = 0;
#pragma unroll
            for (int t = 0; t < taps; ++t) {
                const int yy = y + (KSZ == 3 ? (t / 3 - 1) * p.dil : 0), xx = x + (KSZ == 3 ? (t % 3 - 1) * p.dil : 0);
                if (row_ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v |= 1u << t;
            }
            valid[i] = v;
            a_row[i] = lp * CS + 8 * h;                    // element offset; + tap offset (ty*dil*W + tx*dil) * CS
        }
        auto a_off = [&](int step, int i) -> int {
            const int tap = step < taps ? step : step - taps;
            const int toff = KSZ == 3 ? ((tap / 3) * p.dil * p.W + (tap % 3) * p.dil) * CS : 0;
            return ((valid[i] >> tap) & 1u) ? a_row[i] + toff : zoff;
        };

        f32x16 acc[RM][RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // fragment registers, double-buffered over the 16-deep sub-steps
        bf16x8 xa[2][RM], wb[2][RN];
        int ao[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) ao[i] = a_off(0, i);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RM; ++i) xa[0][i] = *(const bf16x8*)(Aw + ao[i]);
#pragma unroll
        for (int j = 0; j < RN; ++j) wb[0][j] = *(const bf16x8*)(Bs + b_lane + j * 32 * HLD);

        // one K step; JB (compile time) = position in the register ring, RESTAGE = the last step on channel block 0.
        // No data-dependent branch inside: look-ahead indices are clamped (the last steps reload / restore harmlessly), so the
        // compiler's vmcnt counting stays exact and BD - 1 weight steps really stay in flight across the barrier.
        auto do_step = [&](int step, auto JB_, auto RESTAGE_) {
            constexpr int jb = decltype(JB_)::value;
            constexpr bool restage = decltype(RESTAGE_)::value;
            // weight tiles: three LDS buffers, buffer = step % 3 = jb.  Step k multiplies from buffer k % 3, writes step k + 2's tile
            // (ring set (jb + 2) % 3, requested two steps ago) and requests step k + 4 into the set that held step k + 1 (in LDS since
            // step k - 1).  Step k + 1's tile is therefore complete BEFORE this step's barrier, so its first fragments — like the
            // activation ones — are read ahead of the barrier and the MFMAs restart right behind it (with two buffers every step
            // began with an exposed LDS round trip: ~20 % of the loop).
            if (!(DBG & 8)) load_b(step + 4 < nsteps ? step + 4 : nsteps - 1, b_reg[(jb + 1) % 3]);
            const __bf16* b = Bs + jb * BN * HLD + b_lane;
            const __bf16* bnext = Bs + ((jb + 1) % 3) * BN * HLD + b_lane;
            int an[RM];                                          // next step's activation offsets
#pragma unroll
            for (int i = 0; i < RM; ++i) an[i] = a_off(step + 1 < nsteps ? step + 1 : step, i);
#pragma unroll
            for (int s = 0; s < HBK / 16; ++s) {
                const int c = s & 1, n = c ^ 1;
                if (s + 1 < HBK / 16) {
#pragma unroll
                    for (int i = 0; i < RM; ++i) xa[n][i] = *(const bf16x8*)(Aw + ao[i] + 16 * (s + 1));
#pragma unroll
                    for (int j = 0; j < RN; ++j) wb[n][j] = *(const bf16x8*)(b + j * 32 * HLD + 16 * (s + 1));
                } else {
#pragma unroll
                    for (int j = 0; j < RN; ++j) wb[n][j] = *(const bf16x8*)(bnext + j * 32 * HLD);   // next step, sub-step 0: before the barrier
                    if (!restage) {
#pragma unroll
                        for (int i = 0; i < RM; ++i) xa[n][i] = *(const bf16x8*)(Aw + an[i]);
                    }
                }
                if (DBG & 2) {
#pragma unroll
                    for (int i = 0; i < RM; ++i) asm volatile("" ::"v"(xa[c][i]));
#pragma unroll
                    for (int j = 0; j < RN; ++j) asm volatile("" ::"v"(wb[c][j]));
                } else {
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[c][j], xa[c][i], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < RM; ++i) ao[i] = an[i];
            if (!(DBG & 8)) store_b((jb + 2) % 3, b_reg[(jb + 2) % 3]);
            __syncthreads();
            if (restage) {                                       // every wave is past its last read of block 0: block 1 takes its place
                if (!(DBG & 4)) win_write();
                __syncthreads();
#pragma unroll
                for (int i = 0; i < RM; ++i) xa[0][i] = *(const bf16x8*)(Aw + ao[i]);
            }
        };
        if (!(DBG & 16)) {
            for (int kb = 0; kb < kblocks; ++kb) {
                const int s00 = kb * taps;
                for (int s0 = s00; s0 < s00 + taps - BD; s0 += BD) {
                    do_step(s0, integral_constant<int, 0>{}, no_t{});
                    if (BD > 1) do_step(s0 + 1, integral_constant<int, 1 % BD>{}, no_t{});
                    if (BD > 2) do_step(s0 + 2, integral_constant<int, 2 % BD>{}, no_t{});
                }
                const int sl = s00 + taps - BD;
                if (kb == 0) {
                    if (BD == 1) do_step(sl, integral_constant<int, 0>{}, yes_t{});
                    if (BD == 3) { do_step(sl, integral_constant<int, 0>{}, no_t{}); do_step(sl + 1, integral_constant<int, 1 % BD>{}, no_t{}); do_step(sl + 2, integral_constant<int, 2 % BD>{}, yes_t{}); }
                } else {
                    do_step(sl, integral_constant<int, 0>{}, no_t{});
                    if (BD > 1) do_step(sl + 1, integral_constant<int, 1 % BD>{}, no_t{});
                    if (BD > 2) do_step(sl + 2, integral_constant<int, 2 % BD>{}, no_t{});
                }
            }
        }

        // ---- the next tile's weights and first window block are requested now: they travel while this tile's epilogue runs
        if (tile + (int)gridDim.x < ntiles) prefetch_tile(tile + gridDim.x);

        // ---- epilogue.  D row = channel (e&3) + 8*(e>>2) + 4*h of the 32-channel tile, col = pixel i32: bias + activation in
        // registers, f32 tile [BM][132] through LDS (all waves are past the last barrier: window and weight tiles are dead)
        if (DBG & 1) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
            if (sum == 12345.678f) ((__bf16*)p.out)[0] = (__bf16)sum;
            continue;
        }
        constexpr int OS = BN + 4;                          // 528-byte rows: conflict-free ds_write_b128 for the accumulator layout
        float* Ot = (float*)arm_raw;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int lp = (wm * RM + i) * 32 + i32;
#pragma unroll
            for (int j = 0; j < RN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = (wn * RN + j) * 32 + 8 * g + 4 * h;
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    v += *(const f32x4*)(p.bias + n0_cur + nl);           // bias is padded to cout_pad
                    v.x = act_f(v.x, p.act); v.y = act_f(v.y, p.act); v.z = act_f(v.z, p.act); v.w = act_f(v.w, p.act);
                    *(f32x4*)(Ot + lp * OS + nl) = v;
                }
            }
        }
        __syncthreads();
        __bf16* out = (__bf16*)p.out;
        const __bf16* res = (const __bf16*)p.res;
        constexpr int OCH = BM * (BN / 8) / NT;             // 16-byte output chunks per thread
        constexpr int OG = OCH >= 4 ? 4 : OCH;              // chunks per pass (bounds the residual registers)
        if (F2) {
            // ---- fused second conv (the next refinement block's `initial` 1x1, with_mobilenet.py:57): the block output of this tile
            // (t = relu(conv) + residual, rounded to bf16 exactly as the stored tensor would be) becomes the MFMA operand of
            // out2 = act2(W2 . t + b2) and never goes to memory: 31 MB less written, 31 MB less read and one launch less per block.
            // pass 1: Ot (f32) + residual -> bf16, written IN PLACE over the head of the same Ot row (row stride stays 528 B): the 16
            // chunks of a row belong to 16 consecutive lanes of one wave, whose LDS reads of an iteration precede its LDS writes.
            // The second conv's weights W2 [128][128] go into the TAILS of the first 128 rows (bytes 272 .. 527 of a row are free once
            // the row has been converted): row n of W2 rides with tile row n, 16 bytes per lane of that row's chunk group.
            __bf16* Tt = (__bf16*)arm_raw;
            constexpr int TS = OS * 2;                           // row stride of the bf16 images in elements (528 B)
            constexpr int W2O = BN + 8;                          // element offset of a row's W2 slot (272 B)
            const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, BN * CIN * 2, 0x00020000);
#pragma unroll
            for (int u0 = 0; u0 < OCH; u0 += OG) {
                bf16x8 rv[OG], w2v[OG];
#pragma unroll
                for (int u = 0; u < OG; ++u) {
                    const int ch = tid + (u0 + u) * NT;
                    const int row = ch >> 4, col = (ch & 15) * 8;
                    const int m = m0 + row;
                    if (res) rv[u] = (m < M) ? *(const bf16x8*)(res + (int64_t)m * p.res_ld + n0_cur + col) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    w2v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2r, row < BN ? (unsigned)((row * CIN + col) * 2) : 0x80000000u, 0, 0));
                }
                f32x4 v0[OG], v1[OG];
#pragma unroll
                for (int u = 0; u < OG; ++u) {
                    const int ch = tid + (u0 + u) * NT;
                    const int row = ch >> 4, col = (ch & 15) * 8;
                    v0[u] = *(const f32x4*)(Ot + row * OS + col); v1[u] = *(const f32x4*)(Ot + row * OS + col + 4);
                }
                __builtin_amdgcn_sched_barrier(0);               // all reads of this pass are issued before its first write
#pragma unroll
                for (int u = 0; u < OG; ++u) {
                    const int ch = tid + (u0 + u) * NT;
                    const int row = ch >> 4, col = (ch & 15) * 8;
                    const float f[8] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w, v1[u].x, v1[u].y, v1[u].z, v1[u].w};
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(res ? f[e] + (float)rv[u][e] : f[e]);
                    *(bf16x8*)(Tt + row * TS + col) = o;
                    if (row < BN) *(bf16x8*)(Tt + row * TS + W2O + col) = w2v[u];
                }
            }
            __syncthreads();
            // second GEMM: wave (wm, wn) as in the main loop, D row = channel, col = pixel
            f32x16 acc2[RM][RN];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
#pragma unroll
            for (int s8 = 0; s8 < CIN / 16; ++s8) {
                bf16x8 xt[RM];
#pragma unroll
                for (int i = 0; i < RM; ++i) xt[i] = *(const bf16x8*)(Tt + ((wm * RM + i) * 32 + i32) * TS + 16 * s8 + 8 * h);
                bf16x8 w2f[RN];
#pragma unroll
                for (int j = 0; j < RN; ++j) w2f[j] = *(const bf16x8*)(Tt + ((wn * RN + j) * 32 + i32) * TS + W2O + 16 * s8 + 8 * h);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[j], xt[i], acc2[i][j], 0, 0, 0);
            }
            __syncthreads();                                     // every wave has read the bf16 image: the f32 tile may overwrite it
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int lp = (wm * RM + i) * 32 + i32;
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = (wn * RN + j) * 32 + 8 * g + 4 * h;
                        f32x4 v = {acc2[i][j][4 * g], acc2[i][j][4 * g + 1], acc2[i][j][4 * g + 2], acc2[i][j][4 * g + 3]};
                        v += *(const f32x4*)(p.bias2 + nl);
                        v.x = act_f(v.x, p.act2); v.y = act_f(v.y, p.act2); v.z = act_f(v.z, p.act2); v.w = act_f(v.w, p.act2);
                        *(f32x4*)(Ot + lp * OS + nl) = v;
                    }
            }
            __syncthreads();
            __bf16* out2 = (__bf16*)p.out2;
#pragma unroll
            for (int u = 0; u < OCH; ++u) {
                const int ch = tid + u * NT;
                const int row = ch >> 4, col = (ch & 15) * 8;
                const int m = m0 + row;
                if (m >= M) continue;
                const f32x4 a0 = *(const f32x4*)(Ot + row * OS + col), a1 = *(const f32x4*)(Ot + row * OS + col + 4);
                const bf16x8 o = {(__bf16)a0.x, (__bf16)a0.y, (__bf16)a0.z, (__bf16)a0.w, (__bf16)a1.x, (__bf16)a1.y, (__bf16)a1.z, (__bf16)a1.w};
                *(bf16x8*)(out2 + (int64_t)m * p.out2_ld + col) = o;
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int u0 = 0; u0 < OCH; u0 += OG) {
            bf16x8 rv[OG];
            if (res) {
#pragma unroll
                for (int u = 0; u < OG; ++u) {
                    const int ch = tid + (u0 + u) * NT;
                    const int row = ch >> 4, col = (ch & 15) * 8;
                    const int m = m0 + row;
                    rv[u] = (m < M && n0_cur + col < p.cout) ? *(const bf16x8*)(res + (int64_t)m * p.res_ld + n0_cur + col) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int u = 0; u < OG; ++u) {
                const int ch = tid + (u0 + u) * NT;
                const int row = ch >> 4, col = (ch & 15) * 8;
                const int m = m0 + row;
                if (m >= M || n0_cur + col >= p.cout) continue;     // cout is a multiple of 8 here (host check)
                const f32x4 v0 = *(const f32x4*)(Ot + row * OS + col), v1 = *(const f32x4*)(Ot + row * OS + col + 4);
                float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)(res ? f[e] + (float)rv[u][e] : f[e]);
                *(bf16x8*)(out + (int64_t)m * p.out_ld + n0_cur + col) = o;
            }
        }
        __syncthreads();                                    // the f32 tile has been read: the next tile's window may overwrite it
    }
}

static size_t gemm_bf16_ar_lds(const GemmParams& p, int BM) {
    const int halo = p.ks == 3 ? p.dil * (p.W + 1) : 0;
    const size_t loop = ((size_t)(BM + 2 * halo + 1) * (HBK + 8) + (size_t)3 * 128 * HLD) * 2;
    const size_t epi = (size_t)BM * (128 + 4) * 4;
    return loop > epi ? loop : epi;
}

template <int BM, int WM, int WN, int KSZ, int BD, int WCH, int DBG = 0, bool F2 = false>
static hipError_t launch_gemm_bf16_ar_w(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const int64_t tiles = ((M + BM - 1) / BM) * (p.cout_pad / 128);
    static LdsAttrOnce attr;
    hipError_t e = attr.ensure((const void*)gemm_bf16_ar_kernel<BM, WM, WN, KSZ, BD, WCH, DBG, F2>, 160 * 1024);
    if (e != hipSuccess) return e;
    // persistent workgroups: one per CU (or two when two fit), each walking tiles b, b + G, ...
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    const size_t lds = gemm_bf16_ar_lds(p, BM);
    int64_t grid = tiles;
    if (T.gemmh_persist != 0) {                              // LWP_GEMMH_PERSIST "0": one workgroup per tile (A/B)
        const int64_t slots = (int64_t)device_cu_count() * (lds <= 80 * 1024 ? 2 : 1);
        if (grid > slots) grid = slots;
    }
    hipLaunchKernelGGL((gemm_bf16_ar_kernel<BM, WM, WN, KSZ, BD, WCH, DBG, F2>), dim3((unsigned)grid), dim3(WM * WN * 64), lds, s, p);
    return hipGetLastError();
}
template <int BM, int WM, int WN, int KSZ, int BD, int DBG = 0, bool F2 = false>
static hipError_t launch_gemm_bf16_ar_t(const GemmParams& p, hipStream_t s) {
    // window chunks per thread: two register budgets (the window block waits in registers for nine steps)
    constexpr int NT = WM * WN * 64;
    const int halo = p.ks == 3 ? p.dil * (p.W + 1) : 0;
    const int need = ((BM + 2 * halo) * 8 + NT - 1) / NT;
    constexpr int W_SMALL = ((BM + 2 * 96) * 8 + NT - 1) / NT, W_LARGE = ((BM + 2 * 180) * 8 + NT - 1) / NT;
    if (need <= W_SMALL) return launch_gemm_bf16_ar_w<BM, WM, WN, KSZ, BD, W_SMALL, DBG, F2>(p, s);
    if (need <= W_LARGE) return launch_gemm_bf16_ar_w<BM, WM, WN, KSZ, BD, W_LARGE, DBG, F2>(p, s);
    return hipErrorInvalidValue;
}

// picks the window-resident kernel when the tile's window fits the LDS; *used = false: caller falls back
static hipError_t try_gemm_bf16_ar(const GemmParams& p, hipStream_t s, bool* used) {
    *used = false;
    const int64_t M = (int64_t)p.N * p.H * p.W;
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    if (T.gemmh_ar_off) return hipSuccess;                    // LWP_GEMMH_AR "0": off; "BM,WM,WN,BD": force a configuration (experiments)
    if (p.ks != 3 || p.cout_pad % 128 != 0 || p.cin_pad != 128 || (p.cout & 7) || p.out_nchw || p.out_nchw2 || M >= (1ll << 31) - 512) return hipSuccess;
    if ((p.out_ld & 7) || (((uintptr_t)p.out) & 15) || (p.in_ld & 7) || (((uintptr_t)p.in) & 15)) return hipSuccess;
    if (p.res && ((p.res_ld & 7) || (((uintptr_t)p.res) & 15))) return hipSuccess;
    if (p.dil * (p.W + 1) > 180) return hipSuccess;           // the kernel's register budget for the window block held in registers
    if (M * p.in_ld * 2 >= (1ll << 31)) return hipSuccess;     // 32-bit buffer offsets for the window (larger inputs: the shared-tile kernel and its check)
    // LWP_GEMMH_AR_FORCE "1": the window-resident kernel at every size (tests)
    if (!T.gemmh_ar_force && M < 128 * 256) return hipSuccess;      // small problems: the shared-tile kernel's 64 x 64 tiles fill the chip better
    int bm = 0, wm = 0, wn = 0, bd = 0;
    if (T.has_gemmh_ar) { bm = T.gemmh_ar[0]; wm = T.gemmh_ar[1]; wn = T.gemmh_ar[2]; bd = T.gemmh_ar[3]; }
    else {
        // (128-row tiles, two workgroups per CU, measured slower at batch 32: 64.9 us against 44.3 for 256 rows, dilation 1)
        // (256 x 128 tiles on FOUR waves of 128 x 64 — 0.75 instead of 1 LDS fragment read per MFMA, one wave per SIMD, 255 VGPRs:
        //  63 us against 46)
        if (gemm_bf16_ar_lds(p, 256) <= 160 * 1024) { bm = 256; wm = 4; wn = 2; bd = 3; }
        else if (gemm_bf16_ar_lds(p, 128) <= 160 * 1024) { bm = 128; wm = 2; wn = 2; bd = 3; }
        else return hipSuccess;
    }
    if (gemm_bf16_ar_lds(p, bm) > 160 * 1024) return hipSuccess;
    *used = true;
    LWP_VARIANT(p, "gemm_bf16_ar<%d,%d,%d,%d>", bm, wm, wn, bd);
#ifdef LWP_ABLATION
    const int d = T.gemmh_debug;
#define GAR_DBG(D_) if (bm == 256 && wm == 4 && wn == 2 && d == D_) return launch_gemm_bf16_ar_t<256, 4, 2, 3, 3, D_>(p, s); \
                    if (bm == 128 && wm == 2 && wn == 2 && d == D_) return launch_gemm_bf16_ar_t<128, 2, 2, 3, 3, D_>(p, s);
    GAR_DBG(1) GAR_DBG(2) GAR_DBG(4) GAR_DBG(8) GAR_DBG(16) GAR_DBG(17) GAR_DBG(21) GAR_DBG(10) GAR_DBG(14) GAR_DBG(15)
#undef GAR_DBG
#endif
    // the next layer's 1x1 (128 -> 128) folded into this kernel's epilogue: main configuration, one 128-channel tile per pixel tile
    if (p.w2 && p.fused2 && T.gemmh_fold != 0 && bm == 256 && wm == 4 && wn == 2 && bd == 3 && p.cout_pad == 128 && p.cout == 128 &&
        !(p.out2_ld & 7) && !(((uintptr_t)p.out2) & 15)) {
        *p.fused2 = true;
        LWP_VARIANT(p, "gemm_bf16_ar<256,4,2,3>+1x1");
        return launch_gemm_bf16_ar_t<256, 4, 2, 3, 3, 0, true>(p, s);
    }
#define GAR_CASE(BM_, WM_, WN_, BD_) if (bm == BM_ && wm == WM_ && wn == WN_ && bd == BD_) return launch_gemm_bf16_ar_t<BM_, WM_, WN_, 3, BD_>(p, s);
    GAR_CASE(256, 4, 2, 3) GAR_CASE(128, 2, 2, 3) GAR_CASE(128, 4, 2, 3)
#undef GAR_CASE
    *used = false;
    return hipSuccess;
}

hipError_t launch_gemm_bf16(const GemmParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.H * p.W;
    {
        bool used = false;
        hipError_t e = try_gemm_bf16_ar(p, s, &used);
        if (e != hipSuccess || used) return e;
    }
    // experiments: LWP_GEMMH = "BM,BN,RM,RN"
    const Tuning& T = p.tune ? *p.tune : default_tuning();
    int bm = 0, bn = 0, rm = 0, rn = 0;
    if (T.has_gemmh && T.gemmh[1] > 0 && p.cout_pad % T.gemmh[1] == 0) {
        bm = T.gemmh[0]; bn = T.gemmh[1]; rm = T.gemmh[2]; rn = T.gemmh[3];
    } else if (p.cout_pad % 128 == 0 && ((M + 127) / 128) * (p.cout_pad / 128) >= 512) {
        bm = 128; bn = 128; rm = 2; rn = 1;           // 8 waves x (64 x 32): 6-25 % faster than 4 waves x (64 x 64) on every layer at batch 32
    } else if (((M + 127) / 128) * (p.cout_pad / 64) >= 512) {
        bm = 128; bn = 64; rm = 1; rn = 1;            // 8 waves x (32 x 32): the N = 64 head GEMMs, ~10 % over 4 waves x (64 x 32)
    } else {
        bm = 64; bn = 64; rm = 1; rn = 1;             // 4 waves x (32 x 32): small problems
    }
    LWP_VARIANT(p, "gemm_bf16<%d,%d,%d,%d>", bm, bn, rm, rn);
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
