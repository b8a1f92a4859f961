// Internal declarations shared by the host graph code, the HIP kernels and the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/lwpose.h"

namespace lwp {

// ------------------------------------------------------------------ parameter table
struct ParamSpec {
    std::string key;
    int64_t shape[4];
    int ndim;
    int role;
};
std::vector<ParamSpec> param_table(int nref, int C, int NH, int NP);

// ------------------------------------------------------------------ layer graph
enum LayerKind { L_STEM = 0, L_DW = 1, L_GEMM = 2, L_DWPW = 3 };
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2 };
enum KClass { KC_STEM = 0, KC_DW = 1, KC_PW = 2, KC_C3 = 3, KC_POST = 4, KC_OTHER = 5, KC_COUNT = 6 };

struct BufRef {      // a channel window of an NHWC activation buffer
    int buf = -1;    // index into Graph::bufs
    int coff = 0;    // first channel
    int ld = 0;      // row stride (channels per pixel) the layer addresses the buffer with
};

struct WBlock {              // one source conv of a merged 1x1 layer: W[out_off + o][in_off + i] = conv.weight[o][i]
    std::string conv_key;
    int out_off, in_off, cout, cin;
};

struct Layer {
    int kind = L_GEMM;
    std::string name;          // e.g. "model.3.pw"
    std::string conv_key;      // state_dict prefix of the conv ("model.3.3")
    std::string bn_key;        // state_dict prefix of the BN ("" = none)
    bool has_bias = false;
    int cin = 0, cout = 0, ks = 1, stride = 1, dil = 1, act = ACT_NONE;
    BufRef src, dst, res;      // res.buf < 0: no residual
    int out_index = -1;        // >= 0: this layer also produces stage output #out_index (NCHW)
    int out_index2 = -1, out_split = 0;   // merged heads: channels >= out_split belong to stage output #out_index2
    std::vector<WBlock> blocks;           // non-empty: weight matrix assembled from these convs (zeros elsewhere)
    int64_t macs_per_pixel = 0;           // algorithmic multiply-adds per output pixel (zero blocks not counted)
    // packed weights (float offsets into the blob)
    size_t w_off = 0, b_off = 0;
    int cin_pad = 0, cout_pad = 0;
    // L_DWPW: the pointwise half (conv_key/bn_key/act/stride/dil above describe the depthwise half)
    std::string conv2_key, bn2_key;
    int act2 = ACT_NONE;
    size_t w2_off = 0, b2_off = 0;
};

struct BufSpec {
    int level;      // spatial level: 1 = H/2, 2 = H/4, 3 = H/8
    int channels;   // capacity in channels per pixel (layers may address it with a smaller ld)
    bool has_pad = false;   // some channels are never written but read with zero weights (the concat buffer): keep them finite
};

struct Graph {
    int nref, C, NH, NP;
    std::vector<BufSpec> bufs;
    std::vector<Layer> layers;
    size_t blob_floats = 0;
    int cat_buf = -1;          // the [feat | heat | paf | pad] buffer
    int cat_channels = 0;
    int dtype = LWP_F32;       // storage / MFMA dtype of the conv stack (weights packed accordingly)
};
Graph build_graph(int nref, int C, int NH, int NP, bool fuse_dwpw, int dtype, bool merge_heads = true);

struct HostTensor {
    const void* ptr;
    int64_t shape[4];
    int ndim;
};
// folds BN, packs into `blob` (size g.blob_floats); returns "" or an error message
std::string pack_weights(const Graph& g, const std::vector<std::string>& names,
                         const std::vector<HostTensor>& tensors, std::vector<float>& blob);

// ------------------------------------------------------------------ A/B and experiment switches
// Read from the environment ONCE PER HANDLE (lwp_create) and handed to the launchers through their parameter structs: a
// function-local `static getenv` would freeze whatever value the first launch of the process happened to see, so a test that
// toggles a switch between two engines would compare a kernel with itself.  0 / -1 = "not set": the launcher's own heuristic.
struct Tuning {
    int stem_ty = 0, stem_wl = -1, stem_debug = 0;                 // LWP_STEM_TY, LWP_STEM_WL, LWP_STEM_DEBUG
    int dw_tiled = -1, dw_cc = 0, dw_ph = 0;                       // LWP_DW_TILED (0 never | 1 always), LWP_DW_CC, LWP_DW_PH
    int gemm_wp = -1;                                              // LWP_GEMM_WP (-1 unset, else first digit)
    bool has_c3 = false, has_pw = false; int c3[3] = {0, 0, 0}, pw[3] = {0, 0, 0};   // LWP_GEMM_C3 / LWP_GEMM_PW = "BM,BN,KS"
    int dwpw_bm = 0, dwpw_nw = 0, dwpw_debug = 0, dwpwh_debug = 0; // LWP_DWPW_BM, LWP_DWPW_NW, LWP_DWPW_DEBUG, LWP_DWPWH_DEBUG
    int dwpw_pipe = -1;                                            // LWP_DWPW_PIPE (f32 software-pipelined 512-output fused kernel: 0 off, 1 forced)
    int dwpw_tiled_wgs = 0;                                        // LWP_DWPW_TILED_WGS (experiments: persistent workgroups per CU)
    int dwpw_tiled = -1;                                           // LWP_DWPW_TILED (front blocks, LDS-tiled fused kernel: 0 off, 1 forced)
    int dwpw_pp_grid = 0;                                          // LWP_DWPW_PP_GRID (tests: persistent grid size, to walk several rounds at small M)
    int dwpw_pp = -1;                                              // LWP_DWPW_PP (bf16 two-half-tile fused kernel: 0 off, 1 forced)
    int heads_rm = 0;                                              // LWP_HEADS_RM
    int gemmh_fold = -1;                                           // LWP_GEMMH_FOLD (0: never fold the next 1x1 into the 3x3's epilogue)
    int gemmh_persist = -1, gemmh_ar_off = 0, gemmh_ar_force = 0, gemmh_debug = 0;   // LWP_GEMMH_PERSIST, LWP_GEMMH_AR=0, LWP_GEMMH_AR_FORCE
    bool has_gemmh_ar = false, has_gemmh = false; int gemmh_ar[4] = {0, 0, 0, 0}, gemmh[4] = {0, 0, 0, 0};   // LWP_GEMMH_AR / LWP_GEMMH = "a,b,c,d"
    int upsample_tiled = -1;                                       // LWP_UPSAMPLE_TILED
    int max_frames_per_pass = 0;                                   // LWP_MAX_FRAMES_PER_PASS (tests: split batches as if the 2 GiB limit were reached earlier)
    int peak_tile = -1, pair_form = -1;                            // LWP_PEAK_TILE (find_peaks tile 0..3), LWP_PAIR_FORM (score_pairs variant)
    int dwpw_lds_pad_kb = 0;                                       // LWP_DWPW_LDS_PAD (KB of unused LDS per workgroup of the bf16 fused blocks: occupancy experiments)
    int host_fetch_dma = -1;                                       // LWP_HOST_FETCH_DMA ("1": host frames by hipMemcpyAsync instead of the fetch kernel)
    int ms_tx = 0;                                                 // LWP_MS_TX (8..40: tile width of the fused multi-scale kernel; 0 = the geometry's plan)
    int ms_vec = -1;                                               // LWP_MS_VEC ("0": the scalar fused multi-scale kernel)
    int ms_fused = -1;                                             // LWP_MS_FUSED ("0": multi-scale step as up-sample + resize kernels)
    int heads_f32_lds = -1;                                        // LWP_HEADS_F32_LDS ("0": the f32 stage heads above 4096 pixels as two GEMMs)
    int post_nchw = -1;                                            // LWP_POST_NCHW (f32: "0" = grouping reads the NHWC concat buffer in place)
    int heads_f32_max_m = 0;                                       // LWP_HEADS_F32_MAXM (tests: force the fused fp32 head pair at larger M)
};
Tuning tuning_from_env();
const Tuning& default_tuning();
// name of the kernel variant a launcher picked, written when the caller supplies a buffer (debug / profiling entry points)
constexpr int kVariantCap = 64;
#define LWP_VARIANT(p, ...) do { if ((p).variant) snprintf((p).variant, lwp::kVariantCap, __VA_ARGS__); } while (0)

// ------------------------------------------------------------------ kernel launch parameters
struct StemParams {       // (__restrict__: the weights stay scalar loads although the kernel loops over tiles and stores in between)
    const float* __restrict__ in;     // N x 3 x H x W
    const float* __restrict__ w;      // [27][32]  (ky, kx, ci) major, oc minor
    const float* __restrict__ bias;   // [32]
    float* __restrict__ out;          // N x Ho x Wo x 32
    int N, H, W, Ho, Wo;
    const float* zeros = nullptr;   // >= 16 bytes of zeros (source of out-of-image quads)
    const Tuning* tune = nullptr; char* variant = nullptr;
};
struct DwParams {
    const float* in; int in_ld;      // NHWC, row stride in_ld
    const float* w;                  // [9][C]
    const float* bias;               // [C]
    float* out; int out_ld;
    int N, Hi, Wi, Ho, Wo, C, stride, dil, act;
    const Tuning* tune = nullptr; char* variant = nullptr;
};
struct GemmParams {
    const float* in; int in_ld;      // window start already applied to the pointer
    const float* w;                  // [taps][cout_pad][cin_pad]
    const float* wf = nullptr;       // fp32 only: the same weights in fragment order [tap][k-step][32-ch tile][4][64][4]
    const float* bias;               // [cout_pad]
    float* out; int out_ld;
    const float* res; int res_ld;    // may be null
    float* out_nchw;                 // may be null: N x cout x H x W  (x out_split channels when out_split > 0)
    float* out_nchw2 = nullptr;      // merged heads: channels >= out_split, N x (cout - out_split) x H x W
    int out_split = 0;
    const float* zeros;              // >= 16 bytes of zeros (source of out-of-image taps)
    int N, H, W, cin_pad, cout, cout_pad, ks, dil, act;
    int debug = 0;                   // reserved for timing experiments
    const Tuning* tune = nullptr; char* variant = nullptr;
    // optional second conv fused into the epilogue (bf16 window-resident kernel only): out2 = act2(W2 . bf16(out) + b2), a 1x1 conv
    // 128 -> 128 over the tile the kernel has just produced; `out` itself is then NOT written.  *fused2 reports whether the
    // launcher took it (else the caller launches the second layer itself).
    const void* w2 = nullptr; const float* bias2 = nullptr; void* out2 = nullptr; int out2_ld = 0, act2 = 0;
    bool* fused2 = nullptr;
};
struct DwPwParams {
    const float* in; int in_ld;          // depthwise input, NHWC
    const float* dw_w;                   // [9][C] followed by [C] bias (contiguous)
    const float* pw_w;                   // fragment-packed pointwise weights [C/32][cout/32][4][64][4]
    const float* pw_b;                   // [cout]
    float* out; int out_ld;
    const float* res; int res_ld;        // may be null
    const float* zeros;
    int N, Hi, Wi, Ho, Wo, C, cout, stride, dil, act_dw, act_pw;
    int debug = 0;                       // ablation switches (LWP_DWPW_DEBUG): 1 skip phase 1, 2 skip B loads, 4 skip MFMAs, 8 no XCD tile remap
    const Tuning* tune = nullptr; char* variant = nullptr;
};
bool dwpw_supported(int C, int cout);
hipError_t launch_dwpw(const DwPwParams& p, hipStream_t s);
// LDS-tiled form of the front blocks at large batch (net_kernels_tiled.hip); *used = false: not applicable, the caller goes on
hipError_t try_dwpw_tiled_f32(const DwPwParams& p, hipStream_t s, bool* used);
hipError_t try_dwpw_tiled_bf16(const DwPwParams& p, hipStream_t s, bool* used);
// bf16 storage path (net_kernels_bf16.hip): same parameter structs, activation / packed-weight pointers are bf16
hipError_t launch_stem_bf16(const StemParams& p, hipStream_t s);
hipError_t launch_dwpw_bf16(const DwPwParams& p, hipStream_t s);
hipError_t launch_gemm_bf16(const GemmParams& p, hipStream_t s);
// a stage's head pair (with_mobilenet.py:32-45, 1x1 C -> hidden, ReLU, 1x1 hidden -> NH + NP) as ONE kernel: the hidden tensor
// (247 MB at batch 32 for the initial stage) never leaves the CU.  Weights in the plain [cout_pad][cin_pad] bf16 layout.
struct HeadsParams {
    const void* in; int in_ld;           // [M][128] bf16
    const void* w0; const float* b0;     // [hidden][128] bf16, [hidden]
    const void* w1; const float* b1;     // [64][hidden] bf16 (rows >= cout are zero), [64]
    void* out; int out_ld;               // bf16 NHWC window (the concat buffer at the heat/PAF channels)
    float* out_nchw; float* out_nchw2;   // may be null: stage outputs, split at out_split
    int out_split, N, H, W, hidden, cout;
    const Tuning* tune = nullptr; char* variant = nullptr;
};
bool heads_bf16_supported(int cin_pad, int hidden, int cout_pad);
hipError_t launch_heads_bf16(const HeadsParams& p, hipStream_t s);
// fp32 form for M <= 4096 (batch 1: 3772 pixels): 16-pixel workgroups, the hidden dimension split over the waves, fixed-order
// reduction of the partial outputs.  Pointers are f32 ([hidden][128], [64][hidden], NHWC f32 window).
bool heads_f32_supported(int cin_pad, int hidden, int cout_pad, int64_t M, const Tuning* tune = nullptr);
hipError_t launch_heads_f32(const HeadsParams& p, hipStream_t s);
hipError_t launch_nchw_from_nhwc_bf16(const void* src, int src_ld, float* dst, int N, int HW, int C, hipStream_t s);
hipError_t launch_stem(const StemParams& p, hipStream_t s);
hipError_t launch_dw(const DwParams& p, hipStream_t s);
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);

// ------------------------------------------------------------------ post-processing
struct MapView {          // a float32 map set addressed as base[n*ns + y*ys + x*xs + c*cs]
    const float* base;
    int64_t ns, ys, xs, cs;
    int h, w;             // size of the stored grid
};
struct PostCaps { int max_peaks = 2048, max_kpts = 128, max_conn = 4096, max_entries = 256; };

struct PostWorkspace {    // device buffers, sized for (N frames, caps)
    int N = 0;
    PostCaps caps;
    int* peak_count = nullptr;      // [N*18]
    uint32_t* peak_key = nullptr;   // [N*18*max_peaks]  (x << 16 | y)
    float* peak_val = nullptr;      // [N*18*max_peaks]
    int* kpt_count = nullptr;       // [N*18]
    int* kpt_xy = nullptr;          // [N*18*max_kpts*2]
    float* kpt_score = nullptr;     // [N*18*max_kpts]
    int* conn_count = nullptr;      // [N*19]
    int* conn_ij = nullptr;         // [N*19*max_conn]   (i << 16 | j)
    double* conn_ratio = nullptr;   // [N*19*max_conn]
    unsigned long long* flags = nullptr;  // [N*4]  0: overflow bits (1 peaks, 2 kpts, 4 conns/entries),
                                          //        1: min order of a pair whose mid-point test failed,
                                          //        2: min order of a pair whose mid-point test passed
    int* sel_count = nullptr;       // [N*19]  connections picked by the greedy matching
    int* seen = nullptr;            // [N*37]  debug: peaks per type nms_kernel saw [18], candidates per limb match_kernel saw [19] (the counters themselves are re-armed by assemble_kernel)
    int* sel_ij = nullptr;          // [N*19*max_kpts]
    double* sel_r = nullptr;        // [N*19*max_kpts]
    float* sel_sa = nullptr;        // [N*19*max_kpts] score of the connection's first key-point
    float* sel_sb = nullptr;        // [N*19*max_kpts] score of its second key-point
    double* entries_work = nullptr; // [N*max_entries*20] scratch when the entries do not fit LDS
    double* entries = nullptr;      // [N*max_entries*20]
    int* n_entries = nullptr;       // [N]
    double* kpts_out = nullptr;     // [N*18*max_kpts*4]
    void* result_block = nullptr;   // flags, kpts_out, entries, kpt_count, n_entries: one allocation, one D2H copy
    size_t result_bytes = 0;
};

hipError_t init_cubic_tables();
hipError_t launch_reset_ws(int N, PostWorkspace& ws, hipStream_t s);
hipError_t launch_upsample(const MapView& src, int N, int C, int ratio, float* dst, hipStream_t s, const Tuning* tune = nullptr);
// threshold + strict 4-neighbour maximum on the (virtually) up-sampled heat-maps; ratio == 1: src is already full-res
hipError_t launch_find_peaks(const MapView& heat, int N, int ntypes, int ratio, PostWorkspace& ws, hipStream_t s, const Tuning* tune = nullptr);
hipError_t launch_nms(int N, int ntypes, int Hfull, PostWorkspace& ws, hipStream_t s);
hipError_t launch_score_pairs(const MapView& paf, int N, int ratio, int demo, PostWorkspace& ws, hipStream_t s);
hipError_t launch_match(int N, PostWorkspace& ws, hipStream_t s);
hipError_t launch_assemble(int N, PostWorkspace& ws, hipStream_t s);
// u8 frame -> resized, normalised, padded CHW float32 (demo.py:59-64)
struct PreprocParams {
    const unsigned char* src; int Hs, Ws;            // HWC uint8, 3 channels
    const int *xi, *xw, *yi, *yw;                    // per destination index: 4 clamped source indices, 4 fixed-point weights
    int dh, dw, top, left, Hp, Wp;                   // scaled size, its offset inside the padded Hp x Wp frame
    double mean[3], scale;
    float pad_value[3];
    float* out;                                      // 3 x Hp x Wp
};
void build_resize_table_u8(int n_src, int n_dst, double inv_scale, std::vector<int>& idx, std::vector<int>& w);
hipError_t launch_preprocess_u8(const PreprocParams& p, hipStream_t s);
// N uint8 frames -> normalised float64 image, cubic resize by a ratio (f32 coefficients, f64 sums), pad, NCHW float32 (val.py:84-93)
struct PreScaleParams {
    const void* src; int N, Hs, Ws;                  // N x Hs x Ws x 3 uint8 (float32 when src_f32)
    bool src_f32 = false;
    const int *xi, *yi;                              // 4 clamped source indices per destination index
    const float *xw, *yw;                            // 4 float32 cubic coefficients per destination index
    int dh, dw, top, left, Hp, Wp;
    double mean[3], scale;
    float pad_value[3];
    float* out;                                      // N x 3 x Hp x Wp
};
void build_resize_table_ratio(int n_src, int n_dst, double ratio, std::vector<int>& idx, std::vector<float>& w);
hipError_t launch_preprocess_scaled(const PreScaleParams& p, hipStream_t s);
hipError_t launch_fetch_host(const void* src_host_mapped, void* dst, size_t bytes, hipStream_t s);
hipError_t launch_publish(int N, PostWorkspace& ws, void* host_block, hipStream_t s);   // used rows -> pinned host block
void build_resize_table(int n_src, int n_dst, std::vector<int>& idx, std::vector<float>& w);
void multiscale_fused_extent(const int* xi, const int* yi, int dst_h, int dst_w, int tx, int* uh_max, int* uw_max);
void multiscale_fused_plan(const int* xi, const int* yi, int dst_h, int dst_w, int R, int* tx_best, int* uh_max, int* uw_max);
hipError_t launch_multiscale_fused(const MapView& src, int N, int C, int ratio, int crop_top, int crop_left, const int* xi, const float* xw,
                                   const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum,
                                   int tx, int uh_max, int uw_max, hipStream_t s, bool* used);
void multiscale_fused_plan_v4(const int* xi, const int* yi, int dst_h, int dst_w, int R, int* tx_best, int* uh_max, int* uw_max);
hipError_t launch_multiscale_fused_v4(const MapView& src, int N, int C, int ratio, int crop_top, int crop_left, const int* xi, const float* xw,
                                      const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum,
                                      int tx, int uh_max, int uw_max, hipStream_t s, bool* used);
hipError_t launch_resize_accum(const float* src, int N, int Hs, int Ws, int C, int crop_top, int crop_left, const int* xi, const float* xw,
                               const int* yi, const float* yw, int dst_h, int dst_w, float divisor, int init, float* accum, hipStream_t s);
hipError_t launch_threshold_inplace(float* map, int64_t n, hipStream_t s);
hipError_t launch_nchw_from_nhwc(const float* src, int src_ld, float* dst, int N, int HW, int C, hipStream_t s);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: every launcher keeps one flag per device
// (a process may hold handles on several GPUs; the usual deployment is one process per GPU).
constexpr int kMaxDevices = 64;
struct LdsAttrOnce {
    bool done[kMaxDevices] = {};
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
        if (done[dev]) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done[dev] = true;
        return e;
    }
};

}  // namespace lwp
