// C-ABI entry points (include/lwpose.h).  Host orchestration only: buffer management, the static
// launch sequence of the layer graph on the handle's HIP stream, result fetch, and event timing.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lwp_internal.h"

using namespace lwp;

struct lwp_context {
    int device = 0;
    hipStream_t stream = nullptr;
    int dtype = LWP_F32;
    Graph g;
    bool fuse_heads = true;                // head pairs as one kernel (LWP_FUSE_HEADS=0 at lwp_create: two GEMMs)
    bool post_on_main = false;             // LWP_POST_STREAM=0 at lwp_create: grouping kernels follow the network on the same stream
    Tuning tune;                           // kernel-selection A/B switches, read from the environment at lwp_create
    std::vector<std::string> variants;     // per layer: the kernel variant its last launch picked (debug / profiling entry points only)
    bool record_variants = false;
    char variant_buf[kVariantCap] = {0};
    float* d_blob = nullptr;
    float* d_zeros = nullptr;
    bool weights_loaded = false;
    // activations for the current (N, H, W)
    int cur_N = 0, cur_H = 0, cur_W = 0;
    std::vector<float*> bufs;
    std::vector<size_t> buf_bytes;         // allocated size of each activation buffer (grow-only)
    float* d_in = nullptr; size_t d_in_bytes = 0;
    std::vector<float*> d_outs;            // NCHW staging for host outputs
    std::vector<size_t> d_outs_bytes;
    float* d_tmp = nullptr; size_t d_tmp_bytes = 0;     // generic device staging (upsample / extract / group)
    float* d_tmp2 = nullptr; size_t d_tmp2_bytes = 0;
    struct ResizeTab { int cw, ch, dw, dh; void* d; double ratio; int uh_max = 0, uw_max = 0, tx = 0, up_ratio = 0, tx4 = 0, uh4 = 0, uw4 = 0; };   // cubic resize tables of the multi-scale path, kept on the device (+ the fused kernel's tile extents)
    std::vector<ResizeTab> resize_tabs;                  // (a per-call upload went through SDMA queues: multi-ms stalls on some boxes)
    std::vector<ResizeTab> scale_tabs;                   // image-side tables of lwp_preprocess_scaled_u8: (W, H, dw, dh, ratio)
    unsigned char* d_imgs = nullptr; size_t d_imgs_bytes = 0;   // uint8 frame batch staging (host frames of the multi-scale path)
    float* d_img = nullptr; size_t d_img_bytes = 0;     // uint8 frame staging (pre-processing of host frames)
    // host frames travel through one of two pinned buffers (upload_host): a copy from pageable memory is staged by the runtime
    // anyway, at ~100 us per 720 KB frame and with the calling thread blocked until the DMA has finished
    void* pin_buf[2] = {nullptr, nullptr}; size_t pin_bytes[2] = {0, 0}; hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_busy[2] = {false, false};
    int pin_next = 0;
    float* d_pre_tab = nullptr; size_t d_pre_tab_bytes = 0;   // fixed-point resize tables, cached for (pre_H, pre_W, pre_net_h)
    int pre_H = 0, pre_W = 0, pre_net_h = 0;
    float* d_maps[2] = {nullptr, nullptr}; size_t d_maps_bytes[2] = {0, 0};   // bf16 path: f32 NCHW heat / PAF of the last stage
    // post-processing
    PostCaps caps;
    PostWorkspace ws;
    // pinned host staging for results
    void* h_stage = nullptr; size_t h_stage_bytes = 0;
    int last_N = 0;
    // pipelined streaming mode: two result slots, post-processing + fetch of frame k overlap the network of frame k+1
    struct Slot {
        PostWorkspace ws;
        float* maps[2] = {nullptr, nullptr}; size_t maps_bytes[2] = {0, 0};
        void* h_stage = nullptr; size_t h_stage_bytes = 0;
        hipEvent_t ev_maps = nullptr, ev_done = nullptr;
        int N = 0;
        bool pending = false;
    } slots[2];
    hipStream_t post_stream = nullptr;
    // caller-stream ordering (lwp_set_stream): work the caller queued on ITS stream is waited for with an event (no host
    // block), and the caller's stream is made to wait for the handle's results where they stay on the device
    hipStream_t caller_stream = nullptr;
    bool caller_ordered = false;
    bool hand_over = true;                 // lwp_set_stream mode 1: device results are handed to the caller's stream; mode 2: not
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_copy = nullptr;
    // per-launch profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_class;
    std::vector<int> ev_layer;             // index of the (first) layer a launch covers, -1: post-processing kernel
    int cur_layer = -1;
    size_t ev_used = 0;
    std::string err;
};

static std::string g_err;
static std::mutex g_mu;

static int fail(lwp_context* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else { std::lock_guard<std::mutex> l(g_mu); g_err = msg; }
    return code;
}
#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, LWP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

static int ensure_dev(lwp_context* h, float** p, size_t* have, size_t need) {
    if (*have >= need) return LWP_OK;
    if (*p) HIP_TRY(h, hipFree(*p));
    *p = nullptr; *have = 0;
    HIP_TRY(h, hipMalloc((void**)p, need));
    *have = need;
    return LWP_OK;
}

// caller's stream -> handle's stream: everything the caller has queued so far (producers of our inputs, consumers of the
// buffers we are about to overwrite) is ordered before our next launch.  No-op unless lwp_set_stream enabled it.
static int order_in(lwp_context* h) {
    if (!h->caller_ordered) return LWP_OK;
    // an idle caller stream has nothing to wait for: no event, no cross-queue barrier packet (with the three engine streams of
    // the batch-1 protocol a device-side wait on the shared default stream at every submit cost 20 % of the throughput)
    const hipError_t q = hipStreamQuery(h->caller_stream);
    if (q == hipSuccess) return LWP_OK;
    if (q != hipErrorNotReady) return fail(h, LWP_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    (void)hipGetLastError();                         // "not ready" is an answer, not a failure: keep it out of the launchers' error checks
    HIP_TRY(h, hipEventRecord(h->ev_in, h->caller_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_in, 0));
    return LWP_OK;
}
// handle's stream `from` -> caller's stream: what the caller queues next sees our device-side results.  Returns true in
// *ordered when the hand-over was done with an event (the legacy host synchronisation is then not needed).
static int order_out(lwp_context* h, hipStream_t from, bool* ordered) {
    *ordered = false;
    if (!h->caller_ordered) return LWP_OK;
    if (!h->hand_over) { *ordered = true; return LWP_OK; }        // mode 2: the result stays on the handle's stream (its next call consumes it)
    HIP_TRY(h, hipEventRecord(h->ev_out, from));
    HIP_TRY(h, hipStreamWaitEvent(h->caller_stream, h->ev_out, 0));
    *ordered = true;
    return LWP_OK;
}

extern "C" int lwp_version(void) { return 101; }

extern "C" int lwp_set_stream(lwp_handle h, void* caller_stream, int enable) {
    if (!h) return LWP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (enable && !h->ev_in) {
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    }
    h->caller_stream = (hipStream_t)caller_stream;
    h->caller_ordered = enable != 0;
    h->hand_over = enable != 2;
    return LWP_OK;
}

extern "C" int lwp_param_count(int nref, int C, int NH, int NP) {
    if (nref < 0 || C <= 0 || NH <= 0 || NP <= 0) return LWP_ERR_ARG;
    return (int)param_table(nref, C, NH, NP).size();
}

extern "C" int lwp_param_spec(int nref, int C, int NH, int NP, int index, char* name, int name_cap, int64_t shape[4],
                              int* ndim, int* role) {
    if (nref < 0 || C <= 0 || NH <= 0 || NP <= 0 || !name || !shape || !ndim || !role) return fail(nullptr, LWP_ERR_ARG, "bad argument");
    auto t = param_table(nref, C, NH, NP);
    if (index < 0 || index >= (int)t.size()) return fail(nullptr, LWP_ERR_ARG, "index out of range");
    if ((int)t[index].key.size() + 1 > name_cap) return fail(nullptr, LWP_ERR_ARG, "name buffer too small");
    std::strcpy(name, t[index].key.c_str());
    for (int d = 0; d < 4; ++d) shape[d] = t[index].shape[d];
    *ndim = t[index].ndim;
    *role = t[index].role;
    return LWP_OK;
}

extern "C" const char* lwp_last_error(lwp_handle h) {
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> l(g_mu);
    return g_err.c_str();
}

extern "C" int lwp_create(int device_id, int nref, int C, int NH, int NP, int dtype, lwp_handle* out) {
    if (!out) return fail(nullptr, LWP_ERR_ARG, "out is null");
    *out = nullptr;
    if (nref < 0 || C <= 0 || C % 32 || NH <= 0 || NP <= 0) return fail(nullptr, LWP_ERR_ARG, "bad network shape (num_channels must be a multiple of 32)");
    if (dtype != LWP_F32 && dtype != LWP_BF16) return fail(nullptr, LWP_ERR_ARG, "bad dtype");
    // the bf16 graph has no stand-alone depthwise kernel and its GEMM walks K in 64-channel steps: cpm.trunk must fuse
    // (C in {64, 128, 256, 512}); any other width would run f32 kernels on bf16-sized buffers
    if (dtype == LWP_BF16 && !(C % 64 == 0 && dwpw_supported(C, C)))
        return fail(nullptr, LWP_ERR_ARG, "bf16 path supports num_channels 64, 128, 256 or 512 only");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, LWP_ERR_NOGPU, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, LWP_ERR_ARG, "device_id out of range");
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, LWP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, LWP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(nullptr, LWP_ERR_NOGPU, std::string("this library is built for gfx950 only, found ") + prop.gcnArchName);
    lwp_context* h = new lwp_context();
    h->device = device_id;
    h->dtype = dtype;
    {
        const char* fe = getenv("LWP_FUSE_DWPW");   // "0" keeps depthwise and pointwise as separate launches (A/B, tests)
        const char* me = getenv("LWP_MERGE_HEADS");   // "0": separate heat / PAF head GEMMs (A/B)
        h->g = build_graph(nref, C, NH, NP, !(fe && fe[0] == '0'), dtype, !(me && me[0] == '0'));
        const char* he = getenv("LWP_FUSE_HEADS");
        h->fuse_heads = !(he && he[0] == '0');
        const char* pe = getenv("LWP_POST_STREAM");
        h->post_on_main = pe && pe[0] == '0';
        h->tune = tuning_from_env();
        h->variants.assign(h->g.layers.size(), std::string());
    }
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(nullptr, LWP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    e = init_cubic_tables();
    if (e != hipSuccess) { delete h; return fail(nullptr, LWP_ERR_HIP, std::string("init_cubic_tables: ") + hipGetErrorString(e)); }
    e = hipMalloc((void**)&h->d_blob, h->g.blob_floats * sizeof(float));
    if (e != hipSuccess) { delete h; return fail(nullptr, LWP_ERR_HIP, std::string("hipMalloc(blob): ") + hipGetErrorString(e)); }
    e = hipMalloc((void**)&h->d_zeros, 4096);
    if (e == hipSuccess) e = hipMemset(h->d_zeros, 0, 4096);
    if (e != hipSuccess) { delete h; return fail(nullptr, LWP_ERR_HIP, std::string("hipMalloc(zeros): ") + hipGetErrorString(e)); }
    h->bufs.assign(h->g.bufs.size(), nullptr);
    h->d_outs.assign(2 * (1 + nref), nullptr);
    h->d_outs_bytes.assign(2 * (1 + nref), 0);
    *out = h;
    return LWP_OK;
}

static void free_ws_obj(PostWorkspace& w) {
    // flags / kpt_count / n_entries / kpts_out / entries live in ONE allocation (result_block) so the fetch is one copy
    void* ptrs[] = {w.peak_count, w.peak_key, w.peak_val, w.kpt_xy, w.kpt_score, w.conn_count,
                    w.conn_ij, w.conn_ratio, w.result_block, w.sel_count, w.sel_ij, w.sel_r,
                    w.entries_work, w.sel_sa, w.sel_sb, w.seen};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    w = PostWorkspace();
}
static void free_ws(lwp_context* h) { free_ws_obj(h->ws); }

extern "C" int lwp_destroy(lwp_handle h) {
    if (!h) return LWP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (float* p : h->bufs) if (p) (void)hipFree(p);
    for (float* p : h->d_outs) if (p) (void)hipFree(p);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_tmp) (void)hipFree(h->d_tmp);
    if (h->d_tmp2) (void)hipFree(h->d_tmp2);
    for (float* p : h->d_maps) if (p) (void)hipFree(p);
    for (auto& rt : h->resize_tabs) if (rt.d) (void)hipFree(rt.d);
    for (auto& rt : h->scale_tabs) if (rt.d) (void)hipFree(rt.d);
    if (h->d_imgs) (void)hipFree(h->d_imgs);
    if (h->d_img) (void)hipFree(h->d_img);
    for (int k = 0; k < 2; ++k) {
        if (h->pin_buf[k]) (void)hipHostFree(h->pin_buf[k]);
        if (h->pin_ev[k]) (void)hipEventDestroy(h->pin_ev[k]);
    }
    if (h->d_pre_tab) (void)hipFree(h->d_pre_tab);
    if (h->d_blob) (void)hipFree(h->d_blob);
    if (h->d_zeros) (void)hipFree(h->d_zeros);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    free_ws(h);
    if (h->post_stream) (void)hipStreamSynchronize(h->post_stream);
    for (auto& sl : h->slots) {
        free_ws_obj(sl.ws);
        for (float* p : sl.maps) if (p) (void)hipFree(p);
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        if (sl.ev_maps) (void)hipEventDestroy(sl.ev_maps);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (h->post_stream && h->post_stream != h->stream) (void)hipStreamDestroy(h->post_stream);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : {h->ev_in, h->ev_out, h->ev_copy}) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return LWP_OK;
}

extern "C" int lwp_set_capacity(lwp_handle h, int max_peaks, int max_kpts, int max_conn, int max_entries) {
    if (!h) return LWP_ERR_ARG;
    if (max_peaks < 64 || max_peaks > 8192 || max_kpts < 1 || max_kpts > 1024 || max_conn < 1 || max_conn > (1 << 20) ||
        max_entries < 1 || max_entries > 65535)
        return fail(h, LWP_ERR_ARG, "capacity out of range (peaks 64..8192, kpts 1..1024, conns 1..2^20, entries 1..65535)");
    (void)hipSetDevice(h->device);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_ws(h);
    for (auto& sl : h->slots) { if (sl.pending) return fail(h, LWP_ERR_STATE, "pipeline slot pending"); free_ws_obj(sl.ws); }
    h->caps.max_peaks = max_peaks; h->caps.max_kpts = max_kpts; h->caps.max_conn = max_conn; h->caps.max_entries = max_entries;
    return LWP_OK;
}

extern "C" int lwp_load_weights(lwp_handle h, const char* const* names, const void* const* ptrs, const int64_t* shapes,
                                const int* ndims, int n) {
    if (!h || !names || !ptrs || !shapes || !ndims || n <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    std::vector<std::string> nm(n);
    std::vector<HostTensor> ts(n);
    for (int i = 0; i < n; ++i) {
        nm[i] = names[i] ? names[i] : "";
        ts[i].ptr = ptrs[i];
        ts[i].ndim = ndims[i];
        for (int d = 0; d < 4; ++d) ts[i].shape[d] = shapes[i * 4 + d];
    }
    std::vector<float> blob;
    std::string msg = pack_weights(h->g, nm, ts, blob);
    if (!msg.empty()) return fail(h, LWP_ERR_ARG, msg);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(h->d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    h->weights_loaded = true;
    return LWP_OK;
}

extern "C" int lwp_weights_blob_bytes(lwp_handle h, size_t* bytes) {
    if (!h || !bytes) return LWP_ERR_ARG;
    *bytes = h->g.blob_floats * sizeof(float);
    return LWP_OK;
}
extern "C" int lwp_weights_blob_export(lwp_handle h, void* dst, size_t bytes) {
    if (!h || !dst || bytes != h->g.blob_floats * sizeof(float)) return fail(h, LWP_ERR_ARG, "bad blob size");
    if (!h->weights_loaded) return fail(h, LWP_ERR_STATE, "weights not loaded");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(dst, h->d_blob, bytes, hipMemcpyDeviceToDevice));
    return LWP_OK;
}
extern "C" int lwp_weights_blob_import(lwp_handle h, const void* src, size_t bytes) {
    if (!h || !src || bytes != h->g.blob_floats * sizeof(float)) return fail(h, LWP_ERR_ARG, "bad blob size");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(h->d_blob, src, bytes, hipMemcpyDeviceToDevice));
    h->weights_loaded = true;
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- buffers
static void level_dims(int H, int W, int level, int* h, int* w) {
    int hh = H, ww = W;
    for (int l = 0; l < level; ++l) { hh = (hh - 1) / 2 + 1; ww = (ww - 1) / 2 + 1; }
    *h = hh; *w = ww;
}

static int ensure_activations(lwp_context* h, int N, int H, int W) {
    if (h->cur_N == N && h->cur_H == H && h->cur_W == W) return LWP_OK;
    // Buffers only ever GROW: a caller that alternates shapes (the three scales of val.infer) would otherwise free and
    // re-allocate gigabytes on every call — hipFree / hipMalloc of that size took up to 600 ms on some boxes.
    if (h->buf_bytes.size() != h->bufs.size()) h->buf_bytes.assign(h->bufs.size(), 0);
    bool synced = false;
    for (size_t i = 0; i < h->bufs.size(); ++i) {
        int bh, bw;
        level_dims(H, W, h->g.bufs[i].level, &bh, &bw);
        const size_t bytes = (size_t)N * bh * bw * h->g.bufs[i].channels * (h->dtype == LWP_BF16 ? 2 : 4);
        if (bytes > h->buf_bytes[i]) {
            if (!synced) { HIP_TRY(h, hipStreamSynchronize(h->stream)); synced = true; }
            if (h->bufs[i]) { HIP_TRY(h, hipFree(h->bufs[i])); h->bufs[i] = nullptr; h->buf_bytes[i] = 0; }
            HIP_TRY(h, hipMalloc((void**)&h->bufs[i], bytes));
            h->buf_bytes[i] = bytes;
        }
        // the concat buffer's pad channels are read (with zero weights) but never written: they must hold finite values,
        // and a re-used buffer may hold anything at the new geometry's offsets -> cleared, stream-ordered.  Every other
        // buffer is completely overwritten by its producer before it is read.
        if (h->g.bufs[i].has_pad) HIP_TRY(h, hipMemsetAsync(h->bufs[i], 0, bytes, h->stream));
    }
    h->cur_N = N; h->cur_H = H; h->cur_W = W;
    return LWP_OK;
}

static int ensure_ws_obj(lwp_context* h, PostWorkspace& w, int N, hipStream_t stream) {
    if (w.N >= N && w.peak_count) return LWP_OK;
    HIP_TRY(h, hipStreamSynchronize(stream));
    free_ws_obj(w);
    w.caps = h->caps;
    const PostCaps& c = h->caps;
#define WS_ALLOC(field, count, type) HIP_TRY(h, hipMalloc((void**)&w.field, (size_t)(count) * sizeof(type)))
    WS_ALLOC(peak_count, N * 18, int);
    WS_ALLOC(peak_key, (size_t)N * 18 * c.max_peaks, uint32_t);
    WS_ALLOC(peak_val, (size_t)N * 18 * c.max_peaks, float);
    WS_ALLOC(kpt_xy, (size_t)N * 18 * c.max_kpts * 2, int);
    WS_ALLOC(kpt_score, (size_t)N * 18 * c.max_kpts, float);
    WS_ALLOC(conn_count, N * 19, int);
    WS_ALLOC(conn_ij, (size_t)N * 19 * c.max_conn, int);
    WS_ALLOC(conn_ratio, (size_t)N * 19 * c.max_conn, double);
    {   // result block: [flags N*4 u64][kpts_out N*18*kcap*4 f64][entries N*ecap*20 f64][kpt_count N*18 i32][n_entries N i32]
        const size_t b_fl = (size_t)N * 4 * 8, b_k = (size_t)N * 18 * c.max_kpts * 4 * 8, b_e = (size_t)N * c.max_entries * 20 * 8;
        const size_t b_cnt = (size_t)N * 18 * 4, b_ne = (size_t)N * 4;
        w.result_bytes = b_fl + b_k + b_e + b_cnt + b_ne;
        HIP_TRY(h, hipMalloc((void**)&w.result_block, w.result_bytes));
        char* q = (char*)w.result_block;
        w.flags = (unsigned long long*)q; q += b_fl;
        w.kpts_out = (double*)q; q += b_k;
        w.entries = (double*)q; q += b_e;
        w.kpt_count = (int*)q; q += b_cnt;
        w.n_entries = (int*)q;
    }
    WS_ALLOC(entries_work, (size_t)N * c.max_entries * 20, double);
    WS_ALLOC(sel_count, N * 19, int);
    WS_ALLOC(seen, N * 37, int);
    WS_ALLOC(sel_ij, (size_t)N * 19 * c.max_kpts, int);
    WS_ALLOC(sel_r, (size_t)N * 19 * c.max_kpts, double);
    WS_ALLOC(sel_sa, (size_t)N * 19 * c.max_kpts, float);
    WS_ALLOC(sel_sb, (size_t)N * 19 * c.max_kpts, float);
#undef WS_ALLOC
    w.N = N;
    HIP_TRY(h, launch_reset_ws(N, w, stream));
    return LWP_OK;
}
static int ensure_ws(lwp_context* h, int N) { return ensure_ws_obj(h, h->ws, N, h->stream); }

static int ensure_host_stage(lwp_context* h, size_t bytes) {
    if (h->h_stage_bytes >= bytes) return LWP_OK;
    if (h->h_stage) HIP_TRY(h, hipHostFree(h->h_stage));
    h->h_stage = nullptr; h->h_stage_bytes = 0;
    HIP_TRY(h, hipHostMalloc(&h->h_stage, bytes, hipHostMallocDefault));
    h->h_stage_bytes = bytes;
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- profiling hooks
static int prof_begin(lwp_context* h, int kclass) {
    if (!h->profiling) return LWP_OK;
    if (h->ev_used + 2 > h->ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
        h->ev_class.resize(h->ev.size() / 2);
        h->ev_layer.resize(h->ev.size() / 2);
    }
    h->ev_class[h->ev_used / 2] = kclass;
    h->ev_layer[h->ev_used / 2] = h->cur_layer;
    HIP_TRY(h, hipEventRecord(h->ev[h->ev_used], h->stream));
    return LWP_OK;
}
static int prof_end(lwp_context* h) {
    if (!h->profiling) return LWP_OK;
    HIP_TRY(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
    h->ev_used += 2;
    return LWP_OK;
}
#define LAUNCH(h, kclass, expr)                    \
    do {                                           \
        int rc_ = prof_begin(h, kclass);           \
        if (rc_) return rc_;                       \
        HIP_TRY(h, expr);                          \
        rc_ = prof_end(h);                         \
        if (rc_) return rc_;                       \
    } while (0)

// ---------------------------------------------------------------------------------------------- batch split
// The kernels address a tensor with 32-bit byte offsets (buffer loads), so one launch sequence takes at most as many frames
// as keep EVERY tensor of the pass below 2 GiB.  Larger batches are processed in equal chunks inside the entry points — the
// reference's forward takes any N (models/with_mobilenet.py:114) and so does this library.
static int frames_per_pass(lwp_context* h, int N, int H, int W) {
    size_t per = (size_t)3 * H * W * sizeof(float);
    for (const BufSpec& b : h->g.bufs) {
        int bh, bw;
        level_dims(H, W, b.level, &bh, &bw);
        per = std::max(per, (size_t)bh * bw * b.channels * (h->dtype == LWP_BF16 ? 2 : 4));
    }
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);
    per = std::max(per, (size_t)fh * fw * std::max(h->g.NH, h->g.NP) * sizeof(float));
    const size_t lim = ((size_t)1 << 31) - 4096;
    int64_t nmax = std::max<int64_t>(1, (int64_t)(lim / per));
    if (h->tune.max_frames_per_pass > 0) nmax = std::min<int64_t>(nmax, h->tune.max_frames_per_pass);   // tests
    if (N <= nmax) return N;
    const int64_t chunks = (N + nmax - 1) / nmax;
    return (int)((N + chunks - 1) / chunks);
}

// the per-frame arrays of a post-processing workspace, seen from frame f0 on (every kernel indexes them by frame)
static PostWorkspace ws_frames(const PostWorkspace& w, int f0) {
    if (f0 == 0) return w;
    PostWorkspace v = w;
    const PostCaps& c = w.caps;
    const size_t f = (size_t)f0;
    v.N = w.N - f0;
    v.peak_count += f * 18; v.peak_key += f * 18 * c.max_peaks; v.peak_val += f * 18 * c.max_peaks;
    v.kpt_count += f * 18; v.kpt_xy += f * 18 * c.max_kpts * 2; v.kpt_score += f * 18 * c.max_kpts;
    v.conn_count += f * 19; v.conn_ij += f * 19 * c.max_conn; v.conn_ratio += f * 19 * c.max_conn;
    v.flags += f * 4;
    v.sel_count += f * 19; v.sel_ij += f * 19 * c.max_kpts; v.sel_r += f * 19 * c.max_kpts;
    v.seen += f * 37;
    v.sel_sa += f * 19 * c.max_kpts; v.sel_sb += f * 19 * c.max_kpts;
    v.entries_work += f * c.max_entries * 20; v.entries += f * c.max_entries * 20;
    v.n_entries += f; v.kpts_out += f * 18 * c.max_kpts * 4;
    return v;
}

// ---------------------------------------------------------------------------------------------- forward
// element-addressed window of an activation buffer (f32 or bf16 storage)
static inline float* buf_at(lwp_context* h, const BufRef& r) {
    return (float*)((char*)h->bufs[r.buf] + (size_t)r.coff * (h->dtype == LWP_BF16 ? 2 : 4));
}

static int enqueue_layer(lwp_context* h, const Layer& l, const float* d_in, int N, int H, int W, float* const* d_outs_nchw,
                         const Layer* fold = nullptr, bool* folded = nullptr) {
    const Graph& g = h->g;
    const bool h16 = h->dtype == LWP_BF16;
    const float* wts = h->d_blob + l.w_off;
    const float* bias = h->d_blob + l.b_off;
    int dh, dw;
    level_dims(H, W, g.bufs[l.dst.buf].level, &dh, &dw);
    float* dst = buf_at(h, l.dst);
    char* vb = h->record_variants ? h->variant_buf : nullptr;
    if (vb) vb[0] = 0;
    if (l.kind == L_STEM) {
        StemParams p{d_in, wts, bias, dst, N, H, W, dh, dw, h->d_zeros};
        p.tune = &h->tune; p.variant = vb;
        LAUNCH(h, KC_STEM, h16 ? launch_stem_bf16(p, h->stream) : launch_stem(p, h->stream));
    } else if (l.kind == L_DWPW) {
        int sh, sw;
        level_dims(H, W, g.bufs[l.src.buf].level, &sh, &sw);
        DwPwParams p;
        p.in = buf_at(h, l.src); p.in_ld = l.src.ld;
        p.dw_w = wts; p.pw_w = h->d_blob + l.w2_off; p.pw_b = h->d_blob + l.b2_off;
        p.out = dst; p.out_ld = l.dst.ld;
        p.res = l.res.buf >= 0 ? buf_at(h, l.res) : nullptr; p.res_ld = l.res.ld;
        p.zeros = h->d_zeros;
        p.N = N; p.Hi = sh; p.Wi = sw; p.Ho = dh; p.Wo = dw; p.C = l.cin; p.cout = l.cout;
        p.stride = l.stride; p.dil = l.dil; p.act_dw = l.act; p.act_pw = l.act2;
        p.tune = &h->tune; p.variant = vb;
        LAUNCH(h, KC_PW, h16 ? launch_dwpw_bf16(p, h->stream) : launch_dwpw(p, h->stream));
    } else if (l.kind == L_DW) {
        int sh, sw;
        level_dims(H, W, g.bufs[l.src.buf].level, &sh, &sw);
        DwParams p{buf_at(h, l.src), l.src.ld, wts, bias, dst, l.dst.ld, N, sh, sw, dh, dw, l.cin, l.stride, l.dil, l.act};
        p.tune = &h->tune; p.variant = vb;
        LAUNCH(h, KC_DW, launch_dw(p, h->stream));
    } else {
        GemmParams p;
        p.in = buf_at(h, l.src); p.in_ld = l.src.ld;
        p.w = wts; p.bias = bias;
        p.wf = h16 ? nullptr : h->d_blob + l.w2_off;
        p.out = dst; p.out_ld = l.dst.ld;
        p.res = l.res.buf >= 0 ? buf_at(h, l.res) : nullptr; p.res_ld = l.res.ld;
        p.out_nchw = (l.out_index >= 0 && d_outs_nchw) ? d_outs_nchw[l.out_index] : nullptr;
        p.out_nchw2 = (l.out_index2 >= 0 && d_outs_nchw) ? d_outs_nchw[l.out_index2] : nullptr;
        p.out_split = l.out_split;
        p.zeros = h->d_zeros;
        p.N = N; p.H = dh; p.W = dw;
        p.cin_pad = l.cin_pad; p.cout = l.cout; p.cout_pad = l.cout_pad; p.ks = l.ks; p.dil = l.dil; p.act = l.act;
        p.tune = &h->tune; p.variant = vb;
        if (fold && folded && h16) {                    // the next 1x1 rides in this launch's epilogue if the launcher takes it
            p.w2 = h->d_blob + fold->w_off; p.bias2 = h->d_blob + fold->b_off;
            p.out2 = buf_at(h, fold->dst); p.out2_ld = fold->dst.ld; p.act2 = fold->act;
            p.fused2 = folded;
        }
        LAUNCH(h, l.ks == 1 ? KC_PW : KC_C3, h16 ? launch_gemm_bf16(p, h->stream) : launch_gemm(p, h->stream));
    }
    if (vb && h->cur_layer >= 0 && h->cur_layer < (int)h->variants.size()) h->variants[h->cur_layer] = vb;
    return LWP_OK;
}

// a stage's merged head pair (".heads.0" 1x1 C -> hidden + ReLU, ".heads.1" 1x1 hidden -> NH + NP) runs as one kernel
// that keeps the hidden tensor on the CU.  LWP_FUSE_HEADS=0 launches the two GEMMs (A/B, tests).
static bool heads_pair_fusable(lwp_context* h, size_t i, int64_t M) {
    if (!h->fuse_heads) return false;
    const std::vector<Layer>& ls = h->g.layers;
    if (i + 1 >= ls.size()) return false;
    const Layer& a = ls[i];
    const Layer& b = ls[i + 1];
    auto ends_with = [](const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; };
    if (!ends_with(a.name, ".heads.0") || !ends_with(b.name, ".heads.1")) return false;
    if (a.kind != L_GEMM || b.kind != L_GEMM || a.ks != 1 || b.ks != 1 || a.act != ACT_RELU || b.act != ACT_NONE) return false;
    if (a.res.buf >= 0 || b.res.buf >= 0 || a.out_index >= 0) return false;
    if (b.src.buf != a.dst.buf || b.src.coff != a.dst.coff || b.cin_pad != a.cout_pad || a.cout != a.cout_pad) return false;
    return h->dtype == LWP_BF16 ? heads_bf16_supported(a.cin_pad, a.cout_pad, b.cout_pad) : heads_f32_supported(a.cin_pad, a.cout_pad, b.cout_pad, M, &h->tune);
}

static int enqueue_heads_pair(lwp_context* h, const Layer& a, const Layer& b, int N, int H, int W, float* const* d_outs_nchw) {
    int dh, dw;
    level_dims(H, W, h->g.bufs[b.dst.buf].level, &dh, &dw);
    HeadsParams p;
    p.in = buf_at(h, a.src); p.in_ld = a.src.ld;
    p.w0 = h->d_blob + a.w_off; p.b0 = h->d_blob + a.b_off;
    p.w1 = h->d_blob + b.w_off; p.b1 = h->d_blob + b.b_off;
    p.out = buf_at(h, b.dst); p.out_ld = b.dst.ld;
    p.out_nchw = (b.out_index >= 0 && d_outs_nchw) ? d_outs_nchw[b.out_index] : nullptr;
    p.out_nchw2 = (b.out_index2 >= 0 && d_outs_nchw) ? d_outs_nchw[b.out_index2] : nullptr;
    p.out_split = b.out_split;
    p.N = N; p.H = dh; p.W = dw; p.hidden = a.cout_pad; p.cout = b.cout;
    char* vb = h->record_variants ? h->variant_buf : nullptr;
    if (vb) vb[0] = 0;
    p.tune = &h->tune; p.variant = vb;
    LAUNCH(h, KC_PW, h->dtype == LWP_BF16 ? launch_heads_bf16(p, h->stream) : launch_heads_f32(p, h->stream));
    if (vb && h->cur_layer >= 0 && h->cur_layer + 1 < (int)h->variants.size()) { h->variants[h->cur_layer] = vb; h->variants[h->cur_layer + 1] = vb; }
    return LWP_OK;
}

// enqueue every layer on the handle's stream.  d_outs_nchw: 2*(1+nref) device pointers or null.
static int enqueue_forward(lwp_context* h, const float* d_in, int N, int H, int W, float* const* d_outs_nchw,
                           int max_layers = 1 << 30) {
    const std::vector<Layer>& ls = h->g.layers;
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);
    const int64_t M3 = (int64_t)N * fh * fw;                 // pixels of the stride-8 maps the heads work on
    for (size_t i = 0; i < ls.size() && (int)i < max_layers; ++i) {
        h->cur_layer = (int)i;
        if ((int)i + 1 < max_layers && heads_pair_fusable(h, i, M3)) {
            int rc = enqueue_heads_pair(h, ls[i], ls[i + 1], N, H, W, d_outs_nchw);
            if (rc) { h->cur_layer = -1; return rc; }
            ++i;
            continue;
        }
        // bf16: a dense 3x3 whose output feeds ONLY the next layer, a 1x1 128 -> 128 (refinement block b's last conv and block b+1's
        // `initial`, with_mobilenet.py:57-60), hands that layer to its own epilogue when the window-resident kernel runs
        const Layer* fold = nullptr;
        if (h->dtype == LWP_BF16 && (int)i + 1 < max_layers && i + 1 < ls.size()) {
            const Layer& a = ls[i];
            const Layer& b = ls[i + 1];
            const bool later_reader = [&]() {
                for (size_t k = i + 2; k < ls.size(); ++k) {
                    if ((ls[k].src.buf == a.dst.buf && ls[k].src.coff == a.dst.coff) || (ls[k].res.buf == a.dst.buf)) return true;
                    if (ls[k].dst.buf == a.dst.buf) return false;         // overwritten before anyone else reads it
                }
                return false;
            }();
            if (a.kind == L_GEMM && a.ks == 3 && b.kind == L_GEMM && b.ks == 1 && b.src.buf == a.dst.buf && b.src.coff == a.dst.coff &&
                b.src.ld == a.dst.ld && b.res.buf < 0 && b.out_index < 0 && a.out_index < 0 && b.blocks.empty() && a.cout == 128 &&
                b.cin_pad == 128 && b.cout_pad == 128 && b.cout == 128 && !later_reader)
                fold = &b;
        }
        bool folded = false;
        int rc = enqueue_layer(h, ls[i], d_in, N, H, W, d_outs_nchw, fold, &folded);
        if (rc) { h->cur_layer = -1; return rc; }
        if (folded) {
            if (h->record_variants && i + 1 < h->variants.size()) h->variants[i + 1] = h->variants[i];
            ++i;
        }
    }
    h->cur_layer = -1;
    return LWP_OK;
}

static int check_frame_shape(lwp_context* h, int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return fail(h, LWP_ERR_ARG, "bad frame shape");
    if (H < 8 || W < 8) return fail(h, LWP_ERR_ARG, "frame too small (H, W >= 8)");
    if (!h->weights_loaded) return fail(h, LWP_ERR_STATE, "weights not loaded (call lwp_load_weights first)");
    return LWP_OK;
}

static int stage_input(lwp_context* h, const float* in, int in_mem, size_t bytes, const float** d_in) {
    if (in_mem == LWP_MEM_DEVICE) { *d_in = in; return LWP_OK; }
    int rc = ensure_dev(h, &h->d_in, &h->d_in_bytes, bytes);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_in, in, bytes, hipMemcpyHostToDevice, h->stream));
    *d_in = h->d_in;
    return LWP_OK;
}

extern "C" int lwp_forward(lwp_handle h, const float* in, int in_mem, int N, int H, int W, float* const* outs, int out_mem) {
    if (!h || !in || !outs) return fail(h, LWP_ERR_ARG, "null argument");
    int rc = check_frame_shape(h, N, H, W);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const int Nc = frames_per_pass(h, N, H, W);          // frames per launch sequence (N unless a tensor would reach 2 GiB)
    rc = ensure_activations(h, Nc, H, W);
    if (rc) return rc;
    if (in_mem == LWP_MEM_DEVICE || out_mem == LWP_MEM_DEVICE) { rc = order_in(h); if (rc) return rc; }
    const float* d_in = nullptr;
    rc = stage_input(h, in, in_mem, (size_t)N * 3 * H * W * sizeof(float), &d_in);
    if (rc) return rc;
    const int nout = 2 * (1 + h->g.nref);
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);               // three stride-2 stages: out = (in - 1) / 2 + 1 each
    std::vector<float*> d_outs(nout), d_chunk(nout);
    for (int i = 0; i < nout; ++i) {
        if (!outs[i]) return fail(h, LWP_ERR_ARG, "null output pointer");
        if (out_mem == LWP_MEM_DEVICE) { d_outs[i] = outs[i]; continue; }
        const size_t bytes = (size_t)N * (i % 2 ? h->g.NP : h->g.NH) * fh * fw * sizeof(float);
        rc = ensure_dev(h, &h->d_outs[i], &h->d_outs_bytes[i], bytes);
        if (rc) return rc;
        d_outs[i] = h->d_outs[i];
    }
    for (int f0 = 0; f0 < N; f0 += Nc) {
        const int n = std::min(Nc, N - f0);
        for (int i = 0; i < nout; ++i) d_chunk[i] = d_outs[i] + (size_t)f0 * (i % 2 ? h->g.NP : h->g.NH) * fh * fw;
        if (n != h->cur_N) { rc = ensure_activations(h, n, H, W); if (rc) return rc; }      // ragged last chunk
        rc = enqueue_forward(h, d_in + (size_t)f0 * 3 * H * W, n, H, W, d_chunk.data());
        if (rc) return rc;
    }
    if (out_mem == LWP_MEM_HOST) {
        for (int i = 0; i < nout; ++i) {
            const size_t bytes = (size_t)N * (i % 2 ? h->g.NP : h->g.NH) * fh * fw * sizeof(float);
            HIP_TRY(h, hipMemcpyAsync(outs[i], d_outs[i], bytes, hipMemcpyDeviceToHost, h->stream));
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else {
        bool ordered = false;
        rc = order_out(h, h->stream, &ordered);          // the caller's stream waits for the outputs; without lwp_set_stream the
        if (rc) return rc;                               // caller synchronises (lwp_synchronize) before touching them
    }
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- upsample
extern "C" int lwp_upsample(lwp_handle h, const float* src, int src_mem, int N, int C, int hs, int ws, int ratio, float* dst, int dst_mem) {
    if (!h || !src || !dst || N <= 0 || C <= 0 || hs <= 0 || ws <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    if (ratio != 4 && ratio != 8) return fail(h, LWP_ERR_ARG, "upsample ratio must be 4 or 8");
    HIP_TRY(h, hipSetDevice(h->device));
    if (src_mem == LWP_MEM_DEVICE || dst_mem == LWP_MEM_DEVICE) { int rc0 = order_in(h); if (rc0) return rc0; }
    const size_t sb = (size_t)N * C * hs * ws * sizeof(float), db = sb * ratio * ratio;
    const float* d_src = src;
    if (src_mem == LWP_MEM_HOST) {
        int rc = ensure_dev(h, &h->d_tmp, &h->d_tmp_bytes, sb);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp, src, sb, hipMemcpyHostToDevice, h->stream));
        d_src = h->d_tmp;
    }
    float* d_dst = dst;
    if (dst_mem == LWP_MEM_HOST) {
        int rc = ensure_dev(h, &h->d_tmp2, &h->d_tmp2_bytes, db);
        if (rc) return rc;
        d_dst = h->d_tmp2;
    }
    MapView v{d_src, (int64_t)C * hs * ws, (int64_t)ws, 1, (int64_t)hs * ws, hs, ws};
    LAUNCH(h, KC_POST, launch_upsample(v, N, C, ratio, d_dst, h->stream, &h->tune));
    if (dst_mem == LWP_MEM_HOST) {
        HIP_TRY(h, hipMemcpyAsync(dst, d_dst, db, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else {
        bool ordered = false;
        int rc1 = order_out(h, h->stream, &ordered);
        if (rc1) return rc1;
    }
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- multi-scale accumulate
extern "C" int lwp_multiscale_accumulate(lwp_handle h, const float* maps, int maps_mem, int N, int C, int hs, int ws, int up_ratio,
                                         const int* pad, int dst_h, int dst_w, int n_scales, float* accum, int accum_mem, int init) {
    if (!h || !maps || !pad || !accum || N <= 0 || C <= 0 || hs <= 0 || ws <= 0 || dst_h <= 0 || dst_w <= 0 || n_scales <= 0)
        return fail(h, LWP_ERR_ARG, "bad argument");
    if (up_ratio != 4 && up_ratio != 8) return fail(h, LWP_ERR_ARG, "upsample ratio must be 4 or 8");
    const int Hs = hs * up_ratio, Ws = ws * up_ratio;
    const int ch = Hs - pad[0] - pad[2], cw = Ws - pad[1] - pad[3];
    if (pad[0] < 0 || pad[1] < 0 || pad[2] < 0 || pad[3] < 0 || ch <= 0 || cw <= 0) return fail(h, LWP_ERR_ARG, "bad crop");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t sb = (size_t)N * C * hs * ws * sizeof(float), ub = sb * up_ratio * up_ratio, ab = (size_t)N * dst_h * dst_w * C * sizeof(float);
    const float* d_src = maps;
    int rc;
    if (maps_mem == LWP_MEM_DEVICE || accum_mem == LWP_MEM_DEVICE) { rc = order_in(h); if (rc) return rc; }
    if (maps_mem == LWP_MEM_HOST) {
        rc = ensure_dev(h, &h->d_tmp, &h->d_tmp_bytes, sb);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp, maps, sb, hipMemcpyHostToDevice, h->stream));
        d_src = h->d_tmp;
    }
    float* d_acc = accum;
    if (accum_mem == LWP_MEM_HOST) {
        rc = ensure_dev(h, &h->d_maps[0], &h->d_maps_bytes[0], ab);
        if (rc) return rc;
        if (!init) HIP_TRY(h, hipMemcpyAsync(h->d_maps[0], accum, ab, hipMemcpyHostToDevice, h->stream));
        d_acc = h->d_maps[0];
    }
    // per-geometry tables, uploaded once (blocking copy) and kept: steady-state calls issue no host->device copy
    const size_t nx = (size_t)dst_w * 4, ny = (size_t)dst_h * 4;
    void* d_tabs = nullptr;
    int uh_max = 0, uw_max = 0, ms_tx = 0, tx4 = 0, uh4 = 0, uw4 = 0;
    for (const auto& rt : h->resize_tabs)
        if (rt.cw == cw && rt.ch == ch && rt.dw == dst_w && rt.dh == dst_h && rt.up_ratio == up_ratio) {
            d_tabs = rt.d; uh_max = rt.uh_max; uw_max = rt.uw_max; ms_tx = rt.tx; tx4 = rt.tx4; uh4 = rt.uh4; uw4 = rt.uw4; break;
        }
    if (!d_tabs) {
        std::vector<int> xi, yi;
        std::vector<float> xw, yw;
        build_resize_table(cw, dst_w, xi, xw);
        build_resize_table(ch, dst_h, yi, yw);
        if (h->tune.ms_tx >= 8 && h->tune.ms_tx <= 40) {   // LWP_MS_TX: a forced tile width (tests, A/B)
            ms_tx = h->tune.ms_tx;
            multiscale_fused_extent(xi.data(), yi.data(), dst_h, dst_w, ms_tx, &uh_max, &uw_max);
            tx4 = ms_tx; uh4 = uh_max; uw4 = uw_max;
        } else {
            multiscale_fused_plan(xi.data(), yi.data(), dst_h, dst_w, up_ratio, &ms_tx, &uh_max, &uw_max);
            multiscale_fused_plan_v4(xi.data(), yi.data(), dst_h, dst_w, up_ratio, &tx4, &uh4, &uw4);
        }
        if (h->resize_tabs.size() >= 16) {               // bounded: drop the oldest geometry
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            (void)hipFree(h->resize_tabs.front().d);
            h->resize_tabs.erase(h->resize_tabs.begin());
        }
        HIP_TRY(h, hipMalloc(&d_tabs, (nx + ny) * 8));
        char* t0 = (char*)d_tabs;
        HIP_TRY(h, hipMemcpy(t0, xi.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 4, xw.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 8, yi.data(), ny * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 8 + ny * 4, yw.data(), ny * 4, hipMemcpyHostToDevice));
        h->resize_tabs.push_back({cw, ch, dst_w, dst_h, d_tabs, 0.0, uh_max, uw_max, ms_tx, up_ratio, tx4, uh4, uw4});
    }
    char* t = (char*)d_tabs;
    int* d_xi = (int*)t; t += nx * 4;
    float* d_xw = (float*)t; t += nx * 4;
    int* d_yi = (int*)t; t += ny * 4;
    float* d_yw = (float*)t;
    MapView v{d_src, (int64_t)C * hs * ws, (int64_t)ws, 1, (int64_t)hs * ws, hs, ws};
    bool fused = false;
    if (h->tune.ms_fused != 0 && h->tune.ms_vec != 0) {      // four channels per lane (LWP_MS_VEC=0: the scalar fused kernel)
        LAUNCH(h, KC_POST, launch_multiscale_fused_v4(v, N, C, up_ratio, pad[0], pad[1], d_xi, d_xw, d_yi, d_yw, dst_h, dst_w, (float)n_scales, init ? 1 : 0,
                                                      d_acc, tx4, uh4, uw4, h->stream, &fused));
    }
    if (!fused && h->tune.ms_fused != 0) {                   // LWP_MS_FUSED=0: the two-kernel form (A/B, tests)
        LAUNCH(h, KC_POST, launch_multiscale_fused(v, N, C, up_ratio, pad[0], pad[1], d_xi, d_xw, d_yi, d_yw, dst_h, dst_w, (float)n_scales, init ? 1 : 0,
                                                   d_acc, ms_tx, uh_max, uw_max, h->stream, &fused));
    }
    if (!fused) {
        rc = ensure_dev(h, &h->d_tmp2, &h->d_tmp2_bytes, ub);
        if (rc) return rc;
        LAUNCH(h, KC_POST, launch_upsample(v, N, C, up_ratio, h->d_tmp2, h->stream, &h->tune));
        LAUNCH(h, KC_POST, launch_resize_accum(h->d_tmp2, N, Hs, Ws, C, pad[0], pad[1], d_xi, d_xw, d_yi, d_yw, dst_h, dst_w, (float)n_scales, init ? 1 : 0, d_acc, h->stream));
    }
    if (accum_mem == LWP_MEM_HOST) HIP_TRY(h, hipMemcpyAsync(accum, d_acc, ab, hipMemcpyDeviceToHost, h->stream));
    bool ordered = false;
    if (accum_mem == LWP_MEM_DEVICE) { rc = order_out(h, h->stream, &ordered); if (rc) return rc; }
    // without lwp_set_stream the results are complete on return (callers read accum on other streams)
    if (!ordered) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- pre-processing
extern "C" int lwp_preprocess_dims(int H, int W, int net_input_height, int stride, int* scaled_h, int* scaled_w,
                                   int* out_h, int* out_w, int* pad, double* scale) {
    if (H <= 0 || W <= 0 || net_input_height <= 0 || stride <= 0 || !scaled_h || !scaled_w || !out_h || !out_w || !pad || !scale)
        return fail(nullptr, LWP_ERR_ARG, "bad argument");
    const double sc = (double)net_input_height / (double)H;                       // demo.py:57
    const int dw = (int)nearbyint((double)W * sc), dh = (int)nearbyint((double)H * sc);   // cv2 dsize: round half to even
    if (dw <= 0 || dh <= 0) return fail(nullptr, LWP_ERR_ARG, "scaled frame is empty");
    // val.py:36-49 with min_dims = [net_input_height, max(dw, net_input_height)] (demo.py:61)
    const int h = dh < net_input_height ? dh : net_input_height;
    const int min0 = (int)ceil(net_input_height / (double)stride) * stride;
    const int m1 = dw > net_input_height ? dw : net_input_height;
    const int min1 = (int)ceil(m1 / (double)stride) * stride;
    pad[0] = (int)floor((min0 - h) / 2.0);
    pad[1] = (int)floor((min1 - dw) / 2.0);
    pad[2] = min0 - h - pad[0];
    pad[3] = min1 - dw - pad[1];
    *scaled_h = dh; *scaled_w = dw;
    *out_h = dh + pad[0] + pad[2];
    *out_w = dw + pad[1] + pad[3];
    *scale = sc;
    return LWP_OK;
}

// Host frames -> the device buffer `dst` on the handle's stream.  Up to kPinLimit bytes: memcpy (calling thread) into one of two
// pinned staging buffers, then an asynchronous DMA — the caller's buffer is free when this returns (*consumed = true) and the
// call need not wait for the copy; the staging buffer is reused only after the DMA that read it has finished (its event).
// Larger batches: a plain asynchronous copy from the caller's memory (*consumed = false: the caller must wait for ev_copy).
constexpr size_t kPinLimit = (size_t)64 << 20;
static int upload_host(lwp_context* h, const void* src, size_t bytes, void* dst, bool* consumed) {
    *consumed = false;
    if (bytes <= kPinLimit) {
        const int k = h->pin_next;
        h->pin_next ^= 1;
        if (h->pin_busy[k]) { HIP_TRY(h, hipEventSynchronize(h->pin_ev[k])); h->pin_busy[k] = false; }
        if (h->pin_bytes[k] < bytes) {
            if (h->pin_buf[k]) HIP_TRY(h, hipHostFree(h->pin_buf[k]));
            h->pin_buf[k] = nullptr; h->pin_bytes[k] = 0;
            HIP_TRY(h, hipHostMalloc(&h->pin_buf[k], bytes, hipHostMallocDefault));
            h->pin_bytes[k] = bytes;
        }
        if (!h->pin_ev[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->pin_ev[k], hipEventDisableTiming));
        std::memcpy(h->pin_buf[k], src, bytes);
        if (h->tune.host_fetch_dma == 1) {                   // LWP_HOST_FETCH_DMA=1: the copy engine instead of the fetch kernel (A/B)
            HIP_TRY(h, hipMemcpyAsync(dst, h->pin_buf[k], bytes, hipMemcpyHostToDevice, h->stream));
        } else {
            void* mapped = nullptr;
            HIP_TRY(h, hipHostGetDevicePointer(&mapped, h->pin_buf[k], 0));
            HIP_TRY(h, launch_fetch_host(mapped, dst, bytes, h->stream));
        }
        HIP_TRY(h, hipEventRecord(h->pin_ev[k], h->stream));
        h->pin_busy[k] = true;
        *consumed = true;
        return LWP_OK;
    }
    if (!h->ev_copy) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev_copy, h->stream));
    return LWP_OK;
}

extern "C" int lwp_preprocess_u8(lwp_handle h, const unsigned char* img, int img_mem, int H, int W, int net_input_height,
                                 int stride, const double* pad_value, const double* img_mean, double img_scale, float* out_device) {
    if (!h || !img || !pad_value || !img_mean || !out_device) return fail(h, LWP_ERR_ARG, "null argument");
    int dh, dw, Hp, Wp, pad[4];
    double sc;
    int rc = lwp_preprocess_dims(H, W, net_input_height, stride, &dh, &dw, &Hp, &Wp, pad, &sc);
    if (rc) return fail(h, rc, "bad frame / network size");
    if (pad[0] < 0 || pad[1] < 0 || pad[2] < 0 || pad[3] < 0) return fail(h, LWP_ERR_ARG, "negative padding");
    HIP_TRY(h, hipSetDevice(h->device));
    rc = order_in(h);                                    // device frame produced / output buffer last used on the caller's stream
    if (rc) return rc;
    const unsigned char* d_src = img;
    bool consumed = false;
    if (img_mem == LWP_MEM_HOST) {
        const size_t ib = (size_t)H * W * 3;
        rc = ensure_dev(h, &h->d_img, &h->d_img_bytes, ib);
        if (rc) return rc;
        rc = upload_host(h, img, ib, h->d_img, &consumed);
        if (rc) return rc;
        d_src = (const unsigned char*)h->d_img;
    }
    const size_t nx = (size_t)dw * 4, ny = (size_t)dh * 4;
    if (h->pre_H != H || h->pre_W != W || h->pre_net_h != net_input_height) {   // tables depend on the geometry only
        std::vector<int> xi, xw, yi, yw;
        build_resize_table_u8(W, dw, sc, xi, xw);
        build_resize_table_u8(H, dh, sc, yi, yw);
        HIP_TRY(h, hipStreamSynchronize(h->stream));                            // an earlier launch may still read the old tables
        rc = ensure_dev(h, &h->d_pre_tab, &h->d_pre_tab_bytes, (nx + ny) * 8);
        if (rc) return rc;
        int* t = (int*)h->d_pre_tab;
        HIP_TRY(h, hipMemcpy(t, xi.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t + nx, xw.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t + 2 * nx, yi.data(), ny * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t + 2 * nx + ny, yw.data(), ny * 4, hipMemcpyHostToDevice));
        h->pre_H = H; h->pre_W = W; h->pre_net_h = net_input_height;
    }
    const int* t = (const int*)h->d_pre_tab;
    PreprocParams p;
    p.src = d_src; p.Hs = H; p.Ws = W;
    p.xi = t; p.xw = t + nx; p.yi = t + 2 * nx; p.yw = t + 2 * nx + ny;
    p.dh = dh; p.dw = dw; p.top = pad[0]; p.left = pad[1]; p.Hp = Hp; p.Wp = Wp;
    for (int c = 0; c < 3; ++c) { p.mean[c] = img_mean[c]; p.pad_value[c] = (float)pad_value[c]; }
    p.scale = img_scale;
    p.out = out_device;
    LAUNCH(h, KC_POST, launch_preprocess_u8(p, h->stream));
    bool ordered = false;
    rc = order_out(h, h->stream, &ordered);
    if (rc) return rc;
    // the caller may reuse its host frame buffer on return: it has been copied into the pinned staging buffer already (or, for
    // very large frames, the COPY is waited for); the kernel's output is stream-ordered (consumed by this handle's next call, or
    // by the caller's stream after the event hand-over).  Without a declared caller stream the call completes on return.
    if (img_mem == LWP_MEM_HOST) {
        if (!ordered) HIP_TRY(h, hipStreamSynchronize(h->stream));
        else if (!consumed) HIP_TRY(h, hipEventSynchronize(h->ev_copy));
    }
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- multi-scale image side
extern "C" int lwp_scale_dims(int H, int W, double ratio, int base_height, int stride, int* scaled_h, int* scaled_w,
                              int* out_h, int* out_w, int* pad) {
    if (H <= 0 || W <= 0 || !(ratio > 0.0) || base_height <= 0 || stride <= 0 || !scaled_h || !scaled_w || !out_h || !out_w || !pad)
        return fail(nullptr, LWP_ERR_ARG, "bad argument");
    const double fw = nearbyint((double)W * ratio), fh = nearbyint((double)H * ratio);   // cv2 dsize: round half to even
    if (!(fw >= 1.0 && fh >= 1.0 && fw <= 65535.0 && fh <= 65535.0)) return fail(nullptr, LWP_ERR_ARG, "scaled frame is empty or too large");
    const int dw = (int)fw, dh = (int)fh;
    // val.py:36-49 with min_dims = [base_height, max(dw, base_height)] (val.py:90)
    const int hh = dh < base_height ? dh : base_height;
    const int min0 = (int)ceil(base_height / (double)stride) * stride;
    const int m1 = dw > base_height ? dw : base_height;
    const int min1 = (int)ceil(m1 / (double)stride) * stride;
    pad[0] = (int)floor((min0 - hh) / 2.0);
    pad[1] = (int)floor((min1 - dw) / 2.0);
    pad[2] = min0 - hh - pad[0];
    pad[3] = min1 - dw - pad[1];
    *scaled_h = dh; *scaled_w = dw;
    *out_h = dh + pad[0] + pad[2];
    *out_w = dw + pad[1] + pad[3];
    return LWP_OK;
}

static int preprocess_scaled_impl(lwp_handle h, const void* imgs, int elem, int img_mem, int N, int H, int W, double ratio,
                                  int base_height, int stride, const double* pad_value, const double* img_mean,
                                  double img_scale, float* out_device);
extern "C" int lwp_preprocess_scaled_u8(lwp_handle h, const unsigned char* imgs, int img_mem, int N, int H, int W, double ratio,
                                        int base_height, int stride, const double* pad_value, const double* img_mean,
                                        double img_scale, float* out_device) {
    return preprocess_scaled_impl(h, imgs, 1, img_mem, N, H, W, ratio, base_height, stride, pad_value, img_mean, img_scale, out_device);
}
extern "C" int lwp_preprocess_scaled_f32(lwp_handle h, const float* imgs, int img_mem, int N, int H, int W, double ratio,
                                         int base_height, int stride, const double* pad_value, const double* img_mean,
                                         double img_scale, float* out_device) {
    return preprocess_scaled_impl(h, imgs, 4, img_mem, N, H, W, ratio, base_height, stride, pad_value, img_mean, img_scale, out_device);
}
static int preprocess_scaled_impl(lwp_handle h, const void* imgs, int elem, int img_mem, int N, int H, int W, double ratio,
                                  int base_height, int stride, const double* pad_value, const double* img_mean,
                                  double img_scale, float* out_device) {
    if (!h || !imgs || !pad_value || !img_mean || !out_device || N <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    int dh, dw, Hp, Wp, pad[4];
    int rc = lwp_scale_dims(H, W, ratio, base_height, stride, &dh, &dw, &Hp, &Wp, pad);
    if (rc) return fail(h, rc, "bad frame size / scale ratio");
    if (pad[0] < 0 || pad[1] < 0 || pad[2] < 0 || pad[3] < 0) return fail(h, LWP_ERR_ARG, "negative padding");
    HIP_TRY(h, hipSetDevice(h->device));
    rc = order_in(h);
    if (rc) return rc;
    const void* d_src = imgs;
    bool consumed = false;
    if (img_mem == LWP_MEM_HOST) {
        const size_t ib = (size_t)N * H * W * 3 * elem;
        if (h->d_imgs_bytes < ib) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (h->d_imgs) HIP_TRY(h, hipFree(h->d_imgs));
            h->d_imgs = nullptr; h->d_imgs_bytes = 0;
            HIP_TRY(h, hipMalloc((void**)&h->d_imgs, ib));
            h->d_imgs_bytes = ib;
        }
        rc = upload_host(h, imgs, ib, h->d_imgs, &consumed);
        if (rc) return rc;
        d_src = h->d_imgs;
    }
    const size_t nx = (size_t)dw * 4, ny = (size_t)dh * 4;
    void* d_tabs = nullptr;
    for (const auto& rt : h->scale_tabs)
        if (rt.cw == W && rt.ch == H && rt.dw == dw && rt.dh == dh && rt.ratio == ratio) { d_tabs = rt.d; break; }
    if (!d_tabs) {                                       // per-geometry tables, uploaded once and kept
        std::vector<int> xi, yi;
        std::vector<float> xw, yw;
        build_resize_table_ratio(W, dw, ratio, xi, xw);
        build_resize_table_ratio(H, dh, ratio, yi, yw);
        if (h->scale_tabs.size() >= 16) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            (void)hipFree(h->scale_tabs.front().d);
            h->scale_tabs.erase(h->scale_tabs.begin());
        }
        HIP_TRY(h, hipMalloc(&d_tabs, (nx + ny) * 8));
        char* t0 = (char*)d_tabs;
        HIP_TRY(h, hipMemcpy(t0, xi.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 4, xw.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 8, yi.data(), ny * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(t0 + nx * 8 + ny * 4, yw.data(), ny * 4, hipMemcpyHostToDevice));
        h->scale_tabs.push_back({W, H, dw, dh, d_tabs, ratio});
    }
    char* t = (char*)d_tabs;
    PreScaleParams p;
    p.src = d_src; p.src_f32 = elem == 4; p.N = N; p.Hs = H; p.Ws = W;
    p.xi = (const int*)t; p.xw = (const float*)(t + nx * 4); p.yi = (const int*)(t + nx * 8); p.yw = (const float*)(t + nx * 8 + ny * 4);
    p.dh = dh; p.dw = dw; p.top = pad[0]; p.left = pad[1]; p.Hp = Hp; p.Wp = Wp;
    for (int c = 0; c < 3; ++c) { p.mean[c] = img_mean[c]; p.pad_value[c] = (float)pad_value[c]; }
    p.scale = img_scale;
    p.out = out_device;
    LAUNCH(h, KC_POST, launch_preprocess_scaled(p, h->stream));
    bool ordered = false;
    rc = order_out(h, h->stream, &ordered);
    if (rc) return rc;
    if (!ordered) HIP_TRY(h, hipStreamSynchronize(h->stream));                    // no declared caller stream: complete on return
    else if (img_mem == LWP_MEM_HOST && !consumed) HIP_TRY(h, hipEventSynchronize(h->ev_copy));   // host frames may be reused: the copy only
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- extract_keypoints
extern "C" int lwp_extract_keypoints(lwp_handle h, float* heatmap, int H, int W, int64_t row_stride, int64_t pix_stride,
                                     int64_t* xs, int64_t* ys, float* scores, int cap, int* count) {
    if (!h || !heatmap || !xs || !ys || !scores || !count || H <= 0 || W <= 0 || cap < 0) return fail(h, LWP_ERR_ARG, "bad argument");
    if (H > 65535 || W > 65535) return fail(h, LWP_ERR_ARG, "map too large");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 1);
    if (rc) return rc;
    const size_t bytes = (size_t)H * W * sizeof(float);
    rc = ensure_host_stage(h, bytes + (size_t)h->caps.max_kpts * 12 + 64);
    if (rc) return rc;
    rc = ensure_dev(h, &h->d_tmp, &h->d_tmp_bytes, bytes);
    if (rc) return rc;
    float* hs = (float*)h->h_stage;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) hs[(size_t)y * W + x] = heatmap[y * row_stride + x * pix_stride];
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp, hs, bytes, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, KC_POST, launch_reset_ws(1, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_threshold_inplace(h->d_tmp, (int64_t)H * W, h->stream));
    MapView v{h->d_tmp, 0, (int64_t)W, 1, 0, H, W};
    LAUNCH(h, KC_POST, launch_find_peaks(v, 1, 1, 1, h->ws, h->stream, &h->tune));
    LAUNCH(h, KC_POST, launch_nms(1, 1, H, h->ws, h->stream));
    HIP_TRY(h, hipMemcpyAsync(hs, h->d_tmp, bytes, hipMemcpyDeviceToHost, h->stream));
    int* h_xy = (int*)((char*)h->h_stage + bytes);
    float* h_sc = (float*)(h_xy + (size_t)h->caps.max_kpts * 2);
    int n = 0;
    unsigned long long fl = 0;
    HIP_TRY(h, hipMemcpyAsync(h_xy, h->ws.kpt_xy, (size_t)h->caps.max_kpts * 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_sc, h->ws.kpt_score, (size_t)h->caps.max_kpts * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&n, h->ws.kpt_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&fl, h->ws.flags, sizeof(fl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, launch_reset_ws(1, h->ws, h->stream));   // leave the append counters zeroed (the fused path relies on it)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) heatmap[y * row_stride + x * pix_stride] = hs[(size_t)y * W + x];
    if (fl & 3ull) return fail(h, LWP_ERR_CAPACITY, "extract_keypoints: peak or key-point capacity exceeded (lwp_set_capacity)");
    if (n > cap) return fail(h, LWP_ERR_CAPACITY, "extract_keypoints: output arrays too small");
    for (int i = 0; i < n; ++i) { xs[i] = h_xy[2 * i]; ys[i] = h_xy[2 * i + 1]; scores[i] = h_sc[i]; }
    *count = n;
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- results
static int parse_results(lwp_context* h, const PostWorkspace& ws, const void* host_block, int N, int* kpt_counts, double* kpts,
                         int kpt_cap, double* entries, int entry_cap, int* n_entries);

static int fetch_results(lwp_context* h, int N, int* kpt_counts, double* kpts, int kpt_cap, double* entries, int entry_cap, int* n_entries) {
    int rc = ensure_host_stage(h, h->ws.result_bytes + 64);
    if (rc) return rc;
    HIP_TRY(h, launch_publish(N, h->ws, h->h_stage, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return parse_results(h, h->ws, h->h_stage, N, kpt_counts, kpts, kpt_cap, entries, entry_cap, n_entries);
}

static int parse_results(lwp_context* h, const PostWorkspace& ws, const void* host_block, int N, int* kpt_counts, double* kpts,
                         int kpt_cap, double* entries, int entry_cap, int* n_entries) {
    const PostCaps& c = ws.caps;
    const int WN = ws.N;                         // the block is laid out for the workspace's frame capacity
    const char* p = (const char*)host_block;
    const unsigned long long* h_fl = (const unsigned long long*)p; p += (size_t)WN * 4 * 8;
    const double* h_k = (const double*)p; p += (size_t)WN * 18 * c.max_kpts * 4 * 8;
    const double* h_e = (const double*)p; p += (size_t)WN * c.max_entries * 20 * 8;
    const int* h_cnt = (const int*)p; p += (size_t)WN * 18 * 4;
    const int* h_ne = (const int*)p;
    for (int f = 0; f < N; ++f) {
        if (h_fl[f * 4 + 0]) {
            char msg[160];
            snprintf(msg, sizeof msg, "frame %d: post-processing capacity exceeded (bits 0x%llx: 1 peaks, 2 key-points, 4 connections/entries)", f, h_fl[f * 4]);
            return fail(h, LWP_ERR_CAPACITY, msg);
        }
        if (h_fl[f * 4 + 1] < h_fl[f * 4 + 2])
            return fail(h, LWP_ERR_UNBOUND, "local variable 'ratio' referenced before assignment");
        int total = 0;
        for (int t = 0; t < 18; ++t) { kpt_counts[f * 18 + t] = h_cnt[f * 18 + t]; total += h_cnt[f * 18 + t]; }
        if (total > kpt_cap || h_ne[f] > entry_cap) return fail(h, LWP_ERR_CAPACITY, "result arrays too small");
        std::memcpy(kpts + (size_t)f * kpt_cap * 4, h_k + (size_t)f * 18 * c.max_kpts * 4, (size_t)total * 4 * sizeof(double));
        std::memcpy(entries + (size_t)f * entry_cap * 20, h_e + (size_t)f * c.max_entries * 20, (size_t)h_ne[f] * 20 * sizeof(double));
        n_entries[f] = h_ne[f];
    }
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- group_keypoints
extern "C" int lwp_group_keypoints(lwp_handle h, const double* kpts, const int* type_counts, const float* pafs, int pafs_mem,
                                   int H, int W, int demo, double* pose_entries, int cap_entries, int* n_entries) {
    if (!h || !type_counts || !pafs || !pose_entries || !n_entries || H <= 0 || W <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 1);
    if (rc) return rc;
    const PostCaps& c = h->ws.caps;
    int total = 0;
    for (int t = 0; t < 18; ++t) {
        if (type_counts[t] < 0) return fail(h, LWP_ERR_ARG, "negative type count");
        if (type_counts[t] > c.max_kpts) return fail(h, LWP_ERR_CAPACITY, "group_keypoints: more key-points of one type than max_kpts_per_type");
        total += type_counts[t];
    }
    if (total > 0 && !kpts) return fail(h, LWP_ERR_ARG, "kpts is null");
    std::vector<int> xy((size_t)18 * c.max_kpts * 2, 0), cnt(18);
    std::vector<float> sc((size_t)18 * c.max_kpts, 0.f);
    int r = 0;
    for (int t = 0; t < 18; ++t) {
        cnt[t] = type_counts[t];
        for (int i = 0; i < type_counts[t]; ++i, ++r) {
            const double x = kpts[r * 4], y = kpts[r * 4 + 1];
            if (!(x >= 0 && x < W && y >= 0 && y < H)) return fail(h, LWP_ERR_ARG, "key-point outside the PAF map");
            if (kpts[r * 4 + 3] != (double)r) return fail(h, LWP_ERR_ARG, "key-point ids must be the running index 0..K-1");
            xy[((size_t)t * c.max_kpts + i) * 2] = (int)x;
            xy[((size_t)t * c.max_kpts + i) * 2 + 1] = (int)y;
            sc[(size_t)t * c.max_kpts + i] = (float)kpts[r * 4 + 2];
        }
    }
    const int NPc = h->g.NP;
    const float* d_paf = pafs;
    if (pafs_mem == LWP_MEM_HOST) {
        const size_t pb = (size_t)H * W * NPc * sizeof(float);
        rc = ensure_dev(h, &h->d_tmp2, &h->d_tmp2_bytes, pb);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp2, pafs, pb, hipMemcpyHostToDevice, h->stream));
        d_paf = h->d_tmp2;
    }
    LAUNCH(h, KC_POST, launch_reset_ws(1, h->ws, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->ws.kpt_xy, xy.data(), xy.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->ws.kpt_score, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->ws.kpt_count, cnt.data(), 18 * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // the host vectors above go out of scope
    MapView v{d_paf, 0, (int64_t)W * NPc, (int64_t)NPc, 1, H, W};
    LAUNCH(h, KC_POST, launch_score_pairs(v, 1, 1, demo, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_match(1, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_assemble(1, h->ws, h->stream));
    std::vector<int> kc(18);
    std::vector<double> kout((size_t)std::max(total, 1) * 4);
    return fetch_results(h, 1, kc.data(), kout.data(), std::max(total, 1), pose_entries, cap_entries, n_entries);
}

// ---------------------------------------------------------------------------------------------- fused pipeline
static int enqueue_poses_chunk(lwp_context* h, const float* d_in, int N, int H, int W, int ratio, int demo, bool with_post, PostWorkspace ws) {
    const Graph& g = h->g;
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);
    const int cc = g.cat_channels;
    MapView heat, paf;
    // bf16: the concat buffer is bf16, so the last stage's heads ALSO write f32 NCHW maps for the post-processing.  f32: the maps
    // could be read in place from the (NHWC) concat buffer, but the 4 x 4 footprint of a cubic sample is 4 cache lines in a
    // channel plane and 16 in channels-last rows: pair scoring at batch 32 runs 255 us on the concat buffer and ~80 on planes
    // (LWP_POST_NCHW=0: the in-place form, A/B)
    if (h->dtype == LWP_BF16 || (with_post && h->tune.post_nchw != 0)) {
        const int nout = 2 * (1 + g.nref);
        std::vector<float*> outs(nout, nullptr);
        const size_t hb = (size_t)N * g.NH * fh * fw * sizeof(float), pb = (size_t)N * g.NP * fh * fw * sizeof(float);
        int rc = ensure_dev(h, &h->d_maps[0], &h->d_maps_bytes[0], hb);
        if (rc) return rc;
        rc = ensure_dev(h, &h->d_maps[1], &h->d_maps_bytes[1], pb);
        if (rc) return rc;
        outs[nout - 2] = h->d_maps[0];
        outs[nout - 1] = h->d_maps[1];
        rc = enqueue_forward(h, d_in, N, H, W, outs.data());
        if (rc || !with_post) return rc;
        const int64_t hw = (int64_t)fh * fw;
        heat = MapView{h->d_maps[0], (int64_t)g.NH * hw, (int64_t)fw, 1, hw, fh, fw};
        paf = MapView{h->d_maps[1], (int64_t)g.NP * hw, (int64_t)fw, 1, hw, fh, fw};
    } else {
        int rc = enqueue_forward(h, d_in, N, H, W, nullptr);
        if (rc || !with_post) return rc;
        const float* cat = h->bufs[g.cat_buf];
        heat = MapView{cat + g.C, (int64_t)fh * fw * cc, (int64_t)fw * cc, (int64_t)cc, 1, fh, fw};
        paf = MapView{cat + g.C + g.NH, (int64_t)fh * fw * cc, (int64_t)fw * cc, (int64_t)cc, 1, fh, fw};
    }
    LAUNCH(h, KC_POST, launch_find_peaks(heat, N, 18, ratio, ws, h->stream, &h->tune));
    LAUNCH(h, KC_POST, launch_nms(N, 18, fh * ratio, ws, h->stream));
    LAUNCH(h, KC_POST, launch_score_pairs(paf, N, ratio, demo, ws, h->stream));
    LAUNCH(h, KC_POST, launch_match(N, ws, h->stream));
    LAUNCH(h, KC_POST, launch_assemble(N, ws, h->stream));
    return LWP_OK;
}

// the whole batch, in as many equal launch sequences as keep every tensor below the 2 GiB addressing limit (one for any
// batch the BASELINE configs use); the results of all frames land in h->ws
static int enqueue_poses(lwp_context* h, const float* d_in, int N, int H, int W, int ratio, int demo, bool with_post) {
    const int Nc = frames_per_pass(h, N, H, W);
    for (int f0 = 0; f0 < N; f0 += Nc) {
        const int n = std::min(Nc, N - f0);
        if (n != h->cur_N) { int rc = ensure_activations(h, n, H, W); if (rc) return rc; }
        int rc = enqueue_poses_chunk(h, d_in + (size_t)f0 * 3 * H * W, n, H, W, ratio, demo, with_post, ws_frames(h->ws, f0));
        if (rc) return rc;
    }
    return LWP_OK;
}

static int prepare_poses(lwp_context* h, int N, int H, int W, int ratio) {
    int rc = check_frame_shape(h, N, H, W);
    if (rc) return rc;
    if (ratio != 4 && ratio != 8) return fail(h, LWP_ERR_ARG, "upsample ratio must be 4 or 8");
    if (h->g.NH < 18 || h->g.NP < 38) return fail(h, LWP_ERR_ARG, "pose grouping needs >= 18 heat-maps and >= 38 PAFs");
    if (((int64_t)H / 8 + 1) * ratio > 65535 || ((int64_t)W / 8 + 1) * ratio > 65535) return fail(h, LWP_ERR_ARG, "map too large");
    HIP_TRY(h, hipSetDevice(h->device));
    rc = ensure_activations(h, frames_per_pass(h, N, H, W), H, W);
    if (rc) return rc;
    return ensure_ws(h, N);
}

extern "C" int lwp_infer_poses_async(lwp_handle h, const float* in_device, int N, int H, int W, int ratio, int demo) {
    if (!h || !in_device) return fail(h, LWP_ERR_ARG, "null argument");
    int rc = prepare_poses(h, N, H, W, ratio);
    if (rc) return rc;
    rc = order_in(h);
    if (rc) return rc;
    h->last_N = N;
    return enqueue_poses(h, in_device, N, H, W, ratio, demo, true);
}

extern "C" int lwp_fetch_poses(lwp_handle h, int* kpt_counts, double* kpts, int kpt_cap, double* entries, int entry_cap, int* n_entries) {
    if (!h || !kpt_counts || !kpts || !entries || !n_entries) return fail(h, LWP_ERR_ARG, "null argument");
    if (h->last_N <= 0) return fail(h, LWP_ERR_STATE, "no pipeline run to fetch");
    HIP_TRY(h, hipSetDevice(h->device));
    return fetch_results(h, h->last_N, kpt_counts, kpts, kpt_cap, entries, entry_cap, n_entries);
}

extern "C" int lwp_infer_poses(lwp_handle h, const float* in, int in_mem, int N, int H, int W, int ratio, int demo,
                               int* kpt_counts, double* kpts, int kpt_cap, double* entries, int entry_cap, int* n_entries) {
    if (!h || !in || !kpt_counts || !kpts || !entries || !n_entries) return fail(h, LWP_ERR_ARG, "null argument");
    int rc = prepare_poses(h, N, H, W, ratio);
    if (rc) return rc;
    if (in_mem == LWP_MEM_DEVICE) { rc = order_in(h); if (rc) return rc; }
    const float* d_in = nullptr;
    rc = stage_input(h, in, in_mem, (size_t)N * 3 * H * W * sizeof(float), &d_in);
    if (rc) return rc;
    h->last_N = N;
    rc = enqueue_poses(h, d_in, N, H, W, ratio, demo, true);
    if (rc) return rc;
    return fetch_results(h, N, kpt_counts, kpts, kpt_cap, entries, entry_cap, n_entries);
}

// ---------------------------------------------------------------------------------------------- pipelined streaming
extern "C" int lwp_pipeline_submit(lwp_handle h, const float* in_device, int N, int H, int W, int ratio, int demo, int slot) {
    if (!h || !in_device || slot < 0 || slot > 1) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = check_frame_shape(h, N, H, W);
    if (rc) return rc;
    if (ratio != 4 && ratio != 8) return fail(h, LWP_ERR_ARG, "upsample ratio must be 4 or 8");
    if (h->g.NH < 18 || h->g.NP < 38) return fail(h, LWP_ERR_ARG, "pose grouping needs >= 18 heat-maps and >= 38 PAFs");
    HIP_TRY(h, hipSetDevice(h->device));
    lwp_context::Slot& sl = h->slots[slot];
    if (sl.pending) return fail(h, LWP_ERR_STATE, "slot still pending: call lwp_pipeline_fetch first");
    if (!h->post_stream) {
        if (h->post_on_main) h->post_stream = h->stream;
        else HIP_TRY(h, hipStreamCreateWithFlags(&h->post_stream, hipStreamNonBlocking));
    }
    if (!sl.ev_maps) {
        HIP_TRY(h, hipEventCreateWithFlags(&sl.ev_maps, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    }
    const int Nc = frames_per_pass(h, N, H, W);
    rc = ensure_activations(h, Nc, H, W);
    if (rc) return rc;
    rc = order_in(h);
    if (rc) return rc;
    if (sl.ws.caps.max_peaks != h->caps.max_peaks || sl.ws.caps.max_kpts != h->caps.max_kpts) sl.ws.N = 0;   // (re)allocate lazily
    rc = ensure_ws_obj(h, sl.ws, N, h->post_stream);
    if (rc) return rc;
    const Graph& g = h->g;
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);               // three stride-2 stages: out = (in - 1) / 2 + 1 each
    const size_t hb = (size_t)N * g.NH * fh * fw * sizeof(float), pb = (size_t)N * g.NP * fh * fw * sizeof(float);
    rc = ensure_dev(h, &sl.maps[0], &sl.maps_bytes[0], hb);
    if (rc) return rc;
    rc = ensure_dev(h, &sl.maps[1], &sl.maps_bytes[1], pb);
    if (rc) return rc;
    if (sl.h_stage_bytes < sl.ws.result_bytes) {
        if (sl.h_stage) HIP_TRY(h, hipHostFree(sl.h_stage));
        sl.h_stage = nullptr; sl.h_stage_bytes = 0;
        HIP_TRY(h, hipHostMalloc(&sl.h_stage, sl.ws.result_bytes, hipHostMallocDefault));
        sl.h_stage_bytes = sl.ws.result_bytes;
    }
    // network on the main stream: the last stage's heads also write f32 NCHW maps into this slot
    const int nout = 2 * (1 + g.nref);
    std::vector<float*> outs(nout, nullptr);
    for (int f0 = 0; f0 < N; f0 += Nc) {             // one launch sequence unless a tensor would reach 2 GiB
        const int n = std::min(Nc, N - f0);
        if (n != h->cur_N) { rc = ensure_activations(h, n, H, W); if (rc) return rc; }
        outs[nout - 2] = sl.maps[0] + (size_t)f0 * g.NH * fh * fw;
        outs[nout - 1] = sl.maps[1] + (size_t)f0 * g.NP * fh * fw;
        rc = enqueue_forward(h, in_device + (size_t)f0 * 3 * H * W, n, H, W, outs.data());
        if (rc) return rc;
    }
    HIP_TRY(h, hipEventRecord(sl.ev_maps, h->stream));
    // post-processing + result copy on the second stream
    if (h->post_stream != h->stream) HIP_TRY(h, hipStreamWaitEvent(h->post_stream, sl.ev_maps, 0));
    const int64_t hw = (int64_t)fh * fw;
    MapView heat{sl.maps[0], (int64_t)g.NH * hw, (int64_t)fw, 1, hw, fh, fw};
    MapView paf{sl.maps[1], (int64_t)g.NP * hw, (int64_t)fw, 1, hw, fh, fw};
    HIP_TRY(h, launch_find_peaks(heat, N, 18, ratio, sl.ws, h->post_stream, &h->tune));
    HIP_TRY(h, launch_nms(N, 18, fh * ratio, sl.ws, h->post_stream));
    HIP_TRY(h, launch_score_pairs(paf, N, ratio, demo, sl.ws, h->post_stream));
    HIP_TRY(h, launch_match(N, sl.ws, h->post_stream));
    HIP_TRY(h, launch_assemble(N, sl.ws, h->post_stream));
    HIP_TRY(h, launch_publish(N, sl.ws, sl.h_stage, h->post_stream));
    HIP_TRY(h, hipEventRecord(sl.ev_done, h->post_stream));
    sl.pending = true;
    sl.N = N;
    return LWP_OK;
}

extern "C" int lwp_pipeline_fetch(lwp_handle h, int slot, int* kpt_counts, double* kpts, int kpt_cap, double* entries, int entry_cap,
                                  int* n_entries) {
    if (!h || slot < 0 || slot > 1 || !kpt_counts || !kpts || !entries || !n_entries) return fail(h, LWP_ERR_ARG, "bad argument");
    lwp_context::Slot& sl = h->slots[slot];
    if (!sl.pending) return fail(h, LWP_ERR_STATE, "nothing submitted on this slot");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(sl.ev_done));
    sl.pending = false;
    return parse_results(h, sl.ws, sl.h_stage, sl.N, kpt_counts, kpts, kpt_cap, entries, entry_cap, n_entries);
}

extern "C" int lwp_poses_from_maps(lwp_handle h, const float* heat, const float* paf, int mem, int layout, int N, int hs, int ws, int ratio,
                                   int demo, int* kpt_counts, double* kpts, int kpt_cap, double* entries, int entry_cap,
                                   int* n_entries) {
    if (!h || !heat || !paf || !kpt_counts || !kpts || !entries || !n_entries || N <= 0 || hs <= 0 || ws <= 0)
        return fail(h, LWP_ERR_ARG, "bad argument");
    if (ratio != 4 && ratio != 8 && ratio != 1) return fail(h, LWP_ERR_ARG, "upsample ratio must be 1, 4 or 8");
    if (layout != LWP_LAYOUT_NCHW && layout != LWP_LAYOUT_NHWC) return fail(h, LWP_ERR_ARG, "layout must be LWP_LAYOUT_NCHW or LWP_LAYOUT_NHWC");
    if (h->g.NH < 18 || h->g.NP < 38) return fail(h, LWP_ERR_ARG, "pose grouping needs >= 18 heat-maps and >= 38 PAFs");
    if ((int64_t)hs * ratio > 65535 || (int64_t)ws * ratio > 65535) return fail(h, LWP_ERR_ARG, "map too large");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, N);
    if (rc) return rc;
    if (mem == LWP_MEM_DEVICE) { rc = order_in(h); if (rc) return rc; }
    const size_t hb = (size_t)N * h->g.NH * hs * ws * sizeof(float), pb = (size_t)N * h->g.NP * hs * ws * sizeof(float);
    const float *d_heat = heat, *d_paf = paf;
    if (mem == LWP_MEM_HOST) {
        rc = ensure_dev(h, &h->d_tmp, &h->d_tmp_bytes, hb);
        if (rc) return rc;
        rc = ensure_dev(h, &h->d_tmp2, &h->d_tmp2_bytes, pb);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp, heat, hb, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp2, paf, pb, hipMemcpyHostToDevice, h->stream));
        d_heat = h->d_tmp; d_paf = h->d_tmp2;
    }
    const int64_t hw = (int64_t)hs * ws;
    MapView hv{d_heat, (int64_t)h->g.NH * hw, (int64_t)ws, 1, hw, hs, ws};
    MapView pv{d_paf, (int64_t)h->g.NP * hw, (int64_t)ws, 1, hw, hs, ws};
    if (layout == LWP_LAYOUT_NHWC) {                 // the averaged maps of val.infer are H x W x C
        hv.ys = (int64_t)ws * h->g.NH; hv.xs = h->g.NH; hv.cs = 1;
        pv.ys = (int64_t)ws * h->g.NP; pv.xs = h->g.NP; pv.cs = 1;
    }
    LAUNCH(h, KC_POST, launch_find_peaks(hv, N, 18, ratio, h->ws, h->stream, &h->tune));
    LAUNCH(h, KC_POST, launch_nms(N, 18, hs * ratio, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_score_pairs(pv, N, ratio, demo, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_match(N, h->ws, h->stream));
    LAUNCH(h, KC_POST, launch_assemble(N, h->ws, h->stream));
    h->last_N = N;
    return fetch_results(h, N, kpt_counts, kpts, kpt_cap, entries, entry_cap, n_entries);
}

// ---------------------------------------------------------------------------------------------- introspection
extern "C" int lwp_layer_count(lwp_handle h) { return h ? (int)h->g.layers.size() : LWP_ERR_ARG; }

extern "C" int lwp_layer_info(lwp_handle h, int idx, char* name, int name_cap, int* kind, int* cin, int* cout, int* ks,
                              int* stride, int* dil, int64_t* macs_per_pixel) {
    if (!h || idx < 0 || idx >= (int)h->g.layers.size() || !name) return fail(h, LWP_ERR_ARG, "bad argument");
    const Layer& l = h->g.layers[idx];
    if ((int)l.name.size() + 1 > name_cap) return fail(h, LWP_ERR_ARG, "name buffer too small");
    std::strcpy(name, l.name.c_str());
    if (kind) *kind = l.kind;
    if (cin) *cin = l.cin;
    if (cout) *cout = l.cout;
    if (ks) *ks = l.ks;
    if (stride) *stride = l.stride;
    if (dil) *dil = l.dil;
    if (macs_per_pixel) *macs_per_pixel = l.macs_per_pixel;
    return LWP_OK;
}

extern "C" int lwp_debug_layer_output(lwp_handle h, const float* in, int N, int H, int W, int idx, float* dst, size_t dst_floats,
                                      int out_dims[4]) {
    if (!h || !in || !dst || !out_dims || idx < 0 || idx >= (int)h->g.layers.size()) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = check_frame_shape(h, N, H, W);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = ensure_activations(h, N, H, W);
    if (rc) return rc;
    const float* d_in = nullptr;
    rc = stage_input(h, in, LWP_MEM_HOST, (size_t)N * 3 * H * W * sizeof(float), &d_in);
    if (rc) return rc;
    h->record_variants = true;
    rc = enqueue_forward(h, d_in, N, H, W, nullptr, idx + 1);
    h->record_variants = false;
    if (rc) return rc;
    const Layer& l = h->g.layers[idx];
    int dh, dw;
    level_dims(H, W, h->g.bufs[l.dst.buf].level, &dh, &dw);
    const size_t n = (size_t)N * l.cout * dh * dw;
    if (dst_floats < n) return fail(h, LWP_ERR_ARG, "dst too small");
    rc = ensure_dev(h, &h->d_tmp, &h->d_tmp_bytes, n * sizeof(float));
    if (rc) return rc;
    // NHWC window (ld, coff) -> compact NCHW
    if (h->dtype == LWP_BF16) HIP_TRY(h, launch_nchw_from_nhwc_bf16(buf_at(h, l.dst), l.dst.ld, h->d_tmp, N, dh * dw, l.cout, h->stream));
    else HIP_TRY(h, launch_nchw_from_nhwc(buf_at(h, l.dst), l.dst.ld, h->d_tmp, N, dh * dw, l.cout, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dst, h->d_tmp, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    out_dims[0] = N; out_dims[1] = l.cout; out_dims[2] = dh; out_dims[3] = dw;
    return LWP_OK;
}

extern "C" int lwp_debug_frames_per_pass(lwp_handle h, int N, int H, int W) {
    if (!h || N <= 0 || H <= 0 || W <= 0) return LWP_ERR_ARG;
    return frames_per_pass(h, N, H, W);
}

extern "C" int lwp_debug_post_counts(lwp_handle h, int frame, int* peaks18, int* kpts18, int* candidates19, int* picked19) {
    if (!h || !peaks18 || !kpts18 || !candidates19 || !picked19) return fail(h, LWP_ERR_ARG, "null argument");
    if (frame < 0 || frame >= h->ws.N) return fail(h, LWP_ERR_ARG, "frame outside the last batch");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->post_stream != h->stream) HIP_TRY(h, hipStreamSynchronize(h->post_stream));
    HIP_TRY(h, hipMemcpy(peaks18, h->ws.seen + frame * 37, 18 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(kpts18, h->ws.kpt_count + frame * 18, 18 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(candidates19, h->ws.seen + frame * 37 + 18, 19 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(picked19, h->ws.sel_count + frame * 19, 19 * sizeof(int), hipMemcpyDeviceToHost));
    return LWP_OK;
}

extern "C" int lwp_debug_layer_variant(lwp_handle h, int idx, char* name, int name_cap) {
    if (!h || idx < 0 || idx >= (int)h->g.layers.size() || !name || name_cap <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    const std::string& v = h->variants[idx];
    if ((int)v.size() + 1 > name_cap) return fail(h, LWP_ERR_ARG, "name buffer too small");
    std::strcpy(name, v.c_str());
    return LWP_OK;
}

extern "C" int lwp_debug_time_layer(lwp_handle h, int idx, int N, int H, int W, int iters, float* ms_avg) {
    if (!h || !ms_avg || idx < 0 || idx >= (int)h->g.layers.size() || iters <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = check_frame_shape(h, N, H, W);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = ensure_activations(h, N, H, W);
    if (rc) return rc;
    rc = ensure_dev(h, &h->d_in, &h->d_in_bytes, (size_t)N * 3 * H * W * sizeof(float));
    if (rc) return rc;
    const Layer& l = h->g.layers[idx];
    // a fused head pair is timed at its first layer; its second layer has no launch of its own
    int fh, fw;
    level_dims(H, W, 3, &fh, &fw);
    const int64_t M3 = (int64_t)N * fh * fw;
    const bool pair = heads_pair_fusable(h, (size_t)idx, M3);
    if (idx > 0 && heads_pair_fusable(h, (size_t)idx - 1, M3)) { *ms_avg = 0.f; return LWP_OK; }
    auto one = [&]() { return pair ? enqueue_heads_pair(h, l, h->g.layers[idx + 1], N, H, W, nullptr) : enqueue_layer(h, l, h->d_in, N, H, W, nullptr); };
    hipEvent_t e0, e1;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { rc = one(); if (rc) return rc; }
    HIP_TRY(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < iters; ++i) { rc = one(); if (rc) return rc; }
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    HIP_TRY(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return LWP_OK;
}

extern "C" int lwp_synchronize(lwp_handle h) {
    if (!h) return LWP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->post_stream) HIP_TRY(h, hipStreamSynchronize(h->post_stream));
    return LWP_OK;
}

// ---------------------------------------------------------------------------------------------- measurement
extern "C" int lwp_time_pipeline(lwp_handle h, const float* in_device, int N, int H, int W, int ratio, int demo, int what,
                                 int iters, float* ms_total) {
    if (!h || !in_device || !ms_total || iters <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = prepare_poses(h, N, H, W, ratio);
    if (rc) return rc;
    h->last_N = N;
    hipEvent_t e0, e1;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    HIP_TRY(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < iters; ++i) {
        rc = enqueue_poses(h, in_device, N, H, W, ratio, demo, what != 0);
        if (rc) return rc;
    }
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    HIP_TRY(h, hipEventSynchronize(e1));
    HIP_TRY(h, hipEventElapsedTime(ms_total, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return LWP_OK;
}

extern "C" int lwp_profile_launches(lwp_handle h, const float* in_device, int N, int H, int W, int ratio, int demo, int reps,
                                    float* ms, int* kclass, int cap, int* n_launches) {
    if (!h || !in_device || !ms || !kclass || !n_launches || reps <= 0 || cap <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = prepare_poses(h, N, H, W, ratio);
    if (rc) return rc;
    h->last_N = N;
    for (int i = 0; i < cap; ++i) { ms[i] = 0.f; kclass[i] = -1; }
    size_t nl = 0;
    for (int r = 0; r < reps; ++r) {
        h->profiling = true;
        h->record_variants = true;
        h->ev_used = 0;
        rc = enqueue_poses(h, in_device, N, H, W, ratio, demo, true);
        h->profiling = false;
        h->record_variants = false;
        if (rc) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        nl = h->ev_used / 2;
        if ((int)nl > cap) return fail(h, LWP_ERR_ARG, "launch arrays too small");
        for (size_t i = 0; i < nl; ++i) {
            float t = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&t, h->ev[2 * i], h->ev[2 * i + 1]));
            ms[i] += t / (float)reps;
            kclass[i] = h->ev_class[i] | ((h->ev_layer[i] + 1) << 8);
        }
    }
    *n_launches = (int)nl;
    return LWP_OK;
}

extern "C" int lwp_profile_classes(lwp_handle h, const float* in_device, int N, int H, int W, int ratio, int demo, int reps,
                                   float* ms, int* launches) {
    if (!h || !in_device || !ms || !launches || reps <= 0) return fail(h, LWP_ERR_ARG, "bad argument");
    int rc = prepare_poses(h, N, H, W, ratio);
    if (rc) return rc;
    h->last_N = N;
    for (int k = 0; k < KC_COUNT; ++k) { ms[k] = 0.f; launches[k] = 0; }
    for (int r = 0; r < reps; ++r) {
        h->profiling = true;
        h->ev_used = 0;
        rc = enqueue_poses(h, in_device, N, H, W, ratio, demo, true);
        h->profiling = false;
        if (rc) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (size_t i = 0; i + 1 < h->ev_used + 1 && i < h->ev_used; i += 2) {
            float t = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
            const int k = h->ev_class[i / 2];
            ms[k] += t;
            if (r == 0) launches[k] += 1;
        }
    }
    for (int k = 0; k < KC_COUNT; ++k) ms[k] /= (float)reps;
    return LWP_OK;
}
