// Host side of the network: parameter table, layer graph, BN folding and weight packing.
//
// Mirrors the structure of the reference network (models/with_mobilenet.py:89-123, built from
// modules/conv.py:4-32) as a flat list of three kernel kinds on NHWC activations:
//   L_STEM  3x3 stride-2 conv 3->32 (+BN+ReLU)                    with_mobilenet.py:93
//   L_DW    depthwise 3x3 (+BN+ReLU | +ELU), stride 1/2, dil 1/2  conv.py:15-17, 27-28
//   L_GEMM  1x1 or dense 3x3 conv as an implicit GEMM, bias/BN folded, ReLU|ELU|none,
//           optional residual add after the activation             conv.py:4-10,19-21,30-31
// torch.cat([features, heat, paf]) (with_mobilenet.py:121) is free: the producing layers write
// channel windows of one [C | NH | NP | pad] buffer.
#include <cmath>
#include <cstring>
#include <map>

#include "lwp_internal.h"

namespace lwp {

static const int kBackbone[11][4] = {  // cin, cout, stride, dilation (with_mobilenet.py:94-104)
    {32, 64, 1, 1},   {64, 128, 2, 1},  {128, 128, 1, 1}, {128, 256, 2, 1}, {256, 256, 1, 1}, {256, 512, 1, 1},
    {512, 512, 1, 2}, {512, 512, 1, 1}, {512, 512, 1, 1}, {512, 512, 1, 1}, {512, 512, 1, 1}};

static std::string fmt(const char* f, int a = 0, int b = 0) {
    char buf[128];
    snprintf(buf, sizeof buf, f, a, b);
    return buf;
}

static void add_conv(std::vector<ParamSpec>& t, const std::string& p, int cin, int cout, int k, int groups, bool bias) {
    ParamSpec w{p + ".weight", {cout, cin / groups, k, k}, 4, LWP_ROLE_CONV_W};
    t.push_back(w);
    if (bias) t.push_back(ParamSpec{p + ".bias", {cout, 1, 1, 1}, 1, LWP_ROLE_CONV_B});
}
static void add_bn(std::vector<ParamSpec>& t, const std::string& p, int c) {
    t.push_back(ParamSpec{p + ".weight", {c, 1, 1, 1}, 1, LWP_ROLE_BN_W});
    t.push_back(ParamSpec{p + ".bias", {c, 1, 1, 1}, 1, LWP_ROLE_BN_B});
    t.push_back(ParamSpec{p + ".running_mean", {c, 1, 1, 1}, 1, LWP_ROLE_BN_MEAN});
    t.push_back(ParamSpec{p + ".running_var", {c, 1, 1, 1}, 1, LWP_ROLE_BN_VAR});
    t.push_back(ParamSpec{p + ".num_batches_tracked", {1, 1, 1, 1}, 0, LWP_ROLE_BN_NBT});
}

std::vector<ParamSpec> param_table(int nref, int C, int NH, int NP) {
    std::vector<ParamSpec> t;
    add_conv(t, "model.0.0", 3, 32, 3, 1, false);
    add_bn(t, "model.0.1", 32);
    for (int i = 0; i < 11; ++i) {
        int cin = kBackbone[i][0], cout = kBackbone[i][1];
        add_conv(t, fmt("model.%d.0", i + 1), cin, cin, 3, cin, false);
        add_bn(t, fmt("model.%d.1", i + 1), cin);
        add_conv(t, fmt("model.%d.3", i + 1), cin, cout, 1, 1, false);
        add_bn(t, fmt("model.%d.4", i + 1), cout);
    }
    add_conv(t, "cpm.align.0", 512, C, 1, 1, true);
    for (int j = 0; j < 3; ++j) {
        add_conv(t, fmt("cpm.trunk.%d.0", j), C, C, 3, C, false);
        add_conv(t, fmt("cpm.trunk.%d.2", j), C, C, 1, 1, false);
    }
    add_conv(t, "cpm.conv.0", C, C, 3, 1, true);
    for (int j = 0; j < 3; ++j) add_conv(t, fmt("initial_stage.trunk.%d.0", j), C, C, 3, 1, true);
    add_conv(t, "initial_stage.heatmaps.0.0", C, 512, 1, 1, true);
    add_conv(t, "initial_stage.heatmaps.1.0", 512, NH, 1, 1, true);
    add_conv(t, "initial_stage.pafs.0.0", C, 512, 1, 1, true);
    add_conv(t, "initial_stage.pafs.1.0", 512, NP, 1, 1, true);
    for (int k = 0; k < nref; ++k) {
        for (int b = 0; b < 5; ++b) {
            int cin = b == 0 ? C + NH + NP : C;
            std::string q = fmt("refinement_stages.%d.trunk.%d", k, b);
            add_conv(t, q + ".initial.0", cin, C, 1, 1, true);
            add_conv(t, q + ".trunk.0.0", C, C, 3, 1, true);
            add_bn(t, q + ".trunk.0.1", C);
            add_conv(t, q + ".trunk.1.0", C, C, 3, 1, true);
            add_bn(t, q + ".trunk.1.1", C);
        }
        std::string p = fmt("refinement_stages.%d", k);
        add_conv(t, p + ".heatmaps.0.0", C, C, 1, 1, true);
        add_conv(t, p + ".heatmaps.1.0", C, NH, 1, 1, true);
        add_conv(t, p + ".pafs.0.0", C, C, 1, 1, true);
        add_conv(t, p + ".pafs.1.0", C, NP, 1, 1, true);
    }
    return t;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

namespace {
struct Builder {
    Graph g;
    int new_buf(int level, int ch) {
        g.bufs.push_back(BufSpec{level, ch});
        return (int)g.bufs.size() - 1;
    }
    static BufRef ref(int buf, int ld, int coff = 0) {
        BufRef r;
        r.buf = buf; r.ld = ld; r.coff = coff;
        return r;
    }
    Layer& add(int kind, const std::string& name, const std::string& conv, const std::string& bn, bool bias,
               int cin, int cout, int ks, int stride, int dil, int act, BufRef src, BufRef dst) {
        Layer l;
        l.kind = kind; l.name = name; l.conv_key = conv; l.bn_key = bn; l.has_bias = bias;
        l.cin = cin; l.cout = cout; l.ks = ks; l.stride = stride; l.dil = dil; l.act = act;
        l.src = src; l.dst = dst;
        g.layers.push_back(l);
        return g.layers.back();
    }
};
}  // namespace

Graph build_graph(int nref, int C, int NH, int NP, bool fuse_dwpw, int dtype, bool merge_heads) {
    Builder b;
    Graph& g = b.g;
    g.nref = nref; g.C = C; g.NH = NH; g.NP = NP; g.dtype = dtype;
    const bool h16 = dtype == LWP_BF16;
    if (h16) fuse_dwpw = true;                    // the bf16 path has no stand-alone depthwise kernel
    // level-1/2 buffers are single-use; level 3 ping-pongs two 512-channel slots and three C-channel slots
    int s1 = b.new_buf(1, 32), d1 = b.new_buf(1, 32), p1 = b.new_buf(1, 64);
    int d2 = b.new_buf(2, 64), p2 = b.new_buf(2, 128), d3 = b.new_buf(2, 128), p3 = b.new_buf(2, 128);
    int X = b.new_buf(3, 512), Y = b.new_buf(3, 512);
    int S0 = b.new_buf(3, C), S1 = b.new_buf(3, C), S2 = b.new_buf(3, C);
    int catc = round_up(C + NH + NP, 64);
    int CAT = b.new_buf(3, catc);
    g.bufs[CAT].has_pad = true;
    const int H2 = b.new_buf(3, 1024 > 2 * C ? 1024 : 2 * C);   // [heat hidden | paf hidden] of the merged head GEMMs
    g.cat_buf = CAT; g.cat_channels = catc;

    b.add(L_STEM, "model.0", "model.0.0", "model.0.1", false, 3, 32, 3, 2, 1, ACT_RELU, Builder::ref(-1, 3), Builder::ref(s1, 32));
    // backbone blocks 1..11: dw then pw
    int src = s1, src_ld = 32;
    int dwbuf[11] = {d1, d2, d3, X, X, X, X, X, X, X, X};
    int pwbuf[11] = {p1, p2, p3, Y, Y, Y, Y, Y, Y, Y, Y};
    for (int i = 0; i < 11; ++i) {
        int cin = kBackbone[i][0], cout = kBackbone[i][1], st = kBackbone[i][2], dl = kBackbone[i][3];
        // blocks 4.. alternate X (dw out) / Y (pw out); block 4's dw reads p3 (level 2)
        if (fuse_dwpw && dwpw_supported(cin, cout)) {
            // the fused kernel reads a 3x3 neighbourhood of `src` while other workgroups write: never in place
            const int dstb = (pwbuf[i] == src) ? (src == Y ? X : Y) : pwbuf[i];
            Layer& f = b.add(L_DWPW, fmt("model.%d.pw", i + 1), fmt("model.%d.0", i + 1), fmt("model.%d.1", i + 1), false, cin, cout, 3, st, dl,
                             ACT_RELU, Builder::ref(src, src_ld), Builder::ref(dstb, cout));
            f.conv2_key = fmt("model.%d.3", i + 1); f.bn2_key = fmt("model.%d.4", i + 1); f.act2 = ACT_RELU;
            src = dstb; src_ld = cout;
            continue;
        } else {
            b.add(L_DW, fmt("model.%d.dw", i + 1), fmt("model.%d.0", i + 1), fmt("model.%d.1", i + 1), false, cin, cin, 3, st, dl,
                  ACT_RELU, Builder::ref(src, src_ld), Builder::ref(dwbuf[i], cin));
            b.add(L_GEMM, fmt("model.%d.pw", i + 1), fmt("model.%d.3", i + 1), fmt("model.%d.4", i + 1), false, cin, cout, 1, 1, 1,
                  ACT_RELU, Builder::ref(dwbuf[i], cin), Builder::ref(pwbuf[i], cout));
        }
        src = pwbuf[i]; src_ld = cout;
    }
    // Cpm (with_mobilenet.py:18-21): a = align(x); feat = conv(a + trunk(a))
    b.add(L_GEMM, "cpm.align", "cpm.align.0", "", true, 512, C, 1, 1, 1, ACT_RELU, Builder::ref(src, 512), Builder::ref(S0, C));
    int tin = S0;
    for (int j = 0; j < 3; ++j) {
        if (fuse_dwpw && dwpw_supported(C, C)) {
            // S0 holds `a` (kept for the residual); the chain alternates S1/S2 and must END in S2
            const int dstb = (j % 2 == 0) ? S2 : S1;
            Layer& f = b.add(L_DWPW, fmt("cpm.trunk.%d.pw", j), fmt("cpm.trunk.%d.0", j), "", false, C, C, 3, 1, 1, ACT_ELU,
                             Builder::ref(tin, C), Builder::ref(dstb, C));
            f.conv2_key = fmt("cpm.trunk.%d.2", j); f.act2 = ACT_ELU;
            if (j == 2) f.res = Builder::ref(S0, C);
            tin = dstb;
        } else {
            b.add(L_DW, fmt("cpm.trunk.%d.dw", j), fmt("cpm.trunk.%d.0", j), "", false, C, C, 3, 1, 1, ACT_ELU,
                  Builder::ref(tin, C), Builder::ref(S1, C));
            Layer& pw = b.add(L_GEMM, fmt("cpm.trunk.%d.pw", j), fmt("cpm.trunk.%d.2", j), "", false, C, C, 1, 1, 1, ACT_ELU,
                              Builder::ref(S1, C), Builder::ref(S2, C));
            if (j == 2) pw.res = Builder::ref(S0, C);   // x + trunk(x), fused into the last pw's epilogue
            tin = S2;
        }
    }
    b.add(L_GEMM, "cpm.conv", "cpm.conv.0", "", true, C, C, 3, 1, 1, ACT_RELU, Builder::ref(S2, C), Builder::ref(CAT, catc, 0));
    // initial stage (with_mobilenet.py:41-45)
    b.add(L_GEMM, "initial_stage.trunk.0", "initial_stage.trunk.0.0", "", true, C, C, 3, 1, 1, ACT_RELU, Builder::ref(CAT, catc, 0), Builder::ref(S0, C));
    b.add(L_GEMM, "initial_stage.trunk.1", "initial_stage.trunk.1.0", "", true, C, C, 3, 1, 1, ACT_RELU, Builder::ref(S0, C), Builder::ref(S1, C));
    b.add(L_GEMM, "initial_stage.trunk.2", "initial_stage.trunk.2.0", "", true, C, C, 3, 1, 1, ACT_RELU, Builder::ref(S1, C), Builder::ref(S0, C));
    // stage heads (with_mobilenet.py:32-45): heatmaps.0 and pafs.0 read the same input -> ONE GEMM with concatenated
    // outputs; heatmaps.1 and pafs.1 -> ONE block-diagonal GEMM writing [heat | paf] into the concat buffer and both
    // NCHW stage outputs.  4 launches per stage become 2 (at batch 1 every launch costs a ~6-10 us floor).
    auto add_heads = [&](const std::string& p, int in_buf, int hidden, int out0) {
        if (!merge_heads) {
            b.add(L_GEMM, p + ".heatmaps.0", p + ".heatmaps.0.0", "", true, C, hidden, 1, 1, 1, ACT_RELU, Builder::ref(in_buf, C), Builder::ref(X, hidden));
            b.add(L_GEMM, p + ".pafs.0", p + ".pafs.0.0", "", true, C, hidden, 1, 1, 1, ACT_RELU, Builder::ref(in_buf, C), Builder::ref(Y, hidden));
            b.add(L_GEMM, p + ".heatmaps.1", p + ".heatmaps.1.0", "", true, hidden, NH, 1, 1, 1, ACT_NONE, Builder::ref(X, hidden), Builder::ref(CAT, catc, C)).out_index = out0;
            b.add(L_GEMM, p + ".pafs.1", p + ".pafs.1.0", "", true, hidden, NP, 1, 1, 1, ACT_NONE, Builder::ref(Y, hidden), Builder::ref(CAT, catc, C + NH)).out_index = out0 + 1;
            return;
        }
        Layer& h0 = b.add(L_GEMM, p + ".heads.0", p + ".heatmaps.0.0", "", true, C, 2 * hidden, 1, 1, 1, ACT_RELU,
                          Builder::ref(in_buf, C), Builder::ref(H2, 2 * hidden));
        h0.blocks = {WBlock{p + ".heatmaps.0.0", 0, 0, hidden, C}, WBlock{p + ".pafs.0.0", hidden, 0, hidden, C}};
        Layer& h1 = b.add(L_GEMM, p + ".heads.1", p + ".heatmaps.1.0", "", true, 2 * hidden, NH + NP, 1, 1, 1, ACT_NONE,
                          Builder::ref(H2, 2 * hidden), Builder::ref(CAT, catc, C));
        h1.blocks = {WBlock{p + ".heatmaps.1.0", 0, 0, NH, hidden}, WBlock{p + ".pafs.1.0", NH, hidden, NP, hidden}};
        h1.out_index = out0; h1.out_index2 = out0 + 1; h1.out_split = NH;
    };
    add_heads("initial_stage", S0, 512, 0);
    // refinement stages (with_mobilenet.py:57-60, 82-86)
    for (int k = 0; k < nref; ++k) {
        BufRef in = Builder::ref(CAT, catc, 0);
        int in_c = C + NH + NP;
        for (int bl = 0; bl < 5; ++bl) {
            std::string q = fmt("refinement_stages.%d.trunk.%d", k, bl);
            b.add(L_GEMM, q + ".initial", q + ".initial.0", "", true, in_c, C, 1, 1, 1, ACT_RELU, in, Builder::ref(S0, C));
            b.add(L_GEMM, q + ".trunk.0", q + ".trunk.0.0", q + ".trunk.0.1", true, C, C, 3, 1, 1, ACT_RELU, Builder::ref(S0, C), Builder::ref(S1, C));
            Layer& c2 = b.add(L_GEMM, q + ".trunk.1", q + ".trunk.1.0", q + ".trunk.1.1", true, C, C, 3, 1, 2, ACT_RELU,
                              Builder::ref(S1, C), Builder::ref(S2, C));
            c2.res = Builder::ref(S0, C);           // initial_features + trunk_features
            in = Builder::ref(S2, C);
            in_c = C;
        }
        add_heads(fmt("refinement_stages.%d", k), S2, C, 2 * (k + 1));
    }
    // weight blob layout
    size_t off = 0;
    for (Layer& l : g.layers) {
        if (l.kind == L_STEM) l.macs_per_pixel = 27 * 32;
        else if (l.kind == L_DW) l.macs_per_pixel = 9 * (int64_t)l.cin;
        else if (l.kind == L_DWPW) l.macs_per_pixel = 9 * (int64_t)l.cin + (int64_t)l.cin * l.cout;
        else if (!l.blocks.empty()) { for (const WBlock& wb : l.blocks) l.macs_per_pixel += (int64_t)wb.cin * wb.cout; }
        else l.macs_per_pixel = (int64_t)l.cin * l.cout * l.ks * l.ks;
        if (l.kind == L_STEM) {
            l.cin_pad = 3; l.cout_pad = 32;
            l.w_off = off; off += 27 * 32;
        } else if (l.kind == L_DW) {
            l.cin_pad = l.cin; l.cout_pad = l.cout;
            l.w_off = off; off += (size_t)9 * l.cin;
        } else if (l.kind == L_DWPW) {
            l.cin_pad = l.cin; l.cout_pad = l.cout;
            l.w_off = off; off += (size_t)9 * l.cin;          // depthwise [9][C] ...
            l.b_off = off; off += l.cin;                      // ... immediately followed by its bias [C]
            l.w2_off = off; off += (size_t)l.cin * l.cout / (h16 ? 2 : 1);   // fragment-packed pointwise weights
            l.b2_off = off; off += l.cout;
            off = (off + 63) / 64 * 64;
            continue;
        } else {
            l.cin_pad = round_up(l.cin, h16 ? 64 : 32);
            l.cout_pad = round_up(l.cout, 64);
            l.w_off = off; off += (size_t)l.ks * l.ks * l.cout_pad * l.cin_pad / (h16 ? 2 : 1);
            if (!h16) {                                       // second copy in MFMA fragment order (32-row tile kernel)
                off = (off + 63) / 64 * 64;
                l.w2_off = off; off += (size_t)l.ks * l.ks * l.cout_pad * l.cin_pad;
            }
        }
        l.b_off = off; off += l.cout_pad;
        off = (off + 63) / 64 * 64;
    }
    g.blob_floats = off;
    return g;
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);   // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

std::string pack_weights(const Graph& g, const std::vector<std::string>& names, const std::vector<HostTensor>& tensors,
                         std::vector<float>& blob) {
    std::map<std::string, const HostTensor*> by_name;
    for (size_t i = 0; i < names.size(); ++i) by_name[names[i]] = &tensors[i];
    // every key of the table must be there with the right shape (the Python load_state has already
    // substituted the net's own values for missing / mismatching checkpoint entries)
    for (const ParamSpec& p : param_table(g.nref, g.C, g.NH, g.NP)) {
        auto it = by_name.find(p.key);
        if (it == by_name.end()) return "missing parameter '" + p.key + "'";
        if (p.role == LWP_ROLE_BN_NBT) continue;
        const HostTensor& t = *it->second;
        if (t.ndim != p.ndim) return "rank mismatch for '" + p.key + "'";
        for (int d = 0; d < p.ndim; ++d)
            if (t.shape[d] != p.shape[d]) return "shape mismatch for '" + p.key + "'";
        if (!t.ptr) return "null data for '" + p.key + "'";
    }
    auto f32 = [&](const std::string& k) { return (const float*)by_name[k]->ptr; };
    blob.assign(g.blob_floats, 0.0f);
    auto fold = [&](const std::string& conv_key, const std::string& bn_key, bool has_bias, int co, std::vector<double>& scale,
                    std::vector<double>& shift) {   // y = conv*scale + shift
        scale.assign(co, 1.0); shift.assign(co, 0.0);
        const float* cb = has_bias ? f32(conv_key + ".bias") : nullptr;
        if (!bn_key.empty()) {
            const float* gam = f32(bn_key + ".weight");
            const float* bet = f32(bn_key + ".bias");
            const float* mu = f32(bn_key + ".running_mean");
            const float* var = f32(bn_key + ".running_var");
            for (int o = 0; o < co; ++o) {
                scale[o] = (double)gam[o] / std::sqrt((double)var[o] + 1e-5);
                shift[o] = ((cb ? (double)cb[o] : 0.0) - (double)mu[o]) * scale[o] + (double)bet[o];
            }
        } else if (cb) {
            for (int o = 0; o < co; ++o) shift[o] = cb[o];
        }
    };
    for (const Layer& l : g.layers) {
        std::vector<double> scale, shift;
        const int co = (l.kind == L_DWPW) ? l.cin : l.cout;   // channels of the FIRST conv of the layer
        if (l.blocks.empty()) fold(l.conv_key, l.bn_key, l.has_bias, co, scale, shift);
        else { scale.assign(co, 1.0); shift.assign(co, 0.0); }
        const float* w = f32(l.conv_key + ".weight");
        float* wp = blob.data() + l.w_off;
        float* bp = blob.data() + l.b_off;
        for (int o = 0; o < co; ++o) bp[o] = (float)shift[o];
        if (l.kind == L_STEM) {                 // OIHW (32,3,3,3) -> [(ky,kx,ci)][oc]
            for (int o = 0; o < 32; ++o)
                for (int ci = 0; ci < 3; ++ci)
                    for (int t = 0; t < 9; ++t)
                        wp[(t * 3 + ci) * 32 + o] = (float)((double)w[(o * 3 + ci) * 9 + t] * scale[o]);
        } else if (l.kind == L_DW || l.kind == L_DWPW) {   // (C,1,3,3) -> [tap][C]
            for (int c = 0; c < co; ++c)
                for (int t = 0; t < 9; ++t) wp[(size_t)t * co + c] = (float)((double)w[c * 9 + t] * scale[c]);
        } else if (!l.blocks.empty()) {         // merged 1x1 layer: assemble [cout_pad][cin_pad] from its source convs
            uint16_t* wh = (uint16_t*)wp;
            for (const WBlock& wb : l.blocks) {
                const float* ws = f32(wb.conv_key + ".weight");
                const float* bs = f32(wb.conv_key + ".bias");
                for (int o = 0; o < wb.cout; ++o) {
                    bp[wb.out_off + o] = bs[o];
                    for (int ci = 0; ci < wb.cin; ++ci) {
                        const size_t idx = (size_t)(wb.out_off + o) * l.cin_pad + wb.in_off + ci;
                        const float v = ws[(size_t)o * wb.cin + ci];
                        if (g.dtype == LWP_BF16) wh[idx] = f32_to_bf16_rne(v);
                        else wp[idx] = v;
                    }
                }
            }
        } else {                                // OIHW -> [tap][cout_pad][cin_pad]  (f32 or bf16)
            const int taps = l.ks * l.ks;
            uint16_t* wh = (uint16_t*)wp;
            for (int o = 0; o < co; ++o)
                for (int ci = 0; ci < l.cin; ++ci)
                    for (int t = 0; t < taps; ++t) {
                        const float v = (float)((double)w[((size_t)o * l.cin + ci) * taps + t] * scale[o]);
                        const size_t idx = ((size_t)t * l.cout_pad + o) * l.cin_pad + ci;
                        if (g.dtype == LWP_BF16) wh[idx] = f32_to_bf16_rne(v);
                        else wp[idx] = v;
                    }
        }
        if (l.kind == L_GEMM && g.dtype != LWP_BF16) {
            // fragment order for v_mfma_f32_32x32x2_f32 as gemm_wp_kernel walks K: [tap][k-step 32][32-channel tile][s 0..3]
            // [lane 64][4]: lane (r = lane & 31, h = lane >> 5) holds W[tile*32 + r][step*32 + 8s + 4h + j], j = 0..3, so
            // one wave-wide 16-byte load is 1 KiB contiguous
            float* wf = blob.data() + l.w2_off;
            const int taps = l.ks * l.ks, ksteps = l.cin_pad / 32, ntiles = l.cout_pad / 32;
            for (int t = 0; t < taps; ++t)
                for (int ks_ = 0; ks_ < ksteps; ++ks_)
                    for (int nt = 0; nt < ntiles; ++nt)
                        for (int sq = 0; sq < 4; ++sq)
                            for (int ln = 0; ln < 64; ++ln)
                                for (int j = 0; j < 4; ++j) {
                                    const size_t src = ((size_t)t * l.cout_pad + nt * 32 + (ln & 31)) * l.cin_pad + ks_ * 32 + 8 * sq + 4 * (ln >> 5) + j;
                                    const size_t dst = (((((size_t)t * ksteps + ks_) * ntiles + nt) * 4 + sq) * 64 + ln) * 4 + j;
                                    wf[dst] = wp[src];
                                }
        }
        if (l.kind == L_DWPW) {
            // pointwise half: (cout, C, 1, 1) -> MFMA fragment order [C/32][cout/32][4][64 lanes][4]; lane (q = lane>>4,
            // c = lane&15) holds value v = u*8 + j*2 + t  =  W[n = 32w + 16t + c][k = 32s + 16u + 4q + j]
            std::vector<double> sc2, sh2;
            fold(l.conv2_key, l.bn2_key, false, l.cout, sc2, sh2);
            const float* w2 = f32(l.conv2_key + ".weight");
            float* w2p = blob.data() + l.w2_off;
            float* b2p = blob.data() + l.b2_off;
            for (int o = 0; o < l.cout; ++o) b2p[o] = (float)sh2[o];
            const int nw = l.cout / 32, C_ = l.cin;
            if (g.dtype == LWP_BF16) {
                // bf16: MFMA 16x16x32 operand order [C/32][cout/32][2 tiles][64 lanes][8]; lane (q = lane>>4, i = lane&15)
                // holds W[n = 32w + 8 (i >> 2) + 4 t + (i & 3)][k = 32s + 8q + j], j = 0..7
                uint16_t* wh = (uint16_t*)w2p;
                for (int s = 0; s < C_ / 32; ++s)
                    for (int wv = 0; wv < nw; ++wv)
                        for (int t = 0; t < 2; ++t)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    // rows of a 32-channel block are permuted: MFMA row i of tile t holds channel 8 (i >> 2) + 4 t + (i & 3),
                                    // so that a lane's two accumulator tiles are 8 consecutive channels of its pixel (16-byte stores)
                                    const int i_ = lane & 15;
                                    const int k = 32 * s + 8 * (lane >> 4) + j, n = 32 * wv + 8 * (i_ >> 2) + 4 * t + (i_ & 3);
                                    wh[((((size_t)(s * nw + wv) * 2 + t) * 64 + lane) * 8) + j] =
                                        f32_to_bf16_rne((float)((double)w2[(size_t)n * C_ + k] * sc2[n]));
                                }
            } else {
            for (int s = 0; s < C_ / 32; ++s)
                for (int wv = 0; wv < nw; ++wv)
                    for (int v = 0; v < 16; ++v)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int u = v >> 3, j = (v >> 1) & 3, t = v & 1, q = lane >> 4, c = lane & 15;
                            const int k = 32 * s + 16 * u + 4 * q + j, n = 32 * wv + 16 * t + c;
                            w2p[(((size_t)(s * nw + wv) * 4 + (v >> 2)) * 64 + lane) * 4 + (v & 3)] =
                                (float)((double)w2[(size_t)n * C_ + k] * sc2[n]);
                        }
            }
        }
    }
    return "";
}

}  // namespace lwp
