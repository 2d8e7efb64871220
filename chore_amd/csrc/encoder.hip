// encoder.hip -- host-side program of the stacked-hourglass encoder: weight arena layout, workspace
// planning and the launch sequence behind chore_encode_fwd.
//
// Topology restated from model/HGFilters.py:57-185 (HGFilter, HourGlass) and
// model/net_util.py:346-396 (ConvBlock); see SURVEY.md Appendix A.1.  For one (B,H,W,dtype) the
// launch list is built once (buffers are planned with a small pool allocator so the working set
// stays inside the 256 MB Infinity Cache where possible) and cached on the handle; running it is a
// plain loop of kernel launches on the caller's stream (hipGraph-capturable: no allocation, no sync).
#include "enc_common.h"
#include <map>
#include <memory>
#include <functional>

namespace {

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
size_t esize(int dtype) { return dtype == CHORE_F32 ? 4 : 2; }

// ------------------------------------------------------------------------------------------------
// weight arena: name -> (offset, bytes), deterministic traversal shared by pack and run
// ------------------------------------------------------------------------------------------------
struct WEntry {
    size_t off;
    size_t bytes;
    int kind;  // 0 conv (packed), 1 fp32 vector (bias / gamma / beta), 2 stem weights
    int taps, cin, cout;
};

struct WLayout {
    std::map<std::string, WEntry> e;
    std::vector<std::string> order;
    size_t total = 0;
    void add(const std::string& name, size_t bytes, int kind, int taps = 0, int cin = 0, int cout = 0) {
        total = align_up(total, 256);
        e[name] = WEntry{total, bytes, kind, taps, cin, cout};
        order.push_back(name);
        total += bytes;
    }
};

void layout_gn(WLayout& L, const std::string& n, int C) {
    L.add(n + ".weight", (size_t)C * 4, 1, 0, 0, C);
    L.add(n + ".bias", (size_t)C * 4, 1, 0, 0, C);
}
void layout_conv(WLayout& L, int dtype, const std::string& n, int taps, int cin, int cout, bool bias) {
    L.add(n + ".weight", packed_conv_bytes(dtype, taps, cin, cout), 0, taps, cin, cout);
    if (bias) L.add(n + ".bias", (size_t)cout * 4, 1, 0, 0, cout);
}
void layout_block(WLayout& L, int dtype, const std::string& n, int cin, int cout) {
    layout_conv(L, dtype, n + ".conv1", 9, cin, cout / 2, false);
    layout_conv(L, dtype, n + ".conv2", 9, cout / 2, cout / 4, false);
    layout_conv(L, dtype, n + ".conv3", 9, cout / 4, cout / 4, false);
    layout_gn(L, n + ".bn1", cin);
    layout_gn(L, n + ".bn2", cout / 2);
    layout_gn(L, n + ".bn3", cout / 4);
    layout_gn(L, n + ".bn4", cin);
    if (cin != cout) layout_conv(L, dtype, n + ".downsample.2", 1, cin, cout, false);
}
void layout_hg(WLayout& L, int dtype, const std::string& n, int level) {
    layout_block(L, dtype, n + ".b1_" + std::to_string(level), 256, 256);
    layout_block(L, dtype, n + ".b2_" + std::to_string(level), 256, 256);
    if (level > 1) layout_hg(L, dtype, n, level - 1);
    else layout_block(L, dtype, n + ".b2_plus_" + std::to_string(level), 256, 256);
    layout_block(L, dtype, n + ".b3_" + std::to_string(level), 256, 256);
}

WLayout make_layout(const chore_encoder_cfg& cfg, int dtype) {
    WLayout L;
    const std::string p = "image_filter.";
    L.add(p + "conv1.weight", (size_t)cfg.in_channels * 49 * 64 * 4, 2, 49, cfg.in_channels, 64);
    L.add(p + "conv1.bias", 64 * 4, 1, 0, 0, 64);
    layout_gn(L, p + "bn1", 64);
    layout_block(L, dtype, p + "conv2", 64, 128);
    layout_block(L, dtype, p + "conv3", 128, 128);
    layout_block(L, dtype, p + "conv4", 128, 256);
    for (int i = 0; i < cfg.num_stack; ++i) {
        const std::string s = std::to_string(i);
        layout_hg(L, dtype, p + "m" + s, cfg.num_hourglass);
        layout_block(L, dtype, p + "top_m_" + s, 256, 256);
        layout_conv(L, dtype, p + "conv_last" + s, 1, 256, 256, true);
        layout_gn(L, p + "bn_end" + s, 256);
        layout_conv(L, dtype, p + "l" + s, 1, 256, cfg.hourglass_dim, true);
        if (i < cfg.num_stack - 1) {
            layout_conv(L, dtype, p + "bl" + s, 1, 256, 256, true);
            layout_conv(L, dtype, p + "al" + s, 1, cfg.hourglass_dim, 256, true);
        }
    }
    L.total = align_up(L.total, 256);
    return L;
}

int check_cfg(chore_handle* h, const chore_encoder_cfg* cfg) {
    if (!cfg) CHORE_FAIL(h, CHORE_EINVAL, "encoder: null cfg");
    if (cfg->in_channels < 1 || cfg->in_channels > 8 || cfg->num_stack < 1 || cfg->num_stack > 16 ||
        cfg->num_hourglass < 1 || cfg->num_hourglass > 4 || cfg->hourglass_dim != 256)
        CHORE_FAIL(h, CHORE_EINVAL, "encoder: unsupported cfg (in=%d stacks=%d depth=%d dim=%d)", cfg->in_channels,
                   cfg->num_stack, cfg->num_hourglass, cfg->hourglass_dim);
    return CHORE_OK;
}

// ------------------------------------------------------------------------------------------------
// program = list of closures; buffers are offsets into the workspace (or caller tensors)
// ------------------------------------------------------------------------------------------------
// GroupNorm tile partials attached to a buffer (written by the conv epilogues that produce it)
struct StatInfo {
    bool valid = false;    // storage attached
    bool usable = true;    // false once the buffer has been modified after the producing convs
    size_t off = 0, bytes = 0;
    int max_tiles = 0, nslices = 0;
    int c_end[3] = {0, 0, 0}, ntiles[3] = {0, 0, 0};
};

struct Buf {
    size_t off = 0;       // offset into workspace, or
    int ext = -1;         // index of an external (caller) tensor: 0..n_out-1 feats, 100 tmpx, 101 normx
    int H = 0, W = 0, C = 0;
    size_t bytes = 0;
    StatInfo st;
};

struct Pool {
    size_t top = 0;
    std::multimap<size_t, size_t> free_list;  // bytes -> offset
    size_t alloc(size_t bytes) {
        bytes = align_up(bytes, 256);
        auto it = free_list.lower_bound(bytes);
        if (it != free_list.end() && it->first == bytes) {
            const size_t off = it->second;
            free_list.erase(it);
            return off;
        }
        const size_t off = top;
        top += bytes;
        return off;
    }
    void release(size_t off, size_t bytes) { free_list.emplace(align_up(bytes, 256), off); }
};

struct RunCtx {
    chore_handle* h;
    int dtype;
    hipStream_t s;
    char* ws;
    const char* arena;
    const float* images;
    void* const* feats;
    void* tmpx;
    void* normx;
    int rc = CHORE_OK;
};

enum KClass { K_STEM = 0, K_GN_STATS, K_GN_APPLY, K_CONV3_128, K_CONV3_64, K_CONV3_32, K_CONV1, K_POOL, K_UPADD, K_NUM };
const char* const kclass_names[K_NUM] = {"stem_conv7x7", "gn_stats", "gn_apply_relu", "conv3x3_n128", "conv3x3_n64",
                                         "conv3x3_n32", "conv1x1", "avgpool2", "bicubic_upadd"};

struct Step {
    std::string label;
    std::function<void(RunCtx&)> fn;
    int klass = 0;
    double flops = 0.0;  // algorithmic FLOPs of the launch (convolutions: 2*taps*Cin*Cout*B*H*W)
    double bytes = 0.0;  // algorithmic HBM bytes of the launch (compulsory reads + writes)
};

struct Profile {
    bool on = false;
    double ms[K_NUM] = {0}, flops[K_NUM] = {0}, bytes[K_NUM] = {0};
    long long launches[K_NUM] = {0};
    std::vector<hipEvent_t> ev;
};

struct Program {
    chore_encoder_cfg cfg;
    int B, H, W, dtype, n_out;
    bool want_normx;
    WLayout L;
    std::vector<Step> steps;
    size_t ws_bytes = 0;
};

struct Builder {
    Program& P;
    Pool pool;
    int B, dtype;
    size_t partial_off;
    std::string cur_label = "?";
    int cur_class = 0;
    double cur_flops = 0.0, cur_bytes = 0.0;
    double es() const { return (double)esize(dtype); }

    explicit Builder(Program& p) : P(p), B(p.B), dtype(p.dtype) {
        partial_off = pool.alloc((size_t)B * GN_SPLITS_MAX * GN_GROUPS * 2 * 4);
    }

    Buf alloc(int H, int W, int C) {
        Buf b;
        b.H = H; b.W = W; b.C = C;
        b.bytes = (size_t)B * H * W * C * esize(dtype);
        b.off = pool.alloc(b.bytes);
        return b;
    }
    Buf external(int id, int H, int W, int C) {
        Buf b;
        b.ext = id; b.H = H; b.W = W; b.C = C;
        return b;
    }
    void release(const Buf& b) {
        if (b.ext < 0) pool.release(b.off, b.bytes);
        if (b.st.valid) pool.release(b.st.off, b.st.bytes);
    }
    // attach tile-partial storage for the slices [0,c_end0), [c_end0,c_end1) ... each produced by a conv
    // with `cout_i` output channels (which fixes its tile configuration)
    void attach_stats(Buf& b, int taps, int nslices, const int* c_end, const int* couts) {
        StatInfo& st = b.st;
        st.valid = true;
        st.nslices = nslices;
        st.max_tiles = 0;
        for (int i = 0; i < nslices; ++i) {
            st.c_end[i] = c_end[i];
            st.ntiles[i] = conv_plan(taps, B, b.H, b.W, couts[i]).ntiles;
            if (st.ntiles[i] > st.max_tiles) st.max_tiles = st.ntiles[i];
        }
        st.bytes = (size_t)B * st.max_tiles * b.C * 2 * 4;
        st.off = pool.alloc(st.bytes);
    }
    size_t alloc_ss(int C) { return pool.alloc((size_t)B * C * 2 * 4); }
    void release_ss(size_t off, int C) { pool.release(off, (size_t)B * C * 2 * 4); }

    static void* ptr(RunCtx& r, const Buf& b) {
        if (b.ext < 0) return r.ws + b.off;
        if (b.ext == 100) return r.tmpx;
        if (b.ext == 101) return r.normx;
        return r.feats[b.ext];
    }
    static View view(RunCtx& r, const Buf& b, int co = 0, int C = -1) {
        View v;
        v.p = ptr(r, b);
        v.cs = b.C;
        v.co = co;
        v.C = C < 0 ? b.C : C;
        return v;
    }
    const WEntry& w(const std::string& name) const {
        auto it = P.L.e.find(name);
        if (it == P.L.e.end()) abort();
        return it->second;
    }

    // ---- op emitters ----
    // GroupNorm statistics of view (x, co, C) -> scale/shift for `gn` (and optionally a second set
    // `gn2` sharing the same statistics: bn1 / bn4 of a ConvBlock normalise the same tensor)
    void gn_stats(const Buf& x, int co, int C, const std::string& gn, size_t ss, const std::string* gn2 = nullptr,
                  size_t ss2 = 0) {
        const WEntry g = w(gn + ".weight"), bt = w(gn + ".bias");
        WEntry g2{}, bt2{};
        const bool two = gn2 != nullptr;
        if (two) { g2 = w(*gn2 + ".weight"); bt2 = w(*gn2 + ".bias"); }
        const size_t poff = partial_off;
        const int HW = x.H * x.W;
        if (x.st.valid && x.st.usable && co == 0 && C == x.C) {   // statistics already produced by the conv epilogues
            const StatInfo st = x.st;
            cur_label = "gn_finalize_tiles " + gn;
            cur_class = K_GN_STATS; cur_flops = 0.0; cur_bytes = (double)B * st.max_tiles * C * 8;
            const int Bn = B;
            P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
                if (r.rc) return;
                TileStats ts;
                ts.p = (const float*)(r.ws + st.off);
                ts.max_tiles = st.max_tiles; ts.nslices = st.nslices;
                for (int i = 0; i < 3; ++i) { ts.c_end[i] = st.c_end[i]; ts.ntiles[i] = st.ntiles[i]; }
                r.rc = launch_gn_finalize_tiles(r.h, ts, Bn, HW, C, (const float*)(r.arena + g.off),
                                                (const float*)(r.arena + bt.off), (float*)(r.ws + ss), r.s);
                if (r.rc || !two) return;
                r.rc = launch_gn_finalize_tiles(r.h, ts, Bn, HW, C, (const float*)(r.arena + g2.off),
                                                (const float*)(r.arena + bt2.off), (float*)(r.ws + ss2), r.s);
            }, cur_class, cur_flops, cur_bytes});
            return;
        }
        cur_label = "gn_stats " + gn;
        cur_class = K_GN_STATS; cur_flops = 0.0; cur_bytes = (double)B * HW * C * es();
        const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
        P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
            if (r.rc) return;
            float* partial = (float*)(r.ws + poff);
            r.rc = launch_gn_partial(r.h, r.dtype, view(r, x, co, C), Bn, HW, partial, r.s);
            if (r.rc) return;
            r.rc = launch_gn_finalize(r.h, partial, Bn, HW, C, (const float*)(r.arena + g.off),
                                      (const float*)(r.arena + bt.off), (float*)(r.ws + ss), r.s);
            if (r.rc || !two) return;
            r.rc = launch_gn_finalize(r.h, partial, Bn, HW, C, (const float*)(r.arena + g2.off),
                                      (const float*)(r.arena + bt2.off), (float*)(r.ws + ss2), r.s);
        }, cur_class, cur_flops, cur_bytes});
    }

    struct ConvSpec {
        Buf in; int in_co = 0, in_C = 0;
        bool use_ss = false; size_t ss = 0;
        std::string wname; bool bias = false;
        Buf out; int out_co = 0;
        bool has_raw = false; Buf raw; int raw_co = 0;
        bool has_res = false; Buf res; int res_co = 0;
        bool has_res2 = false; Buf res2; int res2_co = 0;
        int taps = 9, cout = 0;
        bool stat_raw = false, stat_out = false;   // emit GroupNorm tile partials into raw.st / out.st
    };
    void conv(const ConvSpec& c) {
        const WEntry we = w(c.wname + ".weight");
        WEntry be{};
        if (c.bias) be = w(c.wname + ".bias");
        const ConvSpec cs = c;
        cur_label = "conv " + c.wname + " taps=" + std::to_string(c.taps) + " cin=" + std::to_string(c.in_C) +
                    " cout=" + std::to_string(c.cout) + " HxW=" + std::to_string(c.in.H) + "x" + std::to_string(c.in.W);
        {
            const double px = (double)B * c.in.H * c.in.W;
            cur_class = c.taps == 1 ? K_CONV1 : (c.cout % 128 == 0 ? K_CONV3_128 : (c.cout == 64 ? K_CONV3_64 : K_CONV3_32));
            cur_flops = 2.0 * c.taps * c.in_C * c.cout * px;
            cur_bytes = px * es() * (c.in_C + c.cout * (1 + (c.has_raw ? 1 : 0) + (c.has_res ? 1 : 0) + (c.has_res2 ? 1 : 0)));
        }
        const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
        P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
            if (r.rc) return;
            ConvArgs a{};
            a.in = view(r, cs.in, cs.in_co, cs.in_C);
            a.ss = cs.use_ss ? (const float*)(r.ws + cs.ss) : nullptr;
            a.wpk = r.arena + we.off;
            a.bias = cs.bias ? (const float*)(r.arena + be.off) : nullptr;
            a.out = view(r, cs.out, cs.out_co, cs.cout);
            if (cs.has_raw) a.raw = view(r, cs.raw, cs.raw_co, cs.cout);
            if (cs.has_res) a.res = view(r, cs.res, cs.res_co, cs.cout);
            if (cs.has_res2) a.res2 = view(r, cs.res2, cs.res2_co, cs.cout);
            a.B = Bn; a.H = cs.in.H; a.W = cs.in.W; a.Cout = cs.cout;
            if (cs.stat_raw) {
                a.st_raw = (float*)(r.ws + cs.raw.st.off); a.st_raw_C = cs.raw.C; a.st_raw_co = cs.raw_co;
                a.st_raw_tiles = cs.raw.st.max_tiles;
            }
            if (cs.stat_out) {
                a.st_out = (float*)(r.ws + cs.out.st.off); a.st_out_C = cs.out.C; a.st_out_co = cs.out_co;
                a.st_out_tiles = cs.out.st.max_tiles;
            }
            r.rc = launch_conv(r.h, r.dtype, cs.taps, a, r.s);
        }, cur_class, cur_flops, cur_bytes});
    }

    // ConvBlock (net_util.py:374-396): y = cat(o1,o2,o3) + residual.  `out` may be an external tensor.
    Buf conv_block(const Buf& x, const std::string& n, int cin, int cout, const Buf* out_opt = nullptr) {
        const int H = x.H, W = x.W;
        Buf out = out_opt ? *out_opt : alloc(H, W, cout);
        Buf o1 = alloc(H, W, cout / 2), o2 = alloc(H, W, cout / 4);
        {
            const int ce_out[3] = {cout / 2, 3 * cout / 4, cout}, co_out[3] = {cout / 2, cout / 4, cout / 4};
            attach_stats(out, 9, 3, ce_out, co_out);
            const int ce1[1] = {cout / 2}, co1[1] = {cout / 2}, ce2[1] = {cout / 4}, co2[1] = {cout / 4};
            attach_stats(o1, 9, 1, ce1, co1);
            attach_stats(o2, 9, 1, ce2, co2);
        }
        const size_t ss1 = alloc_ss(cin);
        Buf res = x;
        if (cin != cout) {
            const size_t ss4 = alloc_ss(cin);
            const std::string bn4 = n + ".bn4";
            gn_stats(x, 0, cin, n + ".bn1", ss1, &bn4, ss4);
            ConvSpec d;  // residual = conv1x1(relu(gn4(x)))
            d.in = x; d.in_C = cin; d.use_ss = true; d.ss = ss4; d.wname = n + ".downsample.2";
            d.out = out; d.taps = 1; d.cout = cout;
            conv(d);
            release_ss(ss4, cin);
            res = out;
        } else {
            gn_stats(x, 0, cin, n + ".bn1", ss1);
        }
        ConvSpec c1;
        c1.in = x; c1.in_C = cin; c1.use_ss = true; c1.ss = ss1; c1.wname = n + ".conv1";
        c1.out = out; c1.out_co = 0; c1.has_raw = true; c1.raw = o1; c1.has_res = true; c1.res = res; c1.res_co = 0;
        c1.cout = cout / 2; c1.stat_raw = true; c1.stat_out = true;
        conv(c1);
        release_ss(ss1, cin);
        const size_t ss2 = alloc_ss(cout / 2);
        gn_stats(o1, 0, cout / 2, n + ".bn2", ss2);
        ConvSpec c2;
        c2.in = o1; c2.in_C = cout / 2; c2.use_ss = true; c2.ss = ss2; c2.wname = n + ".conv2";
        c2.out = out; c2.out_co = cout / 2; c2.has_raw = true; c2.raw = o2; c2.has_res = true; c2.res = res;
        c2.res_co = cout / 2; c2.cout = cout / 4; c2.stat_raw = true; c2.stat_out = true;
        conv(c2);
        release_ss(ss2, cout / 2);
        const size_t ss3 = alloc_ss(cout / 4);
        gn_stats(o2, 0, cout / 4, n + ".bn3", ss3);
        ConvSpec c3;
        c3.in = o2; c3.in_C = cout / 4; c3.use_ss = true; c3.ss = ss3; c3.wname = n + ".conv3";
        c3.out = out; c3.out_co = 3 * cout / 4; c3.has_res = true; c3.res = res; c3.res_co = 3 * cout / 4;
        c3.cout = cout / 4; c3.stat_out = true;
        conv(c3);
        release_ss(ss3, cout / 4);
        release(o1);
        release(o2);
        return out;
    }

    Buf pool2(const Buf& x, const Buf* out_opt = nullptr) {
        Buf y = out_opt ? *out_opt : alloc(x.H / 2, x.W / 2, x.C);
        cur_label = "avgpool2";
        cur_class = K_POOL; cur_flops = 0.0; cur_bytes = (double)B * x.H * x.W * x.C * es() * 1.25;
        const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
        P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
            if (r.rc) return;
            r.rc = launch_avgpool2(r.h, r.dtype, view(r, x), view(r, y), Bn, x.H, x.W, r.s);
        }, cur_class, cur_flops, cur_bytes});
        return y;
    }
    void upadd(const Buf& a, const Buf& low) {  // a += bicubic_up2(low)
        cur_label = "upadd";
        cur_class = K_UPADD; cur_flops = 0.0; cur_bytes = (double)B * low.H * low.W * low.C * es() * 9.0;
        const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
        P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
            if (r.rc) return;
            r.rc = launch_upadd(r.h, r.dtype, view(r, a), view(r, low), view(r, a), Bn, low.H, low.W, r.s);
        }, cur_class, cur_flops, cur_bytes});
    }

    // HourGlass._forward (HGFilters.py:26-50)
    Buf hourglass(const Buf& x, const std::string& n, int level) {
        const std::string l = std::to_string(level);
        Buf up1 = conv_block(x, n + ".b1_" + l, 256, 256);
        Buf pooled = pool2(x);
        Buf low1 = conv_block(pooled, n + ".b2_" + l, 256, 256);
        release(pooled);
        Buf low2 = (level > 1) ? hourglass(low1, n, level - 1) : conv_block(low1, n + ".b2_plus_" + l, 256, 256);
        release(low1);
        Buf low3 = conv_block(low2, n + ".b3_" + l, 256, 256);
        release(low2);
        upadd(up1, low3);
        up1.st.usable = false;   // modified in place: the conv-epilogue statistics no longer describe it
        release(low3);
        return up1;
    }

    void build() {
        const chore_encoder_cfg& cfg = P.cfg;
        const std::string p = "image_filter.";
        const int H2 = P.H / 2, W2 = P.W / 2, H4 = P.H / 4, W4 = P.W / 4;
        // stem: conv7x7/2 + GN + ReLU -> tmpx (HGFilters.py:149-150)
        Buf c1 = alloc(H2, W2, 64);
        {
            const WEntry we = w(p + "conv1.weight"), be = w(p + "conv1.bias");
            const int Cin = cfg.in_channels, H = P.H, W = P.W;
            cur_label = "stem";
            cur_class = K_STEM; cur_flops = 2.0 * 49 * Cin * 64 * (double)B * (H / 2) * (W / 2);
            cur_bytes = (double)B * H * W * Cin * 4 + (double)B * (H / 2) * (W / 2) * 64 * es();
            const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
            P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
                if (r.rc) return;
                r.rc = launch_stem(r.h, r.dtype, r.images, Bn, Cin, H, W, (const float*)(r.arena + we.off),
                                   (const float*)(r.arena + be.off), ptr(r, c1), r.s);
            }, cur_class, cur_flops, cur_bytes});
        }
        Buf tmpx = external(100, H2, W2, 64);
        {
            const size_t ss = alloc_ss(64);
            gn_stats(c1, 0, 64, p + "bn1", ss);
            cur_label = "gn_apply_relu bn1";
            cur_class = K_GN_APPLY; cur_flops = 0.0; cur_bytes = 2.0 * B * H2 * W2 * 64 * es();
            const int Bn = B;  // lambdas must not capture `this` (the Builder dies after build())
            P.steps.push_back(Step{cur_label, [=](RunCtx& r) {
                if (r.rc) return;
                r.rc = launch_gn_apply_relu(r.h, r.dtype, view(r, c1), (const float*)(r.ws + ss), view(r, tmpx), Bn,
                                            H2 * W2, r.s);
            }, cur_class, cur_flops, cur_bytes});
            release_ss(ss, 64);
        }
        release(c1);
        Buf b2 = conv_block(tmpx, p + "conv2", 64, 128);
        Buf normx = P.want_normx ? external(101, H4, W4, 128) : alloc(H4, W4, 128);
        pool2(b2, &normx);
        release(b2);
        Buf x3 = conv_block(normx, p + "conv3", 128, 128);
        release(normx);
        Buf previous = conv_block(x3, p + "conv4", 128, 256);
        release(x3);
        for (int i = 0; i < cfg.num_stack; ++i) {
            const std::string s = std::to_string(i);
            Buf hg = hourglass(previous, p + "m" + s, cfg.num_hourglass);
            Buf t1 = conv_block(hg, p + "top_m_" + s, 256, 256);
            release(hg);
            Buf t2 = alloc(H4, W4, 256);
            {
                const int ce[1] = {256}, co[1] = {256};
                attach_stats(t2, 1, 1, ce, co);
            }
            ConvSpec cl;
            cl.in = t1; cl.in_C = 256; cl.wname = p + "conv_last" + s; cl.bias = true; cl.out = t2; cl.taps = 1;
            cl.cout = 256; cl.stat_out = true;
            conv(cl);
            release(t1);
            const size_t ss = alloc_ss(256);
            gn_stats(t2, 0, 256, p + "bn_end" + s, ss);
            const int oi = i - (cfg.num_stack - P.n_out);
            Buf out_i = (oi >= 0) ? external(oi, H4, W4, 256) : alloc(H4, W4, 256);
            ConvSpec l;
            l.in = t2; l.in_C = 256; l.use_ss = true; l.ss = ss; l.wname = p + "l" + s; l.bias = true; l.out = out_i;
            l.taps = 1; l.cout = cfg.hourglass_dim;
            conv(l);
            if (i < cfg.num_stack - 1) {
                Buf nprev = alloc(H4, W4, 256);
                {
                    const int ce[1] = {256}, co[1] = {256};
                    attach_stats(nprev, 1, 1, ce, co);
                }
                ConvSpec bl;
                bl.in = t2; bl.in_C = 256; bl.use_ss = true; bl.ss = ss; bl.wname = p + "bl" + s; bl.bias = true;
                bl.out = nprev; bl.has_res = true; bl.res = previous; bl.taps = 1; bl.cout = 256;
                conv(bl);
                ConvSpec al;
                al.in = out_i; al.in_C = 256; al.wname = p + "al" + s; al.bias = true; al.out = nprev;
                al.has_res = true; al.res = nprev; al.taps = 1; al.cout = 256; al.stat_out = true;
                conv(al);
                release(previous);
                previous = nprev;
            }
            release_ss(ss, 256);
            release(t2);
            release(out_i);
        }
        release(previous);
        P.ws_bytes = align_up(pool.top, 256);
    }
};

struct EncCache {
    std::vector<std::unique_ptr<Program>> progs;
    Profile prof;
};

Program* get_program(chore_handle* h, const chore_encoder_cfg& cfg, int B, int H, int W, int dtype, int n_out,
                     bool want_normx) {
    if (!h->enc_cache) h->enc_cache = new EncCache();
    EncCache* c = (EncCache*)h->enc_cache;
    for (auto& p : c->progs)
        if (p->B == B && p->H == H && p->W == W && p->dtype == dtype && p->n_out == n_out &&
            p->want_normx == want_normx && p->cfg.in_channels == cfg.in_channels &&
            p->cfg.num_stack == cfg.num_stack && p->cfg.num_hourglass == cfg.num_hourglass)
            return p.get();
    std::unique_ptr<Program> p(new Program());
    p->cfg = cfg; p->B = B; p->H = H; p->W = W; p->dtype = dtype; p->n_out = n_out; p->want_normx = want_normx;
    p->L = make_layout(cfg, dtype);
    Builder b(*p);
    b.build();
    c->progs.push_back(std::move(p));
    return c->progs.back().get();
}

size_t plan_workspace(const chore_encoder_cfg& cfg, int B, int H, int W, int dtype) {
    // the plan depends on which outputs are caller tensors: take the maximum over the variants
    size_t best = 0;
    const WLayout L = make_layout(cfg, dtype);
    const int outs[3] = {0, 1, cfg.num_stack};
    for (int oi = 0; oi < 3; ++oi)
        for (int nx = 0; nx < 2; ++nx) {
            Program p;
            p.cfg = cfg; p.B = B; p.H = H; p.W = W; p.dtype = dtype; p.n_out = outs[oi]; p.want_normx = nx != 0;
            p.L = L;
            Builder b(p);
            b.build();
            if (p.ws_bytes > best) best = p.ws_bytes;
        }
    return best;
}

}  // namespace

extern "C" {

void chore_encoder_cache_free(chore_handle* h) {
    if (h && h->enc_cache) {
        delete (EncCache*)h->enc_cache;
        h->enc_cache = nullptr;
    }
}

size_t chore_encoder_arena_bytes(const chore_encoder_cfg* cfg, int dtype) {
    if (!cfg || (dtype != CHORE_F32 && dtype != CHORE_BF16)) return 0;
    return make_layout(*cfg, dtype).total;
}

int chore_encoder_pack(chore_handle* h, const chore_encoder_cfg* cfg, const chore_weight_desc* descs, int n_descs,
                       int dtype, void* arena, chore_stream_t stream) {
    if (!h) return CHORE_EINVAL;
    if (int rc = check_cfg(h, cfg)) return rc;
    if (!descs || !arena) CHORE_FAIL(h, CHORE_EINVAL, "chore_encoder_pack: null argument");
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_encoder_pack: bad dtype");
    const WLayout L = make_layout(*cfg, dtype);
    std::unordered_map<std::string, const chore_weight_desc*> by_name;
    for (int i = 0; i < n_descs; ++i)
        if (descs[i].name) by_name[descs[i].name] = &descs[i];
    hipStream_t s = (hipStream_t)stream;
    for (const std::string& name : L.order) {
        const WEntry& e = L.e.at(name);
        auto it = by_name.find(name);
        if (it == by_name.end()) CHORE_FAIL(h, CHORE_ESTATE, "chore_encoder_pack: tensor '%s' missing", name.c_str());
        const chore_weight_desc* d = it->second;
        char* dst = (char*)arena + e.off;
        int64_t expect = 0;
        int rc = CHORE_OK;
        if (e.kind == 0) {
            expect = (int64_t)e.cout * e.cin * e.taps;
            if (d->numel == expect) rc = launch_pack_conv(h, dtype, e.taps, e.cin, e.cout, (const float*)d->ptr, dst, s);
        } else if (e.kind == 1) {
            expect = e.cout;
            if (d->numel == expect) rc = launch_copy_f32(h, (const float*)d->ptr, (float*)dst, (size_t)e.cout, s);
        } else {
            expect = (int64_t)64 * e.cin * 49;
            if (d->numel == expect) rc = launch_pack_stem(h, e.cin, (const float*)d->ptr, (float*)dst, s);
        }
        if (d->numel != expect)
            CHORE_FAIL(h, CHORE_ESTATE, "chore_encoder_pack: tensor '%s' has %lld elements, expected %lld", name.c_str(),
                       (long long)d->numel, (long long)expect);
        if (rc) return rc;
    }
    return CHORE_OK;
}

size_t chore_encoder_workspace_bytes(const chore_encoder_cfg* cfg, int B, int H, int W, int dtype) {
    if (!cfg || B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) return 0;
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) return 0;
    return plan_workspace(*cfg, B, H, W, dtype);
}

int chore_encode_fwd(chore_handle* h, const chore_encoder_cfg* cfg, const float* images, int B, int H, int W,
                     int dtype, const void* arena, void* workspace, size_t workspace_bytes, void* const* feat_out,
                     int n_stack_out, void* tmpx, void* normx, chore_stream_t stream) {
    if (!h) return CHORE_EINVAL;
    if (int rc = check_cfg(h, cfg)) return rc;
    if (!images || !arena || !workspace || !tmpx) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: null argument");
    if (B <= 0 || B > 65535 || H % 16 || W % 16 || H < 16 || W < 16)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad shape B=%d H=%d W=%d (H, W multiples of 16)", B, H, W);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad dtype");
    if (n_stack_out < 0 || n_stack_out > cfg->num_stack || (n_stack_out > 0 && !feat_out))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad n_stack_out");
    for (int i = 0; i < n_stack_out; ++i)
        if (!feat_out[i]) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: null feat_out[%d]", i);
    Program* P = get_program(h, *cfg, B, H, W, dtype, n_stack_out, normx != nullptr);
    if (workspace_bytes < P->ws_bytes)
        CHORE_FAIL(h, CHORE_ENOMEM, "chore_encode_fwd: workspace %zu < %zu bytes", workspace_bytes, P->ws_bytes);
    RunCtx r;
    r.h = h; r.dtype = dtype; r.s = (hipStream_t)stream; r.ws = (char*)workspace; r.arena = (const char*)arena;
    r.images = images; r.feats = feat_out; r.tmpx = tmpx; r.normx = normx;
    static const bool debug_sync = getenv("CHORE_DEBUG_SYNC") != nullptr;
    Profile& prof = ((EncCache*)h->enc_cache)->prof;
    if (prof.on) {  // bench/roofline aid: bracket every step with events on the caller's stream
        const size_t need = 2 * P->steps.size();
        while (prof.ev.size() < need) {
            hipEvent_t e;
            CHORE_HIP_CHECK(h, hipEventCreate(&e));
            prof.ev.push_back(e);
        }
        for (size_t i = 0; i < P->steps.size(); ++i) {
            CHORE_HIP_CHECK(h, hipEventRecord(prof.ev[2 * i], r.s));
            P->steps[i].fn(r);
            if (r.rc) return r.rc;
            CHORE_HIP_CHECK(h, hipEventRecord(prof.ev[2 * i + 1], r.s));
        }
        CHORE_HIP_CHECK(h, hipStreamSynchronize(r.s));
        for (size_t i = 0; i < P->steps.size(); ++i) {
            float ms = 0.f;
            CHORE_HIP_CHECK(h, hipEventElapsedTime(&ms, prof.ev[2 * i], prof.ev[2 * i + 1]));
            const Step& st = P->steps[i];
            prof.ms[st.klass] += ms;
            prof.flops[st.klass] += st.flops;
            prof.bytes[st.klass] += st.bytes;
            prof.launches[st.klass] += 1;
        }
        return CHORE_OK;
    }
    for (auto& st : P->steps) {
        st.fn(r);
        if (r.rc) return r.rc;
        if (debug_sync) {
            fprintf(stderr, "[chore] step %s\n", st.label.c_str());
            hipError_t e = hipStreamSynchronize(r.s);
            if (e != hipSuccess) CHORE_FAIL(h, CHORE_EHIP, "step '%s' failed: %s", st.label.c_str(), hipGetErrorString(e));
        }
    }
    return CHORE_OK;
}

int chore_profile_enable(chore_handle* h, int on) {
    if (!h) return CHORE_EINVAL;
    if (!h->enc_cache) h->enc_cache = new EncCache();
    Profile& p = ((EncCache*)h->enc_cache)->prof;
    p.on = on != 0;
    for (int k = 0; k < K_NUM; ++k) { p.ms[k] = p.flops[k] = p.bytes[k] = 0.0; p.launches[k] = 0; }
    return CHORE_OK;
}

int chore_profile_read(chore_handle* h, int max_classes, const char** names, double* ms, double* flops,
                       double* bytes, int64_t* launches) {
    if (!h || !names || !ms || !flops || !bytes || !launches) return CHORE_EINVAL;
    if (!h->enc_cache) return 0;
    const Profile& p = ((EncCache*)h->enc_cache)->prof;
    int n = 0;
    for (int k = 0; k < K_NUM && n < max_classes; ++k, ++n) {
        names[n] = kclass_names[k];
        ms[n] = p.ms[k]; flops[n] = p.flops[k]; bytes[n] = p.bytes[k]; launches[n] = p.launches[k];
    }
    return n;
}

}  // extern "C"
