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
size_t esize(int dtype) { return (dtype == CHORE_BF16 || dtype == CHORE_F16) ? 2 : 4; }
// element type the non-convolution kernels see: fp16 x 3 keeps fp32 tensors, only the convolutions split their operands
int sdt(int dtype) { return dtype == CHORE_F16X3 ? CHORE_F32 : dtype; }

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
    L.add(p + "conv1.weight", (size_t)cfg.in_channels * 49 * 64 * 4 + stem_x3_bytes(), 2, 49, cfg.in_channels, 64);   // fp32 pack, then the fp16 x 3 fragments
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
            // merged l / bl / al (see Builder::build): derived at pack time, not a tensor of the state dict
            L.add(p + "ml" + s + ".wf32", (size_t)256 * 256 * 4, 3, 1, 256, 256);
            L.add(p + "ml" + s + ".weight", packed_conv_bytes(dtype, 1, 256, 256), 4, 1, 256, 256);
            L.add(p + "ml" + s + ".bias", 256 * 4, 5, 0, 0, 256);
        }
    }
    L.total = align_up(L.total, 256);
    return L;
}

// previous' = previous + bl(ll) + al(l(ll)) (HGFilters.py:176-183) is ONE 1x1 convolution of ll when out_i = l(ll) itself is
// not asked for: W = W_bl + W_al W_l, b = b_bl + b_al + W_al b_l.  One thread per element of W, fp64 accumulation.
__global__ void compose_1x1_kernel(const float* __restrict__ wl, const float* __restrict__ bl_l, const float* __restrict__ wbl,
                                   const float* __restrict__ bbl, const float* __restrict__ wal, const float* __restrict__ bal,
                                   int C, float* __restrict__ wout, float* __restrict__ bout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // output channel o = i / C, input channel c = i % C
    if (i >= C * C) return;
    const int o = i / C, c = i % C;
    double acc = (double)wbl[(size_t)o * C + c];
    for (int k = 0; k < C; ++k) acc += (double)wal[(size_t)o * C + k] * (double)wl[(size_t)k * C + c];
    wout[i] = (float)acc;
    if (c == 0) {
        double b = (double)bbl[o] + (double)bal[o];
        for (int k = 0; k < C; ++k) b += (double)wal[(size_t)o * C + k] * (double)bl_l[k];
        bout[o] = (float)b;
    }
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
// program = list of steps (closures); buffers are offsets into per-stream pools of the workspace
// (or caller tensors).  Independent branches of the hourglass run on auxiliary streams (fork/join
// with events), each with its own pool so concurrent branches never alias.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_STREAMS = 5;   // caller's stream + one per hourglass level

struct Buf {
    size_t off = 0;       // offset inside pool `pool`, or
    int pool = 0;
    int ext = -1;         // index of an external (caller) tensor: 0..n_out-1 feats, 100 tmpx, 101 normx
    int H = 0, W = 0, C = 0;
    size_t bytes = 0;
    // GroupNorm statistics ([B][32] GroupStat accumulators in the stats arena) of the buffer's content
    bool st_valid = false;
    size_t st_off = 0;
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
    hipStream_t s;                 // stream of the step being issued
    char* pool_base[MAX_STREAMS];
    char* stats;                   // stats arena
    const char* arena;
    const float* images;
    void* const* feats;
    void* tmpx;
    void* normx;
    int rc = CHORE_OK;
};

// one class per kernel instantiation, named like the kernel in a rocprofv3 trace so that bench.py's live
// hipEvent numbers can be checked against profiles/*_kernel_stats.csv line by line
enum KClass { K_STEM = 0, K_GN_STATS, K_GN_APPLY, K_POOL, K_UPADD, K_CONV_FIRST, K_PC_FIRST = K_CONV_FIRST + 11, K_RW = K_PC_FIRST + 7, K_MW_FIRST = K_RW + 1, K_NUM = K_MW_FIRST + 8 };
const char* const kclass_names[K_NUM] = {"stem_kernel", "gn_stats_kernel", "gn_apply_relu_kernel", "map_stats_kernel<T,C,PoolOp>",
                                         "map_stats_kernel<T,C,UpAddOp>", "conv_lds_kernel<T,9,128,3>", "conv_lds_kernel<T,9,64,3>",
                                         "conv_lds_kernel<T,9,32,3>", "conv_lds_kernel<T,9,64,9>",
                                         "conv_lds_kernel<T,9,32,9>", "conv_lds_kernel<T,1,128,1>",
                                         "conv_lds_kernel<T,1,64,1>", "conv_lds_kernel<T,1,32,1>", "conv_small_kernel<T,64,ROWS>",
                                         "conv_small_kernel<T,128,ROWS>", "conv_small_kernel<T,256,ROWS>",
                                         "conv_pc_kernel<T,9,8,128,1,3>", "conv_pc_kernel<T,9,8,64,3,2>",
                                         "conv_pc_kernel<T,9,8,32,3,2>", "conv_pc_kernel<T,9,4,64,3,2>", "conv_pc_kernel<T,9,4,32,9,2>",
                                         "conv_pc_kernel<T,1,8,128,1,2>", "conv_pc_kernel<T,1,8,64,1,2>", "conv_rw_kernel<KC,WN,RES,SC>",
                                         "conv_mw_kernel<T,9,8,128,1,3>", "conv_mw_kernel<T,9,8,64,3,2>", "conv_mw_kernel<T,9,8,32,3,2>",
                                         "conv_mw_kernel<T,9,4,64,3,2>", "conv_mw_kernel<T,9,4,32,9,2>", "conv_mw_kernel<T,9,2,128,1,3>",
                                         "conv_mw_kernel<T,9,4,128,1,3>", "conv_mw_kernel<T,9,2,64,1,3>"};
inline int conv_class(const ConvPlan& p, int taps) {
    if (p.tps == 0) return K_CONV_FIRST + 8 + (p.small_cin == 64 ? 0 : (p.small_cin == 128 ? 1 : 2));
    const int ni = p.nt == 128 ? 0 : (p.nt == 64 ? 1 : 2);
    if (taps == 1) return K_CONV_FIRST + 5 + ni;
    if (p.tps == 9) return K_CONV_FIRST + 3 + (ni - 1);
    return K_CONV_FIRST + ni;
}
// class of the kernel launch_conv will pick for a layer (the specialised-wave kernel where it covers the layer)
inline int conv_class_of(int dtype, int taps, int B, int H, int W, int Cin, int Cout) {
    if (conv_mw_on(dtype, taps)) {   // as launch_conv: conv_mw_kernel's own tiling first (it also takes the small maps, CHORE_CONV_MW_FILL)
        const PcPlan mp = conv_mw_plan(dtype, taps, B, H, W, Cin, Cout, CONV_MW_FILL_INFER);
        if (mp.th && conv_mw_has(mp) && (conv_mw_fill(CONV_MW_FILL_INFER) < 256 || !conv_small_eligible(dtype, taps, H, W, Cin, Cout))) {
            if (mp.th == 8) return K_MW_FIRST + (mp.nt == 128 ? 0 : (mp.nt == 64 ? 1 : 2));
            if (mp.th == 4) return K_MW_FIRST + (mp.nt == 128 ? 6 : (mp.nt == 64 ? 3 : 4));
            return K_MW_FIRST + (mp.nt == 128 ? 5 : 7);
        }
    }
    if (conv_rw_covers(dtype, taps, Cin, Cout, false) && conv_use_pc() && !conv_small_eligible(dtype, taps, H, W, Cin, Cout)) return K_RW;
    if ((dtype == CHORE_F16 || (dtype == CHORE_F16X3 && conv_use_pc())) && !conv_small_eligible(dtype, taps, H, W, Cin, Cout)) {
        const PcPlan pp = conv_pc_plan(dtype, taps, B, H, W, Cin, Cout);
        if (pp.th) {
            if (taps == 1) return K_PC_FIRST + (pp.nt == 128 ? 5 : 6);
            if (pp.th == 8) return K_PC_FIRST + (pp.nt == 128 ? 0 : (pp.nt == 64 ? 1 : 2));
            return K_PC_FIRST + (pp.nt == 64 ? 3 : 4);
        }
    }
    return conv_class(conv_plan(dtype, taps, B, H, W, Cin, Cout), taps);
}

enum StepKind { S_KERNEL = 0, S_RECORD, S_WAIT, S_MEMSET };

struct Step {
    int kind = S_KERNEL;
    int stream = 0;        // stream index the step is issued on
    int event = -1;        // S_RECORD / S_WAIT
    std::string label;
    std::function<void(RunCtx&)> fn;
    int klass = 0;
    double flops = 0.0;    // algorithmic FLOPs of the launch (convolutions: 2*taps*Cin*Cout*B*H*W)
    double bytes = 0.0;    // algorithmic HBM bytes of the launch (compulsory reads + writes)
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
    size_t pool_off[MAX_STREAMS] = {0};   // base offset of each stream's pool inside the workspace
    size_t stats_off = 0, stats_bytes = 0;
    int n_events = 0, n_streams = 1;
    size_t ws_bytes = 0;
    std::vector<hipEvent_t> events;       // created on first run: [group][n_events]
};

struct Builder {
    Program& P;
    Pool pools[MAX_STREAMS];
    size_t stat_top = 0;
    int B, dtype;
    int cur = 0;           // stream the next steps are issued on
    bool concurrent;
    bool main_first = false;
    std::string cur_label = "?";
    int cur_class = 0;
    double cur_flops = 0.0, cur_bytes = 0.0;
    double es() const { return (double)esize(dtype); }

    bool no_merge = false;   // CHORE_ENC_NO_MERGE: keep l / bl / al separate in eval too (A/B and bit-comparison with training mode)
    explicit Builder(Program& p) : P(p), B(p.B), dtype(p.dtype) {
        concurrent = getenv("CHORE_ENC_SERIAL") == nullptr;
        // Issue order after a fork: the LOWER branch (the longer chain, on the forking stream) before the upper branch (child
        // stream).  Replayed as a hipGraph the runtime keeps a node's first-recorded successor on the node's queue and moves the
        // others to new queues (~10 us per queue change): 5.43 against 5.53 ms per step.  CHORE_ENC_CHILD_FIRST=1: the old order.
        main_first = getenv("CHORE_ENC_CHILD_FIRST") == nullptr;
        no_merge = getenv("CHORE_ENC_NO_MERGE") != nullptr;
    }

    Buf alloc(int H, int W, int C) {
        Buf b;
        b.H = H; b.W = W; b.C = C;
        b.pool = cur;
        b.bytes = (size_t)B * H * W * C * esize(dtype);
        b.off = pools[cur].alloc(b.bytes);
        return b;
    }
    Buf external(int id, int H, int W, int C) {
        Buf b;
        b.ext = id; b.H = H; b.W = W; b.C = C;
        return b;
    }
    void release(const Buf& b) {
        if (b.ext < 0) pools[b.pool].release(b.off, b.bytes);
    }
    // fresh, never-reused statistics accumulators for the tensor in `b` (zeroed once per encode)
    void new_stats(Buf& b) {
        b.st_valid = true;
        b.st_off = stat_top;
        stat_top += align_up(act_stats_bytes(B), 256);
    }

    static void* ptr(RunCtx& r, const Buf& b) {
        if (b.ext < 0) return r.pool_base[b.pool] + b.off;
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
    void push(std::function<void(RunCtx&)> fn) {
        Step st;
        st.kind = S_KERNEL; st.stream = cur; st.label = cur_label; st.fn = std::move(fn);
        st.klass = cur_class; st.flops = cur_flops; st.bytes = cur_bytes;
        P.steps.push_back(std::move(st));
    }
    // child stream starts after everything issued so far on the current stream
    void fork(int child) {
        if (!concurrent) return;
        const int ev = P.n_events++;
        Step a; a.kind = S_RECORD; a.stream = cur; a.event = ev; a.label = "fork";
        Step b; b.kind = S_WAIT; b.stream = child; b.event = ev; b.label = "fork";
        P.steps.push_back(a);
        P.steps.push_back(b);
        if (child + 1 > P.n_streams) P.n_streams = child + 1;
    }
    // current stream waits for everything issued so far on the child stream
    void join(int child) {
        if (!concurrent) return;
        const int ev = P.n_events++;
        Step a; a.kind = S_RECORD; a.stream = child; a.event = ev; a.label = "join";
        Step b; b.kind = S_WAIT; b.stream = cur; b.event = ev; b.label = "join";
        P.steps.push_back(a);
        P.steps.push_back(b);
    }

    // ---- op emitters ----
    // make sure `x` carries statistics (tensors written by pool / upadd / stem get a stats pass)
    void ensure_stats(Buf& x) {
        if (x.st_valid) return;
        new_stats(x);
        const Buf xb = x;
        const int HW = x.H * x.W, Bn = B;
        cur_label = "gn_stats";
        cur_class = K_GN_STATS; cur_flops = 0.0; cur_bytes = (double)B * HW * x.C * es();
        push([=](RunCtx& r) {
            if (r.rc) return;
            r.rc = launch_gn_stats(r.h, sdt(r.dtype), view(r, xb), Bn, HW, (GroupStat*)(r.stats + xb.st_off), r.s);
        });
    }

    struct ConvSpec {
        Buf in; int in_C = 0;
        std::string gn;                        // GroupNorm (+ReLU) fused into the operand load ("" = none)
        std::string wname; bool bias = false;
        Buf out; int out_co = 0;
        bool has_raw = false; Buf raw; int raw_co = 0;
        bool has_res = false; Buf res; int res_co = 0;
        bool has_res2 = false; Buf res2; int res2_co = 0;
        int taps = 9, cout = 0;
        bool stat_raw = false, stat_out = false;   // accumulate GroupNorm statistics of raw / out
    };
    void conv(const ConvSpec& c) {
        const WEntry we = w(c.wname + ".weight");
        WEntry be{}, ge{}, bte{};
        if (c.bias) be = w(c.wname + ".bias");
        const bool use_gn = !c.gn.empty();
        if (use_gn) { ge = w(c.gn + ".weight"); bte = w(c.gn + ".bias"); }
        if (use_gn && (!c.in.st_valid || c.in_C != c.in.C)) abort();
        const ConvSpec cs = c;
        cur_label = "conv " + c.wname + " taps=" + std::to_string(c.taps) + " cin=" + std::to_string(c.in_C) +
                    " cout=" + std::to_string(c.cout) + " HxW=" + std::to_string(c.in.H) + "x" + std::to_string(c.in.W);
        {
            const double px = (double)B * c.in.H * c.in.W;
            cur_class = conv_class_of(dtype, c.taps, B, c.in.H, c.in.W, c.in_C, c.cout);
            cur_flops = 2.0 * c.taps * c.in_C * c.cout * px;
            cur_bytes = px * es() * (c.in_C + c.cout * (1 + (c.has_raw ? 1 : 0) + (c.has_res ? 1 : 0) + (c.has_res2 ? 1 : 0)));
        }
        const int Bn = B;
        push([=](RunCtx& r) {
            if (r.rc) return;
            ConvArgs a{};
            a.fill = CONV_MW_FILL_INFER;
            a.in = view(r, cs.in, 0, cs.in_C);
            if (use_gn) {
                a.in_st = (const GroupStat*)(r.stats + cs.in.st_off);
                a.gamma = (const float*)(r.arena + ge.off);
                a.beta = (const float*)(r.arena + bte.off);
            }
            a.wpk = r.arena + we.off;
            a.bias = cs.bias ? (const float*)(r.arena + be.off) : nullptr;
            a.out = view(r, cs.out, cs.out_co, cs.cout);
            if (cs.has_raw) a.raw = view(r, cs.raw, cs.raw_co, cs.cout);
            if (cs.has_res) a.res = view(r, cs.res, cs.res_co, cs.cout);
            if (cs.has_res2) a.res2 = view(r, cs.res2, cs.res2_co, cs.cout);
            a.B = Bn; a.H = cs.in.H; a.W = cs.in.W; a.Cout = cs.cout;
            if (cs.stat_raw) {
                a.st_raw = (GroupStat*)(r.stats + cs.raw.st_off); a.st_raw_C = cs.raw.C; a.st_raw_co = cs.raw_co;
            }
            if (cs.stat_out) {
                a.st_out = (GroupStat*)(r.stats + cs.out.st_off); a.st_out_C = cs.out.C; a.st_out_co = cs.out_co;
            }
            r.rc = launch_conv(r.h, r.dtype, cs.taps, a, r.s);
        });
    }

    // ConvBlock (net_util.py:374-396): y = cat(o1,o2,o3) + residual; x must carry statistics
    Buf conv_block(Buf& x, const std::string& n, int cin, int cout) {
        const int H = x.H, W = x.W;
        ensure_stats(x);
        Buf out = alloc(H, W, cout);
        Buf o1 = alloc(H, W, cout / 2), o2 = alloc(H, W, cout / 4);
        new_stats(out);
        new_stats(o1);
        new_stats(o2);
        Buf res = x;
        if (cin != cout) {
            ConvSpec d;  // residual = conv1x1(relu(gn4(x)))
            d.in = x; d.in_C = cin; d.gn = n + ".bn4"; d.wname = n + ".downsample.2";
            d.out = out; d.taps = 1; d.cout = cout;
            conv(d);
            res = out;
        }
        ConvSpec c1;
        c1.in = x; c1.in_C = cin; c1.gn = n + ".bn1"; c1.wname = n + ".conv1";
        c1.out = out; c1.out_co = 0; c1.has_raw = true; c1.raw = o1; c1.has_res = true; c1.res = res; c1.res_co = 0;
        c1.cout = cout / 2; c1.stat_raw = true; c1.stat_out = true;
        conv(c1);
        ConvSpec c2;
        c2.in = o1; c2.in_C = cout / 2; c2.gn = n + ".bn2"; c2.wname = n + ".conv2";
        c2.out = out; c2.out_co = cout / 2; c2.has_raw = true; c2.raw = o2; c2.has_res = true; c2.res = res;
        c2.res_co = cout / 2; c2.cout = cout / 4; c2.stat_raw = true; c2.stat_out = true;
        conv(c2);
        ConvSpec c3;
        c3.in = o2; c3.in_C = cout / 4; c3.gn = n + ".bn3"; c3.wname = n + ".conv3";
        c3.out = out; c3.out_co = 3 * cout / 4; c3.has_res = true; c3.res = res; c3.res_co = 3 * cout / 4;
        c3.cout = cout / 4; c3.stat_out = true;
        conv(c3);
        release(o1);
        release(o2);
        return out;
    }

    Buf pool2(const Buf& x, const Buf* out_opt = nullptr) {
        Buf y = out_opt ? *out_opt : alloc(x.H / 2, x.W / 2, x.C);
        new_stats(y);   // the pooling kernel also accumulates the statistics of its output
        cur_label = "avgpool2";
        cur_class = K_POOL; cur_flops = 0.0; cur_bytes = (double)B * x.H * x.W * x.C * es() * 1.25;
        const int Bn = B;
        push([=](RunCtx& r) {
            if (r.rc) return;
            r.rc = launch_avgpool2(r.h, sdt(r.dtype), view(r, x), view(r, y), Bn, x.H, x.W,
                                   (GroupStat*)(r.stats + y.st_off), r.s);
        });
        return y;
    }
    void upadd(Buf& a, const Buf& low) {  // a += bicubic_up2(low)
        cur_label = "upadd";
        cur_class = K_UPADD; cur_flops = 0.0; cur_bytes = (double)B * low.H * low.W * low.C * es() * 9.0;
        new_stats(a);   // modified in place: fresh accumulators, filled by the same kernel
        const Buf ab = a;
        const int Bn = B;
        push([=](RunCtx& r) {
            if (r.rc) return;
            r.rc = launch_upadd(r.h, sdt(r.dtype), view(r, ab), view(r, low), view(r, ab), Bn, low.H, low.W,
                                (GroupStat*)(r.stats + ab.st_off), r.s);
        });
    }

    // HourGlass._forward (HGFilters.py:26-50).  The upper branch (b1 at full resolution) is independent
    // of the whole lower branch: it runs on its own stream.
    Buf hourglass(Buf& x, const std::string& n, int level) {
        const std::string l = std::to_string(level);
        const int child = P.cfg.num_hourglass - level + 1;   // 1 for the outermost level
        ensure_stats(x);   // on the current stream, before the fork: both branches read them
        const int parent = cur;
        fork(child);
        Buf up1;
        if (!(concurrent && main_first)) {
            if (concurrent) cur = child;
            up1 = conv_block(x, n + ".b1_" + l, 256, 256);
            cur = parent;
        }
        Buf pooled = pool2(x);
        Buf low1 = conv_block(pooled, n + ".b2_" + l, 256, 256);
        release(pooled);
        Buf low2 = (level > 1) ? hourglass(low1, n, level - 1) : conv_block(low1, n + ".b2_plus_" + l, 256, 256);
        release(low1);
        Buf low3 = conv_block(low2, n + ".b3_" + l, 256, 256);
        release(low2);
        if (concurrent && main_first) {
            cur = child;
            up1 = conv_block(x, n + ".b1_" + l, 256, 256);
            cur = parent;
        }
        join(child);
        upadd(up1, low3);
        release(low3);
        return up1;
    }

    void build() {
        const chore_encoder_cfg& cfg = P.cfg;
        const std::string p = "image_filter.";
        const int H2 = P.H / 2, W2 = P.W / 2, H4 = P.H / 4, W4 = P.W / 4;
        const int Bn = B;
        // zero the statistics accumulators of this pass (one memset; the arena is never reused inside a pass)
        {
            Step st;
            st.kind = S_MEMSET; st.stream = 0; st.label = "zero stats";
            P.steps.push_back(st);
        }
        // stem: conv7x7/2 + GN + ReLU -> tmpx (HGFilters.py:149-150)
        Buf c1 = alloc(H2, W2, 64);
        {
            const WEntry we = w(p + "conv1.weight"), be = w(p + "conv1.bias");
            const int Cin = cfg.in_channels, H = P.H, W = P.W;
            cur_label = "stem";
            cur_class = K_STEM; cur_flops = 2.0 * 49 * Cin * 64 * (double)B * (H / 2) * (W / 2);
            cur_bytes = (double)B * H * W * Cin * 4 + (double)B * (H / 2) * (W / 2) * 64 * es();
            push([=](RunCtx& r) {
                if (r.rc) return;
                if (r.dtype == CHORE_F16X3 && stem_x3_on(Cin))
                    r.rc = launch_stem_x3(r.h, r.images, Bn, Cin, H, W, r.arena + we.off + (size_t)Cin * 49 * 64 * 4,
                                          (const float*)(r.arena + be.off), (float*)ptr(r, c1), r.s);
                else
                    r.rc = launch_stem(r.h, sdt(r.dtype), r.images, Bn, Cin, H, W, (const float*)(r.arena + we.off),
                                       (const float*)(r.arena + be.off), ptr(r, c1), r.s);
            });
        }
        Buf tmpx = external(100, H2, W2, 64);
        {
            ensure_stats(c1);
            const WEntry ge = w(p + "bn1.weight"), bte = w(p + "bn1.bias");
            const Buf c1b = c1;
            // tmpx's own statistics (the next ConvBlock's GroupNorm needs them) ride in this launch (round 5: was a gn_stats pass)
            static const bool fold = getenv("CHORE_ENC_NO_TMPX_STATS_FOLD") == nullptr;
            if (fold) new_stats(tmpx);
            const Buf tb = tmpx;
            cur_label = "gn_apply_relu bn1";
            cur_class = K_GN_APPLY; cur_flops = 0.0; cur_bytes = 2.0 * B * H2 * W2 * 64 * es();
            push([=](RunCtx& r) {
                if (r.rc) return;
                r.rc = launch_gn_apply_relu(r.h, sdt(r.dtype), view(r, c1b), (const GroupStat*)(r.stats + c1b.st_off),
                                            (const float*)(r.arena + ge.off), (const float*)(r.arena + bte.off),
                                            view(r, tb), Bn, H2 * W2, r.s, fold ? (GroupStat*)(r.stats + tb.st_off) : nullptr);
            });
        }
        release(c1);
        Buf b2 = conv_block(tmpx, p + "conv2", 64, 128);
        Buf normx = P.want_normx ? external(101, H4, W4, 128) : alloc(H4, W4, 128);
        normx = pool2(b2, &normx);
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
            new_stats(t2);
            ConvSpec cl;
            cl.in = t1; cl.in_C = 256; cl.wname = p + "conv_last" + s; cl.bias = true; cl.out = t2; cl.taps = 1;
            cl.cout = 256; cl.stat_out = true;
            conv(cl);
            release(t1);
            const int oi = i - (cfg.num_stack - P.n_out);
            if (oi < 0 && i < cfg.num_stack - 1 && !no_merge) {
                // out_i = l(ll) is not an output of this call (eval keeps the last stack only, model/chore.py:95-96) and
                // its only reader is al: l, bl and al collapse into the single 1x1 convolution packed as "ml<i>"
                Buf nprev = alloc(H4, W4, 256);
                new_stats(nprev);
                ConvSpec ml;
                ml.in = t2; ml.in_C = 256; ml.gn = p + "bn_end" + s; ml.wname = p + "ml" + s; ml.bias = true;
                ml.out = nprev; ml.has_res = true; ml.res = previous; ml.taps = 1; ml.cout = 256; ml.stat_out = true;
                conv(ml);
                release(previous);
                previous = nprev;
                release(t2);
                continue;
            }
            Buf out_i = (oi >= 0) ? external(oi, H4, W4, 256) : alloc(H4, W4, 256);
            ConvSpec l;
            l.in = t2; l.in_C = 256; l.gn = p + "bn_end" + s; l.wname = p + "l" + s; l.bias = true; l.out = out_i;
            l.taps = 1; l.cout = cfg.hourglass_dim;
            conv(l);
            if (i < cfg.num_stack - 1) {
                Buf nprev = alloc(H4, W4, 256);
                new_stats(nprev);
                ConvSpec bl;
                bl.in = t2; bl.in_C = 256; bl.gn = p + "bn_end" + s; bl.wname = p + "bl" + s; bl.bias = true;
                bl.out = nprev; bl.has_res = true; bl.res = previous; bl.taps = 1; bl.cout = 256;
                conv(bl);
                ConvSpec al;
                al.in = out_i; al.in_C = 256; al.wname = p + "al" + s; al.bias = true; al.out = nprev;
                al.has_res = true; al.res = nprev; al.taps = 1; al.cout = 256; al.stat_out = true;
                conv(al);
                release(previous);
                previous = nprev;
            }
            release(t2);
            release(out_i);
        }
        release(previous);
        // workspace layout: [pool 0][pool 1]...[stats arena]
        size_t off = 0;
        for (int i = 0; i < MAX_STREAMS; ++i) {
            P.pool_off[i] = off;
            off += align_up(pools[i].top, 256);
        }
        P.stats_off = off;
        P.stats_bytes = align_up(stat_top, 256);
        P.ws_bytes = off + P.stats_bytes;
    }
};

constexpr int MAX_GROUPS = 8;
struct EncCache {
    std::vector<std::unique_ptr<Program>> progs;
    Profile prof;
    hipStream_t aux[MAX_GROUPS][MAX_STREAMS] = {};   // [0][0] unused (caller's stream)
    hipEvent_t fork_ev = nullptr, join_ev[MAX_GROUPS] = {};
};

// CHORE_ENC_GROUPS=G encodes the batch as G independent groups of B / G images, each a chain of launches of its own on its
// own streams.  The idea: a launch of specialised-wave workgroups (one per CU) lasts as long as ONE workgroup lives, whether
// it has 256 workgroups or 64, so the chains of half batches could overlap.  Measured (B = 4, fp16x3): G = 1 5.6 ms, G = 2
// 6.9 ms, G = 4 10.7 ms per step with only 2.6 ms of host time at G = 2 -- the device does not run the chains side by side
// (six and more streams share the hardware queues).  Off by default; kept as a switch.  Round 5, with hipGraph replays: two
// recordings of a HALF batch side by side take 2 x 2.72 ms for the four images, one recording of the whole batch 5.31 ms (same
// box): splitting the batch does not pay under replay either.
int enc_groups(int B) {
    static const int want = getenv("CHORE_ENC_GROUPS") ? atoi(getenv("CHORE_ENC_GROUPS")) : 1;
    int g = want < 1 ? 1 : (want > MAX_GROUPS ? MAX_GROUPS : want);
    while (g > 1 && B % g) --g;
    return g;
}

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
        EncCache* c = (EncCache*)h->enc_cache;
        for (int g = 0; g < MAX_GROUPS; ++g)
            for (int i = 0; i < MAX_STREAMS; ++i)
                if (c->aux[g][i]) (void)hipStreamDestroy(c->aux[g][i]);
        if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
        for (hipEvent_t e : c->join_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto& p : c->progs)
            for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->prof.ev) (void)hipEventDestroy(e);
        delete c;
        h->enc_cache = nullptr;
    }
}

size_t chore_encoder_arena_bytes(const chore_encoder_cfg* cfg, int dtype) {
    if (!cfg || (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16)) return 0;
    return make_layout(*cfg, dtype).total;
}

int chore_encoder_pack(chore_handle* h, const chore_encoder_cfg* cfg, const chore_weight_desc* descs, int n_descs,
                       int dtype, void* arena, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (int rc = check_cfg(h, cfg)) return rc;
    if (!descs || !arena) CHORE_FAIL(h, CHORE_EINVAL, "chore_encoder_pack: null argument");
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16) CHORE_FAIL(h, CHORE_EINVAL, "chore_encoder_pack: bad dtype");
    const WLayout L = make_layout(*cfg, dtype);
    std::unordered_map<std::string, const chore_weight_desc*> by_name;
    for (int i = 0; i < n_descs; ++i)
        if (descs[i].name) by_name[descs[i].name] = &descs[i];
    hipStream_t s = (hipStream_t)stream;
    for (const std::string& name : L.order) {
        const WEntry& e = L.e.at(name);
        if (e.kind >= 3) {
            if (e.kind != 3) continue;                         // .weight / .bias of a merged conv are written with its .wf32
            const std::string base = name.substr(0, name.size() - 5);                     // "...ml<i>"
            const std::string idx = base.substr(base.rfind("ml") + 2), pre = base.substr(0, base.rfind("ml"));
            const float* src[6];
            const char* leaf[6] = {"l%s.weight", "l%s.bias", "bl%s.weight", "bl%s.bias", "al%s.weight", "al%s.bias"};
            for (int k = 0; k < 6; ++k) {
                char buf[64];
                snprintf(buf, sizeof(buf), leaf[k], idx.c_str());
                auto f = by_name.find(pre + buf);
                const int64_t want = (k & 1) ? 256 : 256 * 256;
                if (f == by_name.end() || f->second->numel != want)
                    CHORE_FAIL(h, CHORE_ESTATE, "chore_encoder_pack: tensor '%s%s' missing or of the wrong size", pre.c_str(), buf);
                src[k] = (const float*)f->second->ptr;
            }
            float* wf = (float*)((char*)arena + e.off);
            float* bf = (float*)((char*)arena + L.e.at(base + ".bias").off);
            hipLaunchKernelGGL(compose_1x1_kernel, dim3(256), dim3(256), 0, s, src[0], src[1], src[2], src[3], src[4], src[5], 256,
                               wf, bf);
            CHORE_LAUNCH_CHECK(h, s);
            if (int rc = launch_pack_conv(h, dtype, 1, 256, 256, wf, (char*)arena + L.e.at(base + ".weight").off, s)) return rc;
            continue;
        }
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
            if (d->numel == expect && !rc && stem_x3_on(e.cin))
                rc = launch_pack_stem_x3(h, e.cin, (const float*)d->ptr, dst + (size_t)e.cin * 49 * 64 * 4, s);
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
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16) return 0;
    const int G = enc_groups(B);
    return (size_t)G * align_up(plan_workspace(*cfg, B / G, H, W, dtype), 256);
}

int chore_encode_fwd(chore_handle* h, const chore_encoder_cfg* cfg, const float* images, int B, int H, int W,
                     int dtype, const void* arena, void* workspace, size_t workspace_bytes, void* const* feat_out,
                     int n_stack_out, void* tmpx, void* normx, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (int rc = check_cfg(h, cfg)) return rc;
    if (!images || !arena || !workspace || !tmpx) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: null argument");
    if (B <= 0 || B > 65535 || H % 16 || W % 16 || H < 16 || W < 16)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad shape B=%d H=%d W=%d (H, W multiples of 16)", B, H, W);
    if (dtype != CHORE_F32 && dtype != CHORE_BF16 && dtype != CHORE_F16X3 && dtype != CHORE_F16) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad dtype");
    if (n_stack_out < 0 || n_stack_out > cfg->num_stack || (n_stack_out > 0 && !feat_out))
        CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: bad n_stack_out");
    for (int i = 0; i < n_stack_out; ++i)
        if (!feat_out[i]) CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: null feat_out[%d]", i);
    const int G = enc_groups(B), Bg = B / G;
    if (G > 1) {   // CHORE_ENC_GROUPS (an experiment switch, off by default): its fork / join of the groups' chains crashes this
                   // runtime's hipStreamEndCapture (round 5) -- refuse a recording instead
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        CHORE_HIP_CHECK(h, hipStreamIsCapturing((hipStream_t)stream, &cs));
        if (cs != hipStreamCaptureStatusNone)
            CHORE_FAIL(h, CHORE_EINVAL, "chore_encode_fwd: CHORE_ENC_GROUPS=%d cannot be recorded into a hipGraph (eager launches only)", G);
    }
    Program* P = get_program(h, *cfg, Bg, H, W, dtype, n_stack_out, normx != nullptr);
    const size_t ws_group = align_up(P->ws_bytes, 256);
    if (workspace_bytes < (size_t)G * ws_group)
        CHORE_FAIL(h, CHORE_ENOMEM, "chore_encode_fwd: workspace %zu < %zu bytes", workspace_bytes, (size_t)G * ws_group);
    EncCache* cache = (EncCache*)h->enc_cache;
    static const bool debug_sync = getenv("CHORE_DEBUG_SYNC") != nullptr;
    Profile& prof = cache->prof;
    const bool serial = prof.on || debug_sync;   // attribution / debugging: everything on the caller's stream
    hipStream_t streams[MAX_GROUPS][MAX_STREAMS];
    for (int g = 0; g < G; ++g)
        for (int i = 0; i < P->n_streams; ++i) {   // auxiliary streams / events are created once, not per call
            if (g == 0 && i == 0) { streams[0][0] = (hipStream_t)stream; continue; }
            if (!cache->aux[g][i]) CHORE_HIP_CHECK(h, hipStreamCreateWithFlags(&cache->aux[g][i], hipStreamNonBlocking));
            streams[g][i] = cache->aux[g][i];
        }
    while ((int)P->events.size() < G * P->n_events) {
        hipEvent_t e;
        CHORE_HIP_CHECK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        P->events.push_back(e);
    }
    if (G > 1 && !serial) {   // the other groups' chains start where the caller's stream is now
        if (!cache->fork_ev) CHORE_HIP_CHECK(h, hipEventCreateWithFlags(&cache->fork_ev, hipEventDisableTiming));
        CHORE_HIP_CHECK(h, hipEventRecord(cache->fork_ev, (hipStream_t)stream));
        for (int g = 1; g < G; ++g) CHORE_HIP_CHECK(h, hipStreamWaitEvent(streams[g][0], cache->fork_ev, 0));
    }
    if (prof.on) {
        const size_t need = 2 * P->steps.size() * G;
        while (prof.ev.size() < need) {
            hipEvent_t e;
            CHORE_HIP_CHECK(h, hipEventCreate(&e));
            prof.ev.push_back(e);
        }
    }
    // element sizes of the caller's tensors (feature maps and tmpx / normx in the activation type)
    const size_t es = esize(dtype);
    const size_t img_stride = (size_t)cfg->in_channels * H * W;                          // floats per image
    const size_t feat_stride = (size_t)(H / 4) * (W / 4) * cfg->hourglass_dim * es;      // bytes per image
    const size_t tmpx_stride = (size_t)(H / 2) * (W / 2) * 64 * es, normx_stride = (size_t)(H / 4) * (W / 4) * 128 * es;
    for (int g = 0; g < G; ++g) {
        void* feats_g[16];
        for (int i = 0; i < n_stack_out; ++i) feats_g[i] = (char*)feat_out[i] + (size_t)g * Bg * feat_stride;
        RunCtx r;
        r.h = h; r.dtype = dtype; r.s = streams[g][0]; r.arena = (const char*)arena;
        char* ws = (char*)workspace + (size_t)g * ws_group;
        for (int i = 0; i < MAX_STREAMS; ++i) r.pool_base[i] = ws + P->pool_off[i];
        r.stats = ws + P->stats_off;
        r.images = images + (size_t)g * Bg * img_stride;
        r.feats = feats_g;
        r.tmpx = (char*)tmpx + (size_t)g * Bg * tmpx_stride;
        r.normx = normx ? (char*)normx + (size_t)g * Bg * normx_stride : nullptr;
        hipEvent_t* ev = P->events.data() + (size_t)g * P->n_events;
        hipEvent_t* pev = prof.on ? prof.ev.data() + (size_t)g * 2 * P->steps.size() : nullptr;
        for (size_t i = 0; i < P->steps.size(); ++i) {
            Step& st = P->steps[i];
            hipStream_t ss = serial ? (hipStream_t)stream : streams[g][st.stream];
            if (st.kind == S_RECORD) {
                if (!serial) CHORE_HIP_CHECK(h, hipEventRecord(ev[st.event], ss));
                continue;
            }
            if (st.kind == S_WAIT) {
                if (!serial) CHORE_HIP_CHECK(h, hipStreamWaitEvent(ss, ev[st.event], 0));
                continue;
            }
            if (st.kind == S_MEMSET) {
                CHORE_HIP_CHECK(h, hipMemsetAsync(r.stats, 0, P->stats_bytes, ss));
                continue;
            }
            r.s = ss;
            if (prof.on) CHORE_HIP_CHECK(h, hipEventRecord(pev[2 * i], ss));
            st.fn(r);
            if (r.rc) return r.rc;
            if (prof.on) CHORE_HIP_CHECK(h, hipEventRecord(pev[2 * i + 1], ss));
            if (debug_sync) {
                fprintf(stderr, "[chore] group %d step %s\n", g, st.label.c_str());
                hipError_t e = hipStreamSynchronize(ss);
                if (e != hipSuccess) CHORE_FAIL(h, CHORE_EHIP, "step '%s' failed: %s", st.label.c_str(), hipGetErrorString(e));
            }
        }
        if (g > 0 && !serial) {   // the caller's stream continues after every group
            if (!cache->join_ev[g]) CHORE_HIP_CHECK(h, hipEventCreateWithFlags(&cache->join_ev[g], hipEventDisableTiming));
            CHORE_HIP_CHECK(h, hipEventRecord(cache->join_ev[g], streams[g][0]));
            CHORE_HIP_CHECK(h, hipStreamWaitEvent((hipStream_t)stream, cache->join_ev[g], 0));
        }
    }
    if (prof.on) {
        CHORE_HIP_CHECK(h, hipStreamSynchronize((hipStream_t)stream));
        for (int g = 0; g < G; ++g)
            for (size_t i = 0; i < P->steps.size(); ++i) {
                const Step& st = P->steps[i];
                if (st.kind != S_KERNEL) continue;
                float ms = 0.f;
                CHORE_HIP_CHECK(h, hipEventElapsedTime(&ms, prof.ev[(g * P->steps.size() + i) * 2], prof.ev[(g * P->steps.size() + i) * 2 + 1]));
                prof.ms[st.klass] += ms;
                prof.flops[st.klass] += st.flops;
                prof.bytes[st.klass] += st.bytes;
                prof.launches[st.klass] += 1;
            }
    }
    return CHORE_OK;
}

int chore_profile_enable(chore_handle* h, int on) {
    CHORE_ENTER(h);
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
