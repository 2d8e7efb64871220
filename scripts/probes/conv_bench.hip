// conv_bench.hip -- times single convolution layers of the encoder in isolation (development aid, not part of the library).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCHORE_CONV_ABLATE=1 -I chore_amd/csrc \
//         scripts/probes/conv_bench.hip chore_amd/csrc/conv_lds.hip chore_amd/csrc/conv_small.hip chore_amd/csrc/enc_misc.hip \
//         -o scripts/probes/conv_bench
//   scripts/probes/conv_bench [dtype=3] [iters=200]
//
// For every layer shape of the B = 4, 512 x 512 encoder it launches the layer exactly as a ConvBlock does (GroupNorm + ReLU
// prologue from exact statistics, raw copy, residual, statistics of both outputs) `iters` times back to back and prints
// the average time per launch by hipEvents, for the full kernel and for the ablation switches of conv_lds.hip (a
// truncated kernel's time is a lower bound of what the remaining phases cost; the differences are the phase breakdown).
#include "../../chore_amd/csrc/enc_common.h"
#include <cmath>
#include <cstring>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

struct Shape { const char* name; int taps, H, W, Cin, Cout; bool raw, res; };

static float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
}

extern "C" void chore_lds_poison(hipStream_t, const char*, int) {}   // capi.hip is not linked into the probe

int main(int argc, char** argv) {
    const int dtype = argc > 1 ? atoi(argv[1]) : CHORE_F16X3;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    const int B = 4;
    chore_handle hh;
    chore_handle* h = &hh;
    setenv("CHORE_CONV_LDS", "1", 1);   // launch_conv = conv_lds_kernel here; the specialised-wave kernel is called directly
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const Shape shapes[] = {
        {"c2  128^2 128->64 ", 9, 128, 128, 128, 64, true, true},
        {"c3  128^2  64->64 ", 9, 128, 128, 64, 64, false, true},
        {"c1   64^2 256->128", 9, 64, 64, 256, 128, true, true},
        {"c2   64^2 128->64 ", 9, 64, 64, 128, 64, true, true},
        {"c3   64^2  64->64 ", 9, 64, 64, 64, 64, false, true},
        {"c1  128^2 256->128", 9, 128, 128, 256, 128, true, true},
        {"1x1 128^2 256->256", 1, 128, 128, 256, 256, false, true},
        {"c1  256^2  64->64 ", 9, 256, 256, 64, 64, true, true},
        {"c1   32^2 256->128", 9, 32, 32, 256, 128, true, true},
        {"c2   32^2 128->64 ", 9, 32, 32, 128, 64, true, true},
    };
    const int dbgs[] = {0, 16, 256, 8, 8 | 4, 8 | 32, 8 | 1 | 2, 8 | 1 | 2 | 32, 4096 | 1, 8 | 4096 | 1, 1 << 29};
    const char* dbgn[] = {"full", "-atom", "-stats", "-epi", "-epi-mfma", "-epi-publish", "-epi-loads", "-epi-ld-pub", "-Bfrag-w", "-epi-Bfrag-w", "full again"};
    const size_t esz = dtype == CHORE_BF16 ? 2 : 4;
    printf("dtype %d, B %d, %d launches per number (us per launch)\n", dtype, B, iters);
    printf("%-20s %-13s %5s", "layer", "kernel", "WGs");
    for (const char* n : dbgn) printf(" %13s", n);
    printf("   GFLOP  TF(full)\n");
    int si = 0;
    for (const Shape& sh : shapes) {
        if (only >= 0 && si++ != only) continue;
        const size_t px = (size_t)B * sh.H * sh.W;
        const size_t nin = px * sh.Cin, nout = px * 256;
        std::vector<float> hin(nin), hw((size_t)sh.Cout * sh.Cin * sh.taps), hg(sh.Cin), hb(sh.Cin);
        unsigned seed = 12345u + sh.Cin * 7 + sh.H;
        for (auto& v : hin) v = frand(seed) * 1.5f + 0.2f;
        const float wsc = 1.0f / sqrtf((float)sh.Cin * sh.taps);
        for (auto& v : hw) v = frand(seed) * wsc * 1.7f;
        for (auto& v : hg) v = 1.0f + 0.1f * frand(seed);
        for (auto& v : hb) v = 0.1f * frand(seed);
        void *din, *dout, *draw, *dres, *dwpk;
        float *dw, *dgam, *dbet, *dinf;
        GroupStat *st_in, *st_raw, *st_out;
        CK(hipMalloc(&dinf, nin * 4));
        CK(hipMalloc(&din, nin * esz));
        CK(hipMalloc(&dout, nout * esz));
        CK(hipMalloc(&draw, nout * esz));
        CK(hipMalloc(&dres, nout * esz));
        CK(hipMalloc(&dw, hw.size() * 4));
        CK(hipMalloc(&dgam, sh.Cin * 4));
        CK(hipMalloc(&dbet, sh.Cin * 4));
        CK(hipMalloc(&dwpk, packed_conv_bytes(dtype, sh.taps, sh.Cin, sh.Cout)));
        CK(hipMalloc(&st_in, 8 * act_stats_bytes(B)));
        CK(hipMalloc(&st_raw, 8 * act_stats_bytes(B)));
        CK(hipMalloc(&st_out, 8 * act_stats_bytes(B)));
        CK(hipMemset(st_in, 0, act_stats_bytes(B)));
        CK(hipMemset(st_raw, 0, act_stats_bytes(B)));
        CK(hipMemset(st_out, 0, act_stats_bytes(B)));
        CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));   // fp32 / fp16x3: as is (bf16 runs on garbage halves: timing only)
        { std::vector<float> hr(nout); for (auto& v : hr) v = frand(seed); CK(hipMemcpy(dres, hr.data(), nout * 4, hipMemcpyHostToDevice)); }
        CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dgam, hg.data(), sh.Cin * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbet, hb.data(), sh.Cin * 4, hipMemcpyHostToDevice));
        View vin; vin.p = din; vin.cs = sh.Cin; vin.co = 0; vin.C = sh.Cin;
        if (launch_gn_stats(h, dtype == CHORE_BF16 ? CHORE_BF16 : CHORE_F32, vin, B, sh.H * sh.W, st_in, s)) { fprintf(stderr, "%s\n", h->err.c_str()); return 1; }
        if (launch_pack_conv(h, dtype, sh.taps, sh.Cin, sh.Cout, dw, dwpk, s)) { fprintf(stderr, "%s\n", h->err.c_str()); return 1; }
        ConvArgs a{};
        a.in = vin;
        a.in_st = st_in; a.gamma = dgam; a.beta = dbet;
        a.wpk = dwpk;
        const int oco = sh.Cout < 256 ? 32 : 0;
        a.out.p = dout; a.out.cs = 256; a.out.co = oco; a.out.C = sh.Cout;
        if (sh.raw) { a.raw.p = draw; a.raw.cs = sh.Cout; a.raw.co = 0; a.raw.C = sh.Cout; a.st_raw = st_raw; a.st_raw_C = sh.Cout; }
        if (sh.res) { a.res.p = dres; a.res.cs = 256; a.res.co = oco; a.res.C = sh.Cout; }
        a.st_out = st_out; a.st_out_C = 256; a.st_out_co = oco;
        a.B = B; a.H = sh.H; a.W = sh.W; a.Cout = sh.Cout;
        const ConvPlan pl = conv_plan(dtype, sh.taps, B, sh.H, sh.W, sh.Cin, sh.Cout);
        const int wgs = pl.tps == 0 ? -1 : B * pl.ntiles * (sh.Cout / pl.nt);
        unsigned long long* dticks;
        CK(hipMalloc(&dticks, 32 * 8));
        a.dbg_ticks = dticks;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const double gflop = 2.0 * sh.taps * sh.Cin * sh.Cout * (double)px * 1e-9;
        // variant 0 = conv_lds_kernel (or conv_small), then the forced tilings of conv_pc_kernel: th * 1000 + nt
        // 1 = the tiling conv_pc_plan picks  (the persistent kernel of round 4, conv_pp.hip, was removed in round 6: slower on every
        // layer, profiles/r04_conv_pp.txt; `git show b66b39a:chore_amd/csrc/conv_pp.hip` has it)
        const int forces9[] = {0, 1, 8128, 8064, 8032, 4064, 4032};
        const int forces1[] = {0, 1, 8128, 8064};
        std::vector<float> ref_out(nout), ref_raw(px * sh.Cout), got(nout);
        const int* fl = sh.taps == 9 ? forces9 : forces1;
        const int nfl = sh.taps == 9 ? 7 : 4;
        for (int fi = 0; fi < nfl; ++fi) {
            const int force = fl[fi];
            PcPlan pp{0, 0, 0, 0};
            int vw = wgs;
            if (force > 0) {
                pp = conv_pc_plan(dtype, sh.taps, B, sh.H, sh.W, sh.Cin, sh.Cout, force == 1 ? 0 : force);
                if (!pp.th || sh.Cout % pp.nt || sh.H % pp.th) continue;
                vw = B * (sh.H / pp.th) * (sh.W / 32) * (sh.Cout / pp.nt);
            }
            auto run = [&](int dbg) -> int {
                a.dbg = dbg | (1 << 30);
                return force ? launch_conv_pc(h, dtype, sh.taps, pp, a, s) : launch_conv(h, dtype, sh.taps, a, s);
            };
            // correctness against variant 0 (same arithmetic, another summation order over the chunks)
            CK(hipMemsetAsync(dout, 0, nout * esz, s));
            CK(hipMemsetAsync(draw, 0, nout * esz, s));
            if (run(0)) { fprintf(stderr, "%s\n", h->err.c_str()); return 1; }
            CK(hipStreamSynchronize(s));
            double err_out = 0, err_raw = 0, mag = 0;
            if (!force) {
                CK(hipMemcpy(ref_out.data(), dout, nout * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ref_raw.data(), draw, px * sh.Cout * 4, hipMemcpyDeviceToHost));
            } else {
                CK(hipMemcpy(got.data(), dout, nout * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < nout; ++i) { err_out = fmax(err_out, fabs((double)got[i] - ref_out[i])); mag = fmax(mag, fabs((double)ref_out[i])); }
                if (sh.raw) {
                    CK(hipMemcpy(got.data(), draw, px * sh.Cout * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < px * sh.Cout; ++i) err_raw = fmax(err_raw, fabs((double)got[i] - ref_raw[i]));
                }
            }
            printf("%-20s %-13s %5d", sh.name, force ? (std::string(force == 1 ? "auto" : (force == -1 ? "ppauto" : (force < 0 ? "pp" : "pc"))) + std::to_string(pp.th * 1000 + pp.nt) + "/" + std::to_string(pp.tps) + "/" + std::to_string(pp.nslot)).c_str() : "lds", vw);
            for (int i = 0; i < 400; ++i) run(0);   // clocks up before the first timed variant (it read 5 us high without)
            double full_us = 0;
            for (size_t di = 0; di < sizeof(dbgs) / sizeof(dbgs[0]); ++di) {
                for (int i = 0; i < 5; ++i)
                    if (run(dbgs[di])) { fprintf(stderr, "%s\n", h->err.c_str()); return 1; }
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < iters; ++i) run(dbgs[di]);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1000.0 / iters;
                if (di == 0) full_us = us;
                printf(" %13.2f", us);
            }
            printf("  %6.2f  %7.1f", gflop, gflop / full_us * 1e-3);
            if (force < 0) {   // persistent kernel: stamps 0 start, 1 main loop, 2 loop done, 3 last tile drained
                CK(hipMemset(dticks, 0, 32 * 8));
                run(0);
                CK(hipStreamSynchronize(s));
                unsigned long long tk[32];
                CK(hipMemcpy(tk, dticks, sizeof(tk), hipMemcpyDeviceToHost));
                for (int w = 0; w < 2; ++w) {
                    const unsigned long long* q = tk + 16 * w;
                    printf("  [%s us: pro %.2f loop %.2f final %.2f]", w ? "prod" : "cons", (double)(q[3] - q[1]) * 0.01, (double)(q[5] - q[3]) * 0.01,
                           (double)(q[7] - q[5]) * 0.01);
                }
            } else if (force) {   // phase stamps of the last full launch (workgroup in the middle of the grid)
                run(getenv("STAMP_DBG") ? atoi(getenv("STAMP_DBG")) : 0);
                CK(hipStreamSynchronize(s));
                unsigned long long tk[32];
                CK(hipMemcpy(tk, dticks, sizeof(tk), hipMemcpyDeviceToHost));
                for (int w = 0; w < 2; ++w) {
                    const unsigned long long* q = tk + 16 * w;
                    const double ghz = (double)(q[2 * 4] - q[0]) / ((double)(q[2 * 4 + 1] - q[1]) * 10.0);
                    printf("  [%s %.2f GHz us:", w ? "prod" : "cons", ghz);
                    for (int i = 1; i <= 5; ++i) printf(" %.2f", (double)(q[2 * i + 1] - q[2 * i - 1]) * 0.01);
                    printf(" | loop %.0f cyc @ %.2f GHz]", (double)(q[4] - q[2]), (double)(q[4] - q[2]) / ((double)(q[5] - q[3]) * 10.0));
                }
            }
            if (force) printf("  maxerr out %.3g raw %.3g (max |out| %.3g)", err_out, err_raw, mag);
            printf("\n");
            fflush(stdout);
        }
        CK(hipFree(dinf)); CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(draw)); CK(hipFree(dres)); CK(hipFree(dw));
        CK(hipFree(dgam)); CK(hipFree(dbet)); CK(hipFree(dwpk)); CK(hipFree(st_in)); CK(hipFree(st_raw)); CK(hipFree(st_out));
    }
    return 0;
}
