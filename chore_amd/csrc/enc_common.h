// enc_common.h -- kernel argument blocks and launchers of the stacked-hourglass encoder.
//
// Activations are NHWC, element type T = float (CHORE_F32) or bf16 stored as unsigned short
// (CHORE_BF16).  A "view" is (pointer, channel stride of the buffer, channel offset, channels):
// ConvBlock's concat (model/net_util.py:388) is never materialised -- each conv writes its slice.
#pragma once
#include "common.h"

typedef unsigned short bf16_t;

struct View {
    void* p = nullptr;
    int cs = 0;   // channel stride (channels of the underlying buffer)
    int co = 0;   // channel offset of this view
    int C = 0;    // channels in the view
};

constexpr int GN_GROUPS = 32;
constexpr int GN_SPLITS_MAX = 64;

struct ConvArgs {
    View in;            // input activations
    const float* ss;    // [B][Cin][2] GroupNorm scale/shift fused with ReLU into the operand load, or null
    const void* wpk;    // fragment-ordered weights (pack_conv_weights)
    const float* bias;  // [Cout] or null
    View out;           // acc + bias + res + res2
    View raw;           // optional: acc + bias
    View res, res2;     // optional residuals (may alias out)
    int B, H, W, Cout;
    // optional fused GroupNorm statistics of what this launch stores: per (image, tile, channel)
    // (sum, sum of squares) partials in [B][tiles][C][2] buffers -- st_raw for the `raw` view,
    // st_out for the `out` view; *_C / *_co = channels of the normalised tensor / offset of this slice
    // *_tiles = tile stride of the buffer (>= tiles of this launch)
    float* st_raw = nullptr; int st_raw_C = 0, st_raw_co = 0, st_raw_tiles = 0;
    float* st_out = nullptr; int st_out_C = 0, st_out_co = 0, st_out_tiles = 0;
    int dbg = 0;   // ablation bits for kernel experiments (CHORE_CONV_DBG): 1 no weight loads, 2 no patch
                   // prefetch, 4 no MFMA, 8 no epilogue -- results are wrong when set
};

struct ConvPlan { int nt, th, ntiles; };
ConvPlan conv_plan(int taps, int B, int H, int W, int Cout);   // tile configuration launch_conv will use

int launch_conv(chore_handle* h, int dtype, int taps /*1|9*/, const ConvArgs& a, hipStream_t s);
size_t packed_conv_bytes(int dtype, int taps, int Cin, int Cout);
int launch_pack_conv(chore_handle* h, int dtype, int taps, int Cin, int Cout, const float* w /*(O,C,k,k)*/,
                     void* dst, hipStream_t s);

// --- misc kernels (enc_misc.hip) ---
int launch_stem(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W,
                const float* wk /*[Cin*49][64]*/, const float* bias, void* out /*(B,H/2,W/2,64)*/, hipStream_t s);
int launch_pack_stem(chore_handle* h, int Cin, const float* w /*(64,Cin,7,7)*/, float* dst, hipStream_t s);
int gn_splits(int HW);
int launch_gn_partial(chore_handle* h, int dtype, const View& x, int B, int HW, float* partial, hipStream_t s);
int launch_gn_finalize(chore_handle* h, const float* partial, int B, int HW, int C, const float* gamma,
                       const float* beta, float* ss, hipStream_t s);
// statistics assembled from conv-epilogue tile partials: up to 3 channel slices, each written by a
// launch with its own tile count
struct TileStats {
    const float* p = nullptr;   // [B][max_tiles][C][2]
    int max_tiles = 0, nslices = 0;
    int c_end[3] = {0, 0, 0};   // exclusive end channel of each slice
    int ntiles[3] = {0, 0, 0};
};
int launch_gn_finalize_tiles(chore_handle* h, const TileStats& ts, int B, int HW, int C, const float* gamma,
                             const float* beta, float* ss, hipStream_t s);
int launch_gn_apply_relu(chore_handle* h, int dtype, const View& x, const float* ss, const View& y, int B,
                         int HW, hipStream_t s);
int launch_avgpool2(chore_handle* h, int dtype, const View& x, const View& y, int B, int H, int W, hipStream_t s);
// y = a + bicubic_up2(low)   (low is (B,H,W,C), a and y are (B,2H,2W,C); y may alias a)
int launch_upadd(chore_handle* h, int dtype, const View& a, const View& low, const View& y, int B, int H, int W,
                 hipStream_t s);
int launch_copy_f32(chore_handle* h, const float* src, float* dst, size_t n, hipStream_t s);
