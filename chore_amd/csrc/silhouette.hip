// silhouette.hip -- differentiable silhouette rasterisation for the occlusion-aware mask loss of the object fit.
//
// Replaces what SilLossROI.forward (recon/obj_pose_roi.py:159-172) gets from the vendored neural_renderer:
// RasterizeFunction with return_alpha only (external/neural_renderer/neural_renderer/rasterize.py:14-170) and its
// CUDA kernels forward_face_index_map (cuda/rasterize_cuda_kernel.cu:24-215) and backward_pixel_map (:290-549).
// Same mathematics -- back-face culling, inside test in normalised coordinates, clamped barycentric weights from the
// pixel-space inverse, perspective-correct depth, z-buffer with near/far; the backward walks every edge of every
// front-facing triangle along both axes and distributes diff_grad / distance to the two edge vertices -- organised
// for this part instead of translated:
//   * the reference bins triangles into 4x4-pixel blocks through an atomically appended list with room for 512
//     entries (faces beyond that are silently dropped, and the z-buffer winner among equal depths depends on the
//     append order).  Here a setup kernel computes, per triangle, the inverse matrix and an integer pixel bounding
//     box; a workgroup owns a 16x16 pixel tile, streams all triangles of the image through LDS in chunks of 256
//     and every lane tests only those whose box meets the tile.  No list, no dropped faces, the winner is the
//     smallest depth and, among equal depths, the smallest triangle index: deterministic.
//     (5 000 triangles x 65 536 pixels is 0.3 G box tests per image: ~20 us.)
//   * the backward walks the same edges over the same pixels as the reference (which gives a triangle to ONE thread), but
//     a workgroup owns a triangle, a wave one (edge, axis) walk and a lane one edge position, eight pixels of a scan line
//     in flight; the per-vertex sums are combined by a fixed butterfly, so the result is deterministic.
// Degenerate edges (a zero denominator makes the crossing coordinate non-finite) are skipped, like the
// restatement in oracle/silhouette.py does; with real-valued vertices they have measure zero.
#include "common.h"

namespace {

struct TriSetup {            // per (image, triangle)
    float f[9];              // projected vertices: x0 y0 z0 x1 y1 z1 x2 y2 z2 (normalised [-1,1] + depth)
    float inv[9];            // pixel-space inverse (barycentric weights = inv * (xi, yi, 1))
    int x0, x1, y0, y1;      // inclusive pixel bounding box, x0 > x1 if the triangle is culled
};

__device__ __forceinline__ bool tri_backside(const float* f) {
    return __fmul_rn(f[7] - f[1], f[3] - f[0]) < __fmul_rn(f[4] - f[1], f[6] - f[0]);
}

__global__ void sil_setup_kernel(const float* __restrict__ faces, int n /*B*F*/, int size, TriSetup* __restrict__ ts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TriSetup t;
#pragma unroll
    for (int k = 0; k < 9; ++k) t.f[k] = faces[(size_t)i * 9 + k];
    t.x0 = 1; t.x1 = 0; t.y0 = 1; t.y1 = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) t.inv[k] = 0.f;
    if (!tri_backside(t.f)) {
        const float S = (float)size;
        float p[3][2];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int d = 0; d < 2; ++d) p[v][d] = 0.5f * ((t.f[3 * v + d] * S + S) - 1.0f);
        const float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1])) + p[1][0] * (p[2][1] - p[0][1]);
        const float m[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                            p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                            p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
#pragma unroll
        for (int k = 0; k < 9; ++k) t.inv[k] = m[k] / den;
        // conservative box: pixel centres inside the triangle lie within [min, max] of the vertex pixel coordinates;
        // one pixel of slack covers the rounding of the normalised-coordinate inside test
        const float xmin = fminf(fminf(p[0][0], p[1][0]), p[2][0]), xmax = fmaxf(fmaxf(p[0][0], p[1][0]), p[2][0]);
        const float ymin = fminf(fminf(p[0][1], p[1][1]), p[2][1]), ymax = fmaxf(fmaxf(p[0][1], p[1][1]), p[2][1]);
        if (xmin == xmin && ymin == ymin && xmax == xmax && ymax == ymax) {   // not NaN
            t.x0 = (int)fmaxf(floorf(xmin) - 1.f, 0.f);
            t.y0 = (int)fmaxf(floorf(ymin) - 1.f, 0.f);
            t.x1 = (int)fminf(ceilf(xmax) + 1.f, S - 1.f);
            t.y1 = (int)fminf(ceilf(ymax) + 1.f, S - 1.f);
        }
    }
    ts[i] = t;
}

constexpr int TILE_PX = 16, CHUNK = 256;

__global__ __launch_bounds__(256) void sil_fwd_kernel(const TriSetup* __restrict__ ts, int F, int size, float near,
                                                      float far, int* __restrict__ face_index,
                                                      float* __restrict__ alpha) {
    __shared__ TriSetup tri[CHUNK];      // 256 x 88 B = 22 KB
    __shared__ int hits[CHUNK];          // indices (within the chunk) of the triangles whose box meets this tile
    __shared__ int wcnt[4];
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * TILE_PX, ty0 = blockIdx.y * TILE_PX;
    const int lx = threadIdx.x % TILE_PX, ly = threadIdx.x / TILE_PX;
    const int xi = tx0 + lx, yi = ty0 + ly;
    const bool inside = xi < size && yi < size;
    const float xp = (float)((2.0 * xi + 1 - size) / size), yp = (float)((2.0 * yi + 1 - size) / size);
    const float xf = (float)xi, yf = (float)yi;
    float depth = far;
    int best = -1;
    for (int c0 = 0; c0 < F; c0 += CHUNK) {
        const int n = min(CHUNK, F - c0);
        bool meets = false;
        if ((int)threadIdx.x < n) {
            const TriSetup t = ts[(size_t)b * F + c0 + threadIdx.x];
            meets = t.x0 <= t.x1 && t.x0 <= tx0 + TILE_PX - 1 && t.x1 >= tx0 && t.y0 <= ty0 + TILE_PX - 1 && t.y1 >= ty0;
            tri[threadIdx.x] = t;
        }
        // compact in index order (keeps the z-buffer order deterministic): ballots and a prefix over the four waves (one
        // thread walking the 256 flags was 2.4 us per chunk, most of the kernel)
        const unsigned long long mb = __ballot(meets);
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (ln == 0) wcnt[wv] = __popcll(mb);
        __syncthreads();
        {
            int base = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) base += q < wv ? wcnt[q] : 0;
            if (meets) hits[base + __popcll(mb & ((1ull << ln) - 1ull))] = threadIdx.x;
        }
        const int nh = (wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]);
        __syncthreads();
        if (inside) {
            for (int h = 0; h < nh; ++h) {
                const int j = hits[h];
                const float* f = tri[j].f;
                if (__fmul_rn(yp - f[1], f[3] - f[0]) < __fmul_rn(xp - f[0], f[4] - f[1]) ||
                    __fmul_rn(yp - f[4], f[6] - f[3]) < __fmul_rn(xp - f[3], f[7] - f[4]) ||
                    __fmul_rn(yp - f[7], f[0] - f[6]) < __fmul_rn(xp - f[6], f[1] - f[7]))
                    continue;
                const float* m = tri[j].inv;
                float w[3], ws = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float v = (m[3 * k] * xf + m[3 * k + 1] * yf) + m[3 * k + 2];
                    v = fminf(fmaxf(v, 0.f), 1.f);     // NaN -> 0 like CUDA's fmax/fmin
                    w[k] = v;
                }
                ws = (w[0] + w[1]) + w[2];
#pragma unroll
                for (int k = 0; k < 3; ++k) w[k] = w[k] / ws;
                const float zp = 1.0f / ((w[0] / f[2] + w[1] / f[5]) + w[2] / f[8]);
                if (zp <= near || far <= zp) continue;
                if (zp < depth) { depth = zp; best = c0 + j; }
            }
        }
        __syncthreads();
    }
    if (inside) {
        const size_t o = ((size_t)b * size + yi) * size + xi;
        face_index[o] = best;
        alpha[o] = best >= 0 ? 1.f : 0.f;
    }
}

// one walk along edge (P0 -> P1) of triangle fn on one axis; P2 is the opposite vertex.  u = coordinate along the
// walk axis, v = across it.  Adds to g0 / g1 (the gradient of P0 / P1 along v).
struct Img {
    const int* fim; const float* alpha; const float* grad; int size;
    __device__ __forceinline__ size_t at(int axis, int d0, int d1) const {
        return axis == 0 ? (size_t)d1 * size + d0 : (size_t)d0 * size + d1;
    }
};

constexpr int SIL_U = 8;     // pixels of a line whose loads are in flight together
// The 64 lanes of a wave share one walk: lane l takes the positions d0_from + l, + 64, ... along the edge.
__device__ void sil_edge_walk(const Img& im, int fn, int axis, float u0, float v0, float u1, float v1, float u2, float v2,
                              float eps, int lane, float& g0, float& g1) {
    const int size = im.size;
    const float S = (float)size;
    int direction;
    if (axis == 0) direction = (u0 < u1) ? -1 : 1;
    else direction = (u0 < u1) ? 1 : -1;
    const int d0_from = (int)fmaxf(ceilf(fminf(u0, u1)), 0.f);
    const int d0_to = (int)fminf(fmaxf(u0, u1), S - 1.f);
    for (int d0 = d0_from + lane; d0 <= d0_to; d0 += 64) {
        const float fd0 = (float)d0;
        const float cross = (v1 - v0) / (u1 - u0) * (fd0 - u0) + v0;
        if (!(fabsf(cross) <= 3.0e38f)) continue;          // non-finite: degenerate edge
        const int d1_in = direction > 0 ? (int)floorf(cross) : (int)ceilf(cross);
        const int d1_out = d1_in + direction;
        if (d1_in < 0 || d1_in >= size || d1_out < 0 || d1_out >= size) continue;
        const float a_in = im.alpha[im.at(axis, d0, d1_in)], a_out = im.alpha[im.at(axis, d0, d1_out)];
        auto push = [&](int d1, float diff) {
            if (!(diff > 0.f)) return;
            const float t = ((float)d1 - cross);
            if (u1 != fd0) {
                float dist = (u1 - u0) / (u1 - fd0) * t * 2.0f / S;
                dist = dist > 0.f ? dist + eps : dist - eps;
                g0 -= diff / dist;
            }
            if (u0 != fd0) {
                float dist = (u1 - u0) / (fd0 - u0) * t * 2.0f / S;
                dist = dist > 0.f ? dist + eps : dist - eps;
                g1 -= diff / dist;
            }
        };
        if (im.fim[im.at(axis, d0, d1_in)] == fn) {          // 'out': beyond the edge up to the image border
            const int lim = direction > 0 ? size - 1 : 0;
            const int lo = max(min(d1_out, lim), 0), hi = min(max(d1_out, lim), size - 1);
            for (int d1 = lo; d1 <= hi; d1 += SIL_U) {      // SIL_U pixels' loads requested together, pushed in pixel order
                float al[SIL_U], gr[SIL_U];
#pragma unroll
                for (int u = 0; u < SIL_U; ++u) {
                    const size_t q = im.at(axis, d0, min(d1 + u, hi));
                    al[u] = im.alpha[q]; gr[u] = im.grad[q];
                }
#pragma unroll
                for (int u = 0; u < SIL_U; ++u)
                    if (d1 + u <= hi) push(d1 + u, (al[u] - a_in) * gr[u]);
            }
        }
        float c2;                                            // 'in': this face's pixels up to the opposite edge
        if ((fd0 - u0) * (fd0 - u2) < 0.f) c2 = (v2 - v0) / (u2 - u0) * (fd0 - u0) + v0;
        else c2 = (v1 - v2) / (u1 - u2) * (fd0 - u2) + v2;
        if (!(fabsf(c2) <= 3.0e38f)) continue;
        const int lim = direction > 0 ? (int)ceilf(c2) : (int)floorf(c2);
        const int lo = max(min(d1_in, lim), 0), hi = min(max(d1_in, lim), size - 1);
        for (int d1 = lo; d1 <= hi; d1 += SIL_U) {
            float al[SIL_U], gr[SIL_U];
            int fm[SIL_U];
#pragma unroll
            for (int u = 0; u < SIL_U; ++u) {
                const size_t q = im.at(axis, d0, min(d1 + u, hi));
                fm[u] = im.fim[q]; al[u] = im.alpha[q]; gr[u] = im.grad[q];
            }
#pragma unroll
            for (int u = 0; u < SIL_U; ++u)
                if (d1 + u <= hi && fm[u] == fn) push(d1 + u, (al[u] - a_out) * gr[u]);
        }
    }
}

// One workgroup per triangle, one wave per walk (3 edges x 2 axes), one lane per position along the edge: the reference
// (rasterize_cuda_kernel.cu:290-549) gives a whole triangle to ONE thread, whose six walks of up to `size` positions x up
// to `size` pixels each are a serial chain of tens of thousands of dependent loads -- 5.4 ms for a 1 024-triangle
// template that fills the 256-px ROI.  The per-lane partial sums are combined with a fixed butterfly and the six walks
// in the reference's order, so the result is deterministic; it differs from the serial sum by fp32 round-off only.
__global__ __launch_bounds__(384) void sil_bwd_kernel(const float* __restrict__ faces, const int* __restrict__ face_index,
                                                      const float* __restrict__ alpha, const float* __restrict__ grad_alpha,
                                                      int B, int F, int size, float eps, float* __restrict__ grad_faces) {
    const int i = blockIdx.x;
    const int b = i / F, fn = i % F;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float part[6][2];
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = faces[(size_t)i * 9 + k];
    const bool back = tri_backside(f);
    float g0 = 0.f, g1 = 0.f;
    if (!back) {
        const size_t img = (size_t)b * size * size;
        const Img im{face_index + img, alpha + img, grad_alpha + img, size};
        const float S = (float)size;
        float px[3], py[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            px[v] = 0.5f * ((f[3 * v] * S + S) - 1.0f);
            py[v] = 0.5f * ((f[3 * v + 1] * S + S) - 1.0f);
        }
        const int e = wave >> 1, axis = wave & 1;
        const int i0 = e, i1 = (e + 1) % 3, i2 = (e + 2) % 3;
        // axis 0: walk x, the gradient goes to y; axis 1: walk y, the gradient goes to x
        if (axis == 0) sil_edge_walk(im, fn, 0, px[i0], py[i0], px[i1], py[i1], px[i2], py[i2], eps, lane, g0, g1);
        else sil_edge_walk(im, fn, 1, py[i0], px[i0], py[i1], px[i1], py[i2], px[i2], eps, lane, g0, g1);
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            g0 += __shfl_xor(g0, o);
            g1 += __shfl_xor(g1, o);
        }
    }
    if (lane == 0) { part[wave][0] = g0; part[wave][1] = g1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float g[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = 0.f;
        for (int e = 0; e < 3; ++e) {           // the reference's order: edges 0, 1, 2, axis 0 before axis 1
            const int i0 = e, i1 = (e + 1) % 3;
            g[3 * i0 + 1] += part[2 * e][0];
            g[3 * i1 + 1] += part[2 * e][1];
            g[3 * i0 + 0] += part[2 * e + 1][0];
            g[3 * i1 + 0] += part[2 * e + 1][1];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) grad_faces[(size_t)i * 9 + k] = g[k];
    }
}

// ---- placement + camera projection of the template, straight into the rasteriser's triangle list -----------------------
// SilLossROI.apply_transformation (recon/obj_pose_roi.py:159-162: s (v Ro + to)) -> neural_renderer projection
// (external/neural_renderer/neural_renderer/projection.py:6-43: v Rc^T + tc, perspective divide by z + eps, the radial /
// tangential distortion polynomial, K, the [-1,1] mapping with the vertical flip) -> vertices_to_faces with both windings.
struct SilCam {
    float k1, k2, p1, p2, k3, orig, eps;
};
struct SilProj {      // one vertex through the chain, with what the backward needs
    float w[3];       // s (v0 Ro + to)
    float pre[3];     // v0 Ro + to
    float c[3];       // camera space
    float out[3];     // u, v, z
};
__device__ __forceinline__ void sil_project_vertex(const float* v0, const float* Ro, const float* to, float s, const float* Rc,
                                                   const float* tc, const float* K, const SilCam& cam, SilProj& p) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p.pre[c] = ((v0[0] * Ro[c] + v0[1] * Ro[3 + c]) + v0[2] * Ro[6 + c]) + to[c];
        p.w[c] = s * p.pre[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p.c[c] = ((p.w[0] * Rc[c * 3] + p.w[1] * Rc[c * 3 + 1]) + p.w[2] * Rc[c * 3 + 2]) + tc[c];
    const float zi = p.c[2] + cam.eps;
    const float x_ = p.c[0] / zi, y_ = p.c[1] / zi;
    const float r2 = x_ * x_ + y_ * y_;
    const float rad = 1.f + cam.k1 * r2 + cam.k2 * (r2 * r2) + cam.k3 * (r2 * r2 * r2);
    const float xd = x_ * rad + 2.f * cam.p1 * x_ * y_ + cam.p2 * (r2 + 2.f * (x_ * x_));
    const float yd = y_ * rad + cam.p1 * (r2 + 2.f * (y_ * y_)) + 2.f * cam.p2 * x_ * y_;
    const float u = (K[0] * xd + K[1] * yd) + K[2];
    const float vv = cam.orig - ((K[3] * xd + K[4] * yd) + K[5]);
    p.out[0] = 2.f * (u - cam.orig / 2.0f) / cam.orig;
    p.out[1] = 2.f * (vv - cam.orig / 2.0f) / cam.orig;
    p.out[2] = p.c[2];
}

// thread = (frame, face of the doubled list): faces2[f] for f < F, the reversed corner order for f >= F (fill_back)
__global__ __launch_bounds__(256) void sil_project_fwd_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                              const float* __restrict__ Ro, const float* __restrict__ to,
                                                              const float* __restrict__ sc, const float* __restrict__ K,
                                                              const float* __restrict__ Rc, const float* __restrict__ tc, int cam_bcast,
                                                              SilCam cam, int V, int F, float* __restrict__ tri) {
    const int b = blockIdx.y, f2 = blockIdx.x * 256 + threadIdx.x;
    if (f2 >= 2 * F) return;
    const int f = f2 < F ? f2 : f2 - F;
    const float* rc = Rc + (cam_bcast ? 0 : (size_t)b * 9);
    const float* t3 = tc + (cam_bcast ? 0 : (size_t)b * 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int corner = f2 < F ? c : 2 - c;
        const int vi = faces[((size_t)b * F + f) * 3 + corner];
        SilProj p;
        sil_project_vertex(verts + ((size_t)b * V + vi) * 3, Ro + (size_t)b * 9, to + (size_t)b * 3, sc[b], rc, t3, K + (size_t)b * 9, cam, p);
        float* o = tri + (((size_t)b * 2 * F + f2) * 3 + c) * 3;
        o[0] = p.out[0]; o[1] = p.out[1]; o[2] = p.out[2];
    }
}

// one workgroup per frame: every vertex collects the gradients of its triangle corners in the order of the adjacency list
// (adj_off (V+1), adj (entries = (f2 * 3 + c) of the doubled list)), pulls them back through the projection, and the
// 13 sums over the vertices (d Ro, d to, d s) are fixed-order fp64 trees
__global__ __launch_bounds__(256) void sil_project_bwd_kernel(const float* __restrict__ verts, const float* __restrict__ Ro,
                                                              const float* __restrict__ to, const float* __restrict__ sc,
                                                              const float* __restrict__ K, const float* __restrict__ Rc,
                                                              const float* __restrict__ tc, int cam_bcast, SilCam cam, int V, int F,
                                                              const int* __restrict__ adj_off, const int* __restrict__ adj,
                                                              const float* __restrict__ g_tri, float* __restrict__ dRo,
                                                              float* __restrict__ dto, float* __restrict__ dsc) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* rc = Rc + (cam_bcast ? 0 : (size_t)b * 9);
    const float* t3 = tc + (cam_bcast ? 0 : (size_t)b * 3);
    const float* Kb = K + (size_t)b * 9;
    const float s = sc[b];
    float a[13];
#pragma unroll
    for (int e = 0; e < 13; ++e) a[e] = 0.f;
    for (int v = tid; v < V; v += 256) {
        float g[3] = {0.f, 0.f, 0.f};
        for (int q = adj_off[v]; q < adj_off[v + 1]; ++q) {
            const float* gt = g_tri + ((size_t)b * 2 * F * 3 + adj[q]) * 3;
            g[0] += gt[0]; g[1] += gt[1]; g[2] += gt[2];
        }
        const float* v0 = verts + ((size_t)b * V + v) * 3;
        SilProj p;
        sil_project_vertex(v0, Ro + (size_t)b * 9, to + (size_t)b * 3, s, rc, t3, Kb, cam, p);
        const float zi = p.c[2] + cam.eps;
        const float x_ = p.c[0] / zi, y_ = p.c[1] / zi, r2 = x_ * x_ + y_ * y_;
        const float rad = 1.f + cam.k1 * r2 + cam.k2 * (r2 * r2) + cam.k3 * (r2 * r2 * r2);
        const float D = cam.k1 + 2.f * cam.k2 * r2 + 3.f * cam.k3 * (r2 * r2);
        const float gu = g[0] * (2.f / cam.orig), gv = -g[1] * (2.f / cam.orig);
        const float gxd = gu * Kb[0] + gv * Kb[3], gyd = gu * Kb[1] + gv * Kb[4];
        const float gx_ = gxd * (rad + 2.f * D * x_ * x_ + 2.f * cam.p1 * y_ + 6.f * cam.p2 * x_) +
                          gyd * (2.f * D * x_ * y_ + 2.f * cam.p1 * x_ + 2.f * cam.p2 * y_);
        const float gy_ = gxd * (2.f * D * x_ * y_ + 2.f * cam.p1 * x_ + 2.f * cam.p2 * y_) +
                          gyd * (rad + 2.f * D * y_ * y_ + 6.f * cam.p1 * y_ + 2.f * cam.p2 * x_);
        const float gc[3] = {gx_ / zi, gy_ / zi, g[2] - (gx_ * x_ + gy_ * y_) / zi};
        float gw[3];      // gradient of the placed vertex w = s (v0 Ro + to):  c = Rc w + tc
#pragma unroll
        for (int k = 0; k < 3; ++k) gw[k] = (gc[0] * rc[k] + gc[1] * rc[3 + k]) + gc[2] * rc[6 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[k * 3 + c] += v0[k] * gw[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[9 + c] += gw[c];
            a[12] += gw[c] * p.pre[c];
        }
    }
#pragma unroll
    for (int e = 0; e < 13; ++e) {
        sh[tid] = (double)a[e];
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) sh[tid] += sh[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            if (e < 9) dRo[(size_t)b * 9 + e] = (float)(sh[0] * (double)s);
            else if (e < 12) dto[b * 3 + e - 9] = (float)(sh[0] * (double)s);
            else dsc[b] = (float)sh[0];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t chore_silhouette_workspace_bytes(int B, int F) { return (size_t)B * F * sizeof(TriSetup); }

extern "C" int chore_silhouette_fwd(chore_handle* h, const float* faces, int B, int F, int size, float near_z,
                                    float far_z, int* face_index, float* alpha, void* workspace,
                                    chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!faces || !face_index || !alpha || !workspace) CHORE_FAIL(h, CHORE_EINVAL, "chore_silhouette_fwd: null argument");
    if (B <= 0 || F <= 0 || size <= 0 || size > 4096 || B > 65535)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_silhouette_fwd: bad sizes B=%d F=%d size=%d", B, F, size);
    hipStream_t s = (hipStream_t)stream;
    TriSetup* ts = (TriSetup*)workspace;
    const int n = B * F;
    hipLaunchKernelGGL(sil_setup_kernel, dim3((n + 255) / 256), dim3(256), 0, s, faces, n, size, ts);
    const int tiles = (size + TILE_PX - 1) / TILE_PX;
    hipLaunchKernelGGL(sil_fwd_kernel, dim3(tiles, tiles, B), dim3(256), 0, s, ts, F, size, near_z, far_z, face_index,
                       alpha);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

extern "C" int chore_silhouette_bwd(chore_handle* h, const float* faces, const int* face_index, const float* alpha,
                                    const float* grad_alpha, int B, int F, int size, float eps, float* grad_faces,
                                    chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!faces || !face_index || !alpha || !grad_alpha || !grad_faces)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_silhouette_bwd: null argument");
    if (B <= 0 || F <= 0 || size <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_silhouette_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int n = B * F;
    hipLaunchKernelGGL(sil_bwd_kernel, dim3(n), dim3(384), 0, s, faces, face_index, alpha, grad_alpha, B, F, size, eps,
                       grad_faces);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

// dist5: HOST floats {k1, k2, p1, p2, k3} or NULL (no distortion); cam_R / cam_t: (B,3,3) / (B,3), or one matrix / vector for
// all frames when cam_broadcast != 0
extern "C" int chore_sil_project_fwd(chore_handle* h, const float* verts, const int* faces, const float* obj_R, const float* obj_t,
                                     const float* obj_s, const float* K, const float* cam_R, const float* cam_t, int cam_broadcast,
                                     const float* dist5, float orig_size, float eps, int B, int V, int F, float* tri,
                                     chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!verts || !faces || !obj_R || !obj_t || !obj_s || !K || !cam_R || !cam_t || !tri)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_sil_project_fwd: null argument");
    if (B <= 0 || B > 65535 || V <= 0 || F <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_sil_project_fwd: bad sizes B=%d V=%d F=%d", B, V, F);
    SilCam cam{0.f, 0.f, 0.f, 0.f, 0.f, orig_size, eps};
    if (dist5) { cam.k1 = dist5[0]; cam.k2 = dist5[1]; cam.p1 = dist5[2]; cam.p2 = dist5[3]; cam.k3 = dist5[4]; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sil_project_fwd_kernel, dim3((2 * F + 255) / 256, B), dim3(256), 0, s, verts, faces, obj_R, obj_t, obj_s, K, cam_R,
                       cam_t, cam_broadcast, cam, V, F, tri);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}

extern "C" int chore_sil_project_bwd(chore_handle* h, const float* verts, const float* obj_R, const float* obj_t, const float* obj_s,
                                     const float* K, const float* cam_R, const float* cam_t, int cam_broadcast, const float* dist5,
                                     float orig_size, float eps, int B, int V, int F, const int* adj_off, const int* adj,
                                     const float* g_tri, float* d_obj_R, float* d_obj_t, float* d_obj_s, chore_stream_t stream) {
    CHORE_ENTER(h);
    if (!verts || !obj_R || !obj_t || !obj_s || !K || !cam_R || !cam_t || !adj_off || !adj || !g_tri || !d_obj_R || !d_obj_t || !d_obj_s)
        CHORE_FAIL(h, CHORE_EINVAL, "chore_sil_project_bwd: null argument");
    if (B <= 0 || V <= 0 || F <= 0) CHORE_FAIL(h, CHORE_EINVAL, "chore_sil_project_bwd: bad sizes B=%d V=%d F=%d", B, V, F);
    SilCam cam{0.f, 0.f, 0.f, 0.f, 0.f, orig_size, eps};
    if (dist5) { cam.k1 = dist5[0]; cam.k2 = dist5[1]; cam.p1 = dist5[2]; cam.p2 = dist5[3]; cam.k3 = dist5[4]; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sil_project_bwd_kernel, dim3(B), dim3(256), 0, s, verts, obj_R, obj_t, obj_s, K, cam_R, cam_t, cam_broadcast, cam, V,
                       F, adj_off, adj, g_tri, d_obj_R, d_obj_t, d_obj_s);
    CHORE_LAUNCH_CHECK(h, s);
    return CHORE_OK;
}
