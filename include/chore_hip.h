/*
 * chore_hip.h -- C ABI of libchore_hip.so, the MI355X (gfx950) implementation of CHORE's
 * implicit-field query + fitting hot path.
 *
 * The reference (xiexh20/CHORE) has no FFI seam of its own for this path: the hot path is pure
 * ATen called from Python (model/chore.py:87-167).  The only C boundary in the reference tree is
 * the vendored rasterizer's pybind module (external/neural_renderer/neural_renderer/cuda/
 * rasterize_cuda.cpp:201-207), whose conventions we mirror: the CALLER allocates every buffer
 * (inputs, outputs, workspaces), functions fill them and return; work is enqueued on the caller's
 * stream; no hidden synchronisation, no internal threads.  Each entry point below cites the
 * reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise
 *   - return 0 (CHORE_OK) on success, negative CHORE_E* otherwise; chore_last_error() gives text
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream
 *   - nothing here calls hipMalloc/hipFree/hipDeviceSynchronize on the hot path, so every call is
 *     hipGraph-capturable
 *   - one handle per (process, device); a handle is not thread-safe
 */
#ifndef CHORE_HIP_H
#define CHORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct chore_handle chore_handle;
typedef void* chore_stream_t;

enum {
    CHORE_OK = 0,
    CHORE_EINVAL = -1, /* bad argument (shape, dtype, null pointer)   */
    CHORE_EHIP = -2,   /* a HIP runtime call failed                    */
    CHORE_ESTATE = -3, /* missing weights / wrong call order           */
    CHORE_ENOMEM = -4  /* caller workspace too small                   */
};

/* storage / MFMA-operand type of feature maps and packed weights */
enum { CHORE_F32 = 0, CHORE_BF16 = 1, CHORE_F16X3 = 2, CHORE_F16 = 3 };
/* CHORE_F16 ("fp16 fields", inference only): IEEE half feature maps; the encoder's convolutions multiply the fp16 activation
 * by the weight's fp16 hi and lo parts (two MFMAs per product, fp32 accumulation); the heads run as under CHORE_HEADS_X3. */
/* query entry points (chore_query_fwd / _bwd_points / _fwd_train / _bwd_train, chore_heads_wgrad): OR into the map type
 * to run the MLP heads on the fp16 matrix cores with hi/lo split operands (fp32-grade results, see csrc/heads_x3.h)
 * instead of the native fp32 MFMA.  CHORE_F16X3 there means CHORE_F32 | CHORE_HEADS_X3. */
enum { CHORE_HEADS_X3 = 0x100 };

/* one tensor of a reference state_dict (Appendix D of SURVEY.md): name, device pointer to its
 * contiguous fp32 data in the reference layout, element count */
typedef struct {
    const char* name;
    const void* ptr;
    int64_t numel;
} chore_weight_desc;

/* encoder topology = the fields of config/chore-release.json the hot path reads
 * (model/HGFilters.py:57-142) */
typedef struct {
    int in_channels;   /* 5 for input_type RGBM3                         */
    int num_stack;     /* 5                                              */
    int num_hourglass; /* 2 (recursion depth of one hourglass)           */
    int hourglass_dim; /* 256                                            */
} chore_encoder_cfg;

int chore_version(void);
int chore_create(chore_handle** out, int device_ordinal);
int chore_destroy(chore_handle* h);
const char* chore_last_error(const chore_handle* h);

/* Streams restricted to a subset of the compute units (no reference counterpart: the reference's loader loop,
 * recon/recon_fit_behave.py:41-76, is serial; this serves the pipelined loop of ReconFitterBehave.fit_recon, where the
 * preparation of batch k+1 must leave CUs free for the small kernels of batch k's optimisation).
 * mask: n_words x 32 bits, bit i = one CU (dealt round robin over the XCDs); chore_cu_count = CUs of the handle's device.
 * The stream is a plain hipStream_t: hand it to any entry point, destroy it with chore_stream_destroy (waits for it). */
int chore_cu_count(chore_handle* h);
int chore_stream_create_cu_mask(chore_handle* h, const uint32_t* mask, int n_words, chore_stream_t* out);
int chore_stream_destroy(chore_handle* h, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-point MLP heads + fused query  (replaces CHORE.query, model/chore.py:107-154:
 * camera.project_points model/camera.py:44-88, index()/grid_sample model/geometry.py:4-14 twice,
 * torch.cat, CHORE.decode model/chore.py:156-167, the df[~in_img]=OUT_DIST fill :147-150)
 * ------------------------------------------------------------------------------------------- */

/* bytes of the packed (MFMA-fragment-ordered) weight arena of the four heads */
size_t chore_heads_arena_bytes(int dtype);

/* Repack the four decoders' Conv1d weights into the arena.  `descs` must contain
 * {df,part_predictor,pca_predictor,center_predictor}.{0,2,4,6}.{weight,bias} (fp32, reference
 * layout (out,in,1)).  model/chore.py:49-55,74-85. */
int chore_heads_pack(chore_handle* h, const chore_weight_desc* descs, int n_descs, int dtype,
                     void* arena, chore_stream_t stream);

/* Camera constants in the order (fx_px, fy_px, cx_px, cy_px, crop/2, crop) -- HOST pointer to 6
 * floats, values as computed in model/camera.py:26-42. */

/* Fused forward query.
 *   points      (B,N,3) fp32 camera-space      crop_center (B,2) fp32
 *   feat        (B,FH,FW,256) NHWC `dtype`     tmpx (B,TH,TW,64) NHWC `dtype`
 *   df (B,2,N)  pca (B,9,N)  parts (B,14,N)  centers (B,6,N)  fp32;  in_img (B,N) uint8 (may be NULL)
 * df is written with OUT_DIST (5.0) where the projected point falls outside [-1,1]^2.
 * Any of df / pca / parts / centers may be NULL (not all four): that head is not evaluated (its wave leaves after the
 * gather).  chore_query_bwd_points skips the chain of every head whose upstream gradient is NULL. */
int chore_query_fwd(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                    const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                    const void* heads_arena, const float* cam6_host, float* df, float* pca,
                    float* parts, float* centers, uint8_t* in_img, chore_stream_t stream);

/* chore_query_fwd in SORTED ORDER: with a workspace of chore_query_fwd_workspace_bytes(B, N) bytes (NULL = chore_query_fwd) the points of
 * a large query with the fp16 x 3 heads are first ordered by the 8 x 8-texel tile of the feature map their sample falls into (two
 * small launches writing a permutation into the workspace) and the tiles of 64 points are formed in that order, so that a tile's
 * gather (BasePIFuNet.index, model/geometry.py:4-14, called from model/chore.py:134-143) reads a texel row once instead of once per
 * point: a third of the fetched bytes.  Every point's arithmetic is unchanged and its outputs go to its own column: results are
 * bit-identical to chore_query_fwd's.  The two extra launches cost more time than the gather saves on one MI355X (DESIGN.md section
 * 4): the host code passes a workspace only with CHORE_QUERY_SORTED=1.  Shapes the ordering does not cover (fewer than 8 192 points,
 * feature maps of more than 256 tiles, the native-fp32 heads) run exactly as chore_query_fwd. */
size_t chore_query_fwd_workspace_bytes(int B, int N);
int chore_query_fwd_ws(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                       const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                       const void* heads_arena, const float* cam6_host, float* df, float* pca,
                       float* parts, float* centers, uint8_t* in_img, void* workspace, chore_stream_t stream);

/* Pixel-aligned feature sample without the heads: BasePIFuNet.index (model/BasePIFuNet.py:23 ->
 * model/geometry.py:4-14) on both maps plus z_feat, concatenated as in model/chore.py:139-143.
 *   features (B,N,323) fp32 point-major [feat 0..255 | x y z-2.2 | tmpx 0..63]
 *   nxy (B,N,2) normalised image coordinates of model/camera.py:44-88 (may be NULL)
 *   in_img (B,N) uint8 (may be NULL) */
int chore_sample_features(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                          const float* cam6_host, float* features, float* nxy, uint8_t* in_img,
                          chore_stream_t stream);

/* Backward of chore_query_fwd with respect to the POINTS only (what
 * recon/generator.py:62-77 `df_target.sum().backward()` and every fitting loss need):
 * g_* are the upstream gradients of the four outputs (any may be NULL = zero); the forward is
 * recomputed inside the kernel, nothing is saved.  Gradients of df at out-of-image points are
 * dropped (the in-place fill of model/chore.py:149 cuts the graph there).
 *   dpoints (B,N,3) fp32, overwritten. */
int chore_query_bwd_points(chore_handle* h, const float* points, const float* crop_center, int B,
                           int N, const void* feat, int FH, int FW, const void* tmpx, int TH,
                           int TW, int dtype, const void* heads_arena, const float* cam6_host,
                           const float* g_df, const float* g_pca, const float* g_parts,
                           const float* g_centers, float* dpoints, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training backward of the query (first half of the backward of CHORE.forward, model/chore.py:176-190: what the
 * reference's autograd computes for Trainer.train_step, trainer/trainer.py:76-131, through decode :156-167 and
 * index / grid_sample model/geometry.py:4-14).  chore_query_bwd_train recomputes the forward like
 * chore_query_bwd_points and, besides dpoints (optional), stages in `staging` (chore_query_train_bytes):
 *   X   [B*N][328]          the 323-vector of every point, zero padded
 *   H   [3][4][B*N][128]    relu outputs of hidden layers 1..3, heads in the order df, parts, pca, centers
 *   dZ  [3][4][B*N][128]    gradients w.r.t. the pre-activations of layers 1..3
 *   dX  [B*N][328]          gradient w.r.t. the 323-vector, summed over the heads
 * chore_heads_wgrad turns the staging into the gradients of all 32 head parameters (dW_l = dZ_l^T H_{l-1}, db_l = column
 * sums of dZ_l; the output layers use the upstream gradients directly -- pass g_df already zeroed where the point is
 * outside the image, model/chore.py:147-150): `grads` = chore_heads_wgrad_floats() floats, per head in kernel order
 * (df, parts, pca, centers): W1 (128,323) b1 (128) W2 (128,128) b2 W3 b3 W4 (out,128) b4 (out); ordered partial sums,
 * bit-reproducible.  workspace: chore_heads_wgrad_workspace_bytes().
 * chore_scatter_features turns dX into the gradients of the two feature maps: dfeat (B,FH,FW,256) and dtmpx
 * (B,TH,TW,64), fp32 NHWC, written (accumulate = 0) or added to (accumulate = 1); tile-gather, no atomics.
 * ------------------------------------------------------------------------------------------- */
size_t chore_query_train_bytes(int B, int N);
/* training forward: chore_query_fwd that also stages the 323-vectors and the ReLU outputs (X, H of the layout above) */
int chore_query_fwd_train(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int map_dtype,
                          const void* heads_arena, const float* camera, float* df, float* pca, float* parts,
                          float* centers, uint8_t* in_img, void* staging, chore_stream_t stream);
/* have_forward != 0: `staging` already holds X and H from chore_query_fwd_train of the same inputs; the backward
 * then recomputes nothing (ReLU masks are read back) and only adds dZ and dX */
int chore_query_bwd_train(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                          const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int map_dtype,
                          const void* heads_arena, const float* camera, const float* g_df, const float* g_pca,
                          const float* g_parts, const float* g_centers, void* staging, float* dpoints,
                          int have_forward, chore_stream_t stream);
size_t chore_heads_wgrad_floats(void);
size_t chore_heads_wgrad_workspace_bytes(void);
int chore_heads_wgrad(chore_handle* h, const void* staging, int B, int N, const float* g_df, const float* g_pca,
                      const float* g_parts, const float* g_centers, float* grads, void* workspace,
                      int heads_x3 /* bit 0: fp16 matrix cores, split operands, per-chunk scales (see CHORE_HEADS_X3); bit 1: add to
                                      `grads` instead of overwriting (the stacks of a step into one arena) */,
                      chore_stream_t stream);
int chore_scatter_features(chore_handle* h, const float* points, const float* crop_center, int B, int N, int FH, int FW,
                           int TH, int TW, const float* camera, const void* staging, float* dfeat, float* dtmpx,
                           int accumulate, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stacked-hourglass encoder  (replaces HGFilter.forward model/HGFilters.py:144-185,
 * HourGlass._forward :26-50, ConvBlock.forward model/net_util.py:374-396)
 * ------------------------------------------------------------------------------------------- */

/* bytes of the packed encoder weight arena */
size_t chore_encoder_arena_bytes(const chore_encoder_cfg* cfg, int dtype);

/* Repack `image_filter.*` tensors of the state_dict (fp32, reference layouts) into the arena. */
int chore_encoder_pack(chore_handle* h, const chore_encoder_cfg* cfg,
                       const chore_weight_desc* descs, int n_descs, int dtype, void* arena,
                       chore_stream_t stream);

/* caller-allocated scratch needed by chore_encode_fwd for this shape */
size_t chore_encoder_workspace_bytes(const chore_encoder_cfg* cfg, int B, int H, int W, int dtype);

/* images (B,C,H,W) fp32 NCHW in [0,1] as data/test_data.py:107-125 delivers them.
 * feat_out[i] (B,H/4,W/4,256) NHWC `dtype` for the last `n_stack_out` stacks (eval: 1, training:
 * num_stack; model/chore.py:93-96), tmpx (B,H/2,W/2,64), normx (B,H/4,W/4,128) (may be NULL). */
int chore_encode_fwd(chore_handle* h, const chore_encoder_cfg* cfg, const float* images, int B,
                     int H, int W, int dtype, const void* arena, void* workspace,
                     size_t workspace_bytes, void* const* feat_out, int n_stack_out, void* tmpx,
                     void* normx, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SMPL / SMPL-H linear blend skinning  (replaces SMPL_Layer.forward,
 * lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:72-175 -- batch_rodrigues
 * rodrigues_layer.py:41-52, th_posemap_axisang / subtract_flat_id tensutils.py:6-53 -- and its
 * autograd w.r.t. pose, betas and trans, which is what recon/recon_fit_behave.py:224-291 optimises)
 *   model buffers (fp32, reference layouts): v_template (V,3), shapedirs (V,3,num_betas),
 *   posedirs (V,3,9(J-1)), J_regressor (J,V) dense, weights (V,J); parents: HOST array of J ints
 *   (kintree_table[0]; parents[0] ignored, parents[i] < i).
 * ------------------------------------------------------------------------------------------- */
size_t chore_smpl_arena_bytes(int V, int J, int num_betas);
size_t chore_smpl_workspace_bytes(int V, int J, int num_betas, int B);
int chore_smpl_pack(chore_handle* h, int V, int J, int num_betas, const float* v_template,
                    const float* shapedirs, const float* posedirs, const float* J_regressor,
                    const float* weights, const int* parents_host, void* arena, chore_stream_t stream);
/* pose (B,3J) axis-angle, betas (B,num_betas), trans (B,3), offsets (B,V,3) or NULL ->
 * verts (B,V,3), joints (B,J,3), v_posed (B,V,3), naked (B,V,3).  `workspace` must be kept untouched
 * until the matching chore_smpl_lbs_bwd call (it holds the rotations / transforms of this call). */
int chore_smpl_lbs_fwd(chore_handle* h, const void* arena, int V, int J, int num_betas, const float* pose,
                       const float* betas, const float* trans, const float* offsets, float scale, int B,
                       float* verts, float* joints, float* v_posed, float* naked, void* workspace,
                       chore_stream_t stream);
/* g_verts (B,V,3) / g_joints (B,J,3) upstream gradients (either may be NULL) ->
 * dpose (B,3J), dbetas (B,num_betas), dtrans (B,3) */
int chore_smpl_lbs_bwd(chore_handle* h, const void* arena, int V, int J, int num_betas, const float* pose,
                       float scale, int B, const float* v_posed, const float* g_verts, const float* g_joints,
                       float* dpose, float* dbetas, float* dtrans, void* workspace, chore_stream_t stream);
/* landmark regression on the skinned vertices (replaces the per-regressor torch.sparse.mm loop of
 * lib_smpl/torch_functions.py:52-76 behind SMPLPyTorchWrapperBatch.get_landmarks, lib_smpl/wrapper_pytorch.py:186-205):
 * reg (R,V) dense fp32 (rows of the body-25 / face / hand regressors stacked), verts (B,V,3) -> out (B,R,3); backward:
 * g (B,R,3) -> dverts (B,V,3), written (not accumulated); R <= 2048 */
int chore_landmarks_fwd(chore_handle* h, const float* reg, const float* verts, int R, int V, int B, float* out,
                        chore_stream_t stream);
int chore_landmarks_bwd(chore_handle* h, const float* reg, const float* g, int R, int V, int B, float* dverts,
                        chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Projection onto SO(3)  (replaces ReconFitterBase.project_so3, recon/recon_fit_base.py:168-188:
 * R = U diag(1,1,det(UV^T)) V^T) and its backward.  M, R, G, dM are (B,3,3) fp32 row-major;
 * `aux` (chore_so3_aux_bytes(B) bytes, may be NULL if no backward follows) carries U, V, sigma.
 * ------------------------------------------------------------------------------------------- */
size_t chore_so3_aux_bytes(int B);
int chore_so3_project_fwd(chore_handle* h, const float* M, int B, float* R, void* aux, chore_stream_t stream);
int chore_so3_project_bwd(chore_handle* h, const void* aux, const float* G, int B, float* dM,
                          chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Human-object contact term of the joint fitting phase  (replaces ReconFitterBase.compute_contact_loss,
 * recon/recon_fit_base.py:553-608, and the pytorch3d.loss.chamfer_distance call at :605-607 with its
 * defaults: squared-L2 nearest neighbour, mean over the points of a cloud, mean over clouds, both
 * directions added).  hum (B,Nh,3), obj (B,No,3); df_hum_o (B,Nh) = object distance field at the human
 * vertices, df_obj_h (B,No) = human distance field at the object points (contact = value < thres);
 * label_h (Nh) int32 part label of every human vertex; part_logits (B,P,No) object part logits (argmax
 * inside).  loss: 1 float on the device (0 when no (frame, part) pair has contact points on both sides).
 * `workspace` (chore_contact_workspace_bytes) carries the nearest-neighbour indices to the backward.
 * No host synchronisation: the step can be captured in a hipGraph.
 * ------------------------------------------------------------------------------------------- */
size_t chore_contact_workspace_bytes(int B, int Nh, int No, int P);
int chore_contact_fwd(chore_handle* h, const float* hum, const float* obj, const float* df_hum_o,
                      const float* df_obj_h, const int* label_h, const float* part_logits, int B, int Nh, int No,
                      int P, float thres, float* loss, void* workspace, chore_stream_t stream);
/* g_loss: 1 float on the device (upstream gradient) -> d_hum (B,Nh,3), d_obj (B,No,3); one of the two may be NULL (not
 * computed) */
int chore_contact_bwd(chore_handle* h, const float* hum, const float* obj, const int* label_h, int B, int Nh,
                      int No, int P, const float* g_loss, const void* workspace, float* d_hum, float* d_obj,
                      chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Differentiable silhouette rasterisation for the mask loss of the object fit  (replaces neural_renderer's
 * RasterizeFunction with return_alpha only, external/neural_renderer/neural_renderer/rasterize.py:14-170, i.e.
 * the CUDA kernels forward_face_index_map cuda/rasterize_cuda_kernel.cu:24-215 and backward_pixel_map
 * :290-549, as used by SilLossROI.forward recon/obj_pose_roi.py:159-172).
 * faces (B,F,3,3) fp32: projected triangle vertices [u, v in [-1,1], depth] (after fill_back, i.e. both windings);
 * face_index (B,size,size) int32 (-1 = background), alpha (B,size,size) fp32 in {0,1}; rows are NOT flipped
 * (the caller flips like rasterize_rgbad does, rasterize.py:345).  near/far/eps defaults of the reference:
 * 0.1, 100, 1e-4.  No face is dropped (the reference keeps at most 512 per 4x4-pixel block) and equal depths
 * resolve to the smaller face index.
 * ------------------------------------------------------------------------------------------- */
size_t chore_silhouette_workspace_bytes(int B, int F);
int chore_silhouette_fwd(chore_handle* h, const float* faces, int B, int F, int size, float near_z, float far_z,
                         int* face_index, float* alpha, void* workspace, chore_stream_t stream);
/* grad_alpha (B,size,size) -> grad_faces (B,F,3,3) (depth components are zero) */
int chore_silhouette_bwd(chore_handle* h, const float* faces, const int* face_index, const float* alpha,
                         const float* grad_alpha, int B, int F, int size, float eps, float* grad_faces,
                         chore_stream_t stream);
/* The rasteriser's triangle list from the object pose in one launch: SilLossROI.apply_transformation (recon/obj_pose_roi.py:
 * 159-162, w = s (v Ro + to)), neural_renderer's projection (external/neural_renderer/neural_renderer/projection.py:6-43:
 * c = Rc w + tc, divide by z + eps, distortion polynomial, K, [-1,1] with the vertical flip) and vertices_to_faces with both
 * windings (renderer.py:119-152, fill_back).  verts (B,V,3) template, faces (B,F,3) int32, obj_R (B,3,3) applied as v Ro,
 * obj_t (B,3), obj_s (B), K (B,3,3), cam_R / cam_t (B,3,3) / (B,3) or one for all frames (cam_broadcast != 0), dist5 HOST
 * {k1,k2,p1,p2,k3} or NULL -> tri (B,2F,3,3): face f and, at F + f, its reversed winding.
 * Backward: g_tri (B,2F,3,3) -> d_obj_R, d_obj_t, d_obj_s; adj_off (V+1) / adj: for every vertex the entries (f2 * 3 + corner)
 * of the doubled list that reference it, ascending (the order its corner gradients are added in; one topology for all frames). */
int chore_sil_project_fwd(chore_handle* h, const float* verts, const int* faces, const float* obj_R, const float* obj_t,
                          const float* obj_s, const float* K, const float* cam_R, const float* cam_t, int cam_broadcast,
                          const float* dist5, float orig_size, float eps, int B, int V, int F, float* tri,
                          chore_stream_t stream);
int chore_sil_project_bwd(chore_handle* h, const float* verts, const float* obj_R, const float* obj_t, const float* obj_s,
                          const float* K, const float* cam_R, const float* cam_t, int cam_broadcast, const float* dist5,
                          float orig_size, float eps, int B, int V, int F, const int* adj_off, const int* adj,
                          const float* g_tri, float* d_obj_R, float* d_obj_t, float* d_obj_s, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation metrics, fp64  (replace recon/eval/chamfer_distance.py:10-52 = sklearn kd-tree nearest neighbours, and
 * recon/eval/pose_utils.py compute_transform :145-180 / compute_similarity_transform :103-143).
 *   chore_eval_chamfer   out[0] = mean_i min_j |x_i - y_j| (direction 'x_to_y'), out[1] = the other direction
 *                        ('y_to_x'); 'bi' = out[0] + out[1].  Euclidean distances (not squared).  x (Nx,3), y (Ny,3) fp64.
 *   chore_eval_procrustes  params[13] = R (row-major), t, scale of the similarity that takes S1 closest to S2
 *                        (R = V Z U^T of K = X1 X2^T, det R = +1); chore_eval_apply_similarity: scale R p + t.
 * ------------------------------------------------------------------------------------------- */
size_t chore_eval_chamfer_workspace_bytes(int Nx, int Ny);
int chore_eval_chamfer(chore_handle* h, const double* x, int Nx, const double* y, int Ny, double* out, void* workspace,
                       chore_stream_t stream);
int chore_eval_procrustes(chore_handle* h, const double* S1, const double* S2, int N, double* params,
                          chore_stream_t stream);
int chore_eval_apply_similarity(chore_handle* h, const double* pts, int N, const double* params, double* out,
                                chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Device-resident bookkeeping of the surface-point generator  (replaces the boolean-mask indexing, Python lists and CPU
 * randint of Generator.gen_pc_batch / parse_preds, recon/generator.py:149-188, 190-217).
 *   chore_gen_compact   order[b][j] = index of the j-th set byte of mask[b][0..N), counts[b] = number of set bytes
 *                       (the order of x[mask]).
 *   chore_gen_append    dst[b][c][offsets[b] + j] = src[b][c][order[b][j]] for j < counts[b]; element strides given for
 *                       (b, c, n) of src and dst, so (B,N,3) and (B,C,N) tensors both fit; entries past `cap` dropped.
 *   chore_gen_advance   offsets[b] += counts[b]; *total += min_b counts[b]   (generator.py:156-158).
 *   chore_gen_resample  out (B,M,3): if counts[b] > 1, samples[b][order[b][floor(u * counts[b])]] + sigma * noise, else
 *                       init[b][floor(u * Ninit)] + 0.5 * noise (generator.py:163-177); u (B,M) uniform in [0,1), noise
 *                       (B,M,3) standard normal, both drawn by the caller on the device.
 * ------------------------------------------------------------------------------------------- */
int chore_gen_compact(chore_handle* h, const unsigned char* mask, int B, int N, int* order, int* counts,
                      chore_stream_t stream);
int chore_gen_append(chore_handle* h, const float* src, long long ss_b, long long ss_c, long long ss_n, const int* order,
                     const int* counts, const int* offsets, float* dst, long long ds_b, long long ds_c, long long ds_n,
                     int B, int C, int N, int cap, chore_stream_t stream);
int chore_gen_advance(chore_handle* h, const int* counts, int B, int* offsets, int* total, chore_stream_t stream);
int chore_gen_resample(chore_handle* h, const float* samples, int B, int N, const int* order, const int* counts,
                       const float* init, int Ninit, const float* u, const float* noise, int M, float sigma, float* out,
                       chore_stream_t stream);
/* one projection step of Alg. 1 (Generator.approx_surface, recon/generator.py:50-79) around chore_query_fwd /
 * chore_query_bwd_points: chore_gen_clamp_mask writes the upstream gradient of sum(clamp(df[:, k], max = thr)) -- g (B,2,N):
 * 1 where df[b,k,n] <= thr, else 0, the other channel 0 --, chore_gen_surface_step moves the points:
 * out = points - normalize(grad) * min(df[:, k], thr), normalize as F.normalize (v / max(||v||, 1e-12)) */
int chore_gen_clamp_mask(chore_handle* h, const float* df, int k, float thr, int B, int N, float* g, chore_stream_t stream);
int chore_gen_surface_step(chore_handle* h, const float* points, const float* grad, const float* df, int k, float thr, int B,
                           int N, float* out, chore_stream_t stream);
/* the four launches above (chore_query_fwd -> chore_gen_clamp_mask -> chore_query_bwd_points -> chore_gen_surface_step) as
 * one, with the same result bit for bit: arguments as chore_query_fwd; only with the fp16 x 3 heads (dtype CHORE_F16X3, or
 * the maps' type | CHORE_HEADS_X3).  out_points (B,N,3) must not alias points. */
int chore_gen_surface_step_fused(chore_handle* h, const float* points, const float* crop_center, int B, int N,
                                 const void* feat, int FH, int FW, const void* tmpx, int TH, int TW, int dtype,
                                 const void* heads_arena, const float* cam6_host, int k, float thr, float* out_points,
                                 chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Image preparation of the test loader on the device  (replaces, from the decoded uint8 images on,
 * TestData.prepare_image_crop data/test_data.py:59-125 with use_mean_center=False = BaseDataset.masks2bbox
 * data/base_data.py:92-112 + cv2.resize + BaseDataset.crop :131-162 + BaseDataset.resize :164-176 + compose_images
 * :178-192).  Integer arithmetic; cv2's 8-bit INTER_LINEAR algorithm is restated (oracle/image_prep.py): PARITY
 * UNPINNED at cv2, which the reference neither vendors nor pins.
 *   chore_prep_masks2bbox    bbox4 = {xmin, ymin, xmax + 1, ymax + 1} (int32, device) of the pixels where the uint8
 *                            wrap-around sum mask0 + mask1 exceeds thres; {50000, 50000, -100, -100} when there is none.
 *                            mask1 may be NULL.
 *   chore_prep_resize_u8     (sh, sw, C) uint8 -> (dh, dw, C) uint8, C <= 4.
 *   chore_prep_crop_compose  crop of the (H, W) images with corners tl = round(center - size / 2), br = round(center +
 *                            size / 2) (zero padded, base_data.py's clipping), resized to S x S, / 255, RGB zeroed where
 *                            neither mask exceeds 0.5, stacked as images (5, S, S) fp32 = [R, G, B, person, object].
 *                            rgb (H, W, 3), masks (H, W) uint8.
 * ------------------------------------------------------------------------------------------- */
int chore_prep_masks2bbox(chore_handle* h, const unsigned char* mask0, const unsigned char* mask1, int H, int W, int thres,
                          int* bbox4, chore_stream_t stream);
int chore_prep_resize_u8(chore_handle* h, const unsigned char* src, int sh, int sw, int C, unsigned char* dst, int dh, int dw,
                         chore_stream_t stream);
int chore_prep_crop_compose(chore_handle* h, const unsigned char* rgb, const unsigned char* person_mask,
                            const unsigned char* obj_mask, int H, int W, int tl_x, int tl_y, int br_x, int br_y, int S,
                            float* images, chore_stream_t stream);
/* use_mean_center=True (the COCO loader, recon/recon_fit_coco.py:28; data/test_data.py:127-160): the images are moved so
 * that the crop centre (cc, in the 2048-px space) lands on the mean crop centre of the BEHAVE training set (pad_image's
 * zero float64 canvas, at least 2048 x 1536, the pasted part clipped to that rectangle), the crop with corners tl / br is
 * taken around the mean centre, and the resize is cv2's generic float path (float weights, double sums) -- restated in
 * oracle/image_prep.py, UNPINNED like the 8-bit one. */
int chore_prep_crop_compose_mean(chore_handle* h, const unsigned char* rgb, const unsigned char* person_mask,
                                 const unsigned char* obj_mask, int H, int W, double cc_x, double cc_y, double mean_x, double mean_y,
                                 int tl_x, int tl_y, int br_x, int br_y, int S, float* images, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Interpenetration term of the joint fit  (replaces ReconFitterBase.smpl_obj_collision recon/recon_fit_base.py:610-624 =
 * mesh_intersection.BVH(max_collisions=8) + DistanceFieldPenetrationLoss(sigma=0.5, point2plane=False), constructed at
 * :78-86).  PARITY UNPINNED: that package (github.com/vchoutas/torch-mesh-isect, no revision pinned) is not in the
 * reference tree; oracle/collision.py states the published method implemented here.
 *   verts  (B,V,3) fp32: the concatenated mesh (SMPL vertices, then object vertices), faces (F,3) int32 into it.
 *   loss   (B) per-batch-element sums over all intersecting, non-adjacent triangle pairs (the caller takes the mean,
 *          like torch.mean(self.pen_distance(...)) :623);  gverts (B,V,3) = d loss[b] / d verts[b], kept for backward;
 *   counts (B+2 int32, or NULL): pairs found per batch element, then candidates / pairs dropped for lack of list space.
 * chore_collision_bwd: dverts = gout[b] * gverts.  No host synchronisation, fixed launch grids (hipGraph-capturable),
 * fixed-point accumulation (bit-reproducible).
 * ------------------------------------------------------------------------------------------- */
size_t chore_collision_workspace_bytes(int B, int V, int F);
int chore_collision_fwd(chore_handle* h, const float* verts, const int* faces, int B, int V, int F, float* loss,
                        float* gverts, int* counts, void* workspace, chore_stream_t stream);
int chore_collision_bwd(chore_handle* h, const float* gverts, const float* gout, int B, int V, float* dverts,
                        chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder layers as differentiable operators (training path).  Replace what torch autograd runs for
 * the reference's encoder modules: nn.Conv2d 3x3 / 1x1 (model/net_util.py:346-349, 356-372,
 * model/HGFilters.py:88-116), nn.GroupNorm(32, C) + F.relu (model/net_util.py:374-396, HGFilters.py:153-176),
 * F.interpolate(scale 2, bicubic, align_corners=True) + add (HGFilters.py:47-50), and their backward passes.
 * The host (chore_amd/ops.py, model/hgfilter_train.py) wires them into torch.autograd.Function nodes.
 *
 * Activations: NHWC (B,H,W,C), `dtype` = CHORE_F32, CHORE_BF16 or -- round 5 -- CHORE_F16X3 (fp32 tensors; every convolution,
 * data gradient and weight gradient on the fp16 matrix cores with hi / lo split operands, fp32-grade: the mode that trains at the
 * reference's precision, trainer/trainer.py:76-85, at matrix-core speed).  Parameters and parameter gradients: fp32
 * in the reference layouts (weight (Cout,Cin,k,k), bias/gamma/beta (C)).  taps = 1 (1x1) or 9 (3x3, pad 1).
 * C, Cin, Cout multiples of 32.  `stats`: chore_gn_stats_bytes(B) bytes, the exact per-(image, group)
 * sum / sum of squares of x, filled by chore_gn_stats; wherever an operator takes (stats, gamma, beta) != NULL
 * it sees relu(groupnorm(x)) instead of x, applied while the tile is staged (the normalised tensor is never
 * written).  Workspaces are caller-owned, sized by the *_bytes functions, and may be reused between calls on
 * one stream.  All reductions are order-fixed (integer accumulators or ordered partial sums): results are
 * bit-reproducible run to run.
 * ------------------------------------------------------------------------------------------- */
size_t chore_conv2d_workspace_bytes(int dtype, int taps, int Cin, int Cout);
size_t chore_gn_stats_bytes(int B);
/* CHORE_F16X3 training: a gradient that feeds a data- or weight-gradient GEMM has any magnitude, while an fp16 hi / lo pair keeps
 * 22 bits only for |x| in 2^-3 .. 2^16; the kernels therefore scale it by a power of two derived from max |x|, which travels
 * with the tensor: chore_amax_bytes() bytes of partial maxima (float bits), written by chore_absmax_f32 (n a multiple of 4,
 * x 16-byte aligned) -- inside chore_convblock_bwd by the kernels that produce the gradients. */
size_t chore_amax_bytes(void);
int chore_absmax_f32(chore_handle* h, const float* x, size_t n, void* amax, chore_stream_t stream);
/* zeroed != 0: `stats` already holds zeros (a slice of an arena cleared once per pass) */
int chore_gn_stats(chore_handle* h, int dtype, const void* x, int B, int HW, int C, void* stats, int zeroed,
                   chore_stream_t stream);
/* y = relu(groupnorm(x)) */
int chore_gn_relu_fwd(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma,
                      const float* beta, void* y, int B, int HW, int C, chore_stream_t stream);
/* y (B,H,W,Cout) = conv(a) + bias (bias may be NULL); out_stats (or NULL): ZEROED statistics accumulators filled with
 * the statistics of y by the convolution's epilogue (for a GroupNorm(32, Cout) that follows) */
int chore_conv2d_fwd(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                     const void* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                     int Cout, void* y, void* out_stats, void* workspace, chore_stream_t stream);
/* dx (B,H,W,Cin) = gradient w.r.t. the tensor the convolution saw (a).  dy_amax: chore_absmax_f32 of dy, required with
 * CHORE_F16X3 (NULL otherwise); the same for chore_conv2d_bwd_weight */
int chore_conv2d_bwd_data(chore_handle* h, int dtype, int taps, const void* dy, int B, int H, int W, int Cout,
                          const float* w, int Cin, void* dx, void* workspace, const void* dy_amax, chore_stream_t stream);
/* dw (Cout,Cin,k,k), dbias (Cout, or NULL) */
size_t chore_conv2d_wgrad_workspace_bytes(int taps, int B, int H, int W, int Cin, int Cout);
int chore_conv2d_bwd_weight(chore_handle* h, int dtype, int taps, const void* x, int B, int H, int W, int Cin,
                            const void* stats, const float* gamma, const float* beta, const void* dy, int Cout,
                            float* dw, float* dbias, void* workspace, const void* dy_amax, chore_stream_t stream);
/* da = gradient w.r.t. relu(groupnorm(x))  ->  dx, dgamma (C), dbeta (C) */
size_t chore_gn_relu_bwd_workspace_bytes(int B, int C);
int chore_gn_relu_bwd(chore_handle* h, int dtype, const void* x, const void* stats, const float* gamma,
                      const float* beta, const void* da, int B, int HW, int C, void* dx, float* dgamma,
                      float* dbeta, void* workspace, int workspace_zeroed, chore_stream_t stream);
/* One ConvBlock (reference model/net_util.py:346-396: three GroupNorm -> ReLU -> conv3x3 stages, their concat, and the
 * identity or GroupNorm -> ReLU -> conv1x1 residual) as a training operator, forward and backward in one call each: the
 * convolutions write their slices of y with the residual added in the epilogue and produce the GroupNorm statistics of
 * o1, o2 and y there; the backward reads dy in channel-strided slices and sums the skip gradients inside the GroupNorm
 * backward.  dtype: CHORE_F32 | CHORE_BF16 | CHORE_F16X3; Cout in {128, 256}, Cin a multiple of 32 (<= 256); conv weights in the
 * reference layout (O,C,kh,kw) fp32, no biases; wd and gb[6], gb[7] only when Cin != Cout.
 *   gb      host array of 8 device pointers: gamma, beta of bn1, bn2, bn3, bn4
 *   x_stats chore_gn_stats_bytes(B) statistics of x (the previous block's: `saved` + chore_convblock_out_stats_offset(B)),
 *           or NULL: computed by the forward (and kept in `saved` for the backward)
 *   saved   chore_convblock_saved_bytes: o1, o2 and all statistics, forward -> backward
 *   grads   chore_convblock_grad_floats: dW1 dW2 dW3 [dWd] then (dgamma, dbeta) of bn1, bn2, bn3 [, bn4] */
size_t chore_convblock_saved_bytes(int dtype, int B, int H, int W, int Cin, int Cout);
size_t chore_convblock_out_stats_offset(int B);
size_t chore_convblock_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout);
size_t chore_convblock_grad_floats(int Cin, int Cout);
int chore_convblock_fwd(chore_handle* h, int dtype, const void* x, const void* x_stats, int B, int H, int W, int Cin,
                        int Cout, const float* w1, const float* w2, const float* w3, const float* wd,
                        const float* const* gb, void* y, void* saved, void* workspace, chore_stream_t stream);
int chore_convblock_bwd(chore_handle* h, int dtype, const void* x, const void* x_stats, const void* dy, int B, int H,
                        int W, int Cin, int Cout, const float* w1, const float* w2, const float* w3, const float* wd,
                        const float* const* gb, const void* saved, void* dx, float* grads, void* workspace,
                        chore_stream_t stream);
/* The training loss of one stack (reference model/chore.py:192-237, CHORE.get_errors) and its gradients with respect to
 * the four predictions, one pass: clamped-L1 of the two distance fields, cross entropy of the part logits, masked MSE of
 * the PCA axes and the two centre predictions.  losses: 7 device floats -- the six terms in the reference's order (h, o,
 * parts, pca, smpl, obj) and their sum, each times `scale`; accumulate != 0 adds to them.  weights: host array, the
 * reference's self.loss_weights.  g_*: gradients of the added sum.  Exact (order-independent) reductions. */
size_t chore_train_loss_workspace_bytes(void);
int chore_train_loss(chore_handle* h, const float* df, const float* pca, const float* parts, const float* centers,
                     const float* df_h, const float* df_o, const int64_t* parts_gt, const float* pca_gt,
                     const float* body_center, const float* obj_center, int B, int N, float max_dist,
                     const float* weights, float scale, float* g_df, float* g_pca, float* g_parts, float* g_centers,
                     float* losses, int accumulate, void* workspace, chore_stream_t stream);
/* y (B,2H,2W,C) = a + bicubic_up2(low (B,H,W,C));  d_low = transpose of the upsampling applied to dy */
/* out_stats (both operators; or NULL): ZEROED chore_gn_stats_bytes(B) accumulators that receive the GroupNorm statistics
 * of y from the same pass (what a ConvBlock consuming y takes as x_stats) */
int chore_upadd_fwd(chore_handle* h, int dtype, const void* a, const void* low, void* y, int B, int H, int W, int C,
                    void* out_stats, chore_stream_t stream);
int chore_up2_bwd(chore_handle* h, int dtype, const void* dy, void* dlow, int B, int H, int W, int C,
                  chore_stream_t stream);
/* y (B,H/2,W/2,C) = 2x2 average pooling of x (B,H,W,C) (nn.AvgPool2d / F.avg_pool2d(2, stride 2), HGFilters.py:33,153),
 * C in {64,128,256}; dx (B,H,W,C) = its transpose applied to dy */
int chore_avgpool2_fwd(chore_handle* h, int dtype, const void* x, void* y, int B, int H, int W, int C, void* out_stats,
                       chore_stream_t stream);
int chore_avgpool2_bwd(chore_handle* h, int dtype, const void* dy, void* dx, int B, int H, int W, int C,
                       chore_stream_t stream);
/* stem: y (B,H/2,W/2,64) = conv 7x7 stride 2 pad 3 of images (B,Cin,H,W) fp32 NCHW, + bias (model/HGFilters.py:102,149;
 * Cin <= 8 forward, <= 5 for the weight gradient; the images take no gradient).  dy has y's layout and dtype. */
size_t chore_stem_workspace_bytes(int Cin);
int chore_stem_fwd(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const float* w,
                   const float* bias, void* y, void* workspace, chore_stream_t stream);
size_t chore_stem_wgrad_workspace_bytes(int B, int Cin, int H, int W);
int chore_stem_bwd_weight(chore_handle* h, int dtype, const float* images, int B, int Cin, int H, int W, const void* dy,
                          float* dw, float* dbias, void* workspace, chore_stream_t stream);
/* C (M,N) = A^T B, fp32 (exact fp32 matrix-core arithmetic), A (P,M) row stride lda, B (P,N) row stride ldb;
 * M, N multiples of 32, any P.  The weight gradients of the MLP heads (what autograd computes for the
 * nn.Conv1d layers of model/net_util.py:218-262). */
size_t chore_gemm_tn_workspace_bytes(int P, int M, int N);
int chore_gemm_tn_f32(chore_handle* h, const float* A, int lda, const float* B, int ldb, int P, int M, int N,
                      float* C, void* workspace, chore_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline): while enabled, chore_encode_fwd brackets every kernel launch
 * with hipEvents on the caller's stream, synchronises at the end of the call and accumulates, per
 * kernel class, the elapsed milliseconds, the algorithmic FLOPs and bytes and the launch count.
 * Enabling/disabling resets the counters.  Not for the timed region (it serialises the host).
 * chore_profile_read fills HOST arrays and returns the number of classes written.
 * ------------------------------------------------------------------------------------------- */
int chore_profile_enable(chore_handle* h, int on);
int chore_profile_read(chore_handle* h, int max_classes, const char** names, double* ms, double* flops,
                       double* bytes, int64_t* launches);

/* The tail of one inner fitting step (reference recon/recon_fit_behave.py:143-160, 270-287): Adam -- the formulas of
 * torch.optim.Adam(capturable=True) -- on nt <= 16 small fp32 tensors in one launch, and the early-stop rule
 * |prev - loss| / prev < prev * tol with its latch in another.  p / g / m / v: host arrays of device pointers, n: lengths;
 * step, prev, loss, loss_out: device floats; stop, armed: device bytes.  A set `stop` freezes the parameters (the moments and
 * the counter still advance).  chore_fit_stop_rule increments `step` (if not NULL): call it after chore_fit_adam_step. */
int chore_fit_adam_step(chore_handle* h, float* const* p, const float* const* g, float* const* m, float* const* v,
                        const int* n, int nt, const float* step, float lr, float beta1, float beta2, float eps,
                        const uint8_t* stop, chore_stream_t stream);
/* the same update with autograd's gradient accumulation folded in: g[k] is the ACCUMULATED gradient (the parameter's .grad,
 * read and written), gnew[k] (or NULL: nothing new) this step's fresh gradient, added to g[k] first -- the reference
 * accumulates over the inner steps of an outer iteration (recon_fit_behave.py:117-118,143-146), which as tensor ops is one
 * `grad += new` launch per parameter and step.  p[k] == NULL (then m[k], v[k] are ignored): a leaf that only accumulates
 * (stepped by a later phase's optimiser, which starts from these sums).  cols / pstride (host int arrays, or NULL = dense):
 * parameter k is a column slice of a wider tensor -- rows of cols[k] elements, pstride[k] apart (the split SMPL parameters of a
 * multi-frame batch, lib_smpl/wrapper_pytorch.py's SMPLPyTorchWrapperBatchSplitParams); g, gnew, m, v stay dense.  nt <= 16. */
int chore_fit_adam_step_acc(chore_handle* h, float* const* p, float* const* g, const float* const* gnew, float* const* m,
                            float* const* v, const int* n, const int* cols, const int* pstride, int nt, const float* step,
                            float lr, float beta1, float beta2, float eps, const uint8_t* stop, int step_counted,
                            chore_stream_t stream);   /* step_counted != 0: `step` already counts this step (see
                                                         chore_fit_weighted_sum_step) */
int chore_fit_stop_rule(chore_handle* h, const float* loss, float* prev, uint8_t* stop, const uint8_t* armed, float tol,
                        float* loss_out, float* step, chore_stream_t stream);
/* the weighting of the fit's loss dictionary (recon_fit_behave.py:339-358): out = sum_k coeff[k] * loss[k] / denom over
 * n <= 16 device scalars (losses: host array of device pointers; coeffs: host floats), and grads[k] = g * coeff[k] / denom */
int chore_fit_weighted_sum(chore_handle* h, const float* const* losses, const float* coeffs, int n, const float* denom,
                           float* out, chore_stream_t stream);
int chore_fit_weighted_sum_bwd(chore_handle* h, const float* coeffs, int n, const float* denom, const float* g, float* grads,
                               chore_stream_t stream);
/* The three single-thread launches of a step as one: chore_fit_weighted_sum, its backward for the upstream gradient *seed
 * (the step's d loss / d loss), and -- when prev != NULL -- chore_fit_stop_rule on the sum (recon_fit_behave.py:143-160,
 * 270-287).  The rule runs BEFORE the step's Adam launch here: `frozen` receives the stop flag as earlier steps left it
 * (hand THAT to chore_fit_adam_step_acc as `stop`: the reference applies the update of the step that meets the rule), and
 * `step` is advanced here (step_counted = 1 for the Adam launch). */
int chore_fit_weighted_sum_step(chore_handle* h, const float* const* losses, const float* coeffs, int n, const float* denom,
                                float* out, const float* seed, float* grads, float* prev, uint8_t* stop, uint8_t* frozen,
                                const uint8_t* armed, float tol, float* loss_out, float* step, chore_stream_t stream);

/* The small loss terms of forward_smpl (recon_fit_behave.py:293-337) in one launch each way.  pose (B,156) SMPL-H
 * axis-angle, pose_init (B,69) = the initial pose[3:72], J (B,R,3) landmarks whose first 25 rows are the body-25 keypoints
 * (row 8 = MidHip), kpts (B,25,3) = (x, y, confidence) in pixels of the network input or NULL (no keypoint term),
 * crop_center (B,2); body prior mean (63) / precision (63,63) applied to pose[3:66] (th_smpl_prior.py:32-39), hand prior
 * mean (90) and the two (45,45) precisions applied to pose[66:156] (th_hand_prior.py:63-72, incl. its sum over frames and
 * hands / 45); cam8 = HOST floats {fx, fy, cx, cy in pixels, crop/2, crop, network input size, z_0}.
 * out5 / up5: HOST arrays of 5 device scalars in the order pose prior, hand prior, pose-init, depth (smplz), keypoints (j2d);
 * a NULL entry of up5 is a zero gradient.  dpose (B,156) and dJ (B,R,3) are written completely. */
int chore_fit_smpl_terms_fwd(chore_handle* h, const float* pose, const float* pose_init, const float* J, const float* kpts,
                             const float* crop_center, const float* body_mean, const float* body_prec, const float* hand_mean,
                             const float* lhand_prec, const float* rhand_prec, int B, int P, int R, const float* cam8,
                             float* const* out5, chore_stream_t stream);
int chore_fit_smpl_terms_bwd(chore_handle* h, const float* pose, const float* pose_init, const float* J, const float* kpts,
                             const float* crop_center, const float* body_mean, const float* body_prec, const float* hand_mean,
                             const float* lhand_prec, const float* rhand_prec, int B, int P, int R, const float* cam8,
                             const float* const* up5, float* dpose, float* dJ, chore_stream_t stream);
/* Per-point terms: out_clamped_mean = mean over (B,N) of min(df[:, channel, :], clamp_max)  (df_h with channel 0 / 0.1,
 * recon_fit_base.py:520-526; the object term with channel 1 / 0.8, :505-511) and, if logits (B,C,N) is not NULL,
 * out_cross_entropy = mean over B of sum over N of the cross entropy against labels (B,N) int64 (recon_fit_behave.py:318-320),
 * C <= 16.  workspace: chore_fit_point_terms_workspace_bytes(B, N).  Backward: up_* device scalars (NULL = 0) -> ddf (B,2,N)
 * (the other channel zero) and dlogits (B,C,N). */
size_t chore_fit_point_terms_workspace_bytes(int B, int N);
int chore_fit_point_terms_fwd(chore_handle* h, const float* df, int channel, float clamp_max, const float* logits,
                              const int64_t* labels, int B, int N, int C, float* out_clamped_mean, float* out_cross_entropy,
                              void* workspace, chore_stream_t stream);
int chore_fit_point_terms_bwd(chore_handle* h, const float* df, int channel, float clamp_max, const float* logits,
                              const int64_t* labels, int B, int N, int C, const float* up_clamped_mean,
                              const float* up_cross_entropy, float* ddf, float* dlogits, chore_stream_t stream);
/* Object side of forward_step (recon_fit_behave.py:165-186).  chore_fit_obj_transform: out (B,N,3) = (verts (B,N,3) R (B,3,3)
 * + t (B,3)) * s (B)  (transform_obj_verts, recon_fit_base.py:367-371); backward: g (B,N,3) -> dR, dt, ds (no gradient for
 * the template points).  chore_fit_obj_terms: out_scale = mean_b (s - scale0)^2 and out_ocent = mean_b sum_k (mean_n
 * object[b,n,k] - (smpl_center[b,k] + mean_n centers[b,3+k,n]))^2 with centers (B,6,N); `diff` (B,3) carries the bracket to
 * the backward, which writes dobject (B,N,3), dcenters (B,6,N) and dscale (B) completely (up_*: device scalars, NULL = 0). */
int chore_fit_obj_transform_fwd(chore_handle* h, const float* verts, const float* R, const float* t, const float* s, int B, int N,
                                float* out, chore_stream_t stream);
int chore_fit_obj_transform_bwd(chore_handle* h, const float* verts, const float* R, const float* t, const float* s, const float* g,
                                int B, int N, float* dR, float* dt, float* ds, chore_stream_t stream);
size_t chore_fit_obj_terms_workspace_bytes(int B);
int chore_fit_obj_terms_fwd(chore_handle* h, const float* object, const float* centers, const float* smpl_center, const float* obj_s,
                            float scale0, int B, int N, float* diff, float* out_scale, float* out_ocent, void* workspace,
                            chore_stream_t stream);
int chore_fit_obj_terms_bwd(chore_handle* h, const float* diff, const float* obj_s, float scale0, const float* up_scale,
                            const float* up_ocent, int B, int N, float* dobject, float* dcenters, float* dscale,
                            chore_stream_t stream);
/* the perturbation of the raw rotation parameter of a step (ReconFitterBase.decopose_axis, recon/recon_fit_base.py:374-384:
 * rot + 1e-4 * U[0,1)), with the draws of all steps laid out up front: out (B,3,3) = rot + scale * noise[*k] with noise
 * (steps,B,3,3), then *k += 1 (k: device int64, the step counter a recorded graph advances; clamped into [0, steps)) */
int chore_fit_rot_noise(chore_handle* h, const float* rot, const float* noise, int64_t* k, float scale, int B, int64_t steps,
                        float* out, chore_stream_t stream);

/* debug aid: with CHORE_NAN_CHECK=1 in the environment chore_query_fwd / chore_query_bwd_points scan their inputs and
 * outputs for non-finite values (extra launches on the caller's stream); out32[0..15] = counts per site (0 points, 1-4 the
 * forward's df / pca / parts / centers, 5 / 6 the maps, 8 points, 9-12 the upstream gradients, 13 dpoints), out32[16..31] =
 * sequence number of the first scan of that site that saw one */
int chore_debug_nan_counts(unsigned* out32);

#ifdef __cplusplus
}
#endif
#endif /* CHORE_HIP_H */
