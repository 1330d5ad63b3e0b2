/*
 * gdm.h -- C ABI of libgdm_hip.so, the MI355X (gfx950) implementation of the geoMatch
 * dense-correspondence hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every `*_hip` entry point takes DEVICE pointers and a HIP stream (hipStream_t passed
 *     as void*; NULL = the null stream), enqueues its kernels on that stream and returns
 *     without synchronising.  All are re-entrant (no global mutable state), so they can be
 *     driven from several host threads on several streams, as evaluator.py:294-303 does.
 *   - return value: 0 on success, a positive hipError_t, or a negative GDM_E* code.
 *     gdm_last_error() returns a thread-local description of the last failure.
 *   - all tensors are dense, row-major, fp32 unless said otherwise; indices are int32.
 *   - the reference interface each entry point replaces is cited as file:line under
 *     /root/reference.
 */
#ifndef GDM_H
#define GDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDM_EINVAL (-1)   /* bad argument (shape, K, alignment) */
#define GDM_ENOMEM (-2)   /* workspace too small */

const char* gdm_last_error(void);
/* ABI version, bumped when a signature changes. */
int gdm_version(void);

/* ---------------------------------------------------------------------------------------
 * Exact K-nearest-neighbour search (fp32 squared L2, ascending; ties by ascending index).
 * ------------------------------------------------------------------------------------- */

/* Drop-in for the reference's only native entry point on the live path:
 *   void cpp_knn_batch_omp(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
 *                          const float* queries, size_t nqueries, size_t K, long* batch_indices)
 *   models/RandLA/utils/nearest_neighbors/knn_.h:17-19, knn_.cxx:104-135
 * Same argument list, HOST pointers, caller-allocated output, no return code (as the
 * reference).  Runs on the GPU: H2D copy, gdm_knn_batch_hip, D2H copy, widened to long.
 * dim must be 3.  On failure indices are left untouched and gdm_last_error() is set.   */
void gdm_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                   const float* queries, size_t nqueries, size_t K, long* batch_indices);

/* Device-resident form. support f32[B,S,3], query f32[B,Q,3] -> idx i32[B,Q,K],
 * d2 f32[B,Q,K] (may be NULL).  1 <= K <= 32.  Slots beyond S (K > S) hold index 0, as the
 * zero-initialised output of knn.pyx:93 does.                                           */
int gdm_knn_batch_hip(const float* support, const float* query, int B, int S, int Q, int K,
                      int32_t* idx, float* d2, void* stream);

/* One launch for a whole table of independent searches (the 22 calls per crop of
 * datasets/lm/linemod_pbr.py:534-569, for all crops of a batch).                        */
typedef struct gdm_knn_job {
    const float* support;     /* f32[B,S,3], batch item b at support + b*support_bstride */
    const float* query;       /* f32[B,Q,3], batch item b at query + b*query_bstride     */
    int32_t* idx;             /* i32[B,Q,K] dense */
    float* d2;                /* f32[B,Q,K] dense, or NULL */
    int64_t support_bstride;  /* in floats; S*3 when dense. A prefix slice cld[:, :S] of a  */
    int64_t query_bstride;    /* [B,N,3] array keeps bstride N*3 (linemod_pbr.py:538)     */
    int32_t S, Q, K;
    int32_t grid_w;           /* 0, or: the support is an ORGANISED map (e.g. the xyz of a depth crop) of S / grid_w rows x grid_w columns
                               * in row-major pixel order -- a hint that lets K > 1 searches with a workspace prune by pixel columns / rows;
                               * results are identical with and without it, for any data                                            */
} gdm_knn_job;
#define GDM_KNN_MAX_JOBS 32
int gdm_knn_jobs_hip(const gdm_knn_job* jobs /* host array */, int njobs, int B, void* stream);
/* Same, with a device workspace (16-byte aligned, >= gdm_knn_jobs_workspace_bytes) in which every distinct support set of the
 * K > 1 jobs is re-laid out once: small unorganised supports as hashed float4 tiles that the search blocks stream with coalesced
 * loads, unorganised supports of >= 1024 points as (x, y) cell lists (a counting sort per crop; a query then visits cells ring by
 * ring instead of every point), organised supports (grid_w) as per-column / per-row ratio ranges; without it (gdm_knn_jobs_hip)
 * each block gathers its tiles from the [S,3] array.  Results are identical in every form, for any data.                       */
size_t gdm_knn_jobs_workspace_bytes(const gdm_knn_job* jobs, int njobs, int B);
int gdm_knn_jobs_ws_hip(const gdm_knn_job* jobs, int njobs, int B, void* workspace, size_t workspace_bytes, void* stream);

/* Ball query (lib/pointops/functions/pointops.py:205-219 `ballquery_cuda(b,n,m,radius,nsample,
 * new_xyz,xyz,idx)`): for each centre the first `nsample` support indices (ascending index)
 * with d2 < radius^2; remaining slots repeat the first hit; all zero when none.
 * xyz f32[B,n,3], new_xyz f32[B,m,3] -> idx i32[B,m,nsample].                           */
int gdm_ballquery_hip(int B, int n, int m, float radius, int nsample,
                      const float* new_xyz, const float* xyz, int32_t* idx, void* stream);

/* Furthest point sampling (pointops.py:40-50 `furthestsampling_cuda(b,n,m,xyz,temp,idx)`):
 * starts from index 0, `temp` f32[B,n] is scratch (initialised inside).                  */
int gdm_furthestsampling_hip(int B, int n, int m, const float* xyz, float* temp, int32_t* idx, void* stream);

/* Three-point interpolation (pointops.py:114-144 `interpolation_forward/backward_cuda`): feat f32[b,c,m], idx i32[b,n,3],
 * weight f32[b,n,3] -> out f32[b,c,n]; backward scatters grad_out f32[b,c,n] into a ZERO-FILLED grad_feat f32[b,c,m].  */
int gdm_interpolation_forward_hip(int B, int c, int m, int n, const float* feat, const int32_t* idx, const float* weight, float* out, void* stream);
int gdm_interpolation_backward_hip(int B, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                                   float* grad_feat, void* stream);

/* Label histograms (pointops.py:289-338): label_stat i32[b,n,nclass] summed over the points inside the ball of each centre
 * (`labelstat_ballrange_cuda(b,n,m,radius,nclass,new_xyz,xyz,label_stat,new_label_stat)`) or over idx i32[b,m,nsample]
 * (`labelstat_idx_cuda(b,n,m,nsample,nclass,label_stat,idx,new_label_stat)`) -> i32[b,m,nclass].                        */
int gdm_labelstat_ballrange_hip(int B, int n, int m, float radius, int nclass, const float* new_xyz, const float* xyz,
                                const int32_t* label_stat, int32_t* new_label_stat, void* stream);
int gdm_labelstat_idx_hip(int B, int n, int m, int nsample, int nclass, const int32_t* label_stat, const int32_t* idx,
                          int32_t* new_label_stat, void* stream);

/* ---------------------------------------------------------------------------------------
 * Feature gather / scatter (channel-major features, as the reference network keeps them).
 * ------------------------------------------------------------------------------------- */

/* out[b,c,j,k] = feat[b,c,idx[b,j,k]]   feat f32[B,C,n], idx i32[B,m,K] -> f32[B,C,m,K]
 * replaces Building_block.gather_neighbour + permute (models/RandLA/RandLANet.py:729-738,
 * 704-716) and pointops grouping_forward_cuda (pointops.py:151-176).                     */
int gdm_group_gather_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m, int K,
                         float* out, void* stream);
/* grad_feat[b,c,idx[b,j,k]] += grad_out[b,c,j,k]; grad_feat must be zeroed by the caller. */
int gdm_group_gather_bwd_hip(const float* grad_out, const int32_t* idx, int B, int C, int n, int m, int K,
                             float* grad_feat, void* stream);
/* the same with grad_out read in place from a wider tensor: go_bstride = floats between two batch items (rows of m*K floats, channel
 * stride m*K; >= C*m*K) -- the gradient slice a torch.cat backward hands to nearest_interpolation (models/ffb6d.py:148-163) */
int gdm_group_gather_bwd2_hip(const float* grad_out, long go_bstride, const int32_t* idx, int B, int C, int n, int m, int K,
                              float* grad_feat, void* stream);

/* out[b,c,j] = max_k feat[b,c,idx[b,j,k]]; arg i32[B,C,m] (may be NULL) = winning source index.
 * replaces FFB6DEmb.random_sample (models/ffb6d.py:128-146).                             */
int gdm_gather_max_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m, int K,
                       float* out, int32_t* arg, void* stream);
/* grad_feat[b,c,arg[b,c,j]] += grad_out[b,c,j]; grad_feat zeroed by the caller.          */
int gdm_gather_max_bwd_hip(const float* grad_out, const int32_t* arg, int B, int C, int n, int m,
                           float* grad_feat, void* stream);

/* out[b,c,j] = feat[b,c,idx[b,j]]      (K == 1)
 * replaces FFB6DEmb.nearest_interpolation (models/ffb6d.py:148-163), the `choose` gather
 * (ffb6d.py:278-281) and pointops gathering_forward_cuda (pointops.py:61-82).
 * idx_bstride = elements between batch items of idx (0 broadcasts one index row).        */
int gdm_gather_nn_hip(const float* feat, const int32_t* idx, int B, int C, int n, int m,
                      float* out, void* stream);
int gdm_gather_nn_bwd_hip(const float* grad_out, const int32_t* idx, int B, int C, int n, int m,
                          float* grad_feat, void* stream);

/* Relative position encoding (RandLANet.py:720-727 + the permute of :701-702):
 * xyz f32[B,n,3], idx i32[B,n,K] -> out f32[B,10,n,K] with channels
 * [ |p_i-p_j|, p_i-p_j (3), p_i (3), p_j (3) ].                                          */
int gdm_rel_pos_enc_hip(const float* xyz, const int32_t* idx, int B, int n, int K, float* out, void* stream);

/* Attentive pooling core (RandLANet.py:749-752): softmax of `att` over K, weighted sum of
 * `feat` over K.  att, feat f32[B,C,n,K] -> out f32[B,C,n].                              */
int gdm_att_pool_hip(const float* att, const float* feat, int B, int C, int n, int K, float* out, void* stream);
int gdm_att_pool_bwd_hip(const float* att, const float* feat, const float* grad_out, int B, int C, int n, int K,
                         float* grad_att, float* grad_feat, void* stream);

/* ---------------------------------------------------------------------------------------
 * N x M descriptor matching (cosine similarity + row arg-max).
 * ------------------------------------------------------------------------------------- */

#define GDM_MATCH_BF16X3 0   /* split-bf16 MFMA: hi*hi + hi*lo + lo*hi, fp32 accumulate (|err| <= ~1.2e-5) */
#define GDM_MATCH_F32    1   /* f32-input MFMA, exact fp32 products                                        */

/* Workspace bytes needed by gdm_match_hip for the given sizes. */
size_t gdm_match_workspace_bytes(int B, int N, int M);

/* Fused inference matching (evaluator.py:87-93): L2-normalise each scene descriptor (over D),
 * each model descriptor (over D), similarity = scene . model, per scene point the maximum
 * over the M model vertices and its index (first maximum on exact ties).
 *   scene f32[B,D,N] channel-major (GeoMatch end_points['rgbd'], geoMatch.py:199)
 *   model f32[D,M]   channel-major (end_points['mesh'][0])
 *   -> best_idx i32[B,N], best_sim f32[B,N]; sim f32[B,N,M] is also written when non-NULL
 *      (the materialised matrix of evaluator.py:91).
 * D must be 128; B, N, M arbitrary (an (M+1)-column padded model, geoMatch.py:117-119, works).
 * = gdm_match_pack_hip(scene) + gdm_match_pack_hip(model) + gdm_match_packed_hip.            */
int gdm_match_hip(const float* scene, const float* model, int B, int D, int N, int M, int precision,
                  int32_t* best_idx, float* best_sim, float* sim,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The two stages separately, so that an object's model rows are packed once and reused
 * (they are constant in eval: models/SplineCNN.py:234 takes no input).
 * pack: x f32[R,D,n] channel-major -> R*n normalised rows of 512 bytes (gdm_match_rows_bytes(R*n)). */
size_t gdm_match_rows_bytes(int rows);
size_t gdm_match_partial_bytes(int B, int N);
int gdm_match_pack_hip(const float* x, int R, int D, int n, int precision, void* rows, void* stream);
/* two packs in one launch (the scene and the model descriptors of one step; evaluator.py:80-81 normalises both): the same bytes as
 * gdm_match_pack_hip(x1, R1, D, n1, ..., rows1) + gdm_match_pack_hip(x2, R2, D, n2, ..., rows2) */
int gdm_match_pack2_hip(const float* x1, int R1, int n1, void* rows1, const float* x2, int R2, int n2, void* rows2,
                        int D, int precision, void* stream);
int gdm_match_packed_hip(const void* scene_rows, const void* model_rows, int R /* B*N */, int M, int precision,
                         int32_t* best_idx, float* best_sim, float* sim,
                         void* partial, size_t partial_bytes, void* stream);

/* seg f32[B,2,N] -> mask u8[B,N] = (argmax over the 2 classes == 1) (evaluator.py:79-83),
 * count i32[B] of selected points (zeroed inside).                                        */
int gdm_seg_mask_hip(const float* seg, int B, int N, uint8_t* mask, int32_t* count, void* stream);

/* ---------------------------------------------------------------------------------------
 * SplineConv sparse part for the object-model branch (models/SplineCNN.py:136-140,234-239;
 * arithmetic of the un-vendored torch_spline_conv, dim=3, degree 1, open splines, mean aggr).
 * xw f32[M, ks^3, C] = X @ [W_0|...|W_{ks^3-1}] (dense GEMM done by the caller),
 * CSR edges sorted by target: rowptr i32[M+1], src i32[E], attr f32[E,3] in [0,1],
 * root f32[M,C] (x W_root) or NULL, bias f32[C] or NULL -> out f32[M,C] (optional ReLU).     */
int gdm_spline_aggregate_hip(const float* xw, const int32_t* rowptr, const int32_t* src, const float* attr,
                             const float* root, const float* bias, int M, int C, int kernel_size, int relu,
                             float* out, void* stream);
/* grad_xw[src, wi_s, :] += b_s * grad_out[target, :] / deg(target); grad_xw zeroed by the caller. */
int gdm_spline_aggregate_bwd_hip(const float* grad_out, const int32_t* rowptr, const int32_t* src, const float* attr,
                                 int M, int C, int kernel_size, float* grad_xw, void* stream);
/* The same layer for few input channels (Cin <= 16; the first mesh layer, 9 -> 128) without the [M, 125*C] table:
 * out_i = mean_e sum_s b_s (x_j . W[wi_s]) + x_i . W_root + bias; x f32[M,Cin], weight f32[ks^3,Cin,C] (SplineConv.weight as stored),
 * root_t f32[Cin,C] (lin.weight transposed, may be NULL), bias f32[C] (may be NULL). */
int gdm_spline_direct_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                          const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                          float* out, void* stream);
/* Edge-grouped form of the 128-channel layers: only the (source vertex, kernel index) pairs that some edge needs are multiplied,
 * 4x fewer FLOPs than the dense form and no [M, 125*C] table.  gdm_gemm_grouped_hip: Y[r, 0:128] = Wpk[tile_co0[r/256] + 0:128, :] .
 * X[rowidx[r], :] for R % 256 == 0 rows (pairs sorted by kernel index, groups padded to 256 rows with rowidx = 0), X packed by
 * gdm_conv3x3_pack_act_hip(x^T, 1, Cin, 1, M), weights by gdm_conv1x1_pack_weight_hip (125*C rows, row = wi*C + co).
 * gdm_spline_pairs_aggregate_hip: out_i = mean_e sum_s basis[e,s] * Y[pos[e,s]] + root_i + bias. */
int gdm_gemm_grouped_hip(const void* xpk, const void* wpk, const int32_t* rowidx, const int32_t* tile_co0, int R, int M,
                         int Cin, int Cout_total, float* out, void* stream);
int gdm_spline_pairs_aggregate_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                   const float* root, const float* bias, int M, int C, int relu, float* out, void* stream);
/* The two kernels above with an optional second output out_t f32[C, M] (channel-major: what the next layer's grouped GEMM, its root
 * product and the final linear read as gdm_pointwise_hip segments), out and out_t may each be NULL but not both; out_t needs
 * C % 4 == 0 and 512 % C == 0. */
int gdm_spline_direct2_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                           const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                           float* out, float* out_t, void* stream);
int gdm_spline_pairs_aggregate2_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                    const float* root, const float* bias, int M, int C, int relu, float* out, float* out_t, void* stream);
/* ... and with an optional THIRD output out_packed: the result as the split-bf16 operand planes of the next layer's grouped GEMM
 * (gdm_conv3x3_act_bytes(1, C, 1, M) bytes, the layout gdm_conv3x3_pack_act_hip(x_cm, 1, C, 1, M) writes; zero border in place, C = 128,
 * 256 or 512): the pack launch between two SplineConv layers (SplineCNN.py:238-239) is then not needed. */
int gdm_spline_direct3_hip(const float* x, const float* weight, const int32_t* rowptr, const int32_t* src, const float* attr,
                           const float* root_t, const float* bias, int M, int Cin, int C, int kernel_size, int relu,
                           float* out, float* out_t, void* out_packed, void* stream);
int gdm_spline_pairs_aggregate3_hip(const float* Y, const int32_t* rowptr, const int32_t* pos, const float* basis,
                                    const float* root, const float* bias, int M, int C, int relu, float* out, float* out_t,
                                    void* out_packed, void* stream);

/* ---------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=True, NCHW fp32 (models/cnn/pspnet.py:26-29,38).
 * in f32[planes,H,W] -> out f32[planes,OH,OW]; planes = B*C.  Backward writes every element of grad_in
 * (gather form: no atomics, deterministic).                                             */
int gdm_upsample_bilinear_hip(const float* in, long planes, int H, int W, int OH, int OW, float* out, void* stream);
int gdm_upsample_bilinear_bwd_hip(const float* grad_out, long planes, int H, int W, int OH, int OW, float* grad_in, void* stream);

/* ---------------------------------------------------------------------------------------
 * DGCNN variant (models/dgcnn.py, models/geoMatch_DGCNN.py).
 * top-k per row of a dense score matrix (dgcnn.py:21-27 `pairwise_distance.topk(k)`): score f32[rows,n]
 * -> idx i32[rows,K] (descending score, ties by ascending column), val f32[rows,K] or NULL.  K <= 32.
 * edge feature (dgcnn.py:30-56): x f32[B,C,n], idx i32[B,n,K] -> out f32[B,2C,n,K] = cat(x_j - x_i, x_i). */
int gdm_topk_rows_hip(const float* score, long rows, int n, int K, int32_t* idx, float* val, void* stream);
/* The same top-k over dgcnn.py:22-25's pairwise_distance without materialising it: gram f32[B,n,n] = X^T X, xx f32[B,n] = sum_c x^2;
 * ranks ((-xx[c]) - (-2*gram[r][c])) - xx[r] formed with torch's operations in torch's order (bit-identical indices). */
int gdm_topk_negdist_hip(const float* gram, const float* xx, int B, int n, int K, int32_t* idx, void* stream);
int gdm_edge_feature_hip(const float* x, const int32_t* idx, int B, int C, int n, int K, float* out, void* stream);
int gdm_edge_feature_bwd_hip(const float* grad_out, const int32_t* idx, int B, int C, int n, int K, float* grad_x, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused circle loss rows for the training matching (models/geoMatch.py:55-83 matching_loss,
 * models/loss.py:441-494 CircleLoss): positive mask evaluated on the fly, two masked LSEs online.
 * sim f32[R, M+1] (all selected points of the batch, concatenated), match i32[R] (ground-truth vertex, M = none),
 * item i32[R] (batch item of the row), xyz f32[M,3], vis u8[B,M], radius / gamma / m scalars
 * -> lse_p, lse_n, loss f32[R] (loss = softplus(lse_p + lse_n)).
 * Backward: grad_rows f32[R] -> dsim f32[R, M+1].                                             */
int gdm_circle_rows_fwd_hip(const float* sim, int R, int Mp, const int32_t* match, const int32_t* item,
                            const float* xyz, const uint8_t* vis, float radius, float gamma, float m,
                            float* lse_p, float* lse_n, float* loss, void* stream);
int gdm_circle_rows_bwd_hip(const float* sim, int R, int Mp, const int32_t* match, const int32_t* item,
                            const float* xyz, const uint8_t* vis, float radius, float gamma, float m,
                            const float* lse_p, const float* lse_n, const float* grad_rows, float* dsim, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training matching loss without the similarity matrix (gdm_circle.hip): replaces models/geoMatch.py:117-136 (normalise, padded
 * matmul), :55-83 / :86-100 (positive masks) and models/loss.py:441-494 (CircleLoss) and their autograd backward for all selected
 * points of a batch.  x f32[R,128] / y f32[M,128] are UNIT rows (F.normalize stays with the caller's autograd).
 *   pack      x -> split-bf16 rows + d-major tiles + row sums (sizes from the two *_bytes functions, rows padded to 128)
 *   nbr       bit table [M][ceil(M/32)] of vertices within `radius` of each vertex (basic_utils.py:86-89 arithmetic); once per model
 *   visbits   visible_flag u8[B,M] -> bits [B][ceil(M/32)]
 *   fwd       per-row lse_p, lse_n, loss f32[Rp]; g i32[Rp] = ground-truth vertex (M = none), item i32[Rp]; symmetric objects:
 *             g / c2 = the two positive columns of the row (geoMatch.py:91-95), nbr / visb unused
 *   bwd       coef f32[Rp] = upstream gradient x sigmoid(lse_p + lse_n) (0 for padding rows and empty positive sets) ->
 *             gx f32[Rp,128], gy_part f32[P][Mp,128] with P = gdm_circle_match_bwd_parts (sum over P = gradient w.r.t. y)    */
size_t gdm_circle_match_rows_bytes(int n);
size_t gdm_circle_match_tp_bytes(int n);
int gdm_circle_match_pack_hip(const float* x, int n, void* rows, void* tp, float* rowsum, void* stream);
int gdm_circle_match_nbr_hip(const float* xyz, int M, float radius, uint32_t* nbr, void* stream);
int gdm_circle_match_visbits_hip(const uint8_t* vis, int B, int M, uint32_t* bits, void* stream);
int gdm_circle_match_fwd_hip(const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp,
                             int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                             const uint32_t* nbr, const uint32_t* visb, float gamma, float m,
                             float* lse_p, float* lse_n, float* loss, void* stream);
/* The same two kernels for the geoMatch_DGCNN variant (/root/reference/models/geoMatch_DGCNN.py:52-135): nbr_per_item != 0 -- `nbr` holds one
 * neighbour table PER BATCH ITEM, u32[B][M][ceil(M/32)] made by gdm_circle_match_nbr_items_hip from a per-item, per-vertex radius
 * (positive_r / 1000 * z of the posed vertex, :66-67); pad_e0 != 0 -- the padding column is the unit vector e0 (:96-99) and `xpad`
 * holds x[r][0] instead of the row sums.  (0, 0) = gdm_circle_match_fwd_hip / _bwd_hip.                                            */
int gdm_circle_match_nbr_items_hip(const float* xyz, int M, const float* rad /* f32[B,M] */, int B, uint32_t* nbr, void* stream);
int gdm_circle_match_fwd2_hip(const void* xrows, const void* xtp, const float* xpad, const void* yrows, const void* ytp,
                              int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                              const uint32_t* nbr, int nbr_per_item, const uint32_t* visb, int pad_e0, float gamma, float m,
                              float* lse_p, float* lse_n, float* loss, void* stream);
int gdm_circle_match_bwd2_hip(const void* xrows, const void* xtp, const float* xpad, const void* yrows, const void* ytp,
                              int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                              const uint32_t* nbr, int nbr_per_item, const uint32_t* visb, int pad_e0, float gamma, float m,
                              const float* lse_p, const float* lse_n, const float* coef, float* gx, float* gy_part, void* stream);
int gdm_circle_match_bwd_parts(int R, int M);
int gdm_circle_match_bwd_hip(const void* xrows, const void* xtp, const float* xsum, const void* yrows, const void* ytp,
                             int R, int M, const int32_t* g, const int32_t* c2, const int32_t* item,
                             const uint32_t* nbr, const uint32_t* visb, float gamma, float m,
                             const float* lse_p, const float* lse_n, const float* coef, float* gx, float* gy_part, void* stream);

/* ---------------------------------------------------------------------------------------
 * One attentive-pooling stage of RandLA-Net's local feature aggregation in a single launch (inference):
 * models/RandLA/RandLANet.py:700-718 Building_block.forward = two such stages; :720-727 relative_pos_encoding, :729-738
 * gather_neighbour, :747-754 Att_pooling.forward.  For every point i with neighbours idx[b,i,0..15]:
 *   f_xyz  = lrelu(s1 * (W1 . pos_enc(i, k)) + b1)                      (10 -> D/2; pos_enc = [dist, rel(3), tile(3), neighbour(3)])
 *   f_xyz  = lrelu(s2 * (W2 . f_xyz) + b2)                              only when w2t != NULL (the block's second stage, mlp2)
 *   f_cat  = [feat[:, idx[b,i,k]] ; f_xyz]                              (D x 16)
 *   att    = Wf . f_cat ; score = softmax_k(att) ; agg = sum_k f_cat * score
 *   out[b,:,i] = lrelu(sm * (Wm . agg) + bm)                            (D -> OUT <= D)
 * xyz f32[B,n,3], idx i32[B,n,16], feat f32[B,D/2,n], out f32[B,OUT,n]; weights TRANSPOSED ([in][out], contiguous):
 * w1t [10,D/2], w2t [D/2,D/2], wft [D,D], wmt [D,OUT]; s*, b* = eval-mode BatchNorm folded to scale / shift.  D in {32,64,128,256}. */
int gdm_lfa_stage_hip(const float* xyz, const int32_t* idx, const float* feat, const float* w1t, const float* s1, const float* b1,
                      const float* w2t, const float* s2, const float* b2, const float* wft, const float* wmt, const float* sm,
                      const float* bm, int B, int n, int K, int D, int OUT, float slope, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pose solve statistics (evaluator.py:85-100 + utils/pvn3d_eval_utils_kpls.py:43-77 best_fit_transform).
 * Per crop, over the points with mask != 0: out[b] = { n, sum A (3), sum B (3), sum A_i B_j (9, row-major) } as f64,
 * A = model_xyz[best_idx] (f32[M,3]), B = scene point.  Scene xyz addressing: element (b, i, c) at
 * scene_xyz[b*scene_bstride + i*pt_stride + c*ch_stride] (so both [B,N,3] and the first rows of cld_rgb_nrm [B,9,N] work). */
int gdm_kabsch_stats_hip(const float* scene_xyz, long scene_bstride, int pt_stride, int ch_stride, const float* model_xyz,
                         const int32_t* best_idx, const uint8_t* mask, int B, int N, int M, double* out, void* stream);

/* The fit itself (pvn3d_eval_utils_kpls.py:55-77: centroids, SVD of H, reflection fix, t = cB - R cA) from those statistics,
 * on the device: RT f32[B,3,4] maps model coordinates (A) to the camera frame (B); valid u8[B] = n >= min_points, else the
 * reference's sentinel pose [I | (0,0,-1000)] (evaluator.py:94-96).  The optimal proper rotation is obtained as Horn's unit
 * quaternion (largest eigenvector of a symmetric 4x4, cyclic Jacobi, f64) -- the same R as SVD + reflection fix. */
int gdm_kabsch_solve_hip(const double* stats, int B, int min_points, float* RT, uint8_t* valid, void* stream);

/* Eval-mode BatchNorm + activation + max over the K neighbours (DGCNN edge convolutions, dgcnn.py:104-117) in one pass:
 * out[plane,i] = max_k act(scale[c]*x[plane,i,k] + shift[c]), c = plane % C; x f32[planes,n,K], K % 4 == 0, planes <= 65535. */
int gdm_affine_act_maxk_hip(const float* x, const float* scale, const float* shift, long planes, int C, long n, int K, int act,
                            float slope, float* out, void* stream);

/* Single-slope PReLU (models/cnn/pspnet.py:41) and its backward, for training: y = x > 0 ? x : a x; grad_x = x > 0 ? go : a go;
 * grad_slope[0] += sum over x <= 0 of x * go (zeroed by the caller).  slope is a DEVICE pointer to the one parameter; n % 4 == 0. */
int gdm_prelu1_hip(const float* x, const float* slope, long n, float* y, void* stream);
int gdm_prelu1_bwd_hip(const float* x, const float* grad_out, const float* slope, long n, float* grad_x, float* grad_slope, void* stream);

/* Inference-mode BatchNorm + activation (+ residual branch with its own folded BatchNorm) in one pass:
 * y = act(x*scale[c] + shift[c] (+ res*res_scale[c] + res_shift[c])), c = plane % C; x,res,y f32[planes, inner],
 * inner % 4 == 0; res / res_scale / res_shift may be NULL (res_scale NULL = plain residual add).
 * act: 0 none, 1 ReLU, 2 leaky ReLU / single-slope PReLU (slope).  May run in place (y == x).
 * Replaces the BN / ReLU / LeakyReLU / PReLU / add launches of pytorch_utils._ConvBase, extractors.BasicBlock
 * (extractors.py:36-58), PSPUpsample (pspnet.py:34-45) and Dilated_res_block (RandLANet.py:683-688) in eval mode. */
int gdm_affine_act_hip(const float* x, const float* scale, const float* shift, const float* res, const float* res_scale,
                       const float* res_shift, long planes, int C, long inner, int act, float slope, float* y, void* stream);

/* conv3x3(pad 1) after bilinear upsample (align_corners), from low-resolution channel mixes (pspnet.py:34-45):
 * z f32[B, 9*Cout, H, W] = conv1x1(x, W rearranged tap-major) at LOW resolution; this gathers, per output pixel,
 * the 9 bilinear taps (zero outside the OHxOW map), applies scale/shift (folded BN incl. the conv bias) and the
 * activation (0 none, 1 ReLU, 2 leaky/PReLU slope) -> out f32[B, Cout, OH, OW].                              */
int gdm_upconv3x3_gather_hip(const float* z, const float* scale, const float* shift, int B, int Cout, int H, int W,
                             int OH, int OW, int act, float slope, float* out, void* stream);
/* The same with eight output channels per workgroup, also writing the result as the packed split-bf16 operand (gdm_conv3x3_act_bytes
 * (B, Cout, OH, OW) bytes, zero border kept by the caller) of the next GEMM over the map: Cout = 64 or a multiple of 128, x2 stages. */
int gdm_upconv3x3_gather2_hip(const float* z, const float* scale, const float* shift, int B, int Cout, int H, int W,
                              int OH, int OW, int act, float slope, float* out, void* outpk, void* stream);
/* Its transpose for training (scale = 1, act = none): grad_z f32[B,9*Cout,H,W] from grad_out f32[B,Cout,OH,OW]; every element of
 * grad_z is written (gather form, no atomics).  B*9*Cout <= 65535. */
int gdm_upconv3x3_gather_bwd_hip(const float* grad_out, int B, int Cout, int H, int W, int OH, int OW, float* grad_z, void* stream);
/* The whole PSPUpsample(64 -> 64) in one kernel (the last up stage, where the 9*64-channel low-resolution tensor of the two-kernel
 * form is 604 MB at batch 16): x f32[B,64,H,W] -> out f32[B,64,OH,OW] = act(scale * conv3x3(upsample(x)) + shift), weights packed by
 * gdm_upconv_fused64_pack_weight_hip from the f32[64,64,3,3] convolution weight (split-bf16 products, fp32 accumulate). */
size_t gdm_upconv_fused64_weight_bytes(void);
int gdm_upconv_fused64_pack_weight_hip(const float* w, void* wpk, void* stream);
int gdm_upconv_fused64_hip(const float* x, const void* wpk, const float* scale, const float* shift, int B, int C, int H, int W,
                           int OH, int OW, int act, float slope, float* out, void* stream);

/* Pyramid-pooling bottleneck tail (pspnet.py:24-31) after splitting the 1x1 convolution over the concat:
 * out = relu(g + bias[c] + sum_k bilinear_align_corners(y_k)), g f32[B,C,H,W] = W_f . feats, y_k f32[B,C,s_k,s_k] = W_k . prior_k. */
int gdm_psp_combine_hip(const float* g, const float* y1, int s1, const float* y2, int s2, const float* y3, int s3,
                        const float* y4, int s4, const float* bias, int B, int C, int H, int W, float* out, void* stream);
/* The same, also writing the result as the packed split-bf16 operand (gdm_conv3x3_pack_act_hip's layout, gdm_conv3x3_act_bytes bytes, zero
 * border kept by the caller) of the next GEMM over the map: C = 64 or a multiple of 128, W % 4 == 0.  outpk NULL = gdm_psp_combine_hip. */
int gdm_psp_combine2_hip(const float* g, const float* y1, int s1, const float* y2, int s2, const float* y3, int s3,
                         const float* y4, int s4, const float* bias, int B, int C, int H, int W, float* out, void* outpk, void* stream);
/* Point->pixel fusion tail (ffb6d.py:216-222,252-258): y[b,c,j] = act(scale[c]*(x[b,c,j] + t[b,c,idx[b,j]]) + shift[c]),
 * x f32[B,C,m] (pixel half of the 1x1 conv), t f32[B,C,n] (point half, computed at the points), idx i32[B,m]. May run in place. */
int gdm_gather_add_affine_act_hip(const float* x, const float* t, const int32_t* idx, const float* scale, const float* shift,
                                  int B, int C, int n, int m, int act, float slope, float* y, void* stream);
/* ... and, with y_packed != NULL, the result ALSO as the packed split-bf16 operand of the next convolution / GEMM over the
 * [B, C, m/W, W] map (gdm_conv3x3_act_bytes(B, C, m/W, W) bytes, zero border in place): no pack launch in front of that layer. */
int gdm_gather_add_affine_act2_hip(const float* x, const float* t, const int32_t* idx, const float* scale, const float* shift,
                                   int B, int C, int n, int m, int act, float slope, float* y, void* y_packed, int W, void* stream);
/* The same tail with the pixel half of the 1x1 convolution inside, for the 64-channel levels (C == 64):
 * y[b,co,j] = act(scale[co]*(sum_ci W[co,ci] x[b,ci,j] + t[b,co,idx[b,j]]) + shift[co]); wt f32[C,C] = W transposed ([ci][co]). */
int gdm_conv1x1_gather_add_act_hip(const float* x, const float* wt, const float* t, const int32_t* idx, const float* scale,
                                   const float* shift, int B, int C, int n, long m, int act, float slope, float* y, void* stream);
/* The same fusion with the K = 64 channel mix on the matrix cores (split-bf16 x3, fp32 accumulate): wpk = the 64 x 64 weight (row =
 * output channel) packed by gdm_pack_rows64_hip; t_point_major != 0: t is f32[B, n, 64] (the gathered term is then one contiguous
 * 256-byte row per pixel instead of 64 scattered floats); other arguments as above. */
int gdm_conv64_gather_add_act_mfma_hip(const float* x, const void* wpk, const float* t, const int32_t* idx, const float* scale,
                                       const float* shift, int B, int n, long m, int act, float slope, int pixel_major, int t_point_major,
                                       float* y, void* stream);
/* The same (NCHW form) also writing the result as the packed split-bf16 operand of the next convolution over the [B, 64, m / W, W] map
 * (gdm_conv3x3_act_bytes(B, 64, m / W, W) bytes, zero border kept by the caller).  ypk NULL = gdm_conv64_gather_add_act_mfma_hip. */
int gdm_conv64_gather_add_act_mfma2_hip(const float* x, const void* wpk, const float* t, const int32_t* idx, const float* scale,
                                        const float* shift, int B, int n, long m, int act, float slope, int pixel_major,
                                        int t_point_major, float* y, void* ypk, int W, void* stream);
/* The same with the output layout selectable: pixel_major != 0 writes y f32[B, m, C] (one 256-byte row per pixel). */
int gdm_conv1x1_gather_add_act2_hip(const float* x, const float* wt, const float* t, const int32_t* idx, const float* scale,
                                    const float* shift, int B, int C, int n, long m, int act, float slope, int pixel_major,
                                    float* y, void* stream);

/* ---------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on split-bf16 MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulate),
 * for the 32x32-resolution ResNet-18 layers (models/cnn/extractors.py:36-58,151-177) where MIOpen's fp32 path is a
 * vector-ALU Winograd kernel.  Cin, Cout multiples of 128, W a multiple of 32.
 *   pack_weight: w f32[Cout,Cin,3,3] -> wpk (gdm_conv3x3_weight_bytes), once per weight change
 *   pack_act   : x f32[B,Cin,H,W]    -> xpk (gdm_conv3x3_act_bytes; pixel-major rows with a one-pixel ZERO border: the
 *                caller provides a zero-filled buffer, only interior rows are written)
 *   packed     : out f32[B,Cout,H,W] = act(scale[co] * conv + shift[co] (+ res)), act 0 none / 1 ReLU; scale/shift/res may be NULL */
size_t gdm_conv3x3_act_bytes(int B, int Cin, int H, int W);
size_t gdm_conv3x3_weight_bytes(int Cout, int Cin);
int gdm_conv3x3_pack_weight_hip(const float* w, int Cout, int Cin, void* wpk, void* stream);
int gdm_conv3x3_pack_act_hip(const float* x, int B, int Cin, int H, int W, void* xpk, void* stream);
int gdm_conv3x3_packed_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                           int B, int Cin, int Cout, int H, int W, int act, float* out, void* stream);
/* The same convolution with the result ALSO (or only: out == NULL) written as the packed operand of the next convolution
 * (outpk: gdm_conv3x3_act_bytes(B, Cout, H, W) bytes, zero-filled once by the caller), which then needs no pack launch.  */
int gdm_conv3x3_packed2_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                            int B, int Cin, int Cout, int H, int W, int act, float* out, void* outpk, void* stream);
/* Strided forms (stride 1 or 2; the first block of ResNet-18 layer2, extractors.py:151-177): H, W are the OUTPUT size, xpk is the
 * (H*stride) x (W*stride) input packed by gdm_conv3x3_pack_act_hip; gdm_conv1x1_strided_hip is the block's 1x1 downsample branch
 * on the same packed input (NCHW output). */
int gdm_conv3x3_strided_hip(const void* xpk, const void* wpk, const float* scale, const float* shift, const float* res,
                            int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* outpk, void* stream);
int gdm_conv1x1_strided_hip(const void* xpk, const void* wpk, const float* scale, const float* shift,
                            int B, int Cin, int Cout, int H, int W, int stride, int act, float* out, void* stream);

/* `final` stage of the image branch (pspnet.py:108-112): out = log_softmax_c(W x + b), x,out f32[B,64,hw], W f32[64,64]. */
/* ResNet stem tail in one pass (extractors.py:128-131 after conv1): y = MaxPool2d(3, stride 2, padding 1)(relu(scale[c] * x + shift[c])),
 * x f32[B,C,H,W] -> y f32[B,C,(H-1)/2+1,(W-1)/2+1]; scale / shift = eval-mode BatchNorm folded. */
int gdm_affine_relu_maxpool_hip(const float* x, const float* scale, const float* shift, int B, int C, int H, int W, float* y, void* stream);
int gdm_conv1x1_logsoftmax_hip(const float* x, const float* w, const float* bias, int B, int C, long hw, float* out, void* stream);
/* The last image stage of FFB6DEmb at the SAMPLED pixels only (inference; /root/reference/models/ffb6d.py:266-285: cnn_up_stages[3] =
 * PSPUpsample(64 -> 64) + `final`, then torch.gather with `choose`): out f32[B,64,N] = log_softmax(Wf . act(scale * conv3x3(up(x)) +
 * shift) + fbias) at pixel choose[b,n] of the OHxOW map.  xpm f32[B, H*W, 64] is the source map PIXEL-major (written by
 * gdm_conv1x1_gather_add_act2_hip(pixel_major = 1)); wpk = gdm_upconv_fused64_pack_weight_hip's rows; wfpk = the 64x64 `final` weight
 * packed by gdm_pack_rows64_hip (R rows of 64 floats -> 256-byte split-bf16 rows); fbias may be NULL.  choose is clamped to the map. */
int gdm_pack_rows64_hip(const float* w, int R, void* out, void* stream);
int gdm_upconv_final_points_hip(const float* xpm, const int32_t* choose, const void* wpk, const float* scale, const float* shift,
                                int act, float slope, const void* wfpk, const float* fbias, int B, int H, int W, int OH, int OW,
                                int N, float* out, void* stream);
/* The four adaptive average pools (1,2,3,6 bins) of the pyramid pooling module (pspnet.py:17-20) in one pass:
 * x f32[planes,H,W] -> o1 f32[planes,1], o2 [planes,4], o3 [planes,9], o6 [planes,36] (PyTorch bin edges). */
int gdm_psp_pools_hip(const float* x, long planes, int H, int W, float* o1, float* o2, float* o3, float* o6, void* stream);
/* Its backward (training): grad_x[planes,H,W] = sum over every bin containing the pixel of grad_bin / bin area, all four sizes in one
 * pass (replaces four adaptive_avg_pool2d backward launches of `PSPModule.stages`, pspnet.py:17-20, and the sums that merge them). */
int gdm_psp_pools_bwd_hip(const float* g1, const float* g2, const float* g3, const float* g6, long planes, int H, int W,
                          float* grad_x, void* stream);

/* 1x1 convolution / GEMM on the same split-bf16 MFMA kernel (one tap): x packed by gdm_conv3x3_pack_act_hip, weights
 * f32[Cout,Cin] packed by gdm_conv1x1_pack_weight_hip.  out = act(scale*(W x)+shift) as f32[B,Cout,H,W], or, with
 * pixel_major != 0, as f32[B*H*W, Cout] (used for the dense part of SplineConv: nodes x (125*128)).  Cin % 128 == 0; Cout is
 * arbitrary: the packed weights carry zero rows up to the next multiple of 128 and the extra outputs are never stored. */
size_t gdm_conv1x1_weight_bytes(int Cout, int Cin);
int gdm_conv1x1_pack_weight_hip(const float* w, int Cout, int Cin, void* wpk, void* stream);
/* training: the packed weights of the INPUT-GRADIENT convolution (the flipped, transposed filter) straight from the forward weight
 * w f32[Cout,Cin,taps], taps 9 (3x3/s1/p1) or 1: a layer with Cin outputs and Cout inputs (Cout % 128 == 0 or Cout == 64); wpk holds
 * gdm_conv3x3_weight_bytes(Cin, Cout) / gdm_conv1x1_weight_bytes(Cin, Cout) bytes.  Replaces the flip / transpose / contiguous copies
 * torch's convolution backward makes of the weights of models/cnn/extractors.py:36-58. */
int gdm_conv_pack_weight_dgrad_hip(const float* w, int Cout, int Cin, int taps, void* wpk, void* stream);
int gdm_conv1x1_packed_hip(const void* xpk, const void* wpk, const float* scale, const float* shift,
                           int B, int Cin, int Cout, int H, int W, int act, int pixel_major, float* out, void* stream);

/* Front end: depth (m) f32[B,H,W] + intrinsics K f32[B,3,3] + integer crop origin (x0,y0) i32[B,2] -> camera-frame
 * xyz f32[B,S,S,3] of the SxS crop (datasets/lm/linemod_pbr.py:398-411 dpt_2_pcld; zeros where depth <= 1e-8). */
int gdm_depth_to_xyz_hip(const float* depth, const float* K, const int32_t* origin, int B, int H, int W, int S,
                         float* out, void* stream);

/* ---- training-mode BatchNorm (+ ReLU / LeakyReLU), forward and backward -------------------------------------------------------
 * Replaces the conv -> nn.BatchNorm{1,2}d -> activation chains of the embedding network in the training step
 * (models/pytorch_utils.py:70-124, models/RandLA/pytorch_utils.py:34-105, models/cnn/extractors.py:36-58; train_lm.py:171-225).
 * x, y, grad f32[B,C,inner] contiguous (inner % 4 == 0, B*C <= 65535, 16-byte aligned).  sums: double[gdm_bn_sums_len(B,C,inner)] =
 * G partial pairs per channel of (sum x, sum x^2) -- or (sum g', sum g' x) in the backward, g' = grad * act'(.) -- then the element
 * count per channel and G; no atomics, no memset, the apply calls add the partials (`groups` = 0: the layout the reduce call of the
 * same shape wrote).  A data-parallel caller (SyncBatchNorm semantics, /root/reference/train_lm.py:412) folds the partials to one
 * pair per channel, all-reduces pairs and count, and passes that buffer -- double[2C+1]: pairs, count -- with `groups` = 1.
 * saved f32[4C] = folded scale | folded shift | mean | rstd, written by the forward apply, read by both backward calls.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  running_mean / running_var (both or neither) are updated with `momentum`. */
long gdm_bn_sums_len(int B, int C, long inner);
int gdm_bn_stats_hip(const float* x, int B, int C, long inner, double* sums, void* stream);
int gdm_bn_fwd_apply_hip(const float* x, const double* sums, int groups, const float* weight, const float* bias, int B, int C, long inner, float eps,
                         float momentum, int act, float slope, float* saved, float* running_mean, float* running_var, float* y, void* stream);
int gdm_bn_bwd_reduce_hip(const float* x, const float* grad_out, const float* saved, int B, int C, long inner, int act, float slope,
                          double* sums, void* stream);
int gdm_bn_bwd_apply_hip(const float* x, const float* grad_out, const double* sums, int groups, const float* weight, const float* saved, int B, int C,
                         long inner, int act, float slope, float* grad_weight, float* grad_bias, float* grad_x, void* stream);

/* ---------------------------------------------------------------------------------------
 * The per-point heads of GeoMatch.forward in one kernel (inference; /root/reference/models/geoMatch.py:159-200: feature_encoding_layer,
 * normalize_feature_layer, the residual add and seg_layer -- nine 1x1 convolutions per scene point).  Input x0 = channels of a
 * f32[B,Ca,N] followed by b f32[B,128-Ca,N] (b may be NULL when Ca == 128).  `nlayer` hidden layers 128 -> 128:
 *   x_{l+1} = act_l(scale_l * (W_l x_l) + shift_l)  [+ x0 after layer res_layer];  out_feat f32[B,128,N] = the affine output of layer
 *   feat_layer (before the residual);  then out_last f32[B,c_last,N] = W_last x_nlayer + shift_last (c_last <= 16).
 * w[l], w_last: weights packed by gdm_conv1x1_pack_weight_hip(Cout, 128) (rows padded to 128); scale[l] / shift[l] may be NULL (1 / 0);
 * act[l]: 0 none, 1 ReLU; feat_layer / res_layer: -1 = none.  w, scale, shift, act are HOST arrays of nlayer entries.
 * Split-bf16 products (hi*hi + hi*lo + lo*hi), fp32 accumulate, as the convolution kernels. */
int gdm_point_heads_hip(const float* a, const float* b, int Ca, int B, int N, int nlayer, const void* const* w,
                        const float* const* scale, const float* const* shift, const int* act, int feat_layer, int res_layer,
                        const void* w_last, const float* shift_last, int c_last, float* out_feat, float* out_last, void* stream);
/* The same chain in pieces: the tensor added at res_layer comes from (ra, rb, rCa) instead of the input (NULL ra = the input), c_last = 0
 * ends the chain with its hidden layers (w_last / out_last unused), out_feat may be NULL with feat_layer < 0.  GeoMatch.forward launches
 * the four feature layers first and the normalise / segmentation layers second, so that the matching kernel -- which reads the features
 * only -- runs beside the second piece. */
int gdm_point_heads2_hip(const float* a, const float* b, int Ca, const float* ra, const float* rb, int rCa, int B, int N, int nlayer,
                         const void* const* w, const float* const* scale, const float* const* shift, const int* act, int feat_layer,
                         int res_layer, const void* w_last, const float* shift_last, int c_last, float* out_feat, float* out_last,
                         void* stream);

/* ---------------------------------------------------------------------------------------
 * One per-point (1x1) layer in a single launch (inference): concat-free, BatchNorm / bias / activation / residual branch folded in.
 * Replaces torch.cat -> Conv1d/Conv2d(1x1) -> BatchNorm -> activation of /root/reference/models/pytorch_utils.py:70-124 and
 * models/RandLA/pytorch_utils.py:34-99, the tail lrelu(mlp2(f) + shortcut(x)) of Dilated_res_block (RandLANet.py:685-688), the
 * fusion layers over cat(point, pooled pixel) features (ffb6d.py:224-231,259-265) and the decoder layers over
 * cat(skip, nearest_interpolation(deeper)) (ffb6d.py:246-250,268-272; the interpolation = `idx` of the second segment).
 *   out[b, out_c0 + c, i] = act( scale[c] * sum_k wt[k][c] * X[b,k,i] + shift[c] )
 * X = the channels of segs[0], segs[1], ... in order (the concat, never formed; nseg in [1,4], four only when K >= 32).  A segment is f32[B,C,n_src]
 * channel-major, read at column i (n_src == n) or at idx[b*n + i] (i32, any n_src).  A residual branch
 * s1*(W1 . x1) + b1 + s2*(W2 . x2) + b2 is the two-segment layer with wt = [s1*W1^T ; s2*W2^T], shift = b1 + b2 (the caller folds).
 * wt f32[K,Cout] TRANSPOSED weight, K = sum of the segments' C; scale / shift f32[Cout] or NULL (1 / 0).
 * act: 0 none, 1 ReLU, 2 leaky ReLU (slope).  out f32[B,out_C,n] (point_major 0) or f32[B*n,out_C] (1), 16-byte aligned;
 * channels [out_c0, out_c0 + Cout) of it are written.  fp32 FMAs; the K axis may be summed as up to four partial sums added in
 * fixed order (deterministic; no atomics). */
typedef struct {
    const float* x;
    const int32_t* idx;
    int32_t C, n_src;
} gdm_pw_seg;
int gdm_pointwise_hip(const gdm_pw_seg* segs, int nseg, const float* wt, const float* scale, const float* shift,
                      int B, int n, int Cout, int act, float slope, float* out, int out_C, int out_c0, int point_major, void* stream);
/* The same layer with the weight as the module holds it when w_rowmajor != 0: wt f32[Cout,K] (nn.Conv1d / nn.Conv2d 1x1 weight), so
 * the TRAINING forward W . x and input gradient W^T . go (wt = the same tensor read as [K' = Cout][Cout' = K], w_rowmajor = 0) of
 * /root/reference/models/pytorch_utils.py:70-124 need no transposed copy of a weight that changes every step. */
int gdm_pointwise2_hip(const gdm_pw_seg* segs, int nseg, const float* wt, int w_rowmajor, const float* scale, const float* shift,
                       int B, int n, int Cout, int act, float slope, float* out, int out_C, int out_c0, int point_major, void* stream);
/* Up to four independent plain layers out_j f32[B,Cout,n_j] = W_j^T . x_j (x_j f32[B,K,n_j], wt_j f32[K,Cout]; no scale / shift /
 * activation) of equal K >= 32 and Cout in ONE launch: the four prior products of the pyramid-pooling module
 * (/root/reference/models/cnn/pspnet.py:17-31, `stage(feats)` of the 1 / 2 / 3 / 6-bin pools folded with the bottleneck's slices).
 * Bit-identical to njobs calls of gdm_pointwise_hip: jobs that would take different K splits alone (gdm_pointwise_hip picks the split
 * from a job's own grid size) are launched apart, one launch per distinct split. */
typedef struct {
    const float* x;
    const float* wt;
    float* out;
    int32_t n;
} gdm_pw_job;
int gdm_pointwise_jobs_hip(const gdm_pw_job* jobs, int njobs, int B, int K, int Cout, void* stream);
/* Two chained narrow per-point layers in one launch, both results written: y0 = act0(s0 (W0 x) + b0) f32[B,C1,n],
 * y1 = act1(s1 (W1 y0) + b1) f32[B,C2,n]; x f32[B,C0,n], w0t f32[C0,C1], w1t f32[C1,C2] (weights transposed), s / b folded
 * BatchNorm or NULL, act 0 none / 1 ReLU / 2 LeakyReLU(slope); C0, C1 <= 16, C2 <= 32.  The RandLA stem fc0 and the first block's
 * mlp1 (/root/reference/models/RandLA/RandLANet.py:19,683-684).  Bit-identical to two gdm_pointwise_hip launches. */
int gdm_pointwise_chain2_hip(const float* x, const float* w0t, const float* s0, const float* b0, int act0, float slope0,
                             const float* w1t, const float* s1, const float* b1, int act1, float slope1,
                             int B, int n, int C0, int C1, int C2, float* y0, float* y1, void* stream);

/* ---------------------------------------------------------------------------------------
 * The ResNet stem in one launch (inference): conv 7x7 / stride 2 / pad 3 (3 -> 64, no bias) + scale / shift (folded BatchNorm) +
 * ReLU + max-pool 3x3 / stride 2 / pad 1 (/root/reference/models/cnn/extractors.py:112-116,181-185: conv1, bn1, relu, maxpool).
 * x f32[B,3,H,W] -> out f32[B,64,PH,PW] (PH = ((H-1)/2)/2 + 1, PW likewise) and / or out_packed = the split-bf16 operand planes of
 * the next 3x3 convolution (gdm_conv3x3_act_bytes(B, 64, PH, PW) bytes, zero border already in place); either may be NULL.
 * wpk: gdm_stem_pack_weight_hip of w f32[64,3,7,7] (gdm_stem_weight_bytes() bytes).  Split-bf16 products, fp32 accumulate. */
size_t gdm_stem_weight_bytes(void);
int gdm_stem_pack_weight_hip(const float* w, void* wpk, void* stream);
int gdm_stem_hip(const float* x, const void* wpk, const float* scale, const float* shift, int B, int H, int W, float* out,
                 void* out_packed, void* stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of a 3x3 / stride 1 / pad 1 convolution on the split-bf16 MFMA GEMM (training; backward of
 * /root/reference/models/cnn/extractors.py:36-58): dW[co,ci,ky,kx] = sum_{b,y,x} go[b,co,y,x] * x[b,ci,y+ky-1,x+kx-1] is the GEMM
 * GO (Cout x pixels) . X_tap^T (pixels x Cin).  These two kernels re-lay the operands for gdm_conv1x1_packed_hip: the contraction (its
 * "Cin") runs over the B*H*W pixels in chunks of 128, its "pixels" are the 9 x Cin grid of (tap, ci), its "weights" the rows of go.
 *   xpk = gdm_wgrad_pack_x_hip(x f32[B,Cin,H,W])   gdm_wgrad_x_bytes(B,Cin,H,W) bytes (the tap shift applied here, zero outside the map)
 *   gpk = gdm_wgrad_pack_go_hip(go f32[B,Cout,H,W]) gdm_wgrad_go_bytes(B,Cout,H,W) bytes
 * then ONE launch of the GEMM with per-image weights, its "images" being P equal parts of n = B*H*W/128/P chunks of the contraction,
 *   gdm_conv1x1_packed_wb_hip(xpk, gpk, n*CoutP*512, P, 128*n, Cout, 9, Cin, parts, stream)      (9*Cin % 256 == 0)
 * gives parts f32[P, Cout, 9, Cin]; dW = their sum over P, permuted to [Cout, Cin, 3, 3].  (Or any chunk range [c0, c0+n) alone:
 * gdm_conv1x1_packed_hip(xpk + c0*32*11*(Cin+2)*16, gpk + c0*CoutP*512, NULL, NULL, 1, 128*n, Cout, 9, Cin, 0, 0, part, stream).)
 * W in {32, 64}, H*W % 128 == 0, Cin % 32 == 0; CoutP = Cout rounded up to 128.  Both byte counts are 0 for an unsupported shape. */
size_t gdm_wgrad_x_bytes(int B, int Cin, int H, int W);
size_t gdm_wgrad_go_bytes(int B, int Cout, int H, int W);
int gdm_wgrad_pack_x_hip(const float* x, int B, int Cin, int H, int W, void* out, void* stream);
int gdm_wgrad_pack_go_hip(const float* go, int B, int Cout, int H, int W, void* out, void* stream);
/* The 1x1 case, dW[co,ci] = sum_{b,p} go[b,co,p] * x[b,ci,p] (x f32[B,Cin,P], P % 128 == 0, Cin % 32 == 0): xpk = gdm_wgrad_pack_x1_hip
 * (gdm_wgrad_x1_bytes bytes), gpk = gdm_wgrad_pack_go_hip(go, B, Cout, P/32, 32), then
 * gdm_conv1x1_packed_wb_hip(xpk, gpk, n*CoutP*512, parts, 128*n, Cout, 1, Cin, out f32[parts, Cout, Cin], stream) (Cin % 256 == 0). */
size_t gdm_wgrad_x1_bytes(int B, int Cin, int P);
int gdm_wgrad_pack_x1_hip(const float* x, int B, int Cin, int P, void* out, void* stream);
/* Measurement aid: `blocks` workgroups of 8 waves, each wave `iters` x 8 independent v_mfma_f32_16x16x32_bf16 on registers (no memory):
 * the rate the matrix pipe sustains on this chip at the clock it holds under that load.  flops = blocks*8*iters*8*16384.  chain = 1: no
 * MFMA depends on its predecessor; chain = 3: three consecutive MFMAs into one accumulator (a split-bf16 product issued back to back). */
int gdm_mfma_probe_hip(int blocks, int iters, int chain, float* sink, void* stream);
/* The same loop with the B operands re-read from LDS: rpu (1, 2, 4) ds_read_b128 per 12 MFMAs (4 = the convolution kernel's ratio).
 * flops = blocks*8*iters*12*16384. */
int gdm_mfma_probe_lds_hip(int blocks, int iters, int rpu, float* sink, void* stream);
/* Small-channel form with no re-layout pass (HBM-bound; the 1x1 layers of the 32 / 64-channel full-resolution stages and of the point
 * branch under training): partial[s][co][ci] = sum over the s-th of nsplit slices of the B*P/32 pixel steps of go[b,co,p] * x[b,ci,p]
 * (rows of P floats, batch strides in elements, P % 32 == 0, split-bf16 MFMA); bias_partial[s][co] (optional) = the row sums of go.
 * The caller adds the nsplit partials. */
int gdm_wgrad_direct_hip(const float* go, long go_bstride, const float* x, long x_bstride, int B, int Cout, int Cin, int P,
                         int nsplit, float* partial, float* bias_partial, void* stream);
/* gdm_conv1x1_packed_hip without epilogue, NCHW output, with the weights of image b at wpk + b*w_bstride (H*W % 256 == 0). */
int gdm_conv1x1_packed_wb_hip(const void* xpk, const void* wpk, long w_bstride, int B, int Cin, int Cout, int H, int W, float* out,
                              void* stream);

/* ---------------------------------------------------------------------------------------
 * Batched strided copies of 4-byte words (f32 / i32), one launch for a table of views: job i fills the dense array
 * dst[B][R1][R2][E] from src[b*sb + r1*s1 + r2*s2 + e] (strides in words).  The neighbour pyramid uses it for the strided xyz
 * grids (datasets/lm/linemod_pbr.py:517-527: R1 x R2 = rows x columns at stride sr), the prefix sub-clouds (:538) and the pooling
 * index prefixes, which the reference makes with numpy slicing + copies in the DataLoader worker. */
typedef struct gdm_copy_job {
    void* dst;
    const void* src;
    int64_t sb, s1, s2;
    int32_t B, R1, R2, E;
} gdm_copy_job;
#define GDM_COPY_MAX_JOBS 16
int gdm_copy_jobs_hip(const gdm_copy_job* jobs /* host array */, int njobs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDM_H */
