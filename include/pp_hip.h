/*
 * pp_hip.h -- C ABI of the MI355X-native PointPillars hot path (libpp_hip.so).
 *
 * The reference (mr3543/3d-Object-Detection) defines no C ABI of its own: its
 * only native surface is the pybind11 module of data/pillars.cpp:429-435 with
 * two functions, create_pillars (pillars.cpp:236-249) and make_ious
 * (pillars.cpp:400-404).  The entry points below are what a binding for that
 * path binds to -- plain pointers, sizes, BYTE strides and scalars, no torch or
 * numpy types.  Each one cites the reference interface it replaces.
 *
 * All functions return 0 on success or a negative PP_ERR_* code;
 * pp_last_error() returns a thread-local description of the last failure.
 * Nothing here ever falls back to a CPU implementation: without a HIP device
 * every compute entry point fails with PP_ERR_HIP.
 */
#ifndef PP_HIP_H
#define PP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_ERR_INDEX (-2)   /* reference: pybind11 index_error -> IndexError      */
#define PP_ERR_VALUE (-3)   /* invalid argument                                    */
#define PP_ERR_WINDING (-4) /* reference: "IOU < 0" -> std::exit(1), pillars.cpp:166-169 */
#define PP_ERR_NOMEM (-5)
#define PP_ERR_HIP (-6)     /* HIP runtime error / no device                       */
#define PP_ERR_INTERNAL (-7)

/* Pillar emission order.  The reference emits pillars in boost::unordered_map
 * iteration order (pillars.cpp:335), which is implementation-defined; these are
 * the deterministic replacements (DESIGN.md "pillar order"). */
#define PP_ORDER_ROW_MAJOR 0 /* ascending (canvas_y, canvas_x)        */
#define PP_ORDER_SCRAMBLED 1 /* ascending (cell * mult) mod ncells    */

#define PP_NUM_FEATURES 9 /* x,y,z,r,xp,yp,xc,yc,zc -- pillars.cpp:48-56 */
#define PP_MAX_BATCH 32

/* The scalar arguments of create_pillars, pillars.cpp:239-249, in its order. */
typedef struct pp_voxel_params {
  int32_t max_points_per_pillar; /* N */
  int32_t max_pillars;           /* P */
  double x_step, y_step;
  double x_min, y_min, z_min;
  double x_max, y_max, z_max;
  double canvas_height;
  int32_t order; /* PP_ORDER_*; not a reference argument (see above) */
  int32_t reserved;
} pp_voxel_params_t;

typedef struct pp_ctx pp_ctx_t; /* per-device, per-stream workspace owner */

const char *pp_last_error(void);
const char *pp_version(void);
int pp_device_count(void);

/* A context owns the scratch buffers (cell ids, counts, buckets).  Use one
 * context per HIP stream; calls on one context are serialised by that stream. */
int pp_ctx_create(int device, pp_ctx_t **out);
void pp_ctx_destroy(pp_ctx_t *ctx);

/* Pre-size the workspace so that later pp_voxelize_dev calls allocate nothing
 * (needed before HIP-graph capture). */
int pp_voxelize_reserve(pp_ctx_t *ctx, int batch, int64_t max_points,
                        const pp_voxel_params_t *prm);

/*
 * Device-resident voxelizer: replaces the whole voxel stage of
 * PPDataset.__getitem__ (data/dataset.py:88-106): np.zeros + create_pillars
 * (pillars.cpp:236-398) + transpose to [9,P,N] + f64->f32 + indices->int64.
 *
 *   points_dev   [batch][points_stride rows][4] f32 (x,y,z,r), device memory;
 *                sweep b uses its first n_points[b] rows (n_points is a HOST array)
 *   pillars_dev  [batch][9][P][N] f32, fully written (zero padded)
 *   indices_dev  [batch][P][3] int64 {1, canvas_x, canvas_y} / {0,0,0}
 *   num_cells_dev[batch][2] int32 or NULL: {non-empty cells (uncapped), in-range points}
 *   stream       hipStream_t (NULL = default stream)
 */
int pp_voxelize_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                    int64_t points_stride, const int32_t *n_points, int batch,
                    const pp_voxel_params_t *prm, float *pillars_dev,
                    int64_t *indices_dev, int32_t *num_cells_dev);

/*
 * The same voxel stage, software-pipelined over consecutive batches -- what the reference gets from
 * DataLoader prefetching (train.py:120-121: the workers voxelize the next batches while the model
 * consumes the current one).  Inside one batch the three stages depend on each other (split ->
 * tile -> order -> emit); across batches they do not.  One call = ONE launch (k_step) whose workgroups
 * take roles side by side:
 *     the SPLIT stage of the batch handed in now            (points_dev),
 *     the TILE  stage of the batch handed in one call ago,
 *     the ORDER stage (descriptors into pillar order) of the batch handed in two calls ago,
 *     the EMIT  stage (the dense store) of the batch handed in THREE calls ago -> pillars_dev ...
 * The launch boundary is the only synchronisation; the latency-bound binning stages hide behind the
 * store.  Outputs are those of pp_voxelize_dev for the same input, bit for bit (the same code on the
 * same data; only the tile size differs, which the outputs do not depend on).
 *   points_dev    as pp_voxelize_dev's, or NULL (pipeline drain: three such calls flush it); read by
 *                 THIS call's launch only
 *   n_points, batch, prm   describe points_dev (ignored when it is NULL)
 *   pillars_dev, indices_dev, num_cells_dev   receive the batch handed in THREE calls ago, with that
 *                 call's batch / prm shapes; may be NULL while no batch is due
 *   emitted       (may be NULL) 1 when this call wrote outputs, else 0
 * Everything runs on `stream` -- the SAME stream for all calls of one pipeline (stream order is what
 * carries a batch from one stage to the next): a call on another stream while batches are in flight is
 * refused with PP_ERR_VALUE and changes nothing (drain first, or pp_voxelize_step_reset); plain calls on
 * the same context are unaffected (own workspace).
 * HIP graphs: a single call must NOT be captured and replayed -- the workspace slot of every stage
 * rotates from call to call on the host, and a replay would repeat one call's slots.  (The plain
 * pp_voxelize_dev is the capturable form.)
 */
int pp_voxelize_step_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                         int64_t points_stride, const int32_t *n_points, int batch,
                         const pp_voxel_params_t *prm, float *pillars_dev,
                         int64_t *indices_dev, int32_t *num_cells_dev, int *emitted);

/*
 * The software-pipelined form of pp_voxelize_pfn_canvas_reuse_dev (PPFeatureNet.forward and
 * PPScatter.forward, model/model.py:31-62, inside the voxelizer): the SAME pipeline as pp_voxelize_step_dev --
 * split, tile and order roles unchanged, batches in flight are shared between the two entry points -- with
 * the EMIT role as the fused feature net: the batch handed in three calls ago comes out as its BEV canvas
 * instead of the dense [9,P,N] tensor.  One launch per call instead of three dependent ones.
 *   canvas_dev        [batch'][64][H][W] f32 (channels_last: memory [batch'][H][W][64]) for the batch that is
 *                     due (batch' = that call's batch).  It must be ALL ZERO on entry: only the pillars'
 *                     pixels are written.
 *   clear_canvas_dev, clear_indices_dev, clear_batch   (optional) ANOTHER canvas of the same shape whose
 *                     only non-zero pixels are those clear_indices_dev [clear_batch][P][3] names (the
 *                     indices this function returned when it filled that canvas): a CLEAR role of the
 *                     same launch zeroes them.  With two canvases used in turn -- emit into one, clear the
 *                     other, which the network has consumed in between -- no call waits for a clear
 *                     (pp_voxelize_pfn_canvas_reuse_dev clears and fills ONE canvas: a kernel boundary).
 *   pfn_params_dev, channels, canvas_h/w, channels_last, indices_dev, num_cells_dev: as
 *                     pp_voxelize_pfn_canvas_dev; points_dev / n_points / batch / prm / emitted / the stream
 *                     rule: as pp_voxelize_step_dev.
 * Results equal pp_voxelize_pfn_canvas_dev's for the same input bit for bit.
 */
int pp_voxelize_step_pfn_canvas_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                                    int64_t points_stride, const int32_t *n_points, int batch,
                                    const pp_voxel_params_t *prm, const float *pfn_params_dev,
                                    int channels, float *canvas_dev, int canvas_h, int canvas_w,
                                    int channels_last, int64_t *indices_dev, int32_t *num_cells_dev,
                                    float *clear_canvas_dev, const int64_t *clear_indices_dev,
                                    int clear_batch, int *emitted);

/* The instance of k_step a pp_voxelize_step_dev call emits a batch of `batch` sweeps with, as a profiler's
 * kernel trace prints it ("pp::k_step<0, 16>": 16-byte stores, write-through; "<0, 0>": plain stores, beyond
 * 128 MB of dense output; "<1, 0>": N not a multiple of 4) -- bench.py names its roofline kernel with it. */
int pp_voxelize_step_kernel_name(const pp_voxel_params_t *prm, int batch, char *name, int cap);

/* Forgets the batches in flight in pp_voxelize_step_dev's pipeline (end of an epoch, an abandoned
 * stream): the next call starts an empty pipeline.  Nothing is launched. */
int pp_voxelize_step_reset(pp_ctx_t *ctx);

/*
 * The optional last step of the voxel stage (data/dataset.py:102-105): pillar -= data_mean,
 * the per-element dataset mean of the [9,P,N] tensor (pillar_means.pkl), f32 - f32 like the
 * reference, the same mean for every sweep.
 *   pillars_dev [batch][elems_per_sweep] f32 in place; mean_dev [elems_per_sweep] f32
 */
int pp_subtract_mean_dev(pp_ctx_t *ctx, void *stream, float *pillars_dev, int batch,
                         int64_t elems_per_sweep, const float *mean_dev);

/*
 * Device-resident voxelizer with the feature net fused in (inference only;
 * SURVEY 8f rank 1): replaces the voxel stage above AND PPFeatureNet.forward
 * (model/model.py:31-40: conv1x1 9->64, ReLU, then BatchNorm2d in eval mode,
 * max over the N slots, zero-padded slots included).  The dense [9,P,N] tensor
 * is never built.
 *   pfn_params_dev [64][12] f32 per output channel: conv weight w[0..8], conv
 *                  bias, BN scale gamma/sqrt(var+eps), BN shift beta-mean*scale
 *   features_dev   [batch][64][P] f32  == PPFeatureNet's output
 *   indices_dev, num_cells_dev, points_dev, n_points: as pp_voxelize_dev
 */
int pp_voxelize_pfn_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                        int64_t points_stride, const int32_t *n_points, int batch,
                        const pp_voxel_params_t *prm, const float *pfn_params_dev,
                        int channels, float *features_dev, int64_t *indices_dev,
                        int32_t *num_cells_dev);

/*
 * PPFeatureNet.forward in eval mode (model/model.py:31-40) on the dense tensor the plain
 * voxelizer wrote -- for callers that keep the reference's [9,P,N] hand-over: the input is
 * read once and the [64,P,N] intermediate never exists.  Same arithmetic as
 * pp_voxelize_pfn_dev (bit-identical features).
 *   pillars_dev  [batch][9][P][N] f32      pfn_params_dev [64][12] f32 (as above)
 *   features_dev [batch][64][P] f32
 */
int pp_pfn_dense_dev(pp_ctx_t *ctx, void *stream, const float *pillars_dev, int batch,
                     int max_pillars, int max_points_per_pillar, const float *pfn_params_dev,
                     int channels, float *features_dev);

/*
 * PPScatter.forward (model/model.py:53-62) alone, for callers that already hold the feature
 * net's output: canvas[b,:,row,col] = features[b,:,p] for every flagged pillar
 * (indices[b,p] = {1, col, row}); the canvas is fully written (zeroed, then scattered).
 *   features_dev [batch][channels][P] f32, indices_dev [batch][P][3] int64
 *   canvas_dev   [batch][channels][H][W] f32, or [batch][H][W][channels] with channels_last
 */
int pp_scatter_canvas_dev(pp_ctx_t *ctx, void *stream, const float *features_dev,
                          const int64_t *indices_dev, int batch, int channels, int max_pillars,
                          float *canvas_dev, int canvas_h, int canvas_w, int channels_last);

/*
 * PPFeatureNet in TRAINING mode (BatchNorm with batch statistics) without the [64,P,N]
 * intermediate.  The forward needs the per-channel batch statistics of r = ReLU(conv(x)); the
 * backward needs, besides the gradient at each pillar's selected element, per-channel sums that
 * do not depend on the incoming gradient -- both come from ONE pass over the dense tensor:
 *   weight_bias_dev [64][10] f32 {w[0..8], bias}
 *   sums_dev        [21][64] f64: #{z>0}, sum (r-c0), sum (r-c0)^2 with c0 = max(bias,0) (the
 *                   value of a zero-padded slot: shifted sums, no cancellation in the
 *                   variance), S1[d] = sum_{z>0} x_d (9 rows), S2[d] = sum r*x_d (9 rows),
 *                   over all batch*P*N slots
 * The forward output is then pp_pfn_dense_dev with scale/shift from the batch statistics.
 */
int pp_pfn_train_stats_dev(pp_ctx_t *ctx, void *stream, const float *pillars_dev, int batch,
                           int max_pillars, int max_points_per_pillar, const float *weight_bias_dev,
                           int channels, double *sums_dev);

/*
 * ... and the gradient-dependent part of the backward: per (b,c,p) the element the forward's
 * max selected is found again (same fmaf chain), and
 *   sums_dev [12][64] f64: dbeta = sum G, dgamma = sum G*xhat*, then the selected elements'
 *            contribution to db (1 row) and dW[d] (9 rows)
 *   pfn_params_dev [64][12] (the forward table: w, bias, scale, shift), mean_dev / invstd_dev [64]
 *   grad_out_dev [batch][64][P] f32
 * The caller combines: dW = sparse + A*S1 + B*S2, db = sparse + A*cnt + B*sum r with
 * A = scale*(-dbeta/M + mean*dgamma*invstd/M), B = -scale*dgamma*invstd/M, M = batch*P*N.
 */
int pp_pfn_train_backward_dev(pp_ctx_t *ctx, void *stream, const float *pillars_dev, int batch,
                              int max_pillars, int max_points_per_pillar,
                              const float *pfn_params_dev, const float *mean_dev,
                              const float *invstd_dev, const float *grad_out_dev, int channels,
                              double *sums_dev);

/*
 * The same, with PPScatter.forward fused in as well (model/model.py:53-62): the feature
 * vector of every flagged pillar goes straight to its pixel of the BEV canvas,
 * canvas[b, :, row, col] = features[b, :, p]; the [batch][64][P] tensor is never built.
 *   canvas_dev   f32, fully written (zeroed, then scattered): [batch][64][canvas_h][canvas_w]
 *                or, with channels_last != 0, [batch][canvas_h][canvas_w][64] (one 256-byte
 *                store per pillar; the layout MIOpen's NHWC convolutions take directly)
 *   canvas_h     must equal prm->canvas_height; a pillar whose pixel falls outside
 *                canvas_h x canvas_w is not written (PPScatter would raise there)
 */
int pp_voxelize_pfn_canvas_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                               int64_t points_stride, const int32_t *n_points, int batch,
                               const pp_voxel_params_t *prm, const float *pfn_params_dev,
                               int channels, float *canvas_dev, int canvas_h, int canvas_w,
                               int channels_last, int64_t *indices_dev, int32_t *num_cells_dev);

/*
 * The same for a canvas that is handed back call after call (a persistent buffer, as PPScatter's
 * output is per step, model/model.py:53-62): `prev_indices_dev` are the indices the PREVIOUS call of
 * this function (or of pp_voxelize_pfn_canvas_dev) wrote for the SAME canvas_dev with the same batch and
 * prm -- the only pixels of it that are non-zero.  Instead of clearing the whole canvas (64 MB per
 * sweep at 500x500) only those pixels are zeroed (3 MB), by extra workgroups of the tile launch.
 * prev_indices_dev may alias indices_dev (it is read before indices_dev is written); NULL = the
 * canvas's contents are unknown: full clear, exactly pp_voxelize_pfn_canvas_dev.
 */
int pp_voxelize_pfn_canvas_reuse_dev(pp_ctx_t *ctx, void *stream, const float *points_dev,
                                     int64_t points_stride, const int32_t *n_points, int batch,
                                     const pp_voxel_params_t *prm, const float *pfn_params_dev,
                                     int channels, float *canvas_dev, int canvas_h, int canvas_w,
                                     int channels_last, int64_t *indices_dev, int32_t *num_cells_dev,
                                     const int64_t *prev_indices_dev);

/*
 * Host drop-in for create_pillars (pillars.cpp:236-249, exported :433).
 * points [n,>=4], tensor [P',N',>=9], indices [P',>=3]: f64 host arrays with
 * arbitrary BYTE strides, mutated in place; nothing is zeroed (pillars.cpp never
 * zeroes); out-of-range writes return PP_ERR_INDEX after the in-range part was
 * written (pybind11 .mutable_at semantics).  num_cells (may be NULL) receives
 * the number of non-empty cells.
 * Host side (one call at a time per context; the call returns with every
 * device operation finished): a pool of host threads gathers the points and
 * scatters the rows (PP_HOST_THREADS); a cloud whose every value is an f32
 * value crosses the link as f32, both ways where it applies; the context's
 * pinned staging buffers are read and written by the kernels themselves
 * (device-visible host memory; PP_DROPIN_DIRECT=0: copy engines instead).
 */
int pp_create_pillars_f64(pp_ctx_t *ctx, const void *points, int64_t n_points,
                          int64_t p_stride0, int64_t p_stride1, void *tensor,
                          const int64_t t_shape[3], const int64_t t_strides[3],
                          void *indices, const int64_t i_shape[2],
                          const int64_t i_strides[2],
                          const pp_voxel_params_t *prm, int64_t *num_cells);

/*
 * Host drop-in for make_ious (pillars.cpp:400-404, exported :432): gated
 * all-pairs rotated IoU; every ious[i][j] is written.  f64 host arrays, BYTE
 * strides.  a_corners [A,4,2], g_corners [G,4,2], a_centers [A,>=2],
 * g_centers [G,>=2], ious [A,G].
 * The anchors of the previous call stay on the device (80 bytes each, freed
 * with the context): every call compares the caller's anchor arrays bit for
 * bit with a pinned mirror and uploads them again only when something changed
 * (utils/box_utils.py:181-183 passes the same arrays for every sample).
 * Returns PP_ERR_WINDING for a box with the wrong corner order (the
 * reference: std::exit(1), pillars.cpp:166-169); the contents of ious are then
 * unspecified.
 */
int pp_make_ious_f64(pp_ctx_t *ctx, const void *a_corners, int64_t A,
                     const int64_t ac_strides[3], const void *g_corners,
                     int64_t G, const int64_t gc_strides[3],
                     const void *a_centers, const int64_t an_strides[2],
                     const void *g_centers, const int64_t gn_strides[2],
                     void *ious, const int64_t io_strides[2]);

/* Error flag of the IoU / target-assignment launches on this context since the
 * last check: synchronises `stream`, returns PP_ERR_WINDING (and clears the flag)
 * if a pair that passed the centre gate had a wrongly wound box (the reference's
 * "IOU < 0" exit, pillars.cpp:166-169), else PP_OK.  pp_make_ious_f64 calls it itself.
 * PP_ERR_VALUE: the box-centric target kernel's list of pairs above the threshold overflowed
 * (it is sized from the candidate anchors per box, so only boxes whose image-space centre is
 * not finite can get there); entries were dropped, the scratch is re-armed before the next call. */
int pp_iou_check(pp_ctx_t *ctx, void *stream);

/* Device-resident make_ious: contiguous f64 device arrays, ious_dev [A][G]. */
int pp_make_ious_dev(pp_ctx_t *ctx, void *stream, const double *a_corners_dev,
                     const double *a_centers_dev, int64_t a_center_cols,
                     int64_t A, const double *g_corners_dev,
                     const double *g_centers_dev, int64_t g_center_cols,
                     int64_t G, double *ious_dev);

/*
 * Device-resident fused target assignment: replaces create_target
 * (utils/box_utils.py:162-232) including make_ious (pillars.cpp:400-427) and
 * make_target (box_utils.py:70-109); the [A,G] IoU matrix is never
 * materialised.  All inputs contiguous f64 device arrays:
 *   a_corners [A,4,2] a_centers [A,3] a_wlh [A,3] a_yaw [A]          (anchors,
 *       image space, box_utils.py:111-159)
 *   g_corners [G,4,2] g_centers_img [G,3]  (boxes_to_image_space, :19-32)
 *   g_centers [G,3] g_wlh [G,3] g_yaw [G]  (canvas space Box fields)
 *   g_class [G] int32
 * Outputs f32 (data/dataset.py:117-118 casts to float):
 *   cls_targets [A,num_classes], reg_targets [A,9]
 * Limits (PP_ERR_VALUE beyond): 1 <= A <= 16 776 960 anchors, 0 <= G <= 65535, 1..1024 classes.
 * (Anchors on the fly are assigned from the box side while a sample's boxes x workgroups per box fit one grid
 * dimension -- about eleven thousand boxes at BASELINE configs[2] -- and from the anchor side beyond; the results are
 * the same either way.)
 */
typedef struct pp_target_params {
  double pos_thresh;    /* cfg.DATA.IOU_POS_THRESH, config.py:122 */
  double canvas_height; /* cfg.DATA.CANVAS_HEIGHT, config.py:60   */
  int32_t num_classes;  /* cfg.DATA.NUM_CLASSES, config.py:97     */
  int32_t reserved;
} pp_target_params_t;

int pp_assign_targets_dev(pp_ctx_t *ctx, void *stream, int64_t A,
                          const double *a_corners, const double *a_centers,
                          const double *a_wlh, const double *a_yaw, int64_t G,
                          const double *g_corners, const double *g_centers_img,
                          const double *g_centers, const double *g_wlh,
                          const double *g_yaw, const int32_t *g_class,
                          const pp_target_params_t *prm, float *cls_targets,
                          float *reg_targets);

/*
 * The same with the anchors evaluated on the fly instead of read from arrays (SURVEY 8f
 * rank 4: no anchor_boxes.pkl / anchor_xy.pkl, train_prep.py:115-120): the grid of
 * make_anchor_boxes (utils/box_utils.py:111-159) -- anchor i = (y*fm_width + x)*per_cell + d,
 * centre ((x+.5)/fm_scale, (y+.5)/fm_scale, z_d), size / yaw of type d.
 *   anchor_types_dev [per_cell][13] f64: the rotated bottom-corner offsets from the centre
 *                    x0,y0,..,x3,y3 (Box.bottom_corners order), then w, l, h, yaw, z
 * Results equal pp_assign_targets_dev on the uploaded arrays bit for bit.
 */
int pp_assign_targets_grid_dev(pp_ctx_t *ctx, void *stream, int fm_height, int fm_width,
                               double fm_scale, int per_cell, const double *anchor_types_dev,
                               int64_t G, const double *g_corners_dev,
                               const double *g_centers_img_dev, const double *g_centers_dev,
                               const double *g_wlh_dev, const double *g_yaw_dev,
                               const int32_t *g_class_dev, const pp_target_params_t *prm,
                               float *cls_targets_dev, float *reg_targets_dev);

/*
 * The target assignment of ALL samples of a step in one launch.  The reference prepares
 * BATCH_SIZE samples per step (config.py:135, train.py:120-121), each through create_target
 * (data/dataset.py:113-118, utils/box_utils.py:162-232); the kernel is a chain of latencies, not
 * bytes, so the samples' chains run side by side (grid.y = sample) instead of one after the other.
 *   batch      1..PP_MAX_BATCH samples; the anchors are shared by all of them
 *   g_counts   HOST array [batch]: sample b has g_counts[b] ground truths (0 allowed: its targets
 *              are all zero), 0..65535 each
 *   g_*        the samples' ground truths CONCATENATED in sample order: sample b owns rows
 *              [g_counts[0]+..+g_counts[b-1], ..+g_counts[b]) of every g_ array (device memory, layouts
 *              as pp_assign_targets_dev)
 *   cls_targets [batch][A][num_classes], reg_targets [batch][A][9] f32
 * Every sample's result equals what pp_assign_targets_dev / pp_assign_targets_grid_dev writes for
 * that sample alone, bit for bit (same code; the list, counter, ticket and column scratch the
 * last-workgroup tail depends on are per sample).
 */
int pp_assign_targets_batch_dev(pp_ctx_t *ctx, void *stream, int32_t batch, const int32_t *g_counts,
                                int64_t A, const double *a_corners, const double *a_centers,
                                const double *a_wlh, const double *a_yaw, const double *g_corners,
                                const double *g_centers_img, const double *g_centers,
                                const double *g_wlh, const double *g_yaw, const int32_t *g_class,
                                const pp_target_params_t *prm, float *cls_targets,
                                float *reg_targets);
int pp_assign_targets_grid_batch_dev(pp_ctx_t *ctx, void *stream, int32_t batch,
                                     const int32_t *g_counts, int fm_height, int fm_width,
                                     double fm_scale, int per_cell, const double *anchor_types_dev,
                                     const double *g_corners_dev, const double *g_centers_img_dev,
                                     const double *g_centers_dev, const double *g_wlh_dev,
                                     const double *g_yaw_dev, const int32_t *g_class_dev,
                                     const pp_target_params_t *prm, float *cls_targets_dev,
                                     float *reg_targets_dev);

/*
 * Lidar sweep ingest pre-pass: replaces, per sweep, the point
 * preparation of PPDataset.__getitem__ (data/dataset.py:65-82) --
 * LidarPointCloud.from_file rows (first four f32 columns of `raw_cols`),
 * .transform(transmat) in f64 stored as f32, .remove_close(min_dist) and the
 * hstack aggregation (write sweep s at points_out_dev + 4*offset_s).
 *   raw_dev        [n_points][raw_cols] f32 (the .bin file's rows), device memory
 *   transform      row-major 4x4 f64 HOST matrix (ref_car_from_global . global_from_car
 *                  . car_from_sensor, dataset.py:76)
 *   points_out_dev [n_points][4] f32; a removed point keeps its row with x = NaN,
 *                  which pp_voxelize_dev drops, so input order is preserved.
 */
int pp_ingest_dev(pp_ctx_t *ctx, void *stream, const float *raw_dev, int64_t n_points,
                  int raw_cols, const double *transform_rowmajor4x4, double min_dist,
                  float *points_out_dev);

/* All sweeps of a sample in one launch (dataset.py:65-82 loops over num_sweeps files and hstacks):
 * raw_dev[s] / n_points[s] / transforms_rowmajor4x4 + 16*s describe sweep s (HOST arrays of device
 * pointers, sizes, 4x4 f64 matrices); sweep s lands at points_out_dev + 4 * (n_points[0] + ... +
 * n_points[s-1]).  Same arithmetic as pp_ingest_dev.  1 <= n_sweeps <= PP_MAX_INGEST_SWEEPS. */
#define PP_MAX_INGEST_SWEEPS 16
int pp_ingest_sweeps_dev(pp_ctx_t *ctx, void *stream, int32_t n_sweeps, const float *const *raw_dev,
                         const int64_t *n_points, int raw_cols, const double *transforms_rowmajor4x4,
                         double min_dist, float *points_out_dev);

/*
 * Inference post-processing on the device: replaces the
 * per-sample tail of evaluate() (evaluate.py:231-245): sigmoid / tanh / class max /
 * score threshold / box_nms (:127-139, torchvision.ops.nms over the ANCHOR
 * rectangles) / first max_out survivors / make_pred_boxes (:33-89) /
 * move_box_to_car_space (:91-125).
 *   cls_dev [Ac*C][H][W] f32, reg_dev [Ac*8][H][W] f32: one sample of PPModel's output
 *   a_centers [A,3] a_wlh [A,3] a_yaw [A] a_xy [A,4]: anchors, f64 device arrays
 *       (box_utils.py:111-159; a_xy = the (x1,y1,x2,y2) rows of anchor_xy.pkl)
 *   boxes_out [max_out][9] f64: car-space x,y,z,w,l,h,yaw,score,class (zeros beyond count)
 *   kept_out  [max_out] int32 anchor ids in keep order (-1 beyond count); count_out [1]
 */
typedef struct pp_decode_params {
  int32_t fm_height, fm_width, anchors_per_cell, num_classes;
  double pos_thresh;    /* cfg.DATA.VAL_POS_THRESH, config.py:153 */
  double nms_thresh;    /* cfg.DATA.VAL_NMS_THRESH, config.py:154 */
  int32_t max_out;      /* 100, evaluate.py:241 */
  int32_t reserved;
  double canvas_height, x_step, y_step, x_min, y_min; /* config.py:46-53,60 */
} pp_decode_params_t;

int pp_decode_dev(pp_ctx_t *ctx, void *stream, const float *cls_dev, const float *reg_dev,
                  const double *a_centers, const double *a_wlh, const double *a_yaw,
                  const double *a_xy, const pp_decode_params_t *prm, double *boxes_out,
                  int32_t *kept_out, int32_t *count_out);

/* The same on strided network outputs: element (channel ch, cell y*W+x) of cls / reg lives at
 * [ch*stride_c + cell*stride_pix] (in floats).  NCHW planes: (H*W, 1) -- what pp_decode_dev
 * passes; channels-last tensors or channel slices of one: (1, row pitch), where an anchor's
 * class logits and box offsets are contiguous. */
int pp_decode_strided_dev(pp_ctx_t *ctx, void *stream, const float *cls_dev, const float *reg_dev,
                          int64_t cls_stride_c, int64_t cls_stride_pix, int64_t reg_stride_c,
                          int64_t reg_stride_pix, const double *a_centers, const double *a_wlh,
                          const double *a_yaw, const double *a_xy, const pp_decode_params_t *prm,
                          double *boxes_out, int32_t *kept_out, int32_t *count_out);

/* A batch of samples in one call (evaluate.py:231-245 loops over the samples of a batch; here every
 * sample gets its own workgroups of the three kernels and they run side by side): sample b of cls /
 * reg starts b*stride_b floats after sample 0; boxes_out [batch][max_out][9], kept_out
 * [batch][max_out], count_out [batch].  1 <= batch <= 1024. */
int pp_decode_batch_dev(pp_ctx_t *ctx, void *stream, int32_t batch, const float *cls_dev,
                        const float *reg_dev, int64_t cls_stride_b, int64_t cls_stride_c,
                        int64_t cls_stride_pix, int64_t reg_stride_b, int64_t reg_stride_c,
                        int64_t reg_stride_pix, const double *a_centers, const double *a_wlh,
                        const double *a_yaw, const double *a_xy, const pp_decode_params_t *prm,
                        double *boxes_out, int32_t *kept_out, int32_t *count_out);

/*
 * Fused conv epilogue for the inference backbone: y = max(x + b_c, 0) * s_c + t_c in
 * place on a contiguous NCHW f32 tensor -- the ReLU -> BatchNorm2d(eval) tail (plus the
 * conv bias) of every block of model/model.py:76-84,105-109 in one pass.
 *   x_dev [batch][channels][hw] f32; params_dev [channels][3] f32 {bias, scale, shift}
 *   y_dev NULL: in place.  Otherwise the result goes to channels
 *   [y_channel_offset, y_channel_offset+channels) of y_dev [batch][y_channels][hw] -- the
 *   up blocks write straight into the concatenated backbone output (model/model.py:140).
 */
int pp_bias_relu_bn_dev(pp_ctx_t *ctx, void *stream, float *x_dev, int64_t batch, int channels,
                        int64_t hw, const float *params_dev, float *y_dev, int64_t y_channels,
                        int64_t y_channel_offset);

/* The same epilogue on a channels-last tensor: x_dev [pixels][channels] f32 (pixels =
 * batch*h*w), channels a multiple of 4; the result goes in place (y_dev NULL) or to
 * channels [y_channel_offset, +channels) of y_dev [pixels][y_channels]. */
int pp_bias_relu_bn_nhwc_dev(pp_ctx_t *ctx, void *stream, float *x_dev, int64_t pixels, int channels,
                             const float *params_dev, float *y_dev, int64_t y_channels,
                             int64_t y_channel_offset);

/*
 * The ReLU -> BatchNorm2d tail of the backbone blocks in TRAINING mode (model/model.py:76-84,
 * 105-109; BatchNorm with batch statistics), forward and backward, on NCHW f32 tensors:
 *   forward   y = gamma*(max(z,0) - mean)*invstd + beta with the batch mean / biased variance
 *             of max(z,0) per channel; mean_out / invstd_out [channels] are kept for the
 *             backward; running_mean / running_var (both or neither NULL) are updated with
 *             `momentum` (unbiased variance), exactly as nn.BatchNorm2d does.
 *   backward  dz = [z > 0] * gamma*invstd * (dy - mean(dy) - xhat*mean(dy*xhat)),
 *             dgamma = sum dy*xhat, dbeta = sum dy.  Only z is needed from the forward.
 *   z_dev, y_dev, dy_dev, dz_dev [batch][channels][hw] f32
 *   dy_batch_stride (floats; 0 = channels*hw): dy may be a channel slice of a wider NCHW
 *             tensor (the gradient of one input of torch.cat), read in place
 *   conv_bias_dev [channels] or NULL: z is then the convolution WITHOUT its bias, the kernels
 *             use z + bias, and the backward also returns dbias_dev = sum dz (may be NULL)
 */
int pp_relu_bn_train_fwd_dev(pp_ctx_t *ctx, void *stream, const float *z_dev,
                             const float *conv_bias_dev, int64_t batch, int channels, int64_t hw,
                             const float *gamma_dev, const float *beta_dev,
                             double eps, double momentum, float *running_mean_dev,
                             float *running_var_dev, float *y_dev, float *mean_out_dev,
                             float *invstd_out_dev);
int pp_relu_bn_train_bwd_dev(pp_ctx_t *ctx, void *stream, const float *z_dev,
                             const float *conv_bias_dev, const float *dy_dev, int64_t dy_batch_stride,
                             int64_t batch, int channels, int64_t hw, const float *gamma_dev,
                             const float *mean_dev,
                             const float *invstd_dev, float *dz_dev, float *dgamma_dev,
                             float *dbeta_dev, float *dbias_dev);

/*
 * The device entry points never synchronise.  pp_voxelize_check synchronises `stream` and
 * returns PP_ERR_HIP (with the runtime's message) if a launch on it failed, else PP_OK.  The
 * voxelizer kernels themselves have no failure mode of their own any more: no workgroup waits
 * for another one (round 1's look-back scan and its time-out flag are gone), every array is
 * written before it is read within one call.  (The reference's only analogue is to fail
 * loudly: pillars.cpp:166-169.)
 */
int pp_voxelize_check(pp_ctx_t *ctx, void *stream);

/*
 * Limits of the voxelizer entry points (PP_ERR_VALUE beyond them):
 *   batch                  <= PP_MAX_BATCH (32) sweeps per call
 *   points per sweep       <  2^30
 *   cell grid              <= 32768 cells per axis and <= 16 777 216 cells in all (a tile of
 *                             up to 4096 cells lives in LDS, at most 4096 tiles); the workspace
 *                             holds 16 bytes per cell and sweep for the tiles' descriptor lists
 *   max_points_per_pillar  <= 65536; dense output: max_pillars * max_points_per_pillar <= 1e8
 *   fused feature net      exactly 64 output channels (model/model.py:28)
 * One context per HIP stream: calls on one context must be stream-ordered.
 */

/* Timing hooks for bench.py: with a ring of `slots` HIP event pairs per kernel
 * (slots = 0 disables), every pp_voxelize*_dev call brackets each of its three launches
 * (PP_KERNEL_SPLIT, PP_KERNEL_TILE, PP_KERNEL_EMIT) with an event pair bound to the dispatch
 * packet.  pp_ctx_read_kernel_ms synchronises on the recorded stop events and returns up to
 * `cap` elapsed times of kernel `which` (oldest first, in milliseconds); reading
 * PP_KERNEL_EMIT empties the ring.  pp_ctx_read_emit_ms = the PP_KERNEL_EMIT row.
 * pp_voxelize_step_dev launches ONE kernel (k_step); its pair is recorded in the PP_KERNEL_EMIT
 * column and the other two columns read as empty. */
#define PP_KERNEL_SPLIT 0
#define PP_KERNEL_TILE 1
#define PP_KERNEL_EMIT 2
int pp_ctx_set_timing(pp_ctx_t *ctx, int slots);
int pp_ctx_read_kernel_ms(pp_ctx_t *ctx, int which, float *ms, int cap, int *count);
int pp_ctx_read_emit_ms(pp_ctx_t *ctx, float *ms, int cap, int *count);

/* Self-test of the host thread pool behind the two host entry points (pp_create_pillars_f64's gather and scatter,
 * pp_make_ious_f64's zero fill): needs no device.  Runs `jobs` jobs on a pool of `threads` threads, alternating the
 * blocking form and the start / wait form, every part adding its own range of 1..n into a per-part slot; returns 0
 * when every part of every job ran exactly once with the right (part, parts) pair, else the index of the first job
 * that went wrong + 1.  (The reference has no counterpart: its module is single-threaded and holds the GIL,
 * data/pillars.cpp:429-435.) */
int pp_host_pool_selftest(int threads, int jobs, int n);

#ifdef __cplusplus
}
#endif
#endif
