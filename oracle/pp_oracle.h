/*
 * pp_oracle.h -- CPU restatement of the reference's PointPillars data-prep path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" for the reference build itself.  The
 * reference (data/pillars.cpp) needs Boost headers that this image lacks, ships
 * no tests, fixtures or golden vectors, and therefore cannot be executed here.
 * This restatement is pinned by (a) the behavioural probes recorded in
 * SURVEY.md section 5.9 / 8c (hand case V1, boundary cases, running-mean and
 * input-order probes) and (b) analytic known answers for the IoU.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#ifndef PP_ORACLE_H
#define PP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (shared with include/pp_hip.h) */
#define PPO_OK 0
#define PPO_ERR_INDEX (-2)   /* reference: pybind11 index_error -> IndexError */
#define PPO_ERR_VALUE (-3)   /* invalid argument */
#define PPO_ERR_WINDING (-4) /* reference: "IOU < 0" -> std::exit(1) */
#define PPO_ERR_NOMEM (-5)

/* pillar emission orders (the reference's order is Boost hash-iteration order,
 * which is implementation-defined; these are the build's deterministic ones) */
#define PPO_ORDER_ROW_MAJOR 0 /* ascending (canvas_y, canvas_x)               */
#define PPO_ORDER_SCRAMBLED 1 /* ascending (cell * mult) mod ncells            */
#define PPO_ORDER_HASH 2      /* reference-style chained hash map (timing leg) */

/* geometry of the implied cell grid: nx = floor((x_max-x_min)/x_step)+1 etc. */
int ppo_grid_dims(double x_step, double y_step, double x_min, double y_min,
                  double x_max, double y_max, int64_t *nx, int64_t *ny);

/* multiplier of the scrambled order for a grid of ncells cells */
int64_t ppo_scramble_mult(int64_t ncells);

/*
 * create_pillars -- data/pillars.cpp:236-398.
 * All arrays are f64 with BYTE strides (any layout, like pybind11 .at()).
 * tensor is [t_shape0, t_shape1, t_shape2], indices is [i_shape0, i_shape1].
 * Never zeroes anything (pillars.cpp never does): untouched slots keep the
 * caller's values.  On PPO_ERR_INDEX the writes made so far persist, as in the
 * reference.  *num_cells receives the number of non-empty cells (uncapped).
 */
int ppo_create_pillars(const void *points, int64_t n_points, int64_t p_stride0,
                       int64_t p_stride1, void *tensor, int64_t t_shape0,
                       int64_t t_shape1, int64_t t_shape2, int64_t t_stride0,
                       int64_t t_stride1, int64_t t_stride2, void *indices,
                       int64_t i_shape0, int64_t i_shape1, int64_t i_stride0,
                       int64_t i_stride1, int max_points_per_pillar,
                       int max_pillars, double x_step, double y_step,
                       double x_min, double y_min, double z_min, double x_max,
                       double y_max, double z_max, double canvas_height,
                       int order, int64_t *num_cells);

/*
 * Per-cell summary used by the size-independent GPU parity tests: for every
 * non-empty cell (row-major order) its canvas_x, canvas_y and point count.
 * cells_out is [cap,3] int64 contiguous; returns number of cells (or <0).
 */
int64_t ppo_cell_counts(const void *points, int64_t n_points, int64_t p_stride0,
                        int64_t p_stride1, int64_t *cells_out, int64_t cap,
                        double x_step, double y_step, double x_min,
                        double y_min, double z_min, double x_max, double y_max,
                        double z_max, double canvas_height);

/* rotated-quad IoU of one pair -- data/pillars.cpp:132-172.  anchor: 4 (x,y)
 * corners declared counter-clockwise; gt: 4 corners declared clockwise.
 * Returns the IoU; *status is PPO_OK or PPO_ERR_WINDING. */
double ppo_iou_pair(const double *anchor8, const double *gt8, int *status);

/*
 * make_ious -- data/pillars.cpp:400-427.  BYTE strides everywhere.
 * a_corners [A,4,2], g_corners [G,4,2], a_centers [A,>=2], g_centers [G,>=2],
 * ious [A,G].  Every entry is written.
 */
int ppo_make_ious(const void *a_corners, int64_t A, int64_t ac_s0,
                  int64_t ac_s1, int64_t ac_s2, const void *g_corners,
                  int64_t G, int64_t gc_s0, int64_t gc_s1, int64_t gc_s2,
                  const void *a_centers, int64_t an_s0, int64_t an_s1,
                  const void *g_centers, int64_t gn_s0, int64_t gn_s1,
                  void *ious, int64_t io_s0, int64_t io_s1);

#ifdef __cplusplus
}
#endif
#endif
