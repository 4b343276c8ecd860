/*
 * pp_oracle.c -- CPU restatement of data/pillars.cpp (create_pillars, make_ious).
 *
 * TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  PARITY STATUS: parity unpinned
 * against the real Boost build (unbuildable here, no reference fixtures);
 * pinned by SURVEY.md 5.9/8c probe values and analytic IoU known answers.
 *
 * Build: see oracle/Makefile.  Compiled WITHOUT fused multiply-add contraction
 * (-ffp-contract=off) because the reference is an x86-64 g++ -O3 build
 * (install_mods.sh:8) whose baseline ISA has no FMA: every product and sum
 * below rounds separately, and the HIP kernels do the same.
 */
#include "pp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* strided f64 access with the bounds checks of pybind11 .at()/.mutable_at()  */
/* ------------------------------------------------------------------------- */

static inline double ld(const void *base, int64_t off) {
  double v;
  memcpy(&v, (const char *)base + off, sizeof v);
  return v;
}
static inline void st(void *base, int64_t off, double v) {
  memcpy((char *)base + off, &v, sizeof v);
}

int ppo_grid_dims(double x_step, double y_step, double x_min, double y_min,
                  double x_max, double y_max, int64_t *nx, int64_t *ny) {
  if (!(x_step > 0.0) || !(y_step > 0.0) || !(x_max > x_min) ||
      !(y_max > y_min))
    return PPO_ERR_VALUE;
  /* every accepted x satisfies floor((x-x_min)/x_step) <= floor((x_max-x_min)/x_step):
   * IEEE subtraction and division are monotonic, so this bound is exact. */
  double qx = floor((x_max - x_min) / x_step);
  double qy = floor((y_max - y_min) / y_step);
  if (!(qx < 1e9) || !(qy < 1e9)) return PPO_ERR_VALUE;
  *nx = (int64_t)qx + 1;
  *ny = (int64_t)qy + 1;
  return PPO_OK;
}

static int64_t gcd64(int64_t a, int64_t b) {
  while (b) {
    int64_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}

int64_t ppo_scramble_mult(int64_t ncells) {
  /* golden-ratio stride, bumped to the next value coprime with ncells */
  if (ncells <= 2) return 1;
  int64_t m = (int64_t)floor((double)ncells * 0.6180339887498949);
  if (m < 1) m = 1;
  while (gcd64(m, ncells) != 1) ++m;
  return m % ncells;
}

/* ------------------------------------------------------------------------- */
/* create_pillars: dense-grid formulation                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
  int64_t head, tail; /* first / last point index of the cell (input order) */
  int64_t count;
  double mean[3]; /* running mean, pillars.cpp:311-328 */
} cell_t;

typedef struct {
  int64_t key, cell;
} keyed_t;

static int cmp_keyed(const void *a, const void *b) {
  const keyed_t *x = (const keyed_t *)a, *y = (const keyed_t *)b;
  return (x->key > y->key) - (x->key < y->key);
}

/* one emitted feature row -- PillarPoint::make_feature, pillars.cpp:38-59;
 * feature order x,y,z,r,xp,yp,xc,yc,zc (pillars.cpp:48-56, 30-31, 381-383) */
static int emit_point(void *tensor, int64_t s0, int64_t s1, int64_t s2,
                      int64_t n0, int64_t n1, int64_t n2, int64_t p, int64_t k,
                      const double f[9]) {
  for (int d = 0; d < 9; ++d) {
    if (p >= n0 || k >= n1 || d >= n2) return PPO_ERR_INDEX;
    st(tensor, p * s0 + k * s1 + d * s2, f[d]);
  }
  return PPO_OK;
}

static int create_pillars_hash(const void *points, int64_t n_points,
                               int64_t ps0, int64_t ps1, void *tensor,
                               int64_t t0, int64_t t1, int64_t t2, int64_t ts0,
                               int64_t ts1, int64_t ts2, void *indices,
                               int64_t i0, int64_t i1, int64_t is0, int64_t is1,
                               int max_pts, int max_pillars, double x_step,
                               double y_step, double x_min, double y_min,
                               double z_min, double x_max, double y_max,
                               double z_max, double canvas_height,
                               int64_t *num_cells);

int ppo_create_pillars(const void *points, int64_t n_points, int64_t ps0,
                       int64_t ps1, void *tensor, int64_t t0, int64_t t1,
                       int64_t t2, int64_t ts0, int64_t ts1, int64_t ts2,
                       void *indices, int64_t i0, int64_t i1, int64_t is0,
                       int64_t is1, int max_pts, int max_pillars,
                       double x_step, double y_step, double x_min, double y_min,
                       double z_min, double x_max, double y_max, double z_max,
                       double canvas_height, int order, int64_t *num_cells) {
  if (order == PPO_ORDER_HASH)
    return create_pillars_hash(points, n_points, ps0, ps1, tensor, t0, t1, t2,
                               ts0, ts1, ts2, indices, i0, i1, is0, is1,
                               max_pts, max_pillars, x_step, y_step, x_min,
                               y_min, z_min, x_max, y_max, z_max, canvas_height,
                               num_cells);
  if (order != PPO_ORDER_ROW_MAJOR && order != PPO_ORDER_SCRAMBLED)
    return PPO_ERR_VALUE;
  int64_t nx, ny;
  int rc = ppo_grid_dims(x_step, y_step, x_min, y_min, x_max, y_max, &nx, &ny);
  if (rc) return rc;
  int64_t ncells = nx * ny;
  cell_t *cells = (cell_t *)calloc((size_t)ncells, sizeof(cell_t));
  int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_points + 1));
  keyed_t *occ = (keyed_t *)malloc(sizeof(keyed_t) * (size_t)(n_points + 1));
  if (!cells || !next || !occ) {
    free(cells);
    free(next);
    free(occ);
    return PPO_ERR_NOMEM;
  }
  int64_t n_occ = 0;

  /* loop 1 -- pillars.cpp:268-329 */
  for (int64_t i = 0; i < n_points; ++i) {
    double x = ld(points, i * ps0 + 0 * ps1);
    double y = ld(points, i * ps0 + 1 * ps1);
    double z = ld(points, i * ps0 + 2 * ps1);
    /* half-open range filter, pillars.cpp:271-275.  Written positively so a
     * NaN coordinate is dropped (the reference lets NaN through to floor(NaN):
     * undefined key; dropping it is the documented deviation). */
    if (!(x >= x_min && x < x_max && y >= y_min && y < y_max && z >= z_min &&
          z < z_max))
      continue;
    /* pillars.cpp:278-280: f64 true division, floor, y flipped into rows */
    double fx = floor((x - x_min) / x_step);
    double fy = floor((y - y_min) / y_step);
    int64_t ix = (int64_t)fx, iy = (int64_t)fy;
    int64_t cell = ((ny - 1) - iy) * nx + ix; /* ascending canvas_y, canvas_x */
    cell_t *c = &cells[cell];
    next[i] = -1;
    if (c->count == 0) { /* pillars.cpp:290-295, 311-319 */
      c->head = c->tail = i;
      c->mean[0] = x;
      c->mean[1] = y;
      c->mean[2] = z;
      c->count = 1;
      occ[n_occ].cell = cell;
      occ[n_occ].key = 0;
      ++n_occ;
    } else { /* pillars.cpp:298-302, 320-328 */
      next[c->tail] = i;
      c->tail = i;
      double n = (double)c->count;
      c->mean[0] = c->mean[0] * (n / (n + 1)) + x / (n + 1);
      c->mean[1] = c->mean[1] * (n / (n + 1)) + y / (n + 1);
      c->mean[2] = c->mean[2] * (n / (n + 1)) + z / (n + 1);
      c->count += 1;
    }
  }
  if (num_cells) *num_cells = n_occ;

  /* emission order (replaces Boost hash-iteration order, pillars.cpp:335) */
  int64_t mult = (order == PPO_ORDER_SCRAMBLED) ? ppo_scramble_mult(ncells) : 1;
  for (int64_t k = 0; k < n_occ; ++k)
    occ[k].key = (int64_t)(((__int128)occ[k].cell * mult) % ncells);
  qsort(occ, (size_t)n_occ, sizeof(keyed_t), cmp_keyed);

  /* loop 2 -- pillars.cpp:335-396 */
  rc = PPO_OK;
  int64_t num_pillars = 0;
  for (int64_t k = 0; k < n_occ && rc == PPO_OK; ++k) {
    if (num_pillars >= max_pillars) break; /* pillars.cpp:339 */
    int64_t cell = occ[k].cell;
    cell_t *c = &cells[cell];
    double canvas_x = (double)(cell % nx);
    double fy = (double)((ny - 1) - cell / nx);
    double canvas_y = (canvas_height - 1) - fy; /* pillars.cpp:280 */
    int64_t num_points = 0;
    for (int64_t i = c->head; i >= 0; i = next[i]) {
      if (num_points >= max_pts) break; /* pillars.cpp:369 */
      double x = ld(points, i * ps0 + 0 * ps1);
      double y = ld(points, i * ps0 + 1 * ps1);
      double z = ld(points, i * ps0 + 2 * ps1);
      double r = ld(points, i * ps0 + 3 * ps1);
      double f[9];
      f[0] = x;
      f[1] = y;
      f[2] = z;
      f[3] = r;
      f[4] = canvas_x - x; /* pillars.cpp:30 */
      f[5] = canvas_y - y; /* pillars.cpp:31 */
      f[6] = c->mean[0] - x; /* pillars.cpp:381 */
      f[7] = c->mean[1] - y;
      f[8] = c->mean[2] - z;
      rc = emit_point(tensor, ts0, ts1, ts2, t0, t1, t2, num_pillars,
                      num_points, f);
      if (rc) break;
      ++num_points;
    }
    if (rc) break;
    /* pillars.cpp:390-392 */
    if (num_pillars >= i0 || 2 >= i1) {
      rc = PPO_ERR_INDEX;
      break;
    }
    st(indices, num_pillars * is0 + 0 * is1, 1.0);
    st(indices, num_pillars * is0 + 1 * is1, canvas_x);
    st(indices, num_pillars * is0 + 2 * is1, canvas_y);
    ++num_pillars;
  }
  free(cells);
  free(next);
  free(occ);
  return rc;
}

int64_t ppo_cell_counts(const void *points, int64_t n_points, int64_t ps0,
                        int64_t ps1, int64_t *cells_out, int64_t cap,
                        double x_step, double y_step, double x_min,
                        double y_min, double z_min, double x_max, double y_max,
                        double z_max, double canvas_height) {
  int64_t nx, ny;
  int rc = ppo_grid_dims(x_step, y_step, x_min, y_min, x_max, y_max, &nx, &ny);
  if (rc) return rc;
  int64_t ncells = nx * ny;
  int64_t *cnt = (int64_t *)calloc((size_t)ncells, sizeof(int64_t));
  if (!cnt) return PPO_ERR_NOMEM;
  for (int64_t i = 0; i < n_points; ++i) {
    double x = ld(points, i * ps0 + 0 * ps1);
    double y = ld(points, i * ps0 + 1 * ps1);
    double z = ld(points, i * ps0 + 2 * ps1);
    if (!(x >= x_min && x < x_max && y >= y_min && y < y_max && z >= z_min &&
          z < z_max))
      continue;
    int64_t ix = (int64_t)floor((x - x_min) / x_step);
    int64_t iy = (int64_t)floor((y - y_min) / y_step);
    cnt[((ny - 1) - iy) * nx + ix] += 1;
  }
  int64_t m = 0;
  for (int64_t c = 0; c < ncells; ++c) {
    if (!cnt[c]) continue;
    if (m < cap) {
      double fy = (double)((ny - 1) - c / nx);
      cells_out[3 * m + 0] = c % nx;
      cells_out[3 * m + 1] = (int64_t)((canvas_height - 1) - fy);
      cells_out[3 * m + 2] = cnt[c];
    }
    ++m;
  }
  free(cnt);
  return m;
}

/* ------------------------------------------------------------------------- */
/* create_pillars: reference-style data structures (timing leg, order "hash") */
/* ------------------------------------------------------------------------- */
/* Mirrors the COST structure of pillars.cpp:259-398: one heap object per
 * in-range point (PillarPoint, pillars.h:6-42: 11 doubles), a growable pointer
 * vector per pillar (Pillar, pillars.h:44-63), two hash maps keyed on the
 * (canvas_x, canvas_y) doubles (pillars.cpp:259-260) and a separately
 * allocated double[4] running mean per cell (pillars.cpp:313).  The emission
 * order is this table's bucket order: arbitrary, like Boost's. */

typedef struct {
  double x, y, z, r, xp, yp, xc, yc, zc, canvas_x, canvas_y;
} hpoint_t;

typedef struct {
  double canvas_x, canvas_y;
  hpoint_t **pts;
  int64_t n, cap;
} hpillar_t;

typedef struct hnode {
  double k0, k1;
  void *val;
  struct hnode *next;
} hnode_t;

typedef struct {
  hnode_t **b;
  int64_t nb, n;
} hmap_t;

static uint64_t hash2(double a, double b) {
  /* hash_combine over the two doubles' bit patterns */
  uint64_t u, v;
  memcpy(&u, &a, 8);
  memcpy(&v, &b, 8);
  uint64_t h = u * 0x9E3779B97F4A7C15ull;
  h ^= (v * 0xC2B2AE3D27D4EB4Full) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h ^= h >> 29;
  return h;
}

static int hmap_init(hmap_t *m, int64_t nb) {
  m->b = (hnode_t **)calloc((size_t)nb, sizeof(hnode_t *));
  m->nb = nb;
  m->n = 0;
  return m->b ? 0 : -1;
}

static void *hmap_find(const hmap_t *m, double k0, double k1) {
  for (hnode_t *p = m->b[hash2(k0, k1) % (uint64_t)m->nb]; p; p = p->next)
    if (p->k0 == k0 && p->k1 == k1) return p->val;
  return NULL;
}

static int hmap_insert(hmap_t *m, double k0, double k1, void *val) {
  if (m->n >= m->nb) { /* rehash at load factor 1, like boost::unordered */
    int64_t nb2 = m->nb * 2 + 1;
    hnode_t **b2 = (hnode_t **)calloc((size_t)nb2, sizeof(hnode_t *));
    if (!b2) return -1;
    for (int64_t i = 0; i < m->nb; ++i)
      for (hnode_t *p = m->b[i]; p;) {
        hnode_t *nx = p->next;
        uint64_t h = hash2(p->k0, p->k1) % (uint64_t)nb2;
        p->next = b2[h];
        b2[h] = p;
        p = nx;
      }
    free(m->b);
    m->b = b2;
    m->nb = nb2;
  }
  hnode_t *nd = (hnode_t *)malloc(sizeof(hnode_t));
  if (!nd) return -1;
  uint64_t h = hash2(k0, k1) % (uint64_t)m->nb;
  nd->k0 = k0;
  nd->k1 = k1;
  nd->val = val;
  nd->next = m->b[h];
  m->b[h] = nd;
  m->n += 1;
  return 0;
}

static void hmap_free(hmap_t *m) {
  for (int64_t i = 0; i < m->nb; ++i)
    for (hnode_t *p = m->b[i]; p;) {
      hnode_t *nx = p->next;
      free(p);
      p = nx;
    }
  free(m->b);
}

static int create_pillars_hash(const void *points, int64_t n_points,
                               int64_t ps0, int64_t ps1, void *tensor,
                               int64_t t0, int64_t t1, int64_t t2, int64_t ts0,
                               int64_t ts1, int64_t ts2, void *indices,
                               int64_t i0, int64_t i1, int64_t is0, int64_t is1,
                               int max_pts, int max_pillars, double x_step,
                               double y_step, double x_min, double y_min,
                               double z_min, double x_max, double y_max,
                               double z_max, double canvas_height,
                               int64_t *num_cells) {
  hmap_t pillar_map, means_map;
  if (hmap_init(&pillar_map, 53) || hmap_init(&means_map, 53))
    return PPO_ERR_NOMEM;
  int rc = PPO_OK;
  for (int64_t i = 0; i < n_points; ++i) { /* pillars.cpp:268-329 */
    double x = ld(points, i * ps0 + 0 * ps1);
    double y = ld(points, i * ps0 + 1 * ps1);
    double z = ld(points, i * ps0 + 2 * ps1);
    if (!(x >= x_min && x < x_max && y >= y_min && y < y_max && z >= z_min &&
          z < z_max))
      continue;
    double canvas_x = floor((x - x_min) / x_step);
    double canvas_y = floor((y - y_min) / y_step);
    canvas_y = (canvas_height - 1) - canvas_y;
    hpoint_t *pp = (hpoint_t *)malloc(sizeof(hpoint_t)); /* pillars.cpp:282 */
    if (!pp) {
      rc = PPO_ERR_NOMEM;
      break;
    }
    pp->x = x;
    pp->y = y;
    pp->z = z;
    pp->r = ld(points, i * ps0 + 3 * ps1);
    pp->xp = canvas_x - x;
    pp->yp = canvas_y - y;
    pp->xc = pp->yc = pp->zc = 0;
    hpillar_t *pl = (hpillar_t *)hmap_find(&pillar_map, canvas_x, canvas_y);
    if (!pl) {
      pl = (hpillar_t *)calloc(1, sizeof(hpillar_t));
      if (!pl) {
        rc = PPO_ERR_NOMEM;
        break;
      }
      pl->canvas_x = canvas_x;
      pl->canvas_y = canvas_y;
      hmap_insert(&pillar_map, canvas_x, canvas_y, pl);
    }
    if (pl->n == pl->cap) {
      pl->cap = pl->cap ? pl->cap * 2 : 1;
      pl->pts = (hpoint_t **)realloc(pl->pts, sizeof(hpoint_t *) * (size_t)pl->cap);
    }
    pl->pts[pl->n++] = pp;
    double *means = (double *)hmap_find(&means_map, canvas_x, canvas_y);
    if (!means) {
      means = (double *)malloc(4 * sizeof(double));
      means[0] = x;
      means[1] = y;
      means[2] = z;
      means[3] = 1;
      hmap_insert(&means_map, canvas_x, canvas_y, means);
    } else {
      double n = means[3];
      means[0] = means[0] * (n / (n + 1)) + x / (n + 1);
      means[1] = means[1] * (n / (n + 1)) + y / (n + 1);
      means[2] = means[2] * (n / (n + 1)) + z / (n + 1);
      means[3] = n + 1;
    }
  }
  if (num_cells) *num_cells = pillar_map.n;
  int64_t num_pillars = 0;
  for (int64_t b = 0; b < pillar_map.nb; ++b) {
    for (hnode_t *nd = pillar_map.b[b]; nd; nd = nd->next) {
      hpillar_t *pl = (hpillar_t *)nd->val;
      double *pm = (double *)hmap_find(&means_map, nd->k0, nd->k1);
      if (rc == PPO_OK && num_pillars < max_pillars) {
        /* pillars.cpp:364: the reference copies the pointer vector */
        hpoint_t **copy = (hpoint_t **)malloc(sizeof(hpoint_t *) * (size_t)(pl->n ? pl->n : 1));
        memcpy(copy, pl->pts, sizeof(hpoint_t *) * (size_t)pl->n);
        int64_t num_points = 0;
        for (int64_t i = 0; i < pl->n && rc == PPO_OK; ++i) {
          if (num_points >= max_pts) break;
          hpoint_t *p = copy[i];
          p->xc = pm[0] - p->x;
          p->yc = pm[1] - p->y;
          p->zc = pm[2] - p->z;
          double f[9] = {p->x, p->y, p->z, p->r, p->xp, p->yp, p->xc, p->yc, p->zc};
          rc = emit_point(tensor, ts0, ts1, ts2, t0, t1, t2, num_pillars,
                          num_points, f);
          ++num_points;
        }
        free(copy);
        if (rc == PPO_OK) {
          if (num_pillars >= i0 || 2 >= i1) {
            rc = PPO_ERR_INDEX;
          } else {
            st(indices, num_pillars * is0 + 0 * is1, 1.0);
            st(indices, num_pillars * is0 + 1 * is1, pl->canvas_x);
            st(indices, num_pillars * is0 + 2 * is1, pl->canvas_y);
            ++num_pillars;
          }
        }
      }
      for (int64_t i = 0; i < pl->n; ++i) free(pl->pts[i]);
      free(pl->pts);
      free(pl);
      free(pm);
    }
  }
  hmap_free(&pillar_map);
  hmap_free(&means_map);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* rotated IoU                                                                */
/* ------------------------------------------------------------------------- */
/* bg::intersection / bg::area (Boost.Geometry, version unpinned and absent;
 * call sites pillars.cpp:160,164,165) are restated as the published algorithm
 * for convex polygons: Sutherland-Hodgman clipping of the anchor quad against
 * the four half-planes of the ground-truth quad, shoelace area.  The HIP kernel
 * executes the SAME operation sequence, so HIP-vs-oracle parity is bit-exact. */

static double shoelace(const double *q, int n) { /* positive when CCW */
  double s = 0.0;
  for (int k = 0; k < n; ++k) {
    int j = (k + 1 == n) ? 0 : k + 1;
    s = s + (q[2 * k] * q[2 * j + 1] - q[2 * j] * q[2 * k + 1]);
  }
  return 0.5 * s;
}

double ppo_iou_pair(const double *a, const double *g, int *status) {
  if (status) *status = PPO_OK;
  double area_a = shoelace(a, 4);  /* Polygon_cc: declared CCW, pillars.cpp:16 */
  double area_g = -shoelace(g, 4); /* Polygon: declared CW, pillars.cpp:15 */
  /* reference: wrong winding shows up as IoU < 0 -> std::exit(1)
   * (pillars.cpp:166-169).  Here: any negative declared-orientation area is
   * reported, so wrong winding is never silently accepted. */
  if (area_a < 0.0 || area_g < 0.0) {
    if (status) *status = PPO_ERR_WINDING;
    return -1.0;
  }
  double poly[2][16];
  int n = 4, cur = 0;
  memcpy(poly[0], a, 8 * sizeof(double));
  for (int e = 0; e < 4 && n > 0; ++e) {
    /* clip edges walk the GT quad counter-clockwise: g0, g3, g2, g1 */
    int ia = (4 - e) & 3, ib = (3 - e) & 3;
    double ax = g[2 * ia], ay = g[2 * ia + 1];
    double ex = g[2 * ib] - ax, ey = g[2 * ib + 1] - ay;
    const double *in = poly[cur];
    double *out = poly[cur ^ 1];
    int m = 0;
    double px = in[2 * (n - 1)], py = in[2 * (n - 1) + 1];
    double dp = ex * (py - ay) - ey * (px - ax);
    for (int i = 0; i < n; ++i) {
      double cx = in[2 * i], cy = in[2 * i + 1];
      double dc = ex * (cy - ay) - ey * (cx - ax);
      if ((dc >= 0.0) != (dp >= 0.0)) {
        double t = dp / (dp - dc);
        out[2 * m] = px + t * (cx - px);
        out[2 * m + 1] = py + t * (cy - py);
        ++m;
      }
      if (dc >= 0.0) {
        out[2 * m] = cx;
        out[2 * m + 1] = cy;
        ++m;
      }
      px = cx;
      py = cy;
      dp = dc;
    }
    n = m;
    cur ^= 1;
  }
  if (n < 3) return 0.0; /* pillars.cpp:161-163: no output polygon */
  double inter = shoelace(poly[cur], n);
  if (!(inter > 0.0)) return 0.0;
  return inter / (area_a + area_g - inter); /* pillars.cpp:164-165 */
}

int ppo_make_ious(const void *a_corners, int64_t A, int64_t ac0, int64_t ac1,
                  int64_t ac2, const void *g_corners, int64_t G, int64_t gc0,
                  int64_t gc1, int64_t gc2, const void *a_centers, int64_t an0,
                  int64_t an1, const void *g_centers, int64_t gn0, int64_t gn1,
                  void *ious, int64_t io0, int64_t io1) {
  int rc = PPO_OK;
  for (int64_t i = 0; i < A; ++i) { /* pillars.cpp:416-425 */
    double acx = ld(a_centers, i * an0), acy = ld(a_centers, i * an0 + an1);
    for (int64_t j = 0; j < G; ++j) {
      double gcx = ld(g_centers, j * gn0), gcy = ld(g_centers, j * gn0 + gn1);
      if (fabs(acx - gcx) > 10 || fabs(acy - gcy) > 10) { /* pillars.cpp:418 */
        st(ious, i * io0 + j * io1, 0.0);
        continue;
      }
      double a[8], g[8];
      for (int k = 0; k < 4; ++k) {
        a[2 * k] = ld(a_corners, i * ac0 + k * ac1);
        a[2 * k + 1] = ld(a_corners, i * ac0 + k * ac1 + ac2);
        g[2 * k] = ld(g_corners, j * gc0 + k * gc1);
        g[2 * k + 1] = ld(g_corners, j * gc0 + k * gc1 + gc2);
      }
      int s;
      double v = ppo_iou_pair(a, g, &s);
      if (s != PPO_OK) {
        rc = s;
        v = -1.0;
      }
      st(ious, i * io0 + j * io1, v);
    }
  }
  return rc;
}
