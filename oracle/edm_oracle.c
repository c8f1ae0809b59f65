/*
 * edm_oracle.c -- CPU restatement of the EDM bias hot path (see edm_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker for tests/, smoke() and bench.py's
 * cpu_baseline leg.  Never linked into, or called from, the product path.
 *
 * Plain C, runtime dimension 1..3, IEEE double, same operation order as the
 * reference so that x86-64 results are bit-identical (build with
 * -ffp-contract=off).  file:line citations are relative to /root/reference.
 * Parity pinned against oracle/_ref (the reference itself) by
 * tests/test_oracle_vs_ref.py and against tests/golden/*.json.
 *
 * Deliberate, documented deviations from the reference (all in places where
 * the reference has undefined behaviour):
 *   - all state is zero-initialised (reference leaves hills_added_,
 *     b_skip_hill_add_, total_volume_, overflow_buffer_ uninitialised);
 *   - duplicate_boundary skips index combinations that fall outside the
 *     array instead of writing out of bounds (gaussian_grid.h:578-628);
 *   - negative double -> size_t conversions follow the x86-64 result
 *     (via long long) instead of being undefined.
 */
#define _GNU_SOURCE
#include "edm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GAUSS_SUPPORT 8.0      /* gaussian_grid.h:10 */
#define BC_TABLE_SIZE 65536    /* gaussian_grid.h:11 */
#define BC_MAR 2.0             /* gaussian_grid.h:12 */
#define GRID_TYPE 32           /* grid.h:14 */
#define BIAS_CLAMP 1.0         /* edm_bias.h:14 */
#define BIAS_BUFFER_SIZE 2048  /* edm_bias.h:15 */
#define BIAS_BUFFER_DBLS 8192  /* edm_bias.h:16 */

static void ora_error(const char *msg, const char *where) {
  /* edm.cpp:4-7 */
  fprintf(stderr, "[EDM:%s] %s\n", where, msg);
  abort();
}

/* grid.h:17-20.  The cast binds tighter than the comparison, so the test is
 * on the truncated integer; both branches equal floor() for finite input. */
static int ifloor(double a) {
  double r = ((int)a < 0.0) ? -ceil(fabs(a)) : floor(a);
  return (int)r;
}

/* grid.h:22-26 */
static double round_half(double a) {
  return a < 0.0 ? ceil(a - 0.5) : floor(a + 0.5);
}

static size_t to_index(double a) { return (size_t)(long long)a; }

/* gaussian_grid.h:16-32 */
static double smooth_step(double t) {
  if (t < 0) return 1;
  if (t > 1) return 0;
  return 2 * t * t * t - 3 * t * t + 1;
}
static double smooth_step_dt(double t) {
  if (t < 0) return 0;
  if (t > 1) return 0;
  return 6 * t * t - 6 * t;
}

/* ====================================================================== */
/* plain grid                                                             */
/* ====================================================================== */
struct ora_grid {
  int dim;
  int b_deriv;
  int b_interp;
  size_t total;
  double *v;
  double *dv;
  double dx[EDM_MAXDIM], min[EDM_MAXDIM], max[EDM_MAXDIM];
  int n[EDM_MAXDIM];
  int periodic[EDM_MAXDIM];
};

/* grid.h:892-904 (the reference over-allocates by DIM; sizes are irrelevant
 * to results, so the restatement allocates exactly what is addressed). */
static void grid_alloc(ora_grid *g) {
  int d;
  g->total = 1;
  for (d = 0; d < g->dim; d++) g->total *= (size_t)g->n[d];
  free(g->v);
  free(g->dv);
  g->v = (double *)calloc(g->total ? g->total : 1, sizeof(double));
  g->dv = NULL;
  if (g->b_deriv)
    g->dv = (double *)calloc((g->total ? g->total : 1) * (size_t)g->dim, sizeof(double));
}

/* grid.h:190-213 */
ora_grid *ora_grid_create(int dim, const double *min, const double *max,
                          const double *spacing, const int *periodic,
                          int b_derivatives, int b_interpolate) {
  int d;
  ora_grid *g = (ora_grid *)calloc(1, sizeof(ora_grid));
  g->dim = dim;
  g->b_deriv = b_derivatives;
  g->b_interp = b_interpolate;
  for (d = 0; d < dim; d++) {
    g->min[d] = min[d];
    g->max[d] = max[d];
    g->periodic[d] = periodic[d];
    g->n[d] = (int)ceil((g->max[d] - g->min[d]) / spacing[d]);
    g->dx[d] = (g->max[d] - g->min[d]) / g->n[d];
    g->n[d] = g->periodic[d] ? g->n[d] : g->n[d] + 1;
    if (!g->periodic[d]) g->max[d] += g->dx[d];
  }
  grid_alloc(g);
  return g;
}

void ora_grid_free(ora_grid *g) {
  if (!g) return;
  free(g->v);
  free(g->dv);
  free(g);
}

int ora_grid_dim(const ora_grid *g) { return g->dim; }
size_t ora_grid_size(const ora_grid *g) { return g->total; }
const int *ora_grid_number(const ora_grid *g) { return g->n; }
const double *ora_grid_dx(const ora_grid *g) { return g->dx; }
const double *ora_grid_min(const ora_grid *g) { return g->min; }
const double *ora_grid_max(const ora_grid *g) { return g->max; }
const int *ora_grid_periodic(const ora_grid *g) { return g->periodic; }
int ora_grid_has_deriv(const ora_grid *g) { return g->b_deriv; }
double *ora_grid_values(ora_grid *g) { return g->v; }
double *ora_grid_derivs(ora_grid *g) { return g->dv; }
void ora_grid_set_interpolation(ora_grid *g, int b) { g->b_interp = b; }

/* grid.h:264-273 */
void ora_grid_get_index(const ora_grid *g, const double *x, size_t *out) {
  int d;
  for (d = 0; d < g->dim; d++) {
    double xi = x[d];
    if (g->periodic[d])
      xi -= (g->max[d] - g->min[d]) * ifloor((xi - g->min[d]) / (g->max[d] - g->min[d]));
    out[d] = to_index(floor((xi - g->min[d]) / g->dx[d]));
  }
}

/* grid.h:315-325: dimension 0 runs fastest */
size_t ora_grid_multi2one(const ora_grid *g, const size_t *idx) {
  size_t r = idx[g->dim - 1];
  int d;
  for (d = g->dim - 1; d > 0; d--) r = r * (size_t)g->n[d - 1] + idx[d - 1];
  return r;
}

/* grid.h:330-338 */
void ora_grid_one2multi(const ora_grid *g, size_t index, size_t *out) {
  int d;
  for (d = 0; d < g->dim - 1; d++) {
    out[d] = index % (size_t)g->n[d];
    index = (index - out[d]) / (size_t)g->n[d];
  }
  out[d] = index;
}

/* grid.h:865-874 */
int ora_grid_in_grid(const ora_grid *g, const double *x) {
  int d;
  for (d = 0; d < g->dim; d++)
    if (!g->periodic[d] && (x[d] < g->min[d] || x[d] >= g->max[d] - g->dx[d])) return 0;
  return 1;
}

/* grid.h:52-139: product-of-cubic-Hermite blend over the 2^dim corners, with
 * the derivative term dropped wherever |f| < 1e-7 (grid.h:113-116). */
static double cubic_blend(int dim, const double *dx, const double *where,
                          const double *tabf, const double *tabder,
                          const long *stride, double *der) {
  int corner, ncorner = 1 << dim, d, e;
  double f = 0;
  for (d = 0; d < dim; d++) der[d] = 0;
  for (corner = 0; corner < ncorner; corner++) {
    int bit[EDM_MAXDIM];
    long shift = 0;
    double C[EDM_MAXDIM], D[EDM_MAXDIM], fd[EDM_MAXDIM];
    double ff = 1.0;
    int tmp = corner;
    for (d = 0; d < dim; d++) {
      bit[d] = tmp % 2;
      tmp /= 2;
      shift += stride[d] * bit[d];
    }
    for (d = 0; d < dim; d++) {
      double X = fabs(where[d] / dx[d] - bit[d]);
      double X2 = X * X;
      double X3 = X2 * X;
      double qq;
      int sgn = bit[d] ? -1 : 1;
      if (fabs(tabf[shift]) < 0.0000001)
        qq = 0.0;
      else
        qq = -tabder[shift * dim + d] / tabf[shift];
      C[d] = (1 - 3 * X2 + 2 * X3) - sgn * qq * (X - 2 * X2 + X3) * dx[d];
      D[d] = (-6 * X + 6 * X2) - sgn * qq * (1 - 4 * X + 3 * X2) * dx[d];
      D[d] *= sgn / dx[d];
      ff *= C[d];
    }
    for (d = 0; d < dim; d++) {
      fd[d] = D[d];
      for (e = 0; e < dim; e++)
        if (e != d) fd[d] *= C[e];
    }
    f += tabf[shift] * ff;
    for (d = 0; d < dim; d++) der[d] += tabf[shift] * fd[d];
  }
  return f;
}

/* grid.h:390-446 */
double ora_grid_get_value_deriv(const ora_grid *g, const double *x, double *der) {
  size_t idx[EDM_MAXDIM], flat;
  int d;
  if (!ora_grid_in_grid(g, x)) {
    for (d = 0; d < g->dim; d++) der[d] = 0;
    return 0;
  }
  ora_grid_get_index(g, x, idx);
  flat = ora_grid_multi2one(g, idx);
  if (g->b_interp) {
    double where[EDM_MAXDIM];
    long stride[EDM_MAXDIM];
    stride[0] = 1;
    for (d = 1; d < g->dim; d++) stride[d] = stride[d - 1] * g->n[d - 1];
    for (d = 0; d < g->dim; d++) {
      double wx = x[d];
      if (g->periodic[d])
        wx -= (g->max[d] - g->min[d]) * ifloor((wx - g->min[d]) / (g->max[d] - g->min[d]));
      where[d] = wx - g->min[d] - idx[d] * g->dx[d];
      if (g->periodic[d] && idx[d] == (size_t)(g->n[d] - 1)) stride[d] *= (1 - g->n[d]);
    }
    return cubic_blend(g->dim, g->dx, where, &g->v[flat], &g->dv[flat * (size_t)g->dim], stride, der);
  }
  for (d = 0; d < g->dim; d++) der[d] = g->dv[flat * (size_t)g->dim + d];
  return g->v[flat];
}

/* grid.h:343-365 */
double ora_grid_get_value(const ora_grid *g, const double *x) {
  size_t idx[EDM_MAXDIM];
  if (!ora_grid_in_grid(g, x)) return 0;
  if (g->b_interp && g->b_deriv) {
    double tmp[EDM_MAXDIM];
    return ora_grid_get_value_deriv(g, x, tmp);
  }
  ora_grid_get_index(g, x, idx);
  return g->v[ora_grid_multi2one(g, idx)];
}

/* grid.h:370-385 */
double ora_grid_add_value(ora_grid *g, const double *x, double value) {
  size_t idx[EDM_MAXDIM];
  if (g->b_interp) return -1e300; /* reference aborts here */
  if (!ora_grid_in_grid(g, x)) return 0;
  ora_grid_get_index(g, x, idx);
  g->v[ora_grid_multi2one(g, idx)] += value;
  return value;
}

/* grid.h:679-688 */
void ora_grid_clear(ora_grid *g) {
  size_t i;
  int d;
  for (i = 0; i < g->total; i++) {
    g->v[i] = 0;
    if (g->b_deriv)
      for (d = 0; d < g->dim; d++) g->dv[i * (size_t)g->dim + d] = 0;
  }
}

/* grid.h:292-309 */
double ora_grid_max_value(const ora_grid *g) {
  double m = g->v[0];
  size_t i;
  for (i = 0; i < g->total; i++) m = fmax(m, g->v[i]);
  return m;
}
double ora_grid_min_value(const ora_grid *g) {
  double m = g->v[0];
  size_t i;
  for (i = 0; i < g->total; i++) m = fmin(m, g->v[i]);
  return m;
}

/* grid.h:692-710 */
double ora_grid_expected_bias(const ora_grid *g) {
  double Z = 0, offset = 0, avg = 0;
  size_t i;
  for (i = 0; i < g->total; i++) offset = fmax(offset, g->v[i]);
  for (i = 0; i < g->total; i++) Z += exp(-g->v[i] - offset);
  for (i = 0; i < g->total; i++) avg += g->v[i] * exp(-g->v[i] - offset);
  return avg / Z;
}

/* grid.h:275-290 */
void ora_grid_add_grid(ora_grid *g, const ora_grid *other, double scale, double offset) {
  size_t i, idx[EDM_MAXDIM];
  double x[EDM_MAXDIM], der[EDM_MAXDIM];
  int d;
  for (i = 0; i < g->total; i++) {
    ora_grid_one2multi(g, i, idx);
    for (d = 0; d < g->dim; d++) x[d] = g->min[d] + g->dx[d] * idx[d];
    g->v[i] += scale * ora_grid_get_value_deriv(other, x, der) + offset;
    for (d = 0; d < g->dim; d++) g->dv[i * (size_t)g->dim + d] += scale * der[d];
  }
}

/* PLUMED-1 style header shared by write and multi_write */
static void put_header(FILE *fp, int b_deriv, int dim, const long *bins,
                       const double *lo, const double *hi, const int *pbc) {
  int d;
  fprintf(fp, "#! FORCE %d\n", b_deriv);
  fprintf(fp, "#! NVAR %d\n", dim);
  fprintf(fp, "#! TYPE ");
  for (d = 0; d < dim; d++) fprintf(fp, "%d ", GRID_TYPE);
  fprintf(fp, "\n#! BIN ");
  for (d = 0; d < dim; d++) fprintf(fp, "%ld ", bins[d]);
  fprintf(fp, "\n#! MIN ");
  for (d = 0; d < dim; d++) fprintf(fp, "%g ", lo[d]);
  fprintf(fp, "\n#! MAX ");
  for (d = 0; d < dim; d++) fprintf(fp, "%g ", hi[d]);
  fprintf(fp, "\n#! PBC ");
  for (d = 0; d < dim; d++) fprintf(fp, "%d ", pbc[d]);
  fprintf(fp, "\n");
}

/* grid.h:448-503 */
void ora_grid_write(const ora_grid *g, const char *filename) {
  FILE *fp = fopen(filename, "w");
  long bins[EDM_MAXDIM];
  double hi[EDM_MAXDIM];
  size_t i, idx[EDM_MAXDIM];
  int d;
  if (!fp) return;
  for (d = 0; d < g->dim; d++) {
    bins[d] = g->periodic[d] ? g->n[d] : g->n[d] - 1;
    hi[d] = g->periodic[d] ? g->max[d] : g->max[d] - g->dx[d];
  }
  put_header(fp, g->b_deriv, g->dim, bins, g->min, hi, g->periodic);
  for (i = 0; i < g->total; i++) {
    ora_grid_one2multi(g, i, idx);
    for (d = 0; d < g->dim; d++) fprintf(fp, "%.8f ", g->min[d] + g->dx[d] * idx[d]);
    fprintf(fp, "%.8f ", g->v[i]);
    if (g->b_deriv)
      for (d = 0; d < g->dim; d++) fprintf(fp, "%.8f ", -g->dv[i * (size_t)g->dim + d]);
    fprintf(fp, "\n");
    if (idx[0] == (size_t)(g->n[0] - 1)) fprintf(fp, "\n");
  }
  fclose(fp);
}

/* grid.h:509-674 evaluated for a single rank: the rank always wins the
 * MPI_MAX election, so a point is written iff it is in_grid. */
void ora_grid_multi_write(const ora_grid *g, const char *filename,
                          const double *box_min, const double *box_max,
                          const int *b_periodic, int b_lammps_format) {
  FILE *fp;
  unsigned int counts[EDM_MAXDIM];
  unsigned int extra_n = 0;
  size_t i, total = 1, sup[EDM_MAXDIM], tmp;
  double x[EDM_MAXDIM], der[EDM_MAXDIM], value;
  int d;
  if (b_lammps_format == 1 && g->dim > 1) ora_error("Lammps format only valid for 1D grids", "grid.h:multi_write");
  if (b_lammps_format) extra_n = (unsigned int)(box_min[0] / g->dx[0]);
  for (d = 0; d < g->dim; d++) {
    counts[d] = (unsigned int)(int)ceil((box_max[d] - box_min[d]) / g->dx[d]);
    counts[d] = b_periodic[d] ? counts[d] : counts[d] + 1;
  }
  fp = fopen(filename, "w");
  if (!fp) return;
  if (!b_lammps_format) {
    long bins[EDM_MAXDIM];
    for (d = 0; d < g->dim; d++) bins[d] = b_periodic[d] ? (long)counts[d] : (long)counts[d] - 1;
    put_header(fp, g->b_deriv, g->dim, bins, box_min, box_max, b_periodic);
  } else {
    fprintf(fp, "#Auto generated by electronic-dance-music\n\n");
    fprintf(fp, "EDM\n");
    fprintf(fp, "N %u R %g %g\n\n", extra_n + counts[0], g->dx[0], box_max[0]);
    for (i = 1; i < extra_n; i++) fprintf(fp, "%zu %g 0.0 0.0\n", i, i * g->dx[0]);
  }
  for (d = 0; d < g->dim; d++) total *= counts[d];
  for (i = 0; i < total; i++) {
    tmp = i;
    for (d = 0; d < g->dim - 1; d++) {
      sup[d] = tmp % counts[d];
      tmp = (tmp - sup[d]) / counts[d];
      x[d] = sup[d] * g->dx[d] + box_min[d];
    }
    sup[d] = tmp;
    x[d] = sup[d] * g->dx[d] + box_min[d];
    if (!ora_grid_in_grid(g, x)) continue;
    if (b_lammps_format) fprintf(fp, "%zu ", i + extra_n);
    for (d = 0; d < g->dim; d++) fprintf(fp, "%.8f ", x[d]);
    if (g->b_deriv)
      value = ora_grid_get_value_deriv(g, x, der);
    else
      value = ora_grid_get_value(g, x);
    fprintf(fp, "%.8f ", value);
    if (g->b_deriv)
      for (d = 0; d < g->dim; d++) fprintf(fp, "%.8f ", -der[d]);
    fprintf(fp, "\n");
    if (sup[0] == counts[0] - 1) fprintf(fp, "\n");
  }
  fclose(fp);
}

static int next_word(FILE *fp, char *buf, size_t cap) {
  char fmt[32];
  snprintf(fmt, sizeof fmt, "%%%zus", cap - 1);
  return fscanf(fp, fmt, buf) == 1;
}

/* grid.h:712-835 */
static void grid_read_into(ora_grid *g, const char *filename) {
  FILE *fp = fopen(filename, "r");
  char w[256];
  int d;
  size_t i;
  if (!fp) {
    fprintf(stderr, "Cannot open input file \"%s\"\n", filename);
    ora_error("", "grid.h:read");
  }
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "FORCE") != 0) ora_error("Mangled grid file: no FORCE", "grid.h:read");
  if (fscanf(fp, "%d", &g->b_deriv) != 1) ora_error("bad FORCE", "grid.h:read");
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "NVAR") == 0) {
    int nv = 0;
    if (fscanf(fp, "%d", &nv) != 1 || nv != g->dim)
      ora_error("Dimension of this grid does not match the one found in the file", "grid.h:read");
  }
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "TYPE") != 0) ora_error("Mangled grid file: no TYPE", "grid.h:read");
  for (d = 0; d < g->dim; d++) {
    int t;
    if (fscanf(fp, "%d", &t) != 1) ora_error("bad TYPE", "grid.h:read");
  }
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "BIN") != 0) ora_error("Mangled grid file: no BIN", "grid.h:read");
  for (d = 0; d < g->dim; d++)
    if (fscanf(fp, "%d", &g->n[d]) != 1) ora_error("bad BIN", "grid.h:read");
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "MIN") != 0) ora_error("Mangled grid file: no MIN", "grid.h:read");
  for (d = 0; d < g->dim; d++)
    if (fscanf(fp, "%lf", &g->min[d]) != 1) ora_error("bad MIN", "grid.h:read");
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "MAX") != 0) ora_error("Mangled grid file: no MAX", "grid.h:read");
  for (d = 0; d < g->dim; d++)
    if (fscanf(fp, "%lf", &g->max[d]) != 1) ora_error("bad MAX", "grid.h:read");
  next_word(fp, w, sizeof w);
  next_word(fp, w, sizeof w);
  if (strcmp(w, "PBC") != 0) ora_error("Mangled grid file: no PBC", "grid.h:read");
  for (d = 0; d < g->dim; d++)
    if (fscanf(fp, "%d", &g->periodic[d]) != 1) ora_error("bad PBC", "grid.h:read");
  for (d = 0; d < g->dim; d++) {
    g->dx[d] = (g->max[d] - g->min[d]) / g->n[d];
    if (!g->periodic[d]) {
      g->max[d] += g->dx[d];
      g->n[d] += 1;
    }
  }
  grid_alloc(g);
  for (i = 0; i < g->total; i++) {
    for (d = 0; d < g->dim; d++) next_word(fp, w, sizeof w);
    if (fscanf(fp, "%lf", &g->v[i]) != 1) g->v[i] = 0;
    if (g->b_deriv)
      for (d = 0; d < g->dim; d++) {
        double t = 0;
        if (fscanf(fp, "%lf", &t) != 1) t = 0;
        g->dv[i * (size_t)g->dim + d] = t;
        g->dv[i * (size_t)g->dim + d] *= -1;
      }
  }
  fclose(fp);
}

/* grid.h:218-227 */
ora_grid *ora_grid_read(int dim, const char *filename, int b_interpolate) {
  ora_grid *g = (ora_grid *)calloc(1, sizeof(ora_grid));
  g->dim = dim;
  g->b_deriv = 0;
  g->b_interp = b_interpolate;
  grid_read_into(g, filename);
  return g;
}

/* ====================================================================== */
/* gaussian grid                                                          */
/* ====================================================================== */
struct ora_gauss {
  ora_grid *grid;
  int dim;
  size_t minisize[EDM_MAXDIM];
  size_t minisize_total;
  double sigma[EDM_MAXDIM]; /* = user sigma * sqrt(2) */
  double bmin[EDM_MAXDIM], bmax[EDM_MAXDIM];
  int bper[EDM_MAXDIM];
  double *denom_tab[EDM_MAXDIM];
  double *dderiv_tab[EDM_MAXDIM];
  int dirty;
};

/* gaussian_grid.h:559-569 */
static void gauss_update_stencil(ora_gauss *g) {
  int d;
  g->minisize_total = 1;
  for (d = 0; d < g->dim; d++) {
    double dist = sqrt(2 * GAUSS_SUPPORT) * g->sigma[d];
    g->minisize[d] = (size_t)ifloor(dist / g->grid->dx[d]);
    g->minisize_total *= (2 * g->minisize[d] + 1);
  }
}

/* gaussian_grid.h:378-435 */
void ora_gauss_set_boundary(ora_gauss *g, const double *min, const double *max,
                            const int *periodic) {
  int d;
  size_t j;
  g->dirty = 0;
  for (d = 0; d < g->dim; d++) {
    g->bmin[d] = min[d];
    g->bmax[d] = max[d];
    g->bper[d] = periodic[d];
  }
  for (d = 0; d < g->dim; d++) {
    double lo, hi, sg;
    if (g->bper[d]) continue;
    lo = g->bmin[d];
    hi = g->bmax[d];
    sg = g->sigma[d];
    for (j = 0; j < BC_TABLE_SIZE; j++) {
      double s = j * (hi - lo) / (BC_TABLE_SIZE - 1) + lo;
      double t1 = sqrt(M_PI) * sg / 2. * (erf((s - lo) / sg) + erf((hi - s) / sg));
      double t2 = sqrt(M_PI) * sg / 2. * erf((hi - lo) / sg);
      double t3;
      g->denom_tab[d][j] = t1;
      g->denom_tab[d][j] += (t2 - t1) * smooth_step((s - lo) / (BC_MAR * sg));
      g->denom_tab[d][j] += (t2 - t1) * smooth_step((hi - s) / (BC_MAR * sg));
      t3 = 1. * (exp(-((s - lo) * (s - lo)) / (sg * sg)) - exp(-((hi - s) * (hi - s)) / (sg * sg)));
      g->dderiv_tab[d][j] = t3;
      g->dderiv_tab[d][j] += (t2 - t1) * smooth_step_dt((s - lo) / (BC_MAR * sg)) / (BC_MAR * sg) -
                             t3 * smooth_step((s - lo) / (BC_MAR * sg));
      g->dderiv_tab[d][j] += -(t2 - t1) * smooth_step_dt((hi - s) / (BC_MAR * sg)) / (BC_MAR * sg) -
                             t3 * smooth_step((hi - s) / (BC_MAR * sg));
    }
  }
}

static ora_gauss *gauss_shell(int dim) {
  int d;
  ora_gauss *g = (ora_gauss *)calloc(1, sizeof(ora_gauss));
  g->dim = dim;
  for (d = 0; d < EDM_MAXDIM; d++) {
    g->denom_tab[d] = (double *)calloc(BC_TABLE_SIZE, sizeof(double));
    g->dderiv_tab[d] = (double *)calloc(BC_TABLE_SIZE, sizeof(double));
  }
  return g;
}

/* gaussian_grid.h:65-80 */
ora_gauss *ora_gauss_create(int dim, const double *min, const double *max,
                            const double *spacing, const int *periodic,
                            int b_interpolate, const double *sigma) {
  int d;
  ora_gauss *g = gauss_shell(dim);
  g->grid = ora_grid_create(dim, min, max, spacing, periodic, 1, b_interpolate);
  for (d = 0; d < dim; d++) g->sigma[d] = sigma[d] * sqrt(2.);
  ora_gauss_set_boundary(g, min, max, periodic);
  gauss_update_stencil(g);
  return g;
}

/* gaussian_grid.h:85-93 */
ora_gauss *ora_gauss_read(int dim, const char *filename, const double *sigma) {
  int d;
  ora_gauss *g = gauss_shell(dim);
  g->grid = ora_grid_read(dim, filename, 1);
  for (d = 0; d < dim; d++) g->sigma[d] = sigma[d] * sqrt(2.);
  ora_gauss_set_boundary(g, g->grid->min, g->grid->max, g->grid->periodic);
  gauss_update_stencil(g);
  return g;
}

void ora_gauss_free(ora_gauss *g) {
  int d;
  if (!g) return;
  for (d = 0; d < EDM_MAXDIM; d++) {
    free(g->denom_tab[d]);
    free(g->dderiv_tab[d]);
  }
  ora_grid_free(g->grid);
  free(g);
}

ora_grid *ora_gauss_grid(ora_gauss *g) { return g->grid; }
const double *ora_gauss_sigma(const ora_gauss *g) { return g->sigma; }
const size_t *ora_gauss_minisize(const ora_gauss *g) { return g->minisize; }
size_t ora_gauss_minisize_total(const ora_gauss *g) { return g->minisize_total; }
const double *ora_gauss_bc_table(const ora_gauss *g, int d, int deriv) {
  return deriv ? g->dderiv_tab[d] : g->denom_tab[d];
}
const double *ora_gauss_boundary_min(const ora_gauss *g) { return g->bmin; }
const double *ora_gauss_boundary_max(const ora_gauss *g) { return g->bmax; }
const int *ora_gauss_boundary_periodic(const ora_gauss *g) { return g->bper; }
void ora_gauss_write(const ora_gauss *g, const char *filename) { ora_grid_write(g->grid, filename); }

/* gaussian_grid.h:151-157 */
void ora_gauss_multi_write(const ora_gauss *g, const char *filename, int b_lammps_format) {
  ora_grid_multi_write(g->grid, filename, g->bmin, g->bmax, g->bper, b_lammps_format);
}

/* gaussian_grid.h:437-444 */
double ora_gauss_get_volume(const ora_gauss *g) {
  double vol = 1;
  int d;
  for (d = 0; d < g->dim; d++) vol *= g->bmax[d] - g->bmin[d];
  return vol;
}

/* gaussian_grid.h:490-499 (closed interval on the boundary) */
int ora_gauss_in_bounds(const ora_gauss *g, const double *x) {
  int d;
  for (d = 0; d < g->dim; d++)
    if (x[d] < g->bmin[d] || x[d] > g->bmax[d]) return 0;
  return 1;
}

/* gaussian_grid.h:504-541: nearest image (to the grid), not minimal image */
void ora_gauss_remap(const ora_gauss *g, double *x) {
  const ora_grid *q = g->grid;
  int d;
  for (d = 0; d < g->dim; d++) {
    if (x[d] < q->min[d] || x[d] > q->max[d]) {
      if (q->periodic[d]) {
        x[d] -= (q->max[d] - q->min[d]) * ifloor((x[d] - q->min[d]) / (q->max[d] - q->min[d]));
      } else if (g->bper[d]) {
        double period = g->bmax[d] - g->bmin[d];
        double s0 = round_half((q->min[d] - x[d]) / (g->bmax[d] - g->bmin[d])) * period;
        double s1 = round_half((q->max[d] - x[d]) / (g->bmax[d] - g->bmin[d])) * period;
        if (fabs(q->min[d] - x[d] - s0) < fabs(q->max[d] - x[d] - s1))
          x[d] += s0;
        else
          x[d] += s1;
      }
    }
  }
}

/* gaussian_grid.h:99-116 */
double ora_gauss_get_value(const ora_gauss *g, const double *x) {
  double xx[EDM_MAXDIM];
  int d;
  for (d = 0; d < g->dim; d++) xx[d] = x[d];
  if (!ora_gauss_in_bounds(g, xx)) {
    ora_gauss_remap(g, xx);
    if (!ora_gauss_in_bounds(g, xx)) return 0;
  }
  return ora_grid_get_value(g->grid, xx);
}

/* gaussian_grid.h:118-138 */
double ora_gauss_get_value_deriv(const ora_gauss *g, const double *x, double *der) {
  double xx[EDM_MAXDIM];
  int d;
  for (d = 0; d < g->dim; d++) xx[d] = x[d];
  if (!ora_gauss_in_bounds(g, xx)) {
    ora_gauss_remap(g, xx);
    if (!ora_gauss_in_bounds(g, xx)) {
      for (d = 0; d < g->dim; d++) der[d] = 0;
      return 0;
    }
  }
  return ora_grid_get_value_deriv(g->grid, xx, der);
}

/* gaussian_grid.h:571-630: copies the VALUE of the first/last in-boundary
 * node into its outward neighbour, for the 4^dim index combinations only. */
static void gauss_duplicate_boundary(ora_gauss *g) {
  ora_grid *q = g->grid;
  size_t lo_i[EDM_MAXDIM], hi_i[EDM_MAXDIM], outer[EDM_MAXDIM], inner[EDM_MAXDIM];
  size_t combos = 1, c;
  int d;
  ora_grid_get_index(q, g->bmin, lo_i);
  ora_grid_get_index(q, g->bmax, hi_i);
  for (d = 0; d < g->dim; d++) {
    while (lo_i[d] * q->dx[d] + q->min[d] < g->bmin[d]) lo_i[d] += 1;
    while (hi_i[d] * q->dx[d] + q->min[d] > g->bmax[d] || hi_i[d] == (size_t)q->n[d]) hi_i[d] -= 1;
  }
  for (d = 0; d < g->dim; d++) combos *= 4;
  for (c = 0; c < combos; c++) {
    int skip = 0, oob = 0;
    size_t tmp = c;
    for (d = 0; d < g->dim; d++) {
      int which = (int)(tmp % 4);
      tmp = (tmp - (size_t)which) / 4;
      switch (which) {
        case 0:
          skip |= g->bper[d];
          skip |= (lo_i[d] == 0);
          outer[d] = lo_i[d] - 1;
          inner[d] = lo_i[d];
          break;
        case 1:
          outer[d] = lo_i[d];
          inner[d] = lo_i[d];
          break;
        case 2:
          outer[d] = hi_i[d];
          inner[d] = hi_i[d];
          break;
        default:
          skip |= g->bper[d];
          skip |= (hi_i[d] == (size_t)(q->n[d] - 1));
          outer[d] = hi_i[d] + 1;
          inner[d] = hi_i[d];
          break;
      }
    }
    if (skip) continue;
    for (d = 0; d < g->dim; d++)
      if (outer[d] >= (size_t)q->n[d] || inner[d] >= (size_t)q->n[d]) oob = 1;
    if (oob) continue; /* reference would write out of bounds */
    q->v[ora_grid_multi2one(q, outer)] = q->v[ora_grid_multi2one(q, inner)];
  }
}

/* gaussian_grid.h:176-372 */
double ora_gauss_add_value(ora_gauss *g, const double *x0, double height);
/* n calls of add_value in one C call (bench.py's CPU baseline: no per-hill FFI overhead) */
double ora_gauss_add_values(ora_gauss *g, long long n, const double *x, int stride, double height) {
  double total = 0;
  long long i;
  for (i = 0; i < n; i++) total += ora_gauss_add_value(g, x + i * stride, height);
  return total;
}
double ora_gauss_add_value(ora_gauss *g, const double *x0, double height) {
  ora_grid *q = g->grid;
  const int dim = g->dim;
  double x[EDM_MAXDIM], xx[EDM_MAXDIM], dp[EDM_MAXDIM], force[EDM_MAXDIM];
  int centre[EDM_MAXDIM], off[EDM_MAXDIM];
  size_t node[EDM_MAXDIM];
  double vol = 1, added = 0;
  size_t s;
  int d;

  for (d = 0; d < dim; d++) vol *= q->dx[d];
  for (d = 0; d < dim; d++) x[d] = x0[d];
  ora_gauss_remap(g, x);
  for (d = 0; d < dim; d++)
    if (!g->bper[d] && (x[d] < g->bmin[d] || x[d] > g->bmax[d])) return 0;
  for (d = 0; d < dim; d++) centre[d] = ifloor((x[d] - q->min[d]) / q->dx[d]);

  for (s = 0; s < g->minisize_total; s++) {
    int rest = (int)s, outside = 0;
    double dp2 = 0, expo, denom, corr;
    size_t flat;
    for (d = 0; d < dim - 1; d++) {
      off[d] = (int)((size_t)rest % (2 * g->minisize[d] + 1));
      rest = (int)((rest - off[d]) / (long)(2 * g->minisize[d] + 1));
    }
    off[d] = rest;
    for (d = 0; d < dim; d++) off[d] -= (int)g->minisize[d];

    for (d = 0; d < dim; d++) {
      off[d] += centre[d];
      if (off[d] >= q->n[d]) {
        if (q->periodic[d]) {
          off[d] %= q->n[d];
        } else {
          outside = 1;
          break;
        }
      }
      if (off[d] < 0) {
        if (q->periodic[d]) {
          off[d] += q->n[d];
        } else {
          outside = 1;
          break;
        }
      }
      node[d] = (size_t)off[d];
      xx[d] = q->min[d] + q->dx[d] * node[d];
      if (!g->bper[d] && (xx[d] < g->bmin[d] || xx[d] > g->bmax[d])) {
        outside = 1;
        break;
      }
    }
    if (outside) continue;

    for (d = 0; d < dim; d++) {
      dp[d] = xx[d] - x[d];
      if (q->periodic[d]) dp[d] -= round_half(dp[d] / (q->max[d] - q->min[d])) * (q->max[d] - q->min[d]);
      dp[d] /= g->sigma[d];
      dp2 += dp[d] * dp[d];
    }
    if (!(dp2 < GAUSS_SUPPORT)) continue;

    expo = exp(-dp2);
    denom = 1.0;
    corr = 0;
    for (d = 0; d < dim; d++) {
      if (!g->bper[d]) {
        const double sg = g->sigma[d];
        size_t ti = to_index((BC_TABLE_SIZE - 1) * (xx[d] - g->bmin[d]) / (g->bmax[d] - g->bmin[d]));
        double t1 = exp(-((x[d] - g->bmin[d]) * (x[d] - g->bmin[d])) / (sg * sg));
        double t2 = smooth_step((xx[d] - g->bmin[d]) / (sg * BC_MAR));
        double t3 = exp(-((x[d] - g->bmax[d]) * (x[d] - g->bmax[d])) / (sg * sg));
        double t4 = smooth_step((g->bmax[d] - xx[d]) / (sg * BC_MAR));
        double t5, t6, t7;
        corr = (t1 - expo) * t2 + (t3 - expo) * t4; /* overwritten per dim, not accumulated */
        denom *= g->denom_tab[d][ti];
        t5 = -2 * dp[d] / sg;
        t6 = smooth_step_dt((xx[d] - g->bmin[d]) / (sg * BC_MAR)) / (BC_MAR * sg);
        t7 = -smooth_step_dt((g->bmax[d] - xx[d]) / (sg * BC_MAR)) / (BC_MAR * sg);
        force[d] = t5 * expo;
        force[d] += (t1 - expo) * t6 - t5 * expo * t2 + (t3 - expo) * t7 - t5 * expo * t4;
        force[d] = force[d] * denom - g->dderiv_tab[d][ti] * (expo + corr);
        force[d] /= denom * denom;
        corr /= denom;
      } else {
        denom *= sqrt(M_PI) * g->sigma[d];
      }
    }
    expo /= denom;

    flat = ora_grid_multi2one(q, node);
    q->v[flat] += height * (expo + corr);
    added += height * (expo + corr) * vol;
    for (d = 0; d < dim; d++) {
      if (g->bper[d])
        q->dv[flat * (size_t)dim + d] -= height * (2 * dp[d] / g->sigma[d] * expo);
      else
        q->dv[flat * (size_t)dim + d] += height * force[d];
    }
    if (!g->dirty && corr * corr > 0) g->dirty = 1;
  }

  if (g->dirty) {
    gauss_duplicate_boundary(g);
    g->dirty = 0;
  }
  return added;
}

/* ====================================================================== */
/* bias controller (serial build: EDM_SERIAL, lib/CMakeLists.txt:1)        */
/* ====================================================================== */
struct ora_bias {
  int b_tempering, b_targeting;
  int mpi_rank, mpi_size;
  unsigned int dim;
  double global_tempering, bias_factor, boltzmann_factor, temperature;
  double hill_prefactor, bias_per_step, hill_density, cum_bias, total_volume;
  double expected_target;
  int b_outofbounds;
  double *bias_dx, *bias_sigma, *min, *max;
  int *bper;
  ora_grid *target, *initial_bias, *hist;
  ora_gauss *bias;
  const int *mask;
  double temp_hill_cum, temp_hill_prefactor;
  int est_hill_count;
  int hills_added;
  long long steps;
  char hist_name[1024];
  FILE *hills_fp;
  double overflow[BIAS_BUFFER_DBLS + EDM_MAXDIM + 1];
  size_t overflow_left, overflow_right;
  int b_skip_hill_add;
};

/* --- config file: "key rest-of-line" pairs, first occurrence wins
 * (edm_bias.cpp:19-24, :997-1004) --- */
typedef struct {
  char key[128];
  char val[1024];
} cfg_pair;

static const char *cfg_find(const cfg_pair *p, int n, const char *key) {
  int i;
  for (i = 0; i < n; i++)
    if (strcmp(p[i].key, key) == 0) return p[i].val;
  return NULL;
}

/* edm_bias.cpp:933-950: a value that parses to exactly 0 is rejected */
static int cfg_double(const cfg_pair *p, int n, const char *key, int required, double *out) {
  const char *v = cfg_find(p, n, key);
  if (v) {
    *out = atof(v);
    if (*out == 0.0) {
      fprintf(stderr, "Invalid value found for %s\n", key);
      return 0;
    }
    return 1;
  }
  if (required) fprintf(stderr, "Could not find key %s\n", key);
  return 0;
}

/* edm_bias.cpp:952-966 */
static int cfg_double_array(const cfg_pair *p, int n, const char *key, int required, double *out, int len) {
  const char *v = cfg_find(p, n, key);
  int i;
  if (v) {
    const char *cur = v;
    for (i = 0; i < len; i++) {
      char *end;
      double t = strtod(cur, &end);
      if (end == cur) break;
      out[i] = t;
      cur = end;
    }
    return 1;
  }
  if (required) fprintf(stderr, "Could not find key %s\n", key);
  return 0;
}

/* edm_bias.cpp:968-979 */
static int cfg_int(const cfg_pair *p, int n, const char *key, int required, int *out) {
  const char *v = cfg_find(p, n, key);
  if (v) {
    *out = atoi(v);
    return 1;
  }
  if (required) fprintf(stderr, "Could not find key %s\n", key);
  return 0;
}

/* edm_bias.cpp:1098-1111: strips LEADING blanks/tabs only */
static void cfg_clean(const char *in, int append_rank, int rank, char *out, size_t cap) {
  size_t k = strspn(in, " \t");
  if (in[k] == '\0') k = 0;
  if (append_rank)
    snprintf(out, cap, "%s_%d", in + k, rank);
  else
    snprintf(out, cap, "%s", in + k);
}

/* edm_bias.cpp:986-1095 */
static int bias_read_input(ora_bias *b, const char *filename) {
  FILE *fp = fopen(filename, "r");
  cfg_pair *pairs;
  int npairs = 0, cap = 64, tmp;
  char key[128], name[1024];
  const char *v;
  if (!fp) {
    fprintf(stderr, "Cannot open input file %s\n", filename);
    return 0;
  }
  pairs = (cfg_pair *)calloc((size_t)cap, sizeof(cfg_pair));
  while (fscanf(fp, "%127s", key) == 1) {
    char line[1024];
    size_t len;
    int got_any = 0, c;
    len = 0;
    while ((c = fgetc(fp)) != EOF) {
      got_any = 1;
      if (c == '\n') break;
      if (len + 1 < sizeof line) line[len++] = (char)c;
    }
    line[len] = '\0';
    if (!got_any) break; /* getline at EOF fails: the pair is dropped */
    if (cfg_find(pairs, npairs, key)) continue;
    if (npairs == cap) {
      cap *= 2;
      pairs = (cfg_pair *)realloc(pairs, (size_t)cap * sizeof(cfg_pair));
    }
    snprintf(pairs[npairs].key, sizeof pairs[npairs].key, "%s", key);
    snprintf(pairs[npairs].val, sizeof pairs[npairs].val, "%s", line);
    npairs++;
  }
  fclose(fp);

  if (!cfg_int(pairs, npairs, "tempering", 1, &b->b_tempering)) goto fail;
  if (b->b_tempering) {
    if (!cfg_double(pairs, npairs, "bias_factor", 1, &b->bias_factor)) goto fail;
    cfg_double(pairs, npairs, "global_tempering", 0, &b->global_tempering);
  }
  if (!cfg_double(pairs, npairs, "hill_prefactor", 1, &b->hill_prefactor)) goto fail;
  if (!cfg_double(pairs, npairs, "bias_per_step", 0, &b->bias_per_step)) b->bias_per_step = b->hill_prefactor;
  cfg_double(pairs, npairs, "hill_density", 0, &b->hill_density);
  if (!cfg_int(pairs, npairs, "dimension", 1, &tmp)) goto fail;
  b->dim = (unsigned int)tmp;
  if (b->dim == 0 || b->dim > 3) {
    fprintf(stderr, "Invalid dimesion %u\n", b->dim);
    goto fail;
  }
  b->bias_dx = (double *)calloc(b->dim, sizeof(double));
  b->bias_sigma = (double *)calloc(b->dim, sizeof(double));
  b->min = (double *)calloc(b->dim, sizeof(double));
  b->max = (double *)calloc(b->dim, sizeof(double));
  b->bper = (int *)calloc(b->dim, sizeof(int));
  if (!cfg_double_array(pairs, npairs, "bias_spacing", 1, b->bias_dx, (int)b->dim)) goto fail;
  if (!cfg_double_array(pairs, npairs, "bias_sigma", 1, b->bias_sigma, (int)b->dim)) goto fail;
  if (!cfg_double_array(pairs, npairs, "box_low", 1, b->min, (int)b->dim)) goto fail;
  if (!cfg_double_array(pairs, npairs, "box_high", 1, b->max, (int)b->dim)) goto fail;

  v = cfg_find(pairs, npairs, "target_filename");
  if (!v) {
    b->b_targeting = 0;
    b->expected_target = 0;
  } else {
    b->b_targeting = 1;
    cfg_clean(v, 0, 0, name, sizeof name);
    b->target = ora_grid_read((int)b->dim, name, 0);
    b->expected_target = ora_grid_expected_bias(b->target);
  }
  v = cfg_find(pairs, npairs, "initial_bias_filename");
  if (!v) {
    b->initial_bias = NULL;
  } else {
    cfg_clean(v, 0, 0, name, sizeof name);
    b->initial_bias = ora_grid_read((int)b->dim, name, 1);
  }
  v = cfg_find(pairs, npairs, "hills_filename");
  cfg_clean(v ? v : "HILLS", 1, b->mpi_rank, name, sizeof name);
  b->hills_fp = fopen(name, "w");
  v = cfg_find(pairs, npairs, "histogram_filename");
  cfg_clean(v ? v : "HIST", 0, 0, b->hist_name, sizeof b->hist_name);
  free(pairs);
  return 1;
fail:
  free(pairs);
  return 0;
}

/* edm_bias.cpp:34-69 */
ora_bias *ora_bias_create(const char *input_filename) {
  ora_bias *b = (ora_bias *)calloc(1, sizeof(ora_bias));
  b->temperature = -1.0;
  b->hill_density = -1;
  b->temp_hill_cum = -1;
  b->temp_hill_prefactor = -1;
  bias_read_input(b, input_filename);
  return b;
}

/* edm_bias.cpp:71-91 */
void ora_bias_free(ora_bias *b) {
  if (!b) return;
  ora_grid_free(b->target);
  ora_grid_free(b->initial_bias);
  ora_grid_free(b->hist);
  ora_gauss_free(b->bias);
  free(b->bias_dx);
  free(b->bias_sigma);
  free(b->min);
  free(b->max);
  free(b->bper);
  if (b->hills_fp) fclose(b->hills_fp);
  free(b);
}

/* edm_bias.cpp:264-269 */
void ora_bias_setup(ora_bias *b, double temperature, double boltzmann) {
  b->temperature = temperature;
  b->boltzmann_factor = boltzmann * temperature;
}

/* edm_bias.cpp:98-222 (serial branch) */
void ora_bias_subdivide(ora_bias *b, const double *sublo, const double *subhi,
                        const double *boxlo, const double *boxhi,
                        const int *b_periodic, const double *skin) {
  int grid_period[3] = {0, 0, 0};
  double lo[3], hi[3];
  unsigned int d;
  int never_inside = 1;
  if (b->bias != NULL) return;
  if (b->temperature < 0) ora_error("Must call setup before subdivide", "edm_bias.cpp:subdivide");
  for (d = 0; d < b->dim; d++) {
    b->bper[d] = 0;
    if (fabs(boxlo[d] - b->min[d]) < 0.000001 && fabs(boxhi[d] - b->max[d]) < 0.000001)
      b->bper[d] = b_periodic[d];
  }
  for (d = 0; d < b->dim; d++) {
    lo[d] = sublo[d];
    hi[d] = subhi[d];
    if (fabs(sublo[d] - b->min[d]) < 0.000001 && fabs(subhi[d] - b->max[d]) < 0.000001) {
      grid_period[d] = b_periodic[d];
      never_inside = 0;
    } else {
      lo[d] -= skin[d];
      hi[d] += skin[d];
    }
    never_inside &= (lo[d] >= b->max[d] || hi[d] <= b->min[d]);
  }
  b->bias = ora_gauss_create((int)b->dim, lo, hi, b->bias_dx, grid_period, 1, b->bias_sigma);
  b->hist = ora_grid_create((int)b->dim, lo, hi, b->bias_sigma, grid_period, 0, 0);
  ora_gauss_set_boundary(b->bias, b->min, b->max, b->bper);
  if (b->initial_bias != NULL) ora_grid_add_grid(b->bias->grid, b->initial_bias, 1.0, 0.0);
  if (never_inside) {
    b->b_outofbounds = 1;
    return;
  }
  b->total_volume = 0;
  b->total_volume += ora_gauss_get_volume(b->bias);
}

/* edm_bias.cpp:276-295 */
double ora_bias_update_forces(const ora_bias *b, int n, const double *positions,
                              double *forces, int stride, int apply_mask) {
  double der[3] = {0, 0, 0};
  double energy = 0;
  int i;
  unsigned int d;
  if (b->b_outofbounds) return 0.0;
  for (i = 0; i < n; i++) {
    if (apply_mask < 0 || (b->mask[i] & apply_mask)) {
      energy += ora_gauss_get_value_deriv(b->bias, &positions[(size_t)i * stride], der);
      for (d = 0; d < b->dim; d++) forces[(size_t)i * stride + d] -= der[d];
    }
  }
  return energy;
}

/* edm_bias.cpp:297-311 */
double ora_bias_update_force(const ora_bias *b, const double *position, double *force) {
  double der[3] = {0, 0, 0};
  double energy;
  unsigned int d;
  if (b->b_outofbounds) return 0.0;
  energy = ora_gauss_get_value_deriv(b->bias, position, der);
  for (d = 0; d < b->dim; d++) force[d] -= der[d];
  return energy;
}

void ora_bias_set_mask(ora_bias *b, const int *mask) { b->mask = mask; }

/* edm_bias.cpp:586-612 */
static void bias_log_hill(ora_bias *b, const double *pos, double height, double added, char type) {
  unsigned int d;
  if (b->hills_fp) {
    fprintf(b->hills_fp, "%lld %c %d ", b->steps, type, b->hills_added);
    for (d = 0; d < b->dim; d++) fprintf(b->hills_fp, "%.8f ", pos[d]);
    fprintf(b->hills_fp, "%.8f %.8f %.8f\n", height, added, b->cum_bias / b->total_volume);
    fflush(b->hills_fp);
  }
  if (type == 'n' || type == 'b' || type == 'h')
    ora_grid_add_value(b->hist, pos, 1);
  else if (type == 'u' || type == 'v')
    ora_grid_add_value(b->hist, pos, -1);
}

/* edm_bias.cpp:313-380 */
static double bias_flush_overflow(ora_bias *b, double max_bias) {
  const size_t w = b->dim + 1;
  double added = 0, t, h;
  for (; b->overflow_left < b->overflow_right; b->overflow_left++) {
    double *rec = &b->overflow[b->overflow_left * w];
    t = ora_gauss_add_value(b->bias, rec, rec[b->dim]);
    b->hills_added++;
    added += t;
    bias_log_hill(b, rec, rec[b->dim], t, 'b');
    if (added > max_bias) {
      h = fmax(max_bias - added, -rec[b->dim]);
      rec[b->dim] = -h;
      t = ora_gauss_add_value(b->bias, rec, h);
      bias_log_hill(b, rec, h, t, 'v');
      b->hills_added++;
      added += t;
      break;
    }
  }
  if (b->overflow_left == b->overflow_right) b->overflow_left = b->overflow_right = 0;
  return added;
}

/* edm_bias.cpp:413-442 */
void ora_bias_pre_add_hill(ora_bias *b, int est_hill_count) {
  if (b->b_outofbounds) return;
  b->est_hill_count = est_hill_count;
  b->temp_hill_prefactor = b->hill_prefactor;
  if (b->global_tempering > 0)
    if (b->cum_bias / b->total_volume >= b->global_tempering)
      b->temp_hill_prefactor *= exp(-(b->cum_bias / b->total_volume - b->global_tempering) /
                                    (b->global_tempering * (b->bias_factor - 1) * b->boltzmann_factor));
  b->temp_hill_cum = 0;
  b->hills_added = 0;
  b->temp_hill_cum += bias_flush_overflow(b, b->bias_per_step);
  if (b->overflow_left == 0 && b->overflow_right == 0)
    b->b_skip_hill_add = 0;
  else
    b->b_skip_hill_add = 1;
}

/* edm_bias.cpp:444-526 (serial: the MPI packing branch never fires).  The
 * "append right" path increments the index BEFORE storing (:518-521), which
 * is the reference's off-by-one: slot 0 is never written and the newest
 * record sits one past what the flush reads. Reproduced as is. */
static double bias_place_hill(ora_bias *b, const double *pos, double this_h) {
  const size_t w = b->dim + 1;
  int defer = 0;
  unsigned int d;
  double added = 0, undo_h;
  if (b->temp_hill_cum < b->bias_per_step) {
    added = ora_gauss_add_value(b->bias, pos, this_h);
    b->temp_hill_cum += added;
    b->hills_added++;
    bias_log_hill(b, pos, this_h, added, 'h');
    if (b->temp_hill_cum > b->bias_per_step) {
      undo_h = fmax(b->bias_per_step - b->temp_hill_cum, -this_h);
      added = ora_gauss_add_value(b->bias, pos, undo_h);
      b->hills_added++;
      bias_log_hill(b, pos, undo_h, added, 'u');
      b->temp_hill_cum += added;
      defer = 1;
      this_h = -undo_h;
    }
  } else {
    bias_log_hill(b, pos, 0, 0, 'h');
    defer = 1;
  }
  if (defer) {
    if (b->overflow_right == BIAS_BUFFER_SIZE) {
      if (b->overflow_left == 0) {
        ora_error("The bias overflow buffer is full. Too many hills. Either increase & recompile, lower hill_density, or lower bias",
                  "edm_bias.cpp:add_hill");
      } else {
        b->overflow_left--;
        for (d = 0; d < b->dim; d++) b->overflow[b->overflow_left * w + d] = pos[d];
        b->overflow[b->overflow_left * w + d] = this_h;
      }
    } else {
      b->overflow_right++;
      for (d = 0; d < b->dim; d++) b->overflow[b->overflow_right * w + d] = pos[d];
      b->overflow[b->overflow_right * w + d] = this_h;
    }
  }
  return added;
}

/* edm_bias.cpp:528-563 */
void ora_bias_add_hill(ora_bias *b, const double *pos, double runiform) {
  double this_h;
  if (b->temp_hill_prefactor < 0) ora_error("Must call pre_add_hill before add_hill", "edm_bias.cpp:add_hill");
  if (b->b_skip_hill_add) return;
  this_h = b->temp_hill_prefactor;
  if (b->b_outofbounds) return;
  if (b->hill_density < 0 || runiform < b->hill_density / b->est_hill_count) {
    if (b->b_targeting) this_h *= exp(ora_grid_get_value(b->target, pos) - b->expected_target);
    if (b->b_tempering && b->global_tempering < 0)
      this_h *= exp(-ora_gauss_get_value(b->bias, pos) / ((b->bias_factor - 1) * b->boltzmann_factor));
    if (b->hill_density < 0)
      this_h /= b->est_hill_count;
    else
      this_h /= b->hill_density;
    this_h = fmin(this_h, BIAS_CLAMP * b->bias_per_step);
    bias_place_hill(b, pos, this_h);
  }
}

/* edm_bias.cpp:565-583, :922-931 (serial: flushes are no-ops) */
void ora_bias_post_add_hill(ora_bias *b) {
  b->cum_bias += b->temp_hill_cum;
  b->temp_hill_cum = -1;
  b->temp_hill_prefactor = -1;
  b->steps++;
}

/* edm_bias.cpp:401-411 */
void ora_bias_add_hills(ora_bias *b, int n, const double *positions, int stride,
                        const double *runiform, int apply_mask) {
  int i;
  ora_bias_pre_add_hill(b, n);
  for (i = 0; i < n; i++)
    if (apply_mask < 0 || (apply_mask & b->mask[i]))
      ora_bias_add_hill(b, &positions[(size_t)i * stride], runiform[i]);
  ora_bias_post_add_hill(b);
}

/* The per-step loop of the reference's pair fix (lammps/fix_edm_pair.cpp:173-247) over flat pair records, in ITS
 * order: on a hill step pre_add_hill(est) first (:173-174); then for every pair k in list order update_force(&r_k)
 * (:217) -- which therefore reads a bias that already holds the hills of pairs 0..k-1 of this very step -- followed by
 * add_hill(&r_k, u) once, and once more when j is owned (second[k] != 0; :230-237); post_add_hill last (:244-247).
 * force[k] = edm_force[0] after update_force on a zeroed accumulator; the uniforms are consumed one per add_hill
 * call (random->uniform()); *ncalls = number of add_hill calls (the next hill step's estimate, :245). */
double ora_bias_pair_loop(ora_bias *b, int n, const double *r, const int *second, const double *runiform,
                          int hill_step, int est_hill_count, double *force, int *ncalls) {
  int k, c = 0;
  double energy = 0;
  if (hill_step) ora_bias_pre_add_hill(b, est_hill_count);
  for (k = 0; k < n; k++) {
    double f[1] = {0};
    energy += ora_bias_update_force(b, &r[k], f);
    force[k] = f[0];
    if (hill_step) {
      ora_bias_add_hill(b, &r[k], runiform[c++]);
      if (second[k]) ora_bias_add_hill(b, &r[k], runiform[c++]);
    }
  }
  if (hill_step) ora_bias_post_add_hill(b);
  if (ncalls) *ncalls = c;
  return energy;
}

/* edm_bias.cpp:224-262: the serial build routes all three writers to the
 * plain PLUMED writer. */
void ora_bias_write_bias(const ora_bias *b, const char *filename) { ora_gauss_write(b->bias, filename); }
void ora_bias_write_lammps_table(const ora_bias *b, const char *filename) { ora_gauss_write(b->bias, filename); }
void ora_bias_write_histogram(const ora_bias *b) { ora_grid_write(b->hist, b->hist_name); }
void ora_bias_clear_histogram(ora_bias *b) { ora_grid_clear(b->hist); }
ora_gauss *ora_bias_gauss(ora_bias *b) { return b->bias; }
ora_grid *ora_bias_hist(ora_bias *b) { return b->hist; }

double ora_bias_get(const ora_bias *b, const char *name) {
#define G(n, expr) if (strcmp(name, n) == 0) return (double)(expr)
  G("dim", b->dim);
  G("b_tempering", b->b_tempering);
  G("b_targeting", b->b_targeting);
  G("global_tempering", b->global_tempering);
  G("bias_factor", b->bias_factor);
  G("boltzmann_factor", b->boltzmann_factor);
  G("temperature", b->temperature);
  G("hill_prefactor", b->hill_prefactor);
  G("bias_per_step", b->bias_per_step);
  G("hill_density", b->hill_density);
  G("cum_bias", b->cum_bias);
  G("total_volume", b->total_volume);
  G("expected_target", b->expected_target);
  G("b_outofbounds", b->b_outofbounds);
  G("overflow_left", b->overflow_left);
  G("overflow_right", b->overflow_right);
  G("b_skip_hill_add", b->b_skip_hill_add);
  G("hills_added", b->hills_added);
  G("steps", b->steps);
#undef G
  return NAN;
}

void ora_bias_set(ora_bias *b, const char *name, double value) {
#define S(n, lhs, type) if (strcmp(name, n) == 0) { lhs = (type)value; return; }
  S("b_tempering", b->b_tempering, int)
  S("global_tempering", b->global_tempering, double)
  S("bias_factor", b->bias_factor, double)
  S("hill_prefactor", b->hill_prefactor, double)
  S("bias_per_step", b->bias_per_step, double)
  S("hill_density", b->hill_density, double)
  S("cum_bias", b->cum_bias, double)
  S("total_volume", b->total_volume, double)
#undef S
}

const double *ora_bias_array(const ora_bias *b, const char *name) {
  if (strcmp(name, "bias_dx") == 0) return b->bias_dx;
  if (strcmp(name, "bias_sigma") == 0) return b->bias_sigma;
  if (strcmp(name, "min") == 0) return b->min;
  if (strcmp(name, "max") == 0) return b->max;
  return NULL;
}
