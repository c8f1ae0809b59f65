// edm_gauss.cpp -- runtime helpers, the device-resident grid / gaussian-grid objects
// and the host text writers of libedm_hip.so.  Host arithmetic that defines grid
// geometry and the boundary tables follows the reference's operation order
// (citations: file:line in the reference tree) so it is bit-identical to it.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>

#include "edm_comm.h"
#include "edm_internal.h"

#include <utility>
#include <vector>

namespace edm {

static constexpr long long EDM_TAG_CAP_HOST = 1024;   // = EDM_TAG_CAP of edm_kernels.hip

// development aid (EDM_HIP_TRACE): host clock, microseconds after the traced step's entry, at marked place `slot`
// EDM_HIP_TEST_FORCE=<token>[,<token>...] in the environment (tests only): sends work down the paths production
// reaches only under other conditions -- long lists, grids without a ball list, batches that cannot be fused -- so
// that they stay covered.  Tokens: no_ball_list, no_lookup_prep, no_tagged_integrals, no_fast_header,
// no_add_values_chain, dup_ticket_all, gather_wgs=<n>.
static std::string test_force_env() {
  const char *e = getenv("EDM_HIP_TEST_FORCE");
  return e ? std::string(",") + e + "," : std::string();
}
bool test_force(const char *token) {
  static const std::string env = test_force_env();
  return !env.empty() && env.find(std::string(",") + token + ",") != std::string::npos;
}
long long test_force_value(const char *key) {
  static const std::string env = test_force_env();
  const size_t at = env.find(std::string(",") + key + "=");
  return at == std::string::npos ? 0 : atoll(env.c_str() + at + strlen(key) + 2);
}
void ht_mark(edm_hip_gauss *g, int slot) {
  static const bool on = getenv("EDM_HIP_TRACE") != nullptr;
  if (!on || !g || slot < 0 || slot >= 12) return;
  g->ht_marks[slot] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - g->ht_ref_us;
}


static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return EDM_HIP_ERR_HIP;
}

void HillWorkspace::release() {
  gath.release(); slots.release(); heights.release(); hx.release(); hx0.release(); ht.release(); added.release(); partial.release(); scratch.release();
  tail_h1.release(); tail_h2.release(); tail_a2.release(); tail_cum.release();
  hc.release(); tail_flags.release(); tile_flags.release(); tile_list.release(); result.release(); rb.release(); tagged.release();
}

// grid.h:190-213
void make_geometry(Geom &g, int dim, const double *min, const double *max, const double *spacing,
                   const int *periodic, int has_deriv, int interpolate) {
  memset(&g, 0, sizeof(g));
  g.dim = dim;
  g.interp = interpolate;
  g.has_deriv = has_deriv;
  g.rec = has_deriv ? (dim == 1 ? 2 : 4) : 1;
  g.total = 1;
  for (int d = 0; d < 3; d++) {
    g.n[d] = 1;
    g.bper[d] = 1;
    g.bmin[d] = -std::numeric_limits<double>::infinity();
    g.bmax[d] = std::numeric_limits<double>::infinity();
  }
  for (int d = 0; d < dim; d++) {
    g.min[d] = min[d];
    g.max[d] = max[d];
    g.periodic[d] = periodic[d];
    g.n[d] = (int)ceil((g.max[d] - g.min[d]) / spacing[d]);
    g.dx[d] = (g.max[d] - g.min[d]) / g.n[d];
    g.n[d] = g.periodic[d] ? g.n[d] : g.n[d] + 1;
    if (!g.periodic[d]) g.max[d] += g.dx[d];
    g.total *= g.n[d];
  }
}

// grid.h:799-806
void finish_read_geometry(Geom &g) {
  g.total = 1;
  for (int d = 0; d < g.dim; d++) {
    g.dx[d] = (g.max[d] - g.min[d]) / g.n[d];
    if (!g.periodic[d]) {
      g.max[d] += g.dx[d];
      g.n[d] += 1;
    }
    g.total *= g.n[d];
  }
}

void fill_public_geometry(const Geom &g, edm_hip_geometry *out) {
  memset(out, 0, sizeof(*out));
  out->dim = g.dim;
  out->interpolate = g.interp;
  out->total = g.total;
  for (int d = 0; d < g.dim; d++) {
    out->n[d] = g.n[d];
    out->periodic[d] = g.periodic[d];
    out->min[d] = g.min[d];
    out->max[d] = g.max[d];
    out->dx[d] = g.dx[d];
    out->sigma[d] = g.sigma[d];
    out->boundary_periodic[d] = g.bper[d];
    out->boundary_min[d] = g.bmin[d];
    out->boundary_max[d] = g.bmax[d];
    out->minisize[d] = g.msize[d];
  }
}

static void one2multi(const Geom &g, long long index, long long *out) {
  int d;
  for (d = 0; d < g.dim - 1; d++) {
    out[d] = index % g.n[d];
    index = (index - out[d]) / g.n[d];
  }
  out[d] = index;
}

static void put_header(FILE *fp, int b_deriv, int dim, const long long *bins, const double *lo, const double *hi,
                       const int *pbc) {
  fprintf(fp, "#! FORCE %d\n", b_deriv);
  fprintf(fp, "#! NVAR %d\n", dim);
  fprintf(fp, "#! TYPE ");
  for (int d = 0; d < dim; d++) fprintf(fp, "%d ", EDM_GRID_TYPE);
  fprintf(fp, "\n#! BIN ");
  for (int d = 0; d < dim; d++) fprintf(fp, "%lld ", bins[d]);
  fprintf(fp, "\n#! MIN ");
  for (int d = 0; d < dim; d++) fprintf(fp, "%g ", lo[d]);
  fprintf(fp, "\n#! MAX ");
  for (int d = 0; d < dim; d++) fprintf(fp, "%g ", hi[d]);
  fprintf(fp, "\n#! PBC ");
  for (int d = 0; d < dim; d++) fprintf(fp, "%d ", pbc[d]);
  fprintf(fp, "\n");
}

// grid.h:448-503
int write_plumed(const Geom &g, const double *values, const double *derivs, const char *filename) {
  FILE *fp = fopen(filename, "w");
  if (!fp) {
    set_error(std::string("cannot open ") + filename);
    return EDM_HIP_ERR_IO;
  }
  long long bins[3], idx[3];
  double hi[3];
  for (int d = 0; d < g.dim; d++) {
    bins[d] = g.periodic[d] ? g.n[d] : g.n[d] - 1;
    hi[d] = g.periodic[d] ? g.max[d] : g.max[d] - g.dx[d];
  }
  put_header(fp, derivs ? 1 : 0, g.dim, bins, g.min, hi, g.periodic);
  for (long long i = 0; i < g.total; i++) {
    one2multi(g, i, idx);
    for (int d = 0; d < g.dim; d++) fprintf(fp, "%.8f ", g.min[d] + g.dx[d] * (size_t)idx[d]);
    fprintf(fp, "%.8f ", values[i]);
    if (derivs)
      for (int d = 0; d < g.dim; d++) fprintf(fp, "%.8f ", -derivs[i * g.dim + d]);
    fprintf(fp, "\n");
    if (idx[0] == g.n[0] - 1) fprintf(fp, "\n");
  }
  fclose(fp);
  return EDM_HIP_OK;
}

static bool next_word(FILE *fp, char *buf, size_t cap) {
  char fmt[32];
  snprintf(fmt, sizeof fmt, "%%%zus", cap - 1);
  return fscanf(fp, fmt, buf) == 1;
}

// grid.h:712-835
int read_plumed(int dim, const char *filename, int b_interpolate, GridFile &out) {
  FILE *fp = fopen(filename, "r");
  if (!fp) {
    set_error(std::string("Cannot open input file \"") + filename + "\"");
    return EDM_HIP_ERR_IO;
  }
  Geom &g = out.g;
  memset(&g, 0, sizeof(g));
  g.dim = dim;
  g.interp = b_interpolate;
  for (int d = 0; d < 3; d++) {
    g.n[d] = 1;
    g.bper[d] = 1;
    g.bmin[d] = -std::numeric_limits<double>::infinity();
    g.bmax[d] = std::numeric_limits<double>::infinity();
  }
  char w[256];
  int b_deriv = 0, ok = 1;
  auto expect = [&](const char *key) {
    ok = ok && next_word(fp, w, sizeof w) && next_word(fp, w, sizeof w) && strcmp(w, key) == 0;
    return ok;
  };
  if (expect("FORCE")) ok = ok && fscanf(fp, "%d", &b_deriv) == 1;
  if (ok && expect("NVAR")) {
    int nv = 0;
    ok = ok && fscanf(fp, "%d", &nv) == 1 && nv == dim;
  }
  if (ok && expect("TYPE"))
    for (int d = 0; d < dim; d++) {
      int t;
      ok = ok && fscanf(fp, "%d", &t) == 1;
    }
  if (ok && expect("BIN"))
    for (int d = 0; d < dim; d++) ok = ok && fscanf(fp, "%d", &g.n[d]) == 1;
  if (ok && expect("MIN"))
    for (int d = 0; d < dim; d++) ok = ok && fscanf(fp, "%lf", &g.min[d]) == 1;
  if (ok && expect("MAX"))
    for (int d = 0; d < dim; d++) ok = ok && fscanf(fp, "%lf", &g.max[d]) == 1;
  if (ok && expect("PBC"))
    for (int d = 0; d < dim; d++) ok = ok && fscanf(fp, "%d", &g.periodic[d]) == 1;
  if (!ok) {
    fclose(fp);
    set_error(std::string("Mangled grid file: ") + filename);
    return EDM_HIP_ERR_IO;
  }
  finish_read_geometry(g);
  g.has_deriv = b_deriv;
  g.rec = b_deriv ? (dim == 1 ? 2 : 4) : 1;
  out.values.assign((size_t)g.total, 0.0);
  out.derivs.assign(b_deriv ? (size_t)g.total * dim : 0, 0.0);
  for (long long i = 0; i < g.total; i++) {
    for (int d = 0; d < dim; d++) next_word(fp, w, sizeof w);
    if (fscanf(fp, "%lf", &out.values[(size_t)i]) != 1) out.values[(size_t)i] = 0;
    if (b_deriv)
      for (int d = 0; d < dim; d++) {
        double t = 0;
        if (fscanf(fp, "%lf", &t) != 1) t = 0;
        out.derivs[(size_t)i * dim + d] = t;
        out.derivs[(size_t)i * dim + d] *= -1;  // files store the force, grids the gradient (:828)
      }
  }
  fclose(fp);
  return EDM_HIP_OK;
}

}  // namespace edm

using namespace edm;

edm::Tables edm_hip_gauss::tables() const {
  Tables t;
  for (int d = 0; d < 3; d++) {
    t.denom[d] = tab[d][0];
    t.dderiv[d] = tab[d][1];
  }
  t.node1d = node_tab;
  t.ball = ball;
  t.nball = nball;
  return t;
}

extern "C" {

// ---- runtime ----------------------------------------------------------------
const char *edm_hip_last_error(void) { return g_last_error.c_str(); }
const char *edm_hip_version(void) { return "edm-hip 0.1 (gfx950)"; }

int edm_hip_device_count(int *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    (void)hipGetLastError();
    return hip_fail(e, "hipGetDeviceCount");
  }
  *count = n;
  return EDM_HIP_OK;
}
int edm_hip_set_device(int device) {
  EDM_HIP_TRY(hipSetDevice(device));
  return EDM_HIP_OK;
}
int edm_hip_device_info(char *name, size_t cap, int *compute_units, size_t *hbm_bytes) {
  int dev = 0;
  EDM_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  EDM_HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (name && cap) snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName);
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return EDM_HIP_OK;
}
int edm_hip_malloc(void **d_ptr, size_t bytes) {
  EDM_HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
  return EDM_HIP_OK;
}
int edm_hip_free(void *d_ptr) {
  if (d_ptr) EDM_HIP_TRY(hipFree(d_ptr));
  return EDM_HIP_OK;
}
int edm_hip_host_malloc(void **h_ptr, size_t bytes) {
  EDM_HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault));
  return EDM_HIP_OK;
}
int edm_hip_host_free(void *h_ptr) {
  if (h_ptr) EDM_HIP_TRY(hipHostFree(h_ptr));
  return EDM_HIP_OK;
}
int edm_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes) {
  EDM_HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return EDM_HIP_OK;
}
int edm_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes) {
  EDM_HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return EDM_HIP_OK;
}
int edm_hip_memset(void *d_dst, int value, size_t bytes) {
  EDM_HIP_TRY(hipMemset(d_dst, value, bytes));
  return EDM_HIP_OK;
}
int edm_hip_device_synchronize(void) {
  EDM_HIP_TRY(hipDeviceSynchronize());
  return EDM_HIP_OK;
}

// ---- plain grid ---------------------------------------------------------------
// DimmedGrid lookups know nothing of a gaussian boundary (no bounds test, no remap)
static Geom plain_geom(const Geom &q) {
  Geom p = q;
  for (int d = 0; d < 3; d++) {
    p.bper[d] = 1;
    p.bmin[d] = -std::numeric_limits<double>::infinity();
    p.bmax[d] = std::numeric_limits<double>::infinity();
  }
  return p;
}
static size_t grid_doubles(const Geom &q) { return (size_t)q.total * (size_t)q.rec; }
// (re)allocates the node storage for the geometry in g->g, zero-filled (grid.h:892-904)
static int grid_alloc(edm_hip_grid *g) {
  if (!g->stream) EDM_HIP_TRY(hipStreamCreate(&g->stream));
  if (g->values) {
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    EDM_HIP_TRY(hipFree(g->values));
    g->values = nullptr;
  }
  const size_t bytes = sizeof(double) * (grid_doubles(g->g) ? grid_doubles(g->g) : 1);
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->values), bytes));
  EDM_HIP_TRY(hipMemset(g->values, 0, bytes));
  if (g->g.has_deriv && !g->scratch)
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->scratch), sizeof(double) * lookup_scratch_doubles()));
  return EDM_HIP_OK;
}
// node values / derivatives of a record array to host arrays (either may be NULL)
static int records_download(const Geom &q, const double *rec, hipStream_t s, double *h_values, double *h_derivs) {
  double *dv = nullptr, *dd = nullptr;
  if (h_values) EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dv), sizeof(double) * (size_t)q.total));
  if (h_derivs) EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dd), sizeof(double) * (size_t)q.total * q.dim));
  EDM_HIP_TRY(launch_unpack(q, rec, dv, dd, s));
  EDM_HIP_TRY(hipStreamSynchronize(s));
  if (h_values) EDM_HIP_TRY(hipMemcpy(h_values, dv, sizeof(double) * (size_t)q.total, hipMemcpyDeviceToHost));
  if (h_derivs) EDM_HIP_TRY(hipMemcpy(h_derivs, dd, sizeof(double) * (size_t)q.total * q.dim, hipMemcpyDeviceToHost));
  if (dv) (void)hipFree(dv);
  if (dd) (void)hipFree(dd);
  return EDM_HIP_OK;
}
static int records_upload(const Geom &q, double *rec, hipStream_t s, const double *h_values, const double *h_derivs) {
  double *dv = nullptr, *dd = nullptr;
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dv), sizeof(double) * (size_t)q.total));
  EDM_HIP_TRY(hipMemcpy(dv, h_values, sizeof(double) * (size_t)q.total, hipMemcpyHostToDevice));
  if (h_derivs) {
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dd), sizeof(double) * (size_t)q.total * q.dim));
    EDM_HIP_TRY(hipMemcpy(dd, h_derivs, sizeof(double) * (size_t)q.total * q.dim, hipMemcpyHostToDevice));
  }
  EDM_HIP_TRY(launch_pack(q, rec, dv, dd, s));
  EDM_HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(dv);
  if (dd) (void)hipFree(dd);
  return EDM_HIP_OK;
}

int edm_hip_grid_create_ex(edm_hip_grid **out, int dim, const double *min, const double *max, const double *spacing,
                           const int *periodic, int b_derivatives, int b_interpolate) {
  if (!out || dim < 1 || dim > 3) {
    set_error("edm_hip_grid_create: bad arguments");
    return EDM_HIP_ERR_ARG;
  }
  edm_hip_grid *g = new edm_hip_grid;
  make_geometry(g->g, dim, min, max, spacing, periodic, b_derivatives ? 1 : 0, b_interpolate ? 1 : 0);
  int rc = grid_alloc(g);
  if (rc) return rc;
  *out = g;
  return EDM_HIP_OK;
}
int edm_hip_grid_create(edm_hip_grid **out, int dim, const double *min, const double *max, const double *spacing,
                        const int *periodic) {
  return edm_hip_grid_create_ex(out, dim, min, max, spacing, periodic, 0, 0);
}
// DimmedGrid::read (grid.h:712-835): geometry, derivative flag and contents come from the file
static int grid_load_file(edm_hip_grid *g, int dim, const char *filename, int b_interpolate) {
  GridFile gf;
  int rc = read_plumed(dim, filename, b_interpolate ? 1 : 0, gf);
  if (rc) return rc;
  g->g = gf.g;
  rc = grid_alloc(g);
  if (rc) return rc;
  if (g->g.has_deriv) return records_upload(g->g, g->values, g->stream, gf.values.data(), gf.derivs.data());
  EDM_HIP_TRY(hipMemcpy(g->values, gf.values.data(), sizeof(double) * gf.values.size(), hipMemcpyHostToDevice));
  return EDM_HIP_OK;
}
int edm_hip_grid_read(edm_hip_grid **out, int dim, const char *filename, int b_interpolate) {
  if (!out || dim < 1 || dim > 3 || !filename) {
    set_error("edm_hip_grid_read: bad arguments");
    return EDM_HIP_ERR_ARG;
  }
  edm_hip_grid *g = new edm_hip_grid;
  int rc = grid_load_file(g, dim, filename, b_interpolate);
  if (rc) {
    edm_hip_grid_destroy(g);
    return rc;
  }
  *out = g;
  return EDM_HIP_OK;
}
int edm_hip_grid_reread(edm_hip_grid *g, const char *filename) {
  return grid_load_file(g, g->g.dim, filename, g->g.interp);   // b_interpolate_ is kept (grid.h:712-835 never sets it)
}
int edm_hip_grid_set_interpolation(edm_hip_grid *g, int b_interpolate) {
  g->g.interp = b_interpolate ? 1 : 0;
  return EDM_HIP_OK;
}
int edm_hip_grid_destroy(edm_hip_grid *g) {
  if (!g) return EDM_HIP_OK;
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->values) (void)hipFree(g->values);
  if (g->scratch) (void)hipFree(g->scratch);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return EDM_HIP_OK;
}
int edm_hip_grid_geometry(const edm_hip_grid *g, edm_hip_geometry *out) {
  fill_public_geometry(g->g, out);
  out->derivatives = g->g.has_deriv;
  return EDM_HIP_OK;
}
int edm_hip_grid_download(const edm_hip_grid *g, double *h_values) {
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  if (g->g.has_deriv) return records_download(g->g, g->values, g->stream, h_values, nullptr);
  EDM_HIP_TRY(hipMemcpy(h_values, g->values, sizeof(double) * (size_t)g->g.total, hipMemcpyDeviceToHost));
  return EDM_HIP_OK;
}
int edm_hip_grid_download_derivs(const edm_hip_grid *g, double *h_derivs) {
  if (!g->g.has_deriv) {
    set_error("edm_hip_grid_download_derivs: the grid stores no derivatives");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return records_download(g->g, g->values, g->stream, nullptr, h_derivs);
}
int edm_hip_grid_upload(edm_hip_grid *g, const double *h_values) {
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  if (g->g.has_deriv) {   // node values replaced, derivative slots kept
    double *dv = nullptr;
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dv), sizeof(double) * (size_t)g->g.total));
    EDM_HIP_TRY(hipMemcpy(dv, h_values, sizeof(double) * (size_t)g->g.total, hipMemcpyHostToDevice));
    EDM_HIP_TRY(launch_set_values(g->g, g->values, dv, g->stream));
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    (void)hipFree(dv);
    return EDM_HIP_OK;
  }
  EDM_HIP_TRY(hipMemcpy(g->values, h_values, sizeof(double) * (size_t)g->g.total, hipMemcpyHostToDevice));
  return EDM_HIP_OK;
}
int edm_hip_grid_upload_derivs(edm_hip_grid *g, const double *h_values, const double *h_derivs) {
  if (!g->g.has_deriv) {
    set_error("edm_hip_grid_upload_derivs: the grid stores no derivatives");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return records_upload(g->g, g->values, g->stream, h_values, h_derivs);
}
int edm_hip_grid_clear(edm_hip_grid *g) {
  EDM_HIP_TRY(hipMemsetAsync(g->values, 0, sizeof(double) * grid_doubles(g->g), g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}
int edm_hip_grid_add_values(edm_hip_grid *g, long long n, const double *d_x, int x_stride, const double *d_w,
                            double w_const) {
  if (g->g.interp) {   // grid.h:371-373
    set_error("Cannot add_value when using derivatives");
    return EDM_HIP_ERR_STATE;
  }
  EDM_HIP_TRY(launch_hist_add(g->g, g->values, n, d_x, x_stride, nullptr, d_w, w_const, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}
// DimmedGrid::get_value / get_value_deriv batched (grid.h:343-365, :390-446): cubic-Hermite interpolation when
// the grid interpolates and stores derivatives, else the nearest-lower node (value and stored derivatives)
int edm_hip_grid_get_value_deriv(const edm_hip_grid *g, long long n, const double *d_x, int x_stride, double *d_value,
                                 double *d_deriv) {
  if (n <= 0) return EDM_HIP_OK;
  if (g->g.has_deriv) {
    LookupArgs a{};
    a.n = n; a.x = d_x; a.x_stride = x_stride; a.energy = d_value; a.f = d_deriv; a.apply_mask = -1;
    EDM_HIP_TRY(launch_lookup(plain_geom(g->g), g->values, LOOKUP_VALUES, a, g->scratch, nullptr, g->stream));
  } else {
    EDM_HIP_TRY(launch_nearest_values(g->g, g->values, n, d_x, x_stride, d_value, d_deriv, g->stream));
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}
int edm_hip_grid_write(const edm_hip_grid *g, const char *filename) {
  std::vector<double> v((size_t)g->g.total), dv;
  int rc = edm_hip_grid_download(g, v.data());
  if (rc) return rc;
  if (g->g.has_deriv) {
    dv.resize((size_t)g->g.total * g->g.dim);
    rc = edm_hip_grid_download_derivs(g, dv.data());
    if (rc) return rc;
  }
  return write_plumed(g->g, v.data(), g->g.has_deriv ? dv.data() : nullptr, filename);
}

// Grid::add (grid.h:275-290): dst += scale * src(x_node) + offset, node by node, `src` evaluated through ITS
// get_value_deriv (src_geom carries the gaussian boundary when src is a GaussGrid, plain_geom otherwise; a
// source without derivative records answers with its nearest-lower node).  Chunked so that a 512^3 grid needs
// a bounded scratch.
static int grid_add_from(const Geom &dst, double *dst_base, hipStream_t s, double *scratch, const Geom &src_geom,
                         const double *src_base, double scale, double offset) {
  const long long chunk = dst.total < (1ll << 22) ? dst.total : (1ll << 22);
  if (chunk <= 0) return EDM_HIP_OK;
  double *dx = nullptr, *dE = nullptr, *dD = nullptr;
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), sizeof(double) * (size_t)chunk * dst.dim));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dE), sizeof(double) * (size_t)chunk));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dD), sizeof(double) * (size_t)chunk * dst.dim));
  for (long long first = 0; first < dst.total; first += chunk) {
    const long long cnt = (dst.total - first < chunk) ? dst.total - first : chunk;
    EDM_HIP_TRY(launch_node_coords(dst, first, cnt, dx, s));
    if (src_geom.has_deriv) {
      LookupArgs a{};
      a.n = cnt; a.x = dx; a.x_stride = dst.dim; a.energy = dE; a.f = dD; a.apply_mask = -1;
      EDM_HIP_TRY(launch_lookup(src_geom, src_base, LOOKUP_VALUES, a, scratch, nullptr, s));
    } else {
      EDM_HIP_TRY(launch_nearest_values(src_geom, src_base, cnt, dx, dst.dim, dE, dD, s));
    }
    EDM_HIP_TRY(launch_axpy_nodes(dst, dst_base, first, cnt, dE, dD, scale, offset, s));
  }
  EDM_HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(dx); (void)hipFree(dE); (void)hipFree(dD);
  return EDM_HIP_OK;
}
static int need_scratch(edm_hip_grid *g) {
  if (!g->scratch) EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->scratch), sizeof(double) * lookup_scratch_doubles()));
  return EDM_HIP_OK;
}
int edm_hip_grid_add_grid(edm_hip_grid *g, const edm_hip_grid *other, double scale, double offset) {
  if (!g || !other || g->g.dim != other->g.dim) {
    set_error("Grid::add: dimensions differ");
    return EDM_HIP_ERR_ARG;
  }
  int rc = need_scratch(g);
  if (rc) return rc;
  EDM_HIP_TRY(hipStreamSynchronize(other->stream));
  return grid_add_from(g->g, g->values, g->stream, g->scratch, plain_geom(other->g), other->values, scale, offset);
}
int edm_hip_grid_add_gauss(edm_hip_grid *g, const edm_hip_gauss *other, double scale, double offset) {
  if (!g || !other || g->g.dim != other->g.dim) {
    set_error("Grid::add: dimensions differ");
    return EDM_HIP_ERR_ARG;
  }
  int rc = need_scratch(g);
  if (rc) return rc;
  EDM_HIP_TRY(hipStreamSynchronize(other->stream));
  return grid_add_from(g->g, g->values, g->stream, g->scratch, other->g, other->rec, scale, offset);
}

static int multi_write_records(const Geom &q, const double *rec, hipStream_t s, double *scratch, const char *filename,
                               const double *box_min, const double *box_max, const int *b_periodic, int b_lammps_format);

// grid.h:509-674 for one rank, no derivatives: nodes are re-sampled by the nearest-lower lookup
// of DimmedGrid::get_value (grid.h:343-365) -- plain indexing of the downloaded bins
int edm_hip_grid_multi_write(const edm_hip_grid *g, const char *filename, const double *box_min,
                             const double *box_max, const int *b_periodic, int b_lammps_format) {
  const Geom &q = g->g;
  if (b_lammps_format == 1 && q.dim > 1) {
    set_error("Lammps format only valid for 1D grids");
    return EDM_HIP_ERR_ARG;
  }
  if (q.has_deriv) {   // b_derivatives_: re-sampled through get_value_deriv, derivative columns written (:650-661)
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    return multi_write_records(q, g->values, g->stream, g->scratch, filename, box_min, box_max, b_periodic, b_lammps_format);
  }
  std::vector<double> v((size_t)q.total);
  int rc = edm_hip_grid_download(g, v.data());
  if (rc) return rc;
  unsigned int counts[3] = {1, 1, 1}, extra_n = 0;
  if (b_lammps_format) extra_n = (unsigned int)(box_min[0] / q.dx[0]);
  size_t total = 1;
  for (int d = 0; d < q.dim; d++) {
    counts[d] = (unsigned int)(int)ceil((box_max[d] - box_min[d]) / q.dx[d]);
    counts[d] = b_periodic[d] ? counts[d] : counts[d] + 1;
    total *= counts[d];
  }
  FILE *fp = fopen(filename, "w");
  if (!fp) {
    set_error(std::string("cannot open ") + filename);
    return EDM_HIP_ERR_IO;
  }
  if (!b_lammps_format) {
    long long bins[3];
    for (int d = 0; d < q.dim; d++) bins[d] = b_periodic[d] ? (long long)counts[d] : (long long)counts[d] - 1;
    edm::put_header(fp, 0, q.dim, bins, box_min, box_max, b_periodic);
  } else {
    fprintf(fp, "#Auto generated by electronic-dance-music\n\n");
    fprintf(fp, "EDM\n");
    fprintf(fp, "N %u R %g %g\n\n", extra_n + counts[0], q.dx[0], box_max[0]);
    for (size_t i = 1; i < extra_n; i++) fprintf(fp, "%zu %g 0.0 0.0\n", i, i * q.dx[0]);
  }
  for (size_t i = 0; i < total; i++) {
    size_t tmp = i, sup[3] = {0, 0, 0};
    double x[3];
    int d;
    for (d = 0; d < q.dim - 1; d++) {
      sup[d] = tmp % counts[d];
      tmp = (tmp - sup[d]) / counts[d];
      x[d] = sup[d] * q.dx[d] + box_min[d];
    }
    sup[d] = tmp;
    x[d] = sup[d] * q.dx[d] + box_min[d];
    bool in = true;
    for (d = 0; d < q.dim; d++)
      if (!q.periodic[d] && (x[d] < q.min[d] || x[d] >= q.max[d] - q.dx[d])) in = false;
    if (!in) continue;
    long long flat = 0, mul = 1;
    for (d = 0; d < q.dim; d++) {
      double w;
      long long idx = node_index(q, d, x[d], &w);
      if (idx < 0) idx = 0;
      if (idx > q.n[d] - 1) idx = q.n[d] - 1;
      flat += idx * mul;
      mul *= q.n[d];
    }
    if (b_lammps_format) fprintf(fp, "%zu ", i + extra_n);
    for (d = 0; d < q.dim; d++) fprintf(fp, "%.8f ", x[d]);
    fprintf(fp, "%.8f ", v[(size_t)flat]);
    fprintf(fp, "\n");
    if (sup[0] == counts[0] - 1) fprintf(fp, "\n");
  }
  fclose(fp);
  return EDM_HIP_OK;
}

// ---- gaussian grid ------------------------------------------------------------
static int gauss_alloc(edm_hip_gauss *g) {
  const size_t bytes = sizeof(double) * (size_t)g->g.total * g->g.rec;
  EDM_HIP_TRY(hipStreamCreate(&g->stream));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->rec), bytes));
  EDM_HIP_TRY(hipMemset(g->rec, 0, bytes));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->scratch), sizeof(double) * lookup_scratch_doubles()));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_scalars), sizeof(double) * 16));
  EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->h_scalars), sizeof(double) * 128, hipHostMallocDefault));
  // block partial sums of the lookup kernels are written straight into host-mapped pinned memory
  // and added up on the host in index order: no reduction kernel, no copy
  EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->h_partials), sizeof(double) * lookup_scratch_doubles(), hipHostMallocMapped));
  EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&g->d_partials), g->h_partials, 0));
  g->h_stage_bytes = (size_t)4096 * (sizeof(int) + sizeof(double) * (3 + 3)) + 1024;
  EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->h_stage), g->h_stage_bytes, hipHostMallocMapped));
  EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&g->d_stage), g->h_stage, 0));
  memset(g->h_stage, 0, g->h_stage_bytes);   // (the last 128 B hold the polled completion words, see apply_hills)
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_dirty), sizeof(int) * 4));
  EDM_HIP_TRY(hipMemset(g->d_dirty, 0, sizeof(int) * 4));
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_tickets), sizeof(int) * 3 * EDM_TICKET_INTS));
  EDM_HIP_TRY(hipMemset(g->d_tickets, 0, sizeof(int) * 3 * EDM_TICKET_INTS));
  return EDM_HIP_OK;
}

// gaussian_grid.h:559-569
static void update_minigrid(Geom &g) {
  for (int d = 0; d < g.dim; d++) {
    const double dist = sqrt(2 * EDM_GAUSS_SUPPORT) * g.sigma[d];
    g.msize[d] = ifloor(dist / g.dx[d]);
  }
}

int edm_hip_gauss_create(edm_hip_gauss **out, int dim, const double *min, const double *max, const double *spacing,
                         const int *periodic, int b_interpolate, const double *sigma) {
  if (!out || dim < 1 || dim > 3) {
    set_error("edm_hip_gauss_create: bad arguments");
    return EDM_HIP_ERR_ARG;
  }
  edm_hip_gauss *g = new edm_hip_gauss;
  make_geometry(g->g, dim, min, max, spacing, periodic, 1, b_interpolate);  // gaussian_grid.h:70
  for (int d = 0; d < dim; d++) g->g.sigma[d] = sigma[d] * sqrt(2.);        // gaussian_grid.h:75
  int rc = gauss_alloc(g);
  if (rc) return rc;
  rc = edm_hip_gauss_set_boundary(g, min, max, periodic);                    // gaussian_grid.h:78
  if (rc) return rc;
  update_minigrid(g->g);
  *out = g;
  return EDM_HIP_OK;
}

// read_gauss_grid (gaussian_grid.h:647, gaussian_grid.cpp:23-33) = DimmedGaussGrid(filename, sigma)
// (gaussian_grid.h:85-93): the underlying grid is read with interpolation on, sigma is given again (files do
// not store it), the boundary is the grid's own extent and periodicity
int edm_hip_gauss_read(edm_hip_gauss **out, int dim, const char *filename, const double *sigma) {
  if (!out || dim < 1 || dim > 3 || !filename || !sigma) {
    set_error("edm_hip_gauss_read: bad arguments");
    return EDM_HIP_ERR_ARG;
  }
  GridFile gf;
  int rc = read_plumed(dim, filename, 1, gf);
  if (rc) return rc;
  if (!gf.g.has_deriv) {
    set_error("a gaussian grid needs the derivative columns (FORCE 1)");
    return EDM_HIP_ERR_IO;
  }
  edm_hip_gauss *g = new edm_hip_gauss;
  g->g = gf.g;
  for (int d = 0; d < dim; d++) g->g.sigma[d] = sigma[d] * sqrt(2.);
  rc = gauss_alloc(g);
  if (rc) return rc;
  rc = edm_hip_gauss_set_boundary(g, g->g.min, g->g.max, g->g.periodic);
  if (rc) return rc;
  update_minigrid(g->g);
  rc = records_upload(g->g, g->rec, g->stream, gf.values.data(), gf.derivs.data());
  if (rc) return rc;
  *out = g;
  return EDM_HIP_OK;
}

int edm_hip_gauss_destroy(edm_hip_gauss *g) {
  if (!g) return EDM_HIP_OK;
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (int d = 0; d < 3; d++)
    for (int k = 0; k < 2; k++)
      if (g->tab[d][k]) (void)hipFree(g->tab[d][k]);
  if (g->rec) (void)hipFree(g->rec);
  if (g->faces) (void)hipFree(g->faces);
  if (g->scratch) (void)hipFree(g->scratch);
  if (g->d_scalars) (void)hipFree(g->d_scalars);
  if (g->h_scalars) (void)hipHostFree(g->h_scalars);
  if (g->h_stage) (void)hipHostFree(g->h_stage);
  if (g->h_partials) (void)hipHostFree(g->h_partials);
  if (g->d_dirty) (void)hipFree(g->d_dirty);
  if (g->d_tickets) (void)hipFree(g->d_tickets);
  if (g->d_ready) (void)hipFree(g->d_ready);
  if (g->node_tab) (void)hipFree(g->node_tab);
  if (g->ball) (void)hipFree(g->ball);
  if (g->prof_ev) {
    for (int i = 0; i < 2 * edm_hip_gauss::PROF_RING; i++) (void)hipEventDestroy(g->prof_ev[i]);
    delete[] g->prof_ev;
  }
  g->ws.release();
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return EDM_HIP_OK;
}

// Tables::ball for the current spacing, sigma and stencil half-widths (see edm_kernels.h); checked before every hill
// batch, rebuilt when one of them changed (creation, re-read of a file with another spacing)
static int ball_list_ensure(edm_hip_gauss *g) {
  const Geom &q = g->g;
  double key[10] = {(double)q.dim, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int d = 0; d < q.dim; d++) {
    key[1 + d] = q.dx[d];
    key[4 + d] = q.sigma[d];
    key[7 + d] = (double)q.msize[d];
  }
  if (memcmp(key, g->ball_key, sizeof(key)) == 0) return EDM_HIP_OK;
  memcpy(g->ball_key, key, sizeof(key));
  if (g->ball) {
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    (void)hipFree(g->ball);
  }
  g->ball = nullptr;
  g->nball = 0;
  static const bool off = test_force("no_ball_list");   // (tests: the stencil box walked point by point)
  if (off || q.dim < 2) return EDM_HIP_OK;
  long long box = 1;
  for (int d = 0; d < q.dim; d++) {
    if (q.msize[d] < 0 || q.msize[d] > 127) return EDM_HIP_OK;
    box *= 2 * q.msize[d] + 1;
  }
  if (box > (1LL << 22)) return EDM_HIP_OK;
  std::vector<std::pair<double, int>> cand;
  const int m0 = q.msize[0], m1 = q.msize[1], m2 = q.dim > 2 ? q.msize[2] : 0;
  for (int o2 = -m2; o2 <= m2; o2++)
    for (int o1 = -m1; o1 <= m1; o1++)
      for (int o0 = -m0; o0 <= m0; o0++) {
        const int o[3] = {o0, o1, o2};
        double sum = 0;
        for (int d = 0; d < q.dim; d++) {
          const int a = o[d] < 0 ? -o[d] : o[d];
          const double e = (a > 1 ? (double)(a - 1) : 0.0) * q.dx[d] / q.sigma[d];
          sum += e * e;
        }
        if (sum <= EDM_GAUSS_SUPPORT * (1.0 + 1e-6) + 1e-9)
          cand.push_back(std::make_pair(sum, (o0 + 128) | ((o1 + 128) << 8) | ((o2 + 128) << 16)));
      }
  // inner offsets first (stable: ties keep stencil order): the cheap rejects of the outer shell share their waves
  std::stable_sort(cand.begin(), cand.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
  if (cand.empty() || (long long)cand.size() * 2 > box) return EDM_HIP_OK;   // (no gain over the box walk)
  std::vector<int> packed(cand.size());
  for (size_t i = 0; i < cand.size(); i++) packed[i] = cand[i].second;
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->ball), sizeof(int) * packed.size()));
  EDM_HIP_TRY(hipMemcpy(g->ball, packed.data(), sizeof(int) * packed.size(), hipMemcpyHostToDevice));
  g->nball = (int)packed.size();
  return EDM_HIP_OK;
}

// gaussian_grid.h:378-435.  libm erf/exp on the host: table bits equal the reference's.
// Tables::node1d of a 1-D grid with walls, for the current geometry and boundary (dropped otherwise)
static int node_table_rebuild(edm_hip_gauss *g) {
  if (g->node_tab) (void)hipFree(g->node_tab);
  g->node_tab = nullptr;
  const Geom &q = g->g;
  if (q.dim != 1 || q.bper[0] || !g->tab[0][0] || !g->tab[0][1]) return EDM_HIP_OK;
  double *nt = nullptr;
  EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&nt), sizeof(double) * 4 * (size_t)q.n[0]));
  EDM_HIP_TRY(launch_build_node_table(q, g->tables(), nt, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  g->node_tab = nt;
  return EDM_HIP_OK;
}

int edm_hip_gauss_set_boundary(edm_hip_gauss *g, const double *min, const double *max, const int *periodic) {
  g->tiles_per_hill = 0;  // (cached bound: recomputed for the new geometry on demand)
  Geom &q = g->g;
  for (int d = 0; d < q.dim; d++) {
    q.bmin[d] = min[d];
    q.bmax[d] = max[d];
    q.bper[d] = periodic[d];
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  EDM_HIP_TRY(hipMemset(g->d_dirty, 0, sizeof(int)));
  std::vector<double> den(EDM_BC_TABLE_SIZE), dder(EDM_BC_TABLE_SIZE);
  for (int d = 0; d < q.dim; d++) {
    if (q.bper[d]) continue;
    const double lo = q.bmin[d], hi = q.bmax[d], sg = q.sigma[d];
    for (size_t j = 0; j < EDM_BC_TABLE_SIZE; j++) {
      const double s = j * (hi - lo) / (EDM_BC_TABLE_SIZE - 1) + lo;
      const double t1 = sqrt(M_PI) * sg / 2. * (erf((s - lo) / sg) + erf((hi - s) / sg));
      const double t2 = sqrt(M_PI) * sg / 2. * erf((hi - lo) / sg);
      den[j] = t1;
      den[j] += (t2 - t1) * smooth_step((s - lo) / (EDM_BC_MAR * sg));
      den[j] += (t2 - t1) * smooth_step((hi - s) / (EDM_BC_MAR * sg));
      const double t3 = 1. * (exp(-((s - lo) * (s - lo)) / (sg * sg)) - exp(-((hi - s) * (hi - s)) / (sg * sg)));
      dder[j] = t3;
      dder[j] += (t2 - t1) * smooth_step_dt((s - lo) / (EDM_BC_MAR * sg)) / (EDM_BC_MAR * sg) -
                 t3 * smooth_step((s - lo) / (EDM_BC_MAR * sg));
      dder[j] += -(t2 - t1) * smooth_step_dt((hi - s) / (EDM_BC_MAR * sg)) / (EDM_BC_MAR * sg) -
                 t3 * smooth_step((hi - s) / (EDM_BC_MAR * sg));
    }
    for (int k = 0; k < 2; k++) {
      if (!g->tab[d][k])
        EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->tab[d][k]), sizeof(double) * EDM_BC_TABLE_SIZE));
      EDM_HIP_TRY(hipMemcpy(g->tab[d][k], k ? dder.data() : den.data(), sizeof(double) * EDM_BC_TABLE_SIZE,
                            hipMemcpyHostToDevice));
    }
  }
  return node_table_rebuild(g);
}

int edm_hip_gauss_geometry(const edm_hip_gauss *g, edm_hip_geometry *out) {
  fill_public_geometry(g->g, out);
  return EDM_HIP_OK;
}

int edm_hip_gauss_download_tables(const edm_hip_gauss *g, int dim_index, double *h_denom, double *h_denom_deriv) {
  if (!g || dim_index < 0 || dim_index >= g->g.dim || g->g.bper[dim_index] || !g->tab[dim_index][0]) {
    set_error("edm_hip_gauss_download_tables: no boundary tables for this dimension (periodic boundary)");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  const size_t bytes = sizeof(double) * EDM_BC_TABLE_SIZE;
  if (h_denom) EDM_HIP_TRY(hipMemcpy(h_denom, g->tab[dim_index][0], bytes, hipMemcpyDeviceToHost));
  if (h_denom_deriv) EDM_HIP_TRY(hipMemcpy(h_denom_deriv, g->tab[dim_index][1], bytes, hipMemcpyDeviceToHost));
  return EDM_HIP_OK;
}

int edm_hip_gauss_download(const edm_hip_gauss *g, double *h_values, double *h_derivs) {
  return records_download(g->g, g->rec, g->stream, h_values, h_derivs);
}

int edm_hip_gauss_upload(edm_hip_gauss *g, const double *h_values, const double *h_derivs) {
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  faces_touch(g);
  return records_upload(g->g, g->rec, g->stream, h_values, h_derivs);
}

int edm_hip_gauss_clear(edm_hip_gauss *g) {
  faces_touch(g);
  EDM_HIP_TRY(hipMemsetAsync(g->rec, 0, sizeof(double) * (size_t)g->g.total * g->g.rec, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

int edm_hip_gauss_device_buffer(edm_hip_gauss *g, double **d_records, int *doubles_per_node, long long *nodes) {
  // the caller may write the records behind the library's back (a collective): wait for queued updates, and
  // keep the lookups on the node records themselves from here on
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  g->faces_mode = 0;
  faces_touch(g);
  if (d_records) *d_records = g->rec;
  if (doubles_per_node) *doubles_per_node = g->g.rec;
  if (nodes) *nodes = g->g.total;
  return EDM_HIP_OK;
}

// event pair for the next timed launch (nullptrs when profiling is off or the ring is full)
}  // extern "C"
void edm::profile_slot(const edm_hip_gauss *gc, hipEvent_t *e0, hipEvent_t *e1) {
  edm_hip_gauss *g = const_cast<edm_hip_gauss *>(gc);
  *e0 = *e1 = nullptr;
  if (!g->profiling || !g->prof_ev) return;
  if ((g->prof_seen++ % g->profiling) != 0 || g->prof_pending >= edm_hip_gauss::PROF_RING) return;
  *e0 = g->prof_ev[2 * g->prof_pending];
  *e1 = g->prof_ev[2 * g->prof_pending + 1];
  g->prof_pending++;
}
extern "C" {
// sums the stamped launches (their kernels must have completed: call after a stream synchronisation)
static void profile_drain(edm_hip_gauss *g) {
  for (int i = 0; i < g->prof_pending; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->prof_ev[2 * i], g->prof_ev[2 * i + 1]) == hipSuccess) {
      g->prof_ms += ms;
      g->prof_launches += 1;
    }
  }
  g->prof_pending = 0;
}

static int fetch_scalar(const edm_hip_gauss *g, int slot, double *out) {
  EDM_HIP_TRY(hipMemcpyAsync(g->h_scalars + slot, g->d_scalars + slot, sizeof(double), hipMemcpyDeviceToHost, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  *out = g->h_scalars[slot];
  return EDM_HIP_OK;
}

int edm_hip_gauss_get_value_deriv(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride,
                                  double *d_energy, double *d_deriv) {
  if (n <= 0) return EDM_HIP_OK;
  LookupArgs a{};
  a.n = n; a.x = d_x; a.x_stride = x_stride; a.energy = d_energy; a.f = d_deriv; a.apply_mask = -1;
  const double *faces = nullptr;
  int rc = faces_prepare(const_cast<edm_hip_gauss *>(g), &faces);
  if (rc) return rc;
  EDM_HIP_TRY(launch_lookup(g->g, g->rec, LOOKUP_VALUES, a, g->scratch, nullptr, g->stream, nullptr, nullptr, nullptr, faces));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

int edm_hip_gauss_sample_index(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride,
                               long long *d_flat) {
  if (n <= 0) return EDM_HIP_OK;
  LookupArgs a{};
  a.n = n; a.x = d_x; a.x_stride = x_stride; a.flat = d_flat; a.apply_mask = -1;
  EDM_HIP_TRY(launch_lookup(g->g, g->rec, LOOKUP_INDEX, a, g->scratch, nullptr, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

int edm_hip_gauss_remap(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride, double *d_out) {
  if (n <= 0) return EDM_HIP_OK;
  EDM_HIP_TRY(launch_remap(g->g, n, d_x, x_stride, d_out, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

int edm_hip_gauss_update_forces(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride, double *d_f,
                                int f_stride, const int *d_mask, int apply_mask, double *energy) {
  if (energy) *energy = 0;
  int nblk = 0;
  // (a forces-only call: the workgroups tag their partial energy sums and the host looks at the slots instead of
  //  waiting for the stream, see edm_hip_gauss_pair_forces)
  const bool poll = n > 0 && edm::forces_poll_enabled();
  edm_hip_gauss *gm = const_cast<edm_hip_gauss *>(g);
  const unsigned long long tag = poll ? ++gm->force_seq : 0ull;
  int rc = edm::update_forces_enqueue(g, n, d_x, x_stride, d_f, f_stride, d_mask, apply_mask, &nblk, tag);
  if (rc) return rc;
  if (n <= 0) return EDM_HIP_OK;
  if (poll) {
    double e = 0;
    if (edm::poll_tagged_partials(g, nblk, tag, &e)) {
      gm->polled_forces++;
      if (energy) *energy = e;
      return EDM_HIP_OK;
    }
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));   // (the poll ran out: the slots are complete now)
    e = 0;
    for (int i = 0; i < nblk; i++) e += g->h_partials[2 * i];
    if (energy) *energy = e;
    return EDM_HIP_OK;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  const double e = edm::pair_forces_finish(g, nblk);
  if (energy) *energy = e;
  return EDM_HIP_OK;
}

int edm_hip_gauss_pair_forces(const edm_hip_gauss *g, long long n, const double *d_r, double *d_force,
                              double *energy) {
  if (energy) *energy = 0;
  int nblk = 0;
  if (g->g.dim == 1 && n > 0 && edm::forces_poll_enabled()) {
    // a forces-only call (every fix edm_pair step between two hill steps): the workgroups tag their partial energy
    // sums, the host looks at the slots instead of waiting for the stream -- the wait's wake-up alone is ~7 us of a
    // 20 us call.  (The force array is complete when every workgroup's sum is: a workgroup stores its forces first.)
    edm_hip_gauss *gm = const_cast<edm_hip_gauss *>(g);
    hipEvent_t e0, e1;
    profile_slot(g, &e0, &e1);
    int tagged = 0;
    const unsigned long long tag = ++gm->force_seq;
    EDM_HIP_TRY(edm::launch_pair_forces(g->g, g->rec, n, d_r, d_force, g->d_partials, nullptr, g->stream, e0, e1, &nblk, tag, &tagged));
    if (tagged) {
      double e = 0;
      if (edm::poll_tagged_partials(g, nblk, tag, &e)) {
        gm->polled_forces++;
        if (energy) *energy = e;
        return EDM_HIP_OK;
      }
      EDM_HIP_TRY(hipStreamSynchronize(g->stream));   // (the poll ran out: the slots are complete now)
      e = 0;
      for (int i = 0; i < nblk; i++) e += g->h_partials[2 * i];
      if (energy) *energy = e;
      return EDM_HIP_OK;
    }
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    const double e = edm::pair_forces_finish(g, nblk);
    if (energy) *energy = e;
    return EDM_HIP_OK;
  }
  int rc = edm::pair_forces_enqueue(g, n, d_r, d_force, &nblk);
  if (rc) return rc;
  if (n <= 0) return EDM_HIP_OK;
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  const double e = edm::pair_forces_finish(g, nblk);
  if (energy) *energy = e;
  return EDM_HIP_OK;
}

int edm_hip_gauss_wait(edm_hip_gauss *g) {
  if (g) EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

// The polled completion's ordering argument, put to the test: `iterations` launches of the protocol on a region of
// `words` 8-byte words that all carry the launch's number; the host polls the flag exactly as apply_hills does and then
// reads the region: a word that is not the launch's number is a flag that overtook its data.
int edm_hip_debug_flag_order_stress(int iterations, long long words, long long *violations, long long *timeouts) {
  if (violations) *violations = 0;
  if (timeouts) *timeouts = 0;
  if (iterations < 0 || words < 1 || words > (1 << 22)) return EDM_HIP_ERR_ARG;
  char *h = nullptr;
  const size_t bytes = (size_t)words * 8 + 128;
  EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocMapped));
  memset(h, 0, bytes);
  char *d = nullptr;
  hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0);
  hipStream_t s = nullptr;
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  long long bad = 0, late = 0;
  volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(h + (size_t)words * 8 + 64);
  volatile long long *payload = reinterpret_cast<volatile long long *>(h);
  for (int it = 1; e == hipSuccess && it <= iterations; it++) {
    e = launch_flag_order_stress(reinterpret_cast<long long *>(d), words, (unsigned long long)it,
                                 reinterpret_cast<unsigned long long *>(d + (size_t)words * 8 + 64), s);
    if (e != hipSuccess) break;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
    bool seen = false;
    for (unsigned spin = 0;; spin++) {
      if (flag[0] == (unsigned long long)it) { seen = true; break; }
      __builtin_ia32_pause();
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) {
      late++;
      e = hipStreamSynchronize(s);
      continue;
    }
    // (last words first: they left the device last)
    for (long long w = words - 1; w >= 0; w--)
      if (payload[w] != (long long)it) bad++;
  }
  if (s) {
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
  }
  (void)hipHostFree(h);
  if (e != hipSuccess) {
    set_error(std::string("flag_order_stress: ") + hipGetErrorString(e));
    return EDM_HIP_ERR_HIP;
  }
  if (violations) *violations = bad;
  if (timeouts) *timeouts = late;
  return EDM_HIP_OK;
}

int edm_hip_gauss_profile_enable(edm_hip_gauss *g, int enabled) {
  if (enabled && !g->prof_ev) {
    g->prof_ev = new hipEvent_t[2 * edm_hip_gauss::PROF_RING];
    for (int i = 0; i < 2 * edm_hip_gauss::PROF_RING; i++) EDM_HIP_TRY(hipEventCreate(&g->prof_ev[i]));
  }
  if (!enabled && g->profiling) {
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    profile_drain(g);
  }
  g->profiling = enabled > 0 ? enabled : 0;
  g->prof_seen = 0;
  return EDM_HIP_OK;
}

int edm_hip_gauss_profile_read(edm_hip_gauss *g, double *kernel_ms_total, long long *launches, int reset) {
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  profile_drain(g);
  if (kernel_ms_total) *kernel_ms_total = g->prof_ms;
  if (launches) *launches = g->prof_launches;
  if (reset) {
    g->prof_ms = 0;
    g->prof_launches = 0;
  }
  return EDM_HIP_OK;
}

}  // extern "C"

// ---- hill application pipeline --------------------------------------------------
namespace edm {

static bool faces_wanted(const edm_hip_gauss *g) {
  const Geom &q = g->g;
  if (g->faces_mode == 0 || q.dim < 2 || !q.interp || q.rec != 4) return false;
  for (int d = 0; d < q.dim; d++)
    if (!q.bper[d]) return false;   // (walls: the boundary duplication writes node values the replica would miss)
  if (g->faces_mode == 1) return true;
  // automatic: only where the node records outgrow the L2s (8 x 4 MB): below that the corner gathers hit cache
  return (size_t)q.total * q.rec * sizeof(double) >= ((size_t)32 << 20) && !g->faces_unavailable;
}
int faces_prepare(edm_hip_gauss *g, const double **faces) {
  *faces = nullptr;
  if (!faces_wanted(g)) return EDM_HIP_OK;
  const size_t bytes = (size_t)g->g.total * 16 * sizeof(double);
  if (!g->faces) {
    size_t free_b = 0, total_b = 0;
    const bool fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > bytes + ((size_t)2 << 30);
    hipError_t e = fits ? hipMalloc(reinterpret_cast<void **>(&g->faces), bytes) : hipErrorOutOfMemory;
    if (e != hipSuccess) {
      (void)hipGetLastError();
      g->faces = nullptr;
      if (g->faces_mode == 1) return hip_fail(e, "hipMalloc(lookup replica)");
      g->faces_unavailable = true;   // an optimisation, not a requirement: the node records serve the lookups
      return EDM_HIP_OK;
    }
    g->faces_state = 2;
  }
  if (g->faces_state != 1) {
    EDM_HIP_TRY(launch_build_faces(g->g, g->rec, g->faces, g->stream));
    g->faces_state = 1;
    g->faces_builds++;
  }
  *faces = g->faces;
  return EDM_HIP_OK;
}

// K1 without the host wait: the launch is queued, the per-workgroup energy sums land in host-mapped
// memory; pair_forces_finish() adds them up after the stream has been synchronised by the caller
int pair_forces_enqueue(const edm_hip_gauss *g, long long n, const double *d_r, double *d_force, int *nblk) {
  *nblk = 0;
  if (g->g.dim != 1) {
    set_error("pair_forces: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  if (n <= 0) return EDM_HIP_OK;
  hipEvent_t e0, e1;
  profile_slot(g, &e0, &e1);
  EDM_HIP_TRY(launch_pair_forces(g->g, g->rec, n, d_r, d_force, g->d_partials, nullptr, g->stream, e0, e1, nblk));
  return EDM_HIP_OK;
}
// K2 (any dimension, strided rows, group mask) without the host wait; finish with pair_forces_finish()
int update_forces_enqueue(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride, double *d_f, int f_stride,
                          const int *d_mask, int apply_mask, int *nblk, unsigned long long tag) {
  *nblk = 0;
  if (n <= 0) return EDM_HIP_OK;
  if (apply_mask >= 0 && !d_mask) {
    set_error("update_forces: apply_mask >= 0 needs a mask");
    return EDM_HIP_ERR_ARG;
  }
  LookupArgs a{};
  a.n = n; a.x = d_x; a.x_stride = x_stride; a.f = d_f; a.f_stride = f_stride; a.mask = d_mask; a.apply_mask = apply_mask;
  a.partial_tag = tag;   // (0: plain partial sums for a caller that waits for the stream)
  hipEvent_t e0, e1;
  profile_slot(g, &e0, &e1);
  const double *faces = nullptr;
  int rcf = faces_prepare(const_cast<edm_hip_gauss *>(g), &faces);
  if (rcf) return rcf;
  EDM_HIP_TRY(launch_lookup(g->g, g->rec, LOOKUP_FORCES, a, g->d_partials, nullptr, g->stream, e0, e1, nblk, faces));
  return EDM_HIP_OK;
}
int pending_forces_flush(const edm_hip_gauss *g, PendingForces *pf) {
  if (!pf || !pf->active) return EDM_HIP_OK;
  pf->active = false;
  if (pf->lookup) {
    hipEvent_t e0, e1;
    profile_slot(g, &e0, &e1);
    const double *faces = nullptr;
    int rcf = faces_prepare(const_cast<edm_hip_gauss *>(g), &faces);
    if (rcf) return rcf;
    EDM_HIP_TRY(launch_lookup(g->g, g->rec, LOOKUP_FORCES, pf->la, g->d_partials, nullptr, g->stream, e0, e1, &pf->nblk, faces));
    pf->launched();
    return EDM_HIP_OK;
  }
  if (pf->list) {
    EDM_HIP_TRY(launch_pairlist_forces(g->g, g->rec, pf->pl, g->d_partials, g->stream, &pf->nblk));
    pf->launched();
    return EDM_HIP_OK;
  }
  if (pf->tag) {
    if (g->g.dim != 1) {
      set_error("pair_forces: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
      return EDM_HIP_ERR_ARG;
    }
    hipEvent_t e0, e1;
    profile_slot(g, &e0, &e1);
    int tagged = 0;
    EDM_HIP_TRY(launch_pair_forces(g->g, g->rec, pf->n, pf->d_r, pf->d_force, g->d_partials, nullptr, g->stream, e0, e1,
                                   &pf->nblk, pf->tag, &tagged));
    pf->tagged = tagged != 0;
    pf->launched();
    return EDM_HIP_OK;
  }
  int rc = pair_forces_enqueue(g, pf->n, pf->d_r, pf->d_force, &pf->nblk);
  if (!rc) pf->launched();
  return rc;
}
int select_prep_enqueue(const edm_hip_gauss *g, const SelectArgs &a_in, const HillList &h, PendingForces *pf) {
  SelectArgs a = a_in;
  // development aid (EDM_HIP_TRACE=select): stamps of one k_pair_forces_select launch to stderr
  static const bool tracing = getenv("EDM_HIP_TRACE") && !strcmp(getenv("EDM_HIP_TRACE"), "select");
  static int launches = 0;
  unsigned long long *d_trace = nullptr;
  size_t trace_wgs = 0;
  if (tracing && pf && pf->active && !pf->list && ++launches == 150) {
    trace_wgs = (size_t)((a.n + 2047) / 2048) + 1024;
    if (hipMalloc(reinterpret_cast<void **>(&d_trace), trace_wgs * 64) == hipSuccess) {
      (void)hipMemset(d_trace, 0, trace_wgs * 64);
      a.trace = d_trace;
    }
  }
  struct TraceDump {
    const edm_hip_gauss *g; unsigned long long *d; size_t wgs, nsel;
    ~TraceDump() {
      if (!d) return;
      (void)hipStreamSynchronize(g->stream);
      std::vector<unsigned long long> tr(wgs * 8);
      (void)hipMemcpy(tr.data(), d, wgs * 64, hipMemcpyDeviceToHost);
      (void)hipFree(d);
      unsigned long long t0 = ~0ull;
      for (size_t w = 0; w < wgs; w++) if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8];
      const char *names[8] = {"start", "published", "ticket (not last)", "ticket (last)", "list prepared", "", "", "end"};
      for (int role = 0; role < 2; role++)
        for (int k = 0; k < 8; k++) {
          std::vector<double> v;
          for (size_t w = role ? nsel : 0; w < (role ? wgs : nsel); w++)
            if (tr[w * 8 + k]) v.push_back((double)(tr[w * 8 + k] - t0) * 0.01);
          if (v.empty()) continue;
          std::sort(v.begin(), v.end());
          fprintf(stderr, "[edm trace] %s %-18s n=%4zu  min %6.2f  med %6.2f  max %6.2f us\n", role ? "forces   " : "selection",
                  names[k], v.size(), v.front(), v[v.size() / 2], v.back());
        }
    }
  } dump{g, d_trace, trace_wgs, (size_t)((a.n + 2047) / 2048)};
  if (pf && pf->active && pf->list && g->g.dim == 1 && a.n > 0 && pf->pl.nall > 0) {
    pf->active = false;
    EDM_HIP_TRY(launch_pairlist_forces_select(a, g->g, h, g->rec, pf->pl, g->d_partials, g->stream, &pf->nblk));
    pf->launched();
    return EDM_HIP_OK;
  }
  static const bool lookup_fuse_env = !test_force("no_lookup_prep");
  if (pf && pf->active && pf->lookup && lookup_fuse_env && !a.pack && g->g.dim > 1 && g->g.interp && g->g.rec == 4 &&
      pf->la.n > 0 && a.n > 0) {
    // fix edm step without an overflow flush: the pending force kernel (K2) and the step's selection share a launch
    pf->active = false;
    hipEvent_t e0, e1;
    profile_slot(g, &e0, &e1);
    const double *faces = nullptr;
    int rcf = faces_prepare(const_cast<edm_hip_gauss *>(g), &faces);
    if (rcf) return rcf;
    EDM_HIP_TRY(launch_lookup_select(g->g, g->rec, pf->la, g->d_partials, g->stream, e0, e1, &pf->nblk, faces, a, h));
    pf->launched();
    const_cast<edm_hip_gauss *>(g)->lookup_prep_launches++;
    return EDM_HIP_OK;
  }
  if (pf && pf->active && !pf->list && !pf->lookup && pair_forces_select_fusable(g->g, pf->n, a.n)) {
    hipEvent_t e0, e1;
    profile_slot(g, &e0, &e1);
    pf->active = false;
    ht_mark(const_cast<edm_hip_gauss *>(g), 1);
    EDM_HIP_TRY(launch_pair_forces_select(a, g->g, h, g->rec, pf->n, pf->d_r, pf->d_force, g->d_partials, g->stream, e0, e1,
                                          &pf->nblk));
    pf->launched();
    ht_mark(const_cast<edm_hip_gauss *>(g), 2);
    return EDM_HIP_OK;
  }
  int rc = pending_forces_flush(g, pf);
  if (rc) return rc;
  ht_mark(const_cast<edm_hip_gauss *>(g), 1);
  EDM_HIP_TRY(launch_select_prep(a, g->g, h, g->stream));
  ht_mark(const_cast<edm_hip_gauss *>(g), 2);
  return EDM_HIP_OK;
}
// forces-only calls poll their workgroups' tagged sums unless EDM_HIP_POLL=0
bool forces_poll_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("EDM_HIP_POLL");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}
// waits (bounded: 2 ms) until every one of the nblk slots {sum, tag} in the host-mapped partial-sum array carries `tag`,
// and adds the sums up in workgroup order; false when the poll ran out
bool poll_tagged_partials(const edm_hip_gauss *g, int nblk, unsigned long long tag, double *energy) {
  const volatile unsigned long long *slots = reinterpret_cast<const volatile unsigned long long *>(g->h_partials);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
  double e = 0;
  unsigned spin = 0;
  for (int i = 0; i < nblk; i++) {
    while (slots[2 * i + 1] != tag) {
      __builtin_ia32_pause();
      if ((++spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) return false;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    unsigned long long bits = slots[2 * i];
    double v;
    memcpy(&v, &bits, 8);
    e += v;
  }
  *energy = e;
  return true;
}
double pair_forces_finish(const edm_hip_gauss *g, int nblk) {
  double e = 0;
  for (int i = 0; i < nblk; i++) e += g->h_partials[i];
  return e;
}

// Completion of a short hill batch is seen by polling two words its last kernel writes behind the read-back
// region (EDM_HIP_POLL=0 in the environment: always wait for the stream instead)
static bool poll_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("EDM_HIP_POLL");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}
static const long long SMALL_BATCH = 4096;  // read-back of a batch this small is one async burst

// Upper bound on the tiles one hill can mark (k_mark_tiles keeps the tiles of the stencil's bounding box whose
// nearest node lies inside the dp2 < 8 ball): the maximum, over every alignment of the hill's centre node
// within its tile, of the tiles whose distance to the hill's CELL (the hill sits anywhere in it) passes the
// same test.  The launch bound of a culled gather is this times the hill count -- idle workgroups of an
// over-sized launch cost ~2 ns each, which is what a 3-D batch of 250 hills would otherwise mostly pay for.
static long long tiles_per_hill_bound(const Geom &q) {
  static const int T1[3] = {256, 1, 1}, T2[3] = {16, 16, 1}, T3[3] = {8, 8, 4};
  const int *T = q.dim == 1 ? T1 : q.dim == 2 ? T2 : T3;
  const int dim = q.dim;
  // per dimension and alignment: for each candidate tile offset, the squared gap (in sigma units)
  std::vector<std::vector<std::vector<double> > > gap2(dim);
  for (int d = 0; d < dim; d++) {
    const int m = q.msize[d];
    gap2[d].resize(T[d]);
    for (int a = 0; a < T[d]; a++) {  // centre node at offset a inside its tile (tile origin = -a)
      const int first = (int)floor((double)(-m + a) / T[d]), last = (int)floor((double)(m + a) / T[d]);
      for (int k = first; k <= last; k++) {
        const double lo = (double)k * T[d] - a, hi = lo + T[d] - 1;  // tile node range relative to the centre node
        double gap = 0;
        if (lo > 1.0) gap = lo - 1.0;   // the hill lies in [0, 1) node units from its centre node
        if (hi < 0.0) gap = 0.0 - hi;
        gap *= q.dx[d] / q.sigma[d];
        gap2[d][a].push_back(gap * gap);
      }
    }
  }
  long long best = 0;
  const double cut = 8.0 * (1.0 + 1e-6);
  for (int a0 = 0; a0 < T[0]; a0++)
    for (int a1 = 0; a1 < (dim > 1 ? T[1] : 1); a1++)
      for (int a2 = 0; a2 < (dim > 2 ? T[2] : 1); a2++) {
        long long cnt = 0;
        const std::vector<double> &g0 = gap2[0][a0];
        for (size_t i = 0; i < g0.size(); i++) {
          if (dim == 1) {
            if (g0[i] < cut) cnt++;
            continue;
          }
          const std::vector<double> &g1 = gap2[1][a1];
          for (size_t j = 0; j < g1.size(); j++) {
            if (dim == 2) {
              if (g0[i] + g1[j] < cut) cnt++;
              continue;
            }
            const std::vector<double> &g2v = gap2[2][a2];
            for (size_t k = 0; k < g2v.size(); k++)
              if (g0[i] + g1[j] + g2v[k] < cut) cnt++;
          }
        }
        if (cnt > best) best = cnt;
      }
  // a periodic seam that cuts a partial tile splits one tile of the count in two, per such dimension
  for (int d = 0; d < dim; d++)
    if (q.periodic[d] && q.n[d] % T[d] != 0) best += best;  // (generous: rare geometry)
  return best;
}

// the read-back region of a batch released by its header line, copied out of the staging buffer (after its
// completion word has arrived: normally long ago) so that the next batch can reuse the buffer
static int save_pending_region(edm_hip_gauss *g) {
  if (!g->rb_pending_seq) return EDM_HIP_OK;
  volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(g->h_stage + g->h_stage_bytes - 128);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
  bool seen = false;
  for (unsigned spin = 0;; spin++) {
    if (w[0] >= g->rb_pending_seq) { seen = true; break; }
    __builtin_ia32_pause();
    if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
  }
  if (!seen) EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  g->rb_saved.assign(g->h_stage, g->h_stage + g->rb_pending_bytes);
  g->rb_pending_seq = 0;
  return EDM_HIP_OK;
}
int apply_hills_fetch_deferred(edm_hip_gauss *g, long long nh_bound, long long nh, std::vector<double> &pos,
                               std::vector<double> &added) {
  const int dim = g->g.dim;
  const size_t off_flags = 64;
  const size_t off_h2 = off_flags + ((sizeof(int) * (size_t)nh_bound + 7) & ~(size_t)7);
  const size_t off_added = off_h2 + 2 * sizeof(double) * (size_t)nh_bound;
  const size_t off_pos = off_added + sizeof(double) * (size_t)nh_bound;
  const size_t need = off_pos + sizeof(double) * (size_t)nh_bound * dim;
  const char *region = nullptr;
  if (g->rb_pending_seq) {
    // still in the host-mapped region (nobody has queued a batch since): wait for its completion word and take the two
    // slices that are wanted straight from there -- the region is laid out for the launch BOUND (23 KB on W1), the
    // hills that exist fill a fifth of it, and this sits in front of the next step's first launch
    if (g->rb_pending_bytes < need || nh > nh_bound) {
      set_error("apply_hills_fetch_deferred: no deferred read-back of that shape");
      return EDM_HIP_ERR_STATE;
    }
    volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(g->h_stage + g->h_stage_bytes - 128);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
    bool seen = false;
    for (unsigned spin = 0;; spin++) {
      if (w[0] >= g->rb_pending_seq) { seen = true; break; }
      __builtin_ia32_pause();
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
    }
    if (!seen) EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
    g->rb_pending_seq = 0;
    region = g->h_stage;
  } else {
    if (g->rb_saved.size() < need || nh > nh_bound) {
      set_error("apply_hills_fetch_deferred: no deferred read-back of that shape");
      return EDM_HIP_ERR_STATE;
    }
    region = g->rb_saved.data();
  }
  const double *a = reinterpret_cast<const double *>(region + off_added);
  const double *p = reinterpret_cast<const double *>(region + off_pos);
  added.assign(a, a + nh);
  pos.assign(p, p + (size_t)nh * dim);
  return EDM_HIP_OK;
}

int apply_hills(edm_hip_gauss *g, const ApplySpec &spec, ApplyOutcome *out, bool want_total) {
  {
    int rcs = save_pending_region(g);   // (a deferred read-back nobody has fetched yet: keep it before its buffer is reused)
    if (rcs) return rcs;
  }
  const Geom &q = g->g;
  HillWorkspace &ws = g->ws;
  const long long nh = spec.nh;
  if (out) {
    memset(&out->res, 0, sizeof(out->res));
    out->res.cum_out = spec.cum_in;
    out->res.k = nh;
    out->total_added = 0;
    out->flags.clear(); out->h2.clear(); out->a2.clear(); out->pos.clear(); out->added.clear(); out->heights.clear();
    out->first = nh;
    out->plain_fast = false;
    out->deferred_fetch = false;
    out->deferred_bound = 0;
    out->terms_emitted = false;
  }
  if (nh <= 0) return EDM_HIP_OK;
  hipStream_t s = g->stream;
  const int dim = q.dim;
  EDM_HIP_TRY(ws.hx.reserve((size_t)nh * dim));
  EDM_HIP_TRY(ws.hx0.reserve((size_t)nh * dim));
  EDM_HIP_TRY(ws.ht.reserve((size_t)nh * 2 * dim));
  EDM_HIP_TRY(ws.hc.reserve((size_t)nh * dim));
  EDM_HIP_TRY(ws.added.reserve((size_t)nh));
  EDM_HIP_TRY(ws.result.reserve(sizeof(LimitResult)));
  EDM_HIP_TRY(ws.tail_h1.reserve(EDM_TAIL_CAP));
  EDM_HIP_TRY(ws.tail_h2.reserve(EDM_TAIL_CAP));
  EDM_HIP_TRY(ws.tail_a2.reserve(EDM_TAIL_CAP));
  EDM_HIP_TRY(ws.tail_cum.reserve(EDM_TAIL_CAP));
  EDM_HIP_TRY(ws.tail_flags.reserve(EDM_TAIL_CAP));
  size_t scratch_need = limit_scratch_doubles(nh);
  if (scratch_need < lookup_scratch_doubles()) scratch_need = lookup_scratch_doubles();
  EDM_HIP_TRY(ws.scratch.reserve(scratch_need));

  // Small limited batches keep everything the host reads back -- limiter result, tail flags, undo
  // heights/bias, per-hill bias and original positions -- in ONE packed device region, so the
  // read-back is a single D2H copy.
  const bool small = (nh <= SMALL_BATCH) && out && spec.limited;
  LimitResult *dres = reinterpret_cast<LimitResult *>(ws.result.p);
  int *p_flags = ws.tail_flags.p;
  double *p_h2 = ws.tail_h2.p, *p_a2 = ws.tail_a2.p, *p_added = ws.added.p, *p_hx0 = ws.hx0.p;
  size_t rb_bytes = 0;
  const size_t off_flags = 64;
  const size_t off_h2 = off_flags + ((sizeof(int) * (size_t)nh + 7) & ~(size_t)7);
  const size_t off_a2 = off_h2 + sizeof(double) * (size_t)nh;
  const size_t off_added = off_a2 + sizeof(double) * (size_t)nh;
  const size_t off_pos = off_added + sizeof(double) * (size_t)nh;
  if (small) {
    rb_bytes = off_pos + sizeof(double) * (size_t)nh * dim;
    EDM_HIP_TRY(ws.rb.reserve(rb_bytes));
    char *base = ws.rb.p;
    dres = reinterpret_cast<LimitResult *>(base);
    p_flags = reinterpret_cast<int *>(base + off_flags);
    p_h2 = reinterpret_cast<double *>(base + off_h2);
    p_a2 = reinterpret_cast<double *>(base + off_a2);
    p_added = reinterpret_cast<double *>(base + off_added);
    p_hx0 = reinterpret_cast<double *>(base + off_pos);
  }
  if (out) out->d_added = p_added;

  HillList hl;
  hl.nh = nh;
  hl.x = spec.d_x;
  hl.x_stride = spec.x_stride;
  hl.pl_x = spec.pl_x;
  hl.pl_i = spec.pl_i;
  hl.pl_j = spec.pl_j;
  hl.sel = spec.d_sel;
  hl.hx = ws.hx.p;
  hl.hc = ws.hc.p;
  hl.ht = ws.ht.p;
  hl.hx0 = p_hx0;
  hl.nh_dev = spec.d_nh;
  {
    int rcb = ball_list_ensure(g);
    if (rcb) return rcb;
  }
  // a force kernel still pending (fix edm step) will read the lookup replica: settle the replica's state NOW -- built
  // if it is due -- before the gather plan below decides whether this batch's gather keeps it current (building it
  // later, when the kernel is launched, would leave a replica of the grid as it was BEFORE this batch flagged current)
  if (spec.forces && spec.forces->active && spec.forces->lookup) {
    const double *faces_now = nullptr;
    int rcf = faces_prepare(g, &faces_now);
    if (rcf) return rcf;
  }
  const Tables tabs = g->tables();
  // the launch that produces the prepared hill list (queued further down, once the gather plan is known: a short
  // fix edm_pair step runs selection, forces, integrals, limiter and gather as ONE launch instead)
  auto enqueue_preparation = [&]() -> int {
    if (spec.sel_chain) {
      // selection + preparation in one launch (and the step's pair forces with them, when they are pending)
      int rc = select_prep_enqueue(g, *spec.sel_chain, hl, spec.forces);
      if (rc) return rc;
    } else if (spec.unpack_chain) {
      int rc = pending_forces_flush(g, spec.forces);
      if (rc) return rc;
      EDM_HIP_TRY(launch_unpack_prep(*spec.unpack_chain, q, hl, s));  // exchange packets -> global prepared list
    } else {
      PendingForces *pf = spec.forces;
      double *fetch_dst = spec.h_fetch_src ? const_cast<double *>(spec.d_h) : nullptr;
      static const bool fuse_env = !test_force("no_lookup_prep");
      if (pf && pf->active && pf->lookup && fuse_env && pf->la.n > 0 && lookup_prep_fusable(q, hl)) {
        // fix edm step: the pending force kernel (K2) and this list's preparation share a launch
        pf->active = false;
        hipEvent_t e0, e1;
        profile_slot(g, &e0, &e1);
        const double *faces = nullptr;
        int rcf = faces_prepare(g, &faces);
        if (rcf) return rcf;
        EDM_HIP_TRY(launch_lookup_prep(q, g->rec, pf->la, g->d_partials, s, e0, e1, &pf->nblk, faces, hl, spec.h_fetch_src, fetch_dst));
        pf->launched();
        g->lookup_prep_launches++;
        return EDM_HIP_OK;
      }
      int rc = pending_forces_flush(g, pf);   // (a pending force kernel goes ahead of everything this batch queues)
      if (rc) return rc;
      EDM_HIP_TRY(launch_hill_prep(q, hl, s, spec.h_fetch_src, fetch_dst));
    }
    return EDM_HIP_OK;
  };

  HillHeights hh;
  hh.h = spec.d_h;
  hh.h_const = spec.h_const;
  hh.k = nh;
  hh.tail_h1 = ws.tail_h1.p;
  hh.tail_h2 = p_h2;
  hh.res_dev = nullptr;
  hh.tail_shift = 0;

  // Dense batches on small grids (the 1-D all-samples regime): fused path -- the gather runs first with
  // the base heights and yields the per-hill integrals as a by-product; the limiter then only has to
  // correct the hills it changed.
  const long long ntiles_all = gather_tiles(q);
  const bool fused = !spec.ordered && !spec.d_nh && (nh >= 4096) && (ntiles_all < 1024);
  GatherPlan fplan;
  memset(&fplan, 0, sizeof(fplan));
  if (fused) {
    long long G = (2048 + ntiles_all - 1) / ntiles_all;
    if (G > 64) G = 64;
    if (G > nh / 256) G = nh / 256;
    if (G < 1) G = 1;
    fplan.groups = (int)G;
    fplan.slots_per_hill = gather_slots_per_hill(q);
    EDM_HIP_TRY(ws.partial.reserve((size_t)(G + 1) * (size_t)q.total * q.rec));
    EDM_HIP_TRY(ws.slots.reserve((size_t)nh * fplan.slots_per_hill));
    fplan.partial = ws.partial.p;
    fplan.slots = ws.slots.p;
  }
  bool rb_pushed = false;   // the packed read-back region is copied to host-mapped memory by a kernel of the chain
  bool polled = false;      // ... and flagged there: the host polls the flag instead of waiting for the stream
  const double *base_heights = spec.d_h;
  const bool chain_limit = spec.limited && !spec.ordered && !fused && hill_integrals_can_chain_limit(nh);
  // sharded (multi-GPU) variant of the fused path
  const bool sharded = fused && (spec.shard_comm || spec.shard_virtual > 1);
  std::vector<std::pair<long long, long long> > slices;
  const size_t grid_doubles = (size_t)q.total * q.rec;
  auto slice_of = [&](const std::pair<long long, long long> &sl) {
    HillList own = hl;
    own.nh = sl.second;
    own.x = nullptr;
    own.sel = nullptr;
    own.hx = hl.hx + sl.first * dim;
    own.hc = hl.hc + sl.first * dim;
    own.ht = hl.ht + sl.first * 2 * dim;
    own.hx0 = hl.hx0 ? hl.hx0 + sl.first * dim : nullptr;
    return own;
  };
  // gather plan: hill groups when a small grid meets a long hill list, tile culling
  // when a large grid meets a short one
  GatherPlan plan;
  plan.groups = 1;
  plan.adaptive = 0;
  plan.partial = nullptr;
  plan.tile_flags = nullptr;
  plan.tile_list = nullptr;
  plan.tile_parity = 0;
  plan.tile_bound = 0;
  plan.tiles_marked = 0;
  plan.compact_waves = 1;
  const long long ntiles = gather_tiles(q);
  if (fused) {
    // planned above
  } else if (nh >= 4096 && ntiles < 1024) {
    long long G = (2048 + ntiles - 1) / ntiles;
    if (G > 64) G = 64;
    if (G > nh / 256) G = nh / 256;
    if (G < 1) G = 1;
    plan.groups = (int)G;
    if (G > 1) {
      EDM_HIP_TRY(ws.partial.reserve((size_t)G * (size_t)q.total * q.rec));
      plan.partial = ws.partial.p;
    }
  } else if (ntiles < 128) {
    // a few hundred hills on a grid of a few dozen tiles (the stochastic 1-D step): the tile loop is a
    // serial chain per node, so spread it over up to ~256 workgroups.  The split is a function of the
    // true hill count alone (resolved on the device when the count is deferred).
    long long G = (256 + ntiles - 1) / ntiles;
    if (G > 16) G = 16;
    // (the cap follows the EXPECTED batch size, which does not depend on how the batch was queued)
    const double expected = spec.expected_nh >= 0 ? spec.expected_nh : (double)nh;
    if (expected < 512) G = 1;
    if (G > nh / 128) G = nh / 128;
    if (G < 1) G = 1;
    plan.groups = (int)G;
    plan.adaptive = 1;
    if (G > 1) {
      EDM_HIP_TRY(ws.partial.reserve((size_t)G * (size_t)q.total * q.rec));
      plan.partial = ws.partial.p;
    }
  } else if (ntiles > 2048) {
    if (g->tiles_per_hill <= 0) g->tiles_per_hill = tiles_per_hill_bound(q);
    long long per_hill = g->tiles_per_hill;
    bool narrow = true;  // (a stencil wider than a periodic dimension crosses more than one seam: no culling)
    for (int d = 0; d < q.dim; d++)
      if (q.periodic[d] && 2 * q.msize[d] + 1 > q.n[d]) narrow = false;
    // a batch queued against a launch bound (deferred count) is sized by its EXPECTED hill count: the
    // workgroups of a culled gather stride over the tile list, so an optimistic launch stays correct
    long long plan_nh = nh;
    if (spec.d_nh && spec.expected_nh >= 0) {
      const long long e = (long long)(1.5 * spec.expected_nh) + 64;
      if (e < plan_nh) plan_nh = e;
    }
    if (narrow && plan_nh * per_hill < ntiles / 2) {
      EDM_HIP_TRY(ws.tile_flags.reserve_zeroed((size_t)ntiles));
      EDM_HIP_TRY(ws.tile_list.reserve_zeroed((size_t)ntiles + 2));
      plan.tile_flags = ws.tile_flags.p;
      plan.tile_list = ws.tile_list.p;
      plan.tile_parity = ws.tile_parity;
      ws.tile_parity ^= 1;
      plan.tile_bound = plan_nh * per_hill;
    }
  }
  // lookup replica: the in-place gather keeps it current; every other way of applying the batch leaves it stale
  plan.faces = nullptr;
  plan.slots = nullptr;
  plan.slots_per_hill = 0;
  if (g->faces && g->faces_state == 1) {
    if (!fused && !sharded && !spec.ordered && plan.groups == 1)
      plan.faces = g->faces;
    else
      g->faces_state = 2;
  }
  const bool fused_post = spec.limited && spec.hist_g && spec.hist_values;
  bool chain_post = false;
  bool gather_done = false;   // the gather rode in the integrals' launch (launch_integrals_gather)
  {
    int rcp = enqueue_preparation();
    if (rcp) return rcp;
  }
  const unsigned long long *hook_ready_flag = nullptr;
  unsigned long long hook_ready_seq = 0;
  if (sharded) {
    if (spec.shard_comm) {
      if (spec.shard_off < 0 || spec.shard_cnt < 0 || spec.shard_off + spec.shard_cnt > nh) {
        set_error("apply_hills: shard outside the hill list");
        return EDM_HIP_ERR_ARG;
      }
      slices.push_back(std::make_pair(spec.shard_off, spec.shard_cnt));
    } else {
      const long long R = spec.shard_virtual, per = (nh + R - 1) / R;
      for (long long r = 0; r < R; r++) {
        const long long o = r * per < nh ? r * per : nh;
        const long long c = (o + per < nh ? o + per : nh) - o;
        slices.push_back(std::make_pair(o, c));
      }
    }
    EDM_HIP_TRY(ws.delta.reserve(grid_doubles));
    EDM_HIP_TRY(hipMemsetAsync(ws.delta.p, 0, sizeof(double) * grid_doubles, s));
    EDM_HIP_TRY(hipMemsetAsync(p_added, 0, sizeof(double) * (size_t)nh, s));
    for (size_t k = 0; k < slices.size(); k++) {
      const HillList own = slice_of(slices[k]);
      if (own.nh <= 0) continue;
      EDM_HIP_TRY(launch_hill_gather_fused(q, tabs, own, spec.d_h ? spec.d_h + slices[k].first : nullptr, spec.h_const, fplan,
                                           p_added + slices[k].first, g->d_dirty, s));
      EDM_HIP_TRY(launch_add_partials(q, ws.delta.p, fplan.partial, fplan.groups, nullptr, s));
    }
    if (spec.shard_comm) {
      // every rank holds the integrals of its own slice: ONE all-gather of the (padded) slices completes the list
      // on all ranks -- half the bytes of the all-reduce of the zero-filled full-length array this replaces
      Transport *tr = static_cast<Transport *>(spec.shard_comm);
      const int N = tr->nranks();
      long long maxc = 0, run = 0;
      for (int r = 0; r < N; r++) {
        const long long c = spec.shard_counts ? spec.shard_counts[r] : 0;
        if (c > maxc) maxc = c;
        run += c;
      }
      if (!spec.shard_counts || run != nh) {
        set_error("apply_hills: the ranks' hill counts do not add up to the global list");
        return EDM_HIP_ERR_ARG;
      }
      EDM_HIP_TRY(ws.gath.reserve((size_t)maxc * (size_t)(N + 1)));
      double *sendb = ws.gath.p + (size_t)maxc * N;   // (own slice, padded to the common length)
      EDM_HIP_TRY(hipMemsetAsync(sendb, 0, sizeof(double) * (size_t)maxc, s));
      if (spec.shard_cnt > 0)
        EDM_HIP_TRY(hipMemcpyAsync(sendb, p_added + spec.shard_off, sizeof(double) * (size_t)spec.shard_cnt, hipMemcpyDeviceToDevice, s));
      int rcg = tr->all_gather(sendb, ws.gath.p, sizeof(double) * (size_t)maxc, s);
      if (rcg) return rcg;
      long long off = 0;
      for (int r = 0; r < N; r++) {
        const long long c = spec.shard_counts[r];
        if (c > 0)
          EDM_HIP_TRY(hipMemcpyAsync(p_added + off, ws.gath.p + (size_t)r * maxc, sizeof(double) * (size_t)c, hipMemcpyDeviceToDevice, s));
        off += c;
      }
    }
  } else if (fused) {
    EDM_HIP_TRY(launch_hill_gather_fused(q, tabs, hl, spec.d_h, spec.h_const, fplan, p_added, g->d_dirty, s));
  } else if (spec.ordered) {
    if (nh > EDM_TAIL_CAP) {
      set_error("ordered (locally tempered) hill batches are limited to EDM_TAIL_CAP hills per step");
      return EDM_HIP_ERR_OVERFLOW;
    }
    EDM_HIP_TRY(ws.heights.reserve((size_t)nh));
    OrderedParams op = spec.op;
    op.limit = spec.limit;
    op.cum_in = spec.cum_in;
    LimitTail tail{ws.tail_h1.p, p_h2, p_a2, ws.tail_cum.p, p_flags};
    EDM_HIP_TRY(launch_hills_ordered(q, tabs, g->rec, hl, op, tail, ws.heights.p, p_added, dres, g->d_dirty, s));
    base_heights = ws.heights.p;
  } else if (chain_limit) {
    // short batch: the limiter runs in the last integrals workgroup (one launch instead of two)
    LimitArgs la;
    memset(&la, 0, sizeof(la));
    la.ticket = g->d_tickets + 1 * EDM_TICKET_INTS;
    la.limit = spec.limit;
    la.cum_in = spec.cum_in;
    la.flush_mode = spec.flush_mode;
    la.tail = LimitTail{ws.tail_h1.p, p_h2, p_a2, ws.tail_cum.p, p_flags};
    la.res = dres;
    // the hills' workgroups hand their integrals to the limiter's workgroup as tagged 16-byte stores (LimitArgs::tagged)
    static const bool tagged_env = !test_force("no_tagged_integrals");   // (tests: the last-arrival ticket, which batches of more than 1024 hills take anyway)
    if (tagged_env && nh <= EDM_TAG_CAP_HOST) {
      EDM_HIP_TRY(ws.tagged.reserve_zeroed(2 * (size_t)EDM_TAG_CAP_HOST));
      la.tagged = ws.tagged.p;
      la.tag_seq = ++ws.tag_seq;
    }
    la.expected_hills = (spec.d_nh && spec.expected_nh >= 0) ? (long long)(spec.expected_nh + 0.5) : nh;
    la.shared_device = g->shared_device ? 1 : 0;
    la.tiles_first_mode = g->debug_tiles_first;
    if (small && rb_bytes + 128 <= g->h_stage_bytes) {
      // ... and so does the read-back: everything the host reads is final once the limiter has run, so the
      // same workgroup copies the packed region into host-mapped memory and flags it; the host polls the flag
      // (see below) and the gather runs on behind the host's back
      la.rb_src = ws.rb.p;
      la.rb_dst = g->d_stage;
      rb_pushed = true;
      if (poll_enabled()) {
        la.done_flag = reinterpret_cast<unsigned long long *>(g->d_stage + g->h_stage_bytes - 128);
        la.done_seq = ++g->done_seq;
        polled = true;
        // (the 128 bytes in front of the completion word: the limiter's result as one line, see the poll below)
        if (rb_bytes + 256 <= g->h_stage_bytes) la.fast_line = reinterpret_cast<unsigned long long *>(g->d_stage + g->h_stage_bytes - 256);
      }
    }
    hh.res_dev = dres;
    if (integrals_gather_fusable(q, nh, plan)) {
      // 1-D: integrals + limiter and the in-place gather side by side in ONE launch (the gather waits for the
      // limiter's word only before it applies heights); bookkeeping chained onto the gather as before
      if (!g->d_ready) {
        EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_ready), 128));
        EDM_HIP_TRY(hipMemset(g->d_ready, 0, 128));
      }
      la.ready_flag = g->d_ready;
      la.ready_seq = ++g->ready_seq;
      la.early_word = 1;
      static const bool tracing = getenv("EDM_HIP_TRACE") != nullptr;   // development aid: stamps of one launch to stderr
      const size_t trace_wgs = (size_t)nh + (size_t)((q.n[0] + 31) / 32);
      unsigned long long *d_trace = nullptr;
      if (tracing && g->ready_seq == 150) {
        EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_trace), trace_wgs * 64));
        EDM_HIP_TRY(hipMemset(d_trace, 0, trace_wgs * 64));
        la.trace = d_trace;
      }
      PostSpec ps;
      ps.ticket = g->d_tickets + 2 * EDM_TICKET_INTS;
      ps.hist_geom = spec.hist_g;
      ps.hist = spec.hist_values;
      ps.flags = p_flags;
      ps.flush_mode = spec.flush_mode;
      ps.rb_src = nullptr;
      ps.rb_dst = nullptr;
      ps.rb_bytes = 0;
      chain_post = fused_post && nh <= 4096;
      if (!rb_pushed && chain_post && small && rb_bytes + 128 <= g->h_stage_bytes) {
        ps.rb_src = ws.rb.p;
        ps.rb_dst = g->d_stage;
        ps.rb_bytes = (long long)rb_bytes;
        rb_pushed = true;
      }
      ht_mark(g, 3);
      if (spec.ord_terms && !spec.flush_mode) {   // (a reference-order step: the launch also stores the hills' stencil terms)
        la.ord_terms = spec.ord_terms;
        la.ord_dirty = spec.ord_dirty;
        la.ord_seq = spec.ord_seq;
        la.ord_ready = spec.ord_ready;
        if (out) out->terms_emitted = true;
        hook_ready_flag = la.ready_flag;
        hook_ready_seq = la.ready_seq;
      }
      EDM_HIP_TRY(launch_integrals_gather(q, tabs, g->rec, hl, spec.d_h, spec.h_const, p_added, la, hh, plan, g->d_dirty, s,
                                          chain_post ? &ps : nullptr));
      ht_mark(g, 4);
      gather_done = true;
      if (d_trace) {
        EDM_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> tr(trace_wgs * 8);
        EDM_HIP_TRY(hipMemcpy(tr.data(), d_trace, trace_wgs * 64, hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < trace_wgs; w++)
          if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8];
        const char *names_i[8] = {"start", "integral done", "last wg: ticket", "flag published", "host copy done", "", "", "end"};
        const char *names_g[8] = {"start", "terms parked", "flag seen", "", "", "", "body end", "post end"};
        for (int role = 0; role < 2; role++) {
          for (int k = 0; k < 8; k++) {
            std::vector<double> v;
            for (size_t w = role ? (size_t)nh : 0; w < (role ? trace_wgs : (size_t)nh); w++)
              if (tr[w * 8 + k]) v.push_back((double)(tr[w * 8 + k] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[edm trace] %s %-16s n=%4zu  min %6.2f  med %6.2f  max %6.2f us\n", role ? "gather   " : "integrals",
                    role ? names_g[k] : names_i[k], v.size(), v.front(), v[v.size() / 2], v.back());
          }
        }
      }
    } else {
      static const bool tracing_nd = getenv("EDM_HIP_TRACE") != nullptr;   // development aid: stamps of one launch to stderr
      static int traced_launches = 0;
      unsigned long long *d_trace = nullptr;
      if (tracing_nd && ++traced_launches >= 40 && traced_launches <= 43) {
        EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_trace), (size_t)nh * 64));
        EDM_HIP_TRY(hipMemset(d_trace, 0, (size_t)nh * 64));
        la.trace = d_trace;
      }
      // a culled gather follows (2-D / 3-D, short list on a big grid): its tile list is built by extra workgroups of
      // this launch instead of a launch of its own (the list needs the prepared hills, nothing of the integrals)
      MarkArgs mark{plan.tile_flags, plan.tile_list, ntiles, plan.tile_parity};
      const bool mark_here = dim > 1 && plan.tile_flags && plan.tile_list && plan.groups == 1 && !sharded && !fused && !spec.ordered;
      EDM_HIP_TRY(launch_hill_integrals(q, tabs, hl, spec.d_h, spec.h_const, p_added, s, &la, mark_here ? &mark : nullptr));
      if (mark_here) plan.tiles_marked = 1;
      if (d_trace) {
        EDM_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> tr((size_t)nh * 8);
        EDM_HIP_TRY(hipMemcpy(tr.data(), d_trace, (size_t)nh * 64, hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < (size_t)nh; w++)
          if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8];
        const char *names_i[8] = {"start", "integral done", "last wg: ticket", "", "limiter stage done", "walk begins", "", "end"};
        fprintf(stderr, "[edm trace] k_hill_integrals<%d>, launch bound %lld hills, flush %d\n", dim, nh, spec.flush_mode);
        for (int k = 0; k < 8; k++) {
          std::vector<double> v;
          for (size_t w = 0; w < (size_t)nh; w++)
            if (tr[w * 8 + k]) v.push_back((double)(tr[w * 8 + k] - t0) * 0.01);
          if (v.empty()) continue;
          std::sort(v.begin(), v.end());
          fprintf(stderr, "[edm trace] integrals %-18s n=%4zu  min %6.2f  med %6.2f  max %6.2f us\n", names_i[k], v.size(), v.front(),
                  v[v.size() / 2], v.back());
        }
      }
    }
  } else if (spec.limited || want_total) {
    EDM_HIP_TRY(launch_hill_integrals(q, tabs, hl, spec.d_h, spec.h_const, p_added, s));
  }
  if (spec.ordered || chain_limit) {
    // everything was applied by the ordered kernel / the limiter was chained above
  } else if (spec.limited) {
    LimitTail tail{ws.tail_h1.p, p_h2, p_a2, ws.tail_cum.p, p_flags};
    EDM_HIP_TRY(launch_limit(nh, p_added, spec.d_h, spec.h_const, spec.limit, spec.cum_in, spec.flush_mode, tail,
                             dres, ws.scratch.p, s, spec.d_nh));
    hh.res_dev = dres;  // the gather reads k on the device: no host round trip here
  } else if (want_total) {
    EDM_HIP_TRY(launch_sum(nh, p_added, g->d_scalars + 1, ws.scratch.p, s));
  }

  if (sharded) {
    if (spec.limited) {
      for (size_t k = 0; k < slices.size(); k++) {
        const HillList own = slice_of(slices[k]);
        HillHeights hs = hh;
        hs.h = spec.d_h ? spec.d_h + slices[k].first : nullptr;
        hs.tail_shift = slices[k].first;
        EDM_HIP_TRY(launch_hill_gather_correction(q, tabs, own, hs, fplan, g->d_dirty, s));
        EDM_HIP_TRY(launch_add_partials(q, ws.delta.p, fplan.partial + (size_t)fplan.groups * grid_doubles, 1, dres, s));
      }
    }
    if (spec.shard_comm) {
      int rcd = static_cast<Transport *>(spec.shard_comm)->all_reduce_sum(ws.delta.p, grid_doubles, s);   // the RCCL bias-grid all-reduce
      if (rcd) return rcd;
    }
    EDM_HIP_TRY(launch_add_partials(q, g->rec, ws.delta.p, 1, spec.limited ? dres : nullptr, s));
    // whether some hill of some rank had a boundary correction is not known locally: duplicate whenever
    // the boundary is not periodic (idempotent on a consistent grid)
    bool any_wall = false;
    for (int d = 0; d < dim; d++)
      if (!q.bper[d]) any_wall = true;
    if (any_wall) EDM_HIP_TRY(hipMemsetAsync(g->d_dirty, 1, 1, s));
    if (!fused_post) EDM_HIP_TRY(launch_duplicate_boundary(q, g->rec, g->d_dirty, s));
  } else if (fused) {
    EDM_HIP_TRY(launch_hill_gather_correct_and_apply(q, tabs, g->rec, hl, hh, fplan, spec.limited ? 1 : 0, g->d_dirty, s));
    if (!fused_post) EDM_HIP_TRY(launch_duplicate_boundary(q, g->rec, g->d_dirty, s));
  } else if (gather_done) {
    if (!fused_post) EDM_HIP_TRY(launch_duplicate_boundary(q, g->rec, g->d_dirty, s));
  } else if (!spec.ordered) {
    // short limited batch applied in place: boundary duplication and histogram ride on the gather launch
    // A long list on a large 2-D/3-D grid: without culling every tile workgroup would scan every hill.  The
    // in-place gather adds the hills of a node in list order, so the list can be applied as consecutive
    // sub-batches, each short enough for the culled launch -- same sums, a fraction of the scanning.
    long long sub = 0;
    if (!spec.d_nh && plan.groups == 1 && !plan.tile_list && ntiles > 2048) {
      if (g->tiles_per_hill <= 0) g->tiles_per_hill = tiles_per_hill_bound(q);
      bool narrow = true;
      for (int d = 0; d < q.dim; d++)
        if (q.periodic[d] && 2 * q.msize[d] + 1 > q.n[d]) narrow = false;
      const long long fit = (ntiles / 2 - 1) / g->tiles_per_hill;   // hills per culled launch
      if (narrow && fit >= 256 && nh > fit) sub = fit;
    }
    chain_post = fused_post && plan.groups == 1 && hh.res_dev && nh <= 4096 && sub == 0;
    PostSpec ps;
    ps.ticket = g->d_tickets + 2 * EDM_TICKET_INTS;
    ps.hist_geom = spec.hist_g;
    ps.hist = spec.hist_values;
    ps.flags = p_flags;
    ps.flush_mode = spec.flush_mode;
    // ... and so does the read-back: the last workgroup copies the packed region into host-mapped memory
    ps.rb_src = nullptr;
    ps.rb_dst = nullptr;
    ps.rb_bytes = 0;
    if (!rb_pushed && chain_post && small && rb_bytes + 128 <= g->h_stage_bytes) {
      ps.rb_src = ws.rb.p;
      ps.rb_dst = g->d_stage;
      ps.rb_bytes = (long long)rb_bytes;
      rb_pushed = true;
    }
    if (sub > 0) {
      EDM_HIP_TRY(ws.tile_flags.reserve_zeroed((size_t)ntiles));
      EDM_HIP_TRY(ws.tile_list.reserve_zeroed((size_t)ntiles + 2));
      for (long long off = 0; off < nh; off += sub) {
        const long long cnt = (off + sub < nh) ? sub : nh - off;
        HillList part = hl;
        part.nh = cnt;
        part.x = nullptr;
        part.sel = nullptr;
        part.hx = hl.hx + off * dim;
        part.hc = hl.hc + off * dim;
        part.ht = hl.ht + off * 2 * dim;
        part.hx0 = hl.hx0 ? hl.hx0 + off * dim : nullptr;
        HillHeights hp = hh;
        hp.h = hh.h ? hh.h + off : nullptr;
        if (!hh.res_dev) hp.k = cnt;   // (unlimited batch: every hill at its base height)
        hp.tail_shift = off;
        GatherPlan pp = plan;
        pp.tile_flags = ws.tile_flags.p;
        pp.tile_list = ws.tile_list.p;
        pp.tile_parity = ws.tile_parity;
        ws.tile_parity ^= 1;
        pp.tile_bound = cnt * g->tiles_per_hill;
        EDM_HIP_TRY(launch_hill_gather(q, tabs, g->rec, part, hp, pp, g->d_dirty, s, nullptr));
      }
    } else {
      EDM_HIP_TRY(launch_hill_gather(q, tabs, g->rec, hl, hh, plan, g->d_dirty, s, chain_post ? &ps : nullptr));
    }
    if (!fused_post) EDM_HIP_TRY(launch_duplicate_boundary(q, g->rec, g->d_dirty, s));
  }

  // CV histogram (edm_bias.cpp:601-610): new hills log one 'h' line each (+1) and a 'u' line
  // per undo (-1); a flush logs 'b' (+1) for replayed hills only and 'v' (-1) for its undo
  if (chain_post) {
    // done by the gather's last workgroup
  } else if (fused_post) {
    EDM_HIP_TRY(launch_post_batch(q, g->rec, g->d_dirty, *spec.hist_g, spec.hist_values, nh, p_hx0, dres, p_flags,
                                  spec.flush_mode, s));
  } else if (spec.hist_g && spec.hist_values) {
    if (!spec.limited || !spec.flush_mode)
      EDM_HIP_TRY(launch_hist_add(*spec.hist_g, spec.hist_values, nh, p_hx0, dim, nullptr, nullptr, 1.0, s));
    if (spec.limited)
      EDM_HIP_TRY(launch_hist_tail(*spec.hist_g, spec.hist_values, dres, p_flags, p_hx0, spec.flush_mode ? 1 : 0, s));
  }

  // ---- read-back ----
  LimitResult res;
  memset(&res, 0, sizeof(res));
  res.k = nh;
  res.cum_out = spec.cum_in;
  LimitResult *hres = reinterpret_cast<LimitResult *>(g->h_scalars + 8);
  char *stage = g->h_stage;
  if (small) {
    if (!rb_pushed) EDM_HIP_TRY(hipMemcpyAsync(stage, ws.rb.p, rb_bytes, hipMemcpyDeviceToHost, s));
    hres = reinterpret_cast<LimitResult *>(stage);
  } else if (spec.limited) {
    EDM_HIP_TRY(hipMemcpyAsync(hres, dres, sizeof(LimitResult), hipMemcpyDeviceToHost, s));
  }
  if (want_total && !spec.limited)
    EDM_HIP_TRY(hipMemcpyAsync(g->h_scalars + 1, g->d_scalars + 1, sizeof(double), hipMemcpyDeviceToHost, s));
  const int *st_flags = reinterpret_cast<const int *>(stage + off_flags);
  const double *st_h2 = reinterpret_cast<const double *>(stage + off_h2);
  const double *st_a2 = reinterpret_cast<const double *>(stage + off_a2);
  const double *st_added = reinterpret_cast<const double *>(stage + off_added);
  const double *st_pos = reinterpret_cast<const double *>(stage + off_pos);
  g->wait_polled = false;
  bool plain_fast = false;
  LimitResult header_res;
  memset(&header_res, 0, sizeof(header_res));
  static const bool host_trace = getenv("EDM_HIP_TRACE") != nullptr;   // development aid: host-side stamps
  if (spec.before_wait)
    spec.before_wait(spec.before_wait_ctx, base_heights, ws.tail_h1.p, p_h2, dres, out ? out->terms_emitted : false,
                     hook_ready_flag, hook_ready_seq, spec.d_nh);
  const auto ht_before_poll = std::chrono::steady_clock::now();
  ht_mark(g, 5);
  if (polled) {
    // the limiter's workgroup flags the host-mapped region once it is complete: poll the word instead of waiting
    // for the stream's completion signal (bounded; falls back to the stream wait)
    volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(g->h_stage + g->h_stage_bytes - 128);
    // ... or, sooner, its header line: the limiter's wave writes the 64-byte result to the host with ONE instruction,
    // the batch's number in its last two words (LimitResult).  If that line says every hill was added in full and the
    // caller wants neither positions nor per-hill bias, nothing else of the region is needed and the call returns
    // ~3 us before the acknowledgements of the region's other stores would let the completion word out.
    volatile unsigned long long *hd = reinterpret_cast<volatile unsigned long long *>(g->h_stage + g->h_stage_bytes - 256);
    static const bool header_env = !test_force("no_fast_header");   // (tests: every polled batch waits for its completion word)
    // (A flush of the overflow buffer: the line carries the stop index and that hill's undo height -- all the host's
    // replay needs when there is no log.)
    // (With a HILLS log the caller needs positions and per-hill bias too -- but not NOW: defer_fetch_ok says it will ask
    //  for them later, see ApplyOutcome::deferred_fetch.  New hills only; a flush's log lines carry per-hill heights.)
    const bool defer = spec.fetch_all && spec.defer_fetch_ok && !spec.flush_mode && !spec.d_h && !spec.ordered && small;
    bool header_may_do = header_env && spec.limited && (!spec.fetch_all || defer) && out != nullptr &&
                         rb_bytes + 256 <= g->h_stage_bytes;
    const unsigned long long want = g->done_seq;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
    bool seen = false;
    for (unsigned spin = 0;; spin++) {
      if (header_may_do && hd[0] == want && hd[7] == want) {
        std::atomic_thread_fence(std::memory_order_acquire);
        unsigned long long line[8];
        for (int i = 0; i < 8; i++) line[i] = hd[i];
        LimitResult hr;
        if (edm_header_line_decode(line, want, &hr)) {
          if ((hr.all_plain || spec.flush_mode) && !hr.error) {
            if (spec.fetch_all && !hr.all_plain) {   // (only a batch the limiter left alone defers its log)
              header_may_do = false;
              continue;
            }
            header_res = hr;
            plain_fast = true;
            g->header_releases++;
            seen = true;
            break;
          }
        }
        header_may_do = false;   // (the limiter changed something: the rest of the region is needed)
      }
      if (w[0] == want) { seen = true; break; }
      __builtin_ia32_pause();
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (seen) {
      g->wait_polled = true;
      g->polled_batches++;
    } else {
      g->poll_fallbacks++;
      EDM_HIP_TRY(hipStreamSynchronize(s));
    }
  } else {
    EDM_HIP_TRY(hipStreamSynchronize(s));
  }
  if (host_trace && g->ready_seq == 160) {
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[edm host] apply_hills: polling took %.2f us (entered the poll %.2f us after a reference point)\n",
            std::chrono::duration<double, std::micro>(now - ht_before_poll).count(),
            std::chrono::duration<double, std::micro>(ht_before_poll.time_since_epoch()).count() - g->ht_ref_us);
  }
  ht_mark(g, 6);
  long long nh_act = nh;
  if (spec.limited) {
    res = plain_fast ? header_res : *hres;
    if (res.error == 2) {
      if (out) out->res = res;
      return EDM_APPLY_BOUND_EXCEEDED;  // deferred count above the launch bound: nothing was applied
    }
    if (res.error) {
      set_error("The bias overflow buffer is full. Too many hills. Either increase & recompile, lower hill_density, or lower bias");
      return EDM_HIP_ERR_OVERFLOW;
    }
    nh_act = res.nh;
  }
  if (out) {
    out->res = res;
    out->d_base_heights = base_heights;
    out->d_tail_h1 = ws.tail_h1.p;
    out->d_tail_h2 = p_h2;
    if (want_total && !spec.limited) out->total_added = g->h_scalars[1];
    if (base_heights && spec.limited && spec.fetch_heights) {
      const long long f0 = spec.fetch_all ? 0 : res.k;
      if (nh_act - f0 > 0) {
        out->heights.resize((size_t)(nh_act - f0));
        EDM_HIP_TRY(hipMemcpy(out->heights.data(), base_heights + f0, sizeof(double) * (size_t)(nh_act - f0), hipMemcpyDeviceToHost));
      }
    }
    const long long k = res.k;
    const int ntail = spec.limited ? res.n_tail : 0;
    const long long first = spec.fetch_all ? 0 : k;
    out->first = first;
    const long long need = nh_act - first;
    if (plain_fast) {
      // (new hills: flags 1, undo heights 0, undo bias 0 throughout; a flush: res.stop / res.h2_stop; positions and
      // per-hill bias not asked for -- or asked for later)
      out->plain_fast = true;
      if (spec.fetch_all) {
        out->deferred_fetch = true;
        out->deferred_bound = nh;
        g->rb_pending_seq = g->done_seq;
        g->rb_pending_bytes = rb_bytes;
      }
    } else if (small) {
      out->flags.assign(st_flags, st_flags + ntail);
      out->h2.assign(st_h2, st_h2 + ntail);
      out->a2.assign(st_a2, st_a2 + ntail);
      out->pos.assign(st_pos + (size_t)first * dim, st_pos + (size_t)nh_act * dim);
      out->added.assign(st_added + first, st_added + nh_act);
    } else if (spec.limited || spec.fetch_all) {
      if (ntail > 0) {
        out->flags.resize((size_t)ntail);
        out->h2.resize((size_t)ntail);
        out->a2.resize((size_t)ntail);
        EDM_HIP_TRY(hipMemcpy(out->flags.data(), p_flags, sizeof(int) * (size_t)ntail, hipMemcpyDeviceToHost));
        EDM_HIP_TRY(hipMemcpy(out->h2.data(), p_h2, sizeof(double) * (size_t)ntail, hipMemcpyDeviceToHost));
        EDM_HIP_TRY(hipMemcpy(out->a2.data(), p_a2, sizeof(double) * (size_t)ntail, hipMemcpyDeviceToHost));
      }
      if (need > 0) {
        out->pos.resize((size_t)need * dim);
        EDM_HIP_TRY(hipMemcpy(out->pos.data(), p_hx0 + (size_t)first * dim, sizeof(double) * (size_t)need * dim,
                              hipMemcpyDeviceToHost));
        out->added.resize((size_t)need);
        EDM_HIP_TRY(hipMemcpy(out->added.data(), p_added + first, sizeof(double) * (size_t)need, hipMemcpyDeviceToHost));
      }
    }
  }
  return EDM_HIP_OK;
}

}  // namespace edm

extern "C" {

int edm_hip_gauss_add_values(edm_hip_gauss *g, long long n, const double *d_x, int x_stride, const double *d_h,
                             double h_const, double *d_added, double *total_added) {
  if (total_added) *total_added = 0;
  if (n <= 0) return EDM_HIP_OK;
  ApplySpec spec;
  spec.nh = n;
  spec.d_x = d_x;
  spec.x_stride = x_stride;
  spec.d_h = d_h;
  spec.h_const = h_const;
  ApplyOutcome outc;
  const bool want = (d_added != nullptr) || (total_added != nullptr);
  // A short batch whose total is wanted goes through the chained limiter with a limit nothing reaches: integrals +
  // "limiter" + tile list in one launch and the total on the limiter's header line, polled -- instead of integrals,
  // a sum launch, a tile-list launch, and a copy behind a stream wait.  (The total is then the limiter's ordered sum.)
  static const bool chain_env = !test_force("no_add_values_chain");   // (tests: the launches long lists take)
  const bool chained = want && chain_env && hill_integrals_can_chain_limit(n);
  if (chained) {
    spec.limited = true;
    spec.limit = std::numeric_limits<double>::infinity();
    spec.cum_in = 0;
    spec.flush_mode = 0;
    spec.fetch_heights = false;
  }
  int rc = apply_hills(g, spec, &outc, want);
  if (rc) return rc;
  // (hipMemcpy on the null stream: ordered behind the object's stream, on which the integrals ran)
  if (d_added) EDM_HIP_TRY(hipMemcpy(d_added, outc.d_added ? outc.d_added : g->ws.added.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice));
  if (total_added) *total_added = chained ? outc.res.cum_out : outc.total_added;
  return EDM_HIP_OK;
}

int edm_hip_gauss_hill_integrals(const edm_hip_gauss *gc, long long n, const double *d_x, int x_stride,
                                 const double *d_h, double h_const, double *d_added) {
  if (n <= 0) return EDM_HIP_OK;
  edm_hip_gauss *g = const_cast<edm_hip_gauss *>(gc);
  HillWorkspace &ws = g->ws;
  const Geom &q = g->g;
  EDM_HIP_TRY(ws.hx.reserve((size_t)n * q.dim));
  EDM_HIP_TRY(ws.ht.reserve((size_t)n * 2 * q.dim));
  EDM_HIP_TRY(ws.hc.reserve((size_t)n * q.dim));
  HillList hl;
  hl.nh = n; hl.x = d_x; hl.x_stride = x_stride; hl.sel = nullptr;
  hl.hx = ws.hx.p; hl.hc = ws.hc.p; hl.ht = ws.ht.p; hl.hx0 = nullptr; hl.nh_dev = nullptr;
  EDM_HIP_TRY(launch_hill_prep(q, hl, g->stream));
  {
    int rcb = ball_list_ensure(g);
    if (rcb) return rcb;
  }
  EDM_HIP_TRY(launch_hill_integrals(q, g->tables(), hl, d_h, h_const, d_added, g->stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return EDM_HIP_OK;
}

int edm_hip_gauss_write(const edm_hip_gauss *g, const char *filename) {
  const Geom &q = g->g;
  std::vector<double> v((size_t)q.total), dv((size_t)q.total * q.dim);
  int rc = edm_hip_gauss_download(g, v.data(), dv.data());
  if (rc) return rc;
  return write_plumed(q, v.data(), dv.data(), filename);
}

}  // extern "C" (the record writer below is internal)

// grid.h:509-674 for one rank on a grid with derivative records: points box_min + k*dx that are in_grid are
// re-sampled by DimmedGrid::get_value_deriv (a batched device lookup in the reference's operation order,
// without the gaussian boundary handling) and written with their derivative columns.
static int multi_write_records(const Geom &q, const double *rec, hipStream_t stream, double *scratch, const char *filename,
                               const double *box_min, const double *box_max, const int *b_periodic, int b_lammps_format) {
  unsigned int counts[3] = {1, 1, 1}, extra_n = 0;
  if (b_lammps_format) extra_n = (unsigned int)(box_min[0] / q.dx[0]);
  size_t total = 1;
  for (int d = 0; d < q.dim; d++) {
    counts[d] = (unsigned int)(int)ceil((box_max[d] - box_min[d]) / q.dx[d]);
    counts[d] = b_periodic[d] ? counts[d] : counts[d] + 1;
    total *= counts[d];
  }
  // sample coordinates and the in_grid filter (host geometry only)
  std::vector<double> xs;
  std::vector<size_t> which;
  std::vector<unsigned int> sup0;
  xs.reserve(total * q.dim);
  for (size_t i = 0; i < total; i++) {
    size_t tmp = i, sup[3] = {0, 0, 0};
    double x[3];
    int d;
    for (d = 0; d < q.dim - 1; d++) {
      sup[d] = tmp % counts[d];
      tmp = (tmp - sup[d]) / counts[d];
      x[d] = sup[d] * q.dx[d] + box_min[d];
    }
    sup[d] = tmp;
    x[d] = sup[d] * q.dx[d] + box_min[d];
    bool in = true;
    for (d = 0; d < q.dim; d++)
      if (!q.periodic[d] && (x[d] < q.min[d] || x[d] >= q.max[d] - q.dx[d])) in = false;
    if (!in) continue;
    which.push_back(i);
    sup0.push_back((unsigned int)sup[0]);
    for (d = 0; d < q.dim; d++) xs.push_back(x[d]);
  }
  const size_t m = which.size();
  std::vector<double> E(m), der(m * q.dim);
  if (m) {
    const Geom plain = plain_geom(q);  // DimmedGrid lookup: no boundary test, no remap
    double *dx = nullptr, *dE = nullptr, *dD = nullptr;
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), sizeof(double) * m * q.dim));
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dE), sizeof(double) * m));
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dD), sizeof(double) * m * q.dim));
    EDM_HIP_TRY(hipMemcpy(dx, xs.data(), sizeof(double) * m * q.dim, hipMemcpyHostToDevice));
    LookupArgs a{};
    a.n = (long long)m; a.x = dx; a.x_stride = q.dim; a.energy = dE; a.f = dD; a.apply_mask = -1;
    EDM_HIP_TRY(launch_lookup(plain, rec, LOOKUP_VALUES, a, scratch, nullptr, stream));
    EDM_HIP_TRY(hipStreamSynchronize(stream));
    EDM_HIP_TRY(hipMemcpy(E.data(), dE, sizeof(double) * m, hipMemcpyDeviceToHost));
    EDM_HIP_TRY(hipMemcpy(der.data(), dD, sizeof(double) * m * q.dim, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dE);
    (void)hipFree(dD);
  }
  FILE *fp = fopen(filename, "w");
  if (!fp) {
    set_error(std::string("cannot open ") + filename);
    return EDM_HIP_ERR_IO;
  }
  if (!b_lammps_format) {
    long long bins[3];
    for (int d = 0; d < q.dim; d++) bins[d] = b_periodic[d] ? (long long)counts[d] : (long long)counts[d] - 1;
    edm::put_header(fp, 1, q.dim, bins, box_min, box_max, b_periodic);
  } else {
    fprintf(fp, "#Auto generated by electronic-dance-music\n\n");
    fprintf(fp, "EDM\n");
    fprintf(fp, "N %u R %g %g\n\n", extra_n + counts[0], q.dx[0], box_max[0]);
    for (size_t i = 1; i < extra_n; i++) fprintf(fp, "%zu %g 0.0 0.0\n", i, i * q.dx[0]);
  }
  for (size_t j = 0; j < m; j++) {
    if (b_lammps_format) fprintf(fp, "%zu ", which[j] + extra_n);
    for (int d = 0; d < q.dim; d++) fprintf(fp, "%.8f ", xs[j * q.dim + d]);
    fprintf(fp, "%.8f ", E[j]);
    for (int d = 0; d < q.dim; d++) fprintf(fp, "%.8f ", -der[j * q.dim + d]);
    fprintf(fp, "\n");
    if (sup0[j] == counts[0] - 1) fprintf(fp, "\n");
  }
  fclose(fp);
  return EDM_HIP_OK;
}

extern "C" {

// DimmedGaussGrid::multi_write / lammps_multi_write (gaussian_grid.h:150-158): the grid's own boundary is the box
int edm_hip_gauss_multi_write(const edm_hip_gauss *g, const char *filename, int b_lammps_format) {
  return edm_hip_gauss_multi_write_box(g, filename, g->g.bmin, g->g.bmax, g->g.bper, b_lammps_format);
}
// DimmedGrid::multi_write on the underlying grid with an explicit box (grid.h:509-674)
int edm_hip_gauss_multi_write_box(const edm_hip_gauss *g, const char *filename, const double *box_min,
                                  const double *box_max, const int *b_periodic, int b_lammps_format) {
  if (b_lammps_format == 1 && g->g.dim > 1) {
    set_error("Lammps format only valid for 1D grids");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  return multi_write_records(g->g, g->rec, g->stream, g->scratch, filename, box_min, box_max, b_periodic, b_lammps_format);
}

// Grid::add (grid.h:275-290) with `other` a plain grid / another gaussian grid
int edm_hip_gauss_add_grid(edm_hip_gauss *g, const edm_hip_grid *other, double scale, double offset) {
  if (!g || !other || g->g.dim != other->g.dim) {
    set_error("Grid::add: dimensions differ");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(other->stream));
  faces_touch(g);
  return grid_add_from(g->g, g->rec, g->stream, g->scratch, plain_geom(other->g), other->values, scale, offset);
}
int edm_hip_gauss_add_gauss(edm_hip_gauss *g, const edm_hip_gauss *other, double scale, double offset) {
  if (!g || !other || g->g.dim != other->g.dim) {
    set_error("Grid::add: dimensions differ");
    return EDM_HIP_ERR_ARG;
  }
  EDM_HIP_TRY(hipStreamSynchronize(other->stream));
  faces_touch(g);
  return grid_add_from(g->g, g->rec, g->stream, g->scratch, other->g, other->rec, scale, offset);
}
// Grid::add (grid.h:275-290) from a PLUMED grid file read with interpolation (read_grid(dim, file, 1)):
// the initial_bias_filename path of EDMBias::subdivide (edm_bias.cpp:166-167, :1066-1072)
int edm_hip_gauss_add_from_file(edm_hip_gauss *g, const char *filename, double scale, double offset) {
  edm_hip_grid *other = nullptr;
  int rc = edm_hip_grid_read(&other, g->g.dim, filename, 1);
  if (rc) return rc;
  if (!other->g.has_deriv) {
    edm_hip_grid_destroy(other);
    set_error("initial bias file has no derivatives (FORCE 0)");
    return EDM_HIP_ERR_IO;
  }
  rc = edm_hip_gauss_add_grid(g, other, scale, offset);
  edm_hip_grid_destroy(other);
  return rc;
}

// DimmedGaussGrid::read (gaussian_grid.h:140-142 -> grid.h:712-835): the node storage and grid geometry come
// from the file; sigma, boundary, tables and the stencil half-widths stay as they are (the reference does not
// touch them either)
int edm_hip_gauss_reread(edm_hip_gauss *g, const char *filename) {
  GridFile gf;
  int rc = read_plumed(g->g.dim, filename, g->g.interp, gf);
  if (rc) return rc;
  if (!gf.g.has_deriv) {
    set_error("a gaussian grid needs the derivative columns (FORCE 1)");
    return EDM_HIP_ERR_IO;
  }
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  Geom q = gf.g;
  for (int d = 0; d < 3; d++) {
    q.sigma[d] = g->g.sigma[d];
    q.bmin[d] = g->g.bmin[d];
    q.bmax[d] = g->g.bmax[d];
    q.bper[d] = g->g.bper[d];
    q.msize[d] = g->g.msize[d];
  }
  faces_touch(g);
  if (q.total != g->g.total) {
    if (g->faces) (void)hipFree(g->faces);
    g->faces = nullptr;
    g->faces_state = 0;
    EDM_HIP_TRY(hipFree(g->rec));
    g->rec = nullptr;
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->rec), sizeof(double) * (size_t)q.total * q.rec));
  }
  g->g = q;
  g->tiles_per_hill = 0;
  int rcn = node_table_rebuild(g);
  if (rcn) return rcn;
  return records_upload(g->g, g->rec, g->stream, gf.values.data(), gf.derivs.data());
}
int edm_hip_gauss_set_lookup_replica(edm_hip_gauss *g, int mode) {
  if (mode < -1 || mode > 1) return EDM_HIP_ERR_ARG;
  g->faces_mode = mode;
  g->faces_unavailable = false;
  if (mode == 0 && g->faces) {
    EDM_HIP_TRY(hipStreamSynchronize(g->stream));
    EDM_HIP_TRY(hipFree(g->faces));
    g->faces = nullptr;
    g->faces_state = 0;
  }
  return EDM_HIP_OK;
}
int edm_hip_gauss_lookup_replica_info(const edm_hip_gauss *g, int *in_use, long long *bytes, long long *rebuilds) {
  if (in_use) *in_use = (g->faces && g->faces_state == 1) ? 1 : 0;
  if (bytes) *bytes = g->faces ? (long long)g->g.total * 128 : 0;
  if (rebuilds) *rebuilds = g->faces_builds;
  return EDM_HIP_OK;
}
int edm_hip_gauss_set_interpolation(edm_hip_gauss *g, int b_interpolate) {
  g->g.interp = b_interpolate ? 1 : 0;
  return EDM_HIP_OK;
}

}  // extern "C"

