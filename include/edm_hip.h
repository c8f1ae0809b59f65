/*
 * edm_hip.h -- C ABI of libedm_hip.so: the MI355X (gfx950) implementation of the
 * EDM per-timestep bias hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no HIP, torch or
 * C++ types.  Each entry point names the reference interface it replaces
 * (file:line relative to the reference tree).  The C++ classes in
 * the include/edm/ headers (EDM::EDMBias, EDM::GaussGrid, EDM::Grid -- source compatible
 * with the reference's lib/ headers) are thin wrappers over these calls, and the
 * LAMMPS fixes call those classes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns an int status (EDM_HIP_OK == 0).  Nothing aborts:
 *     the C++ layer turns a non-zero status into EDM::edm_error() -> abort(),
 *     which is the reference's convention (lib/edm.cpp:4-7).
 *   - "d_" parameters are DEVICE pointers (HBM of the current device); "h_" or
 *     unprefixed pointers are host memory, borrowed for the call only.
 *   - handles own their device memory.  One HIP stream per handle.  A call returns when its RESULTS are on the
 *     host (see "Completion" below): what it queued to update the grid may still be running, and every later
 *     call on the object is ordered behind it.
 *   - positions/forces use the LAMMPS layout: row-major [n][stride] doubles, of
 *     which the first `dim` columns are read/updated.
 *   - there is NO CPU fallback behind any of these calls.
 */
#ifndef EDM_HIP_H_
#define EDM_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDM_HIP_OK 0
#define EDM_HIP_ERR_HIP 1          /* a HIP runtime call failed; see edm_hip_last_error() */
#define EDM_HIP_ERR_ARG 2          /* invalid argument */
#define EDM_HIP_ERR_NO_DEVICE 3    /* no gfx950 device visible */
#define EDM_HIP_ERR_OVERFLOW 4     /* bias overflow buffer full (edm_bias.cpp:501-507) */
#define EDM_HIP_ERR_STATE 5        /* call order violated (e.g. add_hill before pre_add_hill) */
#define EDM_HIP_ERR_IO 6           /* file could not be opened / parsed */
#define EDM_HIP_ERR_COMM 7         /* RCCL failure */

#define EDM_HIP_MAXDIM 3
#define EDM_HIP_BIAS_BUFFER_SIZE 2048   /* edm_bias.h:15 */

typedef struct edm_hip_gauss edm_hip_gauss; /* device-resident DimmedGaussGrid<DIM> */
typedef struct edm_hip_grid edm_hip_grid;   /* device-resident plain DimmedGrid<DIM> (histogram, target) */
typedef struct edm_hip_bias edm_hip_bias;   /* EDMBias controller */

/* Completion: every call returns when its RESULTS (energies, forces already written by an earlier kernel of the
 * call, limiter decisions, per-hill bias, log lines) are final and on the host.  The grid and histogram update of a
 * short hill batch may still be executing on the object's stream at that moment: the library reads its results
 * from host-mapped memory flagged by the device instead of waiting for the stream (EDM_HIP_POLL=0 in the
 * environment: always wait).  Every later call on the same object is ordered behind that work; the streams are
 * blocking streams, so edm_hip_memcpy_* / edm_hip_memset (null stream) and the writers wait for it as well;
 * edm_hip_device_synchronize() waits for everything.  The caller's input arrays are not read after the call.
 *
 * DEVICE output arrays (d_force, d_f, d_fdelta of the *_device / d_* entry points): a forces-only call returns as soon
 * as every workgroup's energy sum has reached the host -- the force kernel itself may not have retired, and its
 * stores are made visible by its end-of-kernel release.  The arrays are therefore ordered for work queued LATER on
 * the handle's own stream and on the null stream (blocking streams: hipMemcpy, edm_hip_memcpy_*, kernels launched on
 * stream 0), which is what every consumer in this repository is.  A consumer on a stream of its own created with
 * hipStreamNonBlocking (Kokkos-style), a peer GPU, or the host reading through a mapping must call
 * edm_hip_bias_wait() / edm_hip_gauss_wait() first: they wait for everything the object has queued.
 * (The reference-order pair step computes its forces on a second stream of the object, beside the hill batch; the
 * object's own stream is made to wait for that pass before the call returns control of it, so the same holds.) */

int edm_hip_gauss_wait(edm_hip_gauss *g);
/* Diagnostic: the completion protocol above (relaxed system-scope stores into host-mapped memory, s_waitcnt(0), a
 * barrier, then the relaxed system-scope flag) run `iterations` times on a region of `words` 8-byte words that all carry
 * the launch's number; the host polls the flag as the library does and then reads the region.  *violations = words
 * found older than their flag (a flag that overtook its data; must be 0), *timeouts = launches whose flag did not
 * arrive within 200 ms. */
int edm_hip_debug_flag_order_stress(int iterations, long long words, long long *violations, long long *timeouts);
int edm_hip_bias_wait(edm_hip_bias *b);

/* ---- runtime ---------------------------------------------------------- */
const char *edm_hip_last_error(void);
const char *edm_hip_version(void);
int edm_hip_device_count(int *count);
int edm_hip_set_device(int device);
/* name, CU count and HBM bytes of the current device */
int edm_hip_device_info(char *name, size_t cap, int *compute_units, size_t *hbm_bytes);
/* device memory helpers so that a C / ctypes / cgo host needs no HIP headers */
int edm_hip_malloc(void **d_ptr, size_t bytes);
int edm_hip_free(void *d_ptr);
/* page-locked host memory: arrays handed to the *_host entry points travel by DMA instead of through the
 * runtime's staging copies */
int edm_hip_host_malloc(void **h_ptr, size_t bytes);
int edm_hip_host_free(void *h_ptr);
int edm_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int edm_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
int edm_hip_memset(void *d_dst, int value, size_t bytes);
int edm_hip_device_synchronize(void);

/* ---- geometry (read-only view of a grid's derived fields) --------------- */
typedef struct {
  int dim;
  int interpolate;                         /* b_interpolate_ */
  int n[EDM_HIP_MAXDIM];                   /* grid_number_ (grid.h:204-207) */
  int periodic[EDM_HIP_MAXDIM];            /* b_periodic_ of the grid */
  double min[EDM_HIP_MAXDIM];
  double max[EDM_HIP_MAXDIM];              /* includes the +dx of non-periodic dims (grid.h:209-210) */
  double dx[EDM_HIP_MAXDIM];               /* re-fitted spacing (grid.h:205) */
  double sigma[EDM_HIP_MAXDIM];            /* sigma * sqrt(2) (gaussian_grid.h:75) */
  int boundary_periodic[EDM_HIP_MAXDIM];   /* b_periodic_boundary_ */
  double boundary_min[EDM_HIP_MAXDIM];
  double boundary_max[EDM_HIP_MAXDIM];
  int minisize[EDM_HIP_MAXDIM];            /* gaussian_grid.h:559-569 */
  long long total;                         /* grid_size_ */
  int derivatives;                         /* b_derivatives_ (always 1 for a gaussian grid) */
} edm_hip_geometry;

/* ---- plain grid: DimmedGrid<DIM> (lib/grid.h) ------------------------------ */
/* make_grid (grid.h:911, grid.cpp:3-20) with b_derivatives = b_interpolate = 0: the CV histogram flavour */
int edm_hip_grid_create(edm_hip_grid **out, int dim, const double *min, const double *max,
                        const double *spacing, const int *periodic);
/* make_grid with any flags (DimmedGrid(min, max, spacing, periodic, b_derivatives, b_interpolate), grid.h:190-213).
 * A grid with derivatives keeps one (V, dV/ds_0 ..) record per node in HBM like the gaussian grid. */
int edm_hip_grid_create_ex(edm_hip_grid **out, int dim, const double *min, const double *max,
                           const double *spacing, const int *periodic, int b_derivatives, int b_interpolate);
/* read_grid (grid.h:923,:928, grid.cpp:22-45) = DimmedGrid(filename, b_interpolate) (grid.h:218-227): PLUMED-1
 * grid file, b_derivatives_ taken from its FORCE line (grid.h:712-835).  read_grid(dim, file) is b_interpolate 1. */
int edm_hip_grid_read(edm_hip_grid **out, int dim, const char *filename, int b_interpolate);
/* Grid::read (grid.h:712-835) on an existing grid: geometry, derivative flag and contents replaced */
int edm_hip_grid_reread(edm_hip_grid *g, const char *filename);
/* Grid::set_interpolation (grid.h:837-839) */
int edm_hip_grid_set_interpolation(edm_hip_grid *g, int b_interpolate);
int edm_hip_grid_destroy(edm_hip_grid *g);
int edm_hip_grid_geometry(const edm_hip_grid *g, edm_hip_geometry *out);
/* Grid::get_grid / grid_deriv_ / clear (grid.h:841, :880, :679); derivs are [total][dim] */
int edm_hip_grid_download(const edm_hip_grid *g, double *h_values);
int edm_hip_grid_download_derivs(const edm_hip_grid *g, double *h_derivs);
int edm_hip_grid_upload(edm_hip_grid *g, const double *h_values);
int edm_hip_grid_upload_derivs(edm_hip_grid *g, const double *h_values, const double *h_derivs);
int edm_hip_grid_clear(edm_hip_grid *g);
/* DimmedGrid::add_value batched (grid.h:370-385): values[bin(x_i)] += w_i, silently
 * ignoring samples outside in_grid.  d_w may be NULL (w = w_const).  An interpolating grid refuses
 * (EDM_HIP_ERR_STATE, grid.h:371-373). */
int edm_hip_grid_add_values(edm_hip_grid *g, long long n, const double *d_x, int x_stride,
                            const double *d_w, double w_const);
/* DimmedGrid::get_value / get_value_deriv batched (grid.h:343-365, :390-446): cubic-Hermite interpolation
 * (interp<DIM>, grid.h:52-139) when the grid interpolates AND stores derivatives, else the nearest-lower node's
 * value and stored derivatives (0 where there are none); (0, 0) outside in_grid.  d_value / d_deriv [n][dim]
 * may be NULL. */
int edm_hip_grid_get_value_deriv(const edm_hip_grid *g, long long n, const double *d_x, int x_stride,
                                 double *d_value, double *d_deriv);
/* Grid::add (grid.h:275-290): this(node) += scale * other(x_node) + offset, derivatives += scale * d other */
int edm_hip_grid_add_grid(edm_hip_grid *g, const edm_hip_grid *other, double scale, double offset);
int edm_hip_grid_add_gauss(edm_hip_grid *g, const edm_hip_gauss *other, double scale, double offset);
/* DimmedGrid::write (grid.h:448-503), byte-identical text */
int edm_hip_grid_write(const edm_hip_grid *g, const char *filename);
/* DimmedGrid::multi_write (grid.h:509-674) for one rank: the CV histogram written by the MPI build's
 * write_histogram (edm_bias.cpp:239) when the grid has no derivatives, re-sampled by interpolation with the
 * derivative columns when it has */
int edm_hip_grid_multi_write(const edm_hip_grid *g, const char *filename, const double *box_min,
                             const double *box_max, const int *b_periodic, int b_lammps_format);

/* ---- gaussian grid: DimmedGaussGrid<DIM> (lib/gaussian_grid.h) ----------- */
/* make_gauss_grid (gaussian_grid.h:636, gaussian_grid.cpp:3-18) */
int edm_hip_gauss_create(edm_hip_gauss **out, int dim, const double *min, const double *max,
                         const double *spacing, const int *periodic, int b_interpolate,
                         const double *sigma);
/* read_gauss_grid (gaussian_grid.h:647, gaussian_grid.cpp:23-33) = DimmedGaussGrid(filename, sigma)
 * (gaussian_grid.h:85-93); the file must carry derivative columns (FORCE 1) */
int edm_hip_gauss_read(edm_hip_gauss **out, int dim, const char *filename, const double *sigma);
/* GaussGrid::read (gaussian_grid.h:140-142): node storage and grid geometry replaced from the file, sigma /
 * boundary / tables kept */
int edm_hip_gauss_reread(edm_hip_gauss *g, const char *filename);
/* Lookup replica (no reference counterpart; a memory-for-bandwidth trade sized for 288 GB of HBM): a 2-D / 3-D
 * grid whose boundary is periodic in every dimension (the coordinate CV of fix edm in a periodic box) keeps, next
 * to its node records, one aligned 128-byte block per node with the records of nodes (i0, i1), (i0+1, i1),
 * (i0, i1+1), (i0+1, i1+1) [at i2] -- 4x the grid's bytes (17 GB for 512^3) -- so that the interpolation of a sample
 * (grid.h:390-446) reads 1 (2-D) or 2 (3-D) lines instead of 2.5 / 5.  Results are bit-identical to lookups on
 * the node records.  The in-place hill gather keeps the replica current; any other write of the grid marks it
 * stale and the next lookup rebuilds it.  mode: -1 automatic (default: grids of 32 MB and more, memory
 * permitting), 0 off (and freed), 1 always. */
int edm_hip_gauss_set_lookup_replica(edm_hip_gauss *g, int mode);
int edm_hip_gauss_lookup_replica_info(const edm_hip_gauss *g, int *in_use, long long *bytes, long long *rebuilds);
/* GaussGrid::set_interpolation (gaussian_grid.h:168-170) */
int edm_hip_gauss_set_interpolation(edm_hip_gauss *g, int b_interpolate);
int edm_hip_gauss_destroy(edm_hip_gauss *g);
/* GaussGrid::set_boundary (gaussian_grid.h:378-435): rebuilds the two 65536-entry
 * McGovern-De Pablo tables per non-periodic dimension on the host with libm erf
 * (bit-identical to the reference) and uploads them. */
int edm_hip_gauss_set_boundary(edm_hip_gauss *g, const double *min, const double *max,
                               const int *periodic);
int edm_hip_gauss_geometry(const edm_hip_gauss *g, edm_hip_geometry *out);
/* reads the two McGovern-De Pablo tables of dimension dim_index back from HBM (bc_denom_table_ /
 * bc_denom_deriv_table_, gaussian_grid.h:394-431): EDM_HIP_BC_TABLE_SIZE doubles each, either pointer may be
 * NULL.  A periodic boundary dimension has no tables: EDM_HIP_ERR_ARG. */
#define EDM_HIP_BC_TABLE_SIZE 65536     /* gaussian_grid.h:11 */
int edm_hip_gauss_download_tables(const edm_hip_gauss *g, int dim_index, double *h_denom, double *h_denom_deriv);
/* Grid::get_grid + grid_deriv_ (grid.h:879-880): host layout values[total],
 * derivs[total][dim]; the device keeps one (V, dV/ds_0..) record per node. */
int edm_hip_gauss_download(const edm_hip_gauss *g, double *h_values, double *h_derivs);
int edm_hip_gauss_upload(edm_hip_gauss *g, const double *h_values, const double *h_derivs);
int edm_hip_gauss_clear(edm_hip_gauss *g);
/* the raw device record array ((1+dim) doubles per node padded to 2 or 4) for a caller's own collectives:
 * pointer, doubles per node, node count.  The call waits for the updates queued on the handle's stream, and --
 * since the caller may now write the records behind the library's back -- switches the lookup replica of this
 * grid off for good. */
int edm_hip_gauss_device_buffer(edm_hip_gauss *g, double **d_records, int *doubles_per_node,
                                long long *nodes);

/* GaussGrid::get_value_deriv batched (gaussian_grid.h:118-138 -> grid.h:390-446 ->
 * interp<DIM> grid.h:52-139).  Per sample: d_energy[i] = V (may be NULL),
 * d_deriv[i*dim + j] = dV/ds_j (positive gradient, may be NULL). */
int edm_hip_gauss_get_value_deriv(const edm_hip_gauss *g, long long n, const double *d_x,
                                  int x_stride, double *d_energy, double *d_deriv);
/* flat node index the lookup of each sample starts from (grid.h:264-273,:315-325) after
 * bounds/remap handling, or -1 where the reference returns 0: the integer half of parity */
int edm_hip_gauss_sample_index(const edm_hip_gauss *g, long long n, const double *d_x,
                               int x_stride, long long *d_flat);

/* DimmedGaussGrid::remap batched (gaussian_grid.h:504-541): the image of each sample the lookup and hill kernels
 * work with (periodic grid: wrapped into the grid; non-periodic grid inside a periodic boundary: shifted by the
 * boundary period that lands nearest the grid's minimum or maximum).  d_out rows of dim doubles. */
int edm_hip_gauss_remap(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride, double *d_out);

/* EDMBias::update_forces (edm_bias.cpp:276-295): for every sample with
 * (apply_mask < 0 || d_mask[i] & apply_mask): E += V(x_i); f[i][j] -= dV/ds_j.
 * d_mask may be NULL when apply_mask < 0.  *energy = sum of V (host double). */
int edm_hip_gauss_update_forces(const edm_hip_gauss *g, long long n, const double *d_x,
                                int x_stride, double *d_f, int f_stride, const int *d_mask,
                                int apply_mask, double *energy);
/* the fix_edm_pair inner loop (fix_edm_pair.cpp:215-217) batched over a 1-D
 * distance array: d_force[i] = -dV/dr(r_i) (i.e. edm_force[0] after
 * update_force on a zeroed accumulator); *energy = sum V(r_i). */
int edm_hip_gauss_pair_forces(const edm_hip_gauss *g, long long n, const double *d_r,
                              double *d_force, double *energy);

/* measurement support (no reference counterpart): stamps the dominant lookup kernel of
 * update_forces / pair_forces with HIP events on the handle's own stream (the dispatch's own begin/end
 * timestamps) and sums hipEventElapsedTime over the stamped launches.  enabled = N > 0 stamps every
 * N-th launch (a stamp costs the stream a few microseconds); the times are summed by profile_read,
 * never inside the caller's timed loop. */
int edm_hip_gauss_profile_enable(edm_hip_gauss *g, int enabled);
int edm_hip_gauss_profile_read(edm_hip_gauss *g, double *kernel_ms_total, long long *launches, int reset);

/* GaussGrid::add_value batched (gaussian_grid.h:176-372).  Applies n hills
 * (positions d_x, heights d_h or h_const when d_h == NULL) IN LIST ORDER and
 * writes each hill's integrated bias to d_added (may be NULL).  No limiting.
 * *total_added (host, may be NULL) = sum of d_added. */
int edm_hip_gauss_add_values(edm_hip_gauss *g, long long n, const double *d_x, int x_stride,
                             const double *d_h, double h_const, double *d_added,
                             double *total_added);
/* per-hill integrated bias WITHOUT touching the grid (the return value of
 * add_value, which does not depend on grid contents) */
int edm_hip_gauss_hill_integrals(const edm_hip_gauss *g, long long n, const double *d_x,
                                 int x_stride, const double *d_h, double h_const,
                                 double *d_added);
/* DimmedGrid::write / multi_write on the underlying grid (grid.h:448-503, :509-674;
 * multi_write evaluated for one rank, lammps != 0 selects the LAMMPS table format) */
int edm_hip_gauss_write(const edm_hip_gauss *g, const char *filename);
int edm_hip_gauss_multi_write(const edm_hip_gauss *g, const char *filename, int b_lammps_format);
/* GaussGrid::multi_write(filename, box_low, box_high, b_periodic, fmt) (gaussian_grid.h:160-166) */
int edm_hip_gauss_multi_write_box(const edm_hip_gauss *g, const char *filename, const double *box_min,
                                  const double *box_max, const int *b_periodic, int b_lammps_format);
/* Grid::add (grid.h:275-290) with another device grid as `other` (evaluated through ITS get_value_deriv) */
int edm_hip_gauss_add_grid(edm_hip_gauss *g, const edm_hip_grid *other, double scale, double offset);
int edm_hip_gauss_add_gauss(edm_hip_gauss *g, const edm_hip_gauss *other, double scale, double offset);
/* Grid::add (grid.h:275-290) from a PLUMED grid file read with interpolation:
 * the initial_bias_filename path of EDMBias::subdivide (edm_bias.cpp:166-167) */
int edm_hip_gauss_add_from_file(edm_hip_gauss *g, const char *filename, double scale, double offset);

/* ---- EDMBias controller (lib/edm_bias.h) ---------------------------------- */
/* EDMBias::EDMBias(const std::string&) + read_input (edm_bias.cpp:34-69, :986-1095) */
int edm_hip_bias_create(edm_hip_bias **out, const char *input_filename);
int edm_hip_bias_destroy(edm_hip_bias *b);
/* EDMBias::setup (edm_bias.cpp:264-269) */
int edm_hip_bias_setup(edm_hip_bias *b, double temperature, double boltzmann_constant);
/* EDMBias::subdivide (edm_bias.cpp:98-222); arrays hold dim entries */
int edm_hip_bias_subdivide(edm_hip_bias *b, const double *sublo, const double *subhi,
                           const double *boxlo, const double *boxhi, const int *b_periodic,
                           const double *skin);
/* EDMBias::set_mask (edm_bias.cpp:982-984); device pointer, borrowed until replaced */
int edm_hip_bias_set_mask(edm_hip_bias *b, const int *d_mask);
/* EDMBias::update_forces (edm_bias.cpp:276-295) */
int edm_hip_bias_update_forces(edm_hip_bias *b, long long n, const double *d_x, int x_stride,
                               double *d_f, int f_stride, int apply_mask, double *energy);
/* batched EDMBias::update_force over pair distances (edm_bias.cpp:297-311) */
int edm_hip_bias_pair_forces(edm_hip_bias *b, long long n, const double *d_r, double *d_force,
                             double *energy);
/* EDMBias::add_hills (edm_bias.cpp:401-411): pre_add_hill(n); add_hill for every
 * masked sample with uniform d_runiform[i]; post_add_hill.  est_hill_count < 0
 * means "use n" (add_hills semantics); fix_edm_pair passes its own estimate. */
int edm_hip_bias_add_hills(edm_hip_bias *b, long long n, const double *d_x, int x_stride,
                           const double *d_runiform, int apply_mask, long long est_hill_count);
/* One hill-depositing step of fix edm (fix_edm.cpp:131-152) in a single call: EDMBias::update_forces
 * (edm_bias.cpp:276-295) followed by EDMBias::add_hills (:401-411) over the same samples.  Same results as
 * edm_hip_bias_update_forces + edm_hip_bias_add_hills; queued back to back, the host waits once. */
int edm_hip_bias_step(edm_hip_bias *b, long long n, const double *d_x, int x_stride, double *d_f,
                      int f_stride, const double *d_runiform, int apply_mask, long long est_hill_count,
                      double *energy);
/* fix edm's post_force for a caller whose atom arrays are in HOST memory (lammps/fix_edm.cpp:134-162): h_x rows
 * [n][x_stride] go up (the block is page-locked in place the first time it is seen), update_forces -- and, when
 * hill_step != 0, add_hills over the same samples -- run on the device, and the bias force comes back as a delta the
 * library adds to h_f rows [n][f_stride] on the host: the caller's force array is never uploaded (edm_bias.cpp:287-293
 * only ever subtracts dV/ds from it).  h_mask (int[n]) may be NULL when apply_mask < 0, h_runiform NULL without
 * hill_density or with device uniforms.  Per atom 8 * x_stride B (+ 4 B mask, + 8 B uniform on hill steps) travel up
 * and 8 * dim B down.  Same results as edm_hip_bias_step / edm_hip_bias_update_forces on device arrays.
 * The position, uniform and mask blocks are page-locked in place the first time they are seen (and again when a
 * pointer or size changes); the delta starts its way down behind the launch that carries the force kernel, beside the
 * step's hills, and is added into h_f by edm_hip_bias_set("host_add_threads", k) threads (default 4, the caller's
 * included; 1 = no helper threads). */
int edm_hip_bias_step_host(edm_hip_bias *b, long long n, const double *h_x, int x_stride, double *h_f, int f_stride,
                           const int *h_mask, const double *h_runiform, int apply_mask, int hill_step,
                           long long est_hill_count, double *energy);
/* fix edm_pair on a device-resident neighbour list.
 * edm_hip_bias_pair_list_upload: called when LAMMPS has rebuilt the list, with HOST arrays -- the half list
 * flattened in neighbour-list order (pair_i/pair_j, j masked with NEIGHMASK) and the atom types [nall]; the library
 * keeps them, plus per-atom index lists of the entries, on the device.
 * edm_hip_bias_pair_list_step: this step's positions d_x [nall][3] in, the bias force per atom out in d_fdelta
 * [nall][3] (i always, j iff j < nlocal: newton off; ghost atoms zero), energy returned.  Pair distances, lookups
 * and the per-atom force sums (fixed order, no atomics: bit-reproducible) run on the GPU.  On hill steps
 * (hill_step != 0) pre_add_hill(est_hill_count) precedes the forces and every list entry passing the type filter
 * deposits add_hill(r, u) once, and a second time iff j is owned (fix_edm_pair.cpp:230-237), in list order, with
 * device uniforms (edm_hip_bias_set_device_rng; sample index 2 * entry + slot).  *ncalls returns the number of
 * add_hill calls of the step, the next hill step's est_hill_count (fix_edm_pair.cpp:245).  Per atom this moves
 * 24 B of positions in and 24 B of forces out instead of 16 B per PAIR. */
int edm_hip_bias_pair_list_upload(edm_hip_bias *b, long long npairs, const int *h_pair_i, const int *h_pair_j,
                                  long long nall, const int *h_type);
int edm_hip_bias_pair_list_step(edm_hip_bias *b, int nlocal, int itype, int jtype, const double *d_x,
                                double *d_fdelta, int hill_step, long long est_hill_count, double *energy,
                                long long *ncalls);
/* One hill-depositing step of fix edm_pair (fix_edm_pair.cpp:174-246) in a single call:
 * pre_add_hill(est_hill_count) (flushes the overflow buffer), the force evaluation of
 * edm_hip_bias_pair_forces(n, d_r, d_force), add_hill(d_sample_r[i], d_runiform[i]) for the n_samples
 * staged samples, post_add_hill.  Same results as the separate calls in that order; forces and hills
 * are queued back to back and the host waits once. */
int edm_hip_bias_pair_step(edm_hip_bias *b, long long n, const double *d_r, double *d_force,
                           long long n_samples, const double *d_sample_r, const double *d_runiform,
                           long long est_hill_count, double *energy);
/* edm_hip_bias_pair_step for a caller whose arrays are in HOST memory (the host-list fix edm_pair,
 * fix_edm_pair.cpp:139-256): same results; the library stages them through HBM with the copies queued around the
 * kernels -- distances up, force kernel, then the forces come down while the hill samples and uniforms go up
 * (two streams, both directions of the link busy), then the hill cycle.  h_sample_r may be h_r itself. */
int edm_hip_bias_pair_step_host(edm_hip_bias *b, long long n, const double *h_r, double *h_force,
                                long long n_samples, const double *h_sample_r, const double *h_runiform,
                                long long est_hill_count, double *energy);
/* One hill-depositing step of fix edm_pair in the REFERENCE'S OWN ORDER (fix_edm_pair.cpp:173-247): the reference
 * walks the neighbour list once -- update_force for pair k (:217), then that pair's one or two add_hill calls
 * (:230-237) -- so the force of pair k is read from a bias that already holds the hills deposited for pairs 0..k-1 of
 * the same step.  This entry reproduces exactly that: pre_add_hill(est_hill_count); the staged samples are applied as
 * in edm_hip_bias_pair_step (which hills are accepted, the limiter and the grid do not depend on the forces, so grid,
 * histogram, HILLS log and limiter state are those of edm_hip_bias_pair_step bit for bit); d_force[k] and the energy
 * are interpolated on the bias as it stood when the reference's loop reached pair k.
 * d_first_sample[k] (int, n entries, ascending) = index into the sample arrays of pair k's first add_hill call =
 * the number of add_hill calls issued before pair k's update_force.  With a communicator a rank's pairs see the hills
 * of THAT RANK'S earlier add_hill calls -- the reference's ranks replay each other's hills in post_add_hill only
 * (edm_bias.cpp:565-583, :630-706).  The force pass keeps, per 32-node tile of the grid, the
 * tile's records behind every hill that reached it (~3 MB for the ~125 hills of a 1 M-pair step): at most 16 384 hills
 * per step (beyond: EDM_HIP_ERR_ARG -- all-samples deposition of a large system keeps edm_hip_bias_pair_step).  edm_hip_bias_pair_step evaluates every force of the step on the bias as it stands after
 * pre_add_hill instead: faster (the forces share the selection's launch), and on a hill step its forces differ from
 * the reference's by the bias the step itself deposits (INTEGRATION.md has the measured size). */
int edm_hip_bias_pair_step_ordered(edm_hip_bias *b, long long n, const double *d_r, double *d_force,
                                   const int *d_first_sample, long long n_samples, const double *d_sample_r,
                                   const double *d_runiform, long long est_hill_count, double *energy);
int edm_hip_bias_pair_step_ordered_host(edm_hip_bias *b, long long n, const double *h_r, double *h_force,
                                        const int *h_first_sample, long long n_samples, const double *h_sample_r,
                                        const double *h_runiform, long long est_hill_count, double *energy);
/* EDMBias::pre_add_hill / add_hill / post_add_hill (edm_bias.cpp:413-442, :528-563,
 * :565-583).  add_hill stages the sample (host values); the staged batch is
 * applied on the device, in call order, at post_add_hill. */
int edm_hip_bias_pre_add_hill(edm_hip_bias *b, long long est_hill_count);
int edm_hip_bias_add_hill(edm_hip_bias *b, const double *position, double runiform);
int edm_hip_bias_post_add_hill(edm_hip_bias *b);
/* EDMBias::write_bias / write_lammps_table / write_histogram / clear_histogram
 * (edm_bias.cpp:224-262).  serial_format != 0 reproduces the EDM_SERIAL build
 * (all three fall back to DimmedGrid::write), 0 the MPI build's multi_write. */
int edm_hip_bias_write_bias(const edm_hip_bias *b, const char *filename, int serial_format);
int edm_hip_bias_write_lammps_table(const edm_hip_bias *b, const char *filename, int serial_format);
int edm_hip_bias_write_histogram(const edm_hip_bias *b, int serial_format);
int edm_hip_bias_clear_histogram(edm_hip_bias *b);
/* the owned grids (bias_ and cv_hist_); NULL before subdivide */
edm_hip_gauss *edm_hip_bias_gauss(edm_hip_bias *b);
edm_hip_grid *edm_hip_bias_histogram(edm_hip_bias *b);
/* public data members of EDMBias by name (edm_bias.h:118-157 and the private limiter
 * state): dim, b_tempering, b_targeting, global_tempering, bias_factor, boltzmann_factor,
 * temperature, hill_prefactor, bias_per_step, hill_density, cum_bias, total_volume,
 * expected_target, b_outofbounds, overflow_left, overflow_right, b_skip_hill_add,
 * hills_added, steps, mpi_rank, mpi_size.  Unknown name -> EDM_HIP_ERR_ARG. */
int edm_hip_bias_get(const edm_hip_bias *b, const char *name, double *value);
int edm_hip_bias_set(edm_hip_bias *b, const char *name, double value);
/* bias_dx, bias_sigma, min, max (dim doubles) */
int edm_hip_bias_get_array(const edm_hip_bias *b, const char *name, double *out);
/* Fast mode of the fixes' random numbers (no reference counterpart): while enabled, an add_hill
 * cycle that is given NO uniform array (d_runiform == NULL with hill_density set) draws
 * u_i = SplitMix64(seed + cycle * 0x632BE59BD9B4E019, output i + 1) >> 11 * 2^-53 on the device for sample i
 * of add_hill cycle number `cycle` (0, 1, ... since this call).  Nothing is generated on the host, uploaded or
 * read from HBM.  The parity mode (the host's RanMars numbers passed in) stays the default. */
int edm_hip_bias_set_device_rng(edm_hip_bias *b, int enabled, unsigned long long seed);
/* 1 (default): write the per-rank HILLS log like the reference (edm_bias.cpp:586-599);
 * 0: skip the text log (the CV histogram is still updated). */
int edm_hip_bias_set_hill_log(edm_hip_bias *b, int enabled);

/* ---- multi-GPU: one rank per GPU, RCCL over xGMI ---------------------------- */
/* replaces the MPI hill exchange of EDMBias::flush_buffers / update_height
 * (edm_bias.cpp:630-706, :922-931).  id_bytes is an ncclUniqueId (128 bytes)
 * created by rank 0 and distributed by the host program (MPI_Bcast in LAMMPS,
 * torch.distributed in bench.py). */
int edm_hip_comm_unique_id(void *id_bytes, size_t cap);
int edm_hip_bias_comm_init(edm_hip_bias *b, const void *id_bytes, int nranks, int rank);
/* The same exchange protocol with the payloads staged through POSIX shared memory instead of RCCL (device -> host
 * slot, barrier, host -> device): ranks of one host that cannot form an RCCL communicator -- several ranks on ONE
 * GPU in the tests -- or debugging.  shm_name ("/name", unique to the job) is the same on every rank; the call
 * returns when all nranks ranks have attached.  A rank that waits more than two minutes at a barrier gets
 * EDM_HIP_ERR_COMM instead of hanging. */
int edm_hip_bias_comm_init_shm(edm_hip_bias *b, const char *shm_name, int nranks, int rank);
int edm_hip_bias_comm_destroy(edm_hip_bias *b);

#ifdef __cplusplus
}
#endif
#endif /* EDM_HIP_H_ */
