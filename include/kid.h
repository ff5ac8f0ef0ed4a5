/* kid.h -- C ABI of libkid_hip.so: the MI355X (gfx950) implementation of the NOAA-GFDL/icebergs evolve loop.
 *
 * Drop-in boundary.  Each entry point replaces one `subroutine f(bergs)` call site
 * inside icebergs_run() of /root/reference/src/icebergs.F90; the Fortran host keeps icebergs_init() /
 * icebergs_run() and its derived types, flattens the per-cell linked lists into the structure of arrays of
 * kid_types.h in reference traversal order, and calls these through the ISO_C_BINDING module
 * icebergs_amd/fortran/kid_hip_binding.F90 (see INTEGRATION.md for the glue a maintainer adds).
 *
 *   reference call site (icebergs.F90)                     entry point
 *   -----------------------------------------------------  --------------------------------------------
 *   ice_bergs_framework_init, grid copy   FW:1021-1094     kid_create + kid_set_static_grid
 *   forcing ingest result (grd%uo ... )   IB:5236-5383     kid_set_forcing
 *   the forcing ingest block itself       IB:5236-5383     kid_ingest_forcing (+ kid_get_forcing for send_data)
 *   calving block + accumulate_calving + calve_icebergs  IB:5203-5231, 5388, 5403   kid_calving
 *   write_restart_bergs / read_restart_bergs  IO2:124-631, 663-1049   kid_write_restart / kid_read_restart
 *   accumulator zeroing                   IB:5125-5156     kid_zero_accumulators
 *   interp_gridded_fields_to_bergs        IB:5423, 5473    kid_interp_gridded_fields_to_bergs
 *   evolve_icebergs                       IB:5433          kid_evolve_icebergs
 *   move_berg_between_cells               IB:5437          kid_move_berg_between_cells
 *   footloose_calving                     IB:5453          kid_footloose_calving
 *   thermodynamics                        IB:5505          kid_thermodynamics
 *   create_gridded_icebergs_fields        IB:5512          kid_create_gridded_icebergs_fields
 *   (all of the above, one coupling step)                  kid_run_step / kid_step_local + kid_step_gather
 *
 * Conventions: every function returns 0 on success and a negative KID_E* code on failure (the Fortran shim
 * turns non-zero into error_mesg(...,FATAL), the reference's abort-on-error convention, e.g. IB:3207).
 * Host arrays are caller-owned, column-major fp64 / int32 as Fortran owns them; the library owns device
 * memory only.  One handle <-> one HIP device and one stream; handles are not thread-safe (the reference is
 * single-threaded per rank and non-reentrant, IB:5110).  There is no CPU fallback: if no gfx950 device can be
 * used, kid_create fails.
 */
#ifndef KID_H
#define KID_H
#include "kid_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kid_handle kid_handle;

enum {
  KID_OK = 0,
  KID_EINVAL = -1,     /* bad argument / shape mismatch                         */
  KID_EHIP = -2,       /* a HIP runtime call failed (see kid_last_error)        */
  KID_ENODEV = -3,     /* no usable GPU                                         */
  KID_ECAPACITY = -4,  /* more bergs than the handle's capacity                 */
  KID_EUNSUPPORTED = -5 /* a switch combination this build does not implement   */
};

/* ---- lifetime ---- */
int kid_create(const kid_grid_desc *grid, const kid_params *params, int64_t capacity, int device,
               kid_handle **out);
int kid_destroy(kid_handle *h);
int kid_set_params(kid_handle *h, const kid_params *params);
/* Launch on an existing hipStream_t (e.g. PyTorch's current stream).  NULL names the device's default (null) stream;
 * a handle that is never given a stream uses a private non-blocking one. */
int kid_set_stream(kid_handle *h, void *hip_stream);
/* Pipelined launches: with `enable`, a fused per-berg launch goes through the hot build in two halves on the main stream
 * while the general build (cell hops, bounces; ~85 us of single-wave latency) of each half runs on `side_stream`, under
 * the hot build of the other half or of the next step.  Every entry point that reads berg state or accumulators on the
 * main stream first orders itself behind the side stream; a gather launched on the side stream itself is ordered by
 * that stream (icebergs_amd/distributed.py PipelinedStepper).  Results do not depend on the setting.
 * enable = 2, the "slow lane" schedule: the hot build stays one launch; a berg it hands over at step s is stepped by
 * general-build launches on `side_stream` for steps s and s+1 (under the hot builds of s+1 and s+2) and returns to the
 * hot build at s+2, so the general build's latency never sits between two hot builds.  The caller alternates two
 * accumulator blocks (kid_bind_accum_buffer) and launches the gather on the side stream; the per-cell forcing records
 * are double-buffered inside.  Applies to the fused RK4/Verlet step (either interpolation order) without footloose, bonds
 * or interactions and falls back to the plain schedule otherwise. */
int kid_set_side_stream(kid_handle *h, void *side_stream, int enable);
int kid_sync(kid_handle *h);
const char *kid_last_error(const kid_handle *h);
const char *kid_version(void);
int64_t kid_sizeof(int which); /* 0 kid_params, 1 kid_grid_desc, 2 kid_berg_soa, 3 kid_bond_soa, 4 kid_forcing_in, 5 kid_calving_params, 6 kid_calving_in: ABI layout check */

/* ---- grid and forcing (host pointers; KID_G_* / KID_F_* order) ---- */
int kid_set_static_grid(kid_handle *h, const double *const fields[KID_NGRID_STATIC]);
int kid_set_forcing(kid_handle *h, const double *const fields[KID_NFORCING]);
/* Same, but the planes already live in device memory (e.g. an ocean model on the same GPU): device-to-device
 * copy + the per-cell prepass, asynchronous on the handle's stream. */
int kid_set_forcing_device(kid_handle *h, const double *const dev_fields[KID_NFORCING]);

/* ---- berg population ---- */
int kid_upload_bergs(kid_handle *h, const kid_berg_soa *host);        /* sets the population (n <= capacity) */
int kid_download_bergs(kid_handle *h, kid_berg_soa *host);            /* host->n must be >= kid_num_bergs slots */
int kid_num_bergs(kid_handle *h, int64_t *n_slots, int64_t *n_alive); /* slots include dead bergs until compaction */
int kid_compact_bergs(kid_handle *h);                                 /* drop melted / departed bergs, keep order */
/* move_berg_between_cells (IB:5437): device counting sort of the SoA by cell (j-major), dead bergs dropped.  Results
 * never depend on it, speed does.  kid_run_step calls it every `steps` steps (default 16; 0 = never). */
int kid_move_berg_between_cells(kid_handle *h);
int kid_set_resort_interval(kid_handle *h, int steps);

/* berg%uo..od (the last interpolated environment).  With old_interp_flds_order accel re-interpolates (IB:2035) and the
 * stored copy is read only by trajectory records (FW:5422) and bergs_chksum (FW:7040): a run with ignore_traj=T may
 * switch it off (on = 0) and save 13 stores per berg per step.  With .not.old_interp_flds_order the stored environment is
 * an input of evolve_icebergs and thermodynamics called one after the other, but not of the fused step, which interpolates
 * it for itself: switched off there, kid_run_step / kid_step_local still run, and the entry points that read it
 * (kid_evolve_icebergs, kid_thermodynamics) return KID_EINVAL.  Default on. */
int kid_set_store_environment(kid_handle *h, int on);

/* Forcing ingest on the device: the block of icebergs_run that builds grd%uo .. grd%hi from the
 * coupler's arguments, IB:5236-5383 + invert_tau_for_du IB:8272-8296: B/C-grid velocities, B/C/A-grid wind stress,
 * stress -> velocity difference, the Kelvin test on sst, sss = -1 when absent, land / NaN scrub.  One rank owns the
 * whole domain, so mpp_update_domains reduces to the zonal wrap of the halo columns (in->cyclic_x) or to nothing.
 * Replaces the host-side ingest + kid_set_forcing.  Host arrays are staged and the call returns when they may be
 * reused; with in->on_device nothing is copied and the call is asynchronous on the handle's stream.
 * add_iceberg_thickness_to_SSH (IB:5330-5337) stays with the host. */
int kid_ingest_forcing(kid_handle *h, const kid_forcing_in *in);
/* grd%uo .. grd%hi as the handle holds them (KID_F_* order, data-domain planes, NULL entries skipped): what the
 * reference hands to send_data for id_uo, id_vo, ... (IB:5529-5548).  Needs kid_set_forcing or kid_ingest_forcing. */
int kid_get_forcing(kid_handle *h, double *const fields[KID_NFORCING]);

/* ---- calving source ----
 * kid_calving is, in this order, the calving block of icebergs_run (IB:5203-5231: mask, running mean IB:5999-6038, kg/s),
 * accumulate_calving (IB:6153-6222) and calve_icebergs (IB:6225-6402): the buckets grd%stored_ice(:,:,1:10) /
 * grd%stored_heat live on the device, and bergs calved from overflowing buckets are appended to the resident SoA
 * (ids from the per-cell counter, kid_set_iceberg_counter).  bergs%current_year / current_yearday come from kid_params.
 * scalars[KID_NCALV_SCALARS]: what the call adds to the budget scalars of type icebergs (KID_CS_* order; may be NULL).
 * Returns KID_ECAPACITY when the SoA cannot hold the new bergs.  Replaces IB:5203-5231, 5388, 5397, 5403. */
int kid_set_calving_params(kid_handle *h, const kid_calving_params *cp);
/* read_restart_calving: stored_ice (10 data-domain planes), stored_heat, the two running means; NULL keeps the handle's */
int kid_set_calving_state(kid_handle *h, const double *stored_ice, const double *stored_heat, const double *rmean_calving,
                          const double *rmean_calving_hflx);
/* ... and for write_restart_calving / send_data (id_stored_ice, id_real_calving IB:5593-5600); NULL entries are skipped */
int kid_get_calving_state(kid_handle *h, double *stored_ice, double *stored_heat, double *rmean_calving,
                          double *rmean_calving_hflx, double *real_calving);
int kid_calving(kid_handle *h, const kid_calving_in *in, double *scalars);
/* grd%calving (kg/s, the unused remainder: id_unused IB:5396, returned to the coupler at IB:5656) and grd%calving_hflx
 * after kid_calving; data-domain planes, NULL skipped.  The melt the step adds to grd%calving_hflx (IB:3129) is the
 * accumulator KID_A_CALVING_HFLX. */
int kid_get_calving(kid_handle *h, double *calving, double *calving_hflx);

/* ---- restart files straight from the structure of arrays ----
 * icebergs.res.nc as write_restart_bergs writes it (icebergs_fms2io.F90:124-420: one unlimited dimension "i", the same
 * variable names, order, types and long_name / units attributes, ids split into id_cnt / id_ij), bonds_iceberg.res.nc
 * for bonded populations (IO2:466-583) and calving.res.nc
 * (IO2:583-631: stored_ice, stored_heat, iceberg_counter_grd, the running means).  netCDF classic (CDF-2) written and
 * read directly: neither FMS nor libnetcdf is needed.
 * The first three work on host arrays and need no device: */
int kid_restart_write_bergs(const char *path, const kid_params *p, const kid_berg_soa *host);
int kid_restart_count_bergs(const char *path, int64_t *n);
/* fills the arrays of `host` (room for `capacity` rows): the file's fields, zeros elsewhere, *_old = current values and
 * halo_berg = 0 as read_restart_bergs sets them (IO2:895-925); xi / yj need the grid and are left to kid_read_restart,
 * and so are the ids of a file with the old 32-bit iceberg_num (they come back 0 here; generate_id needs the counters) */
int kid_restart_read_bergs(const char *path, kid_berg_soa *host, int64_t capacity);
/* bonds_iceberg.res.nc (IO2:466-583, read_restart_bonds IO2:1190-1481): one record per bond side; `bergs` gives the ids
 * and cells of both ends.  Reading puts every bond at the head of its berg's list, as form_a_bond does. */
int kid_restart_write_bonds(const char *path, const kid_params *p, const kid_berg_soa *bergs, const kid_bond_soa *bonds);
int kid_restart_read_bonds(const char *path, const kid_berg_soa *bergs, kid_bond_soa *bonds);
/* the resident state to <dir>/icebergs.res.nc (+ bonds_iceberg.res.nc with bonds, + calving.res.nc when the calving
 * source is on), and back:
 * bergs outside the computational domain or in cells of zero area are dropped (IO2:880-884, 948-955), xi / yj are
 * recomputed on the device from (lon, lat, ine, jne) (IO2:945) */
int kid_write_restart(kid_handle *h, const char *dir);
int kid_read_restart(kid_handle *h, const char *dir);
/* bergs_chksum (FW:6889-6987: the "write_restart berg chksum=.. chksum2=.. chksum3=.. chksum4=.. chksum5=.. #=.." line of
 * icebergs_save_restart, IB:8145, which the reference's regression tests record) on the resident population:
 * out[0..4] = chksum, chksum2, chksum3, chksum4, chksum5 as the default integers the reference prints, out[5] = #, the number
 * of bergs on the computational domain.  FMS's mpp_chksum is restated (wrap-around sum of bit patterns); see
 * csrc/kid_chksum.inc for the quirks that are reproduced on purpose. */
int kid_bergs_chksum(kid_handle *h, int64_t out[6]);

/* ---- trajectories: record_posn (FW:5328-5498) as a device pass that appends one record per selected berg
 * to buffers in HBM, and iceberg_trajectories.nc (icebergs_fms2io.F90:1631-2103) written from them: dimension "i", lon,
 * lat, year, day, id_cnt, id_ij, then the save_fl_traj group, then the long group unless save_short_traj (same names,
 * types and attributes).  kid_write_trajectories appends to an existing file, like the reference, and empties the buffers.
 * With kid_traj_params.save_bond_traj every bond of a sampled berg is sampled too, from each of its two bergs (FW:5456-5490),
 * and kid_write_bond_trajectories writes / extends bond_trajectories.nc (write_bond_trajectory, icebergs_fms2io.F90:2106-2331:
 * lon, lat, year, day, length, n1, n2, id_cnt1, id_ij1, id_cnt2, id_ij2 and, with dem, tangd1, tangd2, nstress, sstress,
 * rel_rotation, broken); with no bond records pending it leaves the file system alone, as the reference does. */
int kid_set_traj_params(kid_handle *h, const kid_traj_params *tp);
int kid_record_posn(kid_handle *h);
int kid_num_traj_records(kid_handle *h, int64_t *n);
int kid_write_trajectories(kid_handle *h, const char *path);
int kid_num_bond_traj_records(kid_handle *h, int64_t *n);
int kid_write_bond_trajectories(kid_handle *h, const char *path);

/* ---- berg migration between the handles of a domain-decomposed model (first slice; send_bergs_to_other_pes
 * FW:2997-3247).  The handle packs and unpacks the reference's wire format (pack_berg_into_buffer2 FW:3250-3301,
 * unpack_berg_from_buffer2 FW:3455-3680: buffer_width reals per berg, integers as reals); the caller moves the buffers
 * (mpp_send / mpp_recv), so a GPU rank can exchange bergs with ranks running the reference.  Call order per step, as FW:3022-3230:
 * pack E, pack W, send/recv, unpack both, then pack N, pack S, send/recv, unpack both -- after kid_evolve_icebergs /
 * kid_step_local (a berg that left the computational domain is dead on the device with its post-evolve state kept; packing
 * consumes it).  kid_pack_emigrants: *n = bergs selected for `dir` (KID_DIR_*); if *n > capacity nothing is packed and
 * KID_ECAPACITY is returned.  kid_unpack_immigrants appends, resets the *_old fields (FW:3573-3577), finds each berg's cell on
 * this grid (check_and_find_cell FW:5973-6008) and its xi / yj (FW:3634); a berg no cell of the data domain takes is dropped
 * and KID_EINVAL returned (the reference's FATAL, FW:3660).  Bonds, mts and dem: KID_EUNSUPPORTED. */
/* A decomposed host calls kid_buffer_width once at initialisation (ice_bergs_framework_init sizes its buffers there,
 * FW:1263-1292).  From the first migration call on the handle is in "decomposed" mode: a berg that left the tile is a dead row
 * whose cell lies outside the computational domain until a pack call has collected it, and such rows survive whatever the host
 * does between kid_evolve_icebergs and the exchange -- kid_move_berg_between_cells (the reference's own order, IB:5437-5447),
 * the dead-tail drop of kid_num_bergs, kid_compact_bergs; the rows of bergs already packed are reclaimed by those calls, and
 * by kid_unpack_immigrants itself before it reports KID_ECAPACITY.  A handle that never makes a migration call deletes leavers
 * at once, as a PE without neighbours does (FW:3024-3041). */
int kid_buffer_width(kid_handle *h, int32_t *width);
int kid_pack_emigrants(kid_handle *h, int32_t dir, double *buf, int64_t capacity, int64_t *n);
int kid_unpack_immigrants(kid_handle *h, const double *buf, int64_t n);
/* the two directions of one exchange pass in one launch and one host read each way: axis 0 = east (a) and west (b), axis 1 =
 * north (a) and south (b); the pair unpack appends the rows of buf_a, then those of buf_b (the order of the reference's two
 * unpack loops, FW:3064-3097).  Same results as the single calls. */
int kid_pack_emigrants_pair(kid_handle *h, int32_t axis, double *buf_a, int64_t capacity_a, int64_t *n_a, double *buf_b, int64_t capacity_b, int64_t *n_b);
int kid_unpack_immigrants_pair(kid_handle *h, const double *buf_a, int64_t n_a, const double *buf_b, int64_t n_b);

/* kid_set_forcing_device + kid_zero_accumulators for the step about to start, as one per-cell launch (fields == NULL
 * keeps the current forcing and only zeroes); the following kid_step_local does not zero again. */
int kid_step_prepare(kid_handle *h, const double *const device_fields[KID_NFORCING]);

/* ---- the hot path, phase by phase (same order as icebergs_run, IB:5423-5512) ---- */
int kid_zero_accumulators(kid_handle *h);
int kid_interp_gridded_fields_to_bergs(kid_handle *h);
int kid_evolve_icebergs(kid_handle *h);
int kid_footloose_calving(kid_handle *h);
/* Footloose children are placed on their parent's perimeter (displace_fl_bergs, IB:2631, 2664, 6432-6498) with the
 * counter-based generator of include/kid_rng.h: rn = f(kid_params.fl_rng_seed, parent id, footloose step, draw).  The step
 * counts the footloose passes of this handle (0 at kid_create, +1 per kid_footloose_calving or fused footloose step);
 * a restarted run sets it to continue the sequence.  (The reference seeds FMS's stream from the PE and the time, IB:2548.) */
double kid_footloose_uniform(int32_t seed, int64_t berg_id, int64_t step, int32_t draw);   /* the generator, for hosts that want to check a placement */
int kid_set_footloose_step(kid_handle *h, int64_t step);
int kid_get_footloose_step(kid_handle *h, int64_t *step);
int kid_thermodynamics(kid_handle *h);
int kid_create_gridded_icebergs_fields(kid_handle *h);

/* ---- bonded bergs, multiple time stepping, DEM (mts=T) ----
 * Bond lists travel as kid_bond_soa (include/kid_types.h), uploaded after the bergs they belong to; rows of bond
 * partners are re-derived from other_id (connect_all_bonds, FW:4963-5125).  While bonds exist the SoA is never
 * re-binned, so rows are stable between upload and download.
 *   kid_evolve_icebergs_mts   evolve_icebergs_mts IB:5431 / IB:6576-7078 (kid_evolve_icebergs dispatches to it when mts=T)
 *   kid_set_conglom_ids       transfer_mts_bergs IB:5459 -> set_conglom_ids FW:2601-2646 with whole conglomerates resident
 * kid_step_local / kid_run_step run the mts=T sequence of icebergs_run, including the once-after-upload `Visited`
 * block (IB:5409-5420: interpolate, label conglomerates, orig_bond_length). */
int kid_upload_bonds(kid_handle *h, const kid_bond_soa *host);
int kid_download_bonds(kid_handle *h, kid_bond_soa *host);
int kid_evolve_icebergs_mts(kid_handle *h);
int kid_set_conglom_ids(kid_handle *h);
/* evolve_icebergs (IB:7081-7200) with interactive_icebergs_on under the single-time-step Verlet scheme: velocity sweep
 * with the spring/damping terms of interactive_force inside accel, then the position sweep (kid_evolve_icebergs
 * dispatches to it when interactive_icebergs_on and mts=F) */
int kid_evolve_icebergs_interactive(kid_handle *h);

/* grd%iceberg_counter_grd (FW:1017), the per-cell counter generate_id draws berg ids from; (isd:ied,jsd:jed) int32 */
int kid_set_iceberg_counter(kid_handle *h, const int32_t *counter);
int kid_get_iceberg_counter(kid_handle *h, int32_t *counter);

/* ---- the hot path, fused: one launch does evolve + thermodynamics + mass spreading per berg ---- */
int kid_step_local(kid_handle *h);   /* zero accumulators, per-berg kernel(s); accumulators hold LOCAL sums */
int kid_step_gather(kid_handle *h);  /* 9-point gather + derived fields (after the cross-GPU all-reduce) */
int kid_run_step(kid_handle *h, int nsteps); /* nsteps x (kid_step_local; kid_step_gather) */

/* ---- results ---- */
/* acc: KID_NACC fields, out: KID_NOUT fields, each (ied-isd+1)*(jed-jsd+1); scalars: KID_NSCALAR running totals
 * since kid_create (the reference keeps them on `bergs`). NULL skips. */
int kid_get_accumulators(kid_handle *h, double *acc, double *out, double *scalars);
/* Device view of the accumulator block for RCCL: KID_NSCALAR + KID_NACC*ncell contiguous doubles -- the step's scalar
 * increments first, then the planes, so that the scalars and the planes a step fills (a prefix of the planes: distributed.py
 * accumulator_views) are one contiguous range, one all-reduce. */
int kid_accum_device_ptr(kid_handle *h, void **dev_ptr, int64_t *count);
/* How many doubles from the start of that block one step fills under the current parameters (the scalar increments + the
 * planes that are zeroed, scattered into and read by the gather): what a sharded run sums across GPUs between
 * kid_step_local and kid_step_gather.  MPI hosts: MPI_Allreduce(MPI_IN_PLACE, dev_ptr, live_count, MPI_DOUBLE, MPI_SUM). */
int kid_accum_live_count(kid_handle *h, int64_t *count);
/* Use caller-owned device memory (e.g. a torch tensor) for the accumulator block instead. */
int kid_bind_accum_buffer(kid_handle *h, void *dev_ptr, int64_t count);
/* find_melt_using_spread_mass (IB:5490-5503): grd%spread_mass_old -- the gridded mass before the thermodynamics -- and, with
 * Iceberg_melt_without_decay, spread_mass_tmp (IB:3411-3413) live in two planes that kid_step_local fills and kid_step_gather
 * reads.  Both are 9-point gathers of per-cell sums, i.e. linear in what each GPU's bergs contribute: a sharded run binds a
 * buffer of its own here (2 planes of ni*nj fp64) and all-reduces it between kid_step_local and kid_step_gather together with
 * the accumulator planes.  NULL returns to the handle's own planes. */
int kid_bind_spread_mass_old(kid_handle *h, void *dev_ptr, int64_t count);

/* ---- measurement: HIP-event timing of the per-berg kernel on the launch stream ---- */
int kid_profile_enable(kid_handle *h, int on);
/* telemetry: bergs that the hot build handed to the general build in the most recent per-berg launch */
int kid_last_redo_count(kid_handle *h, int64_t *count);
int kid_profile_get(kid_handle *h, double *berg_kernel_ms_total, int64_t *berg_kernel_launches,
                    double *all_ms_total);

#ifdef __cplusplus
}
#endif
#endif /* KID_H */
