!> ISO_C_BINDING shim over libkid_hip.so (include/kid.h): the Fortran host side of the MI355X evolve loop.
!!
!! The reference host is Fortran (icebergs_init / icebergs_run in src/icebergs.F90:92-178, 5074-5887) and stays
!! Fortran: this module is all a Fortran caller needs to drive the HIP kernels.  Each wrapper maps one call site of
!! icebergs_run to one C-ABI entry point (see the table in include/kid.h) and turns a non-zero status into the
!! reference's abort-on-error convention through kid_check (the reference calls error_mesg(...,FATAL), e.g.
!! icebergs.F90:3207; here `error stop`, so that the module has no FMS dependency).
!! The derived types and enum constants are generated from include/kid_types.h (tools/gen_fortran_types.py).
module kid_hip_mod
  use, intrinsic :: iso_c_binding
  implicit none
  private

  include 'kid_types_gen.inc'

  integer(c_int), parameter, public :: KID_OK = 0, KID_EINVAL = -1, KID_EHIP = -2, KID_ENODEV = -3, &
                                       KID_ECAPACITY = -4, KID_EUNSUPPORTED = -5

  public :: kid_grid_desc, kid_params, kid_berg_soa
  public :: kid_create, kid_destroy, kid_set_params, kid_sync, kid_set_static_grid, kid_set_forcing
  public :: kid_upload_bergs, kid_download_bergs, kid_num_bergs, kid_compact_bergs
  public :: kid_forcing_in, kid_ingest_forcing, kid_get_forcing
  public :: kid_calving_params, kid_calving_in, kid_set_calving_params, kid_set_calving_state, kid_get_calving_state
  public :: kid_calving, kid_get_calving
  public :: kid_write_restart, kid_read_restart, kid_bergs_chksum
  public :: kid_buffer_width, kid_pack_emigrants, kid_unpack_immigrants, kid_pack_emigrants_pair, kid_unpack_immigrants_pair
  public :: kid_traj_params, kid_set_traj_params, kid_record_posn, kid_write_trajectories, kid_write_bond_trajectories
  public :: kid_zero_accumulators, kid_interp_gridded_fields_to_bergs, kid_evolve_icebergs, kid_footloose_calving
  public :: kid_set_footloose_step, kid_get_footloose_step
  public :: kid_thermodynamics, kid_create_gridded_icebergs_fields, kid_step_local, kid_step_gather, kid_run_step
  public :: kid_get_accumulators, kid_last_error_f, kid_check
  ! every entry point of include/kid.h is usable from Fortran (tests/test_abi.py checks the list against the header)
  public :: kid_bond_soa, kid_accum_device_ptr, kid_accum_live_count, kid_bind_accum_buffer, kid_bind_spread_mass_old, kid_download_bonds
  public :: kid_evolve_icebergs_interactive, kid_evolve_icebergs_mts, kid_footloose_uniform, kid_get_iceberg_counter
  public :: kid_last_error, kid_last_redo_count, kid_move_berg_between_cells, kid_num_bond_traj_records
  public :: kid_num_traj_records, kid_profile_enable, kid_profile_get, kid_restart_count_bergs
  public :: kid_restart_read_bergs, kid_restart_read_bonds, kid_restart_write_bergs, kid_restart_write_bonds
  public :: kid_set_conglom_ids, kid_set_forcing_device, kid_set_iceberg_counter, kid_set_resort_interval
  public :: kid_set_side_stream, kid_set_store_environment, kid_set_stream, kid_sizeof, kid_step_prepare
  public :: kid_upload_bonds, kid_version

  interface
    integer(c_int) function kid_create(grid, params, capacity, device, handle) bind(C, name='kid_create')
      import :: c_int, c_int64_t, c_ptr, kid_grid_desc, kid_params
      type(kid_grid_desc), intent(in) :: grid
      type(kid_params), intent(in) :: params
      integer(c_int64_t), value :: capacity
      integer(c_int), value :: device
      type(c_ptr), intent(out) :: handle
    end function
    integer(c_int) function kid_destroy(h) bind(C, name='kid_destroy')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_set_params(h, params) bind(C, name='kid_set_params')
      import :: c_int, c_ptr, kid_params
      type(c_ptr), value :: h
      type(kid_params), intent(in) :: params
    end function
    integer(c_int) function kid_sync(h) bind(C, name='kid_sync')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    !> fields(KID_NGRID_STATIC): c_loc of grd%lon, grd%lat, grd%lonc, ... (KID_G_* order), each (isd:ied,jsd:jed)
    integer(c_int) function kid_set_static_grid(h, fields) bind(C, name='kid_set_static_grid')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), intent(in) :: fields(*)
    end function
    !> fields(KID_NFORCING): c_loc of grd%uo, grd%vo, ... after the ingest block of icebergs_run (IB:5236-5383)
    integer(c_int) function kid_set_forcing(h, fields) bind(C, name='kid_set_forcing')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), intent(in) :: fields(*)
    end function
    !> the ingest block of icebergs_run on the device (IB:5236-5383): args holds c_loc of the coupler's uo, vo, ui, vi,
    !! tauxa, tauya, ssh, sst, cn, hi, sss (c_null_ptr when sss is absent), their extents and staggers
    integer(c_int) function kid_ingest_forcing(h, args) bind(C, name='kid_ingest_forcing')
      import :: c_int, c_ptr, kid_forcing_in
      type(c_ptr), value :: h
      type(kid_forcing_in), intent(in) :: args
    end function
    !> fields(KID_NFORCING): c_loc of grd%uo, grd%vo, ... to be filled for send_data (IB:5529-5548); c_null_ptr skips one
    integer(c_int) function kid_get_forcing(h, fields) bind(C, name='kid_get_forcing')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), intent(in) :: fields(*)
    end function
    !> the per-hemisphere calving tables of ice_bergs_framework_init (FW:1534-1551), tau_calving, bergs%restarted
    integer(c_int) function kid_set_calving_params(h, cp) bind(C, name='kid_set_calving_params')
      import :: c_int, c_ptr, kid_calving_params
      type(c_ptr), value :: h
      type(kid_calving_params), intent(in) :: cp
    end function
    !> read_restart_calving: c_loc of grd%stored_ice, grd%stored_heat, grd%rmean_calving, grd%rmean_calving_hflx (c_null_ptr keeps)
    integer(c_int) function kid_set_calving_state(h, stored_ice, stored_heat, rmean_calving, rmean_calving_hflx) &
        bind(C, name='kid_set_calving_state')
      import :: c_int, c_ptr
      type(c_ptr), value :: h, stored_ice, stored_heat, rmean_calving, rmean_calving_hflx
    end function
    integer(c_int) function kid_get_calving_state(h, stored_ice, stored_heat, rmean_calving, rmean_calving_hflx, real_calving) &
        bind(C, name='kid_get_calving_state')
      import :: c_int, c_ptr
      type(c_ptr), value :: h, stored_ice, stored_heat, rmean_calving, rmean_calving_hflx, real_calving
    end function
    !> IB:5203-5231 + accumulate_calving (IB:5388) + calve_icebergs (IB:5403); scalars(KID_NCALV_SCALARS) in KID_CS_* order
    integer(c_int) function kid_calving(h, args, scalars) bind(C, name='kid_calving')
      import :: c_int, c_ptr, c_double, kid_calving_in
      type(c_ptr), value :: h
      type(kid_calving_in), intent(in) :: args
      real(c_double), intent(out) :: scalars(*)
    end function
    integer(c_int) function kid_get_calving(h, calving, calving_hflx) bind(C, name='kid_get_calving')
      import :: c_int, c_ptr
      type(c_ptr), value :: h, calving, calving_hflx
    end function
    !> write_restart_bergs / read_restart_bergs (icebergs_fms2io.F90:124-631, 663-1049) from / into the resident state;
    !! dir is a C string (trim(restart_dir)//c_null_char)
    integer(c_int) function kid_write_restart(h, dir) bind(C, name='kid_write_restart')
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: dir(*)
    end function
    integer(c_int) function kid_bergs_chksum(h, chk) bind(C, name='kid_bergs_chksum')   ! bergs_chksum FW:6889: chksum..chksum5, #
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: chk(6)
    end function
    integer(c_int) function kid_read_restart(h, dir) bind(C, name='kid_read_restart')
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: dir(*)
    end function
    !> record_posn (FW:5328-5498) on the device and iceberg_trajectories.nc (icebergs_fms2io.F90:1631-2103) from its records
    integer(c_int) function kid_set_traj_params(h, tp) bind(C, name='kid_set_traj_params')
      import :: c_int, c_ptr, kid_traj_params
      type(c_ptr), value :: h
      type(kid_traj_params), intent(in) :: tp
    end function
    integer(c_int) function kid_record_posn(h) bind(C, name='kid_record_posn')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_write_trajectories(h, path) bind(C, name='kid_write_trajectories')
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function kid_buffer_width(h, width) bind(C, name='kid_buffer_width')
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), intent(out) :: width
    end function
    integer(c_int) function kid_pack_emigrants(h, dir, buf, capacity, n) bind(C, name='kid_pack_emigrants')
      import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
      type(c_ptr), value :: h
      integer(c_int32_t), value :: dir
      real(c_double), intent(out) :: buf(*)          ! obuffer%data(buffer_width, capacity), FW:3250
      integer(c_int64_t), value :: capacity
      integer(c_int64_t), intent(out) :: n
    end function
    integer(c_int) function kid_pack_emigrants_pair(h, axis, buf_a, capacity_a, n_a, buf_b, capacity_b, n_b) bind(C, name='kid_pack_emigrants_pair')
      import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
      type(c_ptr), value :: h
      integer(c_int32_t), value :: axis              ! 0: east (a) + west (b), 1: north (a) + south (b)
      real(c_double), intent(out) :: buf_a(*), buf_b(*)
      integer(c_int64_t), value :: capacity_a, capacity_b
      integer(c_int64_t), intent(out) :: n_a, n_b
    end function
    integer(c_int) function kid_unpack_immigrants_pair(h, buf_a, n_a, buf_b, n_b) bind(C, name='kid_unpack_immigrants_pair')
      import :: c_int, c_ptr, c_int64_t, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: buf_a(*), buf_b(*)
      integer(c_int64_t), value :: n_a, n_b
    end function
    integer(c_int) function kid_unpack_immigrants(h, buf, n) bind(C, name='kid_unpack_immigrants')
      import :: c_int, c_ptr, c_int64_t, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: buf(*)           ! ibuffer%data(buffer_width, n), FW:3455
      integer(c_int64_t), value :: n
    end function
    integer(c_int) function kid_write_bond_trajectories(h, path) bind(C, name='kid_write_bond_trajectories')
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: path(*)
    end function
    integer(c_int) function kid_upload_bergs(h, soa) bind(C, name='kid_upload_bergs')
      import :: c_int, c_ptr, kid_berg_soa
      type(c_ptr), value :: h
      type(kid_berg_soa), intent(in) :: soa
    end function
    integer(c_int) function kid_download_bergs(h, soa) bind(C, name='kid_download_bergs')
      import :: c_int, c_ptr, kid_berg_soa
      type(c_ptr), value :: h
      type(kid_berg_soa), intent(inout) :: soa
    end function
    integer(c_int) function kid_num_bergs(h, n_slots, n_alive) bind(C, name='kid_num_bergs')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: n_slots, n_alive
    end function
    integer(c_int) function kid_compact_bergs(h) bind(C, name='kid_compact_bergs')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_move_berg_between_cells(h) bind(C, name='kid_move_berg_between_cells')   ! IB:5437
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_set_resort_interval(h, steps) bind(C, name='kid_set_resort_interval')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: steps
    end function
    type(c_ptr) function kid_version() bind(C, name='kid_version')   ! a NUL-terminated string: the library and its build switches
      import :: c_ptr
    end function
    integer(c_int64_t) function kid_sizeof(which) bind(C, name='kid_sizeof')   ! ABI layout check against c_sizeof of the types below
      import :: c_int, c_int64_t
      integer(c_int), value :: which
    end function
    integer(c_int) function kid_set_stream(h, hip_stream) bind(C, name='kid_set_stream')
      import :: c_int, c_ptr
      type(c_ptr), value :: h, hip_stream
    end function
    integer(c_int) function kid_set_forcing_device(h, dev_fields) bind(C, name='kid_set_forcing_device')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), intent(in) :: dev_fields(*)
    end function
    integer(c_int) function kid_accum_device_ptr(h, dev_ptr, count) bind(C, name='kid_accum_device_ptr')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      type(c_ptr), intent(out) :: dev_ptr
      integer(c_int64_t), intent(out) :: count
    end function
    integer(c_int) function kid_accum_live_count(h, count) bind(C, name='kid_accum_live_count')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: count
    end function
    integer(c_int) function kid_bind_accum_buffer(h, dev_ptr, count) bind(C, name='kid_bind_accum_buffer')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h, dev_ptr
      integer(c_int64_t), value :: count
    end function
    integer(c_int) function kid_bind_spread_mass_old(h, dev_ptr, count) bind(C, name='kid_bind_spread_mass_old')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h, dev_ptr
      integer(c_int64_t), value :: count
    end function
    real(c_double) function kid_footloose_uniform(seed, berg_id, step, draw) bind(C, name='kid_footloose_uniform')
      import :: c_double, c_int32_t, c_int64_t
      integer(c_int32_t), value :: seed, draw
      integer(c_int64_t), value :: berg_id, step
    end function
    integer(c_int) function kid_num_traj_records(h, n) bind(C, name='kid_num_traj_records')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: n
    end function
    integer(c_int) function kid_num_bond_traj_records(h, n) bind(C, name='kid_num_bond_traj_records')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: n
    end function
    integer(c_int) function kid_profile_enable(h, on) bind(C, name='kid_profile_enable')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function kid_profile_get(h, berg_kernel_ms_total, berg_kernel_launches, all_ms_total) bind(C, name='kid_profile_get')
      import :: c_int, c_ptr, c_double, c_int64_t
      type(c_ptr), value :: h
      real(c_double), intent(out) :: berg_kernel_ms_total, all_ms_total
      integer(c_int64_t), intent(out) :: berg_kernel_launches
    end function
    ! the restart files without a handle (host arrays in, host arrays out): what kid_write_restart / kid_read_restart are made of
    integer(c_int) function kid_restart_count_bergs(path, n) bind(C, name='kid_restart_count_bergs')
      import :: c_int, c_char, c_int64_t
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int64_t), intent(out) :: n
    end function
    integer(c_int) function kid_restart_write_bergs(path, p, host) bind(C, name='kid_restart_write_bergs')
      import :: c_int, c_char, kid_params, kid_berg_soa
      character(kind=c_char), intent(in) :: path(*)
      type(kid_params), intent(in) :: p
      type(kid_berg_soa), intent(in) :: host
    end function
    integer(c_int) function kid_restart_read_bergs(path, host, capacity) bind(C, name='kid_restart_read_bergs')
      import :: c_int, c_char, c_int64_t, kid_berg_soa
      character(kind=c_char), intent(in) :: path(*)
      type(kid_berg_soa), intent(inout) :: host
      integer(c_int64_t), value :: capacity
    end function
    integer(c_int) function kid_restart_write_bonds(path, p, bergs, bonds) bind(C, name='kid_restart_write_bonds')
      import :: c_int, c_char, kid_params, kid_berg_soa, kid_bond_soa
      character(kind=c_char), intent(in) :: path(*)
      type(kid_params), intent(in) :: p
      type(kid_berg_soa), intent(in) :: bergs
      type(kid_bond_soa), intent(in) :: bonds
    end function
    integer(c_int) function kid_restart_read_bonds(path, bergs, bonds) bind(C, name='kid_restart_read_bonds')
      import :: c_int, c_char, kid_berg_soa, kid_bond_soa
      character(kind=c_char), intent(in) :: path(*)
      type(kid_berg_soa), intent(in) :: bergs
      type(kid_bond_soa), intent(inout) :: bonds
    end function
    integer(c_int) function kid_zero_accumulators(h) bind(C, name='kid_zero_accumulators')          ! IB:5125-5156
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_interp_gridded_fields_to_bergs(h) bind(C, name='kid_interp_gridded_fields_to_bergs') ! IB:5423
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_evolve_icebergs(h) bind(C, name='kid_evolve_icebergs')              ! IB:5433
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_footloose_calving(h) bind(C, name='kid_footloose_calving')          ! IB:5453
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_set_footloose_step(h, step) bind(C, name='kid_set_footloose_step')  ! child-placement sequence (kid_rng.h)
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), value :: step
    end function
    integer(c_int) function kid_get_footloose_step(h, step) bind(C, name='kid_get_footloose_step')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: step
    end function
    integer(c_int) function kid_step_prepare(h, fields) bind(C, name='kid_step_prepare')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), intent(in) :: fields(*)
    end function
    integer(c_int) function kid_set_side_stream(h, side_stream, enable) bind(C, name='kid_set_side_stream')
      import :: c_int, c_ptr
      type(c_ptr), value :: h, side_stream
      integer(c_int), value :: enable
    end function
    integer(c_int) function kid_last_redo_count(h, count) bind(C, name='kid_last_redo_count')
      import :: c_int, c_ptr, c_int64_t
      type(c_ptr), value :: h
      integer(c_int64_t), intent(out) :: count
    end function
    integer(c_int) function kid_upload_bonds(h, soa) bind(C, name='kid_upload_bonds')
      import :: c_int, c_ptr, kid_bond_soa
      type(c_ptr), value :: h
      type(kid_bond_soa), intent(in) :: soa
    end function
    integer(c_int) function kid_download_bonds(h, soa) bind(C, name='kid_download_bonds')
      import :: c_int, c_ptr, kid_bond_soa
      type(c_ptr), value :: h
      type(kid_bond_soa), intent(inout) :: soa
    end function
    integer(c_int) function kid_evolve_icebergs_mts(h) bind(C, name='kid_evolve_icebergs_mts')   ! IB:5431
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_evolve_icebergs_interactive(h) bind(C, name='kid_evolve_icebergs_interactive')  ! IB:5433 with interactive bergs
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_set_conglom_ids(h) bind(C, name='kid_set_conglom_ids')           ! IB:5459 / FW:2601
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_set_store_environment(h, on) bind(C, name='kid_set_store_environment')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function kid_set_iceberg_counter(h, counter) bind(C, name='kid_set_iceberg_counter')  ! grd%iceberg_counter_grd, FW:1017
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), intent(in) :: counter(*)
    end function
    integer(c_int) function kid_get_iceberg_counter(h, counter) bind(C, name='kid_get_iceberg_counter')
      import :: c_int, c_ptr, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), intent(inout) :: counter(*)
    end function
    integer(c_int) function kid_thermodynamics(h) bind(C, name='kid_thermodynamics')                ! IB:5505
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_create_gridded_icebergs_fields(h) bind(C, name='kid_create_gridded_icebergs_fields') ! IB:5512
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_step_local(h) bind(C, name='kid_step_local')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_step_gather(h) bind(C, name='kid_step_gather')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function kid_run_step(h, nsteps) bind(C, name='kid_run_step')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: nsteps
    end function
    integer(c_int) function kid_get_accumulators(h, acc, out, scalars) bind(C, name='kid_get_accumulators')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      type(c_ptr), value :: acc, out, scalars
    end function
    type(c_ptr) function kid_last_error(h) bind(C, name='kid_last_error')
      import :: c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_size_t) function c_strlen(s) bind(C, name='strlen')
      import :: c_ptr, c_size_t
      type(c_ptr), value :: s
    end function
  end interface

contains

  !> The library's last error message as a Fortran string
  function kid_last_error_f(h) result(msg)
    type(c_ptr), intent(in) :: h
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: chars(:)
    integer :: n, k
    p = kid_last_error(h)
    msg = ''
    if (.not. c_associated(p)) return
    n = int(c_strlen(p))
    if (n <= 0) return
    call c_f_pointer(p, chars, [n])
    allocate(character(len=n) :: msg)
    do k = 1, n
      msg(k:k) = chars(k)
    end do
  end function

  !> Abort on a non-zero status: the reference's error_mesg(...,FATAL) convention (e.g. icebergs.F90:3207)
  subroutine kid_check(rc, h, what)
    integer(c_int), intent(in) :: rc
    type(c_ptr), intent(in) :: h
    character(len=*), intent(in) :: what
    if (rc /= KID_OK) then
      write(0, '(a,a,a,i0,a,a)') 'KID, ', what, ': status ', rc, ' ', kid_last_error_f(h)
      error stop 1
    end if
  end subroutine

end module kid_hip_mod
