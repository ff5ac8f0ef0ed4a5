!> kid_icebergs_glue -- the piece of Fortran that sits between the reference's linked lists of bergs and the HIP library.
!!
!! In the reference every berg is a heap node of `type iceberg` (icebergs_framework.F90:290-359) in a doubly linked list per
!! ocean cell, `bergs%list(isd:ied,jsd:jed)` (FW:423), kept sorted by `inorder` (FW:4318-4359) when parallel_reprod is on
!! (FW:4270-4305), and the hot loops walk the cells j-outer / i-inner and each list front to back (icebergs.F90:7106).  The
!! library wants one array per member in exactly that order (SURVEY A13).  This module
!!   * declares the berg node with the reference's member names and the per-cell lists (own declarations: the reference's
!!     module needs FMS, which is not available to this build; a maintainer replaces `use kid_icebergs_glue, only: iceberg,
!!     linked_list` by `use ice_bergs_framework` and deletes the two types),
!!   * restates `inorder` and the sorted `insert_berg_into_list`,
!!   * flattens the lists into the structure of arrays in traversal order and uploads it (kid_glue_flatten), downloads and
!!     rebuilds the lists (kid_glue_unflatten),
!!   * and wraps one coupling step behind the argument list of `icebergs_run` (IB:5074-5096; FMS's time_type replaced by
!!     the (year, yearday) pair the path reads from it, IB:5169-5175): forcing ingest, calving source, the evolve loop,
!!     the return of unused calving + melt to the coupler (IB:5654-5679);
!!   * `kid_icebergs_init` with the argument list of `icebergs_init` (IB:92-117, time_type again as (year, yearday)): reads the
!!     reference's own namelist group `icebergs_nml` from input.nml (every variable of FW:823-856 is declared, kid_nml_gen.inc),
!!     derives what ice_bergs_framework_init derives from it (FW:1290-1530), builds the data-domain grid the way FW:1021-1200
!!     does on one PE (halo copies for a cyclic x direction, extrapolated corner coordinates elsewhere, the periodicity fix,
!!     cell centres) and creates the handle;
!!   * `type bond` (FW:362-386) and the per-berg bond lists, flattened to / rebuilt from the library's slot-major bond tables in
!!     list order (form_a_bond puts a new bond at the head, FW:4866-4877).
!! tests/test_fortran_gpu.py::test_glue_* drive it through kid_glue_test.F90 against the oracle.
module kid_icebergs_glue
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  implicit none
  private
  public :: iceberg, bond, linked_list, kid_glue, inorder, insert_berg_into_list, kid_glue_init, kid_glue_set_calving, kid_glue_add_berg, kid_glue_count, &
            kid_glue_flatten, kid_glue_unflatten, kid_glue_clear_lists, kid_glue_end, kid_icebergs_run, kid_icebergs_init, kid_read_icebergs_nml, &
            kid_icebergs_init_bonds, kid_icebergs_run_local, kid_icebergs_run_finish, kid_exchange_sum, row_to_node, node_to_row, form_a_bond, kid_glue_find_berg, KID_CYCLIC_GLOBAL_DOMAIN
  !> FMS's mpp_domains flag for a zonally periodic global domain (mpp_parameter_mod), as DRV:44 passes it in dom_x_flags
  integer, parameter :: KID_CYCLIC_GLOBAL_DOMAIN = 2

#include "kid_nml_gen.inc"


  !> the members of the reference's `type iceberg` (FW:290-359) that the path reads or writes, same names
  type :: iceberg
    type(iceberg), pointer :: prev => null(), next => null()
    real(c_double) :: lon = 0., lat = 0., uvel = 0., vvel = 0., mass = 0., thickness = 0., width = 0., length = 0.
    real(c_double) :: start_lon = 0., start_lat = 0., start_day = 0., start_mass = 0., mass_scaling = 0.
    real(c_double) :: mass_of_bits = 0., mass_of_fl_bits = 0., mass_of_fl_bergy_bits = 0., fl_k = 0., heat_density = 0.
    real(c_double) :: xi = 0., yj = 0.
    real(c_double) :: uo = 0., vo = 0., ui = 0., vi = 0., ua = 0., va = 0., ssh_x = 0., ssh_y = 0., sst = 0., sss = 0., cn = 0., hi = 0., od = 0.
    real(c_double) :: axn = 0., ayn = 0., bxn = 0., byn = 0., uvel_prev = 0., vvel_prev = 0.
    real(c_double) :: uvel_old = 0., vvel_old = 0., lon_old = 0., lat_old = 0.
    real(c_double) :: halo_berg = 0., static_berg = 0.
    real(c_double) :: axn_fast = 0., ayn_fast = 0., bxn_fast = 0., byn_fast = 0., ang_vel = 0., ang_accel = 0., rot = 0.   ! FW:348-358
    integer :: start_year = 0, ine = 0, jne = 0, n_bonds = 0, conglom_id = 0
    integer(c_int64_t) :: id = 0
    type(bond), pointer :: first_bond => null()   ! FW:346
  end type iceberg

  !> FW:362-386 (the DEM members are plain components here, pointers to reals there)
  type :: bond
    type(bond), pointer :: prev_bond => null(), next_bond => null()
    type(iceberg), pointer :: other_berg => null()
    integer(c_int64_t) :: other_id = 0
    integer :: other_berg_ine = 0, other_berg_jne = 0, broken = 0
    real(c_double) :: length = 0., tangd1 = 0., tangd2 = 0., nstress = 0., sstress = 0., rel_rotation = 0.
    real(c_double) :: F_x = 0., F_y = 0., Fd_x = 0., Fd_y = 0., T = 0., T_d = 0.
  end type bond

  type :: linked_list   ! FW:416-419
    type(iceberg), pointer :: first => null()
  end type linked_list

  !> what `type icebergs` (FW:421-616) carries for this path: the handle, the grid extents, the lists, staging arrays
  type :: kid_glue
    type(c_ptr) :: h = c_null_ptr
    type(kid_grid_desc) :: gd
    type(kid_params) :: par
    type(linked_list), allocatable :: list(:,:)               ! (isd:ied, jsd:jed), FW:423
    real(c_double), allocatable :: area(:,:)                   ! grd%area on the data domain (the coupler return needs it, IB:5655)
    real(c_double), allocatable :: static(:,:,:)               ! the grid kid_icebergs_init built, KID_G_* planes on the data domain
    integer(c_int64_t) :: capacity = 0
    logical :: calving_on = .false.        ! the calving source runs on the device too (kid_glue_set_calving)
    logical :: tau_is_velocity = .false.   ! bergs%tau_is_velocity (FW:727): tauxa / tauya are winds, not stresses
    logical :: passive_mode = .false.      ! bergs%passive_mode (FW:728): nothing is returned to the coupler
    integer :: resort_interval = 16, since_sort = 0   ! rows are re-binned by cell every so many steps (kid_set_resort_interval)
    real(c_double), allocatable :: f64(:,:)                    ! (capacity, KID_NB_F64): one column per member
    integer(c_int32_t), allocatable :: i32(:,:)                ! (capacity, KID_NB_I32)
    integer(c_int64_t), allocatable :: ids(:)
    real(c_double), allocatable :: acc(:,:,:), outp(:,:,:), scal(:), gcalv(:,:), ghflx(:,:)
    ! bond tables as the library takes them (kid_bond_soa: slot-major, element (k, s) = row k, slot s)
    integer(c_int32_t), allocatable :: bcount(:), bbroken(:,:)
    integer(c_int64_t), allocatable :: bother(:,:)
    real(c_double), allocatable :: bf64(:,:,:)              ! (capacity, max_bonds, KID_NBOND_F64)
  end type kid_glue

  abstract interface
    !> sums `count` doubles at the DEVICE address `buf` over the ranks that share the grid, in place (GPU-aware MPI:
    !! MPI_Allreduce(MPI_IN_PLACE, buf, count, MPI_DOUBLE_PRECISION, MPI_SUM, comm) on the c_f_pointer view of buf)
    subroutine kid_exchange_sum(buf, count)
      import :: c_ptr, c_int64_t
      type(c_ptr), intent(in) :: buf
      integer(c_int64_t), intent(in) :: count
    end subroutine kid_exchange_sum
  end interface

contains

  !> FW:4318-4359: true when berg1 sorts before (or equal to) berg2: start_year, start_day, start_mass, start_lon, start_lat
  logical function inorder(berg1, berg2)
    type(iceberg), pointer :: berg1, berg2
    inorder = .true.
    if (berg1%start_year /= berg2%start_year) then ; inorder = berg1%start_year < berg2%start_year ; return ; endif
    if (berg1%start_day /= berg2%start_day) then ; inorder = berg1%start_day < berg2%start_day ; return ; endif
    if (berg1%start_mass /= berg2%start_mass) then ; inorder = berg1%start_mass < berg2%start_mass ; return ; endif
    if (berg1%start_lon /= berg2%start_lon) then ; inorder = berg1%start_lon < berg2%start_lon ; return ; endif
    if (berg1%start_lat /= berg2%start_lat) then ; inorder = berg1%start_lat < berg2%start_lat ; return ; endif
  end function inorder

  !> FW:4270-4305 with parallel_reprod: the new berg goes in front of the first berg it is `inorder` with
  subroutine insert_berg_into_list(first, newberg)
    type(iceberg), pointer :: first, newberg
    type(iceberg), pointer :: this, prev
    if (.not. associated(first)) then
      first => newberg ; newberg%next => null() ; newberg%prev => null()
      return
    endif
    this => first ; prev => null()
    do while (associated(this))
      if (inorder(newberg, this)) exit
      prev => this ; this => this%next
    enddo
    newberg%next => this ; newberg%prev => prev
    if (associated(this)) this%prev => newberg
    if (associated(prev)) then ; prev%next => newberg ; else ; first => newberg ; endif
  end subroutine insert_berg_into_list

  !> icebergs_init (IB:92-178) as far as the path needs it: parameters and static grid to the device, empty lists
  subroutine kid_glue_init(g, gd, par, static_planes, capacity)
    type(kid_glue), intent(inout) :: g
    type(kid_grid_desc), intent(in) :: gd
    type(kid_params), intent(in) :: par
    real(c_double), target, intent(in) :: static_planes(:,:,:)   ! (ni, nj, KID_NGRID_STATIC): grd%lon, lat, lonc, ... in KID_G_* order
    integer(c_int64_t), intent(in) :: capacity
    type(c_ptr) :: pst(KID_NGRID_STATIC)
    integer :: k, ni, nj
    g%gd = gd ; g%par = par ; g%capacity = capacity
    ni = gd%ied - gd%isd + 1 ; nj = gd%jed - gd%jsd + 1
    call kid_check(kid_create(gd, par, capacity, 0_c_int, g%h), g%h, 'kid_create')
    do k = 1, KID_NGRID_STATIC
      pst(k) = c_loc(static_planes(1,1,k))
    enddo
    call kid_check(kid_set_static_grid(g%h, pst), g%h, 'kid_set_static_grid')
    allocate(g%list(gd%isd:gd%ied, gd%jsd:gd%jed))
    allocate(g%area(ni, nj)) ; g%area = static_planes(:,:,KID_G_AREA+1)
    allocate(g%f64(capacity, KID_NB_F64), g%i32(capacity, KID_NB_I32), g%ids(capacity))
    allocate(g%acc(ni, nj, KID_NACC), g%outp(ni, nj, KID_NOUT), g%scal(KID_NSCALAR), g%gcalv(ni, nj), g%ghflx(ni, nj))
  end subroutine kid_glue_init

  !> Reads the reference's namelist group from `path` (FMS reads it from input.nml, FW:879-880): defaults first, then the file.
  !! A missing file or a missing group leaves the defaults, as check_nml_error accepts an absent group; any other read error stops.
  subroutine kid_read_icebergs_nml(path)
    character(len=*), intent(in) :: path
    integer :: u, ierr
    logical :: there
    character(len=256) :: msg
#include "kid_nml_defaults_gen.inc"
    inquire(file=path, exist=there)
    if (.not. there) return
    open(newunit=u, file=path, status='old', action='read')
    read(u, nml=icebergs_nml, iostat=ierr, iomsg=msg)
    close(u)
    if (ierr > 0) then
      write(*,'(a)') 'kid_read_icebergs_nml: '//trim(msg)
      error stop 'kid_read_icebergs_nml: error reading icebergs_nml'
    endif
    if (really_debug) debug = .true.   ! FW:883
  end subroutine kid_read_icebergs_nml

  !> icebergs_init (IB:92-117) on one PE: the same argument list (`Time` as the (year, yearday) pair the path reads from it;
  !! layout, io_layout, axes and maskmap are accepted and must describe one PE / are not used: diagnostics and I/O decomposition
  !! are the host model's).  ice_lon, ice_lat, ice_area and ocean_depth have the extents of the computational domain; ice_wet,
  !! ice_dx, ice_dy, cos_rot and sin_rot carry one halo cell more on every side (FW:1021-1038).  What it does:
  !!   namelist (FW:879) -> derived switches (FW:1243-1326, 1433-1531) -> data-domain grid (FW:921-1147) -> kid_create,
  !!   kid_set_static_grid, the class tables of the calving source (FW:1534-1551), the trajectory switches.
  !! `capacity` (not in the reference: its lists grow on the heap) sizes the device arrays; nml_file defaults to 'input.nml'.
  !! Restart reading stays with the caller (kid_read_restart, or kid_glue_add_berg + form_a_bond for a population of its own);
  !! call kid_icebergs_init_bonds after the population is in the lists for the bonded / DEM tail of icebergs_init (IB:141-176).
  subroutine kid_icebergs_init(bergs, gni, gnj, layout, io_layout, axes, dom_x_flags, dom_y_flags, dt, year, yearday, &
                               ice_lon, ice_lat, ice_wet, ice_dx, ice_dy, ice_area, cos_rot, sin_rot, ocean_depth, maskmap, fractional_area, &
                               capacity, nml_file, device)
    type(kid_glue), intent(inout), target :: bergs
    integer, intent(in) :: gni, gnj, layout(2), io_layout(2), axes(2), dom_x_flags, dom_y_flags
    real(c_double), intent(in) :: dt
    integer, intent(in) :: year
    real(c_double), intent(in) :: yearday
    real(c_double), dimension(:,:), intent(in) :: ice_lon, ice_lat, ice_wet, ice_dx, ice_dy, ice_area, cos_rot, sin_rot
    real(c_double), dimension(:,:), intent(in), optional :: ocean_depth
    logical, intent(in), optional :: maskmap(:,:), fractional_area
    integer(c_int64_t), intent(in), optional :: capacity
    character(len=*), intent(in), optional :: nml_file
    integer, intent(in), optional :: device
    real(c_double), parameter :: fms_pi = 3.14159265358979323846_c_double, fms_omega = 7.292e-5_c_double, fms_hlf = 3.34e5_c_double, &
                                 fms_radius = 6371.0e3_c_double   ! constants_mod (FW:6): pi, omega, HLF, radius
    real(c_double), parameter :: big_number = 1.0e15_c_double
    type(kid_grid_desc) :: gd
    type(kid_params) :: par
    type(kid_calving_params) :: cp
    type(kid_traj_params) :: tp
    real(c_double), allocatable, target :: st(:,:,:)
    real(c_double), pointer :: lon(:,:), lat(:,:), lonc(:,:), latc(:,:), gdx(:,:), gdy(:,:), area(:,:), msk(:,:), gcos(:,:), gsin(:,:), depth(:,:)
    type(c_ptr) :: pst(KID_NGRID_STATIC)
    real(c_double) :: lon_mod, mts_fast_dt, dx_dlon, dy_dlat, ddx, ddy, pi_180, total_s, total_n, rem_s, rem_n
    integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, ni, nj, i, j, k, maxk, f, last_s, last_n, dev
    integer(c_int64_t) :: cap
    logical :: cyclic_x, cyclic_y

    if (layout(1) * layout(2) /= 1) error stop 'kid_icebergs_init: one handle owns one rectangle of cells -- call it per PE with that PE''s extents (INTEGRATION.md section 6)'
    if (present(maskmap)) then ; if (.not. all(maskmap)) error stop 'kid_icebergs_init: maskmap must keep the one PE' ; endif
    if (size(ice_lon,1) /= gni .or. size(ice_lon,2) /= gnj) error stop 'kid_icebergs_init: ice_lon must cover the computational domain (gni, gnj)'
    if (size(ice_dx,1) /= gni + 2 .or. size(ice_dx,2) /= gnj + 2) error stop 'kid_icebergs_init: ice_dx must cover the computational domain + 1 halo cell'
    if (present(nml_file)) then ; call kid_read_icebergs_nml(nml_file) ; else ; call kid_read_icebergs_nml('input.nml') ; endif
    pi_180 = fms_pi / 180.
    cyclic_x = dom_x_flags == KID_CYCLIC_GLOBAL_DOMAIN ; cyclic_y = dom_y_flags == KID_CYCLIC_GLOBAL_DOMAIN
    if (.not. cyclic_x .and. dom_x_flags /= 0) error stop 'kid_icebergs_init: dom_x_flags must be 0 or CYCLIC_GLOBAL_DOMAIN'
    if (.not. cyclic_y .and. dom_y_flags /= 0) error stop 'kid_icebergs_init: dom_y_flags must be 0 or CYCLIC_GLOBAL_DOMAIN (no tripolar fold on this path)'

    ! ---- switches that other switches decide (FW:1183-1326) ----
    if (.not. separate_distrib_for_n_hemisphere) then   ! FW:1183-1186
      initial_mass_n = initial_mass ; distribution_n = distribution ; mass_scaling_n = mass_scaling ; initial_thickness_n = initial_thickness
    endif
    if (input_freq_distribution) then   ! FW:1190-1241: a frequency distribution becomes a mass-flux distribution
      total_s = 0. ; total_n = 0.
      do j = 1, nclasses
        total_s = total_s + (distribution(j) * initial_mass(j)) ; total_n = total_n + (distribution_n(j) * initial_mass_n(j))
      enddo
      do j = 1, nclasses
        distribution(j) = (distribution(j) * initial_mass(j)) / total_s ; distribution_n(j) = (distribution_n(j) * initial_mass_n(j)) / total_n
      enddo
      last_s = 1 ; last_n = 1
      do j = 1, nclasses
        if (distribution(j) > 0.) last_s = j
        if (distribution_n(j) > 0.) last_n = j
      enddo
      rem_s = 1. ; rem_n = 1.
      do j = 1, last_s - 1 ; rem_s = rem_s - distribution(j) ; enddo
      distribution(last_s) = rem_s
      do j = 1, last_n - 1 ; rem_n = rem_n - distribution_n(j) ; enddo
      distribution_n(last_n) = rem_n
    endif
    if ((halo < 3) .and. (rotate_icebergs_for_mass_spreading .and. iceberg_bonds_on)) then   ! FW:1243-1249
      halo = 3
    elseif ((halo < 2) .and. (interactive_icebergs_on .or. iceberg_bonds_on)) then
      halo = 2
    endif
    if (halo < 2) halo = 2   ! the library's own floor: the 9-point gather and the bounce read one cell around a berg's cell
    if (.not. iceberg_bonds_on) max_bonds = 0   ! FW:1263
    if (max_bonds > KID_MAX_BONDS) error stop 'kid_icebergs_init: max_bonds exceeds KID_MAX_BONDS'
    mts_fast_dt = 0.
    if (mts) then   ! FW:1291-1309
      if (mts_sub_steps == -1) then
        mts_fast_dt = 0.3 / sqrt(spring_coef)
        mts_sub_steps = ceiling(dt / mts_fast_dt)
      endif
      mts_fast_dt = dt / mts_sub_steps
      Runge_not_Verlet = .false.
    endif
    if (contact_spring_coef <= 0.) contact_spring_coef = spring_coef   ! FW:1313

    ! ---- the data domain of one PE (mpp_define_domains with xhalo = yhalo = halo, FW:915-924) ----
    isc = 1 ; iec = gni ; jsc = 1 ; jec = gnj
    isd = isc - halo ; ied = iec + halo ; jsd = jsc - halo ; jed = jec + halo
    ni = ied - isd + 1 ; nj = jed - jsd + 1
    allocate(st(isd:ied, jsd:jed, KID_NGRID_STATIC))
    lon => st(:,:,KID_G_LON+1) ; lat => st(:,:,KID_G_LAT+1) ; lonc => st(:,:,KID_G_LONC+1) ; latc => st(:,:,KID_G_LATC+1)
    gdx => st(:,:,KID_G_DX+1) ; gdy => st(:,:,KID_G_DY+1) ; area => st(:,:,KID_G_AREA+1) ; msk => st(:,:,KID_G_MSK+1)
    gcos => st(:,:,KID_G_COS+1) ; gsin => st(:,:,KID_G_SIN+1) ; depth => st(:,:,KID_G_OCEAN_DEPTH+1)
    ! (pointer sections of an allocatable with lower bounds isd, jsd start at 1: index them with the offsets below)
    lon = big_number ; lat = big_number ; lonc = 0. ; latc = 0.        ! FW:950-953
    gdx = 0. ; gdy = 0. ; area = 0. ; msk = 0. ; gcos = 1. ; gsin = 0. ; depth = 0.   ! FW:954-960
    ! computational domain (FW:1021-1024, 1042-1046) and the ring the ice model hands over (FW:1049-1055)
    call put(lon, ice_lon, isc, jsc) ; call put(lat, ice_lat, isc, jsc) ; call put(area, ice_area, isc, jsc)
    if (present(fractional_area)) then
      if (fractional_area) call put(area, ice_area * (4. * fms_pi * fms_radius * fms_radius), isc, jsc)
    endif
    if (present(ocean_depth)) call put(depth, ocean_depth, isc, jsc)
    call put(gdx, ice_dx, isc - 1, jsc - 1) ; call put(gdy, ice_dy, isc - 1, jsc - 1) ; call put(msk, ice_wet, isc - 1, jsc - 1)
    call put(gcos, cos_rot, isc - 1, jsc - 1) ; call put(gsin, sin_rot, isc - 1, jsc - 1)
    ! mpp_update_domains on one PE (FW:1057-1065): a cyclic direction copies the other side of the computational domain into
    ! the halo, a closed one leaves the halo as it is
    do f = 1, KID_NGRID_STATIC
      if (f == KID_G_LONC+1 .or. f == KID_G_LATC+1) cycle
      call update_domains(st(:,:,f))
    enddo
    ! FW:1067-1094: lon / lat that no neighbour filled: the southern halo copies lon and extrapolates lat, then every side
    ! extrapolates linearly
    do j = jsc - 1, jsd, -1 ; do i = isd, ied
      if (g2(lon, i, j) >= big_number) call s2(lon, i, j, g2(lon, i, j + 1))
      if (g2(lat, i, j) >= big_number) call s2(lat, i, j, 2. * g2(lat, i, j + 1) - g2(lat, i, j + 2))
    enddo ; enddo
    do j = jsc - 1, jsd, -1 ; do i = isd, ied
      if (g2(lon, i, j) >= big_number) call s2(lon, i, j, 2. * g2(lon, i, j + 1) - g2(lon, i, j + 2))
      if (g2(lat, i, j) >= big_number) call s2(lat, i, j, 2. * g2(lat, i, j + 1) - g2(lat, i, j + 2))
    enddo ; enddo
    do j = jec + 1, jed ; do i = isd, ied
      if (g2(lon, i, j) >= big_number) call s2(lon, i, j, 2. * g2(lon, i, j - 1) - g2(lon, i, j - 2))
      if (g2(lat, i, j) >= big_number) call s2(lat, i, j, 2. * g2(lat, i, j - 1) - g2(lat, i, j - 2))
    enddo ; enddo
    do i = isc - 1, isd, -1 ; do j = jsd, jed
      if (g2(lon, i, j) >= big_number) call s2(lon, i, j, 2. * g2(lon, i + 1, j) - g2(lon, i + 2, j))
      if (g2(lat, i, j) >= big_number) call s2(lat, i, j, 2. * g2(lat, i + 1, j) - g2(lat, i + 2, j))
    enddo ; enddo
    do i = iec + 1, ied ; do j = jsd, jed
      if (g2(lon, i, j) >= big_number) call s2(lon, i, j, 2. * g2(lon, i - 1, j) - g2(lon, i - 2, j))
      if (g2(lat, i, j) >= big_number) call s2(lat, i, j, 2. * g2(lat, i - 1, j) - g2(lat, i - 2, j))
    enddo ; enddo
    if ((.not. grid_is_latlon) .and. (Lx == 360.)) Lx = -1.   ! FW:1118-1123
    if (Lx > 0.) then   ! FW:1127-1148: the halo copies of a periodic direction move by whole periods next to their neighbours
      j = jsc ; do i = isc + 1, ied
        lon_mod = apply_modulo_around_point(g2(lon, i, j), g2(lon, i - 1, j), Lx)
        if (abs(g2(lon, i, j) - lon_mod) > (Lx / 2.)) call s2(lon, i, j, lon_mod)
      enddo
      j = jsc ; do i = isc - 1, isd, -1
        lon_mod = apply_modulo_around_point(g2(lon, i, j), g2(lon, i + 1, j), Lx)
        if (abs(g2(lon, i, j) - lon_mod) > (Lx / 2.)) call s2(lon, i, j, lon_mod)
      enddo
      do j = jsc + 1, jed ; do i = isd, ied
        lon_mod = apply_modulo_around_point(g2(lon, i, j), g2(lon, i, j - 1), Lx)
        if (abs(g2(lon, i, j) - lon_mod) > (Lx / 2.)) call s2(lon, i, j, lon_mod)
      enddo ; enddo
      do j = jsc - 1, jsd, -1 ; do i = isd, ied
        lon_mod = apply_modulo_around_point(g2(lon, i, j), g2(lon, i, j + 1), Lx)
        if (abs(g2(lon, i, j) - lon_mod) > (Lx / 2.)) call s2(lon, i, j, lon_mod)
      enddo ; enddo
    endif
    do j = jsd + 1, jed ; do i = isd + 1, ied   ! FW:1153-1158
      call s2(lonc, i, j, 0.25 * ((g2(lon, i, j) + g2(lon, i - 1, j - 1)) + (g2(lon, i - 1, j) + g2(lon, i, j - 1))))
      call s2(latc, i, j, 0.25 * ((g2(lat, i, j) + g2(lat, i - 1, j - 1)) + (g2(lat, i - 1, j) + g2(lat, i, j - 1))))
    enddo ; enddo
    do j = jsd + 1, jed ; do i = isd + 1, ied   ! FW:1170-1180
      if (g2(lat, i, j) /= g2(lat, i, j)) error stop 'kid_icebergs_init: latitude contains NaNs'
      if (g2(lon, i, j) /= g2(lon, i, j)) error stop 'kid_icebergs_init: longitude contains NaNs'
    enddo ; enddo

    gd%isd = isd ; gd%ied = ied ; gd%jsd = jsd ; gd%jed = jed ; gd%isc = isc ; gd%iec = iec ; gd%jsc = jsc ; gd%jec = jec
    gd%grid_is_latlon = merge(1, 0, grid_is_latlon) ; gd%grid_is_regular = merge(1, 0, grid_is_regular) ; gd%Lx = Lx
    gd%gni = gni ; gd%gnj = gnj ; gd%gi0 = 0 ; gd%gj0 = 0   ! one PE: local and global indices agree (generate_id, FW:6930-6951)

    ! ---- bergs%... = ... (FW:1328-1480) ----
    par%pi = fms_pi ; par%omega = fms_omega ; par%HLF = fms_hlf ; par%dt = dt
    par%current_year = year ; par%pad0 = 0 ; par%current_yearday = yearday
    par%Rearth = Rearth ; par%rho_bergs = rho_bergs ; par%lat_ref = lat_ref
    par%cdrag_grounding = cdrag_grounding ; par%h_to_init_grounding = h_to_init_grounding ; par%ocean_drag_scale = ocean_drag_scale
    par%speed_limit = speed_limit ; par%sicn_shift = sicn_shift ; par%bergy_bit_erosion_fraction = bergy_bit_erosion_fraction
    par%tip_parameter = tip_parameter ; par%grounding_fraction = grounding_fraction ; par%clipping_depth = 0.   ! FW:227, not in the namelist
    par%coastal_drift = coastal_drift ; par%tidal_drift = tidal_drift ; par%initial_orientation = initial_orientation
    par%melt_cutoff = melt_cutoff ; par%cdrag_icebergs = cdrag_icebergs ; par%utide_icebergs = utide_icebergs
    par%ustar_icebergs_bg = ustar_icebergs_bg ; par%Gamma_T_3EQ = Gamma_T_3EQ
    par%fl_youngs = fl_youngs ; par%fl_strength = fl_strength ; par%new_berg_from_fl_bits_mass_thres = new_berg_from_fl_bits_mass_thres
    par%u_override = u_override ; par%v_override = v_override
    par%initial_mass_s = initial_mass ; par%initial_mass_n = initial_mass_n
    par%spring_coef = spring_coef ; par%contact_spring_coef = contact_spring_coef ; par%contact_distance = contact_distance
    par%radial_damping_coef = radial_damping_coef ; par%tangental_damping_coef = tangental_damping_coef
    par%convergence_tolerance = convergence_tolerance ; par%constant_length = constant_length ; par%constant_width = constant_width
    par%dem_spring_coef = dem_spring_coef ; par%dem_damping_coef = dem_damping_coef ; par%poisson = poisson
    par%dem_tests_start_lon = 0. ; par%dem_tests_end_lon = 0.   ! dem_tests_init, kid_icebergs_init_bonds
    par%frac_thres_n = frac_thres_n * frac_thres_scaling ; par%frac_thres_t = frac_thres_t * frac_thres_scaling   ! FW:1355-1356
    par%Runge_not_Verlet = l2i(Runge_not_Verlet) ; par%use_new_predictive_corrective = l2i(use_new_predictive_corrective)
    par%old_interp_flds_order = l2i(.not. (mts .or. dem .or. footloose))   ! FW:1483
    par%old_bug_bilin = l2i(old_bug_bilin) ; par%use_f_plane = l2i(use_f_plane) ; par%use_operator_splitting = l2i(use_operator_splitting)
    par%add_weight_to_ocean = l2i(add_weight_to_ocean) ; par%time_average_weight = l2i(time_average_weight)
    par%use_old_spreading = l2i(use_old_spreading) ; par%hexagonal_icebergs = l2i(hexagonal_icebergs)
    par%allow_bergs_to_roll = l2i(allow_bergs_to_roll) ; par%use_updated_rolling_scheme = l2i(use_updated_rolling_scheme)
    par%set_melt_rates_to_zero = l2i(set_melt_rates_to_zero) ; par%use_mixed_melting = l2i(use_mixed_melting)
    par%melt_icebergs_as_ice_shelf = l2i(melt_icebergs_as_ice_shelf) ; par%Use_three_equation_model = l2i(Use_three_equation_model)
    par%use_mixed_layer_salinity_for_thermo = l2i(use_mixed_layer_salinity_for_thermo) ; par%const_gamma = l2i(const_gamma)
    par%apply_thickness_cutoff_to_bergs_melt = l2i(apply_thickness_cutoff_to_bergs_melt)
    par%apply_thickness_cutoff_to_gridded_melt = l2i(apply_thickness_cutoff_to_gridded_melt)
    par%Iceberg_melt_without_decay = l2i(Iceberg_melt_without_decay) ; par%find_melt_using_spread_mass = l2i(find_melt_using_spread_mass)
    par%override_iceberg_velocities = l2i(override_iceberg_velocities) ; par%iceberg_bonds_on = l2i(iceberg_bonds_on)
    par%internal_bergs_for_drag = l2i(internal_bergs_for_drag) ; par%dem = l2i(dem) ; par%mts = l2i(mts) ; par%footloose = l2i(footloose)
    select case (trim(fl_style))   ! FW:820
    case ('new_bergs') ; par%fl_style = KID_FL_STYLE_NEW_BERGS
    case ('fl_bits') ; par%fl_style = KID_FL_STYLE_FL_BITS
    case default ; error stop 'kid_icebergs_init: fl_style must be new_bergs or fl_bits'
    end select
    par%fl_bits_erosion_to_bergy_bits = l2i(fl_bits_erosion_to_bergy_bits) ; par%displace_fl_bergs = l2i(displace_fl_bergs)
    par%use_roundoff_fix = l2i(use_roundoff_fix) ; par%interactive_icebergs_on = l2i(interactive_icebergs_on)
    par%only_interactive_forces = l2i(only_interactive_forces) ; par%pass_fields_to_ocean_model = l2i(pass_fields_to_ocean_model)
    par%static_icebergs = l2i(Static_icebergs) ; par%old_bug_rotated_weights = l2i(old_bug_rotated_weights)
    par%mts_sub_steps = merge(mts_sub_steps, 1, mts) ; par%explicit_inner_mts = l2i(explicit_inner_mts .or. dem)   ! FW:1436
    par%force_convergence = l2i(force_convergence) ; par%critical_interaction_damping_on = l2i(critical_interaction_damping_on)
    par%tang_crit_int_damp_on = l2i(tang_crit_int_damp_on) ; par%scale_damping_by_pmag = l2i(scale_damping_by_pmag)
    par%constant_interaction_LW = l2i(constant_interaction_LW) ; par%ignore_tangential_force = l2i(ignore_tangential_force)
    select case (trim(fracture_criterion))   ! FW:800
    case ('none') ; par%fracture_criterion_stress = 0
    case ('stress') ; par%fracture_criterion_stress = 1
    case default ; error stop 'kid_icebergs_init: fracture_criterion must be none or stress (the path has no other criterion)'
    end select
    par%max_bonds = max_bonds
    if (break_bonds_on_sub_steps) then   ! FW:1441-1450
      if (use_broken_bonds_for_substep_contact .and. .not. (dem .and. iceberg_bonds_on)) &
        error stop 'kid_icebergs_init: use_broken_bonds_for_substep_contact requires dem and iceberg_bonds_on'
      par%use_broken_bonds_for_substep_contact = l2i(use_broken_bonds_for_substep_contact)
    else
      par%use_broken_bonds_for_substep_contact = 0
    endif
    par%break_bonds_on_sub_steps = l2i(break_bonds_on_sub_steps) ; par%short_step_mts_grounding = l2i(short_step_mts_grounding)
    par%use_grounding_torque = l2i(use_grounding_torque) ; par%radius_based_drag = l2i(radius_based_drag)
    par%orig_dem_moment_of_inertia = l2i(orig_dem_moment_of_inertia) ; par%rev_mind = l2i(rev_mind)
    par%rotate_icebergs_for_mass_spreading = l2i(rotate_icebergs_for_mass_spreading)
    par%diag_mask = 0                                   ! register_diag_field is the host model's: set bits and kid_set_params
    par%periodic_reentry = l2i(cyclic_x .and. Lx > 0.)  ! this handle owns the whole period (include/kid_types.h)
    par%fl_init_child_xy_by_pe = l2i(fl_init_child_xy_by_pe) ; par%fl_rng_seed = 0 ; par%dem_beam_test = dem_beam_test
    if (footloose .and. .not. use_operator_splitting) error stop 'kid_icebergs_init: use_operator_splitting must be true to use footloose'   ! FW:1477
    if (Runge_not_Verlet .and. (mts .or. dem .or. footloose)) error stop 'kid_icebergs_init: Runge_not_Verlet must be false to use MTS, DEM, or footloose'   ! FW:1485
    if (contact_distance > 0.) then   ! FW:1492-1519
      dx_dlon = 1. ; dy_dlat = 1.
      if (grid_is_latlon) dy_dlat = pi_180 * Rearth
      maxk = 0
      do j = jsd, jed ; do i = isd, ied
        if (grid_is_latlon) dx_dlon = pi_180 * Rearth * cos(g2(lat, i, j) * pi_180)
        k = 0
        do while ((k + i) < ied)
          k = k + 1
          ddx = (g2(lon, k + i, j) - g2(lon, i, j)) * dx_dlon
          if (k > maxk) maxk = k
          if (ddx >= contact_distance) exit
        enddo
      enddo ; enddo
      ddy = (g2(lat, isc, jsc + 1) - g2(lat, isc, jsc)) * dy_dlat
      par%contact_cells_lon = max(maxk, 1) ; par%contact_cells_lat = max(int(ceiling(contact_distance / ddy)), 1)
    else
      par%contact_cells_lon = 1 ; par%contact_cells_lat = 1
    endif
    if (.not. mts) then   ! FW:1522-1532
      if ((halo - 1) < par%contact_cells_lon .or. (halo - 1) < par%contact_cells_lat) &
        error stop 'kid_icebergs_init: halo width must be increased to accomodate specified contact distance'
    endif

    ! ---- the handle ----
    cap = 1048576 ; if (present(capacity)) cap = capacity
    dev = 0 ; if (present(device)) dev = device
    bergs%gd = gd ; bergs%par = par ; bergs%capacity = cap
    bergs%tau_is_velocity = tau_is_velocity ; bergs%passive_mode = passive_mode
    call kid_check(kid_create(gd, par, cap, int(dev, c_int), bergs%h), bergs%h, 'kid_create')
    do f = 1, KID_NGRID_STATIC ; pst(f) = c_loc(st(isd, jsd, f)) ; enddo
    call kid_check(kid_set_static_grid(bergs%h, pst), bergs%h, 'kid_set_static_grid')
    allocate(bergs%list(isd:ied, jsd:jed))
    allocate(bergs%area(ni, nj)) ; bergs%area = area
    allocate(bergs%static(ni, nj, KID_NGRID_STATIC)) ; bergs%static = st
    allocate(bergs%f64(cap, KID_NB_F64), bergs%i32(cap, KID_NB_I32), bergs%ids(cap))
    allocate(bergs%acc(ni, nj, KID_NACC), bergs%outp(ni, nj, KID_NOUT), bergs%scal(KID_NSCALAR), bergs%gcalv(ni, nj), bergs%ghflx(ni, nj))
    ! class tables of the calving source (FW:1534-1551)
    cp%distribution_s = distribution ; cp%distribution_n = distribution_n
    cp%mass_scaling_s = mass_scaling ; cp%mass_scaling_n = mass_scaling_n
    cp%initial_thickness_s = initial_thickness ; cp%initial_thickness_n = initial_thickness_n
    cp%initial_width_s = sqrt(initial_mass / (LoW_ratio * rho_bergs * initial_thickness)) ; cp%initial_length_s = LoW_ratio * cp%initial_width_s
    cp%initial_width_n = sqrt(initial_mass_n / (LoW_ratio * rho_bergs * initial_thickness_n)) ; cp%initial_length_n = LoW_ratio * cp%initial_width_n
    cp%tau_calving = tau_calving ; cp%restarted = 0 ; cp%pad = 0
    call kid_glue_set_calving(bergs, cp)
    ! what record_posn keeps (FW:1329-1342)
    tp%traj_area_thres = traj_area_thres ; tp%traj_area_thres_sntbc = traj_area_thres_sntbc ; tp%traj_area_thres_fl = traj_area_thres_fl
    tp%save_all_traj_year = save_all_traj_year
    tp%save_traj_by_class_start_mass_thres_s = save_traj_by_class_start_mass_thres_s
    tp%save_traj_by_class_start_mass_thres_n = save_traj_by_class_start_mass_thres_n
    tp%save_short_traj = l2i(save_short_traj) ; tp%save_fl_traj = l2i(save_fl_traj)
    tp%save_nonfl_traj_by_class = l2i(save_nonfl_traj_by_class) ; tp%save_bond_traj = l2i(save_bond_traj)
    call kid_check(kid_set_traj_params(bergs%h, tp), bergs%h, 'kid_set_traj_params')

  contains

    integer(c_int32_t) function l2i(x)
      logical, intent(in) :: x
      l2i = merge(1, 0, x)
    end function l2i
    real(c_double) function g2(a, i, j)   ! a(i,j) in data-domain indices
      real(c_double), intent(in) :: a(:,:)
      integer, intent(in) :: i, j
      g2 = a(i - isd + 1, j - jsd + 1)
    end function g2
    subroutine s2(a, i, j, v)
      real(c_double), intent(inout) :: a(:,:)
      integer, intent(in) :: i, j
      real(c_double), intent(in) :: v
      a(i - isd + 1, j - jsd + 1) = v
    end subroutine s2
    subroutine put(a, src, i0, j0)   ! a(i0:i0+size-1, j0:...) = src
      real(c_double), intent(inout) :: a(:,:)
      real(c_double), intent(in) :: src(:,:)
      integer, intent(in) :: i0, j0
      a(i0 - isd + 1 : i0 - isd + size(src,1), j0 - jsd + 1 : j0 - jsd + size(src,2)) = src
    end subroutine put
    subroutine update_domains(a)
      real(c_double), intent(inout) :: a(:,:)
      integer :: ii, jj
      if (cyclic_x) then
        do jj = jsc, jec
          do ii = isd, isc - 1 ; a(ii - isd + 1, jj - jsd + 1) = a(ii + gni - isd + 1, jj - jsd + 1) ; enddo
          do ii = iec + 1, ied ; a(ii - isd + 1, jj - jsd + 1) = a(ii - gni - isd + 1, jj - jsd + 1) ; enddo
        enddo
      endif
      if (cyclic_y) then
        do ii = isd, ied
          if (.not. cyclic_x .and. (ii < isc .or. ii > iec)) cycle
          do jj = jsd, jsc - 1 ; a(ii - isd + 1, jj - jsd + 1) = a(ii - isd + 1, jj + gnj - jsd + 1) ; enddo
          do jj = jec + 1, jed ; a(ii - isd + 1, jj - jsd + 1) = a(ii - isd + 1, jj - gnj - jsd + 1) ; enddo
        enddo
      endif
    end subroutine update_domains
  end subroutine kid_icebergs_init

  !> apply_modulo_around_point (FW:6558-6573): x moved by whole periods Lx to within Lx/2 of y
  real(c_double) function apply_modulo_around_point(x, y, Lx) result(r)
    real(c_double), intent(in) :: x, y, Lx
    if (Lx > 0.) then
      r = modulo(x - (y - (Lx / 2.)), Lx) + (y - (Lx / 2.))
    else
      r = x
    endif
  end function apply_modulo_around_point

  !> the tail of icebergs_init once the population and its bonds are in the lists (IB:141-176): n_bonds of every berg
  !! (assign_n_bonds, FW:5226-5252), the DEM beam tests' start positions (dem_tests_init, FW:4687-4710) and the mean element
  !! size of constant_interaction_LW (set_constant_interaction_length_and_width, FW:4641-4684); parameters go to the device
  subroutine kid_icebergs_init_bonds(g)
    type(kid_glue), intent(inout) :: g
    type(iceberg), pointer :: this
    type(bond), pointer :: b
    integer :: grdi, grdj
    real(c_double) :: minlon, maxlon, elem_sum, l_sum, w_sum
    maxlon = -huge(1.0_c_double) ; minlon = huge(1.0_c_double) ; elem_sum = 0. ; l_sum = 0. ; w_sum = 0.
    do grdj = g%gd%jsd, g%gd%jed ; do grdi = g%gd%isd, g%gd%ied
      this => g%list(grdi,grdj)%first
      do while (associated(this))
        if (g%par%iceberg_bonds_on /= 0) then
          this%n_bonds = 0
          b => this%first_bond
          do while (associated(b)) ; this%n_bonds = this%n_bonds + 1 ; b => b%next_bond ; enddo
        endif
        if (g%par%dem_beam_test > 0) then
          this%start_lon = this%lon ; this%start_lat = this%lat
          if (this%lon > maxlon) maxlon = this%lon
          if (this%lon < minlon) minlon = this%lon
        endif
        if (grdi >= g%gd%isc .and. grdi <= g%gd%iec .and. grdj >= g%gd%jsc .and. grdj <= g%gd%jec) then
          elem_sum = elem_sum + 1. ; l_sum = l_sum + this%length ; w_sum = w_sum + this%width
        endif
        this => this%next
      enddo
    enddo ; enddo
    if (g%par%dem_beam_test > 0) then ; g%par%dem_tests_start_lon = minlon ; g%par%dem_tests_end_lon = maxlon ; endif
    if (g%par%constant_interaction_LW /= 0 .and. (g%par%constant_length == 0. .or. g%par%constant_width == 0.) .and. elem_sum > 0.) then   ! IB:173-175
      g%par%constant_length = l_sum / elem_sum ; g%par%constant_width = w_sum / elem_sum
    endif
    call kid_check(kid_set_params(g%h, g%par), g%h, 'kid_set_params')
  end subroutine kid_icebergs_init_bonds

  !> the class tables of ice_bergs_framework_init (FW:1534-1551) for the calving source (accumulate_calving, calve_icebergs)
  subroutine kid_glue_set_calving(g, cp)
    type(kid_glue), intent(inout) :: g
    type(kid_calving_params), intent(in) :: cp
    call kid_check(kid_set_calving_params(g%h, cp), g%h, 'kid_set_calving_params')
    g%calving_on = .true.
  end subroutine kid_glue_set_calving

  !> add_new_berg_to_list (FW:4014): a copy of `vals` becomes a node of the list of its cell
  subroutine kid_glue_add_berg(g, vals, newberg_return)
    type(kid_glue), intent(inout) :: g
    type(iceberg), intent(in) :: vals
    type(iceberg), pointer, optional :: newberg_return
    type(iceberg), pointer :: new
    allocate(new)
    new = vals
    new%prev => null() ; new%next => null() ; new%first_bond => null()
    call insert_berg_into_list(g%list(new%ine, new%jne)%first, new)
    if (present(newberg_return)) newberg_return => new
  end subroutine kid_glue_add_berg

  !> form_a_bond (FW:4818-4883): the new bond goes to the HEAD of the berg's list
  subroutine form_a_bond(berg, other_id, other_berg_ine, other_berg_jne, other_berg)
    type(iceberg), pointer :: berg
    integer(c_int64_t), intent(in) :: other_id
    integer, optional, intent(in) :: other_berg_ine, other_berg_jne
    type(iceberg), pointer, optional :: other_berg
    type(bond), pointer :: new_bond
    if (berg%id == other_id) return
    allocate(new_bond)
    new_bond%other_id = other_id
    if (present(other_berg)) then
      new_bond%other_berg => other_berg
      if (associated(other_berg)) then ; new_bond%other_berg_ine = other_berg%ine ; new_bond%other_berg_jne = other_berg%jne ; endif
    else
      if (present(other_berg_ine)) new_bond%other_berg_ine = other_berg_ine
      if (present(other_berg_jne)) new_bond%other_berg_jne = other_berg_jne
    endif
    new_bond%next_bond => berg%first_bond ; new_bond%prev_bond => null()
    if (associated(berg%first_bond)) berg%first_bond%prev_bond => new_bond
    berg%first_bond => new_bond
  end subroutine form_a_bond

  !> the node with this id (a scan of the lists: what connect_all_bonds does around the bond's recorded cell, FW:4963-5125)
  function kid_glue_find_berg(g, id) result(this)
    type(kid_glue), intent(in) :: g
    integer(c_int64_t), intent(in) :: id
    type(iceberg), pointer :: this
    integer :: grdi, grdj
    do grdj = g%gd%jsd, g%gd%jed ; do grdi = g%gd%isd, g%gd%ied
      this => g%list(grdi,grdj)%first
      do while (associated(this))
        if (this%id == id) return
        this => this%next
      enddo
    enddo ; enddo
    this => null()
  end function kid_glue_find_berg

  subroutine delete_bonds(berg)
    type(iceberg), pointer :: berg
    type(bond), pointer :: b, nxt
    b => berg%first_bond
    do while (associated(b)) ; nxt => b%next_bond ; deallocate(b) ; b => nxt ; enddo
    berg%first_bond => null()
  end subroutine delete_bonds

  integer(c_int64_t) function kid_glue_count(g) result(n)   ! count_bergs FW:5292 over the computational domain
    type(kid_glue), intent(in) :: g
    type(iceberg), pointer :: this
    integer :: grdi, grdj
    n = 0
    do grdj = g%gd%jsc, g%gd%jec ; do grdi = g%gd%isc, g%gd%iec
      this => g%list(grdi,grdj)%first
      do while (associated(this)) ; n = n + 1 ; this => this%next ; enddo
    enddo ; enddo
  end function kid_glue_count

  subroutine kid_glue_clear_lists(g)
    type(kid_glue), intent(inout) :: g
    type(iceberg), pointer :: this, nxt
    integer :: grdi, grdj
    do grdj = g%gd%jsd, g%gd%jed ; do grdi = g%gd%isd, g%gd%ied
      this => g%list(grdi,grdj)%first
      do while (associated(this)) ; nxt => this%next ; call delete_bonds(this) ; deallocate(this) ; this => nxt ; enddo
      g%list(grdi,grdj)%first => null()
    enddo ; enddo
  end subroutine kid_glue_clear_lists

  !> lists -> structure of arrays in the reference's traversal order (cells j outer / i inner, IB:7106; list order inside a
  !! cell), and upload.  Row k of every column is the k-th berg the reference's loops would visit.
  subroutine kid_glue_flatten(g)
    type(kid_glue), intent(inout), target :: g
    type(iceberg), pointer :: this
    type(kid_berg_soa) :: soa
    integer :: grdi, grdj, k
    integer(c_int64_t) :: n
    n = kid_glue_count(g)
    if (n > g%capacity) error stop 'kid_glue_flatten: more bergs than the handle has rows for'
    g%f64 = 0. ; g%i32 = 0 ; g%ids = 0
    n = 0
    do grdj = g%gd%jsc, g%gd%jec ; do grdi = g%gd%isc, g%gd%iec
      this => g%list(grdi,grdj)%first
      do while (associated(this))
        n = n + 1
        call node_to_row(this, g, n)
        this => this%next
      enddo
    enddo ; enddo
    soa%n = n
    do k = 1, KID_NB_F64 ; soa%f64(k) = c_loc(g%f64(1,k)) ; enddo
    do k = 1, KID_NB_I32 ; soa%i32(k) = c_loc(g%i32(1,k)) ; enddo
    soa%id = c_loc(g%ids(1))
    call kid_check(kid_upload_bergs(g%h, soa), g%h, 'kid_upload_bergs')
    if (g%par%iceberg_bonds_on /= 0 .and. g%par%max_bonds > 0) call flatten_bonds(g, n)
  end subroutine kid_glue_flatten

  !> the bond lists in list order: slot s of row k is the s-th bond of the k-th berg of the traversal
  subroutine flatten_bonds(g, n)
    type(kid_glue), intent(inout), target :: g
    integer(c_int64_t), intent(in) :: n
    type(iceberg), pointer :: this
    type(bond), pointer :: b
    integer :: grdi, grdj, s, mb
    integer(c_int64_t) :: k
    mb = g%par%max_bonds
    if (.not. allocated(g%bcount)) then
      allocate(g%bcount(g%capacity), g%bbroken(g%capacity, mb), g%bother(g%capacity, mb), g%bf64(g%capacity, mb, KID_NBOND_F64))
    endif
    g%bcount = 0 ; g%bbroken = 0 ; g%bother = 0 ; g%bf64 = 0.
    k = 0
    do grdj = g%gd%jsc, g%gd%jec ; do grdi = g%gd%isc, g%gd%iec
      this => g%list(grdi,grdj)%first
      do while (associated(this))
        k = k + 1
        s = 0
        b => this%first_bond
        do while (associated(b))
          s = s + 1
          if (s > mb) error stop 'kid_glue_flatten: a berg has more bonds than max_bonds'
          g%bother(k, s) = b%other_id ; g%bbroken(k, s) = b%broken
          g%bf64(k, s, KID_BOND_LENGTH+1) = b%length ; g%bf64(k, s, KID_BOND_TANGD1+1) = b%tangd1 ; g%bf64(k, s, KID_BOND_TANGD2+1) = b%tangd2
          g%bf64(k, s, KID_BOND_NSTRESS+1) = b%nstress ; g%bf64(k, s, KID_BOND_SSTRESS+1) = b%sstress
          g%bf64(k, s, KID_BOND_REL_ROTATION+1) = b%rel_rotation
          g%bf64(k, s, KID_BOND_F_X+1) = b%F_x ; g%bf64(k, s, KID_BOND_F_Y+1) = b%F_y ; g%bf64(k, s, KID_BOND_FD_X+1) = b%Fd_x
          g%bf64(k, s, KID_BOND_FD_Y+1) = b%Fd_y ; g%bf64(k, s, KID_BOND_T+1) = b%T ; g%bf64(k, s, KID_BOND_T_D+1) = b%T_d
          b => b%next_bond
        enddo
        g%bcount(k) = s
        this => this%next
      enddo
    enddo ; enddo
    ! the library's tables are slot-major over n rows (element s*n + k): hand it contiguous (n, mb) copies
    call upload_bond_tables(g, n, mb)
  end subroutine flatten_bonds

  subroutine upload_bond_tables(g, n, mb)
    type(kid_glue), intent(inout), target :: g
    integer(c_int64_t), intent(in) :: n
    integer, intent(in) :: mb
    type(kid_bond_soa) :: bs
    integer(c_int32_t), allocatable, target :: cnt(:), brk(:,:)
    integer(c_int64_t), allocatable, target :: oth(:,:)
    real(c_double), allocatable, target :: f64(:,:,:)
    integer :: f
    allocate(cnt(n), brk(n, mb), oth(n, mb), f64(n, mb, KID_NBOND_F64))
    cnt = g%bcount(1:n) ; brk = g%bbroken(1:n, :) ; oth = g%bother(1:n, :) ; f64 = g%bf64(1:n, :, :)
    bs%n = n ; bs%max_bonds = mb ; bs%pad = 0
    bs%count = c_loc(cnt) ; bs%other_id = c_loc(oth) ; bs%broken = c_loc(brk)
    do f = 1, KID_NBOND_F64 ; bs%f64(f) = c_loc(f64(1,1,f)) ; enddo
    call kid_check(kid_upload_bonds(g%h, bs), g%h, 'kid_upload_bonds')
  end subroutine upload_bond_tables

  !> structure of arrays -> lists: the surviving rows become nodes again, inserted in order (what unpack_berg_from_buffer2 /
  !! add_new_berg_to_list do for a berg that arrives from another PE, FW:3468, 4014); bergs the step removed are gone
  subroutine kid_glue_unflatten(g)
    type(kid_glue), intent(inout), target :: g
    type(kid_berg_soa) :: soa
    type(iceberg) :: vals
    integer(c_int64_t) :: n_slots, n_alive, k
    integer :: q
    call kid_check(kid_num_bergs(g%h, n_slots, n_alive), g%h, 'kid_num_bergs')
    if (n_slots > g%capacity) error stop 'kid_glue_unflatten: the population outgrew the staging arrays'
    soa%n = n_slots
    do q = 1, KID_NB_F64 ; soa%f64(q) = c_loc(g%f64(1,q)) ; enddo
    do q = 1, KID_NB_I32 ; soa%i32(q) = c_loc(g%i32(1,q)) ; enddo
    soa%id = c_loc(g%ids(1))
    call kid_check(kid_download_bergs(g%h, soa), g%h, 'kid_download_bergs')
    call kid_glue_clear_lists(g)
    if (g%par%iceberg_bonds_on /= 0 .and. g%par%max_bonds > 0) then
      call unflatten_with_bonds(g, n_slots)
      return
    endif
    do k = 1, n_slots
      if (g%i32(k, KID_BI_ALIVE+1) == 0) cycle
      call row_to_node(g, k, vals)
      call kid_glue_add_berg(g, vals)
    enddo
  end subroutine kid_glue_unflatten

  !> nodes and their bond lists from the rows and the bond tables: a berg's slots are put back with form_a_bond from the last to
  !! the first (each goes to the head: the list comes out in slot order), then connect_all_bonds (FW:4963-5125) by id
  subroutine unflatten_with_bonds(g, n_slots)
    type(kid_glue), intent(inout), target :: g
    integer(c_int64_t), intent(in) :: n_slots
    type(kid_bond_soa) :: bs
    type(iceberg) :: vals
    type(iceberg), pointer :: node
    type(bond), pointer :: b
    type node_ptr ; type(iceberg), pointer :: p => null() ; end type node_ptr
    type(node_ptr), allocatable :: nodes(:)
    integer(c_int32_t), allocatable, target :: cnt(:), brk(:,:)
    integer(c_int64_t), allocatable, target :: oth(:,:)
    real(c_double), allocatable, target :: f64(:,:,:)
    integer(c_int64_t), allocatable :: order(:)
    integer(c_int64_t) :: k, lo, hi, mid, want
    integer :: s, f, mb
    mb = g%par%max_bonds
    allocate(cnt(n_slots), brk(n_slots, mb), oth(n_slots, mb), f64(n_slots, mb, KID_NBOND_F64), nodes(n_slots), order(n_slots))
    bs%n = n_slots ; bs%max_bonds = mb ; bs%pad = 0
    bs%count = c_loc(cnt) ; bs%other_id = c_loc(oth) ; bs%broken = c_loc(brk)
    do f = 1, KID_NBOND_F64 ; bs%f64(f) = c_loc(f64(1,1,f)) ; enddo
    call kid_check(kid_download_bonds(g%h, bs), g%h, 'kid_download_bonds')
    do k = 1, n_slots
      if (g%i32(k, KID_BI_ALIVE+1) == 0) cycle
      call row_to_node(g, k, vals)
      call kid_glue_add_berg(g, vals, node)
      nodes(k)%p => node
      do s = cnt(k), 1, -1
        call form_a_bond(node, oth(k, s))
        b => node%first_bond
        b%broken = brk(k, s)
        b%length = f64(k, s, KID_BOND_LENGTH+1) ; b%tangd1 = f64(k, s, KID_BOND_TANGD1+1) ; b%tangd2 = f64(k, s, KID_BOND_TANGD2+1)
        b%nstress = f64(k, s, KID_BOND_NSTRESS+1) ; b%sstress = f64(k, s, KID_BOND_SSTRESS+1) ; b%rel_rotation = f64(k, s, KID_BOND_REL_ROTATION+1)
        b%F_x = f64(k, s, KID_BOND_F_X+1) ; b%F_y = f64(k, s, KID_BOND_F_Y+1) ; b%Fd_x = f64(k, s, KID_BOND_FD_X+1)
        b%Fd_y = f64(k, s, KID_BOND_FD_Y+1) ; b%T = f64(k, s, KID_BOND_T+1) ; b%T_d = f64(k, s, KID_BOND_T_D+1)
      enddo
    enddo
    ! connect_all_bonds: other_berg by id (rows sorted by id, binary search)
    call argsort_ids(g%ids, n_slots, order)
    do k = 1, n_slots
      if (.not. associated(nodes(k)%p)) cycle
      b => nodes(k)%p%first_bond
      do while (associated(b))
        want = b%other_id ; lo = 1 ; hi = n_slots
        do while (lo <= hi)
          mid = (lo + hi) / 2
          if (g%ids(order(mid)) < want) then ; lo = mid + 1
          else if (g%ids(order(mid)) > want) then ; hi = mid - 1
          else
            if (associated(nodes(order(mid))%p)) then
              b%other_berg => nodes(order(mid))%p
              b%other_berg_ine = b%other_berg%ine ; b%other_berg_jne = b%other_berg%jne
            endif
            exit
          endif
        enddo
        b => b%next_bond
      enddo
    enddo
  end subroutine unflatten_with_bonds

  subroutine argsort_ids(ids, n, order)   ! bottom-up merge sort of 1..n by ids
    integer(c_int64_t), intent(in) :: ids(:), n
    integer(c_int64_t), intent(out) :: order(:)
    integer(c_int64_t), allocatable :: tmp(:)
    integer(c_int64_t) :: w, lo, mid, hi, i, j, k
    allocate(tmp(n))
    do i = 1, n ; order(i) = i ; enddo
    w = 1
    do while (w < n)
      lo = 1
      do while (lo <= n)
        mid = min(lo + w, n + 1) ; hi = min(lo + 2 * w, n + 1)
        i = lo ; j = mid ; k = lo
        do while (i < mid .or. j < hi)
          if (j >= hi) then ; tmp(k) = order(i) ; i = i + 1
          else if (i >= mid) then ; tmp(k) = order(j) ; j = j + 1
          else if (ids(order(i)) <= ids(order(j))) then ; tmp(k) = order(i) ; i = i + 1
          else ; tmp(k) = order(j) ; j = j + 1
          endif
          k = k + 1
        enddo
        lo = lo + 2 * w
      enddo
      order(1:n) = tmp(1:n)
      w = 2 * w
    enddo
  end subroutine argsort_ids

  subroutine node_to_row(b, g, n)
    type(iceberg), pointer :: b
    type(kid_glue), intent(inout) :: g
    integer(c_int64_t), intent(in) :: n
    g%f64(n, KID_B_LON+1) = b%lon ; g%f64(n, KID_B_LAT+1) = b%lat ; g%f64(n, KID_B_UVEL+1) = b%uvel ; g%f64(n, KID_B_VVEL+1) = b%vvel
    g%f64(n, KID_B_MASS+1) = b%mass ; g%f64(n, KID_B_THICKNESS+1) = b%thickness ; g%f64(n, KID_B_WIDTH+1) = b%width
    g%f64(n, KID_B_LENGTH+1) = b%length ; g%f64(n, KID_B_START_LON+1) = b%start_lon ; g%f64(n, KID_B_START_LAT+1) = b%start_lat
    g%f64(n, KID_B_START_DAY+1) = b%start_day ; g%f64(n, KID_B_START_MASS+1) = b%start_mass ; g%f64(n, KID_B_MASS_SCALING+1) = b%mass_scaling
    g%f64(n, KID_B_MASS_OF_BITS+1) = b%mass_of_bits ; g%f64(n, KID_B_MASS_OF_FL_BITS+1) = b%mass_of_fl_bits
    g%f64(n, KID_B_MASS_OF_FL_BERGY_BITS+1) = b%mass_of_fl_bergy_bits ; g%f64(n, KID_B_FL_K+1) = b%fl_k
    g%f64(n, KID_B_HEAT_DENSITY+1) = b%heat_density ; g%f64(n, KID_B_XI+1) = b%xi ; g%f64(n, KID_B_YJ+1) = b%yj
    g%f64(n, KID_B_UO+1) = b%uo ; g%f64(n, KID_B_VO+1) = b%vo ; g%f64(n, KID_B_UI+1) = b%ui ; g%f64(n, KID_B_VI+1) = b%vi
    g%f64(n, KID_B_UA+1) = b%ua ; g%f64(n, KID_B_VA+1) = b%va ; g%f64(n, KID_B_SSH_X+1) = b%ssh_x ; g%f64(n, KID_B_SSH_Y+1) = b%ssh_y
    g%f64(n, KID_B_SST+1) = b%sst ; g%f64(n, KID_B_SSS+1) = b%sss ; g%f64(n, KID_B_CN+1) = b%cn ; g%f64(n, KID_B_HI+1) = b%hi
    g%f64(n, KID_B_OD+1) = b%od ; g%f64(n, KID_B_AXN+1) = b%axn ; g%f64(n, KID_B_AYN+1) = b%ayn ; g%f64(n, KID_B_BXN+1) = b%bxn
    g%f64(n, KID_B_BYN+1) = b%byn ; g%f64(n, KID_B_UVEL_PREV+1) = b%uvel_prev ; g%f64(n, KID_B_VVEL_PREV+1) = b%vvel_prev
    g%f64(n, KID_B_UVEL_OLD+1) = b%uvel_old ; g%f64(n, KID_B_VVEL_OLD+1) = b%vvel_old ; g%f64(n, KID_B_LON_OLD+1) = b%lon_old
    g%f64(n, KID_B_LAT_OLD+1) = b%lat_old ; g%f64(n, KID_B_HALO_BERG+1) = b%halo_berg ; g%f64(n, KID_B_STATIC_BERG+1) = b%static_berg
    g%f64(n, KID_B_AXN_FAST+1) = b%axn_fast ; g%f64(n, KID_B_AYN_FAST+1) = b%ayn_fast ; g%f64(n, KID_B_BXN_FAST+1) = b%bxn_fast
    g%f64(n, KID_B_BYN_FAST+1) = b%byn_fast ; g%f64(n, KID_B_ANG_VEL+1) = b%ang_vel ; g%f64(n, KID_B_ANG_ACCEL+1) = b%ang_accel
    g%f64(n, KID_B_ROT+1) = b%rot
    g%i32(n, KID_BI_INE+1) = b%ine ; g%i32(n, KID_BI_JNE+1) = b%jne ; g%i32(n, KID_BI_START_YEAR+1) = b%start_year
    g%i32(n, KID_BI_N_BONDS+1) = b%n_bonds ; g%i32(n, KID_BI_CONGLOM_ID+1) = b%conglom_id
    g%i32(n, KID_BI_ALIVE+1) = 1
    g%ids(n) = b%id
  end subroutine node_to_row

  subroutine row_to_node(g, n, b)
    type(kid_glue), intent(in) :: g
    integer(c_int64_t), intent(in) :: n
    type(iceberg), intent(out) :: b
    b%lon = g%f64(n, KID_B_LON+1) ; b%lat = g%f64(n, KID_B_LAT+1) ; b%uvel = g%f64(n, KID_B_UVEL+1) ; b%vvel = g%f64(n, KID_B_VVEL+1)
    b%mass = g%f64(n, KID_B_MASS+1) ; b%thickness = g%f64(n, KID_B_THICKNESS+1) ; b%width = g%f64(n, KID_B_WIDTH+1)
    b%length = g%f64(n, KID_B_LENGTH+1) ; b%start_lon = g%f64(n, KID_B_START_LON+1) ; b%start_lat = g%f64(n, KID_B_START_LAT+1)
    b%start_day = g%f64(n, KID_B_START_DAY+1) ; b%start_mass = g%f64(n, KID_B_START_MASS+1) ; b%mass_scaling = g%f64(n, KID_B_MASS_SCALING+1)
    b%mass_of_bits = g%f64(n, KID_B_MASS_OF_BITS+1) ; b%mass_of_fl_bits = g%f64(n, KID_B_MASS_OF_FL_BITS+1)
    b%mass_of_fl_bergy_bits = g%f64(n, KID_B_MASS_OF_FL_BERGY_BITS+1) ; b%fl_k = g%f64(n, KID_B_FL_K+1)
    b%heat_density = g%f64(n, KID_B_HEAT_DENSITY+1) ; b%xi = g%f64(n, KID_B_XI+1) ; b%yj = g%f64(n, KID_B_YJ+1)
    b%uo = g%f64(n, KID_B_UO+1) ; b%vo = g%f64(n, KID_B_VO+1) ; b%ui = g%f64(n, KID_B_UI+1) ; b%vi = g%f64(n, KID_B_VI+1)
    b%ua = g%f64(n, KID_B_UA+1) ; b%va = g%f64(n, KID_B_VA+1) ; b%ssh_x = g%f64(n, KID_B_SSH_X+1) ; b%ssh_y = g%f64(n, KID_B_SSH_Y+1)
    b%sst = g%f64(n, KID_B_SST+1) ; b%sss = g%f64(n, KID_B_SSS+1) ; b%cn = g%f64(n, KID_B_CN+1) ; b%hi = g%f64(n, KID_B_HI+1)
    b%od = g%f64(n, KID_B_OD+1) ; b%axn = g%f64(n, KID_B_AXN+1) ; b%ayn = g%f64(n, KID_B_AYN+1) ; b%bxn = g%f64(n, KID_B_BXN+1)
    b%byn = g%f64(n, KID_B_BYN+1) ; b%uvel_prev = g%f64(n, KID_B_UVEL_PREV+1) ; b%vvel_prev = g%f64(n, KID_B_VVEL_PREV+1)
    b%uvel_old = g%f64(n, KID_B_UVEL_OLD+1) ; b%vvel_old = g%f64(n, KID_B_VVEL_OLD+1) ; b%lon_old = g%f64(n, KID_B_LON_OLD+1)
    b%lat_old = g%f64(n, KID_B_LAT_OLD+1) ; b%halo_berg = g%f64(n, KID_B_HALO_BERG+1) ; b%static_berg = g%f64(n, KID_B_STATIC_BERG+1)
    b%axn_fast = g%f64(n, KID_B_AXN_FAST+1) ; b%ayn_fast = g%f64(n, KID_B_AYN_FAST+1) ; b%bxn_fast = g%f64(n, KID_B_BXN_FAST+1)
    b%byn_fast = g%f64(n, KID_B_BYN_FAST+1) ; b%ang_vel = g%f64(n, KID_B_ANG_VEL+1) ; b%ang_accel = g%f64(n, KID_B_ANG_ACCEL+1)
    b%rot = g%f64(n, KID_B_ROT+1)
    b%ine = g%i32(n, KID_BI_INE+1) ; b%jne = g%i32(n, KID_BI_JNE+1) ; b%start_year = g%i32(n, KID_BI_START_YEAR+1)
    b%n_bonds = g%i32(n, KID_BI_N_BONDS+1) ; b%conglom_id = g%i32(n, KID_BI_CONGLOM_ID+1)
    b%id = g%ids(n)
  end subroutine row_to_node

  !> One coupling step behind the argument list of icebergs_run (IB:5074-5096).  `bergs` is the glue object; `time` is the
  !! (year, yearday) the reference reads out of FMS's time_type (IB:5169-5175).  The arrays have the extents the coupler
  !! gives them (compute domain for calving, calving_hflx, sst, sss and the three optional returns; compute + 1 halo for the
  !! rest, shifted by the stagger -- DRV:386-392).  The bergs are resident on the device between calls; flatten before the
  !! first call and whenever the lists were changed by the host, unflatten where the host needs the lists.
  subroutine kid_icebergs_run(bergs, year, yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, &
                              stagger, stress_stagger, sss, mass_berg, ustar_berg, area_berg, exchange)
    type(kid_glue), intent(inout), target :: bergs
    integer, intent(in) :: year
    real(c_double), intent(in) :: yearday
    real(c_double), dimension(:,:), intent(inout), target :: calving, calving_hflx
    real(c_double), dimension(:,:), intent(in), target :: uo, vo, ui, vi, tauxa, tauya, ssh, sst, cn, hi
    integer, optional, intent(in) :: stagger, stress_stagger
    real(c_double), dimension(:,:), optional, intent(in), target :: sss
    real(c_double), dimension(:,:), optional, pointer :: mass_berg, ustar_berg, area_berg
    procedure(kid_exchange_sum), optional :: exchange   ! several GPUs on one grid: sums the block over the ranks (INTEGRATION.md section 6)
    type(c_ptr) :: dev
    integer(c_int64_t) :: total, live
    call kid_icebergs_run_local(bergs, year, yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, stagger, stress_stagger, sss)
    if (present(exchange)) then
      call kid_check(kid_sync(bergs%h), bergs%h, 'kid_sync')   ! the local sums are complete before another library touches them
      call kid_check(kid_accum_device_ptr(bergs%h, dev, total), bergs%h, 'kid_accum_device_ptr')
      call kid_check(kid_accum_live_count(bergs%h, live), bergs%h, 'kid_accum_live_count')
      call exchange(dev, live)
    endif
    call kid_icebergs_run_finish(bergs, calving, calving_hflx, mass_berg, ustar_berg, area_berg)
  end subroutine kid_icebergs_run

  !> first half of kid_icebergs_run: clock, forcing ingest, calving source, the per-berg work of this GPU's bergs
  !! (kid_step_local).  The accumulator block then holds this GPU's sums only.
  subroutine kid_icebergs_run_local(bergs, year, yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, &
                                    stagger, stress_stagger, sss)
    type(kid_glue), intent(inout), target :: bergs
    integer, intent(in) :: year
    real(c_double), intent(in) :: yearday
    real(c_double), dimension(:,:), intent(inout), target :: calving, calving_hflx
    real(c_double), dimension(:,:), intent(in), target :: uo, vo, ui, vi, tauxa, tauya, ssh, sst, cn, hi
    integer, optional, intent(in) :: stagger, stress_stagger
    real(c_double), dimension(:,:), optional, intent(in), target :: sss
    type(kid_forcing_in) :: fi
    type(kid_calving_in) :: ci
    real(c_double), target :: cscal(KID_NCALV_SCALARS)
    integer :: vel_stagger, str_stagger
    vel_stagger = KID_BGRID_NE ; if (present(stagger)) vel_stagger = stagger                ! IB:5121-5122
    str_stagger = vel_stagger ; if (present(stress_stagger)) str_stagger = stress_stagger
    ! the model clock (bergs%current_year / current_yearday, IB:5173-5175)
    bergs%par%current_year = year ; bergs%par%current_yearday = yearday
    call kid_check(kid_set_params(bergs%h, bergs%par), bergs%h, 'kid_set_params')
    ! forcing ingest, IB:5236-5383
    fi%uo = c_loc(uo) ; fi%vo = c_loc(vo) ; fi%ui = c_loc(ui) ; fi%vi = c_loc(vi)
    fi%tauxa = c_loc(tauxa) ; fi%tauya = c_loc(tauya)
    fi%ssh = c_loc(ssh) ; fi%sst = c_loc(sst) ; fi%cn = c_loc(cn) ; fi%hi = c_loc(hi)
    fi%sss = c_null_ptr ; if (present(sss)) fi%sss = c_loc(sss)
    fi%u_ni = size(uo,1) ; fi%u_nj = size(uo,2) ; fi%v_ni = size(vo,1) ; fi%v_nj = size(vo,2)
    fi%taux_ni = size(tauxa,1) ; fi%taux_nj = size(tauxa,2) ; fi%tauy_ni = size(tauya,1) ; fi%tauy_nj = size(tauya,2)
    fi%vel_stagger = vel_stagger ; fi%stress_stagger = str_stagger
    fi%tau_is_velocity = merge(1, 0, bergs%tau_is_velocity) ; fi%cyclic_x = merge(1, 0, bergs%gd%Lx > 0.) ; fi%on_device = 0 ; fi%pad = 0
    call kid_check(kid_ingest_forcing(bergs%h, fi), bergs%h, 'kid_ingest_forcing')
    ! calving source, IB:5203-5231, 5388, 5403
    if (bergs%calving_on) then
      ci%calving = c_loc(calving) ; ci%calving_hflx = c_loc(calving_hflx) ; ci%on_device = 0 ; ci%pad = 0
      call kid_check(kid_calving(bergs%h, ci, cscal), bergs%h, 'kid_calving')
    endif
    ! the hot path, IB:5423-5512: this GPU's bergs
    call kid_check(kid_step_local(bergs%h), bergs%h, 'kid_step_local')
  end subroutine kid_icebergs_run_local

  !> second half: the 9-point gather and the derived fields from the (summed) block, then what goes back to the coupler
  subroutine kid_icebergs_run_finish(bergs, calving, calving_hflx, mass_berg, ustar_berg, area_berg)
    type(kid_glue), intent(inout), target :: bergs
    real(c_double), dimension(:,:), intent(inout), target :: calving, calving_hflx
    real(c_double), dimension(:,:), optional, pointer :: mass_berg, ustar_berg, area_berg
    integer :: i0, j0, i1, j1
    call kid_check(kid_step_gather(bergs%h), bergs%h, 'kid_step_gather')
    ! move_berg_between_cells (IB:5437), amortised as kid_run_step does it; bonded bergs keep their rows
    bergs%since_sort = bergs%since_sort + 1
    if (bergs%par%mts == 0 .and. bergs%par%interactive_icebergs_on == 0 .and. bergs%resort_interval > 0 .and. bergs%since_sort >= bergs%resort_interval) then
      call kid_check(kid_move_berg_between_cells(bergs%h), bergs%h, 'kid_move_berg_between_cells')
      bergs%since_sort = 0
    endif
    call kid_check(kid_get_accumulators(bergs%h, c_loc(bergs%acc), c_loc(bergs%outp), c_loc(bergs%scal)), bergs%h, 'kid_get_accumulators')
    ! what goes back to the coupler, IB:5654-5679 (not in passive_mode): unused calving + melt, the heat flux, the optional fields
    i0 = bergs%gd%isc - bergs%gd%isd + 1 ; i1 = bergs%gd%iec - bergs%gd%isd + 1
    j0 = bergs%gd%jsc - bergs%gd%jsd + 1 ; j1 = bergs%gd%jec - bergs%gd%jsd + 1
    bergs%gcalv = 0. ; bergs%ghflx = 0.
    if (bergs%calving_on) call kid_check(kid_get_calving(bergs%h, c_loc(bergs%gcalv), c_loc(bergs%ghflx)), bergs%h, 'kid_get_calving')
    if (bergs%passive_mode) return
    where (bergs%area(i0:i1, j0:j1) > 0.)
      calving(:,:) = bergs%gcalv(i0:i1, j0:j1) / bergs%area(i0:i1, j0:j1) + bergs%acc(i0:i1, j0:j1, KID_A_FLOATING_MELT+1)
    elsewhere
      calving(:,:) = 0.
    end where
    calving_hflx(:,:) = bergs%ghflx(i0:i1, j0:j1) + bergs%acc(i0:i1, j0:j1, KID_A_CALVING_HFLX+1)   ! grd%calving_hflx after the melt's share, IB:3129
    if (present(mass_berg)) then
      if (associated(mass_berg) .and. bergs%par%add_weight_to_ocean /= 0) mass_berg(:,:) = bergs%outp(i0:i1, j0:j1, KID_O_SPREAD_MASS+1)
    endif
    if (present(ustar_berg)) then
      if (associated(ustar_berg)) ustar_berg(:,:) = bergs%outp(i0:i1, j0:j1, KID_O_USTAR_ICEBERG+1)
    endif
    if (present(area_berg)) then
      if (associated(area_berg)) area_berg(:,:) = bergs%outp(i0:i1, j0:j1, KID_O_SPREAD_AREA+1)
    endif
  end subroutine kid_icebergs_run_finish

  subroutine kid_glue_end(g)   ! icebergs_end
    type(kid_glue), intent(inout) :: g
    call kid_glue_clear_lists(g)
    call kid_check(kid_destroy(g%h), g%h, 'kid_destroy')
    g%h = c_null_ptr
  end subroutine kid_glue_end
end module kid_icebergs_glue
