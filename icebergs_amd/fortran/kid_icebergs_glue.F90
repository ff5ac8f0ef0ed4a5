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
!!     the return of unused calving + melt to the coupler (IB:5654-5679).
!! tests/test_fortran_gpu.py::test_glue_* drive it through kid_glue_test.F90 against the oracle.
module kid_icebergs_glue
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  implicit none
  private
  public :: iceberg, linked_list, kid_glue, inorder, insert_berg_into_list, kid_glue_init, kid_glue_set_calving, kid_glue_add_berg, kid_glue_count, &
            kid_glue_flatten, kid_glue_unflatten, kid_glue_clear_lists, kid_glue_end, kid_icebergs_run

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
    integer :: start_year = 0, ine = 0, jne = 0
    integer(c_int64_t) :: id = 0
  end type iceberg

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
    integer(c_int64_t) :: capacity = 0
    logical :: calving_on = .false.        ! the calving source runs on the device too (kid_glue_set_calving)
    logical :: tau_is_velocity = .false.   ! bergs%tau_is_velocity (FW:727): tauxa / tauya are winds, not stresses
    logical :: passive_mode = .false.      ! bergs%passive_mode (FW:728): nothing is returned to the coupler
    real(c_double), allocatable :: f64(:,:)                    ! (capacity, KID_NB_F64): one column per member
    integer(c_int32_t), allocatable :: i32(:,:)                ! (capacity, KID_NB_I32)
    integer(c_int64_t), allocatable :: ids(:)
    real(c_double), allocatable :: acc(:,:,:), outp(:,:,:), scal(:), gcalv(:,:), ghflx(:,:)
  end type kid_glue

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

  !> the class tables of ice_bergs_framework_init (FW:1534-1551) for the calving source (accumulate_calving, calve_icebergs)
  subroutine kid_glue_set_calving(g, cp)
    type(kid_glue), intent(inout) :: g
    type(kid_calving_params), intent(in) :: cp
    call kid_check(kid_set_calving_params(g%h, cp), g%h, 'kid_set_calving_params')
    g%calving_on = .true.
  end subroutine kid_glue_set_calving

  !> add_new_berg_to_list (FW:4014): a copy of `vals` becomes a node of the list of its cell
  subroutine kid_glue_add_berg(g, vals)
    type(kid_glue), intent(inout) :: g
    type(iceberg), intent(in) :: vals
    type(iceberg), pointer :: new
    allocate(new)
    new = vals
    new%prev => null() ; new%next => null()
    call insert_berg_into_list(g%list(new%ine, new%jne)%first, new)
  end subroutine kid_glue_add_berg

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
      do while (associated(this)) ; nxt => this%next ; deallocate(this) ; this => nxt ; enddo
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
  end subroutine kid_glue_flatten

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
    do k = 1, n_slots
      if (g%i32(k, KID_BI_ALIVE+1) == 0) cycle
      call row_to_node(g, k, vals)
      call kid_glue_add_berg(g, vals)
    enddo
  end subroutine kid_glue_unflatten

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
    g%i32(n, KID_BI_INE+1) = b%ine ; g%i32(n, KID_BI_JNE+1) = b%jne ; g%i32(n, KID_BI_START_YEAR+1) = b%start_year
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
    b%ine = g%i32(n, KID_BI_INE+1) ; b%jne = g%i32(n, KID_BI_JNE+1) ; b%start_year = g%i32(n, KID_BI_START_YEAR+1)
    b%id = g%ids(n)
  end subroutine row_to_node

  !> One coupling step behind the argument list of icebergs_run (IB:5074-5096).  `bergs` is the glue object; `time` is the
  !! (year, yearday) the reference reads out of FMS's time_type (IB:5169-5175).  The arrays have the extents the coupler
  !! gives them (compute domain for calving, calving_hflx, sst, sss and the three optional returns; compute + 1 halo for the
  !! rest, shifted by the stagger -- DRV:386-392).  The bergs are resident on the device between calls; flatten before the
  !! first call and whenever the lists were changed by the host, unflatten where the host needs the lists.
  subroutine kid_icebergs_run(bergs, year, yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, &
                              stagger, stress_stagger, sss, mass_berg, ustar_berg, area_berg)
    type(kid_glue), intent(inout), target :: bergs
    integer, intent(in) :: year
    real(c_double), intent(in) :: yearday
    real(c_double), dimension(:,:), intent(inout), target :: calving, calving_hflx
    real(c_double), dimension(:,:), intent(in), target :: uo, vo, ui, vi, tauxa, tauya, ssh, sst, cn, hi
    integer, optional, intent(in) :: stagger, stress_stagger
    real(c_double), dimension(:,:), optional, intent(in), target :: sss
    real(c_double), dimension(:,:), optional, pointer :: mass_berg, ustar_berg, area_berg
    type(kid_forcing_in) :: fi
    type(kid_calving_in) :: ci
    real(c_double), target :: cscal(KID_NCALV_SCALARS)
    integer :: vel_stagger, str_stagger, i0, j0, i1, j1
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
    ! the hot path, IB:5423-5512
    call kid_check(kid_run_step(bergs%h, 1_c_int), bergs%h, 'kid_run_step')
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
  end subroutine kid_icebergs_run

  subroutine kid_glue_end(g)   ! icebergs_end
    type(kid_glue), intent(inout) :: g
    call kid_glue_clear_lists(g)
    call kid_check(kid_destroy(g%h), g%h, 'kid_destroy')
    g%h = c_null_ptr
  end subroutine kid_glue_end
end module kid_icebergs_glue
