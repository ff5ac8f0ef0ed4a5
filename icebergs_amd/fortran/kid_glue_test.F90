!> Self-contained test driver of kid_icebergs_glue: the reference's data structures on the host, the HIP library underneath.
!!
!! It builds per-cell linked lists of `iceberg` nodes from a population that arrives in shuffled order (sorted insertion,
!! `inorder`), flattens them in traversal order, steps them with kid_icebergs_run -- the argument list of icebergs_run --
!! and rebuilds the lists from the device.  Case file = the one kid_couple reads (see there) with magic 1263093764; written
!! and checked against the oracle by tests/test_fortran_gpu.py::test_glue_lists_and_icebergs_run.
!! Output: int64 n, the ids in flattening order; per call the calving and calving_hflx the wrapper handed back;
!! mass_berg of the last call; int64 m and the bergs of the rebuilt lists in traversal order (KID_NB_F64 columns, KID_NB_I32
!! columns, ids).
program kid_glue_test
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  use kid_icebergs_glue
  implicit none
  character(len=1024) :: fin, fout
  type(kid_glue), target :: bergs
  type(kid_grid_desc) :: gd
  type(kid_params) :: par
  type(kid_calving_params) :: cp
  type(iceberg) :: vals
  type(iceberg), pointer :: this
  integer(c_int32_t) :: magic, vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls, ext(8)
  integer(c_int64_t) :: n, capacity, m, k
  integer :: ni, nj, nic, njc, q, s, u, uo_, grdi, grdj
  real(c_double), allocatable, target :: gstatic(:,:,:), bf(:,:)
  real(c_double), allocatable, target :: uo(:,:), vo(:,:), ui(:,:), vi(:,:), tauxa(:,:), tauya(:,:), ssh(:,:), sst(:,:), &
      cn(:,:), hi(:,:), sss(:,:), calving(:,:), calving_hflx(:,:)
  real(c_double), pointer :: mass_berg(:,:), none(:,:)
  integer(c_int32_t), allocatable, target :: bi(:,:)
  integer(c_int64_t), allocatable, target :: bid(:)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', action='read')
  read(u) magic
  if (magic /= 1263093764) error stop 'kid_glue_test: bad magic'
  read(u) gd ; read(u) par ; read(u) cp
  read(u) vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls
  read(u) ext
  read(u) n, capacity
  ni = gd%ied - gd%isd + 1 ; nj = gd%jed - gd%jsd + 1
  nic = gd%iec - gd%isc + 1 ; njc = gd%jec - gd%jsc + 1
  allocate(gstatic(ni, nj, KID_NGRID_STATIC), bf(max(n,1_c_int64_t), KID_NB_F64), bi(max(n,1_c_int64_t), KID_NB_I32), bid(max(n,1_c_int64_t)))
  allocate(uo(ext(1), ext(2)), ui(ext(1), ext(2)), vo(ext(3), ext(4)), vi(ext(3), ext(4)), tauxa(ext(5), ext(6)), tauya(ext(7), ext(8)))
  allocate(ssh(nic + 2, njc + 2), cn(nic + 2, njc + 2), hi(nic + 2, njc + 2), sst(nic, njc), sss(nic, njc))
  allocate(calving(nic, njc), calving_hflx(nic, njc), mass_berg(nic, njc))
  read(u) gstatic
  do q = 1, KID_NB_F64 ; read(u) bf(1:n, q) ; enddo
  do q = 1, KID_NB_I32 ; read(u) bi(1:n, q) ; enddo
  read(u) bid(1:n)

  ! icebergs_init: device handle, calving tables, and the lists -- the bergs arrive in file order (shuffled by the test) and
  ! every one is inserted where `inorder` puts it, as read_restart_bergs / add_new_berg_to_list do
  call kid_glue_init(bergs, gd, par, gstatic, capacity)
  call kid_glue_set_calving(bergs, cp)
  bergs%tau_is_velocity = tau_is_velocity /= 0
  do k = 1, n
    call row_to_vals(k, vals)
    call kid_glue_add_berg(bergs, vals)
  enddo
  call kid_glue_flatten(bergs)

  open(newunit=uo_, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  m = kid_glue_count(bergs)
  write(uo_) m
  write(uo_) bergs%ids(1:m)          ! flattening order = traversal order (A13)

  none => null()
  do s = 1, ncalls
    read(u) uo ; read(u) ui ; read(u) vo ; read(u) vi ; read(u) tauxa ; read(u) tauya
    read(u) ssh ; read(u) cn ; read(u) hi ; read(u) sst
    if (has_sss /= 0) read(u) sss
    read(u) calving ; read(u) calving_hflx
    mass_berg = -1.
    if (has_sss /= 0) then
      call kid_icebergs_run(bergs, par%current_year, par%current_yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, &
                            stagger=vel_stagger, stress_stagger=stress_stagger, sss=sss, mass_berg=mass_berg, ustar_berg=none)
    else
      call kid_icebergs_run(bergs, par%current_year, par%current_yearday, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, &
                            stagger=vel_stagger, stress_stagger=stress_stagger, mass_berg=mass_berg)
    endif
    write(uo_) calving
    write(uo_) calving_hflx
  enddo
  close(u)
  write(uo_) mass_berg

  ! back to lists (the host needs them for restarts, trajectories, migration), then once through them in traversal order
  call kid_glue_unflatten(bergs)
  m = kid_glue_count(bergs)
  deallocate(bf, bi, bid)
  allocate(bf(max(m,1_c_int64_t), KID_NB_F64), bi(max(m,1_c_int64_t), KID_NB_I32), bid(max(m,1_c_int64_t)))
  bf = 0. ; bi = 0 ; bid = 0
  ! the glue's own flattening puts the rebuilt lists into its staging columns in traversal order: reuse it (upload included:
  ! the device order is then the reference's order again)
  call kid_glue_flatten(bergs)
  write(uo_) m
  do q = 1, KID_NB_F64 ; write(uo_) bergs%f64(1:m, q) ; enddo
  do q = 1, KID_NB_I32 ; write(uo_) bergs%i32(1:m, q) ; enddo
  write(uo_) bergs%ids(1:m)
  ! every list is sorted: the property insert_berg_into_list maintains
  do grdj = gd%jsc, gd%jec ; do grdi = gd%isc, gd%iec
    this => bergs%list(grdi,grdj)%first
    do while (associated(this))
      if (associated(this%next)) then
        if (.not. inorder(this, this%next)) error stop 'kid_glue_test: a rebuilt list is out of order'
      endif
      if (this%ine /= grdi .or. this%jne /= grdj) error stop 'kid_glue_test: a berg sits in the wrong list'
      this => this%next
    enddo
  enddo ; enddo
  close(uo_)
  write(*,'(a,i0,a,i0)') 'kid_glue_test: calls=', ncalls, ' bergs=', m
  call kid_glue_end(bergs)

contains
  subroutine row_to_vals(r, b)
    integer(c_int64_t), intent(in) :: r
    type(iceberg), intent(out) :: b
    b%lon = bf(r, KID_B_LON+1) ; b%lat = bf(r, KID_B_LAT+1) ; b%uvel = bf(r, KID_B_UVEL+1) ; b%vvel = bf(r, KID_B_VVEL+1)
    b%mass = bf(r, KID_B_MASS+1) ; b%thickness = bf(r, KID_B_THICKNESS+1) ; b%width = bf(r, KID_B_WIDTH+1) ; b%length = bf(r, KID_B_LENGTH+1)
    b%start_lon = bf(r, KID_B_START_LON+1) ; b%start_lat = bf(r, KID_B_START_LAT+1) ; b%start_day = bf(r, KID_B_START_DAY+1)
    b%start_mass = bf(r, KID_B_START_MASS+1) ; b%mass_scaling = bf(r, KID_B_MASS_SCALING+1) ; b%mass_of_bits = bf(r, KID_B_MASS_OF_BITS+1)
    b%heat_density = bf(r, KID_B_HEAT_DENSITY+1) ; b%xi = bf(r, KID_B_XI+1) ; b%yj = bf(r, KID_B_YJ+1)
    b%axn = bf(r, KID_B_AXN+1) ; b%ayn = bf(r, KID_B_AYN+1) ; b%bxn = bf(r, KID_B_BXN+1) ; b%byn = bf(r, KID_B_BYN+1)
    b%uvel_old = bf(r, KID_B_UVEL_OLD+1) ; b%vvel_old = bf(r, KID_B_VVEL_OLD+1) ; b%lon_old = bf(r, KID_B_LON_OLD+1) ; b%lat_old = bf(r, KID_B_LAT_OLD+1)
    b%halo_berg = bf(r, KID_B_HALO_BERG+1) ; b%static_berg = bf(r, KID_B_STATIC_BERG+1)
    b%ine = bi(r, KID_BI_INE+1) ; b%jne = bi(r, KID_BI_JNE+1) ; b%start_year = bi(r, KID_BI_START_YEAR+1)
    b%id = bid(r)
  end subroutine row_to_vals
end program kid_glue_test
