!> Test driver of the boundary as a maintainer of the reference meets it: icebergs_init's argument list, the namelist group
!! icebergs_nml read from ./input.nml, bergs as heap nodes in per-cell lists, bonds as per-berg lists made by form_a_bond.
!!
!! The stand-alone driver's Cartesian grid (driver/icebergs_driver.F90:274-286: lon = gridres*i, lat = gridres*j, all wet,
!! 1000 m deep; halo 1 handed to icebergs_init, DRV:337-342) goes through kid_icebergs_init; the population arrives in file
!! order and is inserted with kid_glue_add_berg; every bond is formed with form_a_bond in the order the case file lists them
!! (the test writes each berg's bonds last slot first: head insertion then gives the slot order of the tables the oracle was
!! given); kid_icebergs_init_bonds does the bonded tail of icebergs_init; nsteps calls of kid_icebergs_run with the ocean at
!! rest; the lists and bond lists are rebuilt from the device and written out.
!! Case file (stream): int32 magic 1263093765, gni, gnj, dom_x_flags, nsteps; real64 gridres, dt, sst, sss; int64 capacity, n;
!! KID_NB_F64 columns of n, KID_NB_I32 columns, ids; int64 nb; nb pairs (id, other_id).
!! Output: kid_grid_desc, kid_params, the static planes; int64 m; the rebuilt bergs in traversal order (columns as above); per
!! berg: int32 count, then per bond in list order int64 other_id, int64 id of other_berg (-1 when not connected), int32
!! broken, int32 other_berg_ine, 12 reals (KID_BOND_* order).
!! Written and checked by tests/test_fortran_gpu.py::test_icebergs_init_and_bond_lists_cantilever.
program kid_init_test
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  use kid_icebergs_glue
  implicit none
  character(len=1024) :: fin, fout
  type(kid_glue), target :: bergs
  type(iceberg) :: vals
  type(iceberg), pointer :: this, other
  type(bond), pointer :: b
  integer(c_int32_t) :: magic, gni, gnj, dom_x_flags, nsteps, cnt
  real(c_double) :: gridres, dt, sst0, sss0
  integer(c_int64_t) :: capacity, n, nb, k, m, ida, idb, oid
  integer :: u, uo_, q, s, i, j, grdi, grdj
  real(c_double), allocatable :: lon(:,:), lat(:,:), wet(:,:), dx(:,:), dy(:,:), area(:,:), cos_rot(:,:), sin_rot(:,:), depth(:,:)
  real(c_double), allocatable, target :: uo(:,:), vo(:,:), ui(:,:), vi(:,:), tauxa(:,:), tauya(:,:), ssh(:,:), sst(:,:), cn(:,:), hi(:,:), &
      sss(:,:), calving(:,:), calving_hflx(:,:)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', action='read')
  read(u) magic
  if (magic /= 1263093765) error stop 'kid_init_test: bad magic'
  read(u) gni, gnj, dom_x_flags, nsteps
  read(u) gridres, dt, sst0, sss0
  read(u) capacity, n

  ! the driver's grid on the ice model's data domain (halo 1)
  allocate(lon(0:gni+1, 0:gnj+1), lat(0:gni+1, 0:gnj+1), wet(0:gni+1, 0:gnj+1), dx(0:gni+1, 0:gnj+1), dy(0:gni+1, 0:gnj+1), &
           area(0:gni+1, 0:gnj+1), cos_rot(0:gni+1, 0:gnj+1), sin_rot(0:gni+1, 0:gnj+1), depth(0:gni+1, 0:gnj+1))
  do j = 0, gnj + 1 ; do i = 0, gni + 1
    lon(i,j) = gridres * real(i, c_double) ; lat(i,j) = gridres * real(j, c_double)
    dx(i,j) = gridres ; dy(i,j) = gridres ; area(i,j) = gridres * gridres
    wet(i,j) = 1. ; cos_rot(i,j) = 1. ; sin_rot(i,j) = 0. ; depth(i,j) = 1000.
  enddo ; enddo

  call kid_icebergs_init(bergs, gni, gnj, (/1, 1/), (/1, 1/), (/0, 0/), dom_x_flags, 0, dt, 1, 0._c_double, &
                         lon(1:gni,1:gnj), lat(1:gni,1:gnj), wet, dx, dy, area(1:gni,1:gnj), cos_rot, sin_rot, &
                         ocean_depth=depth(1:gni,1:gnj), fractional_area=.false., capacity=capacity)

  ! the population: staged columns -> nodes -> lists (file order; insert_berg_into_list sorts)
  do q = 1, KID_NB_F64 ; read(u) bergs%f64(1:n, q) ; enddo
  do q = 1, KID_NB_I32 ; read(u) bergs%i32(1:n, q) ; enddo
  read(u) bergs%ids(1:n)
  do k = 1, n
    call row_to_node(bergs, k, vals)
    call kid_glue_add_berg(bergs, vals)
  enddo
  read(u) nb
  do k = 1, nb
    read(u) ida, idb
    this => kid_glue_find_berg(bergs, ida)
    other => kid_glue_find_berg(bergs, idb)
    if (.not. associated(this) .or. .not. associated(other)) error stop 'kid_init_test: a bond names a berg that is not in the lists'
    call form_a_bond(this, idb, other_berg=other)
  enddo
  close(u)
  call kid_icebergs_init_bonds(bergs)
  call kid_glue_flatten(bergs)

  allocate(uo(gni+2, gnj+2), vo(gni+2, gnj+2), ui(gni+2, gnj+2), vi(gni+2, gnj+2), tauxa(gni, gnj), tauya(gni, gnj), &
           ssh(gni+2, gnj+2), cn(gni+2, gnj+2), hi(gni+2, gnj+2), sst(gni, gnj), sss(gni, gnj), calving(gni, gnj), calving_hflx(gni, gnj))
  uo = 0. ; vo = 0. ; ui = 0. ; vi = 0. ; tauxa = 0. ; tauya = 0. ; ssh = 0. ; cn = 0. ; hi = 0. ; sst = sst0 ; sss = sss0
  do s = 1, nsteps
    calving = 0. ; calving_hflx = 0.
    call kid_icebergs_run(bergs, 1, real(s - 1, c_double) * dt / 86400._c_double, calving, uo, vo, ui, vi, tauxa, tauya, ssh, sst, calving_hflx, cn, hi, sss=sss)
  enddo

  call kid_glue_unflatten(bergs)
  open(newunit=uo_, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(uo_) bergs%gd ; write(uo_) bergs%par ; write(uo_) bergs%static
  m = kid_glue_count(bergs)
  write(uo_) m
  k = 0
  do grdj = bergs%gd%jsc, bergs%gd%jec ; do grdi = bergs%gd%isc, bergs%gd%iec
    this => bergs%list(grdi,grdj)%first
    do while (associated(this))
      k = k + 1
      call node_to_row(this, bergs, k)
      this => this%next
    enddo
  enddo ; enddo
  do q = 1, KID_NB_F64 ; write(uo_) bergs%f64(1:m, q) ; enddo
  do q = 1, KID_NB_I32 ; write(uo_) bergs%i32(1:m, q) ; enddo
  write(uo_) bergs%ids(1:m)
  do grdj = bergs%gd%jsc, bergs%gd%jec ; do grdi = bergs%gd%isc, bergs%gd%iec
    this => bergs%list(grdi,grdj)%first
    do while (associated(this))
      cnt = 0
      b => this%first_bond
      do while (associated(b)) ; cnt = cnt + 1 ; b => b%next_bond ; enddo
      write(uo_) cnt
      b => this%first_bond
      do while (associated(b))
        oid = -1 ; if (associated(b%other_berg)) oid = b%other_berg%id
        write(uo_) b%other_id, oid, int(b%broken, c_int32_t), int(b%other_berg_ine, c_int32_t)
        write(uo_) b%length, b%tangd1, b%tangd2, b%nstress, b%sstress, b%rel_rotation, b%F_x, b%F_y, b%Fd_x, b%Fd_y, b%T, b%T_d
        b => b%next_bond
      enddo
      this => this%next
    enddo
  enddo ; enddo
  close(uo_)
  write(*,'(a,i0,a,i0)') 'kid_init_test: steps=', nsteps, ' bergs=', m
  call kid_glue_end(bergs)
end program kid_init_test
