!> The multi-GPU pattern of INTEGRATION.md section 6 from Fortran, with the cross-rank sum done by hand.
!!
!! In a coupled model every MPI rank owns one GPU and one share of the bergs on a replicated grid; per step each rank runs
!! kid_icebergs_run_local (= kid_step_local), the ranks sum the live prefix of the accumulator block (MPI_Allreduce on the
!! device pointer), and each rank runs kid_icebergs_run_finish (= kid_step_gather + the coupler return).  This program plays
!! two ranks in one process: two glue objects / two handles on the same GPU, the bergs dealt alternately to them in list
!! order, and the sum done with hipMemcpy through the host -- exactly the arithmetic of a two-rank MPI_SUM (a + b).
!! Both "ranks" must then hand the coupler the same fields, and those must be what one handle with all the bergs gives
!! (tests/test_fortran_gpu.py::test_fortran_two_handles_summed_by_hand checks both against the oracle).
!! Case file: the one of kid_glue_test (magic 1263093764), calving fields ignored (the calving source is a per-cell, not a
!! per-berg, computation: one rank of a replicated grid runs it, INTEGRATION.md section 6).
!! Output: int64 live count, int64 n_a, n_b; per call and per rank (a then b) calving, calving_hflx; mass_berg of rank a and
!! of rank b after the last call; then for each rank: int64 m and the bergs of its rebuilt lists (columns).
program kid_multi_test
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  use kid_icebergs_glue
  implicit none
  interface
    integer(c_int) function hipMemcpy(dst, src, nbytes, kind) bind(C, name='hipMemcpy')
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: nbytes
      integer(c_int), value :: kind
    end function hipMemcpy
  end interface
  integer(c_int), parameter :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2
  character(len=1024) :: fin, fout
  type(kid_glue), target :: rank(2)
  type(kid_grid_desc) :: gd
  type(kid_params) :: par
  type(kid_calving_params) :: cp
  type(iceberg) :: vals
  type(iceberg), pointer :: this
  integer(c_int32_t) :: magic, vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls, ext(8)
  integer(c_int64_t) :: n, capacity, m, k, total, live, live2, nab(2)
  integer :: ni, nj, nic, njc, q, s, u, uo_, grdi, grdj, r
  real(c_double), allocatable, target :: gstatic(:,:,:), sum_a(:), sum_b(:)
  real(c_double), allocatable, target :: uo(:,:), vo(:,:), ui(:,:), vi(:,:), tauxa(:,:), tauya(:,:), ssh(:,:), sst(:,:), &
      cn(:,:), hi(:,:), sss(:,:), calving(:,:,:), calving_hflx(:,:,:), skip(:,:)
  real(c_double), pointer :: mass_a(:,:), mass_b(:,:)
  type(kid_glue), target :: all_bergs   ! host-only lists: deals the population in traversal order
  type(c_ptr) :: dev(2)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', action='read')
  read(u) magic
  if (magic /= 1263093764) error stop 'kid_multi_test: bad magic'
  read(u) gd ; read(u) par ; read(u) cp
  read(u) vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls
  read(u) ext
  read(u) n, capacity
  ni = gd%ied - gd%isd + 1 ; nj = gd%jed - gd%jsd + 1
  nic = gd%iec - gd%isc + 1 ; njc = gd%jec - gd%jsc + 1
  allocate(gstatic(ni, nj, KID_NGRID_STATIC))
  allocate(uo(ext(1), ext(2)), ui(ext(1), ext(2)), vo(ext(3), ext(4)), vi(ext(3), ext(4)), tauxa(ext(5), ext(6)), tauya(ext(7), ext(8)))
  allocate(ssh(nic + 2, njc + 2), cn(nic + 2, njc + 2), hi(nic + 2, njc + 2), sst(nic, njc), sss(nic, njc), skip(nic, njc))
  allocate(calving(nic, njc, 2), calving_hflx(nic, njc, 2), mass_a(nic, njc), mass_b(nic, njc))
  read(u) gstatic

  ! two "ranks": the same grid and parameters on each, half of the bergs each
  do r = 1, 2
    call kid_glue_init(rank(r), gd, par, gstatic, capacity)
    rank(r)%tau_is_velocity = tau_is_velocity /= 0
  enddo
  ! the population through sorted lists first (file order is shuffled), then dealt alternately in traversal order
  all_bergs%gd = gd ; all_bergs%par = par ; all_bergs%capacity = n
  allocate(all_bergs%list(gd%isd:gd%ied, gd%jsd:gd%jed), all_bergs%f64(n, KID_NB_F64), all_bergs%i32(n, KID_NB_I32), all_bergs%ids(n))
  do q = 1, KID_NB_F64 ; read(u) all_bergs%f64(1:n, q) ; enddo
  do q = 1, KID_NB_I32 ; read(u) all_bergs%i32(1:n, q) ; enddo
  read(u) all_bergs%ids(1:n)
  do k = 1, n
    call row_to_node(all_bergs, k, vals)
    call kid_glue_add_berg(all_bergs, vals)
  enddo
  k = 0
  do grdj = gd%jsc, gd%jec ; do grdi = gd%isc, gd%iec
    this => all_bergs%list(grdi,grdj)%first
    do while (associated(this))
      k = k + 1
      vals = this
      call kid_glue_add_berg(rank(1 + int(mod(k, 2_c_int64_t))), vals)
      this => this%next
    enddo
  enddo ; enddo
  call kid_glue_clear_lists(all_bergs)
  do r = 1, 2
    call kid_glue_flatten(rank(r))
    nab(r) = kid_glue_count(rank(r))
  enddo

  call kid_check(kid_accum_live_count(rank(1)%h, live), rank(1)%h, 'kid_accum_live_count')
  call kid_check(kid_accum_live_count(rank(2)%h, live2), rank(2)%h, 'kid_accum_live_count')
  if (live /= live2) error stop 'kid_multi_test: the two handles disagree on what a step fills'
  allocate(sum_a(live), sum_b(live))
  open(newunit=uo_, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(uo_) live, nab

  do s = 1, ncalls
    read(u) uo ; read(u) ui ; read(u) vo ; read(u) vi ; read(u) tauxa ; read(u) tauya
    read(u) ssh ; read(u) cn ; read(u) hi ; read(u) sst
    if (has_sss /= 0) read(u) sss
    read(u) skip ; read(u) skip
    calving = 0. ; calving_hflx = 0.
    ! 1. every rank: its own bergs
    do r = 1, 2
      if (has_sss /= 0) then
        call kid_icebergs_run_local(rank(r), par%current_year, par%current_yearday, calving(:,:,r), uo, vo, ui, vi, tauxa, tauya, ssh, sst, &
                                    calving_hflx(:,:,r), cn, hi, stagger=vel_stagger, stress_stagger=stress_stagger, sss=sss)
      else
        call kid_icebergs_run_local(rank(r), par%current_year, par%current_yearday, calving(:,:,r), uo, vo, ui, vi, tauxa, tauya, ssh, sst, &
                                    calving_hflx(:,:,r), cn, hi, stagger=vel_stagger, stress_stagger=stress_stagger)
      endif
    enddo
    ! 2. the exchange: MPI_Allreduce(MPI_IN_PLACE, dev, live, MPI_DOUBLE_PRECISION, MPI_SUM, comm), by hand
    do r = 1, 2
      call kid_check(kid_sync(rank(r)%h), rank(r)%h, 'kid_sync')
      call kid_check(kid_accum_device_ptr(rank(r)%h, dev(r), total), rank(r)%h, 'kid_accum_device_ptr')
    enddo
    if (hipMemcpy(c_loc(sum_a), dev(1), int(8 * live, c_size_t), hipMemcpyDeviceToHost) /= 0) error stop 'kid_multi_test: hipMemcpy'
    if (hipMemcpy(c_loc(sum_b), dev(2), int(8 * live, c_size_t), hipMemcpyDeviceToHost) /= 0) error stop 'kid_multi_test: hipMemcpy'
    sum_a = sum_a + sum_b
    do r = 1, 2
      if (hipMemcpy(dev(r), c_loc(sum_a), int(8 * live, c_size_t), hipMemcpyHostToDevice) /= 0) error stop 'kid_multi_test: hipMemcpy'
    enddo
    ! 3. every rank: the gather and the coupler return from the summed block
    call kid_icebergs_run_finish(rank(1), calving(:,:,1), calving_hflx(:,:,1), mass_berg=mass_a)
    call kid_icebergs_run_finish(rank(2), calving(:,:,2), calving_hflx(:,:,2), mass_berg=mass_b)
    do r = 1, 2
      write(uo_) calving(:,:,r)
      write(uo_) calving_hflx(:,:,r)
    enddo
  enddo
  close(u)
  write(uo_) mass_a
  write(uo_) mass_b
  do r = 1, 2
    call kid_glue_unflatten(rank(r))
    m = kid_glue_count(rank(r))
    call kid_glue_flatten(rank(r))
    write(uo_) m
    do q = 1, KID_NB_F64 ; write(uo_) rank(r)%f64(1:m, q) ; enddo
    do q = 1, KID_NB_I32 ; write(uo_) rank(r)%i32(1:m, q) ; enddo
    write(uo_) rank(r)%ids(1:m)
  enddo
  close(uo_)
  write(*,'(a,i0,a,i0,a,i0)') 'kid_multi_test: calls=', ncalls, ' bergs=', nab(1), '+', nab(2)
  call kid_glue_end(rank(1))
  call kid_glue_end(rank(2))
end program kid_multi_test
