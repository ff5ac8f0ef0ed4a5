!> Stand-alone Fortran driver over the ISO_C_BINDING shim: replays a case file through the HIP library.
!!
!! It plays the role of driver/icebergs_driver.F90 for the evolve-loop path: it owns the grid, the forcing and
!! the berg arrays on the host, calls the library at the same call sites icebergs_run has (icebergs.F90:5423-5512)
!! and writes the state back.  The case file (little-endian stream) is produced by tests/test_fortran_gpu.py:
!!   int32 magic(=1263093761), kid_grid_desc, kid_params, int32 nsteps, int32 mode(0 fused, 1 phase by phase),
!!   int64 n, KID_NGRID_STATIC planes, KID_NFORCING planes (ni*nj fp64 each), KID_NB_F64 arrays of n fp64,
!!   KID_NB_I32 arrays of n int32, n int64 ids.
!! With magic 1263093762 a bonds section follows (bonded conglomerates, IB:5409-5431): int32 max_bonds, n int32 counts,
!!   max_bonds*n int64 partner ids, max_bonds*n int32 broken marks, KID_NBOND_F64 arrays of max_bonds*n fp64 (slot-major,
!!   include/kid_types.h kid_bond_soa) -- uploaded with kid_upload_bonds, stepped with kid_run_step (which dispatches to
!!   the MTS / DEM path), downloaded with kid_download_bonds.
!! Output file: int64 n_slots, the berg arrays in the same order, KID_NACC + KID_NOUT planes, KID_NSCALAR scalars
!!   (and, with bonds, the counts, partner ids, broken marks and bond fields as they came back).
program kid_replay
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  implicit none
  character(len=1024) :: fin, fout
  type(kid_grid_desc) :: gd
  type(kid_params) :: par
  type(kid_berg_soa) :: soa
  type(kid_bond_soa) :: bsoa
  type(c_ptr) :: h
  logical :: with_bonds
  integer(c_int32_t) :: mb
  integer(c_int32_t), allocatable, target :: bcount(:), bbroken(:,:)
  integer(c_int64_t), allocatable, target :: bother(:,:)
  real(c_double), allocatable, target :: bstate(:,:,:)
  integer(c_int32_t) :: magic, nsteps, mode
  integer(c_int64_t) :: n, n_slots, n_alive
  integer :: ni, nj, k, s, u
  real(c_double), allocatable, target :: gstatic(:,:,:), gforc(:,:,:), bf(:,:), acc(:,:,:), outp(:,:,:), scal(:)
  integer(c_int32_t), allocatable, target :: bi(:,:)
  integer(c_int64_t), allocatable, target :: bid(:)
  type(c_ptr) :: pst(KID_NGRID_STATIC), pfo(KID_NFORCING)

  if (command_argument_count() < 2) then
    write(0,*) 'usage: kid_replay <case.bin> <result.bin>'
    error stop 2
  end if
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)

  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', action='read')
  read(u) magic
  if (magic /= 1263093761 .and. magic /= 1263093762) error stop 'kid_replay: bad magic'
  with_bonds = magic == 1263093762
  read(u) gd
  read(u) par
  read(u) nsteps, mode
  read(u) n
  ni = gd%ied - gd%isd + 1
  nj = gd%jed - gd%jsd + 1
  allocate(gstatic(ni, nj, KID_NGRID_STATIC), gforc(ni, nj, KID_NFORCING))
  allocate(bf(max(n,1_c_int64_t), KID_NB_F64), bi(max(n,1_c_int64_t), KID_NB_I32), bid(max(n,1_c_int64_t)))
  read(u) gstatic
  read(u) gforc
  if (n > 0) then
    read(u) bf(1:n,:)
    read(u) bi(1:n,:)
    read(u) bid(1:n)
  end if
  mb = 0
  if (with_bonds) then
    read(u) mb
    allocate(bcount(n), bother(n, mb), bbroken(n, mb), bstate(n, mb, KID_NBOND_F64))
    read(u) bcount
    read(u) bother
    read(u) bbroken
    read(u) bstate
  end if
  close(u)

  ! icebergs_init: grid + parameters -> device (icebergs.F90:92-178)
  call kid_check(kid_create(gd, par, max(n, 1_c_int64_t), 0_c_int, h), h, 'kid_create')
  do k = 1, KID_NGRID_STATIC
    pst(k) = c_loc(gstatic(1,1,k))
  end do
  do k = 1, KID_NFORCING
    pfo(k) = c_loc(gforc(1,1,k))
  end do
  call kid_check(kid_set_static_grid(h, pst), h, 'kid_set_static_grid')

  soa%n = n
  do k = 1, KID_NB_F64
    soa%f64(k) = c_loc(bf(1,k))
  end do
  do k = 1, KID_NB_I32
    soa%i32(k) = c_loc(bi(1,k))
  end do
  soa%id = c_loc(bid(1))
  call kid_check(kid_upload_bergs(h, soa), h, 'kid_upload_bergs')
  if (with_bonds) then   ! the bond lists of the conglomerates (INTEGRATION.md section 4)
    bsoa%n = n; bsoa%max_bonds = mb; bsoa%pad = 0
    bsoa%count = c_loc(bcount(1)); bsoa%other_id = c_loc(bother(1,1)); bsoa%broken = c_loc(bbroken(1,1))
    do k = 1, KID_NBOND_F64
      bsoa%f64(k) = c_loc(bstate(1,1,k))
    end do
    call kid_check(kid_upload_bonds(h, bsoa), h, 'kid_upload_bonds')
  end if

  ! the time loop of the driver (driver/icebergs_driver.F90:354-408); forcing is constant in the case file
  do s = 1, nsteps
    call kid_check(kid_set_forcing(h, pfo), h, 'kid_set_forcing')                   ! result of IB:5236-5383
    if (mode == 0) then
      call kid_check(kid_run_step(h, 1_c_int), h, 'kid_run_step')
    else
      call kid_check(kid_zero_accumulators(h), h, 'kid_zero_accumulators')          ! IB:5125-5156
      if (par%mts == 0 .and. par%old_interp_flds_order == 0) &
        call kid_check(kid_interp_gridded_fields_to_bergs(h), h, 'interp (1)')        ! IB:5423
      call kid_check(kid_evolve_icebergs(h), h, 'kid_evolve_icebergs')              ! IB:5433
      if (par%footloose /= 0) call kid_check(kid_footloose_calving(h), h, 'kid_footloose_calving') ! IB:5453
      if (par%old_interp_flds_order == 0) &
        call kid_check(kid_interp_gridded_fields_to_bergs(h), h, 'interp (2)')        ! IB:5473
      call kid_check(kid_thermodynamics(h), h, 'kid_thermodynamics')                ! IB:5505
      call kid_check(kid_create_gridded_icebergs_fields(h), h, 'kid_create_gridded_icebergs_fields') ! IB:5512
    end if
  end do

  call kid_check(kid_num_bergs(h, n_slots, n_alive), h, 'kid_num_bergs')
  soa%n = n_slots
  call kid_check(kid_download_bergs(h, soa), h, 'kid_download_bergs')
  if (with_bonds) call kid_check(kid_download_bonds(h, bsoa), h, 'kid_download_bonds')
  allocate(acc(ni, nj, KID_NACC), outp(ni, nj, KID_NOUT), scal(KID_NSCALAR))
  call kid_check(kid_get_accumulators(h, c_loc(acc), c_loc(outp), c_loc(scal)), h, 'kid_get_accumulators')
  call kid_check(kid_destroy(h), h, 'kid_destroy')

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(u) n_slots
  if (n_slots > 0) then
    write(u) bf(1:n_slots,:)
    write(u) bi(1:n_slots,:)
    write(u) bid(1:n_slots)
  end if
  write(u) acc
  write(u) outp
  write(u) scal
  if (with_bonds) then
    write(u) bcount
    write(u) bother
    write(u) bbroken
    write(u) bstate
  end if
  close(u)
  write(*,'(a,i0,a,i0,a,i0)') 'kid_replay: steps=', nsteps, ' bergs=', n_slots, ' alive=', n_alive
end program kid_replay
