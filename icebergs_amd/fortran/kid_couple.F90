!> Stand-alone Fortran driver for the front half of icebergs_run on the device: forcing ingest (icebergs.F90:5236-5383),
!! the calving source (IB:5203-5231, accumulate_calving IB:5388, calve_icebergs IB:5403) and the evolve step, through the
!! ISO_C_BINDING module only.  It stands where the coupler stands: it owns the arrays icebergs_run receives as dummy
!! arguments (uo, vo, ui, vi, tauxa, tauya, ssh, sst, cn, hi, sss, calving, calving_hflx) with the extents their stagger
!! implies, and hands them over once per step.
!! Case file (little-endian stream, written by tests/test_fortran_gpu.py):
!!   int32 magic(=1263093762), kid_grid_desc, kid_params, kid_calving_params,
!!   int32 vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls,
!!   int32 u_ni, u_nj, v_ni, v_nj, taux_ni, taux_nj, tauy_ni, tauy_nj,
!!   int64 n, capacity, KID_NGRID_STATIC planes, the berg arrays (as kid_replay),
!!   then per call: uo, ui (u extents), vo, vi (v extents), tauxa, tauya, ssh, cn, hi (nic+2, njc+2), sst [, sss], calving,
!!   calving_hflx (nic, njc).
!! Output: KID_NFORCING planes (kid_get_forcing), stored_ice (10 planes), stored_heat, real_calving (10), grd%calving,
!!   grd%calving_hflx, KID_NCALV_SCALARS of the last call, int64 n_slots, the berg arrays.
!! With a third argument (a directory) the restart files and one trajectory sample are written there before the download.
program kid_couple
  use, intrinsic :: iso_c_binding
  use kid_hip_mod
  implicit none
  character(len=1024) :: fin, fout, rdir
  type(kid_traj_params) :: tp
  type(kid_grid_desc) :: gd
  type(kid_params) :: par
  type(kid_calving_params) :: cp
  type(kid_forcing_in) :: fi
  type(kid_calving_in) :: ci
  type(kid_berg_soa) :: soa
  type(c_ptr) :: h
  integer(c_int32_t) :: magic, vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls, ext(8)
  integer(c_int64_t) :: n, capacity, n_slots, n_alive
  integer :: ni, nj, nic, njc, k, s, u
  real(c_double), allocatable, target :: gstatic(:,:,:), planes(:,:,:), bf(:,:), stored_ice(:,:,:), stored_heat(:,:), &
      real_calving(:,:,:), rm(:,:), rmh(:,:), gcalv(:,:), ghflx(:,:), scal(:)
  real(c_double), allocatable, target :: uo(:,:), vo(:,:), ui(:,:), vi(:,:), tauxa(:,:), tauya(:,:), ssh(:,:), sst(:,:), &
      cn(:,:), hi(:,:), sss(:,:), calving(:,:), calving_hflx(:,:)
  integer(c_int32_t), allocatable, target :: bi(:,:)
  integer(c_int64_t), allocatable, target :: bid(:)
  type(c_ptr) :: pst(KID_NGRID_STATIC), pfo(KID_NFORCING)

  if (command_argument_count() < 2) then
    write(0,*) 'usage: kid_couple <case.bin> <result.bin>'
    error stop 2
  end if
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  rdir = ''
  if (command_argument_count() >= 3) call get_command_argument(3, rdir)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old', action='read')
  read(u) magic
  if (magic /= 1263093762) error stop 'kid_couple: bad magic'
  read(u) gd
  read(u) par
  read(u) cp
  read(u) vel_stagger, stress_stagger, tau_is_velocity, cyclic_x, has_sss, ncalls
  read(u) ext
  read(u) n, capacity
  ni = gd%ied - gd%isd + 1 ; nj = gd%jed - gd%jsd + 1
  nic = gd%iec - gd%isc + 1 ; njc = gd%jec - gd%jsc + 1
  allocate(gstatic(ni, nj, KID_NGRID_STATIC), planes(ni, nj, KID_NFORCING))
  allocate(bf(capacity, KID_NB_F64), bi(capacity, KID_NB_I32), bid(capacity))
  allocate(uo(ext(1), ext(2)), ui(ext(1), ext(2)), vo(ext(3), ext(4)), vi(ext(3), ext(4)))
  allocate(tauxa(ext(5), ext(6)), tauya(ext(7), ext(8)))
  allocate(ssh(nic + 2, njc + 2), cn(nic + 2, njc + 2), hi(nic + 2, njc + 2), sst(nic, njc), sss(nic, njc))
  allocate(calving(nic, njc), calving_hflx(nic, njc))
  read(u) gstatic
  bf = 0. ; bi = 0 ; bid = 0
  if (n > 0) then
    do k = 1, KID_NB_F64
      read(u) bf(1:n, k)
    end do
    do k = 1, KID_NB_I32
      read(u) bi(1:n, k)
    end do
    read(u) bid(1:n)
  end if

  ! icebergs_init (icebergs.F90:92-178): grid, parameters, the calving tables of ice_bergs_framework_init (FW:1534-1551)
  call kid_check(kid_create(gd, par, capacity, 0_c_int, h), h, 'kid_create')
  do k = 1, KID_NGRID_STATIC
    pst(k) = c_loc(gstatic(1,1,k))
  end do
  call kid_check(kid_set_static_grid(h, pst), h, 'kid_set_static_grid')
  call kid_check(kid_set_calving_params(h, cp), h, 'kid_set_calving_params')
  soa%n = n
  do k = 1, KID_NB_F64
    soa%f64(k) = c_loc(bf(1,k))
  end do
  do k = 1, KID_NB_I32
    soa%i32(k) = c_loc(bi(1,k))
  end do
  soa%id = c_loc(bid(1))
  call kid_check(kid_upload_bergs(h, soa), h, 'kid_upload_bergs')

  ! what icebergs_run receives, as the coupler owns it
  fi%uo = c_loc(uo) ; fi%vo = c_loc(vo) ; fi%ui = c_loc(ui) ; fi%vi = c_loc(vi)
  fi%tauxa = c_loc(tauxa) ; fi%tauya = c_loc(tauya)
  fi%ssh = c_loc(ssh) ; fi%sst = c_loc(sst) ; fi%cn = c_loc(cn) ; fi%hi = c_loc(hi)
  fi%sss = c_null_ptr ; if (has_sss /= 0) fi%sss = c_loc(sss)            ! present(sss), IB:5354
  fi%u_ni = size(uo,1) ; fi%u_nj = size(uo,2) ; fi%v_ni = size(vo,1) ; fi%v_nj = size(vo,2)
  fi%taux_ni = size(tauxa,1) ; fi%taux_nj = size(tauxa,2) ; fi%tauy_ni = size(tauya,1) ; fi%tauy_nj = size(tauya,2)
  fi%vel_stagger = vel_stagger ; fi%stress_stagger = stress_stagger
  fi%tau_is_velocity = tau_is_velocity ; fi%cyclic_x = cyclic_x ; fi%on_device = 0 ; fi%pad = 0
  ci%calving = c_loc(calving) ; ci%calving_hflx = c_loc(calving_hflx) ; ci%on_device = 0 ; ci%pad = 0
  allocate(scal(KID_NCALV_SCALARS))

  do s = 1, ncalls
    read(u) uo ; read(u) ui ; read(u) vo ; read(u) vi ; read(u) tauxa ; read(u) tauya
    read(u) ssh ; read(u) cn ; read(u) hi ; read(u) sst
    if (has_sss /= 0) read(u) sss
    read(u) calving ; read(u) calving_hflx
    call kid_check(kid_ingest_forcing(h, fi), h, 'kid_ingest_forcing')             ! IB:5236-5383
    call kid_check(kid_calving(h, ci, scal), h, 'kid_calving')                     ! IB:5203-5231 (commutes with the ingest), 5388, 5403
    call kid_check(kid_run_step(h, 1_c_int), h, 'kid_run_step')                    ! IB:5423-5512
  end do
  close(u)

  do k = 1, KID_NFORCING
    pfo(k) = c_loc(planes(1,1,k))
  end do
  call kid_check(kid_get_forcing(h, pfo), h, 'kid_get_forcing')
  allocate(stored_ice(ni, nj, KID_NCLASSES), stored_heat(ni, nj), real_calving(ni, nj, KID_NCLASSES), rm(ni, nj), rmh(ni, nj))
  allocate(gcalv(ni, nj), ghflx(ni, nj))
  call kid_check(kid_get_calving_state(h, c_loc(stored_ice), c_loc(stored_heat), c_loc(rm), c_loc(rmh), c_loc(real_calving)), h, &
                 'kid_get_calving_state')
  call kid_check(kid_get_calving(h, c_loc(gcalv), c_loc(ghflx)), h, 'kid_get_calving')
  if (len_trim(rdir) > 0) then
    ! write_restart_bergs (icebergs_fms2io.F90:124-631) and one trajectory sample + write_trajectory (FW:5328, IO2:1631)
    ! straight from the device-resident state
    tp%traj_area_thres = 0. ; tp%traj_area_thres_sntbc = 0. ; tp%traj_area_thres_fl = 1.e9 ; tp%save_all_traj_year = 1.e30
    tp%save_traj_by_class_start_mass_thres_s = 0. ; tp%save_traj_by_class_start_mass_thres_n = 0.
    tp%save_short_traj = 1 ; tp%save_fl_traj = 1 ; tp%save_nonfl_traj_by_class = 0 ; tp%save_bond_traj = 0
    call kid_check(kid_set_traj_params(h, tp), h, 'kid_set_traj_params')
    call kid_check(kid_record_posn(h), h, 'kid_record_posn')
    call kid_check(kid_write_trajectories(h, trim(rdir)//'/iceberg_trajectories.nc'//c_null_char), h, 'kid_write_trajectories')
    call kid_check(kid_write_restart(h, trim(rdir)//c_null_char), h, 'kid_write_restart')
  end if
  call kid_check(kid_num_bergs(h, n_slots, n_alive), h, 'kid_num_bergs')
  soa%n = n_slots
  call kid_check(kid_download_bergs(h, soa), h, 'kid_download_bergs')
  ! the packing loops of send_bergs_to_other_pes for east + west (FW:3022-3048) through the module, on the state just
  ! downloaded: what a decomposed model would hand to mpp_send
  block
    integer(c_int32_t) :: bw
    integer(c_int64_t) :: n_e, n_w, rows
    real(c_double), allocatable :: obuf_e(:,:), obuf_w(:,:)
    call kid_check(kid_buffer_width(h, bw), h, 'kid_buffer_width')
    rows = max(n_slots, 1_c_int64_t)
    allocate(obuf_e(bw, rows), obuf_w(bw, rows))
    call kid_check(kid_pack_emigrants_pair(h, 0_c_int32_t, obuf_e, rows, n_e, obuf_w, rows, n_w), h, 'kid_pack_emigrants_pair')
    write(*,'(a,i0,a,i0,a,i0)') 'migration: width ', bw, ' east ', n_e, ' west ', n_w
    if (n_e > 0) write(*,'(a,i0,a,i0)') 'migration: first east record ine ', nint(obuf_e(24,1)), ' id_ij ', nint(obuf_e(33,1), c_int64_t)
  end block
  call kid_check(kid_destroy(h), h, 'kid_destroy')

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(u) planes
  write(u) stored_ice
  write(u) stored_heat
  write(u) real_calving
  write(u) gcalv
  write(u) ghflx
  write(u) scal
  write(u) n_slots
  if (n_slots > 0) then
    do k = 1, KID_NB_F64
      write(u) bf(1:n_slots, k)
    end do
    do k = 1, KID_NB_I32
      write(u) bi(1:n_slots, k)
    end do
    write(u) bid(1:n_slots)
  end if
  close(u)
  write(*,'(a,i0,a,i0,a,i0)') 'kid_couple: calls=', ncalls, ' bergs=', n_slots, ' alive=', n_alive
end program kid_couple
