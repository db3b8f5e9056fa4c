! ecckd_driver.F90 -- Fortran host program over the drop-in module, shaped like the block loop of
! example/rfmip-rad-irf/ecckd_rfmip_lw.F90:107-136 and ecckd_rfmip_sw.F90:112-162: load the ecCKD
! file, then per column block gas_optics() followed by rte_lw()/rte_sw(), fluxes out.
!
!   ecckd_driver lw|sw  <ecckd_file.nc>  <input.bin>  <output.bin>  [block_size] [n_quad_angles] [device_resident 0|1] [repeats] [byband 0|1] [fused 0|1]
!
! device_resident = 1: optical_props / source are the device twins of mo_ecckd_device (tau and the sources stay in
! HBM between gas_optics and the solver; ECCKD_MIXED memory space of the C ABI).
! repeats: the block loop is run that many times and the best wall time is printed ("loop_seconds", bench.py reads it).
! fused = 1: one call ecckd%lw_fluxes(...) / ecckd%sw_fluxes(...) per block instead of gas_optics + rte_lw / rte_sw (the library's
!            fused paths).
! byband = 1: fluxes go through ty_fluxes_byband (per-band arrays; their sum over bands must reproduce the broadband
! fluxes, which are what output.bin holds either way).
!
! input.bin (little-endian, written by tests/test_fortran_shim.py): int32 ncol, nlay, ngas; per gas
! a 32-char name; then float64 arrays in Fortran order: plev(ncol,nlay+1), tlev(ncol,nlay+1),
! tlay(ncol,nlay), tsfc(ncol), sfc_emis(ncol) [lw] or mu0(ncol), albedo(ncol) [sw], then per gas
! vmr(ncol,nlay).  output.bin: flux_up(ncol,nlay+1), flux_dn(ncol,nlay+1).
! (The RFMIP netCDF files the reference drivers read are an FTP download and not available here.)
program ecckd_driver
  use, intrinsic :: iso_fortran_env, only: error_unit, int32
  use gas_optics_ecckd, only: ty_gas_optics_ecckd
  use mo_fluxes, only: ty_fluxes_broadband
  use mo_fluxes_byband, only: ty_fluxes_byband
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_optical_props, only: ty_optical_props_1scl, ty_optical_props_2str
  use mo_ecckd_device, only: ty_optical_props_1scl_dev, ty_optical_props_2str_dev, ty_source_func_lw_dev
  use mo_rte_kind, only: wp
  use mo_rte_lw, only: rte_lw, rte_set_solver_option
  use mo_rte_sw, only: rte_sw
  use mo_source_functions, only: ty_source_func_lw
  implicit none
  character(len=512) :: mode, ecckd_path, in_path, out_path, arg
  integer(int32) :: ncol, nlay, ngas
  integer :: block_size, n_quad_angles, nblocks, b, c0, c1, nc, i, ibnd, nbnd, u, dev_flag, repeats, rep, byband
  integer(kind=8) :: t0, t1, rate
  integer :: nc_alloc = -1, fused = 0
  real(wp) :: best, secs
  real(wp), dimension(:,:,:), allocatable, target :: bnd_up, bnd_dn
  logical :: top_at_1, lw
  character(len=32), dimension(:), allocatable :: gas_names
  real(wp), dimension(:,:), allocatable :: plev, tlev, tlay, play
  real(wp), dimension(:), allocatable :: tsfc, bc1, bc2
  real(wp), dimension(:,:,:), allocatable :: vmr
  real(wp), dimension(:,:), allocatable, target :: flux_up, flux_dn
  real(wp), dimension(:,:), allocatable :: sfc_spec, sfc_spec2, toa
  type(ty_gas_optics_ecckd) :: ecckd
  type(ty_gas_concs), dimension(:), allocatable :: gas_concs   ! one per block, filled before the loop (mo_rfmip_io.F90:177-263)
  class(ty_source_func_lw), allocatable :: source
  class(ty_optical_props_1scl), allocatable :: op1
  class(ty_optical_props_2str), allocatable :: op2
  type(ty_fluxes_broadband), target :: fluxes_bb
  type(ty_fluxes_byband), target :: fluxes_band
  class(ty_fluxes_broadband), pointer :: fluxes

  if (command_argument_count() < 4) then
    write(error_unit, *) "usage: ecckd_driver lw|sw ecckd_file input.bin output.bin [block_size] [n_quad_angles]"
    stop 1
  end if
  ! ECCKD_SOLVER_OPTION=name=value in the environment: one solver option set through the Fortran binding
  call get_environment_variable("ECCKD_SOLVER_OPTION", arg, status=i)
  if (i == 0 .and. len_trim(arg) > 0) then
    u = index(arg, "=")
    if (u < 2) call stop_on_err("ecckd_driver: ECCKD_SOLVER_OPTION must be name=value")
    read(arg(u + 1:), *, iostat=i) secs
    if (i /= 0) call stop_on_err("ecckd_driver: ECCKD_SOLVER_OPTION must be name=value")
    call stop_on_err(rte_set_solver_option(arg(1:u - 1), secs))
  end if
  call get_command_argument(1, mode)
  call get_command_argument(2, ecckd_path)
  call get_command_argument(3, in_path)
  call get_command_argument(4, out_path)
  block_size = 0
  n_quad_angles = 1
  if (command_argument_count() >= 5) then
    call get_command_argument(5, arg)
    read(arg, *) block_size
  end if
  if (command_argument_count() >= 6) then
    call get_command_argument(6, arg)
    read(arg, *) n_quad_angles
  end if
  dev_flag = 0
  if (command_argument_count() >= 7) then
    call get_command_argument(7, arg)
    read(arg, *) dev_flag
  end if
  repeats = 1
  if (command_argument_count() >= 8) then
    call get_command_argument(8, arg)
    read(arg, *) repeats
  end if
  byband = 0
  if (command_argument_count() >= 9) then
    call get_command_argument(9, arg)
    read(arg, *) byband
  end if
  if (command_argument_count() >= 10) then
    call get_command_argument(10, arg)
    read(arg, *) fused
  end if
  if (byband /= 0) then
    fluxes => fluxes_band
  else
    fluxes => fluxes_bb
  end if
  lw = trim(mode) == "lw"
  if (dev_flag /= 0) then
    allocate(ty_source_func_lw_dev :: source)
    allocate(ty_optical_props_1scl_dev :: op1)
    allocate(ty_optical_props_2str_dev :: op2)
  else
    allocate(ty_source_func_lw :: source)
    allocate(ty_optical_props_1scl :: op1)
    allocate(ty_optical_props_2str :: op2)
  end if

  open(newunit=u, file=trim(in_path), access="stream", form="unformatted", status="old")
  read(u) ncol, nlay, ngas
  allocate(gas_names(ngas))
  read(u) gas_names
  allocate(plev(ncol, nlay + 1), tlev(ncol, nlay + 1), tlay(ncol, nlay), play(ncol, nlay), tsfc(ncol), &
           bc1(ncol), bc2(ncol), vmr(ncol, nlay, ngas))
  read(u) plev, tlev, tlay, tsfc
  if (lw) then
    read(u) bc1
  else
    read(u) bc1, bc2
  end if
  read(u) vmr
  close(u)
  play = 0.5_wp * (plev(:, 1:nlay) + plev(:, 2:nlay + 1))
  if (block_size <= 0) block_size = ncol

  call stop_on_err(ecckd%load(trim(ecckd_path)))
  if (lw .neqv. ecckd%source_is_internal()) call stop_on_err("ecckd_driver: k-distribution file doesn't match lw/sw")
  nbnd = ecckd%get_nband()
  top_at_1 = play(1, 1) < play(1, nlay)                   ! ecckd_rfmip_lw.F90:85
  allocate(flux_up(ncol, nlay + 1), flux_dn(ncol, nlay + 1))
  nblocks = (ncol + block_size - 1) / block_size

  ! gas concentrations per block, as read_and_block_gases_ty prepares them before the reference's loop
  ! (mo_rfmip_io.F90:177-263); a field that is uniform goes in as a scalar, as RFMIP's well-mixed gases do
  allocate(gas_concs(nblocks))
  do b = 1, nblocks
    c0 = (b - 1) * block_size + 1
    c1 = min(ncol, b * block_size)
    call stop_on_err(gas_concs(b)%init(gas_names))
    do i = 1, ngas
      if (all(vmr(c0:c1, :, i) == vmr(c0, 1, i))) then
        call stop_on_err(gas_concs(b)%set_vmr(trim(gas_names(i)), vmr(c0, 1, i)))
      else
        call stop_on_err(gas_concs(b)%set_vmr(trim(gas_names(i)), vmr(c0:c1, :, i)))
      end if
    end do
  end do
  best = huge(1._wp)
  do rep = 1, max(1, repeats)
  call system_clock(t0, rate)
  do b = 1, nblocks
    c0 = (b - 1) * block_size + 1
    c1 = min(ncol, b * block_size)
    nc = c1 - c0 + 1
    fluxes%flux_up => flux_up(c0:c1, :)
    fluxes%flux_dn => flux_dn(c0:c1, :)
    if (byband /= 0) then
      if (allocated(bnd_up)) deallocate(bnd_up, bnd_dn)
      allocate(bnd_up(nc, nlay + 1, nbnd), bnd_dn(nc, nlay + 1, nbnd))
      fluxes_band%bnd_flux_up => bnd_up
      fluxes_band%bnd_flux_dn => bnd_dn
    end if
    if (allocated(sfc_spec)) deallocate(sfc_spec)
    allocate(sfc_spec(nbnd, nc))
    do i = 1, nc                                             ! ecckd_rfmip_lw.F90:112-116
      do ibnd = 1, nbnd
        sfc_spec(ibnd, i) = bc1(c0 + i - 1)
      end do
    end do
    if (lw .and. fused /= 0) then
      call stop_on_err(ecckd%lw_fluxes(plev(c0:c1, :), tlay(c0:c1, :), tsfc(c0:c1), tlev(c0:c1, :), gas_concs(b), top_at_1, &
                                       sfc_spec, flux_up(c0:c1, :), flux_dn(c0:c1, :), n_gauss_angles=n_quad_angles))
    else if (lw) then
      if (nc /= nc_alloc) then                                 ! (the reference allocates once, before its loop: :102-103)
        call stop_on_err(source%alloc(nc, nlay, ecckd))
        call stop_on_err(op1%alloc_1scl(nc, nlay, ecckd))
        nc_alloc = nc
      end if
      call stop_on_err(ecckd%gas_optics(play(c0:c1, :), plev(c0:c1, :), tlay(c0:c1, :), tsfc(c0:c1), gas_concs(b), &
                                        op1, source, tlev=tlev(c0:c1, :)))
      ! ecckd level sources hold one value per level (src/gas_optics_ecckd.f90:419-424): each level is read once
      call stop_on_err(rte_lw(op1, top_at_1, source, sfc_spec, fluxes, n_gauss_angles=n_quad_angles, &
                              lev_sources_shared=.true.))
    else if (fused /= 0) then
      if (allocated(sfc_spec2)) deallocate(sfc_spec2)
      allocate(sfc_spec2(nbnd, nc))
      do i = 1, nc
        sfc_spec(:, i) = bc2(c0 + i - 1)                     ! albedo, direct = diffuse (ecckd_rfmip_sw.F90:136-141)
        sfc_spec2(:, i) = bc2(c0 + i - 1)
      end do
      call stop_on_err(ecckd%sw_fluxes(plev(c0:c1, :), tlay(c0:c1, :), gas_concs(b), top_at_1, bc1(c0:c1), sfc_spec, sfc_spec2, &
                                       flux_up(c0:c1, :), flux_dn(c0:c1, :)))
    else
      if (allocated(sfc_spec2)) deallocate(sfc_spec2)
      if (allocated(toa)) deallocate(toa)
      allocate(sfc_spec2(nbnd, nc), toa(nc, ecckd%get_ngpt()))
      do i = 1, nc
        sfc_spec(:, i) = bc2(c0 + i - 1)                     ! albedo, direct = diffuse (ecckd_rfmip_sw.F90:136-141)
        sfc_spec2(:, i) = bc2(c0 + i - 1)
      end do
      if (nc /= nc_alloc) then
        call stop_on_err(op2%alloc_2str(nc, nlay, ecckd))
        nc_alloc = nc
      end if
      call stop_on_err(ecckd%gas_optics(play(c0:c1, :), plev(c0:c1, :), tlay(c0:c1, :), gas_concs(b), op2, toa))
      call stop_on_err(rte_sw(op2, top_at_1, bc1(c0:c1), toa, sfc_spec, sfc_spec2, fluxes))
    end if
    if (byband /= 0) then
      if (maxval(abs(sum(bnd_up, dim=3) - flux_up(c0:c1, :))) > 1.e-9_wp .or. &
          maxval(abs(sum(bnd_dn, dim=3) - flux_dn(c0:c1, :))) > 1.e-9_wp) &
        call stop_on_err("ecckd_driver: per-band fluxes do not add up to the broadband fluxes")
    end if
  end do
  call system_clock(t1)
  secs = real(t1 - t0, wp) / real(rate, wp)
  best = min(best, secs)
  end do
  write(error_unit, "(a,es12.5)") " ecckd_driver: loop_seconds ", best

  open(newunit=u, file=trim(out_path), access="stream", form="unformatted", status="replace")
  write(u) flux_up, flux_dn
  close(u)
  call ecckd%finalize()
  write(error_unit, *) "ecckd_driver: ", ncol, " columns in ", nblocks, " blocks done"

contains
  subroutine stop_on_err(msg)                                ! mo_simple_netcdf.F90:331-339
    character(len=*), intent(in) :: msg
    if (len_trim(msg) > 0) then
      write(error_unit, *) trim(msg)
      stop 1
    end if
  end subroutine stop_on_err
end program ecckd_driver
