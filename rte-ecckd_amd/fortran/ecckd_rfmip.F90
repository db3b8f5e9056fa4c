! ecckd_rfmip.F90 -- RFMIP RAD-IRF drivers over the MI355X path; one source, two executables with the
! reference's names and command line (example/rfmip-rad-irf/Makefile:12-40, utils.f90:26-37):
!
!     ecckd_rfmip_lw rfmip_file ecckd_file [-f 1|2] [-p 1|2] [-b block] [-n nblocks] [-d]   (default build)
!     ecckd_rfmip_sw rfmip_file ecckd_file [-f 1|2]          [-b block] [-n nblocks] [-d]   (-DSHORTWAVE)
!
! -d: device-resident mode.  optical_props / source are the device twins of mo_ecckd_device: tau and the
! sources stay in the GPU's memory between gas_optics and rte_lw / rte_sw, only the atmosphere goes in and the
! fluxes come out (the C ABI's ECCKD_MIXED memory space).  Same calls, same fluxes bit for bit.
!
! Same steps as ecckd_rfmip_lw.F90:38-140 / ecckd_rfmip_sw.F90:40-166: sizes, output names
! r{l,s}{u,d}_Efx_RTE-ecckd_rad-irf_r1i1p<p>f<f>_gn.nc, gas names by forcing index, blocked inputs,
! load the k-distribution, clamp the top level to the table's minimum pressure, per block gas_optics
! then rte_lw / rte_sw, fluxes written into the pre-existing output files.
! Deliberate differences, both switchable: the default block is ALL columns at once (the reference
! hard-codes block_size = 1, :39; use -b 1 for that) and every block is processed (the reference's loop
! bound is the literal 1700, :107; use -n 1700 for that).
program ecckd_rfmip
  use, intrinsic :: iso_fortran_env, only: error_unit
  use gas_optics_ecckd, only: ty_gas_optics_ecckd
  use load_coefficients, only: load_and_init
  use mo_fluxes, only: ty_fluxes_broadband
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_rte_kind, only: wp
  use rfmip_io
  use simple_netcdf, only: stop_on_err
  use utils, only: determine_gas_names, parse_args
#ifdef SHORTWAVE
  use mo_optical_props, only: ty_optical_props_2str
  use mo_ecckd_device, only: ty_optical_props_2str_dev
  use mo_rte_sw, only: rte_sw
#else
  use mo_optical_props, only: ty_optical_props_1scl
  use mo_ecckd_device, only: ty_optical_props_1scl_dev, ty_source_func_lw_dev
  use mo_rte_lw, only: rte_lw
  use mo_source_functions, only: ty_source_func_lw
#endif
  implicit none
  character(len=512) :: rfmip_path, ecckd_path
  character(len=132) :: file_dn, file_up
  character(len=1) :: fchar, pchar
  character(len=32), dimension(6) :: kdist_names, rfmip_names
  integer :: ncol, nlay, nexp, nbnd, nblocks, ndo, block_size, max_blocks, forcing_index, physics_index
  integer :: b, i
  logical :: top_at_1, device_resident
  real(wp), dimension(:,:,:), allocatable :: p_lay, p_lev, t_lay, t_lev
  real(wp), dimension(:,:,:), allocatable, target :: flux_up, flux_dn
  real(wp), dimension(:,:), allocatable :: bc_spec
  type(ty_gas_optics_ecckd) :: ecckd
  type(ty_gas_concs), dimension(:), allocatable :: gases
  type(ty_fluxes_broadband) :: fluxes
#ifdef SHORTWAVE
  real(wp), parameter :: deg_to_rad = acos(-1._wp) / 180._wp
  real(wp), dimension(:,:), allocatable :: albedo, tsi, sza, toa_flux
  real(wp), dimension(:), allocatable :: mu0, def_tsi
  logical, dimension(:,:), allocatable :: usecol
  integer :: ngpt
  class(ty_optical_props_2str), allocatable :: optical_props
#else
  integer :: n_quad_angles
  real(wp), dimension(:,:), allocatable :: sfc_emis, sfc_t
  class(ty_optical_props_1scl), allocatable :: optical_props
  class(ty_source_func_lw), allocatable :: source
#endif

  call parse_args(rfmip_path, ecckd_path, forcing_index, physics_index, block_size, max_blocks, device_resident)
  call read_size(rfmip_path, ncol, nlay, nexp)
  if (block_size <= 0) block_size = ncol * nexp
  if (mod(ncol * nexp, block_size) /= 0) &
    call stop_on_err("ecckd_rfmip: number of columns doesn't fit evenly into blocks.")
  nblocks = (ncol * nexp) / block_size
  ndo = nblocks
  if (max_blocks > 0) ndo = min(nblocks, max_blocks)
  write(error_unit, *) "Using ", nblocks, " blocks of size ", block_size

  write(fchar, "(i1)") forcing_index
  write(pchar, "(i1)") physics_index
#ifdef SHORTWAVE
  file_dn = "rsd_Efx_RTE-ecckd_rad-irf_r1i1p1f" // fchar // "_gn.nc"      ! ecckd_rfmip_sw.F90:56-57
  file_up = "rsu_Efx_RTE-ecckd_rad-irf_r1i1p1f" // fchar // "_gn.nc"
#else
  file_dn = "rld_Efx_RTE-ecckd_rad-irf_r1i1p" // pchar // "f" // fchar // "_gn.nc"   ! ecckd_rfmip_lw.F90:59-62
  file_up = "rlu_Efx_RTE-ecckd_rad-irf_r1i1p" // pchar // "f" // fchar // "_gn.nc"
  n_quad_angles = merge(3, 1, physics_index == 2)                          ! :40-44
#endif
  call determine_gas_names(forcing_index, kdist_names, rfmip_names)

  call read_and_block_pt(rfmip_path, block_size, p_lay, p_lev, t_lay, t_lev)
#ifdef SHORTWAVE
  call read_and_block_sw_bc(rfmip_path, block_size, albedo, tsi, sza)
#else
  call read_and_block_lw_bc(rfmip_path, block_size, sfc_emis, sfc_t)
#endif
  call read_and_block_gases_ty(rfmip_path, block_size, kdist_names, rfmip_names, gases)

  call load_and_init(ecckd, trim(ecckd_path), gases(1))
#ifdef SHORTWAVE
  if (.not. ecckd%source_is_external()) call stop_on_err("ecckd_rfmip_sw: k-distribution file isn't for shortwave.")
  ngpt = ecckd%get_ngpt()
#else
  if (.not. ecckd%source_is_internal()) call stop_on_err("ecckd_rfmip_lw: k-distribution file isn't for longwave.")
#endif
  nbnd = ecckd%get_nband()

  top_at_1 = p_lay(1, 1, 1) < p_lay(1, nlay, 1)
  ! the top level of the RFMIP file is 1e-3 Pa: pretend the layer is a bit less deep (input sanitising
  ! of the reference drivers, ecckd_rfmip_lw.F90:87-94)
  if (top_at_1) then
    p_lev(:, 1, :) = ecckd%get_press_min() + epsilon(ecckd%get_press_min())
  else
    p_lev(:, nlay + 1, :) = ecckd%get_press_min() + epsilon(ecckd%get_press_min())
  end if

  allocate(flux_up(block_size, nlay + 1, nblocks), flux_dn(block_size, nlay + 1, nblocks), bc_spec(nbnd, block_size))
  flux_up = 0._wp
  flux_dn = 0._wp
#ifdef SHORTWAVE
  allocate(mu0(block_size), def_tsi(block_size), toa_flux(block_size, ngpt), usecol(block_size, nblocks))
  if (device_resident) then
    allocate(ty_optical_props_2str_dev :: optical_props)
  else
    allocate(ty_optical_props_2str :: optical_props)
  end if
  call stop_on_err(optical_props%alloc_2str(block_size, nlay, ecckd))
  usecol = sza < 90._wp - 2._wp * spacing(90._wp)                          ! ecckd_rfmip_sw.F90:106-108
#else
  if (device_resident) then
    allocate(ty_source_func_lw_dev :: source)
    allocate(ty_optical_props_1scl_dev :: optical_props)
  else
    allocate(ty_source_func_lw :: source)
    allocate(ty_optical_props_1scl :: optical_props)
  end if
  call stop_on_err(source%alloc(block_size, nlay, ecckd))
  call stop_on_err(optical_props%alloc_1scl(block_size, nlay, ecckd))
#endif

  do b = 1, ndo
    fluxes%flux_up => flux_up(:, :, b)
    fluxes%flux_dn => flux_dn(:, :, b)
#ifdef SHORTWAVE
    call stop_on_err(ecckd%gas_optics(p_lay(:, :, b), p_lev(:, :, b), t_lay(:, :, b), gases(b), optical_props, toa_flux))
    def_tsi = sum(toa_flux, dim=2)                                          ! :126-133 renormalise to the file's TSI
    do i = 1, block_size
      toa_flux(i, :) = toa_flux(i, :) * tsi(i, b) / def_tsi(i)
      bc_spec(:, i) = albedo(i, b)
      mu0(i) = merge(cos(sza(i, b) * deg_to_rad), 1._wp, usecol(i, b))     ! :143-145
    end do
    call stop_on_err(rte_sw(optical_props, top_at_1, mu0, toa_flux, bc_spec, bc_spec, fluxes))
    do i = 1, block_size                                                    ! :156-161 night columns
      if (.not. usecol(i, b)) then
        flux_up(i, :, b) = 0._wp
        flux_dn(i, :, b) = 0._wp
      end if
    end do
#else
    do i = 1, block_size
      bc_spec(:, i) = sfc_emis(i, b)                                        ! :112-116
    end do
    call stop_on_err(ecckd%gas_optics(p_lay(:, :, b), p_lev(:, :, b), t_lay(:, :, b), sfc_t(:, b), gases(b), &
                                      optical_props, source, tlev=t_lev(:, :, b)))
    call stop_on_err(rte_lw(optical_props, top_at_1, source, bc_spec, fluxes, n_gauss_angles=n_quad_angles))
#endif
  end do

#ifdef SHORTWAVE
  call unblock_and_write(trim(file_up), "rsu", flux_up)
  call unblock_and_write(trim(file_dn), "rsd", flux_dn)
#else
  call unblock_and_write(trim(file_up), "rlu", flux_up)
  call unblock_and_write(trim(file_dn), "rld", flux_dn)
#endif
  call ecckd%finalize()
end program ecckd_rfmip
