! mo_rte_solvers.F90 -- modules mo_rte_lw / mo_rte_sw with RTE-RRTMGP's rte_lw / rte_sw interfaces as
! the reference drivers call them (example/rfmip-rad-irf/ecckd_rfmip_lw.F90:130-135,
! ecckd_rfmip_sw.F90:148-154), implemented by the MI355X solvers of librte_ecckd_hip.so
! (ecckd_rte_lw / ecckd_rte_sw: layer recursions + broadband g-point reduction fused).
! Marshalling only; no numerics here.
module mo_rte_lw
  use, intrinsic :: iso_c_binding
  use mo_rte_kind, only: wp
  use mo_optical_props, only: ty_optical_props_arry
  use mo_source_functions, only: ty_source_func_lw
  use mo_fluxes, only: ty_fluxes_broadband
  use gas_optics_ecckd, only: c_error_message
  implicit none
  private
  public :: rte_lw
  interface
    function c_rte_lw(device, ncol, nlay, ngpt, top_at_1, nmus, tau, lay_source, lev_inc, lev_dec, sfc_source, &
                      nband, band2gpt, sfc_emis, flux_up, flux_dn, memspace, stream) &
        bind(C, name="ecckd_rte_lw") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nmus, nband, memspace
      real(c_double), dimension(*), intent(in) :: tau, lay_source, lev_inc, lev_dec, sfc_source, sfc_emis
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_lw
    function c_rte_lw_shared(device, ncol, nlay, ngpt, top_at_1, nmus, tau, lay_source, lev_inc, lev_dec, sfc_source, &
                      nband, band2gpt, sfc_emis, flux_up, flux_dn, memspace, stream) &
        bind(C, name="ecckd_rte_lw_shared_levels") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nmus, nband, memspace
      real(c_double), dimension(*), intent(in) :: tau, lay_source, lev_inc, lev_dec, sfc_source, sfc_emis
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_lw_shared
  end interface
contains
  function rte_lw(optical_props, top_at_1, sources, sfc_emis, fluxes, n_gauss_angles, device, lev_sources_shared) &
      result(error_msg)
    class(ty_optical_props_arry), intent(in) :: optical_props
    logical, intent(in) :: top_at_1
    type(ty_source_func_lw), intent(in) :: sources
    real(wp), dimension(:,:), intent(in) :: sfc_emis        !< (nband, ncol)
    type(ty_fluxes_broadband), intent(inout) :: fluxes
    integer, optional, intent(in) :: n_gauss_angles
    integer, optional, intent(in) :: device
    !> .true.: lev_source_inc(:,l,:) == lev_source_dec(:,l+1,:), as ecckd's gas_optics writes them
    !> (src/gas_optics_ecckd.f90:419-424); each level is then read once (ecckd_rte_lw_shared_levels)
    logical, optional, intent(in) :: lev_sources_shared
    character(len=128) :: error_msg
    integer :: ncol, nlay, ngpt, nmus, dev
    logical :: shared
    integer(c_int) :: rc
    real(wp), dimension(:,:), allocatable :: up, dn
    error_msg = ""
    ncol = size(optical_props%tau, 1)
    nlay = size(optical_props%tau, 2)
    ngpt = size(optical_props%tau, 3)
    nmus = 1
    if (present(n_gauss_angles)) nmus = n_gauss_angles
    dev = 0
    if (present(device)) dev = device
    if (size(sfc_emis, 1) /= optical_props%get_nband() .or. size(sfc_emis, 2) /= ncol) then
      error_msg = "rte_lw: sfc_emis inconsistently sized"
      return
    end if
    if (.not. associated(fluxes%flux_up) .or. .not. associated(fluxes%flux_dn)) then
      error_msg = "rte_lw: fluxes%flux_up and fluxes%flux_dn must be associated"
      return
    end if
    allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1))
    shared = .false.
    if (present(lev_sources_shared)) shared = lev_sources_shared
    if (shared) then
      rc = c_rte_lw_shared(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                           merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), optical_props%tau, sources%lay_source, &
                           sources%lev_source_inc, sources%lev_source_dec, sources%sfc_source, &
                           int(optical_props%get_nband(), c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                           sfc_emis, up, dn, 0_c_int, c_null_ptr)
    else
      rc = c_rte_lw(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                    merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), optical_props%tau, sources%lay_source, &
                    sources%lev_source_inc, sources%lev_source_dec, sources%sfc_source, &
                    int(optical_props%get_nband(), c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                    sfc_emis, up, dn, 0_c_int, c_null_ptr)
    end if
    if (rc /= 0) then
      error_msg = c_error_message()
      return
    end if
    fluxes%flux_up(:, :) = up
    fluxes%flux_dn(:, :) = dn
  end function rte_lw
end module mo_rte_lw


module mo_rte_sw
  use, intrinsic :: iso_c_binding
  use mo_rte_kind, only: wp
  use mo_optical_props, only: ty_optical_props_arry, ty_optical_props_2str
  use mo_fluxes, only: ty_fluxes_broadband
  use gas_optics_ecckd, only: c_error_message
  implicit none
  private
  public :: rte_sw
  interface
    function c_rte_sw(device, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, toa, nband, band2gpt, alb_dir, &
                      alb_dif, flux_up, flux_dn, flux_dir, memspace, stream) bind(C, name="ecckd_rte_sw") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nband, memspace
      real(c_double), dimension(*), intent(in) :: tau, ssa, g, mu0, toa, alb_dir, alb_dif
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn, flux_dir
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_sw
  end interface
contains
  function rte_sw(optical_props, top_at_1, mu0, inc_flux, sfc_alb_dir, sfc_alb_dif, fluxes, device) &
      result(error_msg)
    class(ty_optical_props_arry), intent(in) :: optical_props
    logical, intent(in) :: top_at_1
    real(wp), dimension(:), intent(in) :: mu0                   !< (ncol)
    real(wp), dimension(:,:), intent(in) :: inc_flux            !< (ncol, ngpt)
    real(wp), dimension(:,:), intent(in) :: sfc_alb_dir, sfc_alb_dif   !< (nband, ncol)
    type(ty_fluxes_broadband), intent(inout) :: fluxes
    integer, optional, intent(in) :: device
    character(len=128) :: error_msg
    integer :: ncol, nlay, ngpt, dev
    integer(c_int) :: rc
    real(wp), dimension(:,:), allocatable :: up, dn, dir
    error_msg = ""
    dev = 0
    if (present(device)) dev = device
    select type (optical_props)
      type is (ty_optical_props_2str)
        ncol = size(optical_props%tau, 1)
        nlay = size(optical_props%tau, 2)
        ngpt = size(optical_props%tau, 3)
        if (.not. associated(fluxes%flux_up) .or. .not. associated(fluxes%flux_dn)) then
          error_msg = "rte_sw: fluxes%flux_up and fluxes%flux_dn must be associated"
          return
        end if
        allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1), dir(ncol, nlay + 1))
        rc = c_rte_sw(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                      merge(1_c_int, 0_c_int, top_at_1), optical_props%tau, optical_props%ssa, optical_props%g, &
                      mu0, inc_flux, int(optical_props%get_nband(), c_int), &
                      int(optical_props%get_band_lims_gpoint(), c_int), sfc_alb_dir, sfc_alb_dif, up, dn, dir, &
                      0_c_int, c_null_ptr)
        if (rc /= 0) then
          error_msg = c_error_message()
          return
        end if
        fluxes%flux_up(:, :) = up
        fluxes%flux_dn(:, :) = dn
        if (associated(fluxes%flux_dn_dir)) fluxes%flux_dn_dir(:, :) = dir
      class default
        error_msg = "rte_sw: two-stream optical properties required"
    end select
  end function rte_sw
end module mo_rte_sw
