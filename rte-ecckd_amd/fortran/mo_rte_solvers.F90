! mo_rte_solvers.F90 -- modules mo_rte_lw / mo_rte_sw with RTE-RRTMGP's rte_lw / rte_sw interfaces as
! the reference drivers call them (example/rfmip-rad-irf/ecckd_rfmip_lw.F90:130-135,
! ecckd_rfmip_sw.F90:148-154), implemented by the MI355X solvers of librte_ecckd_hip.so
! (ecckd_rte_lw / ecckd_rte_sw: layer recursions + broadband g-point reduction fused).
! Marshalling only; no numerics here.
!   * optical_props / sources may be the host containers (ECCKD_HOST: everything staged over PCIe) or the
!     device-resident twins of mo_ecckd_device (ECCKD_MIXED: the 3-D arrays are already in HBM, only the
!     boundary conditions go in and the fluxes come out);
!   * fluxes may be ty_fluxes_broadband (what the reference drivers pass) or ty_fluxes_byband (spectral output:
!     one solver pass per band);
!   * rte_lw takes RTE-RRTMGP's optional inc_flux(ncol,ngpt) and n_gauss_angles.
module mo_rte_lw
  use, intrinsic :: iso_c_binding
  use mo_rte_kind, only: wp
  use mo_optical_props, only: ty_optical_props_arry
  use mo_source_functions, only: ty_source_func_lw
  use mo_fluxes, only: ty_fluxes_broadband
  use mo_fluxes_byband, only: ty_fluxes_byband
  use mo_ecckd_device, only: ty_optical_props_1scl_dev, ty_source_func_lw_dev, ECCKD_HOST, ECCKD_MIXED
  use gas_optics_ecckd, only: c_error_message, c_loc_3d, c_loc_2d
  implicit none
  private
  public :: rte_lw, rte_set_solver_option
  interface
    ! version switches of the un-pinned RTE-RRTMGP solvers and implementation choices (include/ecckd_hip.h:
    ! ecckd_set_solver_option): "lw_tau_thresh", "lw_series_terms", "lw_inc_flux_isotropic", "sw_k_floor",
    ! "sw_dir_clamp", "lw_solver", "lw_split_seg", "gas_merge_scalars", "lw_tail_split", "sw_tail_split"
    function c_set_solver_option(name, value) bind(C, name="ecckd_set_solver_option") result(rc)
      import c_int, c_double, c_char
      character(kind=c_char), dimension(*), intent(in) :: name
      real(c_double), value :: value
      integer(c_int) :: rc
    end function c_set_solver_option
    function c_rte_lw(device, ncol, nlay, ngpt, top_at_1, nmus, tau, lay_source, lev_inc, lev_dec, sfc_source, &
                      nband, band2gpt, sfc_emis, inc_flux, flux_up, flux_dn, memspace, stream) &
        bind(C, name="ecckd_rte_lw_inc_flux") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nmus, nband, memspace
      type(c_ptr), value :: tau, lay_source, lev_inc, lev_dec, sfc_source   ! host or device (ECCKD_MIXED)
      real(c_double), dimension(*), intent(in) :: sfc_emis
      type(c_ptr), value :: inc_flux                                         ! host (ncol,ngpt) or null
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
      type(c_ptr), value :: tau, lay_source, lev_inc, lev_dec, sfc_source
      real(c_double), dimension(*), intent(in) :: sfc_emis
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_lw_shared
    function c_rte_lw_byband(device, ncol, nlay, ngpt, top_at_1, nmus, tau, lay_source, lev_inc, lev_dec, sfc_source, &
                             nband, band2gpt, sfc_emis, bnd_up, bnd_dn, flux_up, flux_dn, memspace, stream) &
        bind(C, name="ecckd_rte_lw_byband") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nmus, nband, memspace
      type(c_ptr), value :: tau, lay_source, lev_inc, lev_dec, sfc_source
      real(c_double), dimension(*), intent(in) :: sfc_emis
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: bnd_up, bnd_dn
      type(c_ptr), value :: flux_up, flux_dn                                 ! host (ncol,nlay+1) or null
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_lw_byband
  end interface
contains
  !> Sets a solver option of the library for the whole process (both solvers; thread-safe).  A host that links a later
  !> RTE-RRTMGP release than the v1.5 era the defaults follow sets the matching forms once at start-up, e.g.
  !> `error_msg = rte_set_solver_option("sw_dir_clamp", 1._wp)`.  Empty result = success, as everywhere in RTE.
  function rte_set_solver_option(name, value) result(error_msg)
    character(len=*), intent(in) :: name
    real(wp), intent(in) :: value
    character(len=128) :: error_msg
    error_msg = ""
    if (c_set_solver_option(trim(name) // c_null_char, real(value, c_double)) /= 0) error_msg = c_error_message()
  end function rte_set_solver_option

  function rte_lw(optical_props, top_at_1, sources, sfc_emis, fluxes, inc_flux, n_gauss_angles, device, &
                  lev_sources_shared) result(error_msg)
    class(ty_optical_props_arry), intent(in) :: optical_props
    logical, intent(in) :: top_at_1
    class(ty_source_func_lw), intent(in) :: sources
    real(wp), dimension(:,:), intent(in) :: sfc_emis        !< (nband, ncol)
    class(ty_fluxes_broadband), intent(inout) :: fluxes
    real(wp), dimension(:,:), intent(in), target, optional :: inc_flux   !< (ncol, ngpt) incident flux at the top
    integer, optional, intent(in) :: n_gauss_angles
    integer, optional, intent(in) :: device
    !> .true.: lev_source_inc(:,l,:) == lev_source_dec(:,l+1,:), as ecckd's gas_optics writes them
    !> (src/gas_optics_ecckd.f90:419-424); each level is then read once (ecckd_rte_lw_shared_levels)
    logical, optional, intent(in) :: lev_sources_shared
    character(len=128) :: error_msg
    integer :: ncol, nlay, ngpt, nmus, dev, nband
    logical :: shared, in_place
    integer(c_int) :: rc, memspace
    type(c_ptr) :: p_tau, p_lay, p_inc, p_dec, p_sfc, p_incf
    real(wp), dimension(:,:), allocatable, target :: up, dn, incf
    real(wp), dimension(:,:,:), allocatable :: bup, bdn
    error_msg = ""
    nmus = 1
    if (present(n_gauss_angles)) nmus = n_gauss_angles
    dev = 0
    if (present(device)) dev = device
    memspace = ECCKD_HOST
    select type (optical_props)
      class is (ty_optical_props_1scl_dev)
        select type (sources)
          class is (ty_source_func_lw_dev)
            memspace = ECCKD_MIXED
            ncol = optical_props%ncol
            nlay = optical_props%nlay
            ngpt = optical_props%get_ngpt()
            if (.not. present(device)) dev = optical_props%device
            p_tau = optical_props%d_tau
            p_lay = sources%d_lay_source
            p_inc = sources%d_lev_source_inc
            p_dec = sources%d_lev_source_dec
            p_sfc = sources%d_sfc_source
          class default
            error_msg = "rte_lw: device-resident optical_props needs device-resident sources"
            return
        end select
      class default
        ncol = size(optical_props%tau, 1)
        nlay = size(optical_props%tau, 2)
        ngpt = size(optical_props%tau, 3)
        p_tau = c_loc_3d(optical_props%tau)
        p_lay = c_loc_3d(sources%lay_source)
        p_inc = c_loc_3d(sources%lev_source_inc)
        p_dec = c_loc_3d(sources%lev_source_dec)
        p_sfc = c_loc_2d(sources%sfc_source)
    end select
    nband = optical_props%get_nband()
    if (size(sfc_emis, 1) /= nband .or. size(sfc_emis, 2) /= ncol) then
      error_msg = "rte_lw: sfc_emis inconsistently sized"
      return
    end if
    p_incf = c_null_ptr
    if (present(inc_flux)) then
      if (size(inc_flux, 1) /= ncol .or. size(inc_flux, 2) /= ngpt) then
        error_msg = "rte_lw: incident flux inconsistently sized"
        return
      end if
      allocate(incf(ncol, ngpt))
      incf = inc_flux
      p_incf = c_loc(incf(1, 1))
    end if
    shared = .false.
    if (present(lev_sources_shared)) shared = lev_sources_shared
    select type (fluxes)
      class is (ty_fluxes_byband)
        if (.not. associated(fluxes%bnd_flux_up) .or. .not. associated(fluxes%bnd_flux_dn)) then
          error_msg = "rte_lw: fluxes%bnd_flux_up and fluxes%bnd_flux_dn must be associated"
          return
        end if
        if (present(inc_flux)) then
          error_msg = "rte_lw: inc_flux with per-band fluxes is not implemented"
          return
        end if
        allocate(bup(ncol, nlay + 1, nband), bdn(ncol, nlay + 1, nband), up(ncol, nlay + 1), dn(ncol, nlay + 1))
        rc = c_rte_lw_byband(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                             merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), p_tau, p_lay, p_inc, p_dec, p_sfc, &
                             int(nband, c_int), int(optical_props%get_band_lims_gpoint(), c_int), sfc_emis, bup, bdn, &
                             c_loc(up(1, 1)), c_loc(dn(1, 1)), memspace, c_null_ptr)
        if (rc /= 0) then
          error_msg = c_error_message()
          return
        end if
        fluxes%bnd_flux_up(:, :, :) = bup
        fluxes%bnd_flux_dn(:, :, :) = bdn
        if (associated(fluxes%flux_up)) fluxes%flux_up(:, :) = up
        if (associated(fluxes%flux_dn)) fluxes%flux_dn(:, :) = dn
        return
    end select
    if (.not. associated(fluxes%flux_up) .or. .not. associated(fluxes%flux_dn)) then
      error_msg = "rte_lw: fluxes%flux_up and fluxes%flux_dn must be associated"
      return
    end if
    if (is_contiguous(fluxes%flux_up) .and. is_contiguous(fluxes%flux_dn) .and. size(fluxes%flux_up, 1) == ncol .and. &
        size(fluxes%flux_dn, 1) == ncol .and. size(fluxes%flux_up, 2) == nlay + 1 .and. size(fluxes%flux_dn, 2) == nlay + 1) then
      in_place = .true.        ! the library writes straight into the caller's flux arrays
    else
      in_place = .false.
      allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1))
    end if
    if (in_place) then
      if (shared .and. .not. present(inc_flux)) then
        rc = c_rte_lw_shared(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                             merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), p_tau, p_lay, p_inc, p_dec, p_sfc, &
                             int(nband, c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                             sfc_emis, fluxes%flux_up, fluxes%flux_dn, memspace, c_null_ptr)
      else
        rc = c_rte_lw(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                      merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), p_tau, p_lay, p_inc, p_dec, p_sfc, &
                      int(nband, c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                      sfc_emis, p_incf, fluxes%flux_up, fluxes%flux_dn, memspace, c_null_ptr)
      end if
      if (rc /= 0) error_msg = c_error_message()
      return
    end if
    if (shared .and. .not. present(inc_flux)) then
      rc = c_rte_lw_shared(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                           merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), p_tau, p_lay, p_inc, p_dec, p_sfc, &
                           int(nband, c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                           sfc_emis, up, dn, memspace, c_null_ptr)
    else
      rc = c_rte_lw(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                    merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), p_tau, p_lay, p_inc, p_dec, p_sfc, &
                    int(nband, c_int), int(optical_props%get_band_lims_gpoint(), c_int), &
                    sfc_emis, p_incf, up, dn, memspace, c_null_ptr)
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
  use mo_fluxes_byband, only: ty_fluxes_byband
  use mo_ecckd_device, only: ty_optical_props_2str_dev, ECCKD_HOST, ECCKD_MIXED
  use gas_optics_ecckd, only: c_error_message, c_loc_3d
  implicit none
  private
  public :: rte_sw
  interface
    function c_rte_sw(device, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, toa, nband, band2gpt, alb_dir, &
                      alb_dif, flux_up, flux_dn, flux_dir, memspace, stream) bind(C, name="ecckd_rte_sw") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nband, memspace
      type(c_ptr), value :: tau, ssa, g                                      ! host or device (ECCKD_MIXED)
      real(c_double), dimension(*), intent(in) :: mu0, toa, alb_dir, alb_dif
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn, flux_dir
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_sw
    function c_rte_sw_byband(device, ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, toa, nband, band2gpt, alb_dir, &
                             alb_dif, bnd_up, bnd_dn, bnd_dir, flux_up, flux_dn, flux_dir, memspace, stream) &
        bind(C, name="ecckd_rte_sw_byband") result(rc)
      import c_int, c_double, c_ptr
      integer(c_int), value :: device, ncol, nlay, ngpt, top_at_1, nband, memspace
      type(c_ptr), value :: tau, ssa, g
      real(c_double), dimension(*), intent(in) :: mu0, toa, alb_dir, alb_dif
      integer(c_int), dimension(*), intent(in) :: band2gpt
      real(c_double), dimension(*), intent(inout) :: bnd_up, bnd_dn, bnd_dir
      type(c_ptr), value :: flux_up, flux_dn, flux_dir
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_rte_sw_byband
  end interface
contains
  function rte_sw(optical_props, top_at_1, mu0, inc_flux, sfc_alb_dir, sfc_alb_dif, fluxes, device) &
      result(error_msg)
    class(ty_optical_props_arry), intent(in) :: optical_props
    logical, intent(in) :: top_at_1
    real(wp), dimension(:), intent(in) :: mu0                   !< (ncol)
    real(wp), dimension(:,:), intent(in) :: inc_flux            !< (ncol, ngpt)
    real(wp), dimension(:,:), intent(in) :: sfc_alb_dir, sfc_alb_dif   !< (nband, ncol)
    class(ty_fluxes_broadband), intent(inout) :: fluxes
    integer, optional, intent(in) :: device
    character(len=128) :: error_msg
    integer :: ncol, nlay, ngpt, dev, nband
    integer(c_int) :: rc, memspace
    type(c_ptr) :: p_tau, p_ssa, p_g
    real(wp), dimension(:,:), allocatable, target :: up, dn, dir
    real(wp), dimension(:,:,:), allocatable :: bup, bdn, bdir
    error_msg = ""
    dev = 0
    if (present(device)) dev = device
    memspace = ECCKD_HOST
    select type (optical_props)
      class is (ty_optical_props_2str_dev)
        memspace = ECCKD_MIXED
        ncol = optical_props%ncol
        nlay = optical_props%nlay
        ngpt = optical_props%get_ngpt()
        if (.not. present(device)) dev = optical_props%device
        p_tau = optical_props%d_tau
        p_ssa = optical_props%d_ssa
        p_g = optical_props%d_g
      class is (ty_optical_props_2str)
        ncol = size(optical_props%tau, 1)
        nlay = size(optical_props%tau, 2)
        ngpt = size(optical_props%tau, 3)
        p_tau = c_loc_3d(optical_props%tau)
        p_ssa = c_loc_3d(optical_props%ssa)
        p_g = c_loc_3d(optical_props%g)
      class default
        error_msg = "rte_sw: two-stream optical properties required"
        return
    end select
    nband = optical_props%get_nband()
    allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1), dir(ncol, nlay + 1))
    select type (fluxes)
      class is (ty_fluxes_byband)
        if (.not. associated(fluxes%bnd_flux_up) .or. .not. associated(fluxes%bnd_flux_dn)) then
          error_msg = "rte_sw: fluxes%bnd_flux_up and fluxes%bnd_flux_dn must be associated"
          return
        end if
        allocate(bup(ncol, nlay + 1, nband), bdn(ncol, nlay + 1, nband), bdir(ncol, nlay + 1, nband))
        rc = c_rte_sw_byband(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                             merge(1_c_int, 0_c_int, top_at_1), p_tau, p_ssa, p_g, mu0, inc_flux, int(nband, c_int), &
                             int(optical_props%get_band_lims_gpoint(), c_int), sfc_alb_dir, sfc_alb_dif, bup, bdn, bdir, &
                             c_loc(up(1, 1)), c_loc(dn(1, 1)), c_loc(dir(1, 1)), memspace, c_null_ptr)
        if (rc /= 0) then
          error_msg = c_error_message()
          return
        end if
        fluxes%bnd_flux_up(:, :, :) = bup
        fluxes%bnd_flux_dn(:, :, :) = bdn
        if (associated(fluxes%bnd_flux_dn_dir)) fluxes%bnd_flux_dn_dir(:, :, :) = bdir
        if (associated(fluxes%flux_up)) fluxes%flux_up(:, :) = up
        if (associated(fluxes%flux_dn)) fluxes%flux_dn(:, :) = dn
        if (associated(fluxes%flux_dn_dir)) fluxes%flux_dn_dir(:, :) = dir
        return
    end select
    if (.not. associated(fluxes%flux_up) .or. .not. associated(fluxes%flux_dn)) then
      error_msg = "rte_sw: fluxes%flux_up and fluxes%flux_dn must be associated"
      return
    end if
    rc = c_rte_sw(int(dev, c_int), int(ncol, c_int), int(nlay, c_int), int(ngpt, c_int), &
                  merge(1_c_int, 0_c_int, top_at_1), p_tau, p_ssa, p_g, mu0, inc_flux, int(nband, c_int), &
                  int(optical_props%get_band_lims_gpoint(), c_int), sfc_alb_dir, sfc_alb_dif, up, dn, dir, &
                  memspace, c_null_ptr)
    if (rc /= 0) then
      error_msg = c_error_message()
      return
    end if
    fluxes%flux_up(:, :) = up
    fluxes%flux_dn(:, :) = dn
    if (associated(fluxes%flux_dn_dir)) fluxes%flux_dn_dir(:, :) = dir
  end function rte_sw
end module mo_rte_sw
