! gas_optics_ecckd.F90 -- drop-in replacement of the reference module `gas_optics_ecckd`
! (src/gas_optics_ecckd.f90): same module name, same public type ty_gas_optics_ecckd with the same
! type-bound procedures and argument lists, but the arithmetic runs on an MI355X through the C ABI
! of librte_ecckd_hip.so (include/ecckd_hip.h).  This file contains NO numerics: every procedure is
! a marshalling shim over one `bind(C)` entry point.
!
!   reference                                   here
!   ------------------------------------------  ------------------------------------------------
!   type(ty_gas_optics_ecckd) public members    opaque device-resident handle (type(c_ptr))
!   load_and_init(ecckd, file, gases)           err = ecckd%load(file [, available_gases] [, device])
!     (mo_load_coefficients.F90:19)             -> ecckd_model_load (own netCDF-3 reader)
!   ecckd%gas_optics(play,plev,tlay,tsfc,       same call                 -> ecckd_gas_optics_lw
!          gas_desc,optical_props,sources,tlev=)  (gas_optics_int, :381)
!   ecckd%gas_optics(play,plev,tlay,gas_desc,   same call                 -> ecckd_gas_optics_sw
!          optical_props,toa_src)                 (gas_optics_ext, :431)
!   get_ngas, get_gases, source_is_internal/    same names                -> ecckd_model_get_*
!   external, get_press_min/max, get_temp_min/max
!
! Differences a caller can observe: `this` carries no tables (they live on the GPU); sources%
! lay_source is never reallocated (the reference reallocates it, :266-269 via :407); a failing
! C call returns its message through the same character(len=128) result.
module gas_optics_ecckd
  use, intrinsic :: iso_c_binding
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_ecckd_device, only: ty_optical_props_1scl_dev, ty_optical_props_2str_dev, ty_source_func_lw_dev, &
                             ECCKD_HOST, ECCKD_MIXED
  use mo_gas_optics, only: ty_gas_optics
  use mo_optical_props, only: ty_optical_props_arry, ty_optical_props_2str
  use mo_rte_kind, only: wp
  use mo_source_functions, only: ty_source_func_lw
  implicit none
  private

  integer, parameter, public :: none_ = 0            ! src/gas_optics_ecckd.f90:54-57
  integer, parameter, public :: linear = 1
  integer, parameter, public :: look_up_table = 2
  integer, parameter, public :: relative_linear = 3
  integer, parameter :: name_len = 32

  type, extends(ty_gas_optics), public :: ty_gas_optics_ecckd
    type(c_ptr) :: handle = c_null_ptr     !< ecckd_model_t*, owns the device-resident tables
  contains
    procedure, public :: load
    procedure, public :: finalize
    procedure, public :: source_is_internal
    procedure, public :: source_is_external
    procedure, public :: get_ngas
    procedure, public :: get_gases
    procedure, public :: get_press_min
    procedure, public :: get_press_max
    procedure, public :: get_temp_min
    procedure, public :: get_temp_max
    procedure, public :: gas_optics_int
    procedure, public :: gas_optics_ext
    procedure, public :: lw_fluxes          !< extension: gas_optics + rte_lw in one call (fused longwave path)
    procedure, public :: sw_fluxes          !< extension: gas_optics + rte_sw in one call (fused shortwave path)
  end type ty_gas_optics_ecckd

  interface
    function c_model_load(filename, device, model) bind(C, name="ecckd_model_load") result(rc)
      import c_char, c_int, c_ptr
      character(kind=c_char), dimension(*), intent(in) :: filename
      integer(c_int), value :: device
      type(c_ptr), intent(out) :: model
      integer(c_int) :: rc
    end function c_model_load
    subroutine c_model_destroy(model) bind(C, name="ecckd_model_destroy")
      import c_ptr
      type(c_ptr), value :: model
    end subroutine c_model_destroy
    function c_last_error() bind(C, name="ecckd_last_error") result(msg)
      import c_ptr
      type(c_ptr) :: msg
    end function c_last_error
    pure function c_get_ngpt(model) bind(C, name="ecckd_model_get_ngpt") result(n)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int) :: n
    end function c_get_ngpt
    pure function c_get_nband(model) bind(C, name="ecckd_model_get_nband") result(n)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int) :: n
    end function c_get_nband
    pure function c_get_ngas(model) bind(C, name="ecckd_model_get_ngas") result(n)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int) :: n
    end function c_get_ngas
    pure function c_get_gas_name(model, idx, name) bind(C, name="ecckd_model_get_gas_name") result(rc)
      import c_ptr, c_int, c_char
      type(c_ptr), value :: model
      integer(c_int), value :: idx
      character(kind=c_char), dimension(*), intent(inout) :: name
      integer(c_int) :: rc
    end function c_get_gas_name
    pure function c_is_internal(model) bind(C, name="ecckd_model_source_is_internal") result(n)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int) :: n
    end function c_is_internal
    pure function c_is_external(model) bind(C, name="ecckd_model_source_is_external") result(n)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int) :: n
    end function c_is_external
    pure function c_press_min(model) bind(C, name="ecckd_model_get_press_min") result(v)
      import c_ptr, c_double
      type(c_ptr), value :: model
      real(c_double) :: v
    end function c_press_min
    pure function c_press_max(model) bind(C, name="ecckd_model_get_press_max") result(v)
      import c_ptr, c_double
      type(c_ptr), value :: model
      real(c_double) :: v
    end function c_press_max
    pure function c_temp_min(model) bind(C, name="ecckd_model_get_temp_min") result(v)
      import c_ptr, c_double
      type(c_ptr), value :: model
      real(c_double) :: v
    end function c_temp_min
    pure function c_temp_max(model) bind(C, name="ecckd_model_get_temp_max") result(v)
      import c_ptr, c_double
      type(c_ptr), value :: model
      real(c_double) :: v
    end function c_temp_max
    function c_get_band2gpt(model, band2gpt) bind(C, name="ecckd_model_get_band2gpt") result(rc)
      import c_ptr, c_int
      type(c_ptr), value :: model
      integer(c_int), dimension(*), intent(out) :: band2gpt
      integer(c_int) :: rc
    end function c_get_band2gpt
    function c_get_band_lims(model, lims) bind(C, name="ecckd_model_get_band_lims_wvn") result(rc)
      import c_ptr, c_int, c_double
      type(c_ptr), value :: model
      real(c_double), dimension(*), intent(out) :: lims
      integer(c_int) :: rc
    end function c_get_band_lims
    function c_gas_optics_lw(model, ncol, nlay, plev, tlay, tsfc, tlev, ngas, gas_names, vmr, cs, ls, &
                             scalar, tau, lay_source, lev_inc, lev_dec, sfc_source, memspace, stream) &
        bind(C, name="ecckd_gas_optics_lw") result(rc)
      import c_ptr, c_int, c_double, c_char, c_long_long
      type(c_ptr), value :: model
      integer(c_int), value :: ncol, nlay, ngas, memspace
      real(c_double), dimension(*), intent(in) :: plev, tlay, tsfc
      type(c_ptr), value :: tlev
      character(kind=c_char), dimension(*), intent(in) :: gas_names
      type(c_ptr), dimension(*), intent(in) :: vmr
      integer(c_long_long), dimension(*), intent(in) :: cs, ls
      real(c_double), dimension(*), intent(in) :: scalar
      type(c_ptr), value :: tau, lay_source, lev_inc, lev_dec, sfc_source   ! host (ECCKD_HOST) or device (ECCKD_MIXED)
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_gas_optics_lw
    function c_gas_optics_sw(model, ncol, nlay, plev, tlay, ngas, gas_names, vmr, cs, ls, scalar, tau, &
                             ssa, g, toa_src, memspace, stream) bind(C, name="ecckd_gas_optics_sw") result(rc)
      import c_ptr, c_int, c_double, c_char, c_long_long
      type(c_ptr), value :: model
      integer(c_int), value :: ncol, nlay, ngas, memspace
      real(c_double), dimension(*), intent(in) :: plev, tlay
      character(kind=c_char), dimension(*), intent(in) :: gas_names
      type(c_ptr), dimension(*), intent(in) :: vmr
      integer(c_long_long), dimension(*), intent(in) :: cs, ls
      real(c_double), dimension(*), intent(in) :: scalar
      type(c_ptr), value :: tau, ssa, g                                     ! host (ECCKD_HOST) or device (ECCKD_MIXED)
      real(c_double), dimension(*), intent(inout) :: toa_src
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_gas_optics_sw
    function c_lw_fluxes(model, ncol, nlay, plev, tlay, tsfc, tlev, ngas, gas_names, vmr, cs, ls, scalar, top_at_1, nmus, &
                         sfc_emis, inc_flux, flux_up, flux_dn, memspace, stream) bind(C, name="ecckd_lw_fluxes") result(rc)
      import c_ptr, c_int, c_double, c_char, c_long_long
      type(c_ptr), value :: model
      integer(c_int), value :: ncol, nlay, ngas, top_at_1, nmus, memspace
      real(c_double), dimension(*), intent(in) :: plev, tlay, tsfc, tlev, sfc_emis
      character(kind=c_char), dimension(*), intent(in) :: gas_names
      type(c_ptr), dimension(*), intent(in) :: vmr
      integer(c_long_long), dimension(*), intent(in) :: cs, ls
      real(c_double), dimension(*), intent(in) :: scalar
      type(c_ptr), value :: inc_flux
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_lw_fluxes
    function c_sw_fluxes(model, ncol, nlay, plev, tlay, ngas, gas_names, vmr, cs, ls, scalar, top_at_1, mu0, toa_scale, &
                         sfc_alb_dir, sfc_alb_dif, flux_up, flux_dn, flux_dir, memspace, stream) &
        bind(C, name="ecckd_sw_fluxes") result(rc)
      import c_ptr, c_int, c_double, c_char, c_long_long
      type(c_ptr), value :: model
      integer(c_int), value :: ncol, nlay, ngas, top_at_1, memspace
      real(c_double), dimension(*), intent(in) :: plev, tlay, mu0, sfc_alb_dir, sfc_alb_dif
      character(kind=c_char), dimension(*), intent(in) :: gas_names
      type(c_ptr), dimension(*), intent(in) :: vmr
      integer(c_long_long), dimension(*), intent(in) :: cs, ls
      real(c_double), dimension(*), intent(in) :: scalar
      type(c_ptr), value :: toa_scale, flux_dir
      real(c_double), dimension(*), intent(inout) :: flux_up, flux_dn
      type(c_ptr), value :: stream
      integer(c_int) :: rc
    end function c_sw_fluxes
  end interface

  public :: c_error_message, c_loc_3d, c_loc_2d

contains

  !> Message of the last failing C call, as the reference's character(len=128) error strings.
  function c_error_message() result(msg)
    character(len=128) :: msg
    character(kind=c_char), dimension(:), pointer :: p
    type(c_ptr) :: cp
    integer :: i
    msg = ""
    cp = c_last_error()
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [128])
    do i = 1, 128
      if (p(i) == c_null_char) exit
      msg(i:i) = p(i)
    end do
    if (len_trim(msg) == 0) msg = "ecckd: unknown error"
  end function c_error_message

  !> load_and_init(ecckd, filename, available_gases) of mo_load_coefficients.F90:19-146 as a
  !! type-bound procedure; available_gases is accepted and ignored exactly as there (:19,:23).
  function load(this, filename, available_gases, device) result(error_msg)
    class(ty_gas_optics_ecckd), intent(inout) :: this
    character(len=*), intent(in) :: filename
    class(ty_gas_concs), intent(in), optional :: available_gases
    integer, intent(in), optional :: device
    character(len=128) :: error_msg
    integer(c_int) :: dev, rc, nband
    integer(c_int), dimension(:,:), allocatable :: b2g
    real(c_double), dimension(:,:), allocatable :: lims
    error_msg = ""
    dev = 0
    if (present(device)) dev = int(device, c_int)
    call this%finalize()
    rc = c_model_load(trim(filename) // c_null_char, dev, this%handle)
    if (rc /= 0) then
      error_msg = c_error_message()
      this%handle = c_null_ptr
      return
    end if
    ! ecckd%init(band_lims_wvn, band2gpt) of mo_load_coefficients.F90:74
    nband = c_get_nband(this%handle)
    allocate(b2g(2, nband), lims(2, nband))
    rc = c_get_band2gpt(this%handle, b2g)
    rc = c_get_band_lims(this%handle, lims)
    error_msg = this%init(lims, int(b2g))
  end function load

  subroutine finalize(this)
    class(ty_gas_optics_ecckd), intent(inout) :: this
    if (c_associated(this%handle)) call c_model_destroy(this%handle)
    this%handle = c_null_ptr
  end subroutine finalize

  pure function get_ngas(this)                                   ! src/gas_optics_ecckd.f90:477
    class(ty_gas_optics_ecckd), intent(in) :: this
    integer :: get_ngas
    get_ngas = int(c_get_ngas(this%handle))
  end function get_ngas

  pure function source_is_internal(this)                         ! :487
    class(ty_gas_optics_ecckd), intent(in) :: this
    logical :: source_is_internal
    source_is_internal = c_is_internal(this%handle) /= 0
  end function source_is_internal

  pure function source_is_external(this)                         ! :497
    class(ty_gas_optics_ecckd), intent(in) :: this
    logical :: source_is_external
    source_is_external = c_is_external(this%handle) /= 0
  end function source_is_external

  pure function get_gases(this)                                  ! :507
    class(ty_gas_optics_ecckd), intent(in) :: this
    character(len=32), dimension(this%get_ngas()) :: get_gases
    character(kind=c_char), dimension(name_len) :: buf
    integer :: i, k
    integer(c_int) :: rc
    do i = 1, size(get_gases)
      buf = c_null_char
      rc = c_get_gas_name(this%handle, int(i - 1, c_int), buf)
      get_gases(i) = ""
      do k = 1, name_len
        if (buf(k) == c_null_char) exit
        get_gases(i)(k:k) = buf(k)
      end do
    end do
  end function get_gases

  pure function get_press_min(this)                              ! :517
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp) :: get_press_min
    get_press_min = c_press_min(this%handle)
  end function get_press_min

  pure function get_press_max(this)                              ! :527
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp) :: get_press_max
    get_press_max = c_press_max(this%handle)
  end function get_press_max

  pure function get_temp_min(this)                               ! :537
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp) :: get_temp_min
    get_temp_min = c_temp_min(this%handle)
  end function get_temp_min

  pure function get_temp_max(this)                               ! :547
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp) :: get_temp_max
    get_temp_max = c_temp_max(this%handle)
  end function get_temp_max

  !> gas_desc -> the flat description the C ABI takes, without copying a concentration field: the C ABI takes a
  !! pointer and two strides per gas, which is exactly what ty_gas_concs stores -- concs(i)%conc is (1,1) for a
  !! scalar, (1,nlay) for a profile and (ncol,nlay) for a full field (RTE-RRTMGP's mo_gas_concentrations, same
  !! public components as the minimal type of mo_rte_min.F90), so the broadcast of get_vmr (:351) happens on the
  !! GPU through the strides.  Like the reference (:348-364) a concentration is only looked up for gases the
  !! k-distribution holds a table for: the others cross the boundary as a name with a null pointer (nothing is
  !! staged for them and an unset one is not an error); the get_vmr error texts are kept (:351-354).
  function marshal_gases(this, gas_desc, ncol, nlay, names, ptr, cs, ls, scalar) result(error_msg)
    class(ty_gas_optics_ecckd), intent(in) :: this
    type(ty_gas_concs), intent(in) :: gas_desc
    integer, intent(in) :: ncol, nlay
    character(kind=c_char), dimension(:), allocatable, intent(out) :: names
    type(c_ptr), dimension(:), allocatable, intent(out) :: ptr
    integer(c_long_long), dimension(:), allocatable, intent(out) :: cs, ls
    real(c_double), dimension(:), allocatable, intent(out) :: scalar
    character(len=128) :: error_msg
    character(len=32), dimension(:), allocatable :: gas_names, model_gases
    integer :: n, j, k, nm
    logical :: known
    error_msg = ""
    n = gas_desc%get_num_gases()
    nm = this%get_ngas()
    allocate(gas_names(n), model_gases(nm))
    gas_names = gas_desc%get_gas_names()
    model_gases = this%get_gases()
    allocate(names(max(1, n * name_len)), ptr(max(1, n)), cs(max(1, n)), ls(max(1, n)), scalar(max(1, n)))
    names = " "
    ptr = c_null_ptr
    cs = 0_c_long_long
    ls = 0_c_long_long
    scalar = 0._c_double
    do j = 1, n
      do k = 1, min(name_len, len_trim(gas_names(j)))
        names((j - 1) * name_len + k) = gas_names(j)(k:k)
      end do
      known = .false.
      do k = 1, nm
        if (trim(model_gases(k)) == trim(gas_names(j))) known = .true.
      end do
      if (.not. known) cycle                                  ! :358-364 silently skipped
      if (.not. allocated(gas_desc%concs(j)%conc)) then      ! (get_gas_names() returns gas_name(:): index j)
        error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas_names(j)) // " not found"
        return
      end if
      associate (c => gas_desc%concs(j)%conc)
        if (size(c, 1) > 1 .and. size(c, 1) /= ncol) then
          error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas_names(j)) // " array is inconsistent with ncol"
          return
        end if
        if (size(c, 2) > 1 .and. size(c, 2) /= nlay) then
          error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas_names(j)) // " array is inconsistent with nlay"
          return
        end if
        if (size(c) == 1) then
          scalar(j) = c(1, 1)
        else
          ptr(j) = c_loc_2d(c)
          if (size(c, 1) > 1) cs(j) = 1_c_long_long
          if (size(c, 2) > 1) ls(j) = int(size(c, 1), c_long_long)
        end if
      end associate
    end do
  end function marshal_gases

  !> gas_optics_int, src/gas_optics_ecckd.f90:381-426 (same argument list).  Host containers: ECCKD_HOST
  !! (every array is staged through the GPU).  Device twins (mo_ecckd_device): ECCKD_MIXED -- tau and the
  !! sources stay in HBM for rte_lw.
  function gas_optics_int(this, play, plev, tlay, tsfc, gas_desc, optical_props, sources, col_dry, tlev) &
      result(error_msg)
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp), dimension(:,:), intent(in) :: play, plev, tlay
    real(wp), dimension(:), intent(in) :: tsfc
    type(ty_gas_concs), intent(in) :: gas_desc
    class(ty_optical_props_arry), intent(inout) :: optical_props
    class(ty_source_func_lw), intent(inout) :: sources
    character(len=128) :: error_msg
    real(wp), dimension(:,:), intent(in), target, optional :: col_dry, tlev
    character(kind=c_char), dimension(:), allocatable :: names
    real(wp), dimension(:,:), allocatable, target :: tlev_c
    type(c_ptr), dimension(:), allocatable :: ptr
    integer(c_long_long), dimension(:), allocatable :: cs, ls
    real(c_double), dimension(:), allocatable :: scalar
    type(c_ptr) :: tlev_p, p_tau, p_lay, p_inc, p_dec, p_sfc
    integer :: ncol, nlay, n
    integer(c_int) :: rc, memspace
    ncol = size(tlay, 1)
    nlay = size(tlay, 2)
    error_msg = marshal_gases(this, gas_desc, ncol, nlay, names, ptr, cs, ls, scalar)
    if (trim(error_msg) /= "") return
    n = gas_desc%get_num_gases()
    tlev_p = c_null_ptr
    if (present(tlev)) then
      if (is_contiguous(tlev)) then
        tlev_p = c_loc_2d(tlev)
      else
        allocate(tlev_c(ncol, nlay + 1))
        tlev_c = tlev
        tlev_p = c_loc(tlev_c(1, 1))
      end if
    end if
    memspace = ECCKD_HOST
    select type (optical_props)
      class is (ty_optical_props_1scl_dev)
        select type (sources)
          class is (ty_source_func_lw_dev)
            if (optical_props%ncol /= ncol .or. optical_props%nlay /= nlay .or. sources%ncol /= ncol .or. &
                sources%nlay /= nlay) then
              error_msg = "gas_optics: device-resident optical_props / sources are inconsistently sized"
              return
            end if
            memspace = ECCKD_MIXED
            p_tau = optical_props%d_tau
            p_lay = sources%d_lay_source
            p_inc = sources%d_lev_source_inc
            p_dec = sources%d_lev_source_dec
            p_sfc = sources%d_sfc_source
          class default
            error_msg = "gas_optics: device-resident optical_props needs device-resident sources (ty_source_func_lw_dev)"
            return
        end select
      class default
        select type (sources)
          class is (ty_source_func_lw_dev)
            error_msg = "gas_optics: device-resident sources need device-resident optical_props (ty_optical_props_1scl_dev)"
            return
        end select
        p_tau = c_loc_3d(optical_props%tau)
        p_lay = c_loc_3d(sources%lay_source)
        p_inc = c_loc_3d(sources%lev_source_inc)
        p_dec = c_loc_3d(sources%lev_source_dec)
        p_sfc = c_loc_2d(sources%sfc_source)
    end select
    rc = c_gas_optics_lw(this%handle, int(ncol, c_int), int(nlay, c_int), plev, tlay, tsfc, tlev_p, &
                         int(n, c_int), names, ptr, cs, ls, scalar, p_tau, p_lay, p_inc, p_dec, p_sfc, memspace, &
                         c_null_ptr)
    if (rc /= 0) error_msg = c_error_message()
  end function gas_optics_int

  !> gas_optics_ext, src/gas_optics_ecckd.f90:431-473 (same argument list).
  function gas_optics_ext(this, play, plev, tlay, gas_desc, optical_props, toa_src, col_dry) result(error_msg)
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp), dimension(:,:), intent(in) :: play, plev, tlay
    type(ty_gas_concs), intent(in) :: gas_desc
    class(ty_optical_props_arry), intent(inout) :: optical_props
    real(wp), dimension(:,:), intent(out) :: toa_src
    real(wp), dimension(:,:), intent(in), target, optional :: col_dry
    character(len=128) :: error_msg
    character(kind=c_char), dimension(:), allocatable :: names
    type(c_ptr), dimension(:), allocatable :: ptr
    integer(c_long_long), dimension(:), allocatable :: cs, ls
    real(c_double), dimension(:), allocatable :: scalar
    type(c_ptr) :: tau_p, ssa_p, g_p
    integer :: ncol, nlay, n
    integer(c_int) :: rc, memspace
    ncol = size(tlay, 1)
    nlay = size(tlay, 2)
    error_msg = marshal_gases(this, gas_desc, ncol, nlay, names, ptr, cs, ls, scalar)
    if (trim(error_msg) /= "") return
    n = gas_desc%get_num_gases()
    ssa_p = c_null_ptr
    g_p = c_null_ptr
    memspace = ECCKD_HOST
    select type (optical_props)                  ! :457-464
      class is (ty_optical_props_2str_dev)
        if (optical_props%ncol /= ncol .or. optical_props%nlay /= nlay) then
          error_msg = "gas_optics: device-resident optical_props is inconsistently sized"
          return
        end if
        memspace = ECCKD_MIXED
        tau_p = optical_props%d_tau
        ssa_p = optical_props%d_ssa
        g_p = optical_props%d_g
      class is (ty_optical_props_1scl_dev)
        memspace = ECCKD_MIXED
        tau_p = optical_props%d_tau
      class is (ty_optical_props_2str)
        tau_p = c_loc_3d(optical_props%tau)
        ssa_p = c_loc_3d(optical_props%ssa)
        g_p = c_loc_3d(optical_props%g)
      class default
        tau_p = c_loc_3d(optical_props%tau)
    end select
    rc = c_gas_optics_sw(this%handle, int(ncol, c_int), int(nlay, c_int), plev, tlay, int(n, c_int), names, &
                         ptr, cs, ls, scalar, tau_p, ssa_p, g_p, toa_src, memspace, c_null_ptr)
    if (rc /= 0) error_msg = c_error_message()
  end function gas_optics_ext

  !> Extension (no counterpart in the reference): broadband longwave fluxes in one call -- what the reference's block
  !! loop computes with ecckd%gas_optics(...) followed by rte_lw(...) (ecckd_rfmip_lw.F90:120-135) -- through the fused
  !! path of the library (ecckd_lw_fluxes: tau stays on the GPU, the Planck sources are recomputed inside the solver).
  !! Host arrays in, host fluxes out (60 layers take the fused kernels, other counts the general route).  flux_up / flux_dn are (ncol, nlay+1), sfc_emis (nband, ncol).
  function lw_fluxes(this, plev, tlay, tsfc, tlev, gas_desc, top_at_1, sfc_emis, flux_up, flux_dn, n_gauss_angles) &
      result(error_msg)
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp), dimension(:,:), intent(in) :: plev, tlay, tlev
    real(wp), dimension(:), intent(in) :: tsfc
    type(ty_gas_concs), intent(in) :: gas_desc
    logical, intent(in) :: top_at_1
    real(wp), dimension(:,:), intent(in) :: sfc_emis
    real(wp), dimension(:,:), intent(inout) :: flux_up, flux_dn
    integer, intent(in), optional :: n_gauss_angles
    character(len=128) :: error_msg
    character(kind=c_char), dimension(:), allocatable :: names
    type(c_ptr), dimension(:), allocatable :: ptr
    integer(c_long_long), dimension(:), allocatable :: cs, ls
    real(c_double), dimension(:), allocatable :: scalar
    real(wp), dimension(:,:), allocatable :: up, dn
    integer :: ncol, nlay, n, nmus
    integer(c_int) :: rc
    ncol = size(tlay, 1)
    nlay = size(tlay, 2)
    nmus = 1
    if (present(n_gauss_angles)) nmus = n_gauss_angles
    error_msg = marshal_gases(this, gas_desc, ncol, nlay, names, ptr, cs, ls, scalar)
    if (trim(error_msg) /= "") return
    if (size(sfc_emis, 1) /= this%get_nband() .or. size(sfc_emis, 2) /= ncol) then
      error_msg = "lw_fluxes: sfc_emis inconsistently sized"
      return
    end if
    n = gas_desc%get_num_gases()
    allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1))
    rc = c_lw_fluxes(this%handle, int(ncol, c_int), int(nlay, c_int), plev, tlay, tsfc, tlev, int(n, c_int), names, ptr, cs, &
                     ls, scalar, merge(1_c_int, 0_c_int, top_at_1), int(nmus, c_int), sfc_emis, c_null_ptr, up, dn, &
                     ECCKD_HOST, c_null_ptr)
    if (rc /= 0) then
      error_msg = c_error_message()
      return
    end if
    flux_up = up
    flux_dn = dn
  end function lw_fluxes

  !> Extension (no counterpart in the reference): broadband shortwave fluxes in one call -- what the reference's block
  !! loop computes with ecckd%gas_optics(...), the rescaling of toa_flux and rte_sw(...) (ecckd_rfmip_sw.F90:118-154) --
  !! through the fused path of the library (ecckd_sw_fluxes: only the total optical depth goes through GPU memory; the
  !! solver derives ssa, g = 0 and the incoming beam as gas_optics_ext does, src/gas_optics_ecckd.f90:455-472).  Host
  !! arrays in, host fluxes out; at most 60 layers.  flux_* are (ncol, nlay+1), the albedos (nband, ncol); toa_scale(ncol)
  !! multiplies the incoming beam of a column (the drivers' total-solar-irradiance rescaling); flux_dir is optional.
  function sw_fluxes(this, plev, tlay, gas_desc, top_at_1, mu0, sfc_alb_dir, sfc_alb_dif, flux_up, flux_dn, flux_dir, toa_scale) &
      result(error_msg)
    class(ty_gas_optics_ecckd), intent(in) :: this
    real(wp), dimension(:,:), intent(in) :: plev, tlay
    type(ty_gas_concs), intent(in) :: gas_desc
    logical, intent(in) :: top_at_1
    real(wp), dimension(:), intent(in) :: mu0
    real(wp), dimension(:,:), intent(in) :: sfc_alb_dir, sfc_alb_dif
    real(wp), dimension(:,:), intent(inout) :: flux_up, flux_dn
    real(wp), dimension(:,:), intent(inout), optional :: flux_dir
    real(wp), dimension(:), intent(in), optional, target :: toa_scale
    character(len=128) :: error_msg
    character(kind=c_char), dimension(:), allocatable :: names
    type(c_ptr), dimension(:), allocatable :: ptr
    integer(c_long_long), dimension(:), allocatable :: cs, ls
    real(c_double), dimension(:), allocatable :: scalar
    real(wp), dimension(:,:), allocatable, target :: up, dn, dir
    real(wp), dimension(:), allocatable, target :: scale
    type(c_ptr) :: scale_p, dir_p
    integer :: ncol, nlay, n
    integer(c_int) :: rc
    ncol = size(tlay, 1)
    nlay = size(tlay, 2)
    error_msg = marshal_gases(this, gas_desc, ncol, nlay, names, ptr, cs, ls, scalar)
    if (trim(error_msg) /= "") return
    if (size(sfc_alb_dir, 1) /= this%get_nband() .or. size(sfc_alb_dir, 2) /= ncol .or. &
        size(sfc_alb_dif, 1) /= this%get_nband() .or. size(sfc_alb_dif, 2) /= ncol) then
      error_msg = "sw_fluxes: surface albedos inconsistently sized"
      return
    end if
    n = gas_desc%get_num_gases()
    allocate(up(ncol, nlay + 1), dn(ncol, nlay + 1))
    scale_p = c_null_ptr
    dir_p = c_null_ptr
    if (present(toa_scale)) then
      allocate(scale(ncol))
      scale = toa_scale
      scale_p = c_loc(scale(1))
    end if
    if (present(flux_dir)) then
      allocate(dir(ncol, nlay + 1))
      dir_p = c_loc(dir(1, 1))
    end if
    rc = c_sw_fluxes(this%handle, int(ncol, c_int), int(nlay, c_int), plev, tlay, int(n, c_int), names, ptr, cs, ls, scalar, &
                     merge(1_c_int, 0_c_int, top_at_1), mu0, scale_p, sfc_alb_dir, sfc_alb_dif, up, dn, dir_p, ECCKD_HOST, &
                     c_null_ptr)
    if (rc /= 0) then
      error_msg = c_error_message()
      return
    end if
    flux_up = up
    flux_dn = dn
    if (present(flux_dir)) flux_dir = dir
  end function sw_fluxes

  function c_loc_3d(a) result(p)
    real(wp), dimension(:,:,:), intent(in), target, contiguous :: a
    type(c_ptr) :: p
    p = c_loc(a(1, 1, 1))
  end function c_loc_3d

  function c_loc_2d(a) result(p)
    real(wp), dimension(:,:), intent(in), target, contiguous :: a
    type(c_ptr) :: p
    p = c_loc(a(1, 1))
  end function c_loc_2d

end module gas_optics_ecckd
