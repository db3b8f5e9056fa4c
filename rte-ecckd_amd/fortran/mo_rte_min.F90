! mo_rte_min.F90 -- the RTE-RRTMGP modules the ecCKD plugin is written against, reduced to the
! members the reference touches (SURVEY.md section 8(b)).  RTE-RRTMGP itself is not part of this
! repository; a host model that already links it drops this file and keeps only
! gas_optics_ecckd.F90 / mo_rte_solvers.F90 (same module, type and procedure names, same
! argument lists).  Module, type and member names follow RTE-RRTMGP v1.5.
!
!   mo_rte_kind            wp
!   mo_gas_concentrations  ty_gas_concs: init, set_vmr (scalar / (nlay) / (ncol,nlay)), get_vmr,
!                          get_gas_names, get_num_gases   (src/gas_optics_ecckd.f90:340-351,
!                          example/rfmip-rad-irf/mo_rfmip_io.F90:202-259)
!   mo_optical_props       ty_optical_props (band structure: init, get_nband, get_ngpt,
!                          get_band_lims_gpoint), ty_optical_props_arry%tau, _1scl%alloc_1scl,
!                          _2str%ssa,%g,%alloc_2str    (src/gas_optics_ecckd.f90:346,370,456-460)
!   mo_source_functions    ty_source_func_lw: lay_source, lev_source_inc, lev_source_dec,
!                          sfc_source, alloc           (:407-424, ecckd_rfmip_lw.F90:102)
!   mo_fluxes              ty_fluxes_broadband: flux_up, flux_dn, flux_dn_dir pointers
!                          (ecckd_rfmip_lw.F90:108-109)
!   mo_fluxes_byband       ty_fluxes_byband: bnd_flux_up, bnd_flux_dn, bnd_flux_dn_dir (ncol,nlev,nband) on top of
!                          the broadband members (what RTE-RRTMGP callers pass to rte_lw / rte_sw for spectral output)
!   mo_gas_optics          abstract ty_gas_optics with the deferred interfaces of
!                          src/gas_optics_ecckd.f90:381-395, 431-442, 487-553
module mo_rte_kind
  use, intrinsic :: iso_c_binding, only: c_double
  implicit none
  integer, parameter :: wp = c_double
end module mo_rte_kind


module mo_gas_concentrations
  use mo_rte_kind, only: wp
  implicit none
  private
  type :: conc_field
    real(wp), dimension(:,:), allocatable :: conc   ! (1,1), (1,nlay) or (ncol,nlay)
  end type conc_field
  type, public :: ty_gas_concs
    character(len=32), dimension(:), allocatable :: gas_name
    type(conc_field), dimension(:), allocatable :: concs
  contains
    procedure, public :: init
    procedure, private :: set_vmr_scalar
    procedure, private :: set_vmr_1d
    procedure, private :: set_vmr_2d
    generic, public :: set_vmr => set_vmr_scalar, set_vmr_1d, set_vmr_2d
    procedure, public :: get_vmr
    procedure, public :: get_num_gases
    procedure, public :: get_gas_names
  end type ty_gas_concs
contains
  function init(this, gas_names) result(error_msg)
    class(ty_gas_concs), intent(inout) :: this
    character(len=*), dimension(:), intent(in) :: gas_names
    character(len=128) :: error_msg
    integer :: i, j
    error_msg = ""
    do i = 1, size(gas_names)
      if (len_trim(gas_names(i)) == 0) error_msg = "ty_gas_concs%init: must provide non-empty gas names"
      do j = 1, i - 1
        if (trim(gas_names(i)) == trim(gas_names(j))) &
          error_msg = "ty_gas_concs%init: duplicate gas names aren't allowed"
      end do
    end do
    if (error_msg /= "") return
    if (allocated(this%gas_name)) deallocate(this%gas_name)
    if (allocated(this%concs)) deallocate(this%concs)
    allocate(this%gas_name(size(gas_names)), this%concs(size(gas_names)))
    do i = 1, size(gas_names)
      this%gas_name(i) = trim(gas_names(i))
    end do
  end function init

  integer function find_gas(this, gas)
    class(ty_gas_concs), intent(in) :: this
    character(len=*), intent(in) :: gas
    integer :: i
    find_gas = 0
    if (.not. allocated(this%gas_name)) return
    do i = 1, size(this%gas_name)
      if (trim(this%gas_name(i)) == trim(gas)) then
        find_gas = i
        return
      end if
    end do
  end function find_gas

  function set_vmr_scalar(this, gas, w) result(error_msg)
    class(ty_gas_concs), intent(inout) :: this
    character(len=*), intent(in) :: gas
    real(wp), intent(in) :: w
    character(len=128) :: error_msg
    integer :: i
    error_msg = ""
    if (w < 0._wp .or. w > 1._wp) then
      error_msg = "ty_gas_concs%set_vmr: concentrations should be >= 0, <= 1"
      return
    end if
    i = find_gas(this, gas)
    if (i == 0) then
      error_msg = "ty_gas_concs%set_vmr: trying to set " // trim(gas) // " but name not provided at initialization"
      return
    end if
    if (allocated(this%concs(i)%conc)) deallocate(this%concs(i)%conc)
    allocate(this%concs(i)%conc(1, 1))
    this%concs(i)%conc(1, 1) = w
  end function set_vmr_scalar

  function set_vmr_1d(this, gas, w) result(error_msg)
    class(ty_gas_concs), intent(inout) :: this
    character(len=*), intent(in) :: gas
    real(wp), dimension(:), intent(in) :: w
    character(len=128) :: error_msg
    integer :: i
    error_msg = ""
    if (any(w < 0._wp) .or. any(w > 1._wp)) then
      error_msg = "ty_gas_concs%set_vmr: concentrations should be >= 0, <= 1"
      return
    end if
    i = find_gas(this, gas)
    if (i == 0) then
      error_msg = "ty_gas_concs%set_vmr: trying to set " // trim(gas) // " but name not provided at initialization"
      return
    end if
    if (allocated(this%concs(i)%conc)) deallocate(this%concs(i)%conc)
    allocate(this%concs(i)%conc(1, size(w)))
    this%concs(i)%conc(1, :) = w
  end function set_vmr_1d

  function set_vmr_2d(this, gas, w) result(error_msg)
    class(ty_gas_concs), intent(inout) :: this
    character(len=*), intent(in) :: gas
    real(wp), dimension(:,:), intent(in) :: w
    character(len=128) :: error_msg
    integer :: i
    error_msg = ""
    if (any(w < 0._wp) .or. any(w > 1._wp)) then
      error_msg = "ty_gas_concs%set_vmr: concentrations should be >= 0, <= 1"
      return
    end if
    i = find_gas(this, gas)
    if (i == 0) then
      error_msg = "ty_gas_concs%set_vmr: trying to set " // trim(gas) // " but name not provided at initialization"
      return
    end if
    if (allocated(this%concs(i)%conc)) deallocate(this%concs(i)%conc)
    allocate(this%concs(i)%conc(size(w, 1), size(w, 2)))
    this%concs(i)%conc = w
  end function set_vmr_2d

  ! Broadcast the stored field to array(ncol,nlay), as src/gas_optics_ecckd.f90:351 relies on.
  function get_vmr(this, gas, array) result(error_msg)
    class(ty_gas_concs), intent(in) :: this
    character(len=*), intent(in) :: gas
    real(wp), dimension(:,:), intent(out) :: array
    character(len=128) :: error_msg
    integer :: i, icol, ilay
    error_msg = ""
    i = find_gas(this, gas)
    if (i == 0) then
      error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas) // " not found"
      return
    end if
    if (.not. allocated(this%concs(i)%conc)) then
      error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas) // " not found"
      return
    end if
    associate (c => this%concs(i)%conc)
      if (size(c, 1) > 1 .and. size(c, 1) /= size(array, 1)) then
        error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas) // " array is inconsistent with ncol"
        return
      end if
      if (size(c, 2) > 1 .and. size(c, 2) /= size(array, 2)) then
        error_msg = "ty_gas_concs%get_vmr; gas " // trim(gas) // " array is inconsistent with nlay"
        return
      end if
      do ilay = 1, size(array, 2)
        do icol = 1, size(array, 1)
          array(icol, ilay) = c(min(icol, size(c, 1)), min(ilay, size(c, 2)))
        end do
      end do
    end associate
  end function get_vmr

  pure integer function get_num_gases(this)
    class(ty_gas_concs), intent(in) :: this
    get_num_gases = 0
    if (allocated(this%gas_name)) get_num_gases = size(this%gas_name)
  end function get_num_gases

  pure function get_gas_names(this)
    class(ty_gas_concs), intent(in) :: this
    character(len=32), dimension(this%get_num_gases()) :: get_gas_names
    if (allocated(this%gas_name)) get_gas_names(:) = this%gas_name(:)
  end function get_gas_names
end module mo_gas_concentrations


module mo_optical_props
  use mo_rte_kind, only: wp
  implicit none
  private
  type, public :: ty_optical_props
    integer, dimension(:,:), allocatable :: band2gpt        ! (2,nband)
    real(wp), dimension(:,:), allocatable :: band_lims_wvn  ! (2,nband)
  contains
    procedure, public :: init
    procedure, public :: get_nband
    procedure, public :: get_ngpt
    procedure, public :: get_band_lims_gpoint
    procedure, public :: get_band_lims_wavenumber
  end type ty_optical_props
  type, extends(ty_optical_props), abstract, public :: ty_optical_props_arry
    real(wp), dimension(:,:,:), allocatable :: tau          ! (ncol,nlay,ngpt)
  end type ty_optical_props_arry
  type, extends(ty_optical_props_arry), public :: ty_optical_props_1scl
  contains
    procedure, public :: alloc_1scl
  end type ty_optical_props_1scl
  type, extends(ty_optical_props_arry), public :: ty_optical_props_2str
    real(wp), dimension(:,:,:), allocatable :: ssa, g
  contains
    procedure, public :: alloc_2str
  end type ty_optical_props_2str
contains
  function init(this, band_lims_wvn, band_lims_gpt) result(err_message)
    class(ty_optical_props), intent(inout) :: this
    real(wp), dimension(:,:), intent(in) :: band_lims_wvn
    integer, dimension(:,:), intent(in) :: band_lims_gpt
    character(len=128) :: err_message
    err_message = ""
    if (size(band_lims_wvn, 1) /= 2 .or. size(band_lims_gpt, 1) /= 2 .or. &
        size(band_lims_wvn, 2) /= size(band_lims_gpt, 2)) then
      err_message = "optical_props%init(): band_lims_wvn and band_lims_gpt have inconsistent sizes"
      return
    end if
    if (allocated(this%band2gpt)) deallocate(this%band2gpt)
    if (allocated(this%band_lims_wvn)) deallocate(this%band_lims_wvn)
    allocate(this%band2gpt(2, size(band_lims_gpt, 2)), this%band_lims_wvn(2, size(band_lims_wvn, 2)))
    this%band2gpt = band_lims_gpt
    this%band_lims_wvn = band_lims_wvn
  end function init
  pure integer function get_nband(this)
    class(ty_optical_props), intent(in) :: this
    get_nband = 0
    if (allocated(this%band2gpt)) get_nband = size(this%band2gpt, 2)
  end function get_nband
  pure integer function get_ngpt(this)
    class(ty_optical_props), intent(in) :: this
    get_ngpt = 0
    if (allocated(this%band2gpt)) get_ngpt = maxval(this%band2gpt)
  end function get_ngpt
  pure function get_band_lims_gpoint(this)
    class(ty_optical_props), intent(in) :: this
    integer, dimension(2, this%get_nband()) :: get_band_lims_gpoint
    get_band_lims_gpoint = this%band2gpt
  end function get_band_lims_gpoint
  pure function get_band_lims_wavenumber(this)
    class(ty_optical_props), intent(in) :: this
    real(wp), dimension(2, this%get_nband()) :: get_band_lims_wavenumber
    get_band_lims_wavenumber = this%band_lims_wvn
  end function get_band_lims_wavenumber
  function alloc_1scl(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_optical_props_1scl), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    if (allocated(this%tau)) deallocate(this%tau)
    allocate(this%tau(ncol, nlay, this%get_ngpt()))
  end function alloc_1scl
  function alloc_2str(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_optical_props_2str), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    if (allocated(this%tau)) deallocate(this%tau)
    if (allocated(this%ssa)) deallocate(this%ssa)
    if (allocated(this%g)) deallocate(this%g)
    allocate(this%tau(ncol, nlay, this%get_ngpt()), this%ssa(ncol, nlay, this%get_ngpt()), &
             this%g(ncol, nlay, this%get_ngpt()))
  end function alloc_2str
end module mo_optical_props


module mo_source_functions
  use mo_rte_kind, only: wp
  use mo_optical_props, only: ty_optical_props
  implicit none
  private
  type, extends(ty_optical_props), public :: ty_source_func_lw
    real(wp), dimension(:,:,:), allocatable :: lay_source, lev_source_inc, lev_source_dec
    real(wp), dimension(:,:), allocatable :: sfc_source
  contains
    procedure, public :: alloc
  end type ty_source_func_lw
contains
  function alloc(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_source_func_lw), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    integer :: ngpt
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    ngpt = this%get_ngpt()
    if (allocated(this%lay_source)) deallocate(this%lay_source, this%lev_source_inc, this%lev_source_dec, &
                                               this%sfc_source)
    allocate(this%lay_source(ncol, nlay, ngpt), this%lev_source_inc(ncol, nlay, ngpt), &
             this%lev_source_dec(ncol, nlay, ngpt), this%sfc_source(ncol, ngpt))
  end function alloc
end module mo_source_functions


module mo_fluxes
  use mo_rte_kind, only: wp
  implicit none
  private
  type, public :: ty_fluxes_broadband
    real(wp), dimension(:,:), pointer :: flux_up => null(), flux_dn => null()   ! (ncol,nlay+1)
    real(wp), dimension(:,:), pointer :: flux_net => null(), flux_dn_dir => null()
  end type ty_fluxes_broadband
end module mo_fluxes


module mo_fluxes_byband
  use mo_rte_kind, only: wp
  use mo_fluxes, only: ty_fluxes_broadband
  implicit none
  private
  type, extends(ty_fluxes_broadband), public :: ty_fluxes_byband
    real(wp), dimension(:,:,:), pointer :: bnd_flux_up => null(), bnd_flux_dn => null()   ! (ncol,nlay+1,nband)
    real(wp), dimension(:,:,:), pointer :: bnd_flux_net => null(), bnd_flux_dn_dir => null()
  end type ty_fluxes_byband
end module mo_fluxes_byband


module mo_gas_optics
  use mo_rte_kind, only: wp
  use mo_gas_concentrations, only: ty_gas_concs
  use mo_optical_props, only: ty_optical_props, ty_optical_props_arry
  use mo_source_functions, only: ty_source_func_lw
  implicit none
  private
  type, abstract, extends(ty_optical_props), public :: ty_gas_optics
  contains
    generic, public :: gas_optics => gas_optics_int, gas_optics_ext
    procedure(gas_optics_int_abstract), deferred, public :: gas_optics_int
    procedure(gas_optics_ext_abstract), deferred, public :: gas_optics_ext
    procedure(logical_abstract), deferred, public :: source_is_internal
    procedure(logical_abstract), deferred, public :: source_is_external
    procedure(real_abstract), deferred, public :: get_press_min
    procedure(real_abstract), deferred, public :: get_press_max
    procedure(real_abstract), deferred, public :: get_temp_min
    procedure(real_abstract), deferred, public :: get_temp_max
  end type ty_gas_optics
  abstract interface
    function gas_optics_int_abstract(this, play, plev, tlay, tsfc, gas_desc, optical_props, sources, &
                                     col_dry, tlev) result(error_msg)
      import ty_gas_optics, wp, ty_gas_concs, ty_optical_props_arry, ty_source_func_lw
      class(ty_gas_optics), intent(in) :: this
      real(wp), dimension(:,:), intent(in) :: play, plev, tlay
      real(wp), dimension(:), intent(in) :: tsfc
      type(ty_gas_concs), intent(in) :: gas_desc
      class(ty_optical_props_arry), intent(inout) :: optical_props
      class(ty_source_func_lw), intent(inout) :: sources
      character(len=128) :: error_msg
      real(wp), dimension(:,:), intent(in), target, optional :: col_dry, tlev
    end function gas_optics_int_abstract
    function gas_optics_ext_abstract(this, play, plev, tlay, gas_desc, optical_props, toa_src, col_dry) &
        result(error_msg)
      import ty_gas_optics, wp, ty_gas_concs, ty_optical_props_arry
      class(ty_gas_optics), intent(in) :: this
      real(wp), dimension(:,:), intent(in) :: play, plev, tlay
      type(ty_gas_concs), intent(in) :: gas_desc
      class(ty_optical_props_arry), intent(inout) :: optical_props
      real(wp), dimension(:,:), intent(out) :: toa_src
      character(len=128) :: error_msg
      real(wp), dimension(:,:), intent(in), target, optional :: col_dry
    end function gas_optics_ext_abstract
    pure function logical_abstract(this)
      import ty_gas_optics
      class(ty_gas_optics), intent(in) :: this
      logical :: logical_abstract
    end function logical_abstract
    pure function real_abstract(this)
      import ty_gas_optics, wp
      class(ty_gas_optics), intent(in) :: this
      real(wp) :: real_abstract
    end function real_abstract
  end interface
end module mo_gas_optics
