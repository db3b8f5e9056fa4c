! mo_ecckd_device.F90 -- device-resident twins of the RTE-RRTMGP containers that sit BETWEEN gas_optics and
! the solvers: ty_optical_props_1scl / _2str and ty_source_func_lw whose (ncol,nlay,ngpt) arrays live in the
! MI355X's HBM instead of in host allocatables.
!
! Why: the reference's block loop (example/rfmip-rad-irf/ecckd_rfmip_lw.F90:107-136) calls
! ecckd%gas_optics(..., optical_props, source, ...) and then rte_lw(optical_props, ..., source, ...) with
! nothing in between that reads tau or the sources.  With host allocatables every call ships 64 B per
! (column, layer, g-point) over PCIe twice; with these types only the atmosphere (about 2.4 KB per column) goes
! in and the fluxes (1 KB per column) come out -- the C ABI's ECCKD_MIXED memory space.
!
! Use: declare the container through its parent class and pick the dynamic type at run time,
!       class(ty_optical_props_1scl), allocatable :: optical_props
!       allocate(ty_optical_props_1scl_dev :: optical_props)      ! or ty_optical_props_1scl for host arrays
!       call stop_on_err(optical_props%alloc_1scl(ncol, nlay, ecckd))
! the calls ecckd%gas_optics(...) and rte_lw(...) / rte_sw(...) stay exactly as the reference spells them:
! the shims look at the dynamic type (select type) and pass device pointers.  The host members tau / ssa / g /
! lay_source ... of a device twin are NOT allocated; copy_to_host() fills them for inspection.
! No numerics here: allocation and marshalling only.
module mo_ecckd_device
  use, intrinsic :: iso_c_binding
  use mo_rte_kind, only: wp
  use mo_optical_props, only: ty_optical_props, ty_optical_props_1scl, ty_optical_props_2str
  use mo_source_functions, only: ty_source_func_lw
  implicit none
  private

  integer(c_int), parameter, public :: ECCKD_HOST = 0_c_int, ECCKD_DEVICE = 1_c_int, ECCKD_MIXED = 2_c_int

  type, extends(ty_optical_props_1scl), public :: ty_optical_props_1scl_dev
    type(c_ptr) :: d_tau = c_null_ptr          !< tau(ncol,nlay,ngpt) in device memory
    integer :: ncol = 0, nlay = 0, device = 0
  contains
    procedure, public :: alloc_1scl => alloc_1scl_dev
    procedure, public :: free => free_1scl_dev
    procedure, public :: copy_to_host => copy_1scl_to_host
  end type ty_optical_props_1scl_dev

  type, extends(ty_optical_props_2str), public :: ty_optical_props_2str_dev
    type(c_ptr) :: d_tau = c_null_ptr, d_ssa = c_null_ptr, d_g = c_null_ptr
    integer :: ncol = 0, nlay = 0, device = 0
  contains
    procedure, public :: alloc_2str => alloc_2str_dev
    procedure, public :: free => free_2str_dev
    procedure, public :: copy_to_host => copy_2str_to_host
  end type ty_optical_props_2str_dev

  type, extends(ty_source_func_lw), public :: ty_source_func_lw_dev
    type(c_ptr) :: d_lay_source = c_null_ptr, d_lev_source_inc = c_null_ptr, d_lev_source_dec = c_null_ptr
    type(c_ptr) :: d_sfc_source = c_null_ptr   !< (ncol,ngpt)
    integer :: ncol = 0, nlay = 0, device = 0
  contains
    procedure, public :: alloc => alloc_source_dev
    procedure, public :: free => free_source_dev
    procedure, public :: copy_to_host => copy_source_to_host
  end type ty_source_func_lw_dev

  interface
    function c_device_malloc(device, bytes, ptr) bind(C, name="ecckd_device_malloc") result(rc)
      import c_int, c_size_t, c_ptr
      integer(c_int), value :: device
      integer(c_size_t), value :: bytes
      type(c_ptr), intent(out) :: ptr
      integer(c_int) :: rc
    end function c_device_malloc
    function c_device_free(device, ptr) bind(C, name="ecckd_device_free") result(rc)
      import c_int, c_ptr
      integer(c_int), value :: device
      type(c_ptr), value :: ptr
      integer(c_int) :: rc
    end function c_device_free
    function c_device_memcpy(device, dst, src, bytes, to_device) bind(C, name="ecckd_device_memcpy") result(rc)
      import c_int, c_size_t, c_ptr
      integer(c_int), value :: device, to_device
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: rc
    end function c_device_memcpy
    function c_last_error_dev() bind(C, name="ecckd_last_error") result(msg)
      import c_ptr
      type(c_ptr) :: msg
    end function c_last_error_dev
  end interface

  public :: set_default_device

  integer, save :: default_device = 0   !< device ordinal new containers are allocated on

contains

  !> Device ordinal for containers allocated from now on (one process per GPU: set once at start-up).
  subroutine set_default_device(device)
    integer, intent(in) :: device
    default_device = device
  end subroutine set_default_device

  function dev_error() result(msg)
    character(len=128) :: msg
    character(kind=c_char), dimension(:), pointer :: p
    type(c_ptr) :: cp
    integer :: i
    msg = "ecckd: device allocation failed"
    cp = c_last_error_dev()
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [128])
    msg = ""
    do i = 1, 128
      if (p(i) == c_null_char) exit
      msg(i:i) = p(i)
    end do
  end function dev_error

  function dev_alloc(device, nelem, ptr) result(error_msg)
    integer, intent(in) :: device
    integer(c_size_t), intent(in) :: nelem
    type(c_ptr), intent(inout) :: ptr
    character(len=128) :: error_msg
    integer(c_int) :: rc
    error_msg = ""
    if (c_associated(ptr)) rc = c_device_free(int(device, c_int), ptr)
    ptr = c_null_ptr
    rc = c_device_malloc(int(device, c_int), nelem * c_sizeof(1._wp), ptr)
    if (rc /= 0) error_msg = dev_error()
  end function dev_alloc

  subroutine dev_free(device, ptr)
    integer, intent(in) :: device
    type(c_ptr), intent(inout) :: ptr
    integer(c_int) :: rc
    if (c_associated(ptr)) rc = c_device_free(int(device, c_int), ptr)
    ptr = c_null_ptr
  end subroutine dev_free

  function d2h_3d(device, src, dst) result(error_msg)
    integer, intent(in) :: device
    type(c_ptr), intent(in) :: src
    real(wp), dimension(:,:,:), intent(inout), target, contiguous :: dst
    character(len=128) :: error_msg
    error_msg = ""
    if (c_device_memcpy(int(device, c_int), c_loc(dst(1, 1, 1)), src, size(dst, kind=c_size_t) * c_sizeof(1._wp), &
                        0_c_int) /= 0) error_msg = dev_error()
  end function d2h_3d

  ! ---- ty_optical_props_1scl_dev ----
  function alloc_1scl_dev(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_optical_props_1scl_dev), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    this%ncol = ncol
    this%nlay = nlay
    this%device = default_device
    err_message = dev_alloc(this%device, int(ncol, c_size_t) * nlay * this%get_ngpt(), this%d_tau)
  end function alloc_1scl_dev

  subroutine free_1scl_dev(this)
    class(ty_optical_props_1scl_dev), intent(inout) :: this
    call dev_free(this%device, this%d_tau)
  end subroutine free_1scl_dev

  function copy_1scl_to_host(this) result(err_message)
    class(ty_optical_props_1scl_dev), intent(inout) :: this
    character(len=128) :: err_message
    if (allocated(this%tau)) deallocate(this%tau)
    allocate(this%tau(this%ncol, this%nlay, this%get_ngpt()))
    err_message = d2h_3d(this%device, this%d_tau, this%tau)
  end function copy_1scl_to_host

  ! ---- ty_optical_props_2str_dev ----
  function alloc_2str_dev(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_optical_props_2str_dev), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    integer(c_size_t) :: n
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    this%ncol = ncol
    this%nlay = nlay
    this%device = default_device
    n = int(ncol, c_size_t) * nlay * this%get_ngpt()
    err_message = dev_alloc(this%device, n, this%d_tau)
    if (err_message == "") err_message = dev_alloc(this%device, n, this%d_ssa)
    if (err_message == "") err_message = dev_alloc(this%device, n, this%d_g)
  end function alloc_2str_dev

  subroutine free_2str_dev(this)
    class(ty_optical_props_2str_dev), intent(inout) :: this
    call dev_free(this%device, this%d_tau)
    call dev_free(this%device, this%d_ssa)
    call dev_free(this%device, this%d_g)
  end subroutine free_2str_dev

  function copy_2str_to_host(this) result(err_message)
    class(ty_optical_props_2str_dev), intent(inout) :: this
    character(len=128) :: err_message
    integer :: ngpt
    ngpt = this%get_ngpt()
    if (allocated(this%tau)) deallocate(this%tau)
    if (allocated(this%ssa)) deallocate(this%ssa)
    if (allocated(this%g)) deallocate(this%g)
    allocate(this%tau(this%ncol, this%nlay, ngpt), this%ssa(this%ncol, this%nlay, ngpt), this%g(this%ncol, this%nlay, ngpt))
    err_message = d2h_3d(this%device, this%d_tau, this%tau)
    if (err_message == "") err_message = d2h_3d(this%device, this%d_ssa, this%ssa)
    if (err_message == "") err_message = d2h_3d(this%device, this%d_g, this%g)
  end function copy_2str_to_host

  ! ---- ty_source_func_lw_dev ----
  function alloc_source_dev(this, ncol, nlay, spectral_desc) result(err_message)
    class(ty_source_func_lw_dev), intent(inout) :: this
    integer, intent(in) :: ncol, nlay
    class(ty_optical_props), intent(in) :: spectral_desc
    character(len=128) :: err_message
    integer(c_size_t) :: n
    err_message = this%init(spectral_desc%band_lims_wvn, spectral_desc%band2gpt)
    if (err_message /= "") return
    this%ncol = ncol
    this%nlay = nlay
    this%device = default_device
    n = int(ncol, c_size_t) * nlay * this%get_ngpt()
    err_message = dev_alloc(this%device, n, this%d_lay_source)
    if (err_message == "") err_message = dev_alloc(this%device, n, this%d_lev_source_inc)
    if (err_message == "") err_message = dev_alloc(this%device, n, this%d_lev_source_dec)
    if (err_message == "") err_message = dev_alloc(this%device, int(ncol, c_size_t) * this%get_ngpt(), this%d_sfc_source)
  end function alloc_source_dev

  subroutine free_source_dev(this)
    class(ty_source_func_lw_dev), intent(inout) :: this
    call dev_free(this%device, this%d_lay_source)
    call dev_free(this%device, this%d_lev_source_inc)
    call dev_free(this%device, this%d_lev_source_dec)
    call dev_free(this%device, this%d_sfc_source)
  end subroutine free_source_dev

  function copy_source_to_host(this) result(err_message)
    class(ty_source_func_lw_dev), intent(inout) :: this
    character(len=128) :: err_message
    real(wp), dimension(:,:,:), allocatable :: tmp
    integer :: ngpt
    ngpt = this%get_ngpt()
    if (allocated(this%lay_source)) deallocate(this%lay_source, this%lev_source_inc, this%lev_source_dec, this%sfc_source)
    allocate(this%lay_source(this%ncol, this%nlay, ngpt), this%lev_source_inc(this%ncol, this%nlay, ngpt), &
             this%lev_source_dec(this%ncol, this%nlay, ngpt), this%sfc_source(this%ncol, ngpt), tmp(this%ncol, ngpt, 1))
    err_message = d2h_3d(this%device, this%d_lay_source, this%lay_source)
    if (err_message == "") err_message = d2h_3d(this%device, this%d_lev_source_inc, this%lev_source_inc)
    if (err_message == "") err_message = d2h_3d(this%device, this%d_lev_source_dec, this%lev_source_dec)
    if (err_message == "") err_message = d2h_3d(this%device, this%d_sfc_source, tmp)
    this%sfc_source = tmp(:, :, 1)
  end function copy_source_to_host

end module mo_ecckd_device
