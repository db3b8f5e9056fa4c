! rfmip_support.F90 -- the helper modules of example/rfmip-rad-irf/ that the RFMIP drivers use, written
! against the library's own netCDF-3 access (include/ecckd_nc.h) instead of netcdf-fortran:
!
!   simple_netcdf       stop_on_err (mo_simple_netcdf.F90:331-339) + the few typed accessors needed here
!   load_coefficients   load_and_init(ecckd, filename, available_gases) -> ecckd%load   (mo_load_coefficients.F90:19)
!   rfmip_io            read_size, read_and_block_pt, read_and_block_lw_bc, read_and_block_sw_bc,
!                       read_and_block_gases_ty, unblock_and_write with the file schema and the blocking
!                       conventions of mo_rfmip_io.F90:28-317
!   utils               determine_gas_names, parse_args of utils.f90:41-134 (same flags: -f, -p, -h/--help),
!                       plus -b <block size> and -n <max blocks>
!
! Blocking convention (mo_rfmip_io.F90:78-99): the (site, expt) pairs are flattened site-fastest into
! ncol*nexp columns and cut into nblocks blocks of `blocksize`; arrays come back as
! (blocksize, nlay[+1], nblocks).
module simple_netcdf
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: error_unit
  use mo_rte_kind, only: wp
  use gas_optics_ecckd, only: c_error_message
  implicit none
  private
  public :: stop_on_err, nc_file, nc_open, nc_close, get_dim_size, var_exists, read_all, read_units_scaling, &
            write_field

  type :: nc_file
    type(c_ptr) :: h = c_null_ptr
  end type nc_file

  interface
    function c_nc_open(path, f) bind(C, name="ecckd_nc_open") result(rc)
      import c_char, c_ptr, c_int
      character(kind=c_char), dimension(*), intent(in) :: path
      type(c_ptr), intent(out) :: f
      integer(c_int) :: rc
    end function
    subroutine c_nc_close(f) bind(C, name="ecckd_nc_close")
      import c_ptr
      type(c_ptr), value :: f
    end subroutine
    function c_nc_dim_size(f, name, n) bind(C, name="ecckd_nc_dim_size") result(rc)
      import c_char, c_ptr, c_int
      type(c_ptr), value :: f
      character(kind=c_char), dimension(*), intent(in) :: name
      integer(c_int), intent(out) :: n
      integer(c_int) :: rc
    end function
    function c_nc_var_exists(f, name) bind(C, name="ecckd_nc_var_exists") result(rc)
      import c_char, c_ptr, c_int
      type(c_ptr), value :: f
      character(kind=c_char), dimension(*), intent(in) :: name
      integer(c_int) :: rc
    end function
    function c_nc_var_size(f, name, n) bind(C, name="ecckd_nc_var_size") result(rc)
      import c_char, c_ptr, c_int, c_long_long
      type(c_ptr), value :: f
      character(kind=c_char), dimension(*), intent(in) :: name
      integer(c_long_long), intent(out) :: n
      integer(c_int) :: rc
    end function
    function c_nc_read(f, name, out, n) bind(C, name="ecckd_nc_read_f64") result(rc)
      import c_char, c_ptr, c_int, c_long_long, c_double
      type(c_ptr), value :: f
      character(kind=c_char), dimension(*), intent(in) :: name
      real(c_double), dimension(*), intent(out) :: out
      integer(c_long_long), value :: n
      integer(c_int) :: rc
    end function
    function c_nc_att(f, var, att, buf, buflen) bind(C, name="ecckd_nc_get_att_text") result(rc)
      import c_char, c_ptr, c_int
      type(c_ptr), value :: f
      character(kind=c_char), dimension(*), intent(in) :: var, att
      character(kind=c_char), dimension(*), intent(inout) :: buf
      integer(c_int), value :: buflen
      integer(c_int) :: rc
    end function
    function c_nc_write(path, var, values, n) bind(C, name="ecckd_nc_write_f64") result(rc)
      import c_char, c_int, c_long_long, c_double
      character(kind=c_char), dimension(*), intent(in) :: path, var
      real(c_double), dimension(*), intent(in) :: values
      integer(c_long_long), value :: n
      integer(c_int) :: rc
    end function
  end interface

contains

  !> Print the message and stop with status 1 when it is not blank (mo_simple_netcdf.F90:331-339).
  subroutine stop_on_err(msg)
    character(len=*), intent(in) :: msg
    if (len_trim(msg) > 0) then
      write(error_unit, "(a)") " " // trim(msg)
      stop 1
    end if
  end subroutine stop_on_err

  function nc_open(filename, who) result(f)
    character(len=*), intent(in) :: filename, who
    type(nc_file) :: f
    if (c_nc_open(trim(filename) // c_null_char, f%h) /= 0) &
      call stop_on_err(trim(who) // ": can't find file " // trim(filename))
  end function nc_open

  subroutine nc_close(f)
    type(nc_file), intent(inout) :: f
    if (c_associated(f%h)) call c_nc_close(f%h)
    f%h = c_null_ptr
  end subroutine nc_close

  integer function get_dim_size(f, dimname)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: dimname
    integer(c_int) :: n
    if (c_nc_dim_size(f%h, trim(dimname) // c_null_char, n) /= 0) call stop_on_err(c_error_message())
    get_dim_size = int(n)
  end function get_dim_size

  logical function var_exists(f, varname)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: varname
    var_exists = c_nc_var_exists(f%h, trim(varname) // c_null_char) /= 0
  end function var_exists

  !> read_field: the whole variable, flat, in Fortran order of its (reversed) netCDF shape
  function read_all(f, varname, n) result(v)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: varname
    integer, intent(in) :: n
    real(wp), dimension(n) :: v
    integer(c_long_long) :: nfile
    if (c_nc_var_size(f%h, trim(varname) // c_null_char, nfile) /= 0) call stop_on_err(c_error_message())
    if (nfile /= int(n, c_long_long)) call stop_on_err("read_field: unexpected size of variable " // trim(varname))
    if (c_nc_read(f%h, trim(varname) // c_null_char, v, nfile) /= 0) call stop_on_err(c_error_message())
  end function read_all

  !> read_scaling (mo_rfmip_io.F90:266-282): the numeric value of the variable's "units" attribute
  function read_units_scaling(f, varname) result(s)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: varname
    real(wp) :: s
    character(kind=c_char), dimension(64) :: buf
    character(len=64) :: units
    integer :: i
    if (c_nc_var_exists(f%h, trim(varname) // c_null_char) == 0) &
      call stop_on_err("read_scaling: can't find variable " // trim(varname))
    buf = c_null_char
    if (c_nc_att(f%h, trim(varname) // c_null_char, "units" // c_null_char, buf, 64_c_int) /= 0) &
      call stop_on_err("read_scaling: can't read attribute 'units' from variable " // trim(varname))
    units = ""
    do i = 1, 63
      if (buf(i) == c_null_char) exit
      units(i:i) = buf(i)
    end do
    read(units, *) s
  end function read_units_scaling

  !> write_field into an existing variable of an existing file
  function write_field(filename, varname, values) result(error_msg)
    character(len=*), intent(in) :: filename, varname
    real(wp), dimension(:), intent(in) :: values
    character(len=128) :: error_msg
    error_msg = ""
    if (c_nc_write(trim(filename) // c_null_char, trim(varname) // c_null_char, values, &
                   int(size(values), c_long_long)) /= 0) error_msg = c_error_message()
  end function write_field
end module simple_netcdf


module load_coefficients
  use gas_optics_ecckd, only: ty_gas_optics_ecckd
  use mo_gas_concentrations, only: ty_gas_concs
  use simple_netcdf, only: stop_on_err
  implicit none
  private
  public :: load_and_init
contains
  !> Same call as the reference's loader (mo_load_coefficients.F90:19-23).
  subroutine load_and_init(ecckd, filename, available_gases)
    class(ty_gas_optics_ecckd), intent(inout) :: ecckd
    character(len=*), intent(in) :: filename
    class(ty_gas_concs), intent(in) :: available_gases
    call stop_on_err(ecckd%load(filename, available_gases))
  end subroutine load_and_init
end module load_coefficients


module rfmip_io
  use mo_rte_kind, only: wp
  use mo_gas_concentrations, only: ty_gas_concs
  use simple_netcdf
  implicit none
  private
  public :: read_size, read_and_block_pt, read_and_block_lw_bc, read_and_block_sw_bc, &
            read_and_block_gases_ty, unblock_and_write

  integer :: ncol_l = 0, nlay_l = 0, nexp_l = 0     ! module state, as mo_rfmip_io.F90:19-21

contains

  subroutine need_size(who, blocksize, nblocks)
    character(len=*), intent(in) :: who
    integer, intent(in) :: blocksize
    integer, intent(out) :: nblocks
    if (ncol_l == 0 .or. nlay_l == 0 .or. nexp_l == 0) call stop_on_err(who // ": haven't read problem size yet.")
    if (mod(ncol_l * nexp_l, blocksize) /= 0) &
      call stop_on_err(who // ": number of columns doesn't fit evenly into blocks.")
    nblocks = (ncol_l * nexp_l) / blocksize
  end subroutine need_size

  !> Problem size: sites, layers, experiments (mo_rfmip_io.F90:28-48).
  subroutine read_size(filename, ncol, nlay, nexp)
    character(len=*), intent(in) :: filename
    integer, intent(out) :: ncol, nlay, nexp
    type(nc_file) :: f
    f = nc_open(filename, "read_size")
    ncol = get_dim_size(f, "site")
    nlay = get_dim_size(f, "layer")
    if (get_dim_size(f, "level") /= nlay + 1) call stop_on_err("read_size: number of levels should be nlay+1")
    nexp = get_dim_size(f, "expt")
    call nc_close(f)
    ncol_l = ncol
    nlay_l = nlay
    nexp_l = nexp
  end subroutine read_size

  !> (nz, site[, expt]) field of the file -> (blocksize, nz, nblocks); a field without the expt
  !! dimension is repeated for every experiment (the `spread` of mo_rfmip_io.F90:78-99).
  subroutine block_profile(f, varname, nz, per_expt, blocksize, nblocks, scale, out)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: varname
    integer, intent(in) :: nz, blocksize, nblocks
    logical, intent(in) :: per_expt
    real(wp), intent(in) :: scale
    real(wp), dimension(:,:,:), allocatable, intent(out) :: out
    real(wp), dimension(:), allocatable :: raw
    integer :: k, c, isite, iexp, b, i
    allocate(out(blocksize, nz, nblocks))
    if (per_expt) then
      raw = read_all(f, varname, nz * ncol_l * nexp_l)
    else
      raw = read_all(f, varname, nz * ncol_l)
    end if
    do c = 0, ncol_l * nexp_l - 1           ! flattened column index, site fastest
      isite = mod(c, ncol_l)
      iexp = c / ncol_l
      b = c / blocksize + 1
      i = mod(c, blocksize) + 1
      do k = 1, nz
        if (per_expt) then
          out(i, k, b) = raw(k + nz * (isite + ncol_l * iexp)) * scale
        else
          out(i, k, b) = raw(k + nz * isite) * scale
        end if
      end do
    end do
  end subroutine block_profile

  !> (site[, expt]) field -> (blocksize, nblocks)
  subroutine block_surface(f, varname, per_expt, blocksize, nblocks, out)
    type(nc_file), intent(in) :: f
    character(len=*), intent(in) :: varname
    logical, intent(in) :: per_expt
    integer, intent(in) :: blocksize, nblocks
    real(wp), dimension(:,:), allocatable, intent(out) :: out
    real(wp), dimension(:), allocatable :: raw
    integer :: c
    allocate(out(blocksize, nblocks))
    if (per_expt) then
      raw = read_all(f, varname, ncol_l * nexp_l)
    else
      raw = read_all(f, varname, ncol_l)
    end if
    do c = 0, ncol_l * nexp_l - 1
      if (per_expt) then
        out(mod(c, blocksize) + 1, c / blocksize + 1) = raw(c + 1)
      else
        out(mod(c, blocksize) + 1, c / blocksize + 1) = raw(mod(c, ncol_l) + 1)
      end if
    end do
  end subroutine block_surface

  !> Layer/level pressures and temperatures (mo_rfmip_io.F90:53-102).
  subroutine read_and_block_pt(filename, blocksize, p_lay, p_lev, t_lay, t_lev)
    character(len=*), intent(in) :: filename
    integer, intent(in) :: blocksize
    real(wp), dimension(:,:,:), allocatable, intent(out) :: p_lay, p_lev, t_lay, t_lev
    type(nc_file) :: f
    integer :: nblocks
    call need_size("read_and_block_pt", blocksize, nblocks)
    f = nc_open(filename, "read_and_block_pt")
    call block_profile(f, "pres_layer", nlay_l, .false., blocksize, nblocks, 1._wp, p_lay)
    call block_profile(f, "temp_layer", nlay_l, .true., blocksize, nblocks, 1._wp, t_lay)
    call block_profile(f, "pres_level", nlay_l + 1, .false., blocksize, nblocks, 1._wp, p_lev)
    call block_profile(f, "temp_level", nlay_l + 1, .true., blocksize, nblocks, 1._wp, t_lev)
    call nc_close(f)
  end subroutine read_and_block_pt

  !> Shortwave boundary conditions (mo_rfmip_io.F90:106-140).
  subroutine read_and_block_sw_bc(filename, blocksize, surface_albedo, total_solar_irradiance, solar_zenith_angle)
    character(len=*), intent(in) :: filename
    integer, intent(in) :: blocksize
    real(wp), dimension(:,:), allocatable, intent(out) :: surface_albedo, total_solar_irradiance, solar_zenith_angle
    type(nc_file) :: f
    integer :: nblocks
    call need_size("read_and_block_sw_bc", blocksize, nblocks)
    f = nc_open(filename, "read_and_block_sw_bc")
    call block_surface(f, "surface_albedo", .false., blocksize, nblocks, surface_albedo)
    call block_surface(f, "total_solar_irradiance", .false., blocksize, nblocks, total_solar_irradiance)
    call block_surface(f, "solar_zenith_angle", .false., blocksize, nblocks, solar_zenith_angle)
    call nc_close(f)
  end subroutine read_and_block_sw_bc

  !> Longwave boundary conditions (mo_rfmip_io.F90:144-173).
  subroutine read_and_block_lw_bc(filename, blocksize, surface_emissivity, surface_temperature)
    character(len=*), intent(in) :: filename
    integer, intent(in) :: blocksize
    real(wp), dimension(:,:), allocatable, intent(out) :: surface_emissivity, surface_temperature
    type(nc_file) :: f
    integer :: nblocks
    call need_size("read_and_block_lw_bc", blocksize, nblocks)
    f = nc_open(filename, "read_and_block_lw_bc")
    call block_surface(f, "surface_emissivity", .false., blocksize, nblocks, surface_emissivity)
    call block_surface(f, "surface_temperature", .true., blocksize, nblocks, surface_temperature)
    call nc_close(f)
  end subroutine read_and_block_lw_bc

  logical function is_3d_gas(name)
    character(len=*), intent(in) :: name
    is_3d_gas = trim(name) == "h2o" .or. trim(name) == "o3" .or. trim(name) == "no2"
  end function is_3d_gas

  !> One ty_gas_concs per block (mo_rfmip_io.F90:177-263): water vapour and ozone are (site, layer,
  !! expt) fields, every other gas a per-experiment global mean `<name>_GM`, each scaled by the
  !! numeric `units` attribute; no2 is set to zero; gas order = gas_names, then h2o, o3, no2.
  subroutine read_and_block_gases_ty(filename, blocksize, gas_names, names_in_file, gas_conc_array)
    character(len=*), intent(in) :: filename
    integer, intent(in) :: blocksize
    character(len=*), dimension(:), intent(in) :: gas_names, names_in_file
    type(ty_gas_concs), dimension(:), allocatable, intent(out) :: gas_conc_array
    type(nc_file) :: f
    integer :: nblocks, b, g, i, c
    logical :: has_3d
    character(len=32), dimension(:), allocatable :: all_names
    real(wp), dimension(:,:,:), allocatable :: field
    real(wp), dimension(:), allocatable :: gm
    real(wp), dimension(:,:), allocatable :: col2d
    integer, dimension(:), allocatable :: expt_of
    call need_size("read_and_block_gases_ty", blocksize, nblocks)
    has_3d = .false.
    do g = 1, size(gas_names)
      has_3d = has_3d .or. is_3d_gas(gas_names(g))
    end do
    if (has_3d) then
      allocate(all_names(size(gas_names)))
      all_names = gas_names
    else
      allocate(all_names(size(gas_names) + 3))
      all_names(1:size(gas_names)) = gas_names
      all_names(size(gas_names) + 1) = "h2o"
      all_names(size(gas_names) + 2) = "o3"
      all_names(size(gas_names) + 3) = "no2"
    end if
    allocate(gas_conc_array(nblocks))
    do b = 1, nblocks
      call stop_on_err(gas_conc_array(b)%init(all_names))
    end do

    f = nc_open(filename, "read_and_block_gases_ty")
    call block_profile(f, "water_vapor", nlay_l, .true., blocksize, nblocks, read_units_scaling(f, "water_vapor"), field)
    do b = 1, nblocks
      call stop_on_err(gas_conc_array(b)%set_vmr("h2o", field(:, :, b)))
    end do
    deallocate(field)
    call block_profile(f, "ozone", nlay_l, .true., blocksize, nblocks, read_units_scaling(f, "ozone"), field)
    do b = 1, nblocks
      call stop_on_err(gas_conc_array(b)%set_vmr("o3", field(:, :, b)))
    end do

    allocate(expt_of(blocksize), col2d(blocksize, nlay_l))
    do g = 1, size(gas_names)
      if (is_3d_gas(gas_names(g))) cycle
      gm = read_all(f, trim(names_in_file(g)) // "_GM", nexp_l) * read_units_scaling(f, trim(names_in_file(g)) // "_GM")
      do b = 1, nblocks
        do i = 1, blocksize
          c = (b - 1) * blocksize + i - 1
          expt_of(i) = c / ncol_l + 1
        end do
        if (all(expt_of == expt_of(1))) then           ! one experiment in the block: a scalar
          call stop_on_err(gas_conc_array(b)%set_vmr(trim(gas_names(g)), gm(expt_of(1))))
        else
          do i = 1, blocksize
            col2d(i, :) = gm(expt_of(i))
          end do
          call stop_on_err(gas_conc_array(b)%set_vmr(trim(gas_names(g)), col2d))
        end if
      end do
    end do
    do b = 1, nblocks
      call stop_on_err(gas_conc_array(b)%set_vmr("no2", 0._wp))
    end do
    call nc_close(f)
  end subroutine read_and_block_gases_ty

  !> (blocksize, nlev, nblocks) -> RFMIP order (nlev, site, expt), written into the existing
  !! variable `varname` of the existing file (mo_rfmip_io.F90:288-317).
  subroutine unblock_and_write(filename, varname, values)
    character(len=*), intent(in) :: filename, varname
    real(wp), dimension(:,:,:), intent(in) :: values
    real(wp), dimension(:), allocatable :: flat
    integer :: blocksize, nlev, nblocks, b, i, k, c
    if (ncol_l == 0 .or. nlay_l == 0 .or. nexp_l == 0) call stop_on_err("unblock_and_write: haven't read problem size yet.")
    blocksize = size(values, 1)
    nlev = size(values, 2)
    nblocks = size(values, 3)
    if (nlev /= nlay_l + 1) call stop_on_err("unblock_and_write: array values has the wrong number of levels")
    if (blocksize * nblocks /= ncol_l * nexp_l) &
      call stop_on_err("unblock_and_write: array values has the wrong number of blocks/size")
    allocate(flat(nlev * ncol_l * nexp_l))
    do b = 1, nblocks
      do i = 1, blocksize
        c = (b - 1) * blocksize + i - 1
        do k = 1, nlev
          flat(k + nlev * c) = values(i, k, b)
        end do
      end do
    end do
    call stop_on_err(write_field(filename, varname, flat))
  end subroutine unblock_and_write
end module rfmip_io


module utils
  use, intrinsic :: iso_fortran_env, only: error_unit
  use simple_netcdf, only: stop_on_err
  implicit none
  private
  public :: determine_gas_names, parse_args
contains

  subroutine usage()
    character(len=256) :: command
    call get_command_argument(0, command)
    write(error_unit, "(a)") " Usage: " // trim(command) // " rfmip_file ecckd_file"
  end subroutine usage

  subroutine help()
    call usage()
    write(error_unit, "(a)") " " // ""
    write(error_unit, "(a)") " " // "Args:"
    write(error_unit, "(a)") " " // "rfmip_file - RFMIP input file."
    write(error_unit, "(a)") " " // "ecckd_file - ecckd input file."
    write(error_unit, "(a)") " " // "-f [1,2] - Forcing index."
    write(error_unit, "(a)") " " // "-h|--help - Prints this help message."
    write(error_unit, "(a)") " " // "-p [1,2] - Physics index."
    write(error_unit, "(a)") " " // "-b n - Columns per block (default: all columns in one block; the reference uses 1)."
    write(error_unit, "(a)") " " // "-n n - Process only the first n blocks (the reference processes 1700)."
    write(error_unit, "(a)") " " // "-d - Device-resident optical properties and sources (only inputs and fluxes cross the bus)."
  end subroutine help

  !> Gas names in the k-distribution and in the RFMIP file by forcing index (utils.f90:41-70).
  subroutine determine_gas_names(forcing_index, names_in_kdist, names_in_rfmip)
    integer, intent(in) :: forcing_index
    character(len=32), dimension(:), intent(inout) :: names_in_kdist, names_in_rfmip
    names_in_kdist(1:6) = [character(len=32) :: "co2", "ch4", "n2o", "o2", "cfc11", "cfc12"]
    names_in_rfmip(1:6) = [character(len=32) :: "carbon_dioxide", "methane", "nitrous_oxide", "oxygen", "cfc11", "cfc12"]
    if (forcing_index == 2) then
      names_in_rfmip(5) = "cfc11eq"
    else if (forcing_index /= 1) then
      call stop_on_err("forcing index must equal 1 or 2.")
    end if
  end subroutine determine_gas_names

  !> rfmip_file ecckd_file [-f 1|2] [-p 1|2] [-b block] [-n nblocks] [-h|--help]   (utils.f90:74-134)
  subroutine parse_args(rfmip_path, ecckd_path, forcing_index, physics_index, block_size, max_blocks, device_resident)
    character(len=*), intent(inout) :: rfmip_path, ecckd_path
    integer, intent(inout) :: forcing_index, physics_index
    integer, intent(inout), optional :: block_size, max_blocks
    logical, intent(inout), optional :: device_resident
    character(len=512) :: buffer
    integer :: i, npos, nargs
    forcing_index = 1
    physics_index = 1
    if (present(block_size)) block_size = 0
    if (present(max_blocks)) max_blocks = 0
    if (present(device_resident)) device_resident = .false.
    nargs = command_argument_count()
    if (nargs < 2) then
      call usage()
      stop 1
    end if
    i = 1
    npos = 0
    do while (i <= nargs)
      call get_command_argument(i, buffer)
      select case (trim(buffer))
      case ("-h", "--help")
        call help()
        stop 0
      case ("-d")
        if (present(device_resident)) device_resident = .true.
      case ("-f", "-p", "-b", "-n")
        if (i + 1 > nargs) then
          call usage()
          stop 1
        end if
        call take_int(trim(buffer), i + 1)
        i = i + 1
      case default
        npos = npos + 1
        if (npos == 1) then
          rfmip_path = trim(buffer)
        else if (npos == 2) then
          ecckd_path = trim(buffer)
        else
          call usage()
          stop 1
        end if
      end select
      i = i + 1
    end do
  contains
    subroutine take_int(flag, iarg)
      character(len=*), intent(in) :: flag
      integer, intent(in) :: iarg
      character(len=64) :: val
      integer :: v, ios
      call get_command_argument(iarg, val)
      read(val, *, iostat=ios) v
      if (ios /= 0) call stop_on_err("bad value for " // flag)
      select case (flag)
      case ("-f")
        if (v < 1 .or. v > 2) call stop_on_err("forcing index must be either 1 or 2.")
        forcing_index = v
      case ("-p")
        if (v < 1 .or. v > 2) call stop_on_err("physics index must be either 1 or 2.")
        physics_index = v
      case ("-b")
        if (present(block_size)) block_size = v
      case ("-n")
        if (present(max_blocks)) max_blocks = v
      end select
    end subroutine take_int
  end subroutine parse_args
end module utils
