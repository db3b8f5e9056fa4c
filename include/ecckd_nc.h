/*
 * ecckd_nc.h -- minimal netCDF-3 (classic / 64-bit-offset) access for the RFMIP-shaped drivers.
 *
 * The reference reads RFMIP inputs and writes fluxes through netcdf-fortran
 * (example/rfmip-rad-irf/mo_simple_netcdf.F90, mo_rfmip_io.F90).  Neither libnetcdf nor
 * netcdf-fortran is part of this build; these few calls are what rte-ecckd_amd/fortran/rfmip_io.F90
 * needs from them, on top of the library's own CDF reader (rte-ecckd_amd/csrc/cdf1.cpp).
 * Host-side I/O only -- nothing here touches the GPU.  Return 0 on success; message via
 * ecckd_last_error().
 */
#ifndef ECCKD_NC_H
#define ECCKD_NC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ecckd_nc ecckd_nc_t;

int ecckd_nc_open(const char *path, ecckd_nc_t **file);                       /* nf90_open(NOWRITE) */
void ecckd_nc_close(ecckd_nc_t *file);
int ecckd_nc_dim_size(const ecckd_nc_t *file, const char *dim, int *size);    /* get_dim_size, mo_simple_netcdf.F90 */
int ecckd_nc_var_exists(const ecckd_nc_t *file, const char *var);             /* var_exists */
int ecckd_nc_var_size(const ecckd_nc_t *file, const char *var, long long *n); /* product of the extents */
/* read_field: the whole variable widened to double, on-disk (C) order == Fortran order of the
 * reversed shape */
int ecckd_nc_read_f64(const ecckd_nc_t *file, const char *var, double *out, long long n);
/* text attribute of a variable (e.g. "units", read_scaling at mo_rfmip_io.F90:266-282);
 * var == "" or NULL reads a global attribute */
int ecckd_nc_get_att_text(const ecckd_nc_t *file, const char *var, const char *att, char *buf, int buflen);
/* write_field into an EXISTING variable of an existing file (the reference writes its fluxes into
 * pre-existing template files, mo_rfmip_io.F90:312-316): values are converted to the variable's
 * on-disk type. */
int ecckd_nc_write_f64(const char *path, const char *var, const double *values, long long n);

#ifdef __cplusplus
}
#endif
#endif
