// oracle/ref_build/blas_rename.h -- TEST INFRASTRUCTURE (oracle), not product code.
// Force-included (-include) when compiling the reference's src/cpucompute so that its
// CBLAS / LAPACK calls (declared by /root/reference/tools/CLAPACK/{cblas,clapack}.h)
// bind to SciPy's bundled LP64 OpenBLAS, whose exported symbols carry a scipy_ prefix.
#ifndef EESEN_ORACLE_BLAS_RENAME_H_
#define EESEN_ORACLE_BLAS_RENAME_H_
#define cblas_dasum scipy_cblas_dasum
#define cblas_daxpy scipy_cblas_daxpy
#define cblas_dcopy scipy_cblas_dcopy
#define cblas_ddot scipy_cblas_ddot
#define cblas_dgbmv scipy_cblas_dgbmv
#define cblas_dgemm scipy_cblas_dgemm
#define cblas_dgemv scipy_cblas_dgemv
#define cblas_dger scipy_cblas_dger
#define cblas_drot scipy_cblas_drot
#define cblas_dsbmv scipy_cblas_dsbmv
#define cblas_dscal scipy_cblas_dscal
#define cblas_dspmv scipy_cblas_dspmv
#define cblas_dspr scipy_cblas_dspr
#define cblas_dspr2 scipy_cblas_dspr2
#define cblas_dsymm scipy_cblas_dsymm
#define cblas_dsyrk scipy_cblas_dsyrk
#define cblas_dtpmv scipy_cblas_dtpmv
#define cblas_dtpsv scipy_cblas_dtpsv
#define cblas_sasum scipy_cblas_sasum
#define cblas_saxpy scipy_cblas_saxpy
#define cblas_scopy scipy_cblas_scopy
#define cblas_sdot scipy_cblas_sdot
#define cblas_sgbmv scipy_cblas_sgbmv
#define cblas_sgemm scipy_cblas_sgemm
#define cblas_sgemv scipy_cblas_sgemv
#define cblas_sger scipy_cblas_sger
#define cblas_srot scipy_cblas_srot
#define cblas_ssbmv scipy_cblas_ssbmv
#define cblas_sscal scipy_cblas_sscal
#define cblas_sspmv scipy_cblas_sspmv
#define cblas_sspr scipy_cblas_sspr
#define cblas_sspr2 scipy_cblas_sspr2
#define cblas_ssymm scipy_cblas_ssymm
#define cblas_ssyrk scipy_cblas_ssyrk
#define cblas_stpmv scipy_cblas_stpmv
#define cblas_stpsv scipy_cblas_stpsv
#define cblas_dtrsm scipy_cblas_dtrsm
#define cblas_strsm scipy_cblas_strsm
#define cblas_dsyr scipy_cblas_dsyr
#define cblas_ssyr scipy_cblas_ssyr
#define cblas_dnrm2 scipy_cblas_dnrm2
#define cblas_snrm2 scipy_cblas_snrm2
#define dgesvd_ scipy_dgesvd_
#define dgetrf_ scipy_dgetrf_
#define dgetri_ scipy_dgetri_
#define dsptrf_ scipy_dsptrf_
#define dsptri_ scipy_dsptri_
#define dtptri_ scipy_dtptri_
#define sgesvd_ scipy_sgesvd_
#define sgetrf_ scipy_sgetrf_
#define sgetri_ scipy_sgetri_
#define ssptrf_ scipy_ssptrf_
#define ssptri_ scipy_ssptri_
#define stptri_ scipy_stptri_
#endif
