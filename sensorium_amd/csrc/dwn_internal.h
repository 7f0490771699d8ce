// Internal launch-argument structs and launcher prototypes shared by the kernel files and the C-ABI layer.
#pragma once
#include "dwn_common.h"

enum { EPI_STORE = DWN_EPI_STORE, EPI_READOUT = DWN_EPI_READOUT, EPI_DG = DWN_EPI_DG, EPI_STORE_CAT = DWN_EPI_STORE_CAT, EPI_DH3 = DWN_EPI_DH3 };
typedef dwn_gemm_nn_args GemmNN;
typedef dwn_gemm_tn_args GemmTN;
typedef dwn_dw_spatial_fwd_args DwSpatialFwd;
typedef dwn_dw_spatial_bwd_args DwSpatialBwd;
typedef dwn_dw_temporal_fwd_args DwTemporalFwd;
typedef dwn_dw_temporal_bwd_args DwTemporalBwd;

int launch_gemm_nn(const GemmNN& g, int dtype, hipStream_t s);
int launch_gemm_tn(const GemmTN& g, int dtype, hipStream_t s);
// 256-row / 8-wave LDS-DMA variant for the load-path-bound shapes (dwn_gemm_xl.hip)
bool gemm_nn_xl_eligible(const GemmNN& g, int dtype);
int launch_gemm_nn_xl(const GemmNN& g, hipStream_t s);
// A-direct kernel for the deep-K, narrow-N shapes: only the weights pass through LDS (dwn_gemm_kd.hip)
bool gemm_nn_kd_eligible(const GemmNN& g, int dtype);
int launch_gemm_nn_kd(const GemmNN& g, hipStream_t s);
int launch_dw_spatial_fwd(const DwSpatialFwd& a, int dtype, hipStream_t s);
int launch_dw_spatial_bwd(const DwSpatialBwd& a, int dtype, hipStream_t s);
int launch_dw_temporal_fwd(const DwTemporalFwd& a, int dtype, hipStream_t s);
int launch_dw_temporal_bwd(const DwTemporalBwd& a, int dtype, hipStream_t s);
bool pw_bwd_fused_supported(int dtype, long long M, int E, int Cin);
int launch_pw_bwd_fused(const void* dh1, const void* a0, const void* bp, const float* r3, void* da0, float* tacc,
                        long long M, int E, int Cin, int dtype, const void* res, const float* res_coef, int res_n,
                        const int* hinv, const int* winv, int Hin, int Win, int Hout, int Wout, hipStream_t s);
