// Internal interface between the two Winograd translation units (not part of the C ABI).
#pragma once
#include "fh_common.h"

// n-blocks of a weight panel that run together on one XCD (both kernels use the same block -> work mapping)
#ifndef W_RUN_N
#define W_RUN_N 8
#endif
constexpr int FH_WINO_RUN = W_RUN_N;

// conv_wino2.hip: 4-wave blocks of 64 co x 256 outputs, six transform points per wave (fp32 weights, vector loads)
int fh_wino2_launch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                    int phase_major, hipStream_t stream, const int* run_map, int n_runs);

// conv_wino3.hip: persistent 12-wave workgroups of two independent 6-wave teams, 64 co x 256 outputs per team tile
// (fp32 weights, vector loads).  The descriptor array must be followed by FH_WINO3_WS_BYTES zeroed bytes (work-list
// cursors; the kernel leaves them zeroed).
int fh_wino3_launch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                    int phase_major, hipStream_t stream, const int* run_map, int n_runs);
