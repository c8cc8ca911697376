// The reduced-precision (two bf16 terms per operand, three products: EKAID_GEMM_BF16X2) instantiations of the split GEMM
// kernels of gemm.hip, compiled as a translation unit of their own so that `make -j` builds both arithmetics in parallel.
#define EKAID_GEMM_X2_TU
#include "gemm.hip"
