// The two-term f16 split (EKAID_GEMM_F16X2: per-row / per-column power-of-two scales, h + l, three v_mfma_f32_32x32x16_f16 products per
// product block) instantiations of the split GEMM kernels of gemm.hip -- NP = 2 --, compiled as a translation unit of their own so that
// `make -j` builds both arithmetics in parallel (and with -fno-slp-vectorize: the operand conversion wants scalar f32 multiplies).
#define EKAID_GEMM_X2_TU
#include "gemm.hip"
