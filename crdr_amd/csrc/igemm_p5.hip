// Explicit instantiations of the implicit-GEMM kernel, part 6 of 7 (split over translation units so the library builds in parallel).
#include "igemm_kernel.hpp"

namespace crdr {
template __global__ void igemm_kernel<4, 1, 1, 5, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
template __global__ void igemm_kernel<2, 2, 2, 2, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
template __global__ void igemm_kernel<2, 2, 1, 3, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
template __global__ void igemm_kernel<2, 2, 2, 1, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
template __global__ void igemm_kernel<1, 4, 1, 1, false>(const IgemmArgs, const IgemmTaps, const IgemmGroup);
}  // namespace crdr
