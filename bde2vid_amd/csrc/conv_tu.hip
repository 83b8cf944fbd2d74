// Translation unit of the fp32 convolution-shaped kernels (conv_mfma.h, conv_vec.h, pw_gemm.h, lstm16.h): their many
// template instantiations compile here, in parallel with bde_api.hip and sb_tu.hip (the split-operand kernels), which
// only see the launcher declarations.
#define BDE_CONV_TU 1
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "common.h"
#include "conv_mfma.h"
#include "conv_vec.h"
#include "lstm16.h"
#include "pw_gemm.h"

namespace bde {

int sb_tu_occupancy(const char* kernel);   // sb_tu.hip

// resident workgroups per CU the runtime reports for a named kernel of this translation unit (-1 = unknown name)
int conv_tu_occupancy(const char* kernel) {
    int nb = -1;
    const std::string k(kernel ? kernel : "");
    hipError_t e = hipErrorInvalidValue;
    if (k == "lstm16_1_64") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<1, 64, 1, false>, 256, 0);
    else if (k == "lstm16_1_128_s2") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<1, 128, 2, false>, 256, 0);
    else if (k == "lstm16_2_32") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<2, 32, 1, false>, 256, 0);
    else if (k == "conv_k3_m2n2") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_mfma_kernel<3, 1, 2, 2, 8, false, EPI_GENERIC, conv_maxi(3)>, 256, 42 * 1024);
    else return sb_tu_occupancy(kernel);
    if (e != hipSuccess) return -1;
    return nb;
}

}  // namespace bde
