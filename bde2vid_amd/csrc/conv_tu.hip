// Translation unit of the convolution-shaped kernels (conv_mfma.h, conv_vec.h, pw_gemm.h, lstm16.h): their many
// template instantiations compile here, in parallel with bde_api.hip, which only sees the launcher declarations.
#define BDE_CONV_TU 1
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "common.h"
#include "conv_mfma.h"
#include "conv_vec.h"
#include "conv_sb.h"
#include "lstm16.h"
#include "lstm_sb.h"
#include "pw_gemm.h"

namespace bde {

// resident workgroups per CU the runtime reports for a named kernel of this translation unit (-1 = unknown name)
int conv_tu_occupancy(const char* kernel) {
    int nb = -1;
    const std::string k(kernel ? kernel : "");
    hipError_t e = hipErrorInvalidValue;
    if (k == "lstm16_1_64") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<1, 64, 1, false>, 256, 0);
    else if (k == "lstm16_1_128_s2") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<1, 128, 2, false>, 256, 0);
    else if (k == "lstm16_2_32") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm16_step_kernel<2, 32, 1, false>, 256, 0);
    else if (k == "conv_k3_m2n2") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_mfma_kernel<3, 1, 2, 2, 8, false, EPI_GENERIC, conv_maxi(3)>, 256, 42 * 1024);
    else if (k == "lstm_sb_l0") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_sb_step_kernel<4, 1, 2, 6, true>, 256, lstm_sb_shape(64, 92, 120).lds);
    else if (k == "lstm_sb_l1") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_sb_step_kernel<2, 2, 2, 12, false>, 256, lstm_sb_shape(128, 46, 60).lds);
    else if (k == "lstm_sb_l2") e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_sb_step_kernel<1, 4, 3, 24, false>, 256, lstm_sb_shape(256, 23, 30).lds);
    if (e != hipSuccess) return -1;
    return nb;
}

}  // namespace bde
