// Translation unit of the split-operand kernels (conv_sb.h, lstm_sb.h; formats in split.h): both operand formats of every
// shape compile here, in parallel with bde_api.hip and conv_tu.hip.
#define BDE_SB_TU 1
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "common.h"
#include "conv_mfma.h"
#include "conv_sb.h"
#include "lstm_sb.h"

namespace bde {

// resident workgroups per CU the runtime reports for the recurrent step of config A's levels in the default format (-1 = unknown name)
int sb_tu_occupancy(const char* kernel) {
    int nb = -1;
    const std::string k(kernel ? kernel : "");
    constexpr int T_ = BDE_DEFAULT_SB_TERMS;
    hipError_t e = hipErrorInvalidValue;
    auto occ = [&](int Ch, int H, int W) {
        const LstmSbShape s = lstm_sb_shape(Ch, H, W, T_);
        if (!s.ok) return hipErrorInvalidValue;
        const void* f = lstm_sb_kernel_ptr(s, T_);
        return f ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 256, s.lds) : hipErrorInvalidValue;
    };
    if (k == "lstm_sb_l0") e = occ(64, 92, 120);
    else if (k == "lstm_sb_l1") e = occ(128, 46, 60);
    else if (k == "lstm_sb_l2") e = occ(256, 23, 30);
    if (e != hipSuccess) return -1;
    return nb;
}

}  // namespace bde
