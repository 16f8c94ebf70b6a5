#include "fa_kernels.h"
namespace fa {
bool fwd_mfma_supported(int, int64_t) { return false; }
hipError_t launch_fwd_mfma(const FwdArgs&, hipStream_t) { return hipErrorNotSupported; }
bool bwd_mfma_supported(int, int64_t) { return false; }
hipError_t launch_bwd_mfma(const BwdArgs&, hipStream_t) { return hipErrorNotSupported; }
size_t bwd_mfma_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
bool fwd_fp8_supported(int, int64_t) { return false; }
hipError_t launch_fwd_fp8(const FwdArgs&, void*, hipStream_t) { return hipErrorNotSupported; }
size_t fwd_fp8_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
}
