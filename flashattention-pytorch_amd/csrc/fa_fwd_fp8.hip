// FA3-style fp8 forward — placeholder translation unit until the e4m3 kernel lands: reports "unsupported" so
// fa3_forward(fp8=True) fails loudly (FA_ERR_UNSUPPORTED) instead of silently running the 16-bit path.
#include "fa_kernels.h"
namespace fa {
bool fwd_fp8_supported(int, int64_t) { return false; }
hipError_t launch_fwd_fp8(const FwdArgs&, void*, hipStream_t) { return hipErrorNotSupported; }
size_t fwd_fp8_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }
}  // namespace fa
