// Fused n_fft == 512 throughput kernel (placeholder until the register-resident FFT kernel lands).
#include "mfcc.hpp"

namespace ssp {

bool mfcc_fast_supported(const ssp_mfcc_cfg&) { return false; }

int launch_mfcc_fast(const MfccArgs&, const ssp_mfcc_cfg&, int, int, int, hipStream_t) {
    SSP_FAIL(SSP_ERR_UNSUPPORTED, "mfcc fast kernel not built");
}

}  // namespace ssp
