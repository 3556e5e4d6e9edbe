// The THIRD kernel of a wave-stream MFCC launch: mfcc_stream512_kernel<..., WALK = 1> (mfcc_stream_kernel.hpp) walks the chunks the scan
// kernel flagged — a time step's window held a non-finite cepstrum: a digitally silent frame (ln 0 = -inf, GMM_UBM.py:89 / d_vector.py:96-98),
// a NaN sample — once more, sequentially, with every step formed term by term as GMM_UBM.py:53-69 forms it and the legacy product on every
// window row, and rewrites their rows (and scales them: CM).  Its own translation unit: the instances compile beside the first kernel's.
#include "mfcc_stream_kernel.hpp"

namespace ssp {

int launch_mfcc_stream_walk(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream) {
    return launch_mfcc_stream_impl<1>(args, p, n_chunks, stream, false);
}

}  // namespace ssp
