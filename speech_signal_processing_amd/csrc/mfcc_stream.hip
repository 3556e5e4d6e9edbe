// Host side of the wave-stream MFCC kernel (mfcc_stream_kernel.hpp) and its FIRST kernel's instances; the second kernel's — the rare
// walk over chunks that held a non-finite cepstrum — are compiled in mfcc_stream_walk.hip.
#include "mfcc_stream_kernel.hpp"

namespace ssp {

// ------------------------------------------------------------------------------------------------ host side
// dense-band instance: <= 24 filterbank rows that the piece filterbank cannot hold, identity "DCT", no deltas, the sidekit front end
bool mfcc_stream_dense(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    return mfcc_fast_supported(c) && p->fast_ready && !getenv("SSP_MFCC_NO_STREAM") && p->fast.melv == 0 && p->args.dct_identity &&
           c.n_filt <= 24 && c.delta_order == 0 && c.win_len <= 416 && c.spec_power == 2 && c.preemph_mode != 0;
}

bool mfcc_stream_supported(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    const FastArgs& f = p->fast;
    if (!(mfcc_fast_supported(c) && p->fast_ready)) return false;
    if (getenv("SSP_MFCC_NO_STREAM")) return false;
    if (mfcc_stream_dense(p)) return true;
    const int ks = (c.n_filt + 3) / 4;
    return f.melv >= 2 && f.melv <= (ks <= 6 ? 4 : 5) && c.n_ceps == 13 && ks <= 10 && (c.delta_order == 0 || c.delta_N == 2);
}

// the instances that scale the features themselves (cmvn): the sidekit call-site family
bool mfcc_stream_fuses_cmvn(const ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    return mfcc_stream_supported(p) && !mfcc_stream_dense(p) && c.win_len <= 416 && c.spec_power == 2 && c.preemph_mode != 0 && (c.n_filt + 3) / 4 <= 6;
}


int build_stream_tables(ssp_mfcc_plan* p) {
    const ssp_mfcc_cfg& c = p->cfg;
    const int KS = stream_ks(c);
    // A operand of the DCT product: lane (ceps = l & 15, kq = l >> 4), k-step s <-> filter KS kq + s
    std::vector<float> dcth((size_t)c.n_ceps * c.n_filt), dA((size_t)KS * 64, 0.f);
    SSP_HIP(hipMemcpy(dcth.data(), p->dct.p, dcth.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int s = 0; s < KS; ++s)
        for (int l = 0; l < 64; ++l) {
            const int q = l & 15, jf = KS * (l >> 4) + s;
            if (q < c.n_ceps && jf < c.n_filt) dA[(size_t)s * 64 + l] = dcth[(size_t)q * c.n_filt + jf];
        }
    SSP_TRY(p->s_dctA.alloc(dA.size() * sizeof(float)));
    SSP_HIP(hipMemcpy(p->s_dctA.p, dA.data(), dA.size() * sizeof(float), hipMemcpyHostToDevice));
    if (mfcc_stream_dense(p)) {
        // lane (c = l & 15, b = l >> 4): bands 6 b + k on bins 16 c + i (i < 16) and, on chunk 15 only, bin 256; the split step leaves
        // 2 X[k], the scale of the spectrum is folded in (as the piece filterbank does)
        const int nb = 257;
        std::vector<float> fb((size_t)c.n_filt * nb), dw((size_t)64 * 6 * 20, 0.f);
        SSP_HIP(hipMemcpy(fb.data(), p->fbank_dense.p, fb.size() * sizeof(float), hipMemcpyDeviceToHost));
        const float pscale = (c.spec_power == 2 ? 0.25f : 0.5f) * c.spec_scale;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 6; ++k) {
                const int band = 6 * (l >> 4) + k, ch = l & 15;
                if (band >= c.n_filt) continue;
                float* d = dw.data() + ((size_t)l * 6 + k) * 20;
                for (int i = 0; i < 16; ++i) d[i] = pscale * fb[(size_t)band * nb + 16 * ch + i];
                if (ch == 15) d[16] = pscale * fb[(size_t)band * nb + 256];
            }
        SSP_TRY(p->s_dense.alloc(dw.size() * sizeof(float)));
        SSP_HIP(hipMemcpy(p->s_dense.p, dw.data(), dw.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    p->stream_ready = true;
    return SSP_OK;
}

int launch_mfcc_stream_walk(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream);  // mfcc_stream_walk.hip

int launch_mfcc_stream(const MfccArgs& args, ssp_mfcc_plan* p, int n_chunks, hipStream_t stream, bool dry_run) {
    SSP_TRY(launch_mfcc_stream_impl<0>(args, p, n_chunks, stream, dry_run));
    if (dry_run || n_chunks <= 0 || mfcc_stream_dense(p)) return SSP_OK;
    // the second kernel: the chunks the first one listed (normally none: every workgroup reads the count and leaves)
    return launch_mfcc_stream_walk(args, p, n_chunks, stream);
}

}  // namespace ssp

#ifdef SSP_S_CLOCK
extern "C" int ssp_debug_clock(ssp_mfcc_plan* p, unsigned long long* out2) {
    unsigned long long h[8];
    if (hipMemcpy(h, p->f_counter.p, 64, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    out2[0] = h[4];  // (64-bit words behind the launch counters: work_counter + 8, + 10)
    out2[1] = h[5];
    return 0;
}
#endif
