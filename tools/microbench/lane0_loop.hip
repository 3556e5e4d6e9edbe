// Reduction of the mfcc_stream2048_kernel hang of round 4 (and of two formulations of round 5's mfcc_stream512_kernel that did the same):
// a PERSISTENT wave loop whose body begins with a lane-0 block (the claim: one lane takes the next work item from a global counter, the
// wave reads it back with readfirstlane) and ENDS with another lane-0 block (one lane publishes the wave's result).  Across the loop's
// back edge the two blocks are neighbours; the compiler threads the other 63 lanes around both, the loop is no longer wave-uniform and
// is rebuilt as a loop over LANE MASKS (`s_andn2_b64 exec, exec, s[..]` + `s_cbranch_execz` as its test) in which lanes leave
// SEPARATELY — although readfirstlane, ballots and shuffles in the body are convergent operations of the whole wave.  FORM 3's ISA,
// read block by block: after the bottom block the exit mask is `~exec` taken INSIDE the lane-0 region = lanes 1..63, so those leave
// after the first trip and lane 0 walks every remaining item alone (its shuffles read dead lanes: wrong verdicts, no hang).  In the
// real kernels the other lanes have stores of their own, the split falls the other way round, and the claim runs with lane 0 masked
// off: `s_and_saveexec ..., <lane 0>` skips the atomic, readfirstlane returns the initial 0 of the first live lane, the exit test
// `item >= n_items` never fires, and the wave re-walks item 0 for ever — the hang of mfcc_stream2048_kernel (round 4: the lane-0
// atomics on utt_max at the bottom of its chunk loop, taken only in multi-chunk = ragged batches) and of two round-5 formulations of
// mfcc_stream512_kernel (a `continue` behind a lane-0 append; a lane-0 flag store at the bottom).
// COMPILE-ONLY evidence (do not run FORM 3 / 4: the loop they compile to is not the loop that was written):
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -DFORM=<n> lane0_loop.hip -o - | grep -c "s_andn2_b64 exec, exec"
//   FORM 0: atomics in a lane-0 block at the bottom (in this small kernel the loop survives: 0; in the 2048-point kernel it did not)
//   FORM 1: the cure — the store comes from ALL lanes through a bounds-checked buffer store, every lane but one out of range: 0
//   FORM 2: FORM 0 + a wave-uniform statement between the block and the back edge: 0
//   FORM 3: a lane-0 block of plain stores at the bottom: 1        FORM 4: `continue` behind a lane-0 append: 1
// tests/test_isa_guards.py asserts that no persistent-loop kernel of the library has such a loop test behind its claim.
#include <hip/hip_runtime.h>

__global__ void lane0_loop_kernel(int* counter, int n_items, const float* in, float* out_max) {
    const int lane = threadIdx.x & 63;
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(counter, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        float v = in[item * 64 + lane];
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
#if FORM == 0
        if (lane == 0 && v > -INFINITY) {
            if (v >= 0.f) atomicMax(reinterpret_cast<int*>(out_max + item), __float_as_int(v));
            else atomicMin(reinterpret_cast<unsigned*>(out_max + item), __float_as_uint(v));
        }
#elif FORM == 1
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out_max, 0, n_items * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane == 0 ? item * 4 : 0x7ffffff0, 0, 0);
#elif FORM == 2
        if (lane == 0 && v > -INFINITY) {
            if (v >= 0.f) atomicMax(reinterpret_cast<int*>(out_max + item), __float_as_int(v));
            else atomicMin(reinterpret_cast<unsigned*>(out_max + item), __float_as_uint(v));
        }
        asm volatile("s_nop 0");
#elif FORM == 3
        const bool flagged = __builtin_amdgcn_ballot_w64(v != v) != 0;
        if (lane == 0) {
            out_max[item] = flagged ? 1.f : 0.f;
            if (flagged) counter[1] = 1;
        }
#elif FORM == 4
        const bool flagged = __builtin_amdgcn_ballot_w64(v != v) != 0;
        if (flagged) {
            if (lane == 0) out_max[atomicAdd(counter + 1, 1)] = (float)item;
            continue;
        }
        out_max[n_items + item * 64 + lane] = v * 2.f;
#endif
    }
}

