// Counter-based random numbers of the training step (host + device: tests/test_host_model_math.py checks the host build against an
// independent restatement and the published known-answer vectors of Philox4x32-10).  The reference draws the opacity noise with
// randn_like (dbw.py:301) and the overlap samples with torch.rand (dbw.py:393) from torch's default generator; what has to hold is the
// distribution, and -- under view-sharded data parallelism -- that every rank draws the SAME numbers.  A counter-based
// generator keyed on (seed, step, stream, index) gives that without any state on the device: no generator kernel, no buffer, no
// lock step of ranks to maintain, and a replayed hipGraph draws fresh numbers as soon as the step counter it reads moves.
#pragma once
#include "raster_math.h"      // DBW_HD

namespace dbw {

struct Philox4 { uint32_t x, y, z, w; };

DBW_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
}

// Philox4x32-10 (Salmon et al., SC'11): 10 rounds, key schedule += (0x9E3779B9, 0xBB67AE85)
DBW_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_mulhilo(0xD2511F53u, c0, hi0, lo0);
        philox_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// the draw of (seed, step, stream, index): stream 0 = opacity noise, 1 = overlap samples
DBW_HD Philox4 step_random(uint64_t seed, uint64_t step, uint32_t stream, uint32_t index) {
    return philox4x32_10(index, stream, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform in [0, 1): the top 24 bits (torch's float path: every value is a multiple of 2^-24, 1 is never returned)
DBW_HD float uniform01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

// standard normal from two words (Box-Muller on u1 in (0, 1], u2 in [0, 1))
DBW_HD float normal01(uint32_t a, uint32_t b) {
    const float u1 = 1.f - uniform01(a), u2 = uniform01(b);
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

}  // namespace dbw
