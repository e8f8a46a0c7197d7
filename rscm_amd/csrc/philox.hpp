// Philox4x32-10 (Salmon et al. 2011), counter-based: shared by the Latin-hypercube kernel and the
// device stretch-move sampler.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rscm {

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// 53-bit uniform in [0,1) from two 32-bit words, like rand's Standard f64 (rng.gen::<f64>())
__device__ __forceinline__ double u01_from_bits(uint32_t lo, uint32_t hi)
{
    const uint64_t bits53 = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)bits53 * (1.0 / 9007199254740992.0);
}

}  // namespace rscm
