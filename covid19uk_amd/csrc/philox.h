// Philox4x32-10 counter-based RNG (Salmon et al. 2011), host + device.
// The reference seeds nothing (seed=None throughout inference.py), so its
// random stream is not reproducible; this build defines its own stream:
//   key     = (seed_lo, seed_hi)
//   counter = (draw index, stream id, sweep, global chain id)
// and the CPU oracle (oracle/mcmc_oracle.py) uses the identical function, so
// proposals are bit-comparable between the two.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SEIR_HD __host__ __device__
#else
#define SEIR_HD
#endif

namespace seir {

struct U4 { uint32_t x, y, z, w; };

SEIR_HD inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

// 52-bit uniform in (0,1): ((hi:lo) >> 12 + 0.5) * 2^-52, exactly representable.
SEIR_HD inline double u01(uint32_t hi, uint32_t lo) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 12;
    return ((double)x + 0.5) * 2.220446049250313e-16;
}

// stream ids
enum : uint32_t { RS_MOMENTUM = 0, RS_HMC_ACCEPT = 1, RS_MOVE_BASE = 16 };

struct RngKey { uint32_t k0, k1, chain, sweep; };

// two uniforms from draw slot `idx` of stream `stream`
SEIR_HD inline void rng_uniform2(const RngKey &k, uint32_t stream, uint32_t idx, double &a, double &b) {
    const U4 r = philox4x32_10(U4{idx, stream, k.sweep, k.chain}, k.k0, k.k1);
    a = u01(r.x, r.y);
    b = u01(r.z, r.w);
}

// uniform integer on {0..n-1}
SEIR_HD inline int rng_index(double u, int n) {
    int i = (int)(u * (double)n);
    return i < n - 1 ? i : n - 1;
}

}  // namespace seir
