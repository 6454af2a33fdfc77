// ABI bookkeeping + host-side MT19937 stream used to seed match_als' factor matrix.
#include "mvmc_common.h"

extern "C" int mvmc_abi_version(void) { return MVMC_ABI_VERSION; }

extern "C" const char* mvmc_status_string(int status) {
    switch (status) {
        case MVMC_OK: return "ok";
        case MVMC_ERR_ARG: return "invalid argument";
        case MVMC_ERR_LAUNCH: return "kernel launch failed";
        case MVMC_ERR_UNSUPPORTED: return "size not covered by the compiled kernel variants";
        default: return "unknown status";
    }
}

// numpy.random.RandomState(0).rand(): MT19937 seeded with init_genrand(0); each double is
// (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53  (mv_association.py:271 draws rand(n, r) row-major).
extern "C" int mvmc_als_seed_table(double* out_host, int count) {
    if (!out_host || count < 0) return MVMC_ERR_ARG;
    uint32_t mt[624];
    mt[0] = 0u;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int pos = 624;
    auto next = [&]() -> uint32_t {
        if (pos >= 624) {
            for (int k = 0; k < 624; ++k) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    };
    for (int i = 0; i < count; ++i) {
        uint32_t a = next() >> 5, b = next() >> 6;
        out_host[i] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
    return MVMC_OK;
}
