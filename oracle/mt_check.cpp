// mt_check.cpp -- independent pin for the oracle's mt19937 / rand_perm / rand_float: the same
// quantities computed with libstdc++'s std::mt19937, written the way faiss utils/random.cpp
// writes them (RandomGenerator: mt() % max, mt() / float(mt.max()); rand_perm: Fisher-Yates).
// TEST INFRASTRUCTURE ONLY (built by oracle/Makefile, called from tests/test_oracle_rng.py).
#include <cstdint>
#include <random>
#include <utility>

extern "C" {

void mtc_raw(uint32_t seed, int64_t n, uint32_t* out) {
    std::mt19937 mt((unsigned int)seed);
    for (int64_t i = 0; i < n; i++) out[i] = (uint32_t)mt();
}

void mtc_rand_perm(int* perm, size_t n, int64_t seed) {
    std::mt19937 mt((unsigned int)seed);
    for (size_t i = 0; i < n; i++) perm[i] = (int)i;
    for (size_t i = 0; i + 1 < n; i++) {
        int max = (int)(n - i);
        int i2 = (int)(i + mt() % max);
        std::swap(perm[i], perm[i2]);
    }
}

void mtc_rand_floats(uint32_t seed, int64_t n, float* out) {
    std::mt19937 mt((unsigned int)seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt() / float(mt.max());
}
}
