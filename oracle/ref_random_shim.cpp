// Thin extern "C" shim over the REFERENCE's own random source -- /root/reference/thirdparty/lambdatwist/utils/random.h, compiled from where it lies by
// `make -C oracle ref` into oracle/_ref/librandom_ref.so (header-only, no Ceres needed).  Test infrastructure only: it pins oracle/pnp_oracle.c's
// restatement of the RANSAC sampler's draw sequence (std::default_random_engine + std::uniform_int_distribution as PNP::compute consumes them).
// get4RandomInRange0 itself lives in pnp_ransac.cpp, which does not compile here (ceres/problem.h); its eleven lines are restated below over the
// reference's own mlib::randui, statement for statement (pnp_ransac.cpp:161-183).
#include <set>
#include <utils/random.h>

extern "C" {

// back to the state of a fresh process (the generator is seeded once, with RANDOM_SEED_VALUE = 0: random.h:40-42,77-86)
void ref_rng_reset() { mlib::random::seeded = false; }

// n draws of mlib::randui<int>(0, max - 1)
void ref_randui(int max, int n, int* out) {
    for (int i = 0; i < n; ++i) out[i] = mlib::randui<int>(0, max - 1);
}

// n samples of get4RandomInRange0(max): 4 distinct indices, ascending (std::set order); out[n][4]
void ref_get4(unsigned max, int n, int* out) {
    for (int s = 0; s < n; ++s) {
        std::set<unsigned> set;
        while (set.size() < 4) set.insert(mlib::randui<int>(0, max - 1));
        int k = 0;
        for (unsigned i : set) out[4 * s + k++] = (int)i;
    }
}
}
