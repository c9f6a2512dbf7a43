// orc_rng.h — ORACLE (test infrastructure): the two random-number back-ends of the solver.
//
// (a) ReferenceRandom restates reference src/ik_base.h:49-126 (std::minstd_rand, 8 Mi-entry uniform and
//     Gaussian tables, XORShift64 index generator of src/utils.h:369-385).
// (b) CounterRandom is the counter-based generator the device uses: Philox2x32-10
//     (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 v1.14 constants) for the control
//     draws (query key, pre-selection count, memetic sign, wipe-out), a two-multiply avalanche hash of the counter for the
//     bulk draws of reproduction (one 32-bit word per gene), integer-only post-processing so CPU and GPU produce
//     bit-identical doubles.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------------------------
// Philox (Random123: philox.h; M2x32 = 0xD256D193, M4x32 = {0xD2511F53, 0xCD9E8D57},
// W32 = {0x9E3779B9, 0xBB67AE85})
// ---------------------------------------------------------------------------------------------
inline void philox2x32_10(uint32_t key, uint32_t c0, uint32_t c1, uint32_t* out) {
    for (int r = 0; r < 10; r++) {
        if (r > 0) key += 0x9E3779B9u;
        uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
        uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
        c0 = hi ^ key ^ c1;
        c1 = lo;
    }
    out[0] = c0;
    out[1] = c1;
}

inline void philox4x32_10(const uint32_t* key2, const uint32_t* ctr4, uint32_t* out4) {
    uint32_t k0 = key2[0], k1 = key2[1];
    uint32_t c0 = ctr4[0], c1 = ctr4[1], c2 = ctr4[2], c3 = ctr4[3];
    for (int r = 0; r < 10; r++) {
        if (r > 0) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
    out4[0] = c0;
    out4[1] = c1;
    out4[2] = c2;
    out4[3] = c3;
}

// counter layout and integer->double post-processing
enum { PURPOSE_REPRODUCE = 0, PURPOSE_PRESELECT = 1, PURPOSE_MEMETIC_SIGN = 2, PURPOSE_WIPEOUT = 3, PURPOSE_WIPEOUT_GENE = 4, PURPOSE_POINT_RANDOM = 5 };
// random words of child c in one generation (the bulk of all draws: 1 + D words per child; bioik_device.h states the same, round 6's form):
//     stream        = mix32(key ^ ctr1 * 0x85EBCA77)
//     base of child = mix32((c << 8) * 0x9E3779B1 + stream)        = the child's word 0; its top four bits are the mutation-rate exponent
//     word w >= 1   = mix1(base + w * 0x9E3779B1)                   word 1 + g -> Gaussian of gene g
// mix32 = the 32-bit finaliser of MurmurHash3 (Appleby, public domain: xor-shift 16, * 0x85EBCA6B, xor-shift 13, * 0xC2B2AE35,
// xor-shift 16 -- every input bit flips every output bit with probability ~1/2); mix1 = its first half (one multiply).  Both are bijections, and
// the odd multiplier makes the value under the child's mix32 a bijection of c for a given stream (key, ctr1): no two children share a base, no
// two words of a child their input.
inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
inline uint32_t mix1(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h;
}
inline uint32_t child_stream(uint32_t key, uint32_t ctr1) { return mix32(key ^ (ctr1 * 0x85EBCA77u)); }
inline uint32_t child_word_of(uint32_t stream, uint32_t child, uint32_t w) {
    const uint32_t base = mix32((child << 8) * 0x9E3779B1u + stream);
    return w == 0 ? base : mix1(base + w * 0x9E3779B1u);
}

inline uint32_t ctr0_of(uint32_t child, uint32_t slot) { return (child << 8) | slot; }
inline uint32_t ctr1_of(uint32_t generation, uint32_t species, uint32_t purpose) { return (generation << 4) | (species << 3) | purpose; }

// approximately N(0,1) from ONE 32-bit word: the sum of its four bytes (Irwin-Hall, n = 4), centred and scaled to unit variance:
// mean 510, variance 4 (256^2 - 1) / 12 = 21845
inline double counter_gauss_from32(uint32_t x) {
    const int v = (int)((x & 255u) + ((x >> 8) & 255u) + ((x >> 16) & 255u) + (x >> 24)) - 510;
    return (double)v * 0.006765875086793228;  // 1 / sqrt(21845)
}
inline double counter_uniform_from(uint32_t x0, uint32_t x1) {
    uint64_t u = (((uint64_t)x0 << 32) | (uint64_t)x1) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);  // 2^-53
}
inline uint32_t query_key(uint64_t seed, uint64_t query, uint32_t island) {
    uint32_t o[2];
    philox2x32_10((uint32_t)seed, (uint32_t)query, (uint32_t)(seed >> 32) ^ (island * 0x9E3779B9u) ^ (uint32_t)(query >> 32), o);
    return o[0];
}

struct CounterRandom {
    uint32_t key = 0;
    uint32_t generation = 0;  // step*16 + generation inside the step
    uint32_t species = 0;     // persistent species id
    size_t mu = 2;

    void set_context(uint32_t step, uint32_t gen, uint32_t species_id) {
        generation = step * 16 + gen;
        species = species_id;
    }
    void reproduce_begin(size_t /*n_total*/, size_t /*gene_count*/) {}
    uint32_t child_word(size_t child_index, uint32_t w) {
        return child_word_of(child_stream(key, ctr1_of(generation, species, PURPOSE_REPRODUCE)), (uint32_t)child_index, w);
    }
    unsigned rate_exponent(size_t child_index) { return child_word(child_index, 0) >> 28; }
    double gauss(size_t child_index, size_t gene) { return counter_gauss_from32(child_word(child_index, (uint32_t)gene + 1u)); }
    void child_end(size_t /*gene_count*/) {}
    size_t preselect_count(size_t mu_, size_t lambda) {  // in [mu+1, mu+lambda-1]
        uint32_t o[2];
        philox2x32_10(key, ctr0_of(0, 0), ctr1_of(generation, species, PURPOSE_PRESELECT), o);
        return (size_t)(o[0] % (uint32_t)(lambda - 1)) + 1 + mu_;
    }
    bool memetic_negative() {
        uint32_t o[2];
        philox2x32_10(key, ctr0_of(0, 0), ctr1_of(generation, species, PURPOSE_MEMETIC_SIGN), o);
        return counter_uniform_from(o[0], o[1]) < 0.5;
    }
    double wipeout_u() {
        uint32_t o[2];
        philox2x32_10(key, ctr0_of(0, 0), ctr1_of(generation, species, PURPOSE_WIPEOUT), o);
        return counter_uniform_from(o[0], o[1]);
    }
    double wipeout_gene(size_t gene, double lo, double hi) {
        uint32_t o[2];
        philox2x32_10(key, ctr0_of(0, (uint32_t)gene), ctr1_of(generation, species, PURPOSE_WIPEOUT_GENE), o);
        return counter_uniform_from(o[0], o[1]) * (hi - lo) + lo;
    }
    // the gradient / Jacobian solvers' random configurations (ik_gradient.cpp:157-159, :165-170, :283-285): draw `count` of the island
    // (0: its starting point, s + 1: the reset in front of step s), one uniform number per gene
    double point_random(size_t gene, uint32_t count, double lo, double hi) {
        uint32_t o[2];
        philox2x32_10(key, ctr0_of(0, (uint32_t)gene), ctr1_of(count, 0, PURPOSE_POINT_RANDOM), o);
        return counter_uniform_from(o[0], o[1]) * (hi - lo) + lo;
    }
};

// ---------------------------------------------------------------------------------------------
// reference src/utils.h:369-385
// ---------------------------------------------------------------------------------------------
struct XORShift64 {
    uint64_t v = 88172645463325252ull;
    inline uint64_t operator()() {
        v ^= v << 13;
        v ^= v >> 7;
        v ^= v << 17;
        return v;
    }
};

// reference src/ik_base.h:49-126.  The 2 x 64 MiB tables are a pure function of the seed (they are filled
// from a freshly seeded minstd_rand in the constructor), so they are cached per seed.
struct ReferenceRandom {
    static constexpr size_t random_buffer_size = 1024 * 1024 * 8;
    struct Buffers {
        std::vector<double> uniform, gauss;
        std::minstd_rand rng_after;  // generator state after both fills
        std::normal_distribution<double> normal_after;
    };
    std::minstd_rand rng;
    std::normal_distribution<double> normal_distribution;
    XORShift64 _xorshift;
    std::shared_ptr<Buffers> buffers;
    const double* random_buffer = nullptr;
    size_t random_buffer_index = 0;
    const double* random_gauss_buffer = nullptr;
    size_t random_gauss_index = 0;
    const double* rr = nullptr;  // current slice inside reproduce()

    static std::shared_ptr<Buffers> get_buffers(uint32_t seed) {
        static std::mutex mtx;
        static std::map<uint32_t, std::shared_ptr<Buffers>> cache;
        std::lock_guard<std::mutex> lock(mtx);
        auto it = cache.find(seed);
        if (it != cache.end()) return it->second;
        auto b = std::make_shared<Buffers>();
        std::minstd_rand rng(seed);
        std::normal_distribution<double> nd;
        b->uniform.resize(random_buffer_size);
        for (auto& r : b->uniform) r = std::uniform_real_distribution<double>(0, 1)(rng);  // make_random_buffer
        b->gauss.resize(random_buffer_size);
        for (auto& r : b->gauss) r = nd(rng);  // make_random_gauss_buffer
        b->rng_after = rng;
        b->normal_after = nd;
        cache.clear();  // keep at most one seed resident (128 MiB)
        cache[seed] = b;
        return b;
    }

    explicit ReferenceRandom(uint32_t seed) {  // ik_base.h:118-125
        buffers = get_buffers(seed);
        rng = buffers->rng_after;
        normal_distribution = buffers->normal_after;
        random_buffer = buffers->uniform.data();
        random_buffer_index = _xorshift();
        random_gauss_buffer = buffers->gauss.data();
        random_gauss_index = _xorshift();
    }
    inline double random() { return std::uniform_real_distribution<double>(0, 1)(rng); }
    inline size_t random_index(size_t s) { return std::uniform_int_distribution<size_t>(0, s - 1)(rng); }
    inline double random(double min, double max) { return random() * (max - min) + min; }
    inline size_t fast_random_index(size_t mod) { return _xorshift() % mod; }
    inline double fast_random() {
        double r = random_buffer[random_buffer_index & (random_buffer_size - 1)];
        random_buffer_index++;
        return r;
    }
    inline const double* fast_random_gauss_n(size_t n) {
        size_t i = random_gauss_index;
        random_gauss_index += n;
        if (random_gauss_index >= random_buffer_size) i = 0, random_gauss_index = n;
        return random_gauss_buffer + i;
    }

    // ---- adapter interface used by Evolution2 (same call order as ik_evolution_2.cpp:254-301) ----
    void set_context(uint32_t, uint32_t, uint32_t) {}
    void reproduce_begin(size_t n_total, size_t gene_count) {
        size_t mu = 2;
        size_t s = (n_total - mu) * gene_count + n_total * 4 + 4;  // :254
        rr = fast_random_gauss_n(s);
        // :257 rounds a byte address up to a multiple of 4 — a no-op for 8-byte aligned doubles
    }
    unsigned rate_exponent(size_t) { return (unsigned)fast_random_index(16); }
    double gauss(size_t, size_t gene) { return rr[gene]; }
    void child_end(size_t gene_count) { rr += (gene_count + 3) / 4 * 4; }
    size_t preselect_count(size_t mu, size_t lambda) { return random_index(lambda - 1) + 1 + mu; }  // :369
    bool memetic_negative() { return fast_random() < 0.5; }                                             // :451
    double wipeout_u() { return fast_random(); }                                                        // :622
    double wipeout_gene(size_t, double lo, double hi) { return random(lo, hi); }                        // :629
};

}  // namespace orc
