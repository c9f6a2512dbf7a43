// hostsim_platform.h — TEST INFRASTRUCTURE (tests/hostsim): substitute for the execution-model primitives of
// bio_ik_amd/csrc/bioik_platform.h.  Every lane of a workgroup is a fibre (ucontext) of ONE OS thread and the cross-lane primitives
// rendezvous on barriers that pass control to the next lane (a lane runs until it waits: 64 switches per rendezvous instead of 64 futex
// wake-ups), so the kernel bodies of the product can be stepped against the CPU oracle on a machine without a GPU.  Injected with
// -DBIOIK_PLATFORM_HEADER; never part of the product library.
// ------------------------------------------------------------------------------------------------------------
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#define BIOIK_DEV inline
#define BIOIK_CALL __attribute__((noinline)) inline
typedef double lds_f64;
#define BIOIK_CONTRACT_OFF
typedef const DevProblem* ProbPtr;
typedef const DevProblemLean* LeanProbPtr;

namespace sim {
extern thread_local int tid;
struct Rendezvous {  // of `n` fibres of one OS thread
    int n = 0, arrived = 0;
    unsigned generation = 0;
    int site = 0;  // source line of the collective the first lane of this round arrived from
    unsigned long long round_of_first = 0;  // how many rounds of THIS rendezvous that lane has been to
    std::vector<unsigned long long> rounds;  // per lane of the workgroup
};
void site_mismatch(int first, int now);  // (hostsim_backend.h) lanes of one wavefront / workgroup met at DIFFERENT collectives: a divergent collective
void yield();  // to the next unfinished lane of the workgroup (hostsim_backend.h)
// `site`: the source line of the collective (every p_* collective passes its caller's line on, __builtin_LINE): on the device lanes that
// reach different collectives exchange garbage silently; here the round's first arrival names the site and every other arrival must agree
inline void arrive_and_wait(Rendezvous& r, int site) {
    const unsigned g = r.generation;
    const unsigned long long mine = ++r.rounds[tid];
    if (r.arrived == 0) r.site = site, r.round_of_first = mine;
    else if (r.site != site || r.round_of_first != mine) site_mismatch(r.site, site);
    if (++r.arrived == r.n) {
        r.arrived = 0, r.generation++;
        return;
    }
    while (r.generation == g) yield();
}
struct Block {
    int nthreads = 0;
    int block_id = 0;
    Rendezvous bar;
    std::vector<Rendezvous> wave_bar;
    std::vector<uint64_t> xchg;  // [waves][64]
    char* lds = nullptr;
};
extern thread_local Block* blk;
}  // namespace sim

BIOIK_DEV int p_tid() { return sim::tid; }
BIOIK_DEV int p_nthreads() { return sim::blk->nthreads; }
BIOIK_DEV void p_barrier(int site = __builtin_LINE()) { sim::arrive_and_wait(sim::blk->bar, site); }
BIOIK_DEV void p_wave_sync(int site = __builtin_LINE()) { sim::arrive_and_wait(sim::blk->wave_bar[sim::tid >> 6], site); }
template <class T>
BIOIK_DEV T p_shfl(T v, int src_lane, int site = __builtin_LINE()) {
    static_assert(sizeof(T) <= 8, "");
    int w = sim::tid >> 6, l = sim::tid & 63;
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    uint64_t* x = sim::blk->xchg.data() + (size_t)w * 64;
    x[l] = bits;
    sim::arrive_and_wait(sim::blk->wave_bar[w], site);
    uint64_t r = x[src_lane & 63];
    sim::arrive_and_wait(sim::blk->wave_bar[w], -site);
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T>
BIOIK_DEV T p_shfl_xor(T v, int mask, int site = __builtin_LINE()) { return p_shfl(v, (sim::tid & 63) ^ mask, site); }
template <int MASK, class T>
BIOIK_DEV T p_quad_xor(T v, int site = __builtin_LINE()) { return p_shfl_xor(v, MASK, site); }
template <int HALF, class T>  // lane i <-> 15 - i of its row of 16 (HALF = 0) or 7 - i of its half row (HALF = 1)
BIOIK_DEV T p_row_mirror(T v, int site = __builtin_LINE()) {
    const int l = sim::tid & 63;
    return p_shfl(v, HALF ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15))), site);
}
BIOIK_DEV int p_uniform(int v) { return v; }
template <class T>
BIOIK_DEV T p_read_lane(T v, int lane, int site = __builtin_LINE()) { return p_shfl(v, lane, site); }
BIOIK_DEV int p_fresh(int v) { return v; }
BIOIK_DEV double p_fresh(double v) { return v; }
BIOIK_DEV int p_lane_fresh() { return sim::tid & 63; }
BIOIK_DEV double p_clamp_uniform(double x, double lo, double hi) { return __builtin_fmin(__builtin_fmax(x, lo), hi); }
BIOIK_DEV double p_clamp(double x, double lo, double hi) { return __builtin_fmin(__builtin_fmax(x, lo), hi); }
BIOIK_DEV int p_tid_fresh() { return sim::tid; }
BIOIK_DEV int p_wave_index() { return sim::tid >> 6; }
BIOIK_DEV unsigned long long p_ballot(bool pred, int site = __builtin_LINE()) {  // every lane of the wavefront calls it (two rendezvous, as p_shfl)
    const int w = sim::tid >> 6, l = sim::tid & 63;
    uint64_t* x = sim::blk->xchg.data() + (size_t)w * 64;
    x[l] = pred ? 1u : 0u;
    sim::arrive_and_wait(sim::blk->wave_bar[w], site);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)x[i] << i;
    sim::arrive_and_wait(sim::blk->wave_bar[w], -site);
    return m;
}
BIOIK_DEV int p_popc(uint32_t v) { return __builtin_popcount(v); }
BIOIK_DEV int p_byte_sum(uint32_t v, int addend) { return (int)((v & 255u) + ((v >> 8) & 255u) + ((v >> 16) & 255u) + (v >> 24)) + addend; }
BIOIK_DEV int p_popc64(unsigned long long v) { return __builtin_popcountll(v); }
unsigned long long sim_wall_clock();  // 100 MHz ticks of a steady host clock (defined with the simulator's back end)
BIOIK_DEV unsigned long long p_wall_clock() { return sim_wall_clock(); }
BIOIK_DEV unsigned long long p_stamp_once(unsigned long long* word, unsigned long long value) {  // first caller's value wins; returns the winner
    unsigned long long expected = 0ull;
    return __atomic_compare_exchange_n(word, &expected, value, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) ? value : expected;
}
BIOIK_DEV unsigned int p_atomic_inc(unsigned int* counter) { return __atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL); }
BIOIK_DEV int p_xcc_id() { return 0; }
BIOIK_DEV void p_fence_device() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
BIOIK_DEV void p_prefetch_word_to_lds(const unsigned int* src, double* lds_slot) { *(unsigned int*)lds_slot = __atomic_load_n(src, __ATOMIC_RELAXED); }
BIOIK_DEV unsigned int p_prefetched_word(const double* lds_slot) { return *(const unsigned int*)lds_slot; }
BIOIK_DEV void p_atomic_add(unsigned int* counter, unsigned int v) { (void)__atomic_fetch_add(counter, v, __ATOMIC_ACQ_REL); }
BIOIK_DEV void p_atomic_sub(unsigned int* counter, unsigned int v) { (void)__atomic_fetch_sub(counter, v, __ATOMIC_ACQ_REL); }
BIOIK_DEV void p_atomic_min(unsigned int* word, unsigned int value) {
    unsigned int cur = __atomic_load_n(word, __ATOMIC_RELAXED);
    while (value < cur && !__atomic_compare_exchange_n(word, &cur, value, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {
    }
}
BIOIK_DEV unsigned int p_atomic_load(const unsigned int* word) { return __atomic_load_n(word, __ATOMIC_RELAXED); }
template <class T>
BIOIK_DEV T p_load_device(const T* p) { return *(const volatile T*)p; }
template <class T>
BIOIK_DEV void p_store_device(T* p, T v) { *(volatile T*)p = v; }
// (the lanes of a workgroup are fibres of one thread: a lane that waits for a word hands control on until another lane has written it)
BIOIK_DEV void p_flag_store(unsigned int* word, unsigned int value) { *(volatile unsigned int*)word = value; }
BIOIK_DEV unsigned int p_flag_wait_ge(const unsigned int* word, unsigned int value) {
    for (long spin = 0; spin < (1L << 20); spin++) {  // (a partner that is on its way arrives within a few thousand switches; the give-up itself is a test)
        const unsigned int r = *(const volatile unsigned int*)word;
        if (r >= value) return r;
        sim::yield();
    }
    return 0xffffffffu;
}
#define P_INF (__builtin_inf())

#define BIOIK_FP_STRICT
#define BIOIK_HD inline
