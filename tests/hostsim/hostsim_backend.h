// hostsim_backend.h — TEST INFRASTRUCTURE (tests/hostsim): substitute for the HIP back end of bio_ik_amd/csrc/bioik_hip.hip.
// "Device memory" is host memory and a launch runs every workgroup as a gang of fibres (one per lane) of the calling thread that call the kernel body
// directly.  Injected with -DBIOIK_BACKEND_HEADER; never part of the product library.
#include <chrono>
#include <functional>
#include <ucontext.h>
namespace sim {
thread_local Block* blk = nullptr;
thread_local int tid = 0;
}  // namespace sim
#include <chrono>
unsigned long long sim_wall_clock() {
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
static unsigned long long be_device_clock_now() { return sim_wall_clock(); }
static bool be_can_time() { return false; }  // (the simulator's times say nothing about the device: the launcher's measured mapping choice stays off)
template <class F>
static double be_time_ms(void*, F&& enqueue) {
    enqueue();
    return 0.0;
}
static void be_zero_async(void* p, size_t bytes, void*) { std::memset(p, 0, bytes); }
static void be_fill_ff_async(void* p, size_t bytes, void*) { std::memset(p, 0xff, bytes); }
typedef void* stream_t;
static int be_device_count() { return 1; }
struct DeviceInfo {  // (the simulator stands for an MI355X)
    size_t lds_cu = 160 * 1024;
    int cus = 256, xcds = 8;
};
static DeviceInfo be_device_info(int) { return DeviceInfo{}; }
static void be_set_device(int) {}
static int be_get_device() { return 0; }
static void* be_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
static void be_free(void* p) { std::free(p); }
static void* be_alloc_pinned(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
static void be_free_pinned(void* p) { std::free(p); }
static void* be_alloc_async(size_t bytes, stream_t) { return std::malloc(bytes ? bytes : 1); }
static bool be_stream_capturing(stream_t) { return false; }
static void be_free_async(void* p, stream_t) { std::free(p); }
static void be_h2d(void* d, const void* h, size_t bytes, stream_t) { std::memcpy(d, h, bytes); }
static void be_d2h(void* h, const void* d, size_t bytes, stream_t) { std::memcpy(h, d, bytes); }
static void be_sync(stream_t) {}
static stream_t be_stream_create() { return nullptr; }
static void be_stream_destroy(stream_t) {}
// One workgroup at a time; its lanes are fibres of the calling thread, scheduled round-robin: a lane runs until it waits at a rendezvous
// (or ends), then the next unfinished lane continues.  sim::tid is the running lane.
namespace sim {
struct Fibres {
    static constexpr size_t kStack = 2u << 20;  // per lane (the kernel bodies are one large inlined frame)
    ucontext_t main;
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    std::vector<std::unique_ptr<char[]>> stacks;  // kept across workgroups and launches
    int n = 0, alive = 0;
    std::function<void()> run;  // the kernel body of the running lane
};
static thread_local Fibres fib;
static int next_unfinished(int me) {
    int nx = me;
    do nx = nx + 1 == fib.n ? 0 : nx + 1;
    while (fib.done[nx] && nx != me);
    return nx;
}
static unsigned long long n_site_mismatches = 0;  // (relaxed: a diagnostic counter read by the tests between launches)
void site_mismatch(int first, int now) {
    // (the second rendezvous of a two-phase collective carries the negated line)
    if (__atomic_fetch_add(&n_site_mismatches, 1ull, __ATOMIC_RELAXED) < 8)
        std::fprintf(stderr, "[hostsim] divergent collective: lane %d of workgroup %d arrived from line %d at a rendezvous opened from line %d\n", tid, blk->block_id, now, first);
    if (std::getenv("BIOIK_HOSTSIM_SITES_FATAL")) std::abort();
}
void yield() {
    const int me = tid, nx = next_unfinished(me);
    if (nx == me) return;
    tid = nx;
    swapcontext(&fib.ctx[me], &fib.ctx[nx]);  // (whoever resumes this lane has set tid back to it)
}
static void lane_main() {
    fib.run();
    const int me = tid;
    fib.done[me] = 1;
    if (--fib.alive == 0) setcontext(&fib.main);
    tid = next_unfinished(me);
    setcontext(&fib.ctx[tid]);
}
}  // namespace sim
template <class Body>
static void be_launch(uint64_t grid, int block, size_t lds_bytes, stream_t, Body body) {
    std::vector<double> lds(lds_bytes / 8 + 2);
    sim::Fibres& f = sim::fib;
    while ((int)f.stacks.size() < block) f.stacks.emplace_back(new char[sim::Fibres::kStack]);
    for (uint64_t b = 0; b < grid; b++) {
        sim::Block blk;
        blk.nthreads = block;
        blk.block_id = (int)b;
        blk.bar.n = block;
        blk.bar.rounds.assign((size_t)block, 0ull);
        {
            sim::Rendezvous of_a_wave;
            of_a_wave.n = 64;
            of_a_wave.rounds.assign((size_t)block, 0ull);
            blk.wave_bar.assign((size_t)(block / 64), of_a_wave);
        }
        blk.xchg.assign((size_t)block, 0);
        f.n = f.alive = block;
        f.ctx.assign((size_t)block, ucontext_t());
        f.done.assign((size_t)block, 0);
        f.run = [&]() { body(b, lds.data()); };
        for (int t = 0; t < block; t++) {
            getcontext(&f.ctx[t]);
            f.ctx[t].uc_stack.ss_sp = f.stacks[t].get();
            f.ctx[t].uc_stack.ss_size = sim::Fibres::kStack;
            f.ctx[t].uc_link = nullptr;
            makecontext(&f.ctx[t], sim::lane_main, 0);
        }
        sim::blk = &blk;
        sim::tid = 0;
        swapcontext(&f.main, &f.ctx[0]);  // returns when the last lane has ended
        sim::blk = nullptr;
    }
}
#define LAUNCH(KERNEL, BODYCALL, grid, block, lds, stream, args) be_launch(grid, block, lds, stream, [&](uint64_t b_, double* l_) { BODYCALL; })
static void be_allow_lds(size_t) {}
// What the tests read of the simulator itself (tests/test_hostsim_parity.py): how often lanes met at DIFFERENT collectives so far -- on the device that
// is a silent exchange of garbage --, and a launch that does it on purpose (odd lanes synchronise from another line than even ones)
extern "C" unsigned long long hostsim_divergent_collectives() { return __atomic_load_n(&sim::n_site_mismatches, __ATOMIC_RELAXED); }
extern "C" void hostsim_selftest_divergence(int diverge) {
    be_launch(1, 64, 64, nullptr, [&](uint64_t, double* l) {
        const int lane = p_tid();
        l[0] = 0.0;
        p_wave_sync();
        if (diverge && (lane & 1)) {
            p_wave_sync();
        } else {
            p_wave_sync();
        }
        const int sum = p_read_lane(lane, 63) + p_shfl_xor(lane, 1);  // (collectives from one line each: no report)
        if (lane == 0) l[0] = (double)sum;
    });
}
