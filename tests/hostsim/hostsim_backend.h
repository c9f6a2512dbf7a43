// hostsim_backend.h — TEST INFRASTRUCTURE (tests/hostsim): substitute for the HIP back end of bio_ik_amd/csrc/bioik_hip.hip.
// "Device memory" is host memory and a launch runs every workgroup as a gang of OS threads (one per lane) that call the kernel body
// directly.  Injected with -DBIOIK_BACKEND_HEADER; never part of the product library.
#include <barrier>
#include <chrono>
#include <thread>
namespace sim {
thread_local Block* blk = nullptr;
thread_local int tid = 0;
}  // namespace sim
#include <chrono>
unsigned long long sim_wall_clock() {
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
static void be_zero_async(void* p, size_t bytes, void*) { std::memset(p, 0, bytes); }
typedef void* stream_t;
static int be_device_count() { return 1; }
static void be_set_device(int) {}
static int be_get_device() { return 0; }
static void* be_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
static void be_free(void* p) { std::free(p); }
static void* be_alloc_pinned(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
static void be_free_pinned(void* p) { std::free(p); }
static void* be_alloc_async(size_t bytes, stream_t) { return std::malloc(bytes ? bytes : 1); }
static void be_free_async(void* p, stream_t) { std::free(p); }
static void be_h2d(void* d, const void* h, size_t bytes, stream_t) { std::memcpy(d, h, bytes); }
static void be_d2h(void* h, const void* d, size_t bytes, stream_t) { std::memcpy(h, d, bytes); }
static void be_sync(stream_t) {}
static stream_t be_stream_create() { return nullptr; }
static void be_stream_destroy(stream_t) {}
template <class Body>
static void be_launch(uint64_t grid, int block, size_t lds_bytes, stream_t, Body body) {
    std::vector<double> lds(lds_bytes / 8 + 2);
    for (uint64_t b = 0; b < grid; b++) {
        sim::Block blk;
        blk.nthreads = block;
        blk.block_id = (int)b;
        blk.bar.reset(new std::barrier<>(block));
        for (int w = 0; w < block / 64; w++) blk.wave_bar.emplace_back(new std::barrier<>(64));
        blk.xchg.assign((size_t)block, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < block; t++)
            th.emplace_back([&, t]() {
                sim::blk = &blk;
                sim::tid = t;
                body(b, lds.data());
            });
        for (auto& t : th) t.join();
    }
}
#define LAUNCH(KERNEL, BODYCALL, grid, block, lds, stream, args) be_launch(grid, block, lds, stream, [&](uint64_t b_, double* l_) { BODYCALL; })
static void be_allow_lds(size_t) {}
