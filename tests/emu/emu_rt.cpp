// TEST-ONLY host emulation of the handful of HIP runtime features the kernels use
// (see zero-chain_amd/csrc/gpu_rt.h).  Blocks run one after another; kernels that call
// __syncthreads() get one OS thread per GPU thread and a real barrier, the rest are run as a
// plain loop over thread indices.
#define ZK_EMU 1
#include "../../zero-chain_amd/csrc/gpu_rt.h"

#ifdef ZK_EMU_NO_FIBERS
// The first execution model, kept for the sanitizer build (tests/emu/build_emu.py --sanitize: AddressSanitizer does not
// follow ucontext switches): blocks one after another, one OS thread per GPU thread for kernels with barriers.
thread_local emu_dim3 threadIdx, blockIdx;
emu_dim3 blockDim, gridDim;
unsigned char* emu_dyn_shared = nullptr;

namespace {
std::mutex g_mu;
std::condition_variable g_cv;
unsigned g_count = 0, g_gen = 0, g_parties = 1;
}  // namespace

void emu_barrier_wait() {
    std::unique_lock<std::mutex> lk(g_mu);
    unsigned gen = g_gen;
    if (++g_count == g_parties) {
        g_count = 0;
        g_gen++;
        g_cv.notify_all();
    } else {
        g_cv.wait(lk, [&] { return gen != g_gen; });
    }
}

void emu_launch(emu_dim3 grid, emu_dim3 block, size_t shmem, bool needs_sync, const std::function<void()>& body) {
    // one launch at a time: blockDim / gridDim, the dynamic shared area and the barrier state are process-wide (two host
    // threads driving two handles - tests/abi_multi.c - launched into each other's grid: found by this very build, r5)
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    blockDim = block;
    gridDim = grid;
    std::vector<unsigned char> dyn(shmem + 64);
    emu_dyn_shared = dyn.data();
    const unsigned nthreads = block.x * block.y * block.z;
    if (!needs_sync) {
        for (unsigned bz = 0; bz < grid.z; bz++)
            for (unsigned by = 0; by < grid.y; by++)
                for (unsigned bx = 0; bx < grid.x; bx++) {
                    blockIdx = emu_dim3(bx, by, bz);
                    for (unsigned t = 0; t < nthreads; t++) {
                        threadIdx = emu_dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        body();
                    }
                }
    } else {
        // one OS thread per GPU thread for the whole launch; the blocks are walked in lock step
        // (a barrier between consecutive blocks keeps static __shared__ storage private to a block)
        g_parties = nthreads;
        g_count = 0;
        std::vector<std::thread> ths;
        ths.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; t++)
            ths.emplace_back([&, t] {
                threadIdx = emu_dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                for (unsigned bz = 0; bz < grid.z; bz++)
                    for (unsigned by = 0; by < grid.y; by++)
                        for (unsigned bx = 0; bx < grid.x; bx++) {
                            blockIdx = emu_dim3(bx, by, bz);
                            body();
                            emu_barrier_wait();
                        }
            });
        for (auto& th : ths) th.join();
    }
    emu_dyn_shared = nullptr;
}

#else
thread_local emu_dim3 threadIdx, blockIdx;
emu_dim3 blockDim, gridDim;
thread_local unsigned char* emu_dyn_shared = nullptr;

// Execution model (round 4; the first version created one OS thread per GPU thread per launch and ran barrier-free
// kernels as one serial loop: 10 minutes of mostly thread creation and futex traffic for the CPU suite):
//   * the BLOCKS of a launch are handed out to a persistent pool of worker threads (one per core), so independent
//     blocks run in parallel; static __shared__ arrays are thread_local (gpu_rt.h), i.e. private to the block a worker runs;
//   * inside a block, a kernel without barriers is a plain loop over its threads; a kernel with __syncthreads() runs its
//     threads as FIBERS (ucontext) of the worker, round robin: a barrier yields to the next fiber, and a full round brings
//     every live fiber to the same barrier (threads that returned early simply drop out, as on the hardware).
#include <ucontext.h>
#include <sys/mman.h>
#include <atomic>

namespace {

constexpr size_t FIBER_STACK = 256 * 1024;
constexpr unsigned MAX_FIBERS = 1024;

struct Fibers {
    ucontext_t sched;
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    char* stacks = nullptr;
    unsigned cur = 0;
    bool active = false;
    const std::function<void()>* body = nullptr;
    emu_dim3 block;
    ~Fibers() {
        if (stacks) munmap(stacks, FIBER_STACK * MAX_FIBERS);
    }
};
thread_local Fibers* t_fib = nullptr;

void fiber_entry() {
    Fibers* f = t_fib;
    (*f->body)();
    f->done[f->cur] = 1;
    swapcontext(&f->ctx[f->cur], &f->sched);
}

void run_block_fibers(Fibers& f, unsigned nthreads, const emu_dim3& block, const std::function<void()>& body) {
    if (!f.stacks) {
        f.stacks = (char*)mmap(nullptr, FIBER_STACK * MAX_FIBERS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stacks == MAP_FAILED) abort();
        f.ctx.resize(MAX_FIBERS);
        f.done.resize(MAX_FIBERS);
    }
    if (nthreads > MAX_FIBERS) abort();
    f.body = &body;
    f.block = block;
    f.active = true;
    for (unsigned t = 0; t < nthreads; t++) {
        getcontext(&f.ctx[t]);
        f.ctx[t].uc_stack.ss_sp = f.stacks + (size_t)t * FIBER_STACK;
        f.ctx[t].uc_stack.ss_size = FIBER_STACK;
        f.ctx[t].uc_link = nullptr;
        makecontext(&f.ctx[t], fiber_entry, 0);
        f.done[t] = 0;
    }
    unsigned live = nthreads;
    while (live) {
        live = 0;
        for (unsigned t = 0; t < nthreads; t++) {
            if (f.done[t]) continue;
            f.cur = t;
            threadIdx = emu_dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&f.sched, &f.ctx[t]);
            if (!f.done[t]) live++;
        }
    }
    f.active = false;
}

// ---- the pool
struct Launch {
    emu_dim3 grid, block;
    size_t shmem = 0;
    bool needs_sync = false;
    const std::function<void()>* body = nullptr;
    std::atomic<unsigned> next{0};
    unsigned total = 0;
};
std::mutex g_launch_mu;          // one launch at a time (blockDim / gridDim are process-wide)
std::mutex g_mu;
std::condition_variable g_cv_work, g_cv_done;
Launch* g_launch = nullptr;
unsigned g_generation = 0, g_running = 0;
bool g_stop = false;
std::vector<std::thread> g_workers;

void run_blocks(Launch& L) {
    static thread_local Fibers fibers;
    static thread_local std::vector<unsigned char> dyn;
    t_fib = &fibers;
    if (dyn.size() < L.shmem + 64) dyn.resize(L.shmem + 64);
    emu_dyn_shared = dyn.data();
    const unsigned nthreads = L.block.x * L.block.y * L.block.z;
    for (;;) {
        const unsigned b = L.next.fetch_add(1, std::memory_order_relaxed);
        if (b >= L.total) break;
        blockIdx = emu_dim3(b % L.grid.x, (b / L.grid.x) % L.grid.y, b / (L.grid.x * L.grid.y));
        if (!L.needs_sync) {
            for (unsigned t = 0; t < nthreads; t++) {
                threadIdx = emu_dim3(t % L.block.x, (t / L.block.x) % L.block.y, t / (L.block.x * L.block.y));
                (*L.body)();
            }
        } else {
            run_block_fibers(fibers, nthreads, L.block, *L.body);
        }
    }
}

void worker_main() {
    unsigned seen = 0;
    for (;;) {
        Launch* L;
        {
            std::unique_lock<std::mutex> lk(g_mu);
            g_cv_work.wait(lk, [&] { return g_stop || g_generation != seen; });
            if (g_stop) return;
            seen = g_generation;
            L = g_launch;
        }
        run_blocks(*L);
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (--g_running == 0) g_cv_done.notify_all();
        }
    }
}

struct PoolGuard {
    ~PoolGuard() {
        {
            std::lock_guard<std::mutex> lk(g_mu);
            g_stop = true;
        }
        g_cv_work.notify_all();
        for (auto& t : g_workers) t.join();
    }
} g_pool_guard;

}  // namespace

void emu_barrier_wait() {
    Fibers* f = t_fib;
    if (!f || !f->active) return;   // a barrier in a kernel launched without ZK_LAUNCH_SYNC would be a bug of the sources
    swapcontext(&f->ctx[f->cur], &f->sched);
    // (the scheduler restored threadIdx before it resumed us)
}

void emu_launch(emu_dim3 grid, emu_dim3 block, size_t shmem, bool needs_sync, const std::function<void()>& body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);
    blockDim = block;
    gridDim = grid;
    Launch L;
    L.grid = grid;
    L.block = block;
    L.shmem = shmem;
    L.needs_sync = needs_sync;
    L.body = &body;
    L.total = grid.x * grid.y * grid.z;
    if (!L.total) return;
    unsigned want = std::thread::hardware_concurrency();
    if (const char* env = getenv("ZKAMD_EMU_THREADS")) want = (unsigned)atoi(env);
    if (want < 1) want = 1;
    if (want > 32) want = 32;
    const unsigned helpers = L.total > 1 ? std::min(want, L.total) - 1 : 0;   // the calling thread works too
    {
        std::lock_guard<std::mutex> lk(g_mu);
        while (g_workers.size() < helpers) g_workers.emplace_back(worker_main);
        g_launch = &L;
        g_running = (unsigned)g_workers.size();
        g_generation++;
    }
    g_cv_work.notify_all();
    run_blocks(L);
    {
        std::unique_lock<std::mutex> lk(g_mu);
        g_cv_done.wait(lk, [&] { return g_running == 0; });
        g_launch = nullptr;
    }
    emu_dyn_shared = nullptr;
}

#endif

// TEST-ONLY diagnostics: ZKAMD_EMU_BACKTRACE=1 prints the native stack of a crash inside the emulation build
// (llvm-symbolizer -e tests/emu/libzkamd_emu.so <offset> names the frames).
#include <cstring>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void emu_crash_handler(int sig) {
    void* bt[64];
    const int n = backtrace(bt, 64);
    const char msg[] = "[emu] fatal signal, native stack:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
struct EmuCrashInstall {
    EmuCrashInstall() {
        if (getenv("ZKAMD_EMU_BACKTRACE")) {
            static char alt[1 << 16];          // a stack overflow must still reach the handler (main thread only)
            stack_t ss;
            ss.ss_sp = alt;
            ss.ss_size = sizeof(alt);
            ss.ss_flags = 0;
            sigaltstack(&ss, nullptr);
            struct sigaction sa;
            memset(&sa, 0, sizeof(sa));
            sa.sa_handler = emu_crash_handler;
            sa.sa_flags = SA_ONSTACK;
            sigaction(SIGSEGV, &sa, nullptr);
            sigaction(SIGBUS, &sa, nullptr);
            sigaction(SIGABRT, &sa, nullptr);
        }
    }
} g_emu_crash_install;
}  // namespace
