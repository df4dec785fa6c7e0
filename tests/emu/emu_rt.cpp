// TEST-ONLY host emulation of the handful of HIP runtime features the kernels use
// (see zero-chain_amd/csrc/gpu_rt.h).  Blocks run one after another; kernels that call
// __syncthreads() get one OS thread per GPU thread and a real barrier, the rest are run as a
// plain loop over thread indices.
#define ZK_EMU 1
#include "../../zero-chain_amd/csrc/gpu_rt.h"

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
