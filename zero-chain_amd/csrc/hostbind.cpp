// One process per GPU: keep a rank's host threads on the NUMA node its GPU hangs off.
//
// The host side of a chunk is small (enqueue ~100 launches, encode 1024 proofs, the page-locked staging of 272 B
// per statement) but it is latency that the GPU waits for, and on a two-socket 8-GPU node a rank whose threads sit on
// the far socket crosses the inter-socket link for every doorbell and every staging copy.  The reference has no
// counterpart (its prover is one CPU process: core/proofs/src/confidential.rs:149); this belongs to the multi-GPU row
// of the hot path.
#include "host_common.h"

#include <cstdio>
#include <cstring>
#include <string>
#if defined(__linux__) && !defined(ZK_EMU)
#include <sched.h>
#endif

using namespace zkrt;

namespace {

// "0-15,64-79" -> the CPUs it names that are also in `allowed`
#if defined(__linux__) && !defined(ZK_EMU)
int parse_cpulist(const char* s, const cpu_set_t& allowed, cpu_set_t* out) {
    CPU_ZERO(out);
    int n = 0;
    while (*s) {
        char* end = nullptr;
        long a = strtol(s, &end, 10);
        if (end == s) break;
        long b = a;
        s = end;
        if (*s == '-') {
            b = strtol(s + 1, &end, 10);
            s = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed) && !CPU_ISSET((int)c, out)) {
                CPU_SET((int)c, out);
                n++;
            }
        while (*s == ',' || *s == ' ' || *s == '\n') s++;
    }
    return n;
}
bool read_line(const std::string& path, char* buf, size_t cap) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    return ok;
}
#endif

}  // namespace

extern "C" zk_status zk_bind_host_to_device(int device, int* numa_node_out, int* n_cpus_out) try {
    if (numa_node_out) *numa_node_out = -1;
    if (n_cpus_out) *n_cpus_out = 0;
#if defined(__linux__) && !defined(ZK_EMU)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ZK_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(ZK_ERR_INVALID_ARGUMENT, "device index out of range");
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return ZK_OK;   // nothing known: leave the thread where it is
    if (n_cpus_out) *n_cpus_out = CPU_COUNT(&allowed);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return ZK_OK;
    for (char* c = bus; *c; c++)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');   // sysfs names are lower case
    char line[4096];
    if (!read_line(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", line, sizeof(line))) return ZK_OK;
    const int node = atoi(line);
    if (node < 0) return ZK_OK;   // the platform reports no affinity (single-node box, or a VM that hides it)
    if (!read_line("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", line, sizeof(line))) return ZK_OK;
    cpu_set_t mine;
    const int k = parse_cpulist(line, allowed, &mine);
    if (k <= 0) return ZK_OK;     // the node's CPUs are not ours to use (cgroup cpuset): keep the mask we were given
    if (sched_setaffinity(0, sizeof(mine), &mine) != 0) return ZK_OK;
    if (numa_node_out) *numa_node_out = node;
    if (n_cpus_out) *n_cpus_out = k;
#else
    (void)device;
#endif
    return ZK_OK;
} ZK_ABI_CATCH
