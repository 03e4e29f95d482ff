/* sh_numa.h -- host memory next to the GPU that reads it.  Plain C++ + the HIP runtime API (no device code).
 *
 * A node of eight MI355X has two sockets; a GPU's PCIe root hangs off one of them.  Pinned staging (the loader threads write it, the GPU's DMA
 * reads it) and the pinned result buffers (k_results_out writes them) should live in that socket's memory: hipHostMalloc places pages on
 * the node of the thread that calls it (the kernel's default local policy), so the allocations of an engine / preparer are made inside a
 * ShNumaScope, which binds the calling thread to the CPUs of the device's node for the duration of the allocation and restores its
 * affinity afterwards -- nothing else about the process changes.  The node comes from sysfs (/sys/bus/pci/devices/<bus id>/numa_node, the same
 * number /sys/class/drm/card<N>/device/numa_node shows); -1 / unreadable (single-socket hosts, containers without sysfs): no-op.
 * SCRAPPIE_HIP_NUMA=0 turns it off.  Unmeasured on a two-socket node (none was available to this build): it costs two sched_setaffinity
 * calls per allocation of a grow-only buffer. */
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static inline int sh_device_numa_node(int device) {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess || !bus[0]) return -1;
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');      /* sysfs spells the bus id in lower case */
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

/* the CPUs of a node ("0-15,32-47") intersected with `allowed`; false if none */
static inline bool sh_node_cpus(int node, const cpu_set_t *allowed, cpu_set_t *out) {
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(out);
    int any = 0;
    for (char *p = buf; *p && *p != '\n';) {
        char *end;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET((int)c, allowed)) { CPU_SET((int)c, out); any = 1; }
        p = (*end == ',') ? end + 1 : end;
    }
    return any != 0;
}

struct ShNumaScope {
    cpu_set_t old;
    bool bound = false;
    explicit ShNumaScope(int device) {
        static const bool on = [] { const char *v = getenv("SCRAPPIE_HIP_NUMA"); return !(v && v[0] == '0'); }();
        if (!on) return;
        const int node = sh_device_numa_node(device);
        if (node < 0) return;
        cpu_set_t want;
        if (sched_getaffinity(0, sizeof old, &old) != 0 || !sh_node_cpus(node, &old, &want)) return;
        if (CPU_EQUAL(&want, &old)) return;
        bound = sched_setaffinity(0, sizeof want, &want) == 0;
    }
    ~ShNumaScope() { if (bound) (void)sched_setaffinity(0, sizeof old, &old); }
    ShNumaScope(const ShNumaScope &) = delete;
    ShNumaScope &operator=(const ShNumaScope &) = delete;
};
