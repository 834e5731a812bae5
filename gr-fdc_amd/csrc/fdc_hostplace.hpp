// Where the host side of a member device runs (round 6).  On an 8-GPU node the GPUs hang off different sockets /
// NUMA nodes; a worker thread that feeds device d over PCIe, and the pinned staging it copies through, belong on d's node — left alone
// they land wherever the calling thread lives and seven of eight members cross the socket interconnect with every byte.  Linux sysfs
// only, no libnuma: /sys/bus/pci/devices/<bdf>/numa_node and /sys/devices/system/node/node<k>/cpulist.  Everything here is best
// effort: a missing file, node -1 (single-node machines, most containers) or an empty intersection with the process's own cpuset
// leaves the thread where it is.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace fdc {

// "0000:C1:00.0" (hipDeviceGetPCIBusId) -> its NUMA node, -1 if unknown
inline int pci_numa_node(const char *bdf)
{
    if (!bdf || !*bdf) return -1;
    std::string path = "/sys/bus/pci/devices/";
    for (const char *q = bdf; *q; q++) path += (char)std::tolower((unsigned char)*q);
    path += "/numa_node";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

// the CPUs of a node ("0-15,128-143") as a cpu_set_t; false if the node has no list
inline bool node_cpuset(int node, cpu_set_t *set)
{
    CPU_ZERO(set);
    if (node < 0) return false;
    char path[96];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    bool any = false;
    for (char *q = buf; *q;) {
        while (*q && !std::isdigit((unsigned char)*q)) q++;
        if (!*q) break;
        long a = std::strtol(q, &q, 10), b = a;
        if (*q == '-') b = std::strtol(q + 1, &q, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); any = true; }
    }
    return any;
}

// Restrict the CALLING thread to the CPUs of `node` that the process may use.  Returns 1 = pinned, 0 = left alone (unknown node, no list,
// nothing of the node in the process's mask), -1 = the system call failed.
inline int pin_this_thread_to_node(int node)
{
    cpu_set_t want, have, both;
    if (!node_cpuset(node, &want)) return 0;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return -1;
    CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) == 0) return 0;
    return pthread_setaffinity_np(pthread_self(), sizeof both, &both) == 0 ? 1 : -1;
}

}  // namespace fdc
