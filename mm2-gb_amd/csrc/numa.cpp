// numa.cpp -- where a device's host side should run: the CPUs of the NUMA node its PCIe root hangs off.
// The reference pins nothing and uses device 0 only (gpu/plmem.cu:426,462,499); on an 8-GPU node every rank / pool worker /
// batcher worker here moves its host threads next to its GPU before it allocates page-locked staging (first touch after the move
// puts the pages on that node), so that H2D / D2H copies and the host post-pass do not cross the socket link.
//   /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node      -> node (-1: unknown / single node)
//   /sys/devices/system/node/node<N>/cpulist                 -> "0-31,128-159"
// Both are read below a root directory that the test entry point (mm2gb_numa_cpus_for_bdf) can point at a made-up tree.  The move is an intersection with the CPUs
// the process may use already (affinity mask: a container's cpuset), and nothing happens when that is empty, when the node is
// unknown, or with MM2GB_NUMA=0.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/mm2gb_chain.h"
#include "host_chain.h"

namespace mm2gb {

static std::string sysfs_root(const char *root)
{
	return root && *root ? root : "";      // only the test entry point (mm2gb_numa_cpus_for_bdf) passes a root: production reads the real /sys
}

static bool read_line(const std::string &path, std::string &out)
{
	FILE *fp = fopen(path.c_str(), "r");
	if (!fp) return false;
	char buf[4096];
	const bool ok = fgets(buf, sizeof buf, fp) != nullptr;
	fclose(fp);
	if (!ok) return false;
	out = buf;
	while (!out.empty() && isspace((unsigned char)out.back())) out.pop_back();
	return true;
}

// "0-3,8,10-11" -> 0 1 2 3 8 10 11 (kernel cpulist format); false on anything else
bool parse_cpulist(const std::string &s, std::vector<int> &cpus)
{
	cpus.clear();
	size_t i = 0;
	while (i < s.size()) {
		if (!isdigit((unsigned char)s[i])) return false;
		long a = 0;
		while (i < s.size() && isdigit((unsigned char)s[i])) { a = a * 10 + (s[i++] - '0'); if (a > 1 << 20) return false; }
		long b = a;
		if (i < s.size() && s[i] == '-') {
			++i;
			if (i >= s.size() || !isdigit((unsigned char)s[i])) return false;
			b = 0;
			while (i < s.size() && isdigit((unsigned char)s[i])) { b = b * 10 + (s[i++] - '0'); if (b > 1 << 20) return false; }
			if (b < a) return false;
		}
		for (long c = a; c <= b; ++c) cpus.push_back((int)c);
		if (i < s.size()) { if (s[i] != ',') return false; ++i; if (i >= s.size()) return false; }
	}
	return true;
}

// hipDeviceGetPCIBusId gives "0000:c1:00.0" (upper or lower case hex); sysfs names are lower case
static std::string lower(std::string s) { for (char &c : s) c = (char)tolower((unsigned char)c); return s; }

int numa_node_of_bdf(const char *bdf, const char *root)
{
	if (!bdf || !*bdf) return -1;
	std::string line;
	if (!read_line(sysfs_root(root) + "/sys/bus/pci/devices/" + lower(bdf) + "/numa_node", line)) return -1;
	char *end = nullptr;
	const long v = strtol(line.c_str(), &end, 10);
	return end == line.c_str() || v < 0 ? -1 : (int)v;
}

bool cpus_of_node(int node, const char *root, std::vector<int> &cpus)
{
	cpus.clear();
	if (node < 0) return false;
	std::string line;
	if (!read_line(sysfs_root(root) + "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", line)) return false;
	return parse_cpulist(line, cpus) && !cpus.empty();
}

static bool numa_enabled() { const char *v = getenv("MM2GB_NUMA"); return !(v && *v == '0'); }

} // namespace mm2gb

using namespace mm2gb;

extern "C" {

int mm2gb_numa_cpus_for_bdf(const char *bdf, const char *sysfs_root_dir, int32_t *node_out, int32_t *cpus, int32_t max_cpus)
{
	const int node = numa_node_of_bdf(bdf, sysfs_root_dir);
	if (node_out) *node_out = node;
	std::vector<int> list;
	if (!cpus_of_node(node, sysfs_root_dir, list)) return 0;
	const int n = (int)std::min<size_t>(list.size(), (size_t)std::max(0, max_cpus));
	for (int k = 0; k < n && cpus; ++k) cpus[k] = list[(size_t)k];
	return (int)list.size();
}

int mm2gb_device_numa_node(int device)
{
	char bdf[64] = { 0 };
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
	return numa_node_of_bdf(bdf, nullptr);
}

// Moves the CALLING thread (threads it starts afterwards inherit the mask) onto the CPUs of `device`'s NUMA node that the process may use.
// Returns the number of CPUs in the new mask, 0 when nothing was changed (node unknown, no CPU of it usable, MM2GB_NUMA=0), -1 on error.
int mm2gb_pin_thread_to_device(int device)
{
	if (!numa_enabled()) return 0;
	std::vector<int> cpus;
	if (!cpus_of_node(mm2gb_device_numa_node(device), nullptr, cpus)) return 0;
	cpu_set_t now, want;
	if (sched_getaffinity(0, sizeof now, &now) != 0) return -1;
	CPU_ZERO(&want);
	int n = 0;
	for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &now)) { CPU_SET(c, &want); ++n; }
	if (n == 0 || n == CPU_COUNT(&now)) return 0;                     // no usable CPU there, or already exactly there: nothing changed
	if (sched_setaffinity(0, sizeof want, &want) != 0) return -1;
	return n;
}

} // extern "C"
