// pin_rate.hip -- what page-locked staging costs to MAKE and to GIVE BACK, alone and from 16 threads at once (the drop-in at -t 16 pins a
// stage's staging on every stream at the same moment): hipHostMalloc / hipHostFree against hipHostRegister / hipHostUnregister of a
// 2 MB-aligned malloc block with MADV_HUGEPAGE (first touch included), and H2D straight from pageable memory for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 pin_rate.hip -o pin_rate -lpthread ; run: ./pin_rate [MiB per buffer, default 256] [threads, default 16]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Times { double make = 0, touch = 0, h2d = 0, give_back = 0; };

static Times one(int kind, size_t bytes, void *dev, hipStream_t s)
{
	Times t;
	void *p = nullptr;
	double t0 = now();
	if (kind == 0) { if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) exit(2); }
	else {
		if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) exit(2);
		(void)madvise(p, bytes, MADV_HUGEPAGE);
		if (kind == 1) { memset(p, 0, bytes); if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) exit(2); }
	}
	t.make = now() - t0;
	t0 = now();
	memset(p, 1, bytes);
	t.touch = now() - t0;
	t0 = now();
	if (hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) exit(3);
	t.h2d = now() - t0;
	t0 = now();
	if (kind == 0) (void)hipHostFree(p);
	else { if (kind == 1) (void)hipHostUnregister(p); free(p); }
	t.give_back = now() - t0;
	return t;
}

int main(int argc, char **argv)
{
	const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
	const int nt = argc > 2 ? atoi(argv[2]) : 16;
	const char *names[] = { "hipHostMalloc", "malloc+THP+hipHostRegister", "pageable (malloc+THP), no pinning" };
	std::vector<void*> dev((size_t)nt);
	std::vector<hipStream_t> st((size_t)nt);
	for (int i = 0; i < nt; ++i) { if (hipMalloc(&dev[i], bytes) != hipSuccess) return 1; if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return 1; }
	for (int kind = 0; kind < 3; ++kind) {
		(void)one(kind, bytes, dev[0], st[0]);                       // first use
		const Times a = one(kind, bytes, dev[0], st[0]);
		printf("%-36s 1 thread  x %4zu MiB: make %.3f s (%.2f s/GB) | touch %.3f s | H2D %.3f s (%.1f GB/s) | give back %.3f s\n", names[kind], bytes >> 20, a.make, a.make / (bytes / 1e9), a.touch,
		       a.h2d, bytes / a.h2d / 1e9, a.give_back);
		std::vector<Times> r((size_t)nt);
		std::vector<std::thread> pool;
		const double t0 = now();
		for (int i = 0; i < nt; ++i) pool.emplace_back([&, i] { r[(size_t)i] = one(kind, bytes, dev[(size_t)i], st[(size_t)i]); });
		for (auto &th : pool) th.join();
		const double wall = now() - t0;
		Times m;
		for (const Times &x : r) { m.make = std::max(m.make, x.make); m.touch = std::max(m.touch, x.touch); m.h2d = std::max(m.h2d, x.h2d); m.give_back = std::max(m.give_back, x.give_back); }
		printf("%-36s %d threads x %4zu MiB: make %.3f s max | touch %.3f s | H2D %.3f s (%.1f GB/s aggregate) | give back %.3f s | wall %.3f s\n", names[kind], nt, bytes >> 20, m.make, m.touch, m.h2d,
		       nt * (double)bytes / m.h2d / 1e9, m.give_back, wall);
	}
	return 0;
}
