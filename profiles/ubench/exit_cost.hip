// exit_cost.hip -- what a process that holds many engines' memory pays to END: D GB of device memory in `nd` allocations and P GB of
// page-locked host memory (2 MB-aligned mmap + MADV_HUGEPAGE + hipHostRegister, touched) in `np` blocks, `ns` streams, then one of
//   free  : hipFree / hipHostUnregister + munmap / hipStreamDestroy, timed one kind at a time, then return from main
//   leave : return from main with everything alive (the HIP runtime's own teardown at exit)
//   quick : fflush + _exit(0) with everything alive (no runtime teardown: the kernel reclaims)
// The caller times the whole process (profiles/exit_cost.sh); the program prints the time of day at which main gives up control.
// Build: hipcc --offload-arch=gfx950 -O2 exit_cost.hip -o exit_cost
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/time.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double epoch() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + tv.tv_usec * 1e-6; }
__global__ void touch(char *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i] = 1; }

int main(int argc, char **argv)
{
	if (argc < 7) { fprintf(stderr, "usage: exit_cost <device GB> <device allocations> <pinned GB> <pinned blocks> <streams> free|leave|quick\n"); return 2; }
	const double dgb = atof(argv[1]), pgb = atof(argv[3]);
	const int nd = atoi(argv[2]), np = atoi(argv[4]), ns = atoi(argv[5]);
	const char *mode = argv[6];
	const size_t dbytes = nd ? (size_t)(dgb * 1e9 / nd) : 0, pbytes = np ? ((size_t)(pgb * 1e9 / np) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1) : 0;
	std::vector<void*> dev((size_t)nd), pin((size_t)np);
	std::vector<hipStream_t> st((size_t)ns);
	double t0 = now();
	for (auto &s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
	const double t_streams = now() - t0;
	t0 = now();
	for (auto &d : dev) { if (hipMalloc(&d, dbytes) != hipSuccess) return 1; touch<<<256, 256, 0, ns ? st[0] : nullptr>>>((char*)d, dbytes); }
	(void)hipDeviceSynchronize();
	const double t_dev = now() - t0;
	t0 = now();
	for (auto &p : pin) {
		p = mmap(nullptr, pbytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (p == MAP_FAILED) return 1;
		(void)madvise(p, pbytes, MADV_HUGEPAGE);
		memset(p, 1, pbytes);
		if (hipHostRegister(p, pbytes, hipHostRegisterDefault) != hipSuccess) return 1;
	}
	const double t_pin = now() - t0;
	printf("made: %d streams %.3f s | %.1f GB device in %d allocations %.3f s | %.1f GB page-locked in %d blocks %.3f s\n", ns, t_streams, dgb, nd, t_dev, pgb, np, t_pin);
	if (!strcmp(mode, "free")) {
		t0 = now();
		for (auto &d : dev) (void)hipFree(d);
		const double f_dev = now() - t0;
		t0 = now();
		for (auto &p : pin) (void)hipHostUnregister(p);
		const double f_unreg = now() - t0;
		t0 = now();
		for (auto &p : pin) (void)munmap(p, pbytes);
		const double f_unmap = now() - t0;
		t0 = now();
		for (auto &s : st) (void)hipStreamDestroy(s);
		printf("given back: hipFree %.3f s | hipHostUnregister %.3f s | munmap %.3f s | hipStreamDestroy %.3f s\n", f_dev, f_unreg, f_unmap, now() - t0);
	}
	printf("main ends at epoch %.6f (%s)\n", epoch(), mode);
	fflush(nullptr);
	if (!strcmp(mode, "quick")) _exit(0);
	return 0;
}
