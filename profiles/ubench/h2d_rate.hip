// h2d_rate.hip -- page-locked host -> device copy rate with one and with several streams (does splitting a copy over two SDMA
// queues help?) and device -> host at the same time.   hipcc --offload-arch=gfx950 -O2 h2d_rate.hip -o h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t bytes = (size_t)2 << 30;
	char *h, *h2, *d, *d2;
	CHECK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault)); CHECK(hipHostMalloc((void**)&h2, bytes, hipHostMallocDefault));
	CHECK(hipMalloc((void**)&d, bytes)); CHECK(hipMalloc((void**)&d2, bytes));
	memset(h, 1, bytes); memset(h2, 2, bytes);
	hipStream_t s[4];
	for (auto &q : s) CHECK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
	for (int n : { 1, 2, 4 }) {
		double best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipDeviceSynchronize());
			const double t0 = now();
			for (int k = 0; k < n; ++k) CHECK(hipMemcpyAsync(d + bytes / n * k, h + bytes / n * k, bytes / n, hipMemcpyHostToDevice, s[k]));
			for (int k = 0; k < n; ++k) CHECK(hipStreamSynchronize(s[k]));
			best = std::min(best, now() - t0);
		}
		printf("H2D 2 GiB over %d stream(s): %.1f GB/s\n", n, bytes / best / 1e9);
	}
	{
		double best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipDeviceSynchronize());
			const double t0 = now();
			CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[0]));
			CHECK(hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, s[1]));
			CHECK(hipStreamSynchronize(s[0])); CHECK(hipStreamSynchronize(s[1]));
			best = std::min(best, now() - t0);
		}
		printf("H2D 2 GiB + D2H 2 GiB at once: %.1f GB/s each way\n", bytes / best / 1e9);
	}
	return 0;
}
