// Micro-benchmark (round 6): what one step of a lone lane's bucket walk costs on gfx950, in the forms the sort's radix pass could take.
// Each wave owns N random bytes in LDS and 256 bucket heads; a "step" goes from bucket d to the byte at d's head and advances the head.
//   hipcc --offload-arch=gfx950 -O3 -o walk_step walk_step.hip && ./walk_step
// Forms: 0 head (b32) then byte, perm entry to memory per step (what k_post_sort_level's resident walk does)
//        1 the same without the store to memory
//        2 record = head | four bytes ahead in one 64-bit word: one LDS read per step, a second every fourth visit of a bucket
//        3 form 0 on K lanes at once, each lane its own run (state in LDS side by side)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int W = 64, N = 3584, THREADS = 256;

template <int FORM, int K>
__global__ __launch_bounds__(THREADS) void k_walk(const unsigned char *bytes, int *perm, long long *ticks, int *sink, int steps)
{
	extern __shared__ unsigned char smem[];
	const int wave = threadIdx.x / W, l = threadIdx.x % W;
	constexpr int PER_RUN = 256 * 8 + N;                  // heads/records, bytes
	unsigned char *base = smem + (size_t)wave * K * PER_RUN;
	const size_t wave_g = ((size_t)blockIdx.x * (THREADS / W) + wave) * K;
	for (int j = 0; j < K; ++j) {
		unsigned char *lb = base + j * PER_RUN + 256 * 8;
		const unsigned char *src = bytes + ((wave_g + j) % 64) * N;
		for (int i = l; i < N; i += W) lb[i] = src[i];
		unsigned long long *rec = (unsigned long long*)(base + j * PER_RUN);
		for (int k = l; k < 256; k += W) rec[k] = (unsigned)(k * (N / 256));
	}
	__syncthreads();
	if (FORM == 2) {
		for (int j = 0; j < K; ++j) {
			unsigned char *lb = base + j * PER_RUN + 256 * 8;
			unsigned long long *rec = (unsigned long long*)(base + j * PER_RUN);
			for (int k = l; k < 256; k += W) { const unsigned p = k * (N / 256); rec[k] = (unsigned long long)p | (unsigned long long)*(unsigned*)(lb + (p & ~3u)) << 32; }
		}
		__syncthreads();
	}
	const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
	int d = l & 255, src = 0, acc = 0;
	if (l < K) {
		unsigned char *lb = base + l * PER_RUN + 256 * 8;
		int *perm_w = perm + (wave_g + l) * N;
		if (FORM == 0 || FORM == 1 || FORM == 3) {
			int *head = (int*)(base + l * PER_RUN);           // stride 8 bytes: head[2 d]
			for (int s = 0; s < steps; ++s) {
				int pos = head[2 * d];
				const int nb = lb[pos];
				if (FORM != 1) perm_w[src] = pos;
				int nx = pos + 1; if (nx == (d + 1) * (N / 256)) nx = d * (N / 256);
				head[2 * d] = nx;
				src = pos; d = nb; acc += pos;
			}
		} else {
			unsigned long long *rec = (unsigned long long*)(base + l * PER_RUN);
			for (int s = 0; s < steps; ++s) {
				const unsigned long long r = rec[d];
				const unsigned pos = (unsigned)r; unsigned win = (unsigned)(r >> 32);
				const int nb = win & 255;
				perm_w[src] = (int)pos;
				unsigned nx = pos + 1; if (nx == (unsigned)(d + 1) * (N / 256)) nx = d * (N / 256);
				if ((nx & 3u) == 0 || nx == (unsigned)d * (N / 256)) win = *(unsigned*)(lb + (nx & ~3u)) >> ((nx & 3u) * 8); else win >>= 8;
				rec[d] = (unsigned long long)nx | (unsigned long long)win << 32;
				src = (int)pos; d = nb; acc += (int)pos;
			}
		}
	}
	const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
	if (l == 0) ticks[(size_t)blockIdx.x * (THREADS / W) + wave] = t1 - t0;
	if (acc == 0x7fffffff) sink[0] = acc + d;
}

template <int FORM, int K>
static void run(const char *what, int blocks, const unsigned char *d_bytes, int *d_perm, long long *d_ticks, int *d_sink, int steps)
{
	const size_t lds = (size_t)(THREADS / W) * K * (256 * 8 + N);
	CK(hipFuncSetAttribute((const void*)k_walk<FORM, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_walk<FORM, K>), dim3(blocks), dim3(THREADS), lds, 0, d_bytes, d_perm, d_ticks, d_sink, steps);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	}
	float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<long long> t((size_t)blocks * (THREADS / W));
	CK(hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost));
	double sum = 0; for (long long v : t) sum += (double)v;
	printf("%-64s blocks %5d K %2d LDS/block %6zu B: kernel %.3f ms, %.1f ns per step per walker (wave clock), %.2f G steps/s in all\n", what, blocks, K, lds, ms,
	       sum / t.size() * 10.0 / steps, (double)blocks * (THREADS / W) * K * steps / (ms * 1e6));
}

int main()
{
	const int steps = 20000;
	std::vector<unsigned char> h(64 * N);
	srand(7); for (auto &b : h) b = (unsigned char)(rand() >> 7);
	unsigned char *d_bytes; int *d_perm, *d_sink; long long *d_ticks;
	const int max_walkers = 1024 * 4 * 16;
	CK(hipMalloc(&d_bytes, h.size())); CK(hipMemcpy(d_bytes, h.data(), h.size(), hipMemcpyHostToDevice));
	CK(hipMalloc(&d_perm, (size_t)max_walkers * N * 4)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_ticks, (size_t)1024 * 4 * 8));
	run<0, 1>("head then byte, perm entry to memory (as now), one block", 1, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<1, 1>("the same, no store to memory, one block", 1, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<2, 1>("record with four bytes ahead, one block", 1, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<0, 1>("head then byte, 3 blocks per CU", 768, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<1, 1>("no store to memory, 3 blocks per CU", 768, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<2, 1>("record with four bytes ahead, 3 blocks per CU", 768, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<3, 2>("two lanes, two runs, 3 blocks per CU", 768, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<3, 4>("four lanes, 1 block per CU", 256, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<3, 2>("two lanes, 1 block per CU", 256, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<0, 1>("one lane, 1 block per CU", 256, d_bytes, d_perm, d_ticks, d_sink, steps);
	run<3, 4>("four lanes, 2 blocks per CU", 512, d_bytes, d_perm, d_ticks, d_sink, steps);
	return 0;
}
