// Micro-benchmark (round 6): the radix pass's walk (ksort.h:128-139: an element out of its bucket starts a cycle; every step puts the element in
// hand at the head of its bucket and takes up what was there) as a per-LANE state machine over runs held whole in LDS: a wave takes as many runs
// as fit its LDS ("bundle"), lane j walks run j.  Blocks (heads | bytes | map of the elements not in their bucket) arrive by LDS-direct loads.
// Checked against the host's walk; reports steps per second against the one-lane-per-wave form's measured rates.
//   hipcc --offload-arch=gfx950 -O3 -o bundle_walk bundle_walk.hip && ./bundle_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int W = 64;
struct Task { long long stage; long long perm; int len; int pad; };
__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int block_bytes(int len) { return 1024 + pad16(len) + pad16(((len + 31) / 32 + 1) * 4); }

template <int KMAX>
__global__ __launch_bounds__(W) void k_bundle(const Task *tasks, int n_tasks, const unsigned char *stage, int *perm, int *cursor, int lds_bytes, unsigned long long *stat)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const int l = threadIdx.x;
	int carry = -1;
	unsigned long long iters = 0, bundles = 0;
	for (;;) {
		int K = 0, used = 0;
		int my_base = 0, my_len = 0; int *my_perm = nullptr;
		while (K < KMAX) {
			int q = carry; carry = -1;
			if (q < 0) { if (l == 0) q = atomicAdd(cursor, 1); q = __builtin_amdgcn_readfirstlane(q); }
			if (q >= n_tasks) break;
			const Task t = tasks[q];
			const int need = block_bytes(t.len);
			if (used + need > lds_bytes) { carry = q; break; }
			const unsigned char *src = stage + t.stage;
			for (int u = 0; u < need; u += 16 * W)
				if (u + l * 16 < need)
					__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + u + l * 16), (void __attribute__((address_space(3)))*)(lds + used + u), 16, 0, 0);
			if (l == K) { my_base = used; my_len = t.len; my_perm = perm + t.perm; }
			used += need; ++K;
		}
		if (K == 0) break;
		__builtin_amdgcn_s_waitcnt(0);
		__builtin_amdgcn_wave_barrier();
		++bundles;
		if (l < K) {
			int *heads = (int*)(lds + my_base);
			const unsigned char *lb = lds + my_base + 1024;
			unsigned *bm = (unsigned*)(lds + my_base + 1024 + pad16(my_len));
			int i = 0, home = -1, d = 0, src = 0;
			bool stepping = false, done = my_len <= 0;
			while (!done) {
				++iters;
				int a = 0;
				if (stepping) a = heads[d];
				const unsigned w = bm[i >> 5];
				int pos = -1;
				bool real = false;
				if (stepping) {
					if (a <= home) { my_perm[src] = home; stepping = false; }
					else { pos = a; real = true; }
				}
				if (!stepping) {
					const unsigned m = w & (~0u << (i & 31));
					if (m == 0) { i = (i | 31) + 1; if (i >= my_len) done = true; }
					else { home = (i & ~31) + __builtin_ctz(m); i = home + 1; pos = home; stepping = true; }
				}
				if (pos >= 0) {
					const int nb = lb[pos];
					if (real) { my_perm[src] = pos; heads[d] = pos + 1; atomicAnd(&bm[pos >> 5], ~(1u << (pos & 31))); }
					src = pos; d = nb;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (stat) { atomicAdd(&stat[0], iters); if (l == 0) atomicAdd(&stat[1], bundles); }
}

// host: the reference's walk on one run's bytes -> where every element goes; also heads and the map
static long long host_walk(const unsigned char *b, int len, int *perm, int *heads_out, unsigned *bm)
{
	int cnt[257] = { 0 };
	for (int i = 0; i < len; ++i) ++cnt[b[i] + 1];
	for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
	int head[256], end[256];
	for (int k = 0; k < 256; ++k) { head[k] = cnt[k]; end[k] = cnt[k + 1]; heads_out[k] = cnt[k]; }
	const int words = (len + 31) / 32 + 1;
	for (int w = 0; w < words; ++w) bm[w] = 0;
	for (int i = 0; i < len; ++i) { perm[i] = i; if (i < cnt[b[i]] || i >= cnt[b[i] + 1]) bm[i >> 5] |= 1u << (i & 31); }
	long long steps = 0;
	for (int k = 0; k < 256; ++k)
		while (head[k] < end[k]) {
			int d = b[head[k]];
			if (d == k) { ++head[k]; continue; }
			const int home = head[k];
			int src = home;
			do { const int pos = head[d]++; const int nb = b[pos]; perm[src] = pos; src = pos; d = nb; ++steps; } while (d != k);
			perm[src] = home; ++head[k];
		}
	return steps;
}

template <int KMAX>
static void run(const char *what, const std::vector<int> &lens, int kind, int lds_bytes, int waves_per_cu)
{
	std::mt19937 rng(11);
	std::vector<Task> tasks(lens.size());
	long long stage_n = 0, perm_n = 0;
	for (size_t t = 0; t < lens.size(); ++t) { tasks[t] = { stage_n, perm_n, lens[t], 0 }; stage_n += block_bytes(lens[t]); perm_n += lens[t]; }
	std::vector<unsigned char> stage((size_t)stage_n + 1024, 0);
	std::vector<int> ref((size_t)perm_n);
	long long steps = 0;
	std::vector<unsigned char> b;
	for (size_t t = 0; t < lens.size(); ++t) {
		const int len = lens[t];
		b.resize(len);
		for (int i = 0; i < len; ++i) {
			if (kind == 0) b[i] = (unsigned char)(rng() >> 11);                                   // every byte as likely: nearly all move, long cycles
			else { int v = (int)((long long)i * 256 / len) + (int)(rng() % 7) - 3; b[i] = (unsigned char)std::min(255, std::max(0, v)); }   // nearly in order: short cycles, many at home
		}
		unsigned char *blk = stage.data() + tasks[t].stage;
		steps += host_walk(b.data(), len, ref.data() + tasks[t].perm, (int*)blk, (unsigned*)(blk + 1024 + pad16(len)));
		memcpy(blk + 1024, b.data(), len);
	}
	unsigned char *d_stage; int *d_perm, *d_cursor; Task *d_tasks; unsigned long long *d_stat;
	CK(hipMalloc(&d_stage, stage.size())); CK(hipMemcpy(d_stage, stage.data(), stage.size(), hipMemcpyHostToDevice));
	CK(hipMalloc(&d_perm, (size_t)perm_n * 4)); CK(hipMalloc(&d_cursor, 4)); CK(hipMalloc(&d_stat, 16));
	CK(hipMalloc(&d_tasks, tasks.size() * sizeof(Task))); CK(hipMemcpy(d_tasks, tasks.data(), tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
	CK(hipFuncSetAttribute((const void*)k_bundle<KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e30f;
	std::vector<int> ident((size_t)perm_n);
	for (size_t t = 0; t < lens.size(); ++t) for (int i = 0; i < lens[t]; ++i) ident[tasks[t].perm + i] = i;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipMemcpy(d_perm, ident.data(), ident.size() * 4, hipMemcpyHostToDevice));
		CK(hipMemset(d_cursor, 0, 4)); CK(hipMemset(d_stat, 0, 16));
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_bundle<KMAX>), dim3(256 * waves_per_cu), dim3(W), lds_bytes, 0, d_tasks, (int)tasks.size(), d_stage, d_perm, d_cursor, lds_bytes, d_stat);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
	}
	std::vector<int> got((size_t)perm_n);
	CK(hipMemcpy(got.data(), d_perm, got.size() * 4, hipMemcpyDeviceToHost));
	unsigned long long st[2]; CK(hipMemcpy(st, d_stat, 16, hipMemcpyDeviceToHost));
	size_t bad = 0; for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
	printf("%-58s %6zu runs, %9lld elements, %9lld steps, LDS %6d B x %d per CU: %.3f ms, %.1f G steps/s, %.2f lane iterations per step, %.1f runs per bundle -- %s\n", what, lens.size(), perm_n, steps, lds_bytes, waves_per_cu, best,
	       steps / (best * 1e6), (double)st[0] / steps, (double)lens.size() / st[1], bad ? "MISMATCH" : "same as the host's walk");
	hipFree(d_stage); hipFree(d_perm); hipFree(d_cursor); hipFree(d_tasks); hipFree(d_stat);
}

int main()
{
	std::mt19937 rng(5);
	std::vector<int> big, small;
	// like the second level of the bench's sort (runs beyond the lines' reach): 6 314 runs, 15.8 K elements on average, 52 K the longest
	for (int t = 0; t < 6314; ++t) { double u = (rng() % 10000) / 10000.0; big.push_back(7169 + (int)(u * u * u * 44000)); }
	// like its third level: 34 937 runs of 1 840 elements on average
	for (int t = 0; t < 34937; ++t) { double u = (rng() % 10000) / 10000.0; small.push_back(65 + (int)(u * u * 5400)); }
	std::sort(big.begin(), big.end(), std::greater<int>()); std::sort(small.begin(), small.end(), std::greater<int>());
	run<16>("long runs, every byte as likely", big, 0, 80 * 1024, 2);
	run<16>("long runs, nearly in order", big, 1, 80 * 1024, 2);
	run<32>("short runs, every byte as likely", small, 0, 80 * 1024, 2);
	run<16>("short runs, every byte as likely", small, 0, 40 * 1024, 4);
	run<8>("short runs, every byte as likely", small, 0, 20 * 1024, 8);
	run<32>("short runs, nearly in order", small, 1, 80 * 1024, 2);
	run<16>("long runs, every byte as likely", big, 0, 53 * 1024, 3);
	return 0;
}
