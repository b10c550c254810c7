// LDS throughput probe (gfx950): LDS-array cost of one wave-instruction for random / ring-like addresses,
// with 8 independent reads in flight per wave.  Build: hipcc --offload-arch=gfx950 -O2 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
constexpr int kBytes = 128 * 1024;
template <int MODE>
__global__ void __launch_bounds__(1024) rate(int iters, uint32_t *out, unsigned long long *cyc) {
	extern __shared__ unsigned char smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)smem;
	for (uint32_t i = threadIdx.x; i < kBytes; i += blockDim.x) lds[i] = (unsigned char)(i * 7 + 3);
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t st[8], acc = 0;
	for (int k = 0; k < 8; k++) st[k] = (threadIdx.x * 8 + k + blockIdx.x * 8192u) * 2654435761u;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++) {
		uint32_t a[8], v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			st[k] = st[k] * 1664525u + 1013904223u;
			const uint32_t r = st[k] >> 12;
			if (MODE == 0) a[k] = r & 32767;                                  // u8 random over 32 KB
			else if (MODE == 1) a[k] = (r & 32767) * 2;                        // u16 random over 64 KB
			else if (MODE == 2) a[k] = (r & 16383) * 4;                        // b32 random
			else if (MODE == 3) a[k] = wv * 2048 + lane * 32 + (r & 31);       // u8, rings side by side
			else if (MODE == 4) a[k] = wv * 2048 + ((r & 28) << 6) + lane * 4 + (r & 3); // u8, rings interleaved
			else a[k] = (r & 8191) * 8;                                        // b64 random
		}
#pragma unroll
		for (int k = 0; k < 8; k++) {
			if (MODE == 0 || MODE == 3 || MODE == 4) asm volatile("ds_read_u8 %0, %1" : "=v"(v[k]) : "v"(a[k]) : "memory");
			else if (MODE == 1) asm volatile("ds_read_u16 %0, %1" : "=v"(v[k]) : "v"(a[k]) : "memory");
			else if (MODE == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[k]) : "v"(a[k]) : "memory");
			else { uint64_t w; asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a[k]) : "memory"); v[k] = (uint32_t)w; }
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int k = 0; k < 8; k++) acc += v[k];
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, uint32_t *d_out, unsigned long long *d_cyc) {
	hipFuncSetAttribute((const void *)rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
	for (int waves : {4, 16}) {
		const int iters = 500;
		rate<MODE><<<256, 64 * waves, kBytes>>>(iters, d_out, d_cyc);
		rate<MODE><<<256, 64 * waves, kBytes>>>(iters, d_out, d_cyc);
		hipDeviceSynchronize();
		std::vector<unsigned long long> c(256);
		hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
		double avg = 0;
		for (auto x : c) avg += x;
		avg /= 256;
		printf("%-28s waves %2d: %.2f cycles per wave-instruction per CU\n", name, waves, avg / iters / 8 / waves);
	}
}
int main() {
	uint32_t *d_out; unsigned long long *d_cyc;
	(void)hipMalloc(&d_out, 256 * 1024 * 4); (void)hipMalloc(&d_cyc, 256 * 8);
	run<0>("u8 random 32K", d_out, d_cyc);
	run<1>("u16 random 64K", d_out, d_cyc);
	run<2>("b32 random 64K", d_out, d_cyc);
	run<3>("u8 rings side by side", d_out, d_cyc);
	run<4>("u8 rings interleaved", d_out, d_cyc);
	run<5>("b64 random 64K", d_out, d_cyc);
	return 0;
}
