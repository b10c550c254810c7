// LDS read probe (gfx950): does ds_read_b32 take byte addresses that are not multiples of 4, and
// what do random reads cost?  Build: hipcc --offload-arch=gfx950 -O2 -o lds_probe lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
constexpr int kBytes = 120 * 1024;
// mode 0: u8 + u16 at t and base+2t (SoA); 1: b32 at 3t (unaligned); 2: b32 at 4t (aligned); 3: u8 only; 4: b32 at 3t via two aligned... n/a
__global__ void __launch_bounds__(1024) probe(int mode, int iters, uint32_t nslots, uint32_t *out, unsigned long long *cyc) {
	extern __shared__ unsigned char smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)smem;
	for (uint32_t i = threadIdx.x; i < kBytes; i += blockDim.x) lds[i] = (unsigned char)(i * 7 + 3);
	__syncthreads();
	uint32_t st = (threadIdx.x + blockIdx.x * 1024u) * 2654435761u, t = (st >> 10) & (nslots - 1), acc = 0;
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t base_a = nslots;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; i++) {
		uint32_t v;
		if (mode == 0) {
			uint32_t a, b;
			asm volatile("ds_read_u8 %0, %2\n ds_read_u16 %1, %3\n s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(t), "v"(base_a + 2 * t) : "memory");
			v = a | (b << 8);
		} else if (mode == 1) {
			asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(3 * t) : "memory");
			v &= 0xFFFFFF;
		} else if (mode == 2) {
			asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(4 * (t % (kBytes / 4))) : "memory");
			v &= 0xFFFFFF;
		} else if (mode == 3) {
			asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(t) : "memory");
		} else if (mode == 4) { // ring pattern: lane * 32 + random offset
			asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(wv * 2048 + lane * 32 + (t & 31)) : "memory");
		} else if (mode == 5) { // interleaved ring: dword j of lane l at (j * 64 + l) * 4
			asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(wv * 2048 + ((t & 28) << 6) + lane * 4 + (t & 3)) : "memory");
		} else if (mode == 6) { // u16 random
			asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(2 * t) : "memory");
		} else { // b64 aligned random
			uint64_t w;
			asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(8 * (t & 8191)) : "memory");
			v = (uint32_t)w & 0xFFFFFF;
		}
		acc += v;
		st = st * 1664525u + 1013904223u + (v & 1u);
		t = (st >> 10) & (nslots - 1);
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
	const uint32_t nslots = 32768;
	uint32_t *d_out; unsigned long long *d_cyc;
	hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_cyc, 256 * 8);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
	// correctness of mode 1 against host model
	{
		probe<<<1, 64, kBytes>>>(1, 1, nslots, d_out, d_cyc);
		std::vector<uint32_t> h(64); hipMemcpy(h.data(), d_out, 256, hipMemcpyDeviceToHost);
		int bad = 0;
		for (uint32_t l = 0; l < 64; l++) {
			uint32_t t = ((l * 2654435761u) >> 10) & (nslots - 1), a = 3 * t, want = 0;
			for (int k = 0; k < 3; k++) want |= (uint32_t)(unsigned char)((a + k) * 7 + 3) << (8 * k);
			if (h[l] != want) bad++;
		}
		printf("unaligned ds_read_b32: %s (%d of 64 lanes differ)\n", bad ? "WRONG" : "ok", bad);
	}
	for (int waves : {8, 16})
		for (int mode = 0; mode < 8; mode++) {
			const int iters = 2000;
			probe<<<256, 64 * waves, kBytes>>>(mode, iters, nslots, d_out, d_cyc);
			hipDeviceSynchronize();
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0);
			probe<<<256, 64 * waves, kBytes>>>(mode, iters, nslots, d_out, d_cyc);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			std::vector<unsigned long long> c(256); hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
			double avg = 0; for (auto x : c) avg += x; avg /= 256;
			printf("waves %2d mode %d: %.3f ms, %.1f cycles per dependent step per wave, %.2f cycles per wave-step per CU\n", waves, mode, ms, avg / iters, avg / iters / waves);
		}
	return 0;
}
