// Which XCD does workgroup b of a 1-D grid land on? (the XcdMap of csrc/common.hpp assumes b % 8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(int * xcc, int * cu)
{
	if (threadIdx.x == 0)
	{
		unsigned x, h;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
		xcc[blockIdx.x] = (int) (x & 0xf);
		cu[blockIdx.x] = (int) h;
	}
	// keep the workgroup alive a little so that the grid spreads over the chip
	for (volatile int i = 0; i < 2000; i++) { }
}
int main()
{
	const int n = 4096;
	int * d, * c;
	hipMalloc(&d, n * 4); hipMalloc(&c, n * 4);
	std::vector<int> h(n), hc(n);
	for (int rep = 0; rep < 2; rep++)
	{
		hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, c);
		hipDeviceSynchronize();
	}
	hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
	printf("first 32 blocks -> xcc:");
	for (int i = 0; i < 32; i++) printf(" %d", h[i]);
	int match = 0, hist[16] = {0};
	for (int i = 0; i < n; i++) { hist[h[i] & 15]++; }
	// is xcc(b) == (xcc(0) + b) % 8 ?
	for (int i = 0; i < n; i++) match += (h[i] == (h[0] + i) % 8);
	printf("\nblocks with xcc == (xcc0 + b) %% 8: %d of %d; per-xcc counts:", match, n);
	for (int k = 0; k < 8; k++) printf(" %d", hist[k]);
	printf("\n");
	return 0;
}
