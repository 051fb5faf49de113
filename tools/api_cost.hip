/* Cost of the HIP runtime calls a producer lane needs: stream/event creation, device and
 * pinned allocation, and their release.  hipcc --offload-arch=gfx950 tools/api_cost.hip -o build/api_cost */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
template <class F> static double ms(F f) { auto t0 = std::chrono::steady_clock::now(); f(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
int main()
{
	printf("hipFree(0) (runtime init): %.1f ms\n", ms([] { (void) hipFree(nullptr); }));
	const int N = 16;
	std::vector<hipStream_t> st(N);
	std::vector<hipEvent_t> ev(N);
	std::vector<void *> d(N), h(N);
	printf("hipStreamCreateWithFlags x%d: %.2f ms each\n", N, ms([&] { for (auto &s : st) (void) hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }) / N);
	printf("hipEventCreateWithFlags x%d: %.3f ms each\n", N, ms([&] { for (auto &e : ev) (void) hipEventCreateWithFlags(&e, hipEventDisableTiming); }) / N);
	for (size_t mb : { 4, 16, 64 }) {
		printf("hipMalloc %zu MiB x%d: %.2f ms each", mb, N, ms([&] { for (auto &p : d) (void) hipMalloc(&p, mb << 20); }) / N);
		printf(", hipFree: %.2f ms each\n", ms([&] { for (auto &p : d) (void) hipFree(p); }) / N);
		printf("hipHostMalloc %zu MiB x%d: %.2f ms each", mb, N, ms([&] { for (auto &p : h) (void) hipHostMalloc(&p, mb << 20, hipHostMallocPortable); }) / N);
		printf(", hipHostFree: %.2f ms each\n", ms([&] { for (auto &p : h) (void) hipHostFree(p); }) / N);
	}
	void *big;
	printf("hipHostMalloc 512 MiB: %.1f ms", ms([&] { (void) hipHostMalloc(&big, 512ull << 20, hipHostMallocPortable); }));
	printf(", hipHostFree: %.1f ms\n", ms([&] { (void) hipHostFree(big); }));
	/* first use of a stream: memcpy + sync */
	void *dp, *hp;
	(void) hipMalloc(&dp, 1 << 20); (void) hipHostMalloc(&hp, 1 << 20, 0);
	printf("first async copy+sync on each stream: %.3f ms each\n", ms([&] { for (auto &s : st) { (void) hipMemcpyAsync(dp, hp, 1 << 20, hipMemcpyHostToDevice, s); (void) hipStreamSynchronize(s); } }) / N);
	printf("second async copy+sync on each stream: %.3f ms each\n", ms([&] { for (auto &s : st) { (void) hipMemcpyAsync(dp, hp, 1 << 20, hipMemcpyHostToDevice, s); (void) hipStreamSynchronize(s); } }) / N);
	printf("hipStreamDestroy x%d: %.2f ms each\n", N, ms([&] { for (auto &s : st) (void) hipStreamDestroy(s); }) / N);
	printf("hipEventDestroy x%d: %.3f ms each\n", N, ms([&] { for (auto &e : ev) (void) hipEventDestroy(e); }) / N);
	return 0;
}
