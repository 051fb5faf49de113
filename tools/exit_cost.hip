/* tools/exit_cost.hip -- what a process that has used the HIP runtime costs to START and to END, by what it allocated.
 *   exit_cost [streams] [pinned MiB] [device MiB] [threads] [touch MiB] [mode]
 *   mode: 0 sleepers poll every 1 ms, 1 sleepers in pause(), 2 sleepers spin, 3 like 0 but the main thread leaves with
 *   SYS_exit and a helper calls exit_group 2 ms later, 4 the main thread sleeps 20 ms between the last line and _exit
 * prints the time from main() to "runtime up" and the wall-clock instant right before _exit(0); the caller (tools/r04_exit2.sh)
 * takes its own clock when the process has gone and reports the difference: the kernel-side teardown nobody in the process
 * can see.  hipcc -O2 tools/exit_cost.hip -o build/exit_cost */
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

static double now() { struct timespec t; clock_gettime(CLOCK_REALTIME, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

__global__ void touch(unsigned *p) { p[threadIdx.x] = threadIdx.x; }

int main(int argc, char **argv)
{
	const double t0 = now();
	const int n_streams = argc > 1 ? atoi(argv[1]) : 0;
	const size_t pinned = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 0) << 20, dev = (argc > 3 ? strtoull(argv[3], nullptr, 10) : 0) << 20;
	const int n_threads = argc > 4 ? atoi(argv[4]) : 0;
	const size_t touch_bytes = (argc > 5 ? strtoull(argv[5], nullptr, 10) : 0) << 20;
	const int mode = argc > 6 ? atoi(argv[6]) : 0;
	void *p = nullptr;
	if (hipMalloc(&p, 1 << 20) != hipSuccess) return 1;
	const double t1 = now();
	std::vector<hipStream_t> st((size_t) n_streams);
	for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	void *h = nullptr, *d = nullptr;
	if (pinned) { hipHostMalloc(&h, pinned, hipHostMallocPortable); memset(h, 1, pinned); }
	if (dev) { hipMalloc(&d, dev); hipMemset(d, 0, dev); }
	for (auto &s : st) touch<<<1, 64, 0, s>>>((unsigned *) p);
	touch<<<1, 64>>>((unsigned *) p);
	hipDeviceSynchronize();
	std::atomic<bool> quit { false };
	std::vector<std::thread> pool;
	for (int i = 0; i < n_threads; ++i)
		pool.emplace_back([&]() {
			if (mode == 1) for (;;) pause();
			if (mode == 2) while (!quit.load()) { }
			while (!quit.load()) usleep(1000);
		});
	if (touch_bytes) { char *m = (char *) malloc(touch_bytes); memset(m, 1, touch_bytes); }
	const double t2 = now();
	printf("%.6f init %.4f work %.4f\n", t2, t1 - t0, t2 - t1);
	fflush(stdout);
	if (mode == 3) {
		std::thread([]() { usleep(2000); syscall(SYS_exit_group, 0); }).detach();
		syscall(SYS_exit, 0);
	}
	if (mode == 4) usleep(20000);
	_exit(0);
}
