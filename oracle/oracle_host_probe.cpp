// oracle_host_probe.cpp -- TEST / BENCH INFRASTRUCTURE (never on the product path).
//
// Effective parallelism of the host the CPU baseline runs on: BASELINE.md 3 asks for it next to the core count, because a
// container's `nproc` says little about the cores it actually gets (the survey box: 8 logical CPUs, an 8-thread spin loop
// 5.1x slower per thread than one thread, i.e. ~1.6 effective cores). The probe runs the same fixed dependent-FMA loop on
// one thread and then on `threads` threads at once and returns threads x t_1 / t_threads.
#include <chrono>
#include <thread>
#include <vector>

static double spin(long long iterations) {
	volatile float sink;
	float a = 1.0001f, b = 0.9999f, x = 0.5f;
	auto t0 = std::chrono::steady_clock::now();
	for (long long i = 0; i < iterations; i++) x = __builtin_fmaf(x, a, -0.00001f) * b;
	sink = x; (void)sink;
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

extern "C" double oracle_effective_parallelism(int threads, double seconds_per_run) {
	if (threads < 1) threads = 1;
	long long iterations = 20 * 1000 * 1000;
	double t1 = spin(iterations);
	if (t1 > 0.0) iterations = (long long)(double(iterations) * seconds_per_run / t1);
	if (iterations < 1000) iterations = 1000;
	t1 = spin(iterations);
	std::vector<double> elapsed(size_t(threads), 0.0);
	std::vector<std::thread> pool;
	auto t0 = std::chrono::steady_clock::now();
	for (int t = 0; t < threads; t++) pool.emplace_back([&elapsed, t, iterations]() { elapsed[size_t(t)] = spin(iterations); });
	for (std::thread & t : pool) t.join();
	double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return wall > 0.0 ? double(threads) * t1 / wall : 0.0;
}

#include <cstdlib>
#include <omp.h>
extern "C" int oracle_default_threads(void) {
	if (const char * e = getenv("ORACLE_THREADS")) { int n = atoi(e); if (n > 0) return n; }
	int n = omp_get_max_threads();
	return n < 32 ? n : 32;
}
