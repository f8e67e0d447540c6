// l1_lookup_rate.hip -- what the CU's vector L1 (TCP) of MI355X sustains when every access HITS: the roof the traversal launch's L1 figures are priced
// against (bench.py roofline.binding.l1). MI355X_MICROARCH.md gives the L1's capacity (32 KiB per CU) and no throughput; this measures one.
//
// Every workgroup reads, again and again, a table small enough to stay in its CU's L1 (16 KiB), with the access shapes of kernels_trace.hip:
//   coalesced   lane i reads 16 B at (base + 16 i): a wave's global_load_dwordx4 covers 1 KiB of consecutive bytes
//   node        every lane picks its own 80-byte record (of 128) and reads its five 16-byte parts (the node fetch: 5 instructions, 64 records each)
//   triangle    every lane picks its own 48-byte record (of 256) and reads its three parts (the triangle fetch)
//   line        every lane reads 16 B of its own 128-byte line (one instruction touches 64 lines)
//   same        all lanes read the same 16 bytes
// The loads of an iteration do not depend on earlier loads (addresses come from a counter hash), eight iterations are in flight per lane, the grid is
// 8 workgroups of 256 lanes per CU: the unit under test is the L1's tag / data path, not latency.
// Printed per shape: wave-level load instructions per second (whole chip), per CU and clock (2.4 GHz), bytes per CU and clock. Run it under
//   rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD -- /tmp/l1_lookup_rate
// (tools/gpu_jobs/r06_run3.sh) for the look-ups the TCP counts per instruction and its busy cycles: look-ups per second at busy ~ 1 is the roof.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_lookup_rate tools/microbench/l1_lookup_rate.hip && /tmp/l1_lookup_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { COALESCED = 0, NODE = 1, TRIANGLE = 2, LINE = 3, SAME = 4, SHAPES = 5 };
#define TABLE_BYTES 16384u

template<int SHAPE>
__device__ __forceinline__ unsigned probe(const char * __restrict__ table, unsigned lane_id, unsigned counter) {
	const unsigned h = (lane_id * 2654435761u + counter * 2246822519u) >> 20;   // (cheap on purpose: the vector ALUs must not be what is measured)
	unsigned acc = 0;
	if (SHAPE == COALESCED) {
		uint4 v = *(const uint4 *)(table + ((counter * 1024u + (lane_id & 63u) * 16u) & (TABLE_BYTES - 1u)));
		acc = v.x ^ v.y ^ v.z ^ v.w;
	} else if (SHAPE == NODE) {
		const uint4 * r = (const uint4 *)(table + (h & 127u) * 80u);
		uint4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
		acc = (a.x ^ a.y ^ a.z ^ a.w) ^ (b.x ^ b.y ^ b.z ^ b.w) ^ (c.x ^ c.y ^ c.z ^ c.w) ^ (d.x ^ d.y ^ d.z ^ d.w) ^ (e.x ^ e.y ^ e.z ^ e.w);
	} else if (SHAPE == TRIANGLE) {
		const uint4 * r = (const uint4 *)(table + (h & 255u) * 48u);
		uint4 a = r[0], b = r[1], c = r[2];
		acc = (a.x ^ a.y ^ a.z ^ a.w) ^ (b.x ^ b.y ^ b.z ^ b.w) ^ (c.x ^ c.y ^ c.z ^ c.w);
	} else if (SHAPE == LINE) {
		uint4 v = *(const uint4 *)(table + (h & 127u) * 128u);
		acc = v.x ^ v.y ^ v.z ^ v.w;
	} else {
		uint4 v = *(const uint4 *)(table + (((counter + (lane_id >> 31)) * 16u) & (TABLE_BYTES - 1u)));   // (lane_id >> 31 is 0: a vector load of one address, not a scalar load)
		acc = v.x ^ v.y ^ v.z ^ v.w;
	}
	return acc;
}

template<int SHAPE>
__global__ void __launch_bounds__(256) kernel_l1_probe(const char * __restrict__ table, int iterations, unsigned * sink) {
	const unsigned lane_id = blockIdx.x * 256u + threadIdx.x;
	unsigned acc = 0;
	for (int i = 0; i < iterations; i += 8) {
		#pragma unroll
		for (int k = 0; k < 8; k++) acc ^= probe<SHAPE>(table, lane_id, unsigned(i + k));
	}
	if (acc == 0x9e3779b9u) *sink = acc;   // (never true for this table: keeps the loads alive)
}

int main(int argc, char ** argv) {
	hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double clock_hz = 2.4e9;
	std::vector<unsigned> host(TABLE_BYTES / 4);
	for (size_t i = 0; i < host.size(); i++) host[i] = unsigned(i * 2654435761u) | 1u;
	char * table; (void)hipMalloc(&table, TABLE_BYTES); (void)hipMemcpy(table, host.data(), TABLE_BYTES, hipMemcpyHostToDevice);
	unsigned * sink; (void)hipMalloc(&sink, 64);
	const int iterations = argc > 1 ? atoi(argv[1]) : 8192, blocks = cus * 8;
	const char * names[SHAPES] = { "coalesced (1 KiB per instruction)", "node (5 x 16 B of an 80 B record)", "triangle (3 x 16 B of a 48 B record)", "line (64 lines per instruction)", "same 16 bytes" };
	const int loads_per_probe[SHAPES] = { 1, 5, 3, 1, 1 };
	printf("%s, %d CUs; table %u bytes (L1-resident), %d workgroups x 256 lanes, %d probes per lane\n", prop.name, cus, TABLE_BYTES, blocks, iterations);
	printf("  %-38s %10s %16s %14s %14s\n", "shape", "ms", "wave loads / s", "per CU, clock", "B / CU, clock");
	for (int shape = 0; shape < SHAPES; shape++) {
		hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
		auto launch = [&]() {
			switch (shape) {
				case COALESCED: hipLaunchKernelGGL(kernel_l1_probe<COALESCED>, dim3(blocks), dim3(256), 0, 0, table, iterations, sink); break;
				case NODE:      hipLaunchKernelGGL(kernel_l1_probe<NODE>,      dim3(blocks), dim3(256), 0, 0, table, iterations, sink); break;
				case TRIANGLE:  hipLaunchKernelGGL(kernel_l1_probe<TRIANGLE>,  dim3(blocks), dim3(256), 0, 0, table, iterations, sink); break;
				case LINE:      hipLaunchKernelGGL(kernel_l1_probe<LINE>,      dim3(blocks), dim3(256), 0, 0, table, iterations, sink); break;
				default:        hipLaunchKernelGGL(kernel_l1_probe<SAME>,      dim3(blocks), dim3(256), 0, 0, table, iterations, sink); break;
			}
		};
		launch();
		float best = 1e30f;
		for (int r = 0; r < 3; r++) {
			(void)hipEventRecord(t0); launch(); (void)hipEventRecord(t1); (void)hipEventSynchronize(t1);
			float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1); if (ms < best) best = ms;
		}
		const double wave_loads = double(blocks) * 4.0 * double(iterations) * loads_per_probe[shape];   // 4 waves per workgroup
		const double rate = wave_loads / (best * 1e-3);
		printf("  %-38s %10.3f %16.4g %14.4f %14.1f\n", names[shape], best, rate, rate / cus / clock_hz, rate * 1024.0 / cus / clock_hz);
	}
	return 0;
}
