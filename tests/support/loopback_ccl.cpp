// loopback_ccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so that lets TWO OR MORE PROCESSES THAT SHARE ONE GPU run the product's native frame
// exchange (rt_comm_unique_id -> rt_comm_init_rank(rank, world) -> rt_all_gather_framebuffer: pack kernel, ncclAllGather on the context's stream, unpack
// kernel; gpu-raytracer_amd/csrc/rt_api.hip) with world > 1 on a test box that has one MI355X. RCCL refuses a device twice in one communicator, so on such a
// box the real ncclAllGather only ever runs in a communicator of one rank (tests/test_gpu_rccl.py); this library implements the same eight entry points with the
// semantics the product relies on -- rank r's `count` elements land at recvbuff + r * count on every rank, ordered behind the work already on `stream` -- by
// staging each rank's chunk through a file in /dev/shm. It exists so that the FIRST time rank 1's tiles arrive in rank 0's frame is a test, not the first 8-GPU
// run. Selected with GRT_COLLECTIVE_LIBRARY=<path of this .so> (rt_api.hip: rccl_api); never loaded otherwise; nothing under gpu-raytracer_amd/ links it.
// Not a performance path: every call synchronises the stream and copies through the host.
//   hipcc -O2 -fPIC -shared -o tests/support/libloopback_ccl.so tests/support/loopback_ccl.cpp      (tests/support/Makefile, __graft_entry__.build())
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
enum { OK = 0, UNHANDLED_HIP_ERROR = 1, SYSTEM_ERROR = 2, INTERNAL_ERROR = 3, INVALID_ARGUMENT = 4, INVALID_USAGE = 5 };   // ncclResult_t values
enum { FLOAT32 = 7 };
struct UniqueId { char internal[128]; };
struct Control { std::atomic<int> arrived; std::atomic<int> generation; };
struct Comm {
	std::string token; int world = 0, rank = 0;
	Control * control = nullptr;
	void * own_map = nullptr; size_t own_bytes = 0; int own_fd = -1;
};
std::string control_name(const std::string & token) { return "/grtccl_" + token + "_ctl"; }
std::string data_name(const std::string & token, int rank) { return "/grtccl_" + token + "_r" + std::to_string(rank); }

// every rank arrives; the last one opens the next generation. A rank that waits longer than a minute gives up (a peer died: the test must fail, not hang).
bool barrier(Comm * c) {
	const int generation = c->control->generation.load();
	if (c->control->arrived.fetch_add(1) + 1 == c->world) { c->control->arrived.store(0); c->control->generation.fetch_add(1); return true; }
	const auto started = std::chrono::steady_clock::now();
	while (c->control->generation.load() == generation) {
		std::this_thread::sleep_for(std::chrono::microseconds(50));
		if (std::chrono::steady_clock::now() - started > std::chrono::seconds(60)) return false;
	}
	return true;
}
}   // namespace

extern "C" {

int ncclGetUniqueId(UniqueId * id) {
	if (!id) return INVALID_ARGUMENT;
	uint64_t bits = 0;
	FILE * f = fopen("/dev/urandom", "rb");
	if (!f || fread(&bits, 8, 1, f) != 1) { if (f) fclose(f); return SYSTEM_ERROR; }
	fclose(f);
	memset(id->internal, 0, sizeof(id->internal));
	snprintf(id->internal, sizeof(id->internal), "%016llx", (unsigned long long)bits);
	return OK;
}

int ncclCommInitRank(void ** comm, int world, UniqueId id, int rank) {
	if (!comm || world < 1 || rank < 0 || rank >= world) return INVALID_ARGUMENT;
	id.internal[127] = 0;
	Comm * c = new Comm; c->token = id.internal; c->world = world; c->rank = rank;
	int fd = shm_open(control_name(c->token).c_str(), O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) { if (fd >= 0) close(fd); delete c; return SYSTEM_ERROR; }   // (a new segment reads as zeros: nobody has arrived, generation 0)
	c->control = (Control *)mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (c->control == MAP_FAILED) { delete c; return SYSTEM_ERROR; }
	*comm = c;
	return OK;
}

int ncclCommInitAll(void **, int, const int *) { return INVALID_USAGE; }   // (one process, several GPUs: not what this stand-in is for)

int ncclCommDestroy(void * comm) {
	Comm * c = (Comm *)comm;
	if (!c) return INVALID_ARGUMENT;
	if (c->own_map) munmap(c->own_map, c->own_bytes);
	if (c->own_fd >= 0) close(c->own_fd);
	shm_unlink(data_name(c->token, c->rank).c_str());
	if (c->rank == 0) shm_unlink(control_name(c->token).c_str());   // (the others keep their mapping; a name is all that goes)
	munmap(c->control, sizeof(Control));
	delete c;
	return OK;
}

int ncclAllGather(const void * sendbuff, void * recvbuff, size_t count, int datatype, void * comm, hipStream_t stream) {
	Comm * c = (Comm *)comm;
	if (!c || !sendbuff || !recvbuff || datatype != FLOAT32) return INVALID_ARGUMENT;
	const size_t bytes = count * 4;
	if (hipStreamSynchronize(stream) != hipSuccess) return UNHANDLED_HIP_ERROR;           // what was enqueued before the collective (the pack kernel) has run
	if (c->own_bytes != bytes) {                                                           // this rank's chunk, staged in a file of its own
		if (c->own_map) munmap(c->own_map, c->own_bytes);
		if (c->own_fd < 0) c->own_fd = shm_open(data_name(c->token, c->rank).c_str(), O_CREAT | O_RDWR, 0600);
		if (c->own_fd < 0 || ftruncate(c->own_fd, off_t(bytes)) != 0) return SYSTEM_ERROR;
		c->own_map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->own_fd, 0);
		if (c->own_map == MAP_FAILED) { c->own_map = nullptr; c->own_bytes = 0; return SYSTEM_ERROR; }
		c->own_bytes = bytes;
	}
	// every copy is enqueued on the CALLER'S stream (a non-blocking stream is not ordered against the null stream a plain hipMemcpy uses: the unpack kernel
	// the product enqueues next would race a device-to-device copy made there) and waited for before the ranks meet
	if (hipMemcpyAsync(c->own_map, sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return UNHANDLED_HIP_ERROR;
	if (!barrier(c)) return SYSTEM_ERROR;                                                  // every rank's chunk is in its file
	void * maps[64] = { }; if (c->world > 64) return INVALID_USAGE;
	int result = OK;
	for (int r = 0; r < c->world && result == OK; r++) {
		char * slot = (char *)recvbuff + size_t(r) * bytes;
		if (r == c->rank) { if (hipMemcpyAsync(slot, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) result = UNHANDLED_HIP_ERROR; continue; }
		int fd = shm_open(data_name(c->token, r).c_str(), O_RDONLY, 0600);
		struct stat st;
		if (fd < 0 || fstat(fd, &st) != 0 || size_t(st.st_size) != bytes) { if (fd >= 0) close(fd); result = INVALID_USAGE; break; }   // (ranks must gather the same count)
		maps[r] = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
		close(fd);
		if (maps[r] == MAP_FAILED) { maps[r] = nullptr; result = SYSTEM_ERROR; break; }
		if (hipMemcpyAsync(slot, maps[r], bytes, hipMemcpyHostToDevice, stream) != hipSuccess) result = UNHANDLED_HIP_ERROR;
	}
	if (hipStreamSynchronize(stream) != hipSuccess && result == OK) result = UNHANDLED_HIP_ERROR;
	for (int r = 0; r < c->world; r++) if (maps[r]) munmap(maps[r], bytes);
	if (!barrier(c) && result == OK) result = SYSTEM_ERROR;                                // nobody rewrites its file before everyone has read it
	return result;
}

int ncclGroupStart() { return OK; }
int ncclGroupEnd() { return OK; }
const char * ncclGetErrorString(int result) {
	switch (result) {
		case OK: return "no error";
		case UNHANDLED_HIP_ERROR: return "loopback stand-in: a HIP call failed";
		case SYSTEM_ERROR: return "loopback stand-in: shared memory / a peer did not arrive within a minute";
		case INVALID_ARGUMENT: return "loopback stand-in: invalid argument (float32 only)";
		case INVALID_USAGE: return "loopback stand-in: invalid usage (ranks gathered different counts, or ncclCommInitAll)";
		default: return "loopback stand-in: internal error";
	}
}

}   // extern "C"
