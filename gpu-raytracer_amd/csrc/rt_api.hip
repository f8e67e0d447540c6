// rt_api.hip -- implementation of the C ABI in include/gpu_raytracer_amd.h.
//
// One rt_context owns one HIP device: a stream, every device buffer, and the RtParams block
// handed to the kernels by value. It plays the role of the reference's Device/ layer plus the
// device half of Integrator/Pathtracer (buffer ownership, `buffer_sizes` handling, the
// wavefront launch loop of Pathtracer::render, Pathtracer.cpp:738-855).
#include "rt_tlas_build.h"
#include <dlfcn.h>
#include "rt_types.h"
#include "rt_tlas_build.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

static thread_local std::string g_global_error;

// HIP multiplexes its streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and the variable is read when
// the HIP runtime initialises, i.e. at the first HIP call of the PROCESS. The per-submission scheduler keeps up to 2 x
// (submissions in flight) streams busy and loses their overlap silently when they share queues
// (profiles/r01_small_range_concurrency.txt). A library has no business editing its host's environment (and cannot know
// whether HIP is already up -- PyTorch imported first, say): the applications that ship with it set the variable before
// their first HIP call (bench.py, the Python front end, the command-line renderer), INTEGRATION.md tells embedders to, and
// a context says so once when it is asked for more concurrent streams than the process has queues for.
static void warn_if_streams_share_queues(int streams_needed) {
	static bool warned = false;
	if (warned) return;
	const char * value = getenv("GPU_MAX_HW_QUEUES");
	const int queues = value ? atoi(value) : 4;
	if (queues >= streams_needed) return;
	warned = true;
	fprintf(stderr, "[grt] %d streams in flight but GPU_MAX_HW_QUEUES=%s: HIP will multiplex them onto %d hardware queues and serialise their kernels; "
	                "export GPU_MAX_HW_QUEUES=24 before the process makes its first HIP call\n", streams_needed, value ? value : "(unset, 4)", queues);
}

// Everything one sample per pixel owns while it is in flight. Up to RT_MAX_SAMPLE_SLOTS samples
// are rendered concurrently (rt_set_samples_in_flight): consecutive rt_render_sample calls take the
// slots round-robin, each on its own stream, and only the accumulate step is ordered between them.
// Why: a wavefront pass is a chain of ~40 launches whose deep bounces are too small to fill 256 CUs
// and whose persistent trace launches each end in a tail; a second sample's kernels fill those holes.
#define RT_MAX_SAMPLE_SLOTS 8
#define RT_MAX_BATCH_SAMPLES 16
struct SampleSlot {
	bool created = false;
	hipStream_t stream = nullptr;      // the sample's launch chain
	hipStream_t side   = nullptr;      // shadow rays of bounce b, concurrent with the closest-hit trace of bounce b+1
	hipEvent_t ev_shaded = nullptr, ev_shadowed = nullptr, ev_done = nullptr, ev_frame_start = nullptr, ev_frame_end = nullptr;
	RtTraceBuffer trace[2]; RtMaterialBuffer material[4]; RtShadowBuffer shadow;
	bool queues_allocated = false; size_t queue_capacity = 0;
	RtBufferSizes * sizes = nullptr;
	int * xcd_counters = nullptr;
	void * spill[2] = { nullptr, nullptr };  // traversal stack spill of the closest-hit / shadow launch
	int * counter_totals = nullptr;          // 6 x RT_MAX_BOUNCES ints accumulated over batches
	RtBufferSizes * pinned_counters = nullptr;
	void * aov_framebuffer[RT_AOV_COUNT] = { };  // slots 1..: per-sample frame buffers (slot 0 uses ctx->aov_buffers[i][0])
	int aov_samples = 1;                         // samples per batch the frame buffers of this slot are sized for
	// SVGF g-buffers (normal+depth, mesh+triangle id, previous screen position) are written by the
	// bounce-0 kernels of a frame and read by its filter stage: one set per slot lets frame n+1 be
	// traced while frame n is filtered. Pixels that miss all geometry keep the value of the last
	// frame that hit something (the reference never clears them), so a frame starts from a copy of
	// its predecessor's set, taken as soon as that frame's bounce-0 shading is done (ev_gbuffers).
	void * gbuffers[3] = { };                    // slots 1..; slot 0 uses ctx->svgf_buffers[0..2]
	hipEvent_t ev_gbuffers = nullptr;
};

// Per-frame scene data (the TLAS and the five per-instance tables, rebuilt by Integrator::build_tlas for
// every frame of an animated scene) lives in a ring of versions: an upload fills the next version
// through pinned staging with an asynchronous copy, the kernels of later samples get its address (the
// parameter block is passed by value), samples already in flight keep reading theirs. No upload of this
// kind drains the pipeline; a version is only waited for when the ring wraps around onto a sample
// that still uses it.
#define RT_SCENE_VERSIONS 12
struct SceneRing {
	void * device[RT_SCENE_VERSIONS] = { };
	void * pinned[RT_SCENE_VERSIONS] = { };
	hipEvent_t copied[RT_SCENE_VERSIONS] = { };
	size_t capacity = 0;   // bytes allocated per version (grows only)
	size_t bytes = 0;      // bytes of the current version
	int current = -1;
	// Sample slots that have submitted work reading a version since it was last written. (A slot used to remember only the
	// version of its LAST submission: with the host several frames ahead of the device, an older submission still queued
	// on the same slot lost its claim and the ring wrapped around onto the version it was about to read -- found with 1 500
	// moving instances and 48 frames, tools/animation_bench.py: a TLAS overwritten under a running traversal.)
	unsigned users[RT_SCENE_VERSIONS] = { };
	hipEvent_t last_use[RT_SCENE_VERSIONS][RT_MAX_SAMPLE_SLOTS] = { }; // end of the slot's latest submission that reads the version
};

// ---- merged wavefront (RT_SCHEDULER_MERGED; the idea is described at RtStreamSlot in rt_types.h) -------------------
// Host side: ONE stream, one set of queues sized for `capacity` rays. A submission (one rt_render_samples call) gets a
// run of sample slots, generates its primary rays into the current trace queue and advances the wavefront by one
// iteration; it is complete -- accumulated into the shared accumulators, in submission order -- after the iteration in
// which it reaches its last bounce. What bounds the queues: every path in flight occupies at most one entry of a
// queue, paths only die, so (wavefront size reported by an earlier iteration) + (rays generated since) bounds the
// entries any queue can receive; the host stays RT_STREAM_RUN_AHEAD iterations ahead of the device at most, reads the
// reported sizes from pinned memory without blocking, and runs iterations without new samples while a new
// submission would not fit.
// Small submissions (the tiles of one rank of a multi-GPU split: 1/8 of a frame) would make small iterations again --
// launches that do not fill the machine, fixed costs per iteration that no longer disappear behind the rays. When the
// application pipelines frames (rt_set_frame_pipelining) a submission therefore generates its rays at once but the
// iteration is only enqueued when RT_STREAM_BATCH_PATHS paths have been generated for it (or RT_STREAM_MAX_BATCH
// submissions), so the iterations of a 1/8 split carry 8 frames and are as large as those of a whole frame. Anything that
// needs progress -- rt_advance, every call that flushes, a change of camera -- enqueues the iteration with what is there.
#define RT_STREAM_BATCH_PATHS     (1920 * 1080 * 4)
#define RT_STREAM_MAX_BATCH       8
#define RT_STREAM_PROGRESS_RING   64
#define RT_STREAM_RUN_AHEAD       4
#define RT_STREAM_TABLE_SNAPSHOTS 8
#define RT_STREAM_HISTORY_ROWS    4096
#define RT_STREAM_STATS_ROW       (RT_STAT_KINDS * RT_MAX_BOUNCES)   // ints per submission

struct StreamSubmission {
	int first_sample, sample_count, slot_base, ring, birth, last, paths;
	int range_offset, range_count, tile_pixels, tile_first, tile_stride;
};

struct PathStream {
	bool created = false;
	hipStream_t stream = nullptr;
	hipEvent_t ev_idle = nullptr;         // after the last completion enqueued so far: what main-stream consumers wait for
	RtTraceBuffer trace[2]; RtMaterialBuffer material[4]; RtShadowBuffer shadow;
	bool queues_allocated = false; size_t capacity = 0;
	RtStreamControl * control = nullptr;
	RtStreamTable * table_device = nullptr;
	RtStreamTable table_host;             // what the device table will hold once the copies enqueued so far have run
	RtStreamTable * table_staging = nullptr; hipEvent_t table_copied[RT_STREAM_TABLE_SNAPSHOTS] = { }; int table_next = 0;
	void * spill = nullptr;
	void * aov_framebuffer[RT_AOV_COUNT] = { }; int frame_slots = 0;   // per-sample frames, one per sample slot
	void * gbuffers[3] = { };             // SVGF: one g-buffer set (float4, int2, float2 per pixel) per sample slot
	int last_gbuffer_slot = -1;           // the set of the frame submitted last, if that was a frame of this wavefront
	bool slot_used[RT_STREAM_SAMPLE_SLOTS] = { };
	int next_slot = 0, next_ring = 0;
	int iteration = 0;                    // the next iteration to enqueue
	int base_iteration = 0;               // nothing generated before it is still in flight
	std::deque<StreamSubmission> in_flight;
	int pending = 0; long long pending_paths = 0;   // the newest submissions: rays generated, iteration not enqueued yet
	// ... their slot-table entries and statistics rows reach the device ONCE, with the iteration that first needs them (one copy and the advance launch instead
	// of a copy and a fill per submission: a rank of an 8-GPU split spent 0.1 ms of its 5.4 ms burst on five such pairs, profiles/r05_rank_timeline.txt)
	bool table_dirty = false; int reset_ring_first = 0, reset_ring_count = 0;
	int * progress = nullptr;             // pinned [RING][2] = { iteration, wavefront size }, written by kernel_stream_advance
	hipEvent_t iteration_done[RT_STREAM_PROGRESS_RING] = { };
	int generated[RT_STREAM_PROGRESS_RING] = { };
	int known_iteration = -1; long long known_size = 0;
	unsigned long long submissions_completed = 0;
	int * stats_host = nullptr;           // pinned [RT_STREAM_SUBMISSIONS][RT_STREAM_STATS_ROW]
	hipEvent_t ev_begin[RT_STREAM_SUBMISSIONS] = { }, ev_end[RT_STREAM_SUBMISSIONS] = { };
	int last_completed_ring = -1;
};

struct rt_context {
	int device = 0;
	hipStream_t stream = nullptr;      // "main": uploads, read-backs, pack/unpack, kernel-level entry points
	hipEvent_t ev_main = nullptr, ev_interop = nullptr;
	std::string error;

	SampleSlot slots[RT_MAX_SAMPLE_SLOTS];
	// (Two such wavefronts taking the submissions in turns, their traversal launches serialised by an event chain so that
	// one pipeline's sort / shade kernels would run beside the other's traversal, were built and measured: 3.23 ms per step
	// against 3.02 ms with one -- the persistent traversal launch holds every wave slot of the machine until its queue is
	// drained, nothing can start beside it. Removed; profiles/r02_two_pipelines_*.)
	PathStream path_stream;
	unsigned long long * stream_history = nullptr; int stream_history_rows = 0;   // pinned [ROWS][10]: trace statistics after each traversal launch
	int scheduler = RT_SCHEDULER_MERGED;
	bool last_render_merged = false;
	bool defer_filter = false;         // rt_render_sample_unfiltered: an SVGF frame stops before its filter stage (tile split)
	bool frame_pipelining = false;     // rt_pack_pixels / rt_unpack_pixels follow the completed submissions only (rt_set_frame_pipelining)
	long long stream_batch_paths = 0;  // paths the submissions of one iteration may bring (rt_set_stream_batch); 0: RT_STREAM_BATCH_PATHS
	int samples_in_flight = 3;
	bool overlap_shadows = true;
	unsigned render_counter = 0;
	int last_slot = -1;

	RtParams params;               // zero-initialised in rt_create
	std::vector<void *> owned;     // every hipMalloc'd pointer, freed in rt_destroy

	// named allocations that get replaced on re-upload
	void * triangles = nullptr, * triangle_positions = nullptr, * bvh8_nodes = nullptr, * bvh2_nodes = nullptr, * bvh4_nodes = nullptr;
	size_t tlas_node_bytes = 80;        // what the current TLAS version was uploaded as (80 CWBVH, 32 binary, 128 4-wide)
	int lowest_blas_root = 0x7fffffff;  // over the instances uploaded last: the node slots below it are free for the TLAS copy of the merged wavefront
	unsigned long long tlas_version = 0, tlas_version_in_nodes = ~0ull;   // the merged wavefront traces a copy of the TLAS inside the BLAS node array (stream_sync_tlas)
	bool expand_bc1 = true;   // rt_set_texture_expansion: BC1 textures are decoded once, at upload (rt_types.h: RT_TEXTURE_BC1_EXPANDED)
	size_t texture_bytes = 0; // what rt_upload_textures holds on the device
	std::vector<float> build_boxes; size_t build_boxes_first = 0;   // rt_set_build_boxes: consumed by the next rt_build_geometry
	size_t bvh4_node_count = 0;
	size_t bvh8_node_count = 0, bvh2_node_count = 0, triangle_count = 0;
	size_t mesh_count = 0;
	SceneRing tlas_ring, instance_ring, light_ring;
	int * device_tlas_order = nullptr, * device_tlas_node_count = nullptr; // current TLAS built by rt_build_tlas (else null)
	hipEvent_t ev_scene = nullptr;  // the last asynchronous scene upload on the main stream
	void * material_types = nullptr, * materials = nullptr, * media = nullptr;
	bool has_material[4] = { false, false, false, false };
	bool has_lights = false;
	void * texture_table = nullptr; std::vector<void *> texture_data;
	void * pmj = nullptr, * blue_noise = nullptr;
	void * sky = nullptr;
	void * luts[6] = { }; bool luts_ready = false;
	int bvh_width = 8;

	// frame resources
	void * aov_buffers[RT_AOV_COUNT][2] = { };
	void * final_image = nullptr;
	void * svgf_buffers[16] = { }; bool svgf_allocated = false;   // [15]: RtParams::svgf_young_pixels
	size_t frame_pixels = 0; // pitch * height

	// frame exchange of the tile split (rt_comm_*): this context's rank in a group of `world` contexts, each on its own GPU
	// (RCCL communicator) or, for tests on one GPU, several in one process (peer copies)
	struct FrameExchange {
		int rank = 0, world = 1;
		void * comm = nullptr;                       // ncclComm_t
		std::vector<rt_context *> peers;             // in-process transport: all contexts of the group, by rank
		float4 * packed = nullptr, * gathered = nullptr; size_t packed_pixels = 0;   // per pixel `channels` float4
		hipEvent_t ev_packed = nullptr, ev_copied = nullptr;
	} exchange;

	int * explicit_retired = nullptr;
	int * pixel_query_out = nullptr;   // device { mesh_id, triangle_id }
	int batch_size_request = 0;              // 0 = whole frame (288 GB of HBM: no reason to cut a frame into pieces)
	int pixel_offset = 0, pixel_count = -1;  // -1 = whole frame

	rt_counters last_counters;
	bool profiling = false;          // mode 1: per-stage events, one sample at a time
	bool launch_timing = false;      // mode 2: events around every traversal launch, concurrency untouched
	bool launch_timing_all = false;  // mode 3: ... and around every other launch of the merged wavefront
	bool time_this_sample = false;
	std::vector<hipEvent_t> span_events; std::vector<int> span_kinds; size_t span_used = 0; // mode 2: [begin, end] pairs
	bool trace_statistics = false;
	unsigned long long * trace_stats = nullptr;    // device, 10 x u64
	unsigned long long host_trace_stats[10] = { };
	std::vector<hipEvent_t> stage_events; std::vector<int> stage_kinds; size_t stage_used = 0;
};

static int fail(rt_context * ctx, int status, const char * fmt, ...) {
	char buffer[512];
	va_list args; va_start(args, fmt); vsnprintf(buffer, sizeof(buffer), fmt, args); va_end(args);
	if (ctx) ctx->error = buffer;
	g_global_error = buffer;
	return status;
}

#define RT_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, RT_ERROR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define RT_REQUIRE(ctx, cond, msg) do { if (!(cond)) return fail(ctx, RT_ERROR_INVALID_ARG, "%s", msg); } while (0)

static int device_alloc(rt_context * ctx, void ** out, size_t bytes) {
	if (bytes == 0) bytes = 16;
	RT_HIP(ctx, hipMalloc(out, bytes));
	ctx->owned.push_back(*out);
	return RT_OK;
}
static void device_free(rt_context * ctx, void * p) {
	if (!p) return;
	for (size_t i = 0; i < ctx->owned.size(); i++) if (ctx->owned[i] == p) { ctx->owned[i] = ctx->owned.back(); ctx->owned.pop_back(); break; }
	(void)hipFree(p);
}
// (re)allocate + synchronous upload
static hipError_t quiesce(rt_context * ctx);
static int upload(rt_context * ctx, void ** slot, const void * src, size_t bytes) {
	RT_HIP(ctx, quiesce(ctx)); // samples in flight may still read the old buffer
	device_free(ctx, *slot);
	*slot = nullptr;
	int s = device_alloc(ctx, slot, bytes);
	if (s != RT_OK) return s;
	if (bytes && src) RT_HIP(ctx, hipMemcpy(*slot, src, bytes, hipMemcpyHostToDevice));
	return RT_OK;
}

// Wait for everything the context has in flight: the merged wavefront (run to completion first), every sample
// slot (and its side stream), then main.
static hipError_t stream_flush(rt_context * ctx);
static int stream_launch_pending(rt_context * ctx);
static void stream_destroy(rt_context * ctx);
static void stream_release_frames(rt_context * ctx);
static hipError_t quiesce(rt_context * ctx) {
	if (ctx->path_stream.created) {
		hipError_t e = stream_flush(ctx);                            if (e != hipSuccess) return e;
		e = hipStreamSynchronize(ctx->path_stream.stream);           if (e != hipSuccess) return e;
	}
	for (SampleSlot & slot : ctx->slots) if (slot.created) {
		hipError_t e = hipStreamSynchronize(slot.side);   if (e != hipSuccess) return e;
		e = hipStreamSynchronize(slot.stream);            if (e != hipSuccess) return e;
	}
	return hipStreamSynchronize(ctx->stream);
}

// Main-stream work that reads or writes frame results is ordered after every sample already submitted.
static hipError_t main_waits_for_samples(rt_context * ctx) {
	if (ctx->path_stream.created) {
		if (!ctx->frame_pipelining) { hipError_t e = stream_flush(ctx); if (e != hipSuccess) return e; }
		hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->path_stream.ev_idle, 0); if (e != hipSuccess) return e;
	}
	for (SampleSlot & slot : ctx->slots) if (slot.created) {
		hipError_t e = hipStreamWaitEvent(ctx->stream, slot.ev_done, 0); if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

static int ensure_slot(rt_context * ctx, int index) {
	SampleSlot & slot = ctx->slots[index];
	if (slot.created) return RT_OK;
	if (index > 0) warn_if_streams_share_queues(2 * (index + 1) + 2);   // two streams per slot, the main stream, the merged wavefront's
	memset(slot.trace, 0, sizeof(slot.trace)); memset(slot.material, 0, sizeof(slot.material)); memset(&slot.shadow, 0, sizeof(slot.shadow));
	RT_HIP(ctx, hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
	RT_HIP(ctx, hipStreamCreateWithFlags(&slot.side,   hipStreamNonBlocking));
	RT_HIP(ctx, hipEventCreateWithFlags(&slot.ev_shaded,   hipEventDisableTiming));
	RT_HIP(ctx, hipEventCreateWithFlags(&slot.ev_shadowed, hipEventDisableTiming));
	RT_HIP(ctx, hipEventCreateWithFlags(&slot.ev_done,     hipEventDisableTiming));
	RT_HIP(ctx, hipEventCreate(&slot.ev_frame_start));
	RT_HIP(ctx, hipEventCreate(&slot.ev_frame_end));
	RT_HIP(ctx, hipEventCreateWithFlags(&slot.ev_gbuffers, hipEventDisableTiming));
	RT_HIP(ctx, hipEventRecord(slot.ev_done, slot.stream)); // so that waiting on a never-used slot is a no-op
	RT_HIP(ctx, hipEventRecord(slot.ev_gbuffers, slot.stream));
	int s = device_alloc(ctx, (void **)&slot.sizes, sizeof(RtBufferSizes)); if (s) return s;
	s = device_alloc(ctx, (void **)&slot.counter_totals, 6 * RT_MAX_BOUNCES * sizeof(int)); if (s) return s;
	s = device_alloc(ctx, (void **)&slot.xcd_counters, RT_MAX_BOUNCES * 2 * 8 * sizeof(int)); if (s) return s;
	// spill area for traversal stacks deeper than the LDS part: 24 entries x 8 B x (256 CUs x 8 workgroups x 256 lanes),
	// one per launch that can be resident at the same time
	for (int k = 0; k < 2; k++) { s = device_alloc(ctx, &slot.spill[k], size_t(24) * 8 * 256 * 8 * 256); if (s) return s; }
	RT_HIP(ctx, hipHostMalloc((void **)&slot.pinned_counters, sizeof(RtBufferSizes)));
	memset(slot.pinned_counters, 0, sizeof(RtBufferSizes));
	size_t bytes = ctx->frame_pixels * 16;
	if (index > 0 && bytes) for (int i = 0; i < RT_AOV_COUNT; i++) if (ctx->aov_buffers[i][0]) {
		s = device_alloc(ctx, &slot.aov_framebuffer[i], bytes); if (s) return s;
		RT_HIP(ctx, hipMemset(slot.aov_framebuffer[i], 0, bytes));
	}
	if (index > 0 && ctx->svgf_allocated) {
		const size_t elem[3] = { 16, 8, 8 };
		for (int i = 0; i < 3; i++) {
			s = device_alloc(ctx, &slot.gbuffers[i], ctx->frame_pixels * elem[i]); if (s) return s;
			RT_HIP(ctx, hipMemset(slot.gbuffers[i], 0, ctx->frame_pixels * elem[i]));
		}
	}
	slot.created = true;
	return RT_OK;
}

// The kernels get RtParams by value: the context's block with one slot's per-sample pointers patched in.
static RtParams slot_params(const rt_context * ctx, const SampleSlot & slot, int index) {
	RtParams p = ctx->params;
	memcpy(p.trace, slot.trace, sizeof(p.trace)); memcpy(p.material, slot.material, sizeof(p.material)); p.shadow = slot.shadow;
	p.sizes = slot.sizes; p.xcd_counters = slot.xcd_counters; p.stack_spill = (uint2 *)slot.spill[0];
	if (index > 0) for (int i = 0; i < RT_AOV_COUNT; i++) p.aovs[i].framebuffer = (float4 *)slot.aov_framebuffer[i];
	if (index > 0 && slot.gbuffers[0]) {
		p.gbuffer_normal_and_depth        = (float4 *)slot.gbuffers[0];
		p.gbuffer_mesh_id_and_triangle_id = (int2   *)slot.gbuffers[1];
		p.gbuffer_screen_position_prev    = (float2 *)slot.gbuffers[2];
	}
	return p;
}

// Next version of a scene ring, sized `bytes`: returns its pinned staging buffer for the caller to fill;
// ring_commit() then starts the copy. `which` selects the slot field that remembers the version in use.
static int ring_begin(rt_context * ctx, SceneRing & ring, size_t bytes, void ** staging) {
	if (bytes == 0) bytes = 16;
	// a launch of the merged wavefront traces the rays of every submission in flight against ONE scene version
	if (ctx->path_stream.created) {
		RT_HIP(ctx, stream_flush(ctx));
		// ... and the copy into the next version (main stream) runs after everything the wavefront has enqueued: its kernels read
		// ring versions too, and nothing else orders the two streams (the flush only ENQUEUES the remaining iterations). Without
		// this wait a version was safe from being overwritten under a running iteration only because RT_SCENE_VERSIONS (12)
		// exceeds what RT_STREAM_RUN_AHEAD (4) lets the host get ahead.
		RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->path_stream.ev_idle, 0));
	}
	if (bytes > ring.capacity) { // first use, or the scene outgrew the ring: start over (the only case that drains)
		RT_HIP(ctx, quiesce(ctx));
		size_t capacity = bytes + bytes / 2 + 256; // a TLAS changes its node count a little from frame to frame
		for (int v = 0; v < RT_SCENE_VERSIONS; v++) {
			device_free(ctx, ring.device[v]); ring.device[v] = nullptr;
			if (ring.pinned[v]) { (void)hipHostFree(ring.pinned[v]); ring.pinned[v] = nullptr; }
			int s = device_alloc(ctx, &ring.device[v], capacity); if (s) return s;
			RT_HIP(ctx, hipHostMalloc(&ring.pinned[v], capacity));
			if (!ring.copied[v]) RT_HIP(ctx, hipEventCreateWithFlags(&ring.copied[v], hipEventDisableTiming));
		}
		ring.capacity = capacity;
		ring.current = -1;
		for (unsigned & users : ring.users) users = 0;
	}
	ring.bytes = bytes;
	int v = (ring.current + 1) % RT_SCENE_VERSIONS;
	// wait for the submissions that read version v -- for them only: a slot's ev_done is re-recorded by every later
	// submission, waiting on it would stall the host behind the frame it has just submitted
	for (int k = 0; k < RT_MAX_SAMPLE_SLOTS; k++) if ((ring.users[v] >> k) & 1u) RT_HIP(ctx, hipEventSynchronize(ring.last_use[v][k]));
	ring.users[v] = 0;
	if (ring.current >= 0) RT_HIP(ctx, hipEventSynchronize(ring.copied[v])); // its previous staging copy (recorded when it was last filled)
	*staging = ring.pinned[v];
	ring.current = v;
	return RT_OK;
}

static int ring_commit(rt_context * ctx, SceneRing & ring) {
	int v = ring.current;
	RT_HIP(ctx, hipMemcpyAsync(ring.device[v], ring.pinned[v], ring.bytes, hipMemcpyHostToDevice, ctx->stream));
	RT_HIP(ctx, hipEventRecord(ring.copied[v], ctx->stream));
	RT_HIP(ctx, hipEventRecord(ctx->ev_scene, ctx->stream));
	return RT_OK;
}

// A submission on sample slot `slot_index` (stream `st`, everything enqueued) read the current version of every scene ring
static int mark_scene_versions_in_use(rt_context * ctx, int slot_index, hipStream_t st) {
	for (SceneRing * ring : { &ctx->tlas_ring, &ctx->instance_ring, &ctx->light_ring }) {
		if (ring->current < 0) continue;
		hipEvent_t & e = ring->last_use[ring->current][slot_index];
		if (!e) RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
		RT_HIP(ctx, hipEventRecord(e, st));
		ring->users[ring->current] |= 1u << slot_index;
	}
	return RT_OK;
}

// the node array of the selected BVH type has been uploaded
static bool bvh_nodes_present(const rt_context * ctx) {
	const RtParams & p = ctx->params;
	return ctx->bvh_width == 8 ? p.bvh8_nodes != nullptr : (ctx->bvh_width == 4 ? p.bvh4_nodes != nullptr : p.bvh2_nodes != nullptr);
}

enum { STAGE_GENERATE = 0, STAGE_TRACE, STAGE_SORT, STAGE_SHADE, STAGE_SHADOW, STAGE_POST, STAGE_END };
// what a [begin, end] event pair of rt_set_profiling(ctx, 2 / 3) brackets (the `kind` of rt_get_launch_timings)
enum { SPAN_TRACE = RT_TIMING_TRACE, SPAN_SHADOW = RT_TIMING_SHADOW, SPAN_SORT = RT_TIMING_SORT, SPAN_GENERATE = RT_TIMING_GENERATE, SPAN_ACCUMULATE = RT_TIMING_ACCUMULATE,
       SPAN_MATERIAL = RT_TIMING_MATERIAL_0, SPAN_SVGF = RT_TIMING_SVGF_REPROJECT };

extern "C" {

const char * rt_version(void) { return "gpu-raytracer_amd 0.3 (gfx950, HIP; ABI 4)"; }
int rt_abi_version(void) { return RT_ABI_VERSION; }

const char * rt_last_error(const rt_context * ctx) { return ctx ? ctx->error.c_str() : g_global_error.c_str(); }

int rt_create(int device_ordinal, rt_context ** out_ctx) {
	if (!out_ctx) return fail(nullptr, RT_ERROR_INVALID_ARG, "rt_create: out_ctx is NULL");
	*out_ctx = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count == 0) return fail(nullptr, RT_ERROR_NO_DEVICE, "rt_create: no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	if (device_ordinal < 0 || device_ordinal >= count) return fail(nullptr, RT_ERROR_NO_DEVICE, "rt_create: device ordinal %d out of range [0,%d)", device_ordinal, count);
	if (hipSetDevice(device_ordinal) != hipSuccess) return fail(nullptr, RT_ERROR_NO_DEVICE, "rt_create: hipSetDevice(%d) failed", device_ordinal);

	rt_context * ctx = new rt_context();
	ctx->device = device_ordinal;
	memset(&ctx->params, 0, sizeof(ctx->params));
	ctx->params.entry_tlas_stack_size = RT_INVALID;
	ctx->params.svgf_tiles = 1;
	ctx->params.skip_behind_hit = 1;
	if (const char * e = getenv("GRT_SKIP_BEHIND_HIT")) ctx->params.skip_behind_hit = atoi(e) != 0;   // (A / B runs of one command: bench.py, tools/)
	memset(&ctx->last_counters, 0, sizeof(ctx->last_counters));
	RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
	RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
	RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_interop, hipEventDisableTiming));
	RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_scene, hipEventDisableTiming));
	RT_HIP(ctx, hipEventRecord(ctx->ev_scene, ctx->stream));
	if (const char * e = getenv("GRT_SAMPLES_IN_FLIGHT")) { int n = atoi(e); if (n >= 1 && n <= RT_MAX_SAMPLE_SLOTS) ctx->samples_in_flight = n; }
	if (const char * e = getenv("GRT_OVERLAP_SHADOWS")) ctx->overlap_shadows = atoi(e) != 0;
	if (const char * e = getenv("GRT_SCHEDULER")) ctx->scheduler = strcmp(e, "slots") == 0 ? RT_SCHEDULER_SLOTS : RT_SCHEDULER_MERGED;

	int s = ensure_slot(ctx, 0); if (s) return s;
	s = device_alloc(ctx, (void **)&ctx->explicit_retired, 8 * sizeof(int)); if (s) return s;
	s = device_alloc(ctx, (void **)&ctx->pixel_query_out, 2 * sizeof(int)); if (s) return s;
	RT_HIP(ctx, hipMemset(ctx->pixel_query_out, 0xff, 2 * sizeof(int)));
	ctx->params.pixel_query_out = ctx->pixel_query_out;
	ctx->params.pixel_query_pixel = -1;
	// the kernel-level entry points run on the main stream with slot 0's buffers (after quiesce())
	ctx->params.sizes        = ctx->slots[0].sizes;
	ctx->params.xcd_counters = ctx->slots[0].xcd_counters;
	ctx->params.stack_spill  = (uint2 *)ctx->slots[0].spill[0];

	ctx->params.frame_pixels = 1u << 30; ctx->params.frame_pixels_magic = 5; ctx->params.batch_samples = 1; // until rt_resize
	ctx->params.bvh_width = 8;
	// default config = reference defaults (Common.h:39-67)
	rt_gpu_config c = { RT_FILTER_GAUSSIAN, 1u << RT_AOV_RADIANCE, 10, 1, 1, 1, 1, 0, 1, 1, 0.1f, 0.1f, 6, 4.0f, 16.0f, 10.0f };
	ctx->params.config = c;
	*out_ctx = ctx;
	return RT_OK;
}

void rt_destroy(rt_context * ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)quiesce(ctx);
	stream_destroy(ctx);
	(void)rt_comm_destroy(ctx);
	for (void * p : ctx->owned) (void)hipFree(p);
	for (hipEvent_t e : ctx->stage_events) (void)hipEventDestroy(e);
	for (hipEvent_t e : ctx->span_events) (void)hipEventDestroy(e);
	for (SampleSlot & slot : ctx->slots) if (slot.stream) {
		if (slot.pinned_counters) (void)hipHostFree(slot.pinned_counters);
		for (hipEvent_t e : { slot.ev_shaded, slot.ev_shadowed, slot.ev_done, slot.ev_gbuffers, slot.ev_frame_start, slot.ev_frame_end }) if (e) (void)hipEventDestroy(e);
		(void)hipStreamDestroy(slot.side);
		(void)hipStreamDestroy(slot.stream);
	}
	(void)hipEventDestroy(ctx->ev_main);
	(void)hipEventDestroy(ctx->ev_interop);
	(void)hipEventDestroy(ctx->ev_scene);
	for (SceneRing * ring : { &ctx->tlas_ring, &ctx->instance_ring, &ctx->light_ring }) for (int v = 0; v < RT_SCENE_VERSIONS; v++) {
		if (ring->pinned[v]) (void)hipHostFree(ring->pinned[v]);
		if (ring->copied[v]) (void)hipEventDestroy(ring->copied[v]);
		for (hipEvent_t e : ring->last_use[v]) if (e) (void)hipEventDestroy(e);
	}
	(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

// ---- scene upload ----------------------------------------------------------------------------------

// traversal copy of the triangles: positions only, 48 B stride (first 36 B of each 96 B triangle + 12 B pad)
static int upload_triangle_positions(rt_context * ctx, const void * triangles, size_t triangle_count) {
	std::vector<float> positions(triangle_count * 12, 0.0f);
	const float * src = (const float *)triangles;
	for (size_t t = 0; t < triangle_count; t++) memcpy(&positions[t * 12], src + t * 24, 36);
	int s = upload(ctx, &ctx->triangle_positions, positions.data(), triangle_count * 48); if (s) return s;
	ctx->params.triangle_positions = (const float4 *)ctx->triangle_positions;
	ctx->params.has_triangle_aliases = 0; ctx->params.entry_tlas_stack_size = RT_INVALID;
	return RT_OK;
}

// What the flattened scene's engine (kernel_trace_stream_bvh8_flat*) can address: a node's byte offset is v_mul_u32_u24(index, 80) -- a 24-bit multiply, exact for
// fewer than 2^24 nodes (1.34 GB of them: the 4 GiB of the 32-bit offset is never the limit for nodes) -- and a triangle's a 32-bit index * 48. Beyond either limit
// the general engine walks the scene (64-bit addresses). (Round 5 checked the 4 GiB only: advisor finding.)
int rt_geometry_fits_flat_engine(size_t node_count, size_t triangle_count) {
	return node_count < (size_t(1) << 24) && triangle_count * 48 < (1ull << 32) ? 1 : 0;
}

int rt_upload_geometry(rt_context * ctx, const void * triangles, size_t triangle_count, const void * bvh8_nodes, size_t node_count) {
	RT_REQUIRE(ctx, ctx && triangles && bvh8_nodes, "rt_upload_geometry: NULL argument");
	ctx->build_boxes.clear(); ctx->build_boxes_first = 0;   // (boxes set for a build that never came do not wait for another geometry's)
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->triangles, triangles, triangle_count * 96); if (s) return s;
	s = upload(ctx, &ctx->bvh8_nodes, bvh8_nodes, node_count * 80); if (s) return s;
	s = upload_triangle_positions(ctx, triangles, triangle_count); if (s) return s;
	ctx->triangle_count = triangle_count; ctx->bvh8_node_count = node_count;
	ctx->tlas_version_in_nodes = ~0ull;
	ctx->params.triangles  = (const float4 *)ctx->triangles;
	ctx->params.bvh8_nodes = (const float4 *)ctx->bvh8_nodes;
	ctx->params.geometry_below_4gib = rt_geometry_fits_flat_engine(node_count, triangle_count);
	return RT_OK;
}

// Static geometry flattened into one BLAS: its triangles are COPIES of triangles that other BLASes own, and a hit on a copy is
// reported as the instance and the triangle it was copied from (kernels_trace.hip translates once per ray, when the ray is done).
// The two names are the last 8 bytes of the 48-byte position record (36 B of positions + 12 B of padding), so the translation
// costs one load from a line the triangle test has touched.
} // extern "C"
__global__ void kernel_name_triangles(float4 * positions, const int2 * names, int count) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	float4 last = positions[size_t(i) * 3 + 2];
	last.z = __int_as_float(names[i].x); last.w = __int_as_float(names[i].y);
	positions[size_t(i) * 3 + 2] = last;
}
extern "C" {
int rt_upload_triangle_aliases(rt_context * ctx, const int32_t * mesh_ids, const int32_t * triangle_ids) {
	RT_REQUIRE(ctx, ctx && ctx->triangle_positions, "rt_upload_triangle_aliases: no geometry uploaded");
	(void)hipSetDevice(ctx->device);
	if (!mesh_ids || !triangle_ids) { RT_HIP(ctx, quiesce(ctx)); ctx->params.has_triangle_aliases = 0; return RT_OK; }
	size_t count = ctx->triangle_count;
	std::vector<int32_t> names(2 * count);
	bool any = false;
	for (size_t i = 0; i < count; i++) {
		bool copy = mesh_ids[i] >= 0;
		RT_REQUIRE(ctx, !copy || (triangle_ids[i] >= 0 && size_t(triangle_ids[i]) < count && mesh_ids[triangle_ids[i]] < 0), "rt_upload_triangle_aliases: a copy must name a triangle of the array that is not itself a copy");
		names[2 * i] = copy ? mesh_ids[i] : -1; names[2 * i + 1] = copy ? triangle_ids[i] : -1;
		any = any || copy;
	}
	RT_HIP(ctx, quiesce(ctx));
	void * device_names = nullptr;
	int s = device_alloc(ctx, &device_names, names.size() * 4); if (s) return s;
	hipError_t e = hipMemcpy(device_names, names.data(), names.size() * 4, hipMemcpyHostToDevice);
	if (e == hipSuccess && count) {
		kernel_name_triangles<<<unsigned((count + 255) / 256), 256, 0, ctx->stream>>>((float4 *)ctx->triangle_positions, (const int2 *)device_names, int(count));
		e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	}
	device_free(ctx, device_names);
	if (e != hipSuccess) return fail(ctx, RT_ERROR_HIP, "rt_upload_triangle_aliases: %s", hipGetErrorString(e));
	ctx->params.has_triangle_aliases = any ? 1 : 0;
	return RT_OK;
}

// Replaces nodes [first_node, first_node + node_count) of the uploaded CWBVH node array in place: a tree whose children have been re-seated beside the frame
// loop (host/Integrator.cpp: reseat worker). A ray holds stack entries (child base, mask) only INSIDE one traversal launch, so the swap is safe between launches:
// the context is drained first. Boxes, leaves and triangles must be what they were (the caller's contract: the node count and every index stay).
int rt_update_nodes(rt_context * ctx, const void * nodes, size_t first_node, size_t node_count) {
	RT_REQUIRE(ctx, ctx && (nodes || node_count == 0), "rt_update_nodes: NULL argument");
	RT_REQUIRE(ctx, ctx->bvh8_nodes && first_node <= ctx->bvh8_node_count && node_count <= ctx->bvh8_node_count - first_node, "rt_update_nodes: range outside the uploaded CWBVH node array");
	if (node_count == 0) return RT_OK;
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemcpy((char *)ctx->bvh8_nodes + first_node * 80, nodes, node_count * 80, hipMemcpyHostToDevice));
	return RT_OK;
}

// Where rays start: 0 = at the TLAS root in node slot 0; 1 = the whole scene is ONE world-space bottom-level tree whose root node
// the caller has put into node slot 0 (through rt_upload_tlas), rays are inside it from the start, as instance row 0. A change
// drains the context first: samples in flight were submitted against the old entry.
int rt_set_static_geometry(rt_context * ctx, int32_t whole_scene) {
	RT_REQUIRE(ctx, ctx != nullptr, "rt_set_static_geometry: NULL context");
	int entry = whole_scene ? 0 : RT_INVALID;
	if (ctx->params.entry_tlas_stack_size == entry) return RT_OK;
	RT_REQUIRE(ctx, !whole_scene || ctx->bvh8_nodes, "rt_set_static_geometry: CWBVH geometry only");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	ctx->params.entry_tlas_stack_size = entry;
	return RT_OK;
}

// "Skip behind the hit": see kernels_trace.hip. The wish is stored; the kernels take the walk when the scene is also one tree below 4 GiB (rt_skip_walk).
int rt_set_skip_behind_hit(rt_context * ctx, int32_t enable) {
	RT_REQUIRE(ctx, ctx != nullptr, "rt_set_skip_behind_hit: NULL context");
	if ((ctx->params.skip_behind_hit != 0) == (enable != 0)) return RT_OK;
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));   // (samples in flight were submitted under the other walk; the counters of a statistics pass must belong to one)
	ctx->params.skip_behind_hit = enable != 0;
	return RT_OK;
}
int rt_get_skip_behind_hit(const rt_context * ctx) { return ctx && rt_skip_walk(ctx->params) ? 1 : 0; }

// ---- BLAS build on the device (kernels_blas.hip) ---------------------------------------------------------------------
} // extern "C"
struct BlasBuildArgs { // must match kernels_blas.hip
	int triangle_count, mesh_count, first_node;
	const float4 * triangles; const int * mesh_first;
	float4 * triangles_out, * positions_out; uint32_t * nodes; int * order, * position;
	TlasBox * triangle_boxes, * sorted_boxes, * mesh_boxes, * child_boxes; TlasBox * box_table; int table_levels, split_widest; int * triangle_mesh;
	uint64_t * keys; int * ids; uint64_t * sorted_keys; int * sorted_ids;
	int2 * range; int * runs; int * inner_count, * leaf_count, * inner_base, * leaf_base; int * level_state;
	const float * given_boxes; int given_first, given_count;
};
size_t rt_blas_build_scratch_bytes(size_t triangles, size_t meshes);
hipError_t rt_blas_build(BlasBuildArgs a, void * library_scratch, size_t library_scratch_bytes, int * pinned_state, hipStream_t stream, int * out_node_count, size_t node_capacity);
extern "C" {

int rt_set_build_boxes(rt_context * ctx, const float * boxes, size_t first_triangle, size_t count) {
	RT_REQUIRE(ctx, ctx && (boxes || count == 0), "rt_set_build_boxes: NULL argument");
	RT_REQUIRE(ctx, first_triangle < (size_t(1) << 30) && count < (size_t(1) << 30), "rt_set_build_boxes: range out of bounds");
	ctx->build_boxes.assign(boxes, boxes + 6 * count);
	ctx->build_boxes_first = first_triangle;
	return RT_OK;
}

int rt_build_geometry(rt_context * ctx, const void * triangles, size_t triangle_count, const int32_t * mesh_first_triangle, size_t mesh_count,
                      size_t reserved_tlas_nodes, int32_t * out_root_indices, int32_t * out_triangle_positions, size_t * out_node_count, float * out_build_ms) {
	// rt_set_build_boxes: boxes that come with some of the triangles -- taken by THIS call whatever becomes of it (an early error exit used to leave them pending
	// for the next build, of possibly unrelated geometry: advisor finding, round 5); whatever does not fit the input is ignored
	std::vector<float> given_boxes; size_t given_first = 0;
	if (ctx) { given_boxes.swap(ctx->build_boxes); given_first = ctx->build_boxes_first; ctx->build_boxes_first = 0; }
	RT_REQUIRE(ctx, ctx && triangles && mesh_first_triangle && mesh_count >= 1, "rt_build_geometry: NULL argument");
	RT_REQUIRE(ctx, triangle_count < (size_t(1) << 30) && mesh_count < (size_t(1) << 24), "rt_build_geometry: too many triangles / meshes");
	RT_REQUIRE(ctx, mesh_first_triangle[0] == 0 && size_t(mesh_first_triangle[mesh_count]) == triangle_count, "rt_build_geometry: mesh_first_triangle must run from 0 to triangle_count");
	for (size_t m = 0; m < mesh_count; m++) RT_REQUIRE(ctx, mesh_first_triangle[m] <= mesh_first_triangle[m + 1], "rt_build_geometry: mesh_first_triangle must not decrease");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	const size_t T = triangle_count, M = mesh_count;
	const size_t node_capacity = reserved_tlas_nodes + M + T;          // every inner node has at least two children
	const size_t level_capacity = std::max(M, T / (4) + 1) + 1;        // an inner node holds more than 3 triangles
	// outputs (kept): triangles and positions in leaf order, nodes; everything else is scratch of this call
	void * out_triangles = nullptr, * out_positions = nullptr, * out_nodes = nullptr, * scratch = nullptr;
	int * pinned = nullptr; hipEvent_t t0 = nullptr, t1 = nullptr;
	// ONE way out on failure: whatever of the outputs, the scratch area, the pinned word and the timing events exists is released
	// (a scratch allocation that does not fit is a recoverable error: it must not strand the hundreds of MB allocated before it)
	auto give_up = [&](int status) {
		device_free(ctx, scratch); device_free(ctx, out_triangles); device_free(ctx, out_positions); device_free(ctx, out_nodes);
		if (pinned) (void)hipHostFree(pinned);
		if (t0) (void)hipEventDestroy(t0);
		if (t1) (void)hipEventDestroy(t1);
		return status;
	};
	#define RT_BUILD_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return give_up(fail(ctx, RT_ERROR_HIP, "rt_build_geometry: %s failed: %s", #call, hipGetErrorString(e_))); } while (0)
	int s;
	if ((s = device_alloc(ctx, &out_triangles, T * 96))) return give_up(s);
	if ((s = device_alloc(ctx, &out_positions, T * 48))) return give_up(s);
	if ((s = device_alloc(ctx, &out_nodes, node_capacity * 80))) return give_up(s);
	size_t at = 0;
	auto region = [&](size_t bytes) { size_t begin = at; at += (bytes + 255) / 256 * 256; return begin; };
	const size_t o_in = region(T * 96), o_first = region((M + 1) * 4), o_order = region(T * 4), o_position = region(T * 4),
	             o_tbox = region(T * 24), o_sbox = region(T * 24), o_mbox = region(M * 24), o_cbox = region(level_capacity * 8 * 24), o_tmesh = region(T * 4),
	             o_keys = region(T * 8), o_ids = region(T * 4), o_skeys = region(T * 8), o_sids = region(T * 4), o_range = region(node_capacity * 8),
	             o_runs = region(level_capacity * 48), o_ic = region(level_capacity * 4), o_lc = region(level_capacity * 4), o_ib = region(level_capacity * 4), o_lb = region(level_capacity * 4),
	             o_state = region(16);
	// boxes of all power-of-two runs of the sorted order (kernel_blas_box_table: the area of any run is two look-ups): levels 1 .. floor(log2 T)
	int table_levels = 0; while ((size_t(2) << table_levels) <= T) table_levels++;
	static const bool split_widest = getenv("GRT_BLAS_SPLIT_WIDEST") != nullptr;   // round 3's rule (most triangles first), for tools/blas_bench.py
	const size_t o_table = region(split_widest ? 0 : size_t(table_levels) * T * 24);
	const size_t library_bytes = rt_blas_build_scratch_bytes(T, M + level_capacity);
	const size_t o_library = region(library_bytes);
	const size_t given_count = given_first + given_boxes.size() / 6 <= T ? given_boxes.size() / 6 : 0;
	const size_t o_given = region(given_count * 24);
	if ((s = device_alloc(ctx, &scratch, at))) return give_up(s);
	char * base = (char *)scratch;
	RT_BUILD_HIP(hipMemcpyAsync(base + o_in, triangles, T * 96, hipMemcpyHostToDevice, ctx->stream));
	RT_BUILD_HIP(hipMemcpyAsync(base + o_first, mesh_first_triangle, (M + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	if (given_count) RT_BUILD_HIP(hipMemcpyAsync(base + o_given, given_boxes.data(), given_count * 24, hipMemcpyHostToDevice, ctx->stream));
	RT_BUILD_HIP(hipMemsetAsync(out_nodes, 0, node_capacity * 80, ctx->stream));
	BlasBuildArgs a;
	a.triangle_count = int(T); a.mesh_count = int(M); a.first_node = int(reserved_tlas_nodes);
	a.triangles = (const float4 *)(base + o_in); a.mesh_first = (const int *)(base + o_first);
	a.triangles_out = (float4 *)out_triangles; a.positions_out = (float4 *)out_positions; a.nodes = (uint32_t *)out_nodes;
	a.order = (int *)(base + o_order); a.position = (int *)(base + o_position);
	a.triangle_boxes = (TlasBox *)(base + o_tbox); a.sorted_boxes = (TlasBox *)(base + o_sbox); a.mesh_boxes = (TlasBox *)(base + o_mbox); a.child_boxes = (TlasBox *)(base + o_cbox);
	a.triangle_mesh = (int *)(base + o_tmesh);
	a.box_table = (TlasBox *)(base + o_table); a.table_levels = table_levels; a.split_widest = split_widest ? 1 : 0;
	a.keys = (uint64_t *)(base + o_keys); a.ids = (int *)(base + o_ids); a.sorted_keys = (uint64_t *)(base + o_skeys); a.sorted_ids = (int *)(base + o_sids);
	a.range = (int2 *)(base + o_range); a.runs = (int *)(base + o_runs);
	a.inner_count = (int *)(base + o_ic); a.leaf_count = (int *)(base + o_lc); a.inner_base = (int *)(base + o_ib); a.leaf_base = (int *)(base + o_lb);
	a.level_state = (int *)(base + o_state);
	a.given_boxes = given_count ? (const float *)(base + o_given) : nullptr; a.given_first = int(given_first); a.given_count = int(given_count);
	RT_BUILD_HIP(hipHostMalloc((void **)&pinned, 16));
	RT_BUILD_HIP(hipEventCreate(&t0)); RT_BUILD_HIP(hipEventCreate(&t1));
	RT_BUILD_HIP(hipEventRecord(t0, ctx->stream));
	int node_count = 0;
	hipError_t e = rt_blas_build(a, base + o_library, library_bytes, pinned, ctx->stream, &node_count, node_capacity);
	if (e == hipSuccess) e = hipEventRecord(t1, ctx->stream);
	if (e == hipSuccess && out_triangle_positions) e = hipMemcpyAsync(out_triangle_positions, a.position, T * 4, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	float ms = 0.0f;
	if (e == hipSuccess) (void)hipEventElapsedTime(&ms, t0, t1);
	if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return give_up(fail(ctx, RT_ERROR_HIP, "rt_build_geometry: %s", hipGetErrorString(e))); }
	(void)hipEventDestroy(t0); (void)hipEventDestroy(t1); (void)hipHostFree(pinned);
	device_free(ctx, scratch);
	#undef RT_BUILD_HIP
	// the context's geometry is what was built
	device_free(ctx, ctx->triangles); device_free(ctx, ctx->triangle_positions); device_free(ctx, ctx->bvh8_nodes);
	ctx->triangles = out_triangles; ctx->triangle_positions = out_positions; ctx->bvh8_nodes = out_nodes;
	ctx->triangle_count = T; ctx->bvh8_node_count = size_t(node_count);
	ctx->tlas_version_in_nodes = ~0ull;
	ctx->params.triangles = (const float4 *)out_triangles; ctx->params.triangle_positions = (const float4 *)out_positions; ctx->params.bvh8_nodes = (const float4 *)out_nodes;
	ctx->params.geometry_below_4gib = rt_geometry_fits_flat_engine(size_t(node_count), size_t(T));
	ctx->params.has_triangle_aliases = 0; ctx->params.entry_tlas_stack_size = RT_INVALID;
	if (out_root_indices) for (size_t m = 0; m < M; m++) out_root_indices[m] = int32_t(reserved_tlas_nodes + m);
	if (out_node_count) *out_node_count = size_t(node_count);
	if (out_build_ms) *out_build_ms = ms;
	return RT_OK;
}

int rt_read_geometry(rt_context * ctx, void * out_triangles, void * out_bvh8_nodes) {
	RT_REQUIRE(ctx, ctx && ctx->triangles && ctx->bvh8_nodes, "rt_read_geometry: no CWBVH geometry on the device");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	if (out_triangles)  RT_HIP(ctx, hipMemcpy(out_triangles,  ctx->triangles,  ctx->triangle_count * 96, hipMemcpyDeviceToHost));
	if (out_bvh8_nodes) RT_HIP(ctx, hipMemcpy(out_bvh8_nodes, ctx->bvh8_nodes, ctx->bvh8_node_count * 80, hipMemcpyDeviceToHost));
	return RT_OK;
}

// TLAS of the selected BVH type: node_bytes = 80 (CWBVH), 32 (binary) or 128 (4-wide)
static int upload_tlas_version(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count, size_t node_bytes) {
	void * staging = nullptr;
	int s = ring_begin(ctx, ctx->tlas_ring, tlas_node_count * node_bytes, &staging); if (s) return s;
	memcpy(staging, tlas_nodes, tlas_node_count * node_bytes);
	s = ring_commit(ctx, ctx->tlas_ring); if (s) return s;
	ctx->params.tlas_nodes = (const float4 *)ctx->tlas_ring.device[ctx->tlas_ring.current];
	ctx->params.tlas_node_count = int(tlas_node_count);
	ctx->tlas_node_bytes = node_bytes;
	ctx->tlas_version++;
	return RT_OK;
}

int rt_upload_tlas(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count) {
	RT_REQUIRE(ctx, ctx && tlas_nodes, "rt_upload_tlas: NULL argument");
	RT_REQUIRE(ctx, ctx->bvh8_nodes && tlas_node_count <= ctx->bvh8_node_count, "rt_upload_tlas: geometry not uploaded or TLAS larger than the node array");
	(void)hipSetDevice(ctx->device);
	return upload_tlas_version(ctx, tlas_nodes, tlas_node_count, 80);
}

int rt_upload_geometry_bvh2(rt_context * ctx, const void * triangles, size_t triangle_count, const void * bvh2_nodes, size_t node_count) {
	RT_REQUIRE(ctx, ctx && triangles && bvh2_nodes, "rt_upload_geometry_bvh2: NULL argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->triangles, triangles, triangle_count * 96); if (s) return s;
	s = upload(ctx, &ctx->bvh2_nodes, bvh2_nodes, node_count * 32); if (s) return s;
	s = upload_triangle_positions(ctx, triangles, triangle_count); if (s) return s;
	ctx->triangle_count = triangle_count; ctx->bvh2_node_count = node_count;
	ctx->params.triangles  = (const float4 *)ctx->triangles;
	ctx->params.bvh2_nodes = (const float4 *)ctx->bvh2_nodes;
	return RT_OK;
}

int rt_upload_tlas_bvh2(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count) {
	RT_REQUIRE(ctx, ctx && tlas_nodes, "rt_upload_tlas_bvh2: NULL argument");
	RT_REQUIRE(ctx, ctx->bvh2_nodes && tlas_node_count <= ctx->bvh2_node_count, "rt_upload_tlas_bvh2: geometry not uploaded or TLAS larger than the node array");
	(void)hipSetDevice(ctx->device);
	return upload_tlas_version(ctx, tlas_nodes, tlas_node_count, 32);
}

int rt_upload_geometry_bvh4(rt_context * ctx, const void * triangles, size_t triangle_count, const void * bvh4_nodes, size_t node_count) {
	RT_REQUIRE(ctx, ctx && triangles && bvh4_nodes, "rt_upload_geometry_bvh4: NULL argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->triangles, triangles, triangle_count * 96); if (s) return s;
	s = upload(ctx, &ctx->bvh4_nodes, bvh4_nodes, node_count * 128); if (s) return s;
	s = upload_triangle_positions(ctx, triangles, triangle_count); if (s) return s;
	ctx->triangle_count = triangle_count; ctx->bvh4_node_count = node_count;
	ctx->params.triangles  = (const float4 *)ctx->triangles;
	ctx->params.bvh4_nodes = (const float4 *)ctx->bvh4_nodes;
	return RT_OK;
}

int rt_upload_tlas_bvh4(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count) {
	RT_REQUIRE(ctx, ctx && tlas_nodes, "rt_upload_tlas_bvh4: NULL argument");
	RT_REQUIRE(ctx, ctx->bvh4_nodes && tlas_node_count <= ctx->bvh4_node_count, "rt_upload_tlas_bvh4: geometry not uploaded or TLAS larger than the node array");
	(void)hipSetDevice(ctx->device);
	return upload_tlas_version(ctx, tlas_nodes, tlas_node_count, 128);
}

int rt_set_bvh_type(rt_context * ctx, int bvh_width) {
	RT_REQUIRE(ctx, ctx, "rt_set_bvh_type: NULL context");
	if (bvh_width != 8 && bvh_width != 4 && bvh_width != 2) return fail(ctx, RT_ERROR_INVALID_ARG, "rt_set_bvh_type: the device has kernels for the binary BVH (2), the 4-wide BVH (4) and the 8-wide CWBVH (8), got %d", bvh_width);
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	ctx->bvh_width = bvh_width;
	ctx->params.bvh_width = bvh_width;
	return RT_OK;
}

int rt_upload_instances(rt_context * ctx, const int32_t * root_indices, const int32_t * material_ids,
                        const float * transforms, const float * transforms_inv, const float * transforms_prev, size_t mesh_count) {
	RT_REQUIRE(ctx, ctx && root_indices && material_ids && transforms && transforms_inv && transforms_prev, "rt_upload_instances: NULL argument");
	(void)hipSetDevice(ctx->device);
	// one allocation per version: roots | material ids | transforms | inverse | previous (48-byte rows are 16-byte aligned)
	size_t padded = (mesh_count + 3) / 4 * 4;
	size_t offset[6] = { 0, padded * 4, padded * 8, padded * 8 + mesh_count * 48, padded * 8 + mesh_count * 96, padded * 8 + mesh_count * 144 };
	const void * src[5] = { root_indices, material_ids, transforms, transforms_inv, transforms_prev };
	size_t bytes[5] = { mesh_count * 4, mesh_count * 4, mesh_count * 48, mesh_count * 48, mesh_count * 48 };
	void * staging = nullptr;
	int s = ring_begin(ctx, ctx->instance_ring, offset[5], &staging); if (s) return s;
	for (int i = 0; i < 5; i++) memcpy((char *)staging + offset[i], src[i], bytes[i]);
	s = ring_commit(ctx, ctx->instance_ring); if (s) return s;
	ctx->lowest_blas_root = 0x7fffffff;
	for (size_t i = 0; i < mesh_count; i++) ctx->lowest_blas_root = std::min(ctx->lowest_blas_root, int(root_indices[i] & 0x7fffffff));
	const char * base = (const char *)ctx->instance_ring.device[ctx->instance_ring.current];
	ctx->mesh_count = mesh_count; ctx->params.mesh_count = int(mesh_count);
	ctx->params.mesh_bvh_root_indices = (const int *)(base + offset[0]);
	ctx->params.mesh_material_ids     = (const int *)(base + offset[1]);
	ctx->params.mesh_transforms       = (const float4 *)(base + offset[2]);
	ctx->params.mesh_transforms_inv   = (const float4 *)(base + offset[3]);
	ctx->params.mesh_transforms_prev  = (const float4 *)(base + offset[4]);
	ctx->params.mesh_position = nullptr;   // the host supplies everything in TLAS order
	ctx->device_tlas_order = nullptr; ctx->device_tlas_node_count = nullptr;
	return RT_OK;
}

// ---- TLAS build on the device (kernels_build.hip) ---------------------------------------------------------------
} // extern "C"
struct TlasBuildArgs;
void rt_launch_build_tlas(const TlasBuildArgs & args, hipStream_t stream);
struct TlasBuildArgs { // must match kernels_build.hip
	int count;
	const int * root_indices; const int * material_ids;
	const float4 * transforms, * transforms_inv, * transforms_prev;
	const float * local_boxes;
	uint32_t * nodes; int * out_root_indices, * out_material_ids;
	float4 * out_transforms, * out_transforms_inv, * out_transforms_prev;
	int * order; int * position; int * node_count;
	TlasBox * boxes, * sorted_boxes; int * queue; int * runs; int * bases; TlasBox * child_boxes;
};
extern "C" {

int rt_build_tlas(rt_context * ctx, const int32_t * root_indices, const int32_t * material_ids,
                  const float * transforms, const float * transforms_inv, const float * transforms_prev,
                  const float * local_boxes, size_t mesh_count) {
	RT_REQUIRE(ctx, ctx && root_indices && material_ids && transforms && transforms_inv && transforms_prev && local_boxes, "rt_build_tlas: NULL argument");
	RT_REQUIRE(ctx, mesh_count >= 1 && mesh_count <= RT_TLAS_BUILD_MAX, "rt_build_tlas: 1 .. 4096 instances (larger scenes build the TLAS on the host: rt_upload_tlas)");
	RT_REQUIRE(ctx, ctx->bvh_width == 8 && ctx->bvh8_nodes && 2 * mesh_count <= ctx->bvh8_node_count, "rt_build_tlas: CWBVH geometry with 2 * mesh_count reserved node slots must be uploaded first");
	(void)hipSetDevice(ctx->device);
	const size_t n = mesh_count, padded = (n + 3) / 4 * 4;
	// one allocation per version: outputs | inputs | scratch (every region 16-byte aligned)
	size_t offset[24]; size_t at = 0; int regions = 0;
	auto region = [&](size_t bytes) { offset[regions++] = at; at += (bytes + 15) / 16 * 16; };
	region(padded * 4); region(padded * 4); region(n * 48); region(n * 48); region(n * 48);   // 0..4 out: roots, materials, transforms, inverse, previous
	region(padded * 4); region(padded * 4); region(16);                                      // 5..7 out: position, order, node count
	const size_t inputs_begin = at;
	region(padded * 4); region(padded * 4); region(n * 48); region(n * 48); region(n * 48); region(n * 24); // 8..13 in
	const size_t inputs_end = at;
	region(n * 24); region(n * 24); region(n * 48); region(n * 8); region(n * 192); region(n * 24); // 14..19 scratch: boxes, queues, runs, bases, child boxes, boxes in sorted order
	void * staging = nullptr;
	int s = ring_begin(ctx, ctx->instance_ring, at, &staging); if (s) return s;
	const void * src[6] = { root_indices, material_ids, transforms, transforms_inv, transforms_prev, local_boxes };
	const size_t bytes[6] = { n * 4, n * 4, n * 48, n * 48, n * 48, n * 24 };
	for (int i = 0; i < 6; i++) memcpy((char *)staging + offset[8 + i], src[i], bytes[i]);
	char * base = (char *)ctx->instance_ring.device[ctx->instance_ring.current];
	RT_HIP(ctx, hipMemcpyAsync(base + inputs_begin, (char *)staging + inputs_begin, inputs_end - inputs_begin, hipMemcpyHostToDevice, ctx->stream));
	void * tlas_staging = nullptr;
	s = ring_begin(ctx, ctx->tlas_ring, 2 * n * 80, &tlas_staging); if (s) return s;
	void * tlas_device = ctx->tlas_ring.device[ctx->tlas_ring.current];

	TlasBuildArgs a;
	a.count = int(n);
	a.root_indices = (const int *)(base + offset[8]); a.material_ids = (const int *)(base + offset[9]);
	a.transforms = (const float4 *)(base + offset[10]); a.transforms_inv = (const float4 *)(base + offset[11]); a.transforms_prev = (const float4 *)(base + offset[12]);
	a.local_boxes = (const float *)(base + offset[13]);
	a.nodes = (uint32_t *)tlas_device;
	a.out_root_indices = (int *)(base + offset[0]); a.out_material_ids = (int *)(base + offset[1]);
	a.out_transforms = (float4 *)(base + offset[2]); a.out_transforms_inv = (float4 *)(base + offset[3]); a.out_transforms_prev = (float4 *)(base + offset[4]);
	a.position = (int *)(base + offset[5]); a.order = (int *)(base + offset[6]); a.node_count = (int *)(base + offset[7]);
	a.boxes = (TlasBox *)(base + offset[14]); a.queue = (int *)(base + offset[15]); a.runs = (int *)(base + offset[16]); a.bases = (int *)(base + offset[17]); a.child_boxes = (TlasBox *)(base + offset[18]); a.sorted_boxes = (TlasBox *)(base + offset[19]);
	rt_launch_build_tlas(a, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	RT_HIP(ctx, hipEventRecord(ctx->instance_ring.copied[ctx->instance_ring.current], ctx->stream));
	RT_HIP(ctx, hipEventRecord(ctx->tlas_ring.copied[ctx->tlas_ring.current], ctx->stream));
	RT_HIP(ctx, hipEventRecord(ctx->ev_scene, ctx->stream));

	ctx->mesh_count = n; ctx->params.mesh_count = int(n);
	ctx->params.tlas_nodes = (const float4 *)tlas_device;
	ctx->params.tlas_node_count = int(2 * n);     // the node slots reserved for the TLAS; BLAS nodes start behind them
	ctx->params.entry_tlas_stack_size = RT_INVALID;   // node 0 is a TLAS root from now on: rays start above the instances (rt_set_static_geometry(ctx, 1) does not survive a TLAS build)
	ctx->tlas_version++;
	ctx->params.mesh_bvh_root_indices = a.out_root_indices;
	ctx->params.mesh_material_ids     = a.out_material_ids;
	ctx->params.mesh_transforms       = a.out_transforms;
	ctx->params.mesh_transforms_inv   = a.out_transforms_inv;
	ctx->params.mesh_transforms_prev  = a.out_transforms_prev;
	ctx->params.mesh_position         = a.position;
	ctx->device_tlas_order = a.order; ctx->device_tlas_node_count = a.node_count;
	return RT_OK;
}

int rt_read_tlas(rt_context * ctx, int32_t * order, void * nodes, size_t node_capacity, int32_t * node_count) {
	RT_REQUIRE(ctx, ctx, "rt_read_tlas: NULL context");
	(void)hipSetDevice(ctx->device);
	if (!ctx->device_tlas_order) return fail(ctx, RT_ERROR_NOT_READY, "rt_read_tlas: the current TLAS was not built on the device (rt_build_tlas)");
	RT_HIP(ctx, quiesce(ctx));
	int count = 0;
	RT_HIP(ctx, hipMemcpy(&count, ctx->device_tlas_node_count, sizeof(int), hipMemcpyDeviceToHost));
	if (node_count) *node_count = count;
	if (order) RT_HIP(ctx, hipMemcpy(order, ctx->device_tlas_order, ctx->mesh_count * sizeof(int), hipMemcpyDeviceToHost));
	if (nodes) RT_HIP(ctx, hipMemcpy(nodes, ctx->params.tlas_nodes, std::min(node_capacity, size_t(count)) * 80, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_upload_materials(rt_context * ctx, const uint8_t * types, const void * materials, size_t count) {
	RT_REQUIRE(ctx, ctx && types && materials, "rt_upload_materials: NULL argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->material_types, types, count); if (s) return s;
	s = upload(ctx, &ctx->materials, materials, count * 32); if (s) return s;
	ctx->params.material_types = (const uint8_t *)ctx->material_types;
	ctx->params.materials      = (const float4 *)ctx->materials;

	// Scene::check_materials (Scene.cpp:50-70): which material kernels have to run at all
	for (bool & h : ctx->has_material) h = false;
	ctx->has_lights = false;
	const float * m = (const float *)materials;
	for (size_t i = 0; i < count; i++) {
		switch (types[i]) {
			case RT_MATERIAL_DIFFUSE:    ctx->has_material[0] = true; break;
			case RT_MATERIAL_PLASTIC:    ctx->has_material[1] = true; break;
			case RT_MATERIAL_DIELECTRIC: ctx->has_material[2] = true; break;
			case RT_MATERIAL_CONDUCTOR:  ctx->has_material[3] = true; break;
			case RT_MATERIAL_LIGHT:      ctx->has_lights |= (m[8 * i] * m[8 * i] + m[8 * i + 1] * m[8 * i + 1] + m[8 * i + 2] * m[8 * i + 2]) > 0.0f; break;
			default: return fail(ctx, RT_ERROR_INVALID_ARG, "rt_upload_materials: unknown material type %d at index %zu", int(types[i]), i);
		}
	}
	return RT_OK;
}

int rt_upload_media(rt_context * ctx, const void * media, size_t count) {
	RT_REQUIRE(ctx, ctx && (media || count == 0), "rt_upload_media: NULL argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->media, media, count * 32); if (s) return s;
	ctx->params.media = (const float4 *)ctx->media;
	return RT_OK;
}

int rt_upload_textures(rt_context * ctx, const rt_texture_desc * descs, size_t count) {
	RT_REQUIRE(ctx, ctx && (descs || count == 0), "rt_upload_textures: NULL argument");
	(void)hipSetDevice(ctx->device);
	for (void * p : ctx->texture_data) device_free(ctx, p);
	ctx->texture_data.clear(); ctx->texture_bytes = 0;
	std::vector<RtTexture> table(count);
	for (size_t i = 0; i < count; i++) {
		const rt_texture_desc & d = descs[i];
		RT_REQUIRE(ctx, d.texels && d.width > 0 && d.height > 0 && d.mip_levels > 0, "rt_upload_textures: invalid texture descriptor");
		RT_REQUIRE(ctx, d.format == RT_TEXTURE_RGBA8 || d.format == RT_TEXTURE_BC1, "rt_upload_textures: unknown texture format");
		size_t bytes = 0;
		for (int l = 0; l < d.mip_levels; l++) {
			int w = d.width >> l; if (w < 1) w = 1; int h = d.height >> l; if (h < 1) h = 1;
			bytes += d.format == RT_TEXTURE_BC1 ? size_t((w + 3) / 4) * ((h + 3) / 4) * 8 : size_t(w) * h * 4;
		}
		void * dev = nullptr;
		int s = upload(ctx, &dev, d.texels, bytes); if (s) return s;
		int device_format = d.format;
		if (d.format == RT_TEXTURE_BC1 && ctx->expand_bc1) {   // decode every block once, here, instead of once per texel fetch
			void * expanded = nullptr;
			s = device_alloc(ctx, &expanded, bytes * 8); if (s) { device_free(ctx, dev); return s; }
			rt_launch_expand_bc1((const uint2 *)dev, (uchar4 *)expanded, bytes / 8, ctx->stream);
			hipError_t e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			device_free(ctx, dev);
			if (e != hipSuccess) { device_free(ctx, expanded); return fail(ctx, RT_ERROR_HIP, "rt_upload_textures: expanding BC1 blocks failed: %s", hipGetErrorString(e)); }
			dev = expanded; bytes *= 8; device_format = RT_TEXTURE_BC1_EXPANDED;
		}
		ctx->texture_data.push_back(dev);
		ctx->texture_bytes += bytes;
		table[i].texels = (const uchar4 *)dev;
		table[i].width = d.width; table[i].height = d.height; table[i].mip_levels = d.mip_levels;
		table[i].format = device_format; table[i].pad = 0;
		int lod_width  = d.lod_width  > 0 ? d.lod_width  : d.width;
		int lod_height = d.lod_height > 0 ? d.lod_height : d.height;
		table[i].lod_bias = 0.5f * log2f(float(lod_width * lod_height)); // Integrator.cpp:95
	}
	int s = upload(ctx, &ctx->texture_table, table.data(), count * sizeof(RtTexture)); if (s) return s;
	ctx->params.textures = (const RtTexture *)ctx->texture_table;
	ctx->params.textures_compressed = 0;
	for (const RtTexture & t : table) if (t.format == RT_TEXTURE_BC1) ctx->params.textures_compressed = 1;
	return RT_OK;
}

int rt_upload_lights(rt_context * ctx,
                     const int32_t * light_triangle_indices, const float * light_triangle_cumulative_probability, size_t light_triangle_count,
                     const float * light_mesh_cumulative_probability, const int32_t * light_mesh_triangle_span,
                     const int32_t * light_mesh_transform_indices, size_t light_mesh_count, float lights_total_weight) {
	RT_REQUIRE(ctx, ctx, "rt_upload_lights: NULL context");
	(void)hipSetDevice(ctx->device);
	// the per-mesh tables change with every TLAS rebuild (they are in TLAS order): versioned like the TLAS itself
	const void * src[5] = { light_triangle_indices, light_triangle_cumulative_probability, light_mesh_cumulative_probability, light_mesh_triangle_span, light_mesh_transform_indices };
	size_t bytes[5] = { light_triangle_count * 4, light_triangle_count * 4, light_mesh_count * 4, light_mesh_count * 8, light_mesh_count * 4 };
	size_t offset[6] = { 0 };
	for (int i = 0; i < 5; i++) offset[i + 1] = offset[i] + ((src[i] ? bytes[i] : 0) + 15) / 16 * 16;
	void * staging = nullptr;
	int s = ring_begin(ctx, ctx->light_ring, offset[5], &staging); if (s) return s;
	for (int i = 0; i < 5; i++) if (src[i] && bytes[i]) memcpy((char *)staging + offset[i], src[i], bytes[i]);
	s = ring_commit(ctx, ctx->light_ring); if (s) return s;
	const char * base = (const char *)ctx->light_ring.device[ctx->light_ring.current];
	ctx->params.light_triangle_indices                = (const int *)(base + offset[0]);
	ctx->params.light_triangle_cumulative_probability = (const float *)(base + offset[1]);
	ctx->params.light_mesh_cumulative_probability     = (const float *)(base + offset[2]);
	ctx->params.light_mesh_triangle_span              = (const int2 *)(base + offset[3]);
	ctx->params.light_mesh_transform_indices          = (const int *)(base + offset[4]);
	ctx->params.light_mesh_count    = int(light_mesh_count);
	ctx->params.light_triangle_count = int(light_triangle_count);
	ctx->params.lights_total_weight = lights_total_weight;
	return RT_OK;
}

static int ensure_luts(rt_context * ctx); // needs the RNG tables

int rt_upload_rng(rt_context * ctx, const float * pmj_samples, const uint8_t * blue_noise) {
	RT_REQUIRE(ctx, ctx && pmj_samples && blue_noise, "rt_upload_rng: NULL argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->pmj, pmj_samples, size_t(RT_PMJ_NUM_SEQUENCES) * RT_PMJ_NUM_SAMPLES_PER_SEQUENCE * 8); if (s) return s;
	s = upload(ctx, &ctx->blue_noise, blue_noise, size_t(RT_BLUE_NOISE_NUM_TEXTURES) * RT_BLUE_NOISE_TEXTURE_DIM * RT_BLUE_NOISE_TEXTURE_DIM * 2); if (s) return s;
	ctx->params.pmj_samples = (const float2 *)ctx->pmj;
	ctx->params.blue_noise  = (const uchar2 *)ctx->blue_noise;
	ctx->luts_ready = false;
	return RT_OK;
}

int rt_set_sky(rt_context * ctx, const float * rgba, int width, int height, float scale) {
	RT_REQUIRE(ctx, ctx && rgba && width > 0 && height > 0, "rt_set_sky: invalid argument");
	(void)hipSetDevice(ctx->device);
	int s = upload(ctx, &ctx->sky, rgba, size_t(width) * height * 16); if (s) return s;
	ctx->params.sky = (const float4 *)ctx->sky;
	ctx->params.sky_width = width; ctx->params.sky_height = height; ctx->params.sky_scale = scale;
	return RT_OK;
}

// ---- frame state ---------------------------------------------------------------------------------------

static int alloc_vec3(rt_context * ctx, RtVec3SoA & v, size_t n) {
	int s = device_alloc(ctx, (void **)&v.x, n * 4); if (s) return s;
	s = device_alloc(ctx, (void **)&v.y, n * 4); if (s) return s;
	return device_alloc(ctx, (void **)&v.z, n * 4);
}

static size_t wanted_batch_size(const rt_context * ctx) {
	size_t frame = size_t(ctx->params.screen_width) * ctx->params.screen_height;
	size_t n = ctx->batch_size_request > 0 ? size_t(ctx->batch_size_request) : frame;
	if (n < 1) n = 1;
	return n;
}

static int ensure_queues(rt_context * ctx, int slot_index = 0, size_t pixels = 0) {
	SampleSlot & slot = ctx->slots[slot_index];
	size_t n = pixels > 0 ? pixels : wanted_batch_size(ctx); // entries: pixels of a batch x samples per batch
	if (slot.queues_allocated && slot.queue_capacity >= n) return RT_OK;
	if (slot.queues_allocated) { // grow: release the old queues
		RT_HIP(ctx, quiesce(ctx));
		auto free3 = [&](RtVec3SoA & v) { device_free(ctx, v.x); device_free(ctx, v.y); device_free(ctx, v.z); };
		for (int i = 0; i < 2; i++) { RtTraceBuffer & t = slot.trace[i]; free3(t.origin); free3(t.direction); device_free(ctx, t.hits); device_free(ctx, t.cone_angle); device_free(ctx, t.cone_width); device_free(ctx, t.medium); device_free(ctx, t.pixel_index_and_flags); free3(t.throughput); device_free(ctx, t.last_pdf); }
		for (int i = 0; i < 4; i++) { RtMaterialBuffer & m = slot.material[i]; free3(m.direction); device_free(ctx, m.hits); device_free(ctx, m.cone_angle); device_free(ctx, m.cone_width); device_free(ctx, m.medium); device_free(ctx, m.pixel_index_and_flags); free3(m.throughput); }
		RtShadowBuffer & sh = slot.shadow; free3(sh.origin); free3(sh.direction); device_free(ctx, sh.max_distance); device_free(ctx, sh.illumination_and_pixel_index);
		slot.queues_allocated = false;
	}
	slot.queue_capacity = n;
	int s;
	for (int i = 0; i < 2; i++) {
		RtTraceBuffer & t = slot.trace[i];
		if ((s = alloc_vec3(ctx, t.origin, n))) return s;
		if ((s = alloc_vec3(ctx, t.direction, n))) return s;
		if ((s = device_alloc(ctx, (void **)&t.hits, n * 16))) return s;
		if ((s = device_alloc(ctx, (void **)&t.cone_angle, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&t.cone_width, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&t.medium, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&t.pixel_index_and_flags, n * 4))) return s;
		if ((s = alloc_vec3(ctx, t.throughput, n))) return s;
		if ((s = device_alloc(ctx, (void **)&t.last_pdf, n * 4))) return s;
	}
	for (int i = 0; i < 4; i++) {
		RtMaterialBuffer & m = slot.material[i];
		if ((s = alloc_vec3(ctx, m.direction, n))) return s;
		if ((s = device_alloc(ctx, (void **)&m.hits, n * 16))) return s;
		if ((s = device_alloc(ctx, (void **)&m.cone_angle, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&m.cone_width, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&m.medium, n * 4))) return s;
		if ((s = device_alloc(ctx, (void **)&m.pixel_index_and_flags, n * 4))) return s;
		if ((s = alloc_vec3(ctx, m.throughput, n))) return s;
	}
	RtShadowBuffer & sh = slot.shadow;
	if ((s = alloc_vec3(ctx, sh.origin, n))) return s;
	if ((s = alloc_vec3(ctx, sh.direction, n))) return s;
	if ((s = device_alloc(ctx, (void **)&sh.max_distance, n * 4))) return s;
	if ((s = device_alloc(ctx, (void **)&sh.illumination_and_pixel_index, n * 16))) return s;
	slot.queues_allocated = true;
	if (slot_index == 0) { memcpy(ctx->params.trace, slot.trace, sizeof(slot.trace)); memcpy(ctx->params.material, slot.material, sizeof(slot.material)); ctx->params.shadow = slot.shadow; }
	return RT_OK;
}

static int sync_aovs(rt_context * ctx) {
	size_t bytes = ctx->frame_pixels * 16;
	for (int i = 0; i < RT_AOV_COUNT; i++) {
		bool enabled = (ctx->params.config.aov_mask >> i) & 1u;
		bool allocated = ctx->aov_buffers[i][0] != nullptr;
		if (enabled && !allocated && bytes) {
			RT_HIP(ctx, quiesce(ctx));
			stream_release_frames(ctx);
			for (int k = 0; k < 2; k++) { // [0]: slot 0's per-sample frame buffer(s), [1]: the accumulator
				size_t n = k == 0 ? bytes * ctx->slots[0].aov_samples : bytes;
				int s = device_alloc(ctx, &ctx->aov_buffers[i][k], n); if (s) return s;
				RT_HIP(ctx, hipMemset(ctx->aov_buffers[i][k], 0, n));
			}
			// (the filter's mirror of the radiance accumulators' variances starts from the same zeros)
			if ((i == RT_AOV_RADIANCE_DIRECT || i == RT_AOV_RADIANCE_INDIRECT) && ctx->svgf_buffers[13]) RT_HIP(ctx, hipMemset(ctx->svgf_buffers[13], 0, ctx->frame_pixels * 8));
			for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) if (ctx->slots[k].created) { // per-sample frame buffers of the other slots
				int s = device_alloc(ctx, &ctx->slots[k].aov_framebuffer[i], bytes * ctx->slots[k].aov_samples); if (s) return s;
				RT_HIP(ctx, hipMemset(ctx->slots[k].aov_framebuffer[i], 0, bytes * ctx->slots[k].aov_samples));
			}
		} else if (!enabled && allocated) {
			RT_HIP(ctx, quiesce(ctx));
			stream_release_frames(ctx);
			for (int k = 0; k < 2; k++) { device_free(ctx, ctx->aov_buffers[i][k]); ctx->aov_buffers[i][k] = nullptr; }
			for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) { device_free(ctx, ctx->slots[k].aov_framebuffer[i]); ctx->slots[k].aov_framebuffer[i] = nullptr; }
		}
		ctx->params.aovs[i].framebuffer = (float4 *)ctx->aov_buffers[i][0];
		ctx->params.aovs[i].accumulator = (float4 *)ctx->aov_buffers[i][1];
	}
	return RT_OK;
}

// Frame buffers of one slot for `samples` samples per batch (they only ever grow).
static int ensure_aov_batch(rt_context * ctx, int slot_index, int samples) {
	SampleSlot & slot = ctx->slots[slot_index];
	if (slot.aov_samples >= samples) return RT_OK;
	RT_HIP(ctx, quiesce(ctx));
	size_t bytes = ctx->frame_pixels * 16 * size_t(samples);
	for (int i = 0; i < RT_AOV_COUNT; i++) {
		void ** fb = slot_index == 0 ? &ctx->aov_buffers[i][0] : &slot.aov_framebuffer[i];
		if (!*fb) continue;
		device_free(ctx, *fb); *fb = nullptr;
		int s = device_alloc(ctx, fb, bytes); if (s) return s;
		RT_HIP(ctx, hipMemset(*fb, 0, bytes));
		if (slot_index == 0) ctx->params.aovs[i].framebuffer = (float4 *)*fb;
	}
	slot.aov_samples = samples;
	return RT_OK;
}

static int sync_svgf(rt_context * ctx) {
	bool want = ctx->params.config.enable_svgf != 0;
	if (want == ctx->svgf_allocated || ctx->frame_pixels == 0) return RT_OK;
	if (want) {
		// gbuffers (float4, int2, float2), moment, history x5 (length is int), taa x2, decoded normal + depth, variance pairs x2
		const size_t elem[15] = { 16, 8, 8, 16, 4, 16, 16, 16, 16, 16, 16, 16, 8, 8, 16 };
		for (int i = 0; i < 15; i++) {
			int s = device_alloc(ctx, &ctx->svgf_buffers[i], ctx->frame_pixels * elem[i]); if (s) return s;
			RT_HIP(ctx, hipMemsetAsync(ctx->svgf_buffers[i], 0, ctx->frame_pixels * elem[i], ctx->stream));
		}
		{ int s = device_alloc(ctx, &ctx->svgf_buffers[15], (RT_SVGF_YOUNG_HEADER + ctx->frame_pixels) * 4); if (s) return s; RT_HIP(ctx, hipMemsetAsync(ctx->svgf_buffers[15], 0, RT_SVGF_YOUNG_HEADER * 4, ctx->stream)); }
		for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) if (ctx->slots[k].created) for (int i = 0; i < 3; i++) {
			int s = device_alloc(ctx, &ctx->slots[k].gbuffers[i], ctx->frame_pixels * elem[i]); if (s) return s;
			RT_HIP(ctx, hipMemsetAsync(ctx->slots[k].gbuffers[i], 0, ctx->frame_pixels * elem[i], ctx->stream));
		}
		// The filter keeps (direct.w, indirect.w) of the radiance accumulators mirrored in svgf_variance[1] and writes both only where it
		// filters (not at sky pixels). The mirror starts at zero, so the accumulators' .w have to: radiance accumulated WITHOUT the filter
		// before it was switched on would leave stale .w at the sky pixels the variance blur taps across a silhouette. With SVGF on the
		// two accumulators are the filter's ping-pong images (SVGF.h:416-554) and the sample count restarts, so nothing is lost.
		RT_HIP(ctx, quiesce(ctx));
		for (int aov : { RT_AOV_RADIANCE_DIRECT, RT_AOV_RADIANCE_INDIRECT })
			if (ctx->aov_buffers[aov][1]) RT_HIP(ctx, hipMemsetAsync(ctx->aov_buffers[aov][1], 0, ctx->frame_pixels * 16, ctx->stream));
	} else {
		RT_HIP(ctx, quiesce(ctx));
		for (int i = 0; i < 16; i++) { device_free(ctx, ctx->svgf_buffers[i]); ctx->svgf_buffers[i] = nullptr; }
		for (SampleSlot & slot : ctx->slots) for (int i = 0; i < 3; i++) { device_free(ctx, slot.gbuffers[i]); slot.gbuffers[i] = nullptr; }
		ctx->path_stream.last_gbuffer_slot = -1;
		for (void * & g : ctx->path_stream.gbuffers) { device_free(ctx, g); g = nullptr; }
	}
	ctx->svgf_allocated = want;
	RtParams & p = ctx->params;
	p.gbuffer_normal_and_depth        = (float4 *)ctx->svgf_buffers[0];
	p.gbuffer_mesh_id_and_triangle_id = (int2   *)ctx->svgf_buffers[1];
	p.gbuffer_screen_position_prev    = (float2 *)ctx->svgf_buffers[2];
	p.frame_buffer_moment             = (float4 *)ctx->svgf_buffers[3];
	p.history_length                  = (int    *)ctx->svgf_buffers[4];
	p.history_direct                  = (float4 *)ctx->svgf_buffers[5];
	p.history_indirect                = (float4 *)ctx->svgf_buffers[6];
	p.history_moment                  = (float4 *)ctx->svgf_buffers[7];
	p.history_normal_and_depth        = (float4 *)ctx->svgf_buffers[8];
	p.taa_frame_prev                  = (float4 *)ctx->svgf_buffers[9];
	p.taa_frame_curr                  = (float4 *)ctx->svgf_buffers[10];
	p.svgf_normal_and_depth           = (float4 *)ctx->svgf_buffers[11];
	p.svgf_variance[0]                = (float2 *)ctx->svgf_buffers[12];
	p.svgf_variance[1]                = (float2 *)ctx->svgf_buffers[13];
	p.taa_frame_next                  = (float4 *)ctx->svgf_buffers[14];   // (prev and next trade places after every filtered frame: kernel_taa)
	p.svgf_young_pixels               = (int    *)ctx->svgf_buffers[15];
	return RT_OK;
}

int rt_resize(rt_context * ctx, int width, int height) {
	RT_REQUIRE(ctx, ctx && width > 0 && height > 0, "rt_resize: invalid size");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	int pitch = (width + 31) / 32 * 32; // Math::round_up(width, WARP_SIZE), Pathtracer.cpp:258
	ctx->params.screen_width = width; ctx->params.screen_height = height; ctx->params.screen_pitch = pitch;
	ctx->frame_pixels = size_t(pitch) * height;
	ctx->path_stream.last_gbuffer_slot = -1; // the filter's histories start over with the new size
	stream_release_frames(ctx);

	for (int i = 0; i < RT_AOV_COUNT; i++) for (int k = 0; k < 2; k++) { device_free(ctx, ctx->aov_buffers[i][k]); ctx->aov_buffers[i][k] = nullptr; }
	for (int i = 0; i < RT_AOV_COUNT; i++) for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) { device_free(ctx, ctx->slots[k].aov_framebuffer[i]); ctx->slots[k].aov_framebuffer[i] = nullptr; }
	for (SampleSlot & slot : ctx->slots) slot.aov_samples = 1;
	ctx->params.frame_pixels = unsigned(ctx->frame_pixels);
	ctx->params.frame_pixels_magic = unsigned((1ull << 32) / ctx->frame_pixels) + 1u;
	for (size_t i = 0; i < sizeof(ctx->svgf_buffers) / sizeof(*ctx->svgf_buffers); i++) { device_free(ctx, ctx->svgf_buffers[i]); ctx->svgf_buffers[i] = nullptr; }   // (all 16: the third TAA image [14] and the young-pixel list [15] used to stay behind, 20 B per pixel per resize)
	for (SampleSlot & slot : ctx->slots) for (int i = 0; i < 3; i++) { device_free(ctx, slot.gbuffers[i]); slot.gbuffers[i] = nullptr; }
	ctx->svgf_allocated = false;
	device_free(ctx, ctx->final_image); ctx->final_image = nullptr;
	int s = device_alloc(ctx, &ctx->final_image, ctx->frame_pixels * 16); if (s) return s;
	RT_HIP(ctx, hipMemsetAsync(ctx->final_image, 0, ctx->frame_pixels * 16, ctx->stream));
	ctx->params.final_image = (float4 *)ctx->final_image;
	if ((s = sync_aovs(ctx))) return s;
	if ((s = sync_svgf(ctx))) return s;
	return RT_OK;
}

int rt_set_camera(rt_context * ctx, const rt_camera * camera) {
	RT_REQUIRE(ctx, ctx && camera, "rt_set_camera: NULL argument");
	if (memcmp(&ctx->params.camera, camera, sizeof(*camera)) != 0) { int status = stream_launch_pending(ctx); if (status) return status; } // bounce 0 is shaded with the camera
	ctx->params.camera = *camera;
	return RT_OK;
}

int rt_set_svgf_matrices(rt_context * ctx, const float * view_projection, const float * view_projection_prev) {
	RT_REQUIRE(ctx, ctx && view_projection && view_projection_prev, "rt_set_svgf_matrices: NULL argument");
	memcpy(ctx->params.view_projection,      view_projection,      64);
	memcpy(ctx->params.view_projection_prev, view_projection_prev, 64);
	return RT_OK;
}

int rt_set_svgf_tiles(rt_context * ctx, int enable) {
	RT_REQUIRE(ctx, ctx, "rt_set_svgf_tiles: NULL context");
	ctx->params.svgf_tiles = enable != 0;   // (read at launch time: the filter launches of frames already enqueued keep what they were enqueued with)
	return RT_OK;
}

int rt_set_config(rt_context * ctx, const rt_gpu_config * config) {
	RT_REQUIRE(ctx, ctx && config, "rt_set_config: NULL argument");
	RT_REQUIRE(ctx, config->num_bounces >= 0 && config->num_bounces <= RT_MAX_BOUNCES, "rt_set_config: num_bounces out of range");
	RT_REQUIRE(ctx, config->num_atrous_iterations >= 0 && config->num_atrous_iterations <= RT_MAX_ATROUS_ITERATIONS, "rt_set_config: num_atrous_iterations out of range");
	(void)hipSetDevice(ctx->device);
	if (ctx->path_stream.created && memcmp(&ctx->params.config, config, sizeof(*config)) != 0) RT_HIP(ctx, stream_flush(ctx)); // the submissions in flight were made under the old settings
	ctx->params.config = *config;
	ctx->params.config.aov_mask |= 1u << RT_AOV_RADIANCE;
	if (config->enable_svgf) ctx->params.config.aov_mask |= (1u << RT_AOV_RADIANCE_DIRECT) | (1u << RT_AOV_RADIANCE_INDIRECT) | (1u << RT_AOV_ALBEDO);
	int s = sync_aovs(ctx); if (s) return s;
	return sync_svgf(ctx);
}

int rt_set_pixel_range(rt_context * ctx, int pixel_offset, int pixel_count) {
	RT_REQUIRE(ctx, ctx && pixel_offset >= 0, "rt_set_pixel_range: invalid argument");
	ctx->pixel_offset = pixel_offset;
	ctx->pixel_count  = pixel_count;
	ctx->params.tile_pixels = 0;
	return RT_OK;
}

int rt_set_pixel_tiles(rt_context * ctx, int tile_pixels, int first_tile, int tile_stride) {
	RT_REQUIRE(ctx, ctx && tile_pixels > 0 && first_tile >= 0 && tile_stride > 0 && first_tile < tile_stride, "rt_set_pixel_tiles: invalid argument");
	ctx->params.tile_pixels = tile_pixels;
	ctx->params.tile_first  = first_tile;
	ctx->params.tile_stride = tile_stride;
	return RT_OK;
}

int rt_pack_pixels(rt_context * ctx, void * dst_device, int tile_pixels, int first_tile, int tile_stride, int tiles) {
	RT_REQUIRE(ctx, ctx && dst_device && tile_pixels > 0 && tile_stride > 0 && tiles >= 0, "rt_pack_pixels: invalid argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->final_image) return fail(ctx, RT_ERROR_NOT_READY, "rt_pack_pixels: rt_resize was not called");
	RT_HIP(ctx, main_waits_for_samples(ctx));
	rt_launch_pack_pixels(ctx->params, (float4 *)dst_device, tile_pixels, first_tile, tile_stride, tiles, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

int rt_unpack_pixels(rt_context * ctx, const void * src_device, int tile_pixels, int world, int tiles_per_rank) {
	RT_REQUIRE(ctx, ctx && src_device && tile_pixels > 0 && world > 0 && tiles_per_rank > 0, "rt_unpack_pixels: invalid argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->final_image) return fail(ctx, RT_ERROR_NOT_READY, "rt_unpack_pixels: rt_resize was not called");
	RT_HIP(ctx, main_waits_for_samples(ctx));
	rt_launch_unpack_pixels(ctx->params, (const float4 *)src_device, tile_pixels, world, tiles_per_rank, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

// ---- SVGF frames under the tile split ---------------------------------------------------------------
int rt_render_sample_unfiltered(rt_context * ctx, int sample_index) {
	RT_REQUIRE(ctx, ctx, "rt_render_sample_unfiltered: NULL context");
	if (!ctx->params.config.enable_svgf) return fail(ctx, RT_ERROR_INVALID_ARG, "rt_render_sample_unfiltered: SVGF is not enabled");
	ctx->defer_filter = true;
	int status = rt_render_samples(ctx, sample_index, 1);
	ctx->defer_filter = false;
	return status;
}

int rt_pack_svgf_inputs(rt_context * ctx, void * dst_device, int tile_pixels, int first_tile, int tile_stride, int tiles) {
	RT_REQUIRE(ctx, ctx && dst_device && tile_pixels > 0 && tile_stride > 0 && tiles >= 0, "rt_pack_svgf_inputs: invalid argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->svgf_allocated) return fail(ctx, RT_ERROR_NOT_READY, "rt_pack_svgf_inputs: SVGF is not enabled");
	RT_HIP(ctx, main_waits_for_samples(ctx));
	rt_launch_pack_svgf(slot_params(ctx, ctx->slots[0], 0), (float4 *)dst_device, tile_pixels, first_tile, tile_stride, tiles, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

int rt_unpack_svgf_inputs(rt_context * ctx, const void * src_device, int tile_pixels, int world, int tiles_per_rank) {
	RT_REQUIRE(ctx, ctx && src_device && tile_pixels > 0 && world > 0 && tiles_per_rank > 0, "rt_unpack_svgf_inputs: invalid argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->svgf_allocated) return fail(ctx, RT_ERROR_NOT_READY, "rt_unpack_svgf_inputs: SVGF is not enabled");
	RT_HIP(ctx, main_waits_for_samples(ctx));
	rt_launch_unpack_svgf(slot_params(ctx, ctx->slots[0], 0), (const float4 *)src_device, tile_pixels, world, tiles_per_rank, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

int rt_filter_frame(rt_context * ctx, int sample_index) {
	RT_REQUIRE(ctx, ctx, "rt_filter_frame: NULL context");
	(void)hipSetDevice(ctx->device);
	if (!ctx->params.config.enable_svgf || !ctx->svgf_allocated) return fail(ctx, RT_ERROR_NOT_READY, "rt_filter_frame: SVGF is not enabled");
	SampleSlot & slot = ctx->slots[0];
	RtParams p = slot_params(ctx, slot, 0);
	p.batch_samples = 1;
	hipStream_t st = slot.stream;
	RT_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));      // after the scatter of the gathered tiles (main stream)
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_main, 0));
	rt_launch_svgf_taa(p, sample_index, st);
	if (p.config.enable_taa) std::swap(ctx->params.taa_frame_prev, ctx->params.taa_frame_next);   // (the snapshot that launched kernel_taa decides, not a config set meanwhile)
	for (int i = 0; i < RT_AOV_COUNT; i++) if (p.aovs[i].framebuffer) RT_HIP(ctx, hipMemsetAsync(p.aovs[i].framebuffer, 0, ctx->frame_pixels * 16, st)); // aovs_clear_to_zero
	RT_HIP(ctx, hipEventRecord(slot.ev_done, st));
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

// ---- frame exchange of the tile split without Python ----------------------------------------------------
// One communicator per context. RCCL is bound at RUN TIME (dlopen, RTLD_LOCAL): a process that also hosts PyTorch already
// has torch's own copy of librccl mapped, and a link-time dependency would make every user of this library load a
// collective library most of them never call. Contexts that share a GPU (tests; RCCL refuses a device twice in one
// communicator) exchange by stream-ordered peer copies instead -- the same pack / unpack kernels either way.
extern "C++" {
namespace {
struct RcclUniqueId { char internal[128]; };            // ncclUniqueId (rccl.h)
enum { RCCL_FLOAT32 = 7 };                              // ncclFloat32
struct RcclApi {
	void * handle = nullptr;
	int (*get_unique_id)(RcclUniqueId *) = nullptr;
	int (*comm_init_rank)(void **, int, RcclUniqueId, int) = nullptr;
	int (*comm_init_all)(void **, int, const int *) = nullptr;
	int (*comm_destroy)(void *) = nullptr;
	int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
	int (*group_start)() = nullptr; int (*group_end)() = nullptr;
	const char * (*error_string)(int) = nullptr;
};
RcclApi * rccl_api(std::string & why) {
	static RcclApi api; static bool tried = false; static std::string failure;
	if (!tried) {
		tried = true;
		// GRT_COLLECTIVE_LIBRARY: another library with RCCL's entry points, tried first. tests/support/libloopback_ccl.so uses it to run this very code with two
		// ranks on a box with one GPU (RCCL refuses a device twice); a deployment could name a site's own RCCL build the same way.
		if (const char * named = getenv("GRT_COLLECTIVE_LIBRARY")) { if (named[0] && !(api.handle = dlopen(named, RTLD_NOW | RTLD_LOCAL))) failure = std::string("GRT_COLLECTIVE_LIBRARY: ") + dlerror(); }
		if (!api.handle && failure.empty()) for (const char * name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) if ((api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
		if (!api.handle) { if (failure.empty()) failure = std::string("librccl.so not found: ") + dlerror(); }
		else {
			#define RT_BIND(member, symbol) { *(void **)&api.member = dlsym(api.handle, symbol); if (!api.member) failure = std::string("librccl.so lacks ") + symbol; }
			RT_BIND(get_unique_id, "ncclGetUniqueId") RT_BIND(comm_init_rank, "ncclCommInitRank") RT_BIND(comm_init_all, "ncclCommInitAll") RT_BIND(comm_destroy, "ncclCommDestroy")
			RT_BIND(all_gather, "ncclAllGather") RT_BIND(group_start, "ncclGroupStart") RT_BIND(group_end, "ncclGroupEnd") RT_BIND(error_string, "ncclGetErrorString")
			#undef RT_BIND
		}
	}
	why = failure;
	return failure.empty() ? &api : nullptr;
}
// this rank's share of the frame in float4 pixels, padded so that every rank sends the same amount
size_t exchange_tiles_per_rank(const rt_context * ctx, int tile_pixels, int world) {
	size_t frame = size_t(ctx->params.screen_width) * ctx->params.screen_height;
	size_t tiles = (frame + tile_pixels - 1) / tile_pixels;
	return (tiles + world - 1) / world;
}
int exchange_buffers(rt_context * ctx, size_t pixels) {
	rt_context::FrameExchange & x = ctx->exchange;
	if (x.packed_pixels >= pixels) return RT_OK;
	if (x.packed)   device_free(ctx, x.packed);
	if (x.gathered) device_free(ctx, x.gathered);
	x.packed = x.gathered = nullptr; x.packed_pixels = 0;
	int status = device_alloc(ctx, (void **)&x.packed, pixels * 16); if (status) return status;
	status = device_alloc(ctx, (void **)&x.gathered, pixels * 16 * size_t(x.world)); if (status) return status;
	x.packed_pixels = pixels;
	if (!x.ev_packed) { RT_HIP(ctx, hipEventCreateWithFlags(&x.ev_packed, hipEventDisableTiming)); RT_HIP(ctx, hipEventCreateWithFlags(&x.ev_copied, hipEventDisableTiming)); }
	return RT_OK;
}
} // namespace
} // extern "C++"

int rt_comm_unique_id(void * out_id_128_bytes) {
	if (!out_id_128_bytes) return RT_ERROR_INVALID_ARG;
	std::string why; RcclApi * api = rccl_api(why);
	if (!api) return RT_ERROR_NOT_READY;
	return api->get_unique_id((RcclUniqueId *)out_id_128_bytes) == 0 ? RT_OK : RT_ERROR_HIP;
}

int rt_comm_init_rank(rt_context * ctx, const void * unique_id_128_bytes, int rank, int world) {
	RT_REQUIRE(ctx, ctx && unique_id_128_bytes && world >= 1 && rank >= 0 && rank < world, "rt_comm_init_rank: invalid argument");
	(void)hipSetDevice(ctx->device);
	(void)rt_comm_destroy(ctx);
	std::string why; RcclApi * api = rccl_api(why);
	if (!api) return fail(ctx, RT_ERROR_NOT_READY, "rt_comm_init_rank: %s", why.c_str());
	RcclUniqueId id; memcpy(&id, unique_id_128_bytes, sizeof(id));
	int rc = api->comm_init_rank(&ctx->exchange.comm, world, id, rank);
	if (rc != 0) { ctx->exchange.comm = nullptr; return fail(ctx, RT_ERROR_HIP, "rt_comm_init_rank: ncclCommInitRank: %s", api->error_string(rc)); }
	ctx->exchange.rank = rank; ctx->exchange.world = world;
	return RT_OK;
}

int rt_comm_init_all(rt_context ** contexts, int count) {
	if (!contexts || count < 1) return RT_ERROR_INVALID_ARG;
	for (int i = 0; i < count; i++) if (!contexts[i]) return RT_ERROR_INVALID_ARG;
	bool distinct = true;
	for (int i = 0; i < count; i++) for (int j = 0; j < i; j++) if (contexts[i]->device == contexts[j]->device) distinct = false;
	for (int i = 0; i < count; i++) { (void)rt_comm_destroy(contexts[i]); contexts[i]->exchange.rank = i; contexts[i]->exchange.world = count; }
	if (distinct && count > 1) {   // one communicator over the GPUs of this process (ncclCommInitAll)
		std::string why; RcclApi * api = rccl_api(why);
		if (!api) return fail(contexts[0], RT_ERROR_NOT_READY, "rt_comm_init_all: %s", why.c_str());
		std::vector<void *> comms(count, nullptr); std::vector<int> devices(count);
		for (int i = 0; i < count; i++) devices[i] = contexts[i]->device;
		int rc = api->comm_init_all(comms.data(), count, devices.data());
		if (rc != 0) return fail(contexts[0], RT_ERROR_HIP, "rt_comm_init_all: ncclCommInitAll: %s", api->error_string(rc));
		for (int i = 0; i < count; i++) contexts[i]->exchange.comm = comms[i];
	} else {                       // contexts sharing a GPU: stream-ordered copies between them
		for (int i = 0; i < count; i++) contexts[i]->exchange.peers.assign(contexts, contexts + count);
	}
	return RT_OK;
}

int rt_comm_destroy(rt_context * ctx) {
	if (!ctx) return RT_ERROR_INVALID_ARG;
	rt_context::FrameExchange & x = ctx->exchange;
	if (x.comm) { std::string why; if (RcclApi * api = rccl_api(why)) (void)api->comm_destroy(x.comm); x.comm = nullptr; }
	for (rt_context * peer : x.peers) if (peer && peer != ctx) {   // the others of an in-process group lose this member
		for (rt_context *& p : peer->exchange.peers) if (p == ctx) p = nullptr;
	}
	x.peers.clear();
	if (x.ev_packed || x.packed) {   // (a later group may have another world size: its buffers and events are made again, exchange_buffers)
		(void)hipSetDevice(ctx->device);
		(void)hipStreamSynchronize(ctx->stream);
	}
	if (x.ev_packed) { (void)hipEventDestroy(x.ev_packed); (void)hipEventDestroy(x.ev_copied); x.ev_packed = x.ev_copied = nullptr; }
	if (x.packed)   device_free(ctx, x.packed);
	if (x.gathered) device_free(ctx, x.gathered);
	x.packed = x.gathered = nullptr; x.packed_pixels = 0;
	x.rank = 0; x.world = 1;
	return RT_OK;
}

// what: 0 = the final image (1 float4 per pixel), 1 = the inputs of the SVGF filter stage (5 float4 per pixel)
static int exchange_group(rt_context ** contexts, int count, int what) {
	const int channels = what == 0 ? 1 : 5;
	std::string why; RcclApi * api = nullptr;
	// pack: every context's own tiles (the tile layout is the one rt_set_pixel_tiles gave it)
	for (int i = 0; i < count; i++) {
		rt_context * ctx = contexts[i];
		rt_context::FrameExchange & x = ctx->exchange;
		RT_REQUIRE(ctx, x.world == count || x.comm, "rt_all_gather: the contexts are not one communicator group");
		RT_REQUIRE(ctx, ctx->params.tile_pixels > 0 && ctx->params.tile_stride == x.world && ctx->params.tile_first == x.rank, "rt_all_gather: rt_set_pixel_tiles(tile_pixels, rank, world) first");
		(void)hipSetDevice(ctx->device);
		const size_t per_rank = exchange_tiles_per_rank(ctx, ctx->params.tile_pixels, x.world) * size_t(ctx->params.tile_pixels) * channels;
		int status = exchange_buffers(ctx, per_rank); if (status) return status;
		RT_HIP(ctx, main_waits_for_samples(ctx));
		if (!x.peers.empty()) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, x.ev_copied, 0));   // the peers have read the previous frame's tiles (see below)
		const int tiles = int(per_rank / size_t(ctx->params.tile_pixels) / channels);
		if (what == 0) rt_launch_pack_pixels(ctx->params, x.packed, ctx->params.tile_pixels, x.rank, x.world, tiles, ctx->stream);
		else { if (!ctx->svgf_allocated) return fail(ctx, RT_ERROR_NOT_READY, "rt_all_gather_svgf_inputs: SVGF is not enabled"); rt_launch_pack_svgf(slot_params(ctx, ctx->slots[0], 0), x.packed, ctx->params.tile_pixels, x.rank, x.world, tiles, ctx->stream); }
		RT_HIP(ctx, hipEventRecord(x.ev_packed, ctx->stream));
		if (x.comm && !api) { api = rccl_api(why); if (!api) return fail(ctx, RT_ERROR_NOT_READY, "rt_all_gather: %s", why.c_str()); }
	}
	// exchange
	if (api) {
		if (count > 1) api->group_start();
		for (int i = 0; i < count; i++) {
			rt_context * ctx = contexts[i]; rt_context::FrameExchange & x = ctx->exchange;
			(void)hipSetDevice(ctx->device);
			const size_t per_rank = exchange_tiles_per_rank(ctx, ctx->params.tile_pixels, x.world) * size_t(ctx->params.tile_pixels) * channels;
			int rc = api->all_gather(x.packed, x.gathered, per_rank * 4, RCCL_FLOAT32, x.comm, ctx->stream);
			if (rc != 0) { if (count > 1) api->group_end(); return fail(ctx, RT_ERROR_HIP, "rt_all_gather: ncclAllGather: %s", api->error_string(rc)); }
		}
		if (count > 1) { int rc = api->group_end(); if (rc != 0) return fail(contexts[0], RT_ERROR_HIP, "rt_all_gather: ncclGroupEnd: %s", api->error_string(rc)); }
	} else {
		for (int i = 0; i < count; i++) {
			rt_context * ctx = contexts[i]; rt_context::FrameExchange & x = ctx->exchange;
			RT_REQUIRE(ctx, int(x.peers.size()) == x.world && count == x.world, "rt_all_gather: an in-process group exchanges all its contexts in one call (rt_all_gather_framebuffers)");
			(void)hipSetDevice(ctx->device);
			const size_t per_rank = exchange_tiles_per_rank(ctx, ctx->params.tile_pixels, x.world) * size_t(ctx->params.tile_pixels) * channels;
			for (int r = 0; r < x.world; r++) {
				rt_context * peer = x.peers[r];
				RT_REQUIRE(ctx, peer && peer->exchange.packed_pixels >= per_rank, "rt_all_gather: a member of the group is gone");
				RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, peer->exchange.ev_packed, 0));
				if (peer->device == ctx->device) RT_HIP(ctx, hipMemcpyAsync(x.gathered + size_t(r) * per_rank, peer->exchange.packed, per_rank * 16, hipMemcpyDeviceToDevice, ctx->stream));
				else RT_HIP(ctx, hipMemcpyPeerAsync(x.gathered + size_t(r) * per_rank, ctx->device, peer->exchange.packed, peer->device, per_rank * 16, ctx->stream));
			}
		}
		// a context may pack its next frame only when every peer has copied this one: one event per context, recorded on a
		// stream that has waited for all the copies (its own stream does: the peers' copy streams are joined through ev_packed
		// of the NEXT round only, so join them here explicitly)
		for (int i = 0; i < count; i++) {
			rt_context * ctx = contexts[i];
			(void)hipSetDevice(ctx->device);
			RT_HIP(ctx, hipEventRecord(ctx->ev_interop, ctx->stream));
		}
		for (int i = 0; i < count; i++) {
			rt_context * ctx = contexts[i];
			(void)hipSetDevice(ctx->device);
			for (int r = 0; r < count; r++) if (r != i) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, contexts[r]->ev_interop, 0));
			RT_HIP(ctx, hipEventRecord(ctx->exchange.ev_copied, ctx->stream));
		}
	}
	// unpack: the gathered tiles become every context's whole frame
	for (int i = 0; i < count; i++) {
		rt_context * ctx = contexts[i]; rt_context::FrameExchange & x = ctx->exchange;
		(void)hipSetDevice(ctx->device);
		const int tiles = int(exchange_tiles_per_rank(ctx, ctx->params.tile_pixels, x.world));
		if (what == 0) rt_launch_unpack_pixels(ctx->params, x.gathered, ctx->params.tile_pixels, x.world, tiles, ctx->stream);
		else rt_launch_unpack_svgf(slot_params(ctx, ctx->slots[0], 0), x.gathered, ctx->params.tile_pixels, x.world, tiles, ctx->stream);
		RT_HIP(ctx, hipGetLastError());
	}
	return RT_OK;
}

int rt_all_gather_framebuffer(rt_context * ctx) {
	RT_REQUIRE(ctx, ctx, "rt_all_gather_framebuffer: NULL context");
	if (ctx->exchange.world == 1 && !ctx->exchange.comm) return RT_OK;   // (a 1-rank communicator does run its ncclAllGather: tests/test_gpu_rccl.py)
	RT_REQUIRE(ctx, ctx->exchange.comm, "rt_all_gather_framebuffer: no communicator (rt_comm_init_rank), or an in-process group (use rt_all_gather_framebuffers)");
	return exchange_group(&ctx, 1, 0);
}
int rt_all_gather_framebuffers(rt_context ** contexts, int count) {
	if (!contexts || count < 1) return RT_ERROR_INVALID_ARG;
	if (count == 1 && contexts[0] && contexts[0]->exchange.world == 1) return RT_OK;
	return exchange_group(contexts, count, 0);
}
int rt_all_gather_svgf_inputs(rt_context ** contexts, int count) {
	if (!contexts || count < 1) return RT_ERROR_INVALID_ARG;
	if (count == 1 && contexts[0] && contexts[0]->exchange.world == 1) return RT_OK;
	return exchange_group(contexts, count, 1);
}

int rt_stream_wait_for_context(rt_context * ctx, void * stream) {
	RT_REQUIRE(ctx, ctx, "rt_stream_wait_for_context: NULL context");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, hipEventRecord(ctx->ev_interop, ctx->stream));
	RT_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_interop, 0));
	return RT_OK;
}

int rt_context_wait_for_stream(rt_context * ctx, void * stream) {
	RT_REQUIRE(ctx, ctx, "rt_context_wait_for_stream: NULL context");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, hipEventRecord(ctx->ev_interop, (hipStream_t)stream));
	RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_interop, 0));
	return RT_OK;
}

int rt_set_trace_statistics(rt_context * ctx, int enable) {
	RT_REQUIRE(ctx, ctx, "rt_set_trace_statistics: NULL context");
	(void)hipSetDevice(ctx->device);
	if (enable && !ctx->trace_stats) { int s = device_alloc(ctx, (void **)&ctx->trace_stats, 10 * sizeof(unsigned long long)); if (s) return s; }
	{ int s = stream_launch_pending(ctx); if (s) return s; }
	ctx->trace_statistics = enable != 0;
	return RT_OK;
}

int rt_get_trace_statistics(rt_context * ctx, uint64_t * out10) {
	RT_REQUIRE(ctx, ctx && out10, "rt_get_trace_statistics: NULL argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->trace_stats) return fail(ctx, RT_ERROR_NOT_READY, "rt_get_trace_statistics: statistics were never enabled");
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemcpy(out10, ctx->trace_stats, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_set_batch_size(rt_context * ctx, int batch_size) {
	RT_REQUIRE(ctx, ctx && batch_size >= 0, "rt_set_batch_size: invalid argument");
	ctx->batch_size_request = batch_size;
	return RT_OK;
}

int rt_set_profiling(rt_context * ctx, int enable) {
	RT_REQUIRE(ctx, ctx, "rt_set_profiling: NULL context");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	ctx->profiling = enable == 1;
	ctx->launch_timing = enable == 2 || enable == 3;
	ctx->launch_timing_all = enable == 3;
	ctx->span_used = 0;
	ctx->stage_used = 0;
	return RT_OK;
}

int rt_get_launch_timings(rt_context * ctx, int kind, float * out_ms, int capacity, int * out_count) {
	RT_REQUIRE(ctx, ctx && out_count && (out_ms || capacity == 0) && kind >= 0 && kind < RT_TIMING_KINDS, "rt_get_launch_timings: invalid argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	int n = 0;
	for (size_t i = 0; i + 1 < ctx->span_used; i += 2) {
		if (ctx->span_kinds[i] != kind) continue;
		float d = 0.0f;
		if (hipEventElapsedTime(&d, ctx->span_events[i], ctx->span_events[i + 1]) != hipSuccess) continue;
		if (n < capacity) out_ms[n] = d;
		n++;
	}
	*out_count = n;
	if (n <= capacity) { // everything delivered: start over (a call with too small a buffer only reports the count)
		size_t kept = 0;
		for (size_t i = 0; i + 1 < ctx->span_used; i += 2) if (ctx->span_kinds[i] != kind) {
			std::swap(ctx->span_events[kept], ctx->span_events[i]); std::swap(ctx->span_events[kept + 1], ctx->span_events[i + 1]);
			ctx->span_kinds[kept] = ctx->span_kinds[i]; ctx->span_kinds[kept + 1] = ctx->span_kinds[i + 1];
			kept += 2;
		}
		ctx->span_used = kept;
	}
	return RT_OK;
}

int rt_set_scheduler(rt_context * ctx, int scheduler) {
	RT_REQUIRE(ctx, ctx && (scheduler == RT_SCHEDULER_MERGED || scheduler == RT_SCHEDULER_SLOTS), "rt_set_scheduler: unknown scheduler");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	ctx->scheduler = scheduler;
	return RT_OK;
}

int rt_set_texture_expansion(rt_context * ctx, int enable) {
	RT_REQUIRE(ctx, ctx, "rt_set_texture_expansion: NULL context");
	ctx->expand_bc1 = enable != 0;   // takes effect at the next rt_upload_textures
	return RT_OK;
}
size_t rt_texture_bytes(rt_context * ctx) { return ctx ? ctx->texture_bytes : 0; }

int rt_set_frame_pipelining(rt_context * ctx, int enable) {
	RT_REQUIRE(ctx, ctx, "rt_set_frame_pipelining: NULL context");
	ctx->frame_pipelining = enable != 0;
	return RT_OK;
}

int rt_set_stream_batch(rt_context * ctx, long long paths) {
	RT_REQUIRE(ctx, ctx && paths >= 0, "rt_set_stream_batch: invalid argument");
	ctx->stream_batch_paths = paths;   // (read by the next submission; what already waits for company is enqueued by whatever needs progress, as always)
	return RT_OK;
}

int rt_get_trace_statistics_history(rt_context * ctx, uint64_t * out_rows10, int capacity_rows, int * out_rows) {
	RT_REQUIRE(ctx, ctx && out_rows && (out_rows10 || capacity_rows == 0), "rt_get_trace_statistics_history: invalid argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	int rows = ctx->stream_history ? ctx->stream_history_rows : 0;
	*out_rows = rows;
	if (rows > capacity_rows) rows = capacity_rows;
	if (rows > 0) memcpy(out_rows10, ctx->stream_history, size_t(rows) * 10 * sizeof(uint64_t));
	return RT_OK;
}

// ---- render ------------------------------------------------------------------------------------------------

static int ensure_luts(rt_context * ctx) {
	if (ctx->luts_ready) return RT_OK;
	const size_t bytes[6] = { 4096 * 4, 4096 * 4, 256 * 4, 256 * 4, 1024 * 4, 32 * 4 };
	for (int i = 0; i < 6; i++) if (!ctx->luts[i]) { int s = device_alloc(ctx, &ctx->luts[i], bytes[i]); if (s) return s; }
	RtParams & p = ctx->params;
	rt_launch_integrate_luts(p, (float *)ctx->luts[0], (float *)ctx->luts[1], (float *)ctx->luts[2], (float *)ctx->luts[3], (float *)ctx->luts[4], (float *)ctx->luts[5], ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	p.lut_dielectric_directional_albedo_enter = (const float *)ctx->luts[0];
	p.lut_dielectric_directional_albedo_leave = (const float *)ctx->luts[1];
	p.lut_dielectric_albedo_enter             = (const float *)ctx->luts[2];
	p.lut_dielectric_albedo_leave             = (const float *)ctx->luts[3];
	p.lut_conductor_directional_albedo        = (const float *)ctx->luts[4];
	p.lut_conductor_albedo                    = (const float *)ctx->luts[5];
	ctx->luts_ready = true;
	RT_HIP(ctx, quiesce(ctx)); // once per context: the sample streams read the tables without further ordering
	return RT_OK;
}


static void stage_mark(rt_context * ctx, int kind, hipStream_t stream) {
	if (!ctx->profiling) return;
	if (ctx->stage_used == ctx->stage_events.size()) { hipEvent_t e; (void)hipEventCreate(&e); ctx->stage_events.push_back(e); ctx->stage_kinds.push_back(0); }
	ctx->stage_kinds[ctx->stage_used] = kind;
	(void)hipEventRecord(ctx->stage_events[ctx->stage_used++], stream);
}

// mode 2: an event on the launch's own stream before and after it (does not order other streams)
static void span_mark(rt_context * ctx, int kind, hipStream_t stream) {
	if (!ctx->time_this_sample) return;
	if (kind != SPAN_TRACE && kind != SPAN_SHADOW && !ctx->launch_timing_all) return;
	if (ctx->span_used == ctx->span_events.size()) { hipEvent_t e; (void)hipEventCreate(&e); ctx->span_events.push_back(e); ctx->span_kinds.push_back(0); }
	ctx->span_kinds[ctx->span_used] = kind;
	(void)hipEventRecord(ctx->span_events[ctx->span_used++], stream);
}

// rt_launch_svgf_taa calls this before and after each of its kernels (mode 3)
static void svgf_span_mark(void * user, int svgf_kernel, hipStream_t stream) { span_mark((rt_context *)user, SPAN_SVGF + svgf_kernel, stream); }

__global__ void kernel_accumulate_counters(const RtBufferSizes * sizes, int * totals) {
	int b = threadIdx.x;
	if (b >= RT_MAX_BOUNCES) return;
	totals[0 * RT_MAX_BOUNCES + b] += sizes->trace[b];
	totals[1 * RT_MAX_BOUNCES + b] += sizes->shadow[b];
	totals[2 * RT_MAX_BOUNCES + b] += sizes->diffuse[b];
	totals[3 * RT_MAX_BOUNCES + b] += sizes->plastic[b];
	totals[4 * RT_MAX_BOUNCES + b] += sizes->dielectric[b];
	totals[5 * RT_MAX_BOUNCES + b] += sizes->conductor[b];
}


} // extern "C" -- the merged-wavefront scheduler below is internal

// ---- merged wavefront: host side ---------------------------------------------------------------------------------

static int stream_create(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	if (s.created) return RT_OK;
	if (!ctx->stream_history) RT_HIP(ctx, hipHostMalloc((void **)&ctx->stream_history, sizeof(unsigned long long) * 10 * RT_STREAM_HISTORY_ROWS));
	memset(s.trace, 0, sizeof(s.trace)); memset(s.material, 0, sizeof(s.material)); memset(&s.shadow, 0, sizeof(s.shadow));
	RT_HIP(ctx, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
	RT_HIP(ctx, hipEventCreateWithFlags(&s.ev_idle, hipEventDisableTiming));
	RT_HIP(ctx, hipEventRecord(s.ev_idle, s.stream));
	int status = device_alloc(ctx, (void **)&s.control, sizeof(RtStreamControl)); if (status) return status;
	RT_HIP(ctx, hipMemset(s.control, 0, sizeof(RtStreamControl)));
	status = device_alloc(ctx, (void **)&s.table_device, sizeof(RtStreamTable)); if (status) return status;
	RT_HIP(ctx, hipMemset(s.table_device, 0, sizeof(RtStreamTable)));
	memset(&s.table_host, 0, sizeof(s.table_host));
	RT_HIP(ctx, hipHostMalloc((void **)&s.table_staging, sizeof(RtStreamTable) * RT_STREAM_TABLE_SNAPSHOTS));
	for (hipEvent_t & e : s.table_copied) { RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); RT_HIP(ctx, hipEventRecord(e, s.stream)); }
	// one spill area: the closest-hit and the shadow part of the fused launch run one after the other in the same waves
	status = device_alloc(ctx, &s.spill, size_t(24) * 8 * 256 * 8 * 256); if (status) return status;
	RT_HIP(ctx, hipHostMalloc((void **)&s.progress, sizeof(int) * 2 * RT_STREAM_PROGRESS_RING));
	for (int i = 0; i < 2 * RT_STREAM_PROGRESS_RING; i++) s.progress[i] = -1;
	for (hipEvent_t & e : s.iteration_done) { RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); RT_HIP(ctx, hipEventRecord(e, s.stream)); }
	RT_HIP(ctx, hipHostMalloc((void **)&s.stats_host, sizeof(int) * RT_STREAM_SUBMISSIONS * RT_STREAM_STATS_ROW));
	memset(s.stats_host, 0, sizeof(int) * RT_STREAM_SUBMISSIONS * RT_STREAM_STATS_ROW);
	for (int i = 0; i < RT_STREAM_SUBMISSIONS; i++) { RT_HIP(ctx, hipEventCreate(&s.ev_begin[i])); RT_HIP(ctx, hipEventCreate(&s.ev_end[i])); }
	s.created = true;
	return RT_OK;
}

static void stream_destroy(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	if (ctx->stream_history) { (void)hipHostFree(ctx->stream_history); ctx->stream_history = nullptr; }
	if (!s.stream) return;
	if (s.table_staging) (void)hipHostFree(s.table_staging);
	if (s.progress)      (void)hipHostFree(s.progress);
	if (s.stats_host)    (void)hipHostFree(s.stats_host);
	for (hipEvent_t e : s.table_copied)   if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : s.iteration_done) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : s.ev_begin)       if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : s.ev_end)         if (e) (void)hipEventDestroy(e);
	if (s.ev_idle) (void)hipEventDestroy(s.ev_idle);
	(void)hipStreamDestroy(s.stream);
	s.stream = nullptr; s.created = false;
}


// SVGF: a frame starts from a copy of the g-buffers of the frame before it (the reference has ONE set that every frame
// overwrites where its primary rays hit something). Before the sets of the sample slots are freed or re-allocated the
// latest one moves to the set of slot 0 of the slot scheduler, where the next frame of either scheduler finds it.
static const size_t stream_gbuffer_elem[3] = { 16, 8, 8 };
static void stream_save_gbuffers(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	if (s.last_gbuffer_slot >= 0 && s.gbuffers[0] && ctx->svgf_buffers[0]) {
		(void)hipDeviceSynchronize();
		for (int i = 0; i < 3; i++) (void)hipMemcpy(ctx->svgf_buffers[i], (char *)s.gbuffers[i] + ctx->frame_pixels * stream_gbuffer_elem[i] * size_t(s.last_gbuffer_slot), ctx->frame_pixels * stream_gbuffer_elem[i], hipMemcpyDeviceToDevice);
		ctx->last_slot = 0;
	}
	s.last_gbuffer_slot = -1;
}

// The per-sample frames of the sample slots (they are freed with the frame resources: rt_resize, AOV changes).
static void stream_release_frames(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	for (void * & fb : s.aov_framebuffer) { device_free(ctx, fb); fb = nullptr; }
	stream_save_gbuffers(ctx);
	for (void * & g : s.gbuffers) { device_free(ctx, g); g = nullptr; }
	s.frame_slots = 0;
}

static int stream_ensure_frames(rt_context * ctx, int wanted_slots) {
	PathStream & s = ctx->path_stream;
	int limit = int(std::min<size_t>(RT_STREAM_SAMPLE_SLOTS, ((size_t(1) << 30) - 1) / ctx->frame_pixels));
	int slots = std::min(limit, std::max(wanted_slots, 8));
	bool complete = s.frame_slots >= slots;
	for (int i = 0; i < RT_AOV_COUNT; i++) if ((ctx->aov_buffers[i][0] != nullptr) != (s.aov_framebuffer[i] != nullptr)) complete = false;
	if (ctx->svgf_allocated != (s.gbuffers[0] != nullptr)) complete = false;
	if (complete) return RT_OK;
	RT_HIP(ctx, quiesce(ctx));
	slots = std::max(slots, s.frame_slots);
	stream_save_gbuffers(ctx);
	for (void * & g : s.gbuffers) { device_free(ctx, g); g = nullptr; }
	for (void * & fb : s.aov_framebuffer) { device_free(ctx, fb); fb = nullptr; }
	size_t bytes = ctx->frame_pixels * 16 * size_t(slots);
	for (int i = 0; i < RT_AOV_COUNT; i++) if (ctx->aov_buffers[i][0]) {
		int status = device_alloc(ctx, &s.aov_framebuffer[i], bytes); if (status) return status;
		RT_HIP(ctx, hipMemset(s.aov_framebuffer[i], 0, bytes));
	}
	if (ctx->svgf_allocated) for (int i = 0; i < 3; i++) {
		int status = device_alloc(ctx, &s.gbuffers[i], ctx->frame_pixels * stream_gbuffer_elem[i] * size_t(slots)); if (status) return status;
		RT_HIP(ctx, hipMemset(s.gbuffers[i], 0, ctx->frame_pixels * stream_gbuffer_elem[i] * size_t(slots)));
	}
	s.frame_slots = slots;
	for (bool & used : s.slot_used) used = false;
	s.next_slot = 0;
	return RT_OK;
}

static int stream_ensure_queues(rt_context * ctx, size_t entries) {
	PathStream & s = ctx->path_stream;
	if (s.queues_allocated && s.capacity >= entries) return RT_OK;
	RT_HIP(ctx, quiesce(ctx));
	auto free3 = [&](RtVec3SoA & v) { device_free(ctx, v.x); device_free(ctx, v.y); device_free(ctx, v.z); };
	if (s.queues_allocated) {
		for (RtTraceBuffer & t : s.trace) { free3(t.origin); free3(t.direction); device_free(ctx, t.hits); device_free(ctx, t.cone_angle); device_free(ctx, t.cone_width); device_free(ctx, t.medium); device_free(ctx, t.pixel_index_and_flags); free3(t.throughput); device_free(ctx, t.last_pdf); }
		for (RtMaterialBuffer & m : s.material) { free3(m.direction); device_free(ctx, m.hits); device_free(ctx, m.cone_angle); device_free(ctx, m.cone_width); device_free(ctx, m.medium); device_free(ctx, m.pixel_index_and_flags); free3(m.throughput); }
		free3(s.shadow.origin); free3(s.shadow.direction); device_free(ctx, s.shadow.max_distance); device_free(ctx, s.shadow.illumination_and_pixel_index);
		s.queues_allocated = false;
	}
	size_t n = entries;
	int status;
	for (RtTraceBuffer & t : s.trace) {
		if ((status = alloc_vec3(ctx, t.origin, n))) return status;
		if ((status = alloc_vec3(ctx, t.direction, n))) return status;
		if ((status = device_alloc(ctx, (void **)&t.hits, n * 16))) return status;
		if ((status = device_alloc(ctx, (void **)&t.cone_angle, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&t.cone_width, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&t.medium, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&t.pixel_index_and_flags, n * 4))) return status;
		if ((status = alloc_vec3(ctx, t.throughput, n))) return status;
		if ((status = device_alloc(ctx, (void **)&t.last_pdf, n * 4))) return status;
	}
	for (RtMaterialBuffer & m : s.material) {
		if ((status = alloc_vec3(ctx, m.direction, n))) return status;
		if ((status = device_alloc(ctx, (void **)&m.hits, n * 16))) return status;
		if ((status = device_alloc(ctx, (void **)&m.cone_angle, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&m.cone_width, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&m.medium, n * 4))) return status;
		if ((status = device_alloc(ctx, (void **)&m.pixel_index_and_flags, n * 4))) return status;
		if ((status = alloc_vec3(ctx, m.throughput, n))) return status;
	}
	if ((status = alloc_vec3(ctx, s.shadow.origin, n))) return status;
	if ((status = alloc_vec3(ctx, s.shadow.direction, n))) return status;
	if ((status = device_alloc(ctx, (void **)&s.shadow.max_distance, n * 4))) return status;
	if ((status = device_alloc(ctx, (void **)&s.shadow.illumination_and_pixel_index, n * 16))) return status;
	s.queues_allocated = true;
	s.capacity = n;
	return RT_OK;
}

// The parameter block of iteration `iteration`: the context's block with the stream's queues, control block and
// per-sample frames patched in.
static RtParams stream_params(const rt_context * ctx, int iteration) {
	const PathStream & s = ctx->path_stream;
	RtParams p = ctx->params;
	memcpy(p.trace, s.trace, sizeof(p.trace)); memcpy(p.material, s.material, sizeof(p.material)); p.shadow = s.shadow;
	p.sizes = nullptr; p.xcd_counters = nullptr; p.stack_spill = (uint2 *)s.spill;
	p.stream = s.control; p.stream_table = s.table_device; p.stream_iteration = iteration;
	for (int i = 0; i < RT_AOV_COUNT; i++) p.aovs[i].framebuffer = (float4 *)s.aov_framebuffer[i];
	if (s.gbuffers[0]) { // x + y * pitch of a VIRTUAL pixel is the virtual pixel: the shade kernels write the set of the path's slot
		p.gbuffer_normal_and_depth = (float4 *)s.gbuffers[0]; p.gbuffer_mesh_id_and_triangle_id = (int2 *)s.gbuffers[1]; p.gbuffer_screen_position_prev = (float2 *)s.gbuffers[2];
	}
	p.batch_samples = 1;
	return p;
}

// Reads the wavefront sizes the device has reported so far (pinned memory, no synchronisation).
static void stream_update_known(PathStream & s) {
	for (int k = std::max(s.known_iteration + 1, s.iteration - RT_STREAM_PROGRESS_RING + 1); k < s.iteration; k++) {
		volatile int * entry = s.progress + 2 * (k % RT_STREAM_PROGRESS_RING);
		if (entry[0] != k) break;
		s.known_iteration = k; s.known_size = entry[1];
	}
}

// Upper bound of the paths in flight when the next iteration starts (before anything it generates).
static long long stream_bound(PathStream & s) {
	stream_update_known(s);
	int from = s.base_iteration; long long bound = 0;
	if (s.known_iteration >= s.base_iteration) { from = s.known_iteration + 1; bound = s.known_size; }
	for (int k = from; k < s.iteration; k++) bound += s.generated[k % RT_STREAM_PROGRESS_RING];
	return bound;
}

// The submissions that have passed their last bounce with iteration i (consecutive, in submission order; with batched
// admission up to RT_STREAM_MAX_BATCH of them): folded into the accumulators, their sample slots cleared and released.
static int stream_complete(rt_context * ctx, const StreamSubmission * subs, int count, const RtParams & p) {
	PathStream & s = ctx->path_stream;
	hipStream_t st = s.stream;
	// the accumulate step follows the main-stream work submitted so far (rt_pack_pixels of an earlier frame reads,
	// rt_unpack_pixels writes the image)
	RT_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_main, 0));
	stage_mark(ctx, STAGE_POST, st);
	if (p.config.enable_svgf) { // one filtered frame per submission: reproject / variance / a-trous / finalize / TAA on its own sample frames and g-buffers
		for (int k = 0; k < count; k++) {
			const StreamSubmission & sub = subs[k];
			RtParams pf = p;
			pf.taa_frame_prev = ctx->params.taa_frame_prev; pf.taa_frame_next = ctx->params.taa_frame_next;   // (they trade places after every filtered frame)
			for (int i = 0; i < RT_AOV_COUNT; i++) if (pf.aovs[i].framebuffer) pf.aovs[i].framebuffer += size_t(sub.slot_base) * ctx->frame_pixels;
			pf.gbuffer_normal_and_depth += size_t(sub.slot_base) * ctx->frame_pixels; pf.gbuffer_mesh_id_and_triangle_id += size_t(sub.slot_base) * ctx->frame_pixels; pf.gbuffer_screen_position_prev += size_t(sub.slot_base) * ctx->frame_pixels;
			rt_launch_svgf_taa(pf, sub.first_sample, st, ctx->launch_timing_all && ctx->time_this_sample ? svgf_span_mark : nullptr, ctx);
			if (pf.config.enable_taa) std::swap(ctx->params.taa_frame_prev, ctx->params.taa_frame_next);   // (the submission's own config snapshot launched kernel_taa: the same one decides the swap)
			for (int i = 0; i < RT_AOV_COUNT; i++) if (pf.aovs[i].framebuffer) RT_HIP(ctx, hipMemsetAsync(pf.aovs[i].framebuffer, 0, ctx->frame_pixels * 16 * size_t(sub.sample_count), st)); // aovs_clear_to_zero
		}
	} else
	for (int first = 0; first < count; ) {
		// one launch for a run of submissions over the same pixels
		const StreamSubmission & head = subs[first];
		RtAccumulateGroup group; group.count = 0;
		int k = first;
		for (; k < count && group.count < RT_ACCUMULATE_GROUP; k++) {
			const StreamSubmission & sub = subs[k];
			if (sub.range_offset != head.range_offset || sub.range_count != head.range_count || sub.tile_pixels != head.tile_pixels || sub.tile_first != head.tile_first || sub.tile_stride != head.tile_stride) break;
			group.first_sample[group.count] = sub.first_sample; group.sample_count[group.count] = sub.sample_count; group.slot_base[group.count] = sub.slot_base;
			group.count++;
		}
		RtParams pa = p;
		pa.tile_pixels = head.tile_pixels; pa.tile_first = head.tile_first; pa.tile_stride = head.tile_stride;
		span_mark(ctx, SPAN_ACCUMULATE, st);
		rt_launch_accumulate_group(pa, group, head.range_offset, head.range_count, st);
		span_mark(ctx, SPAN_ACCUMULATE, st);
		first = k;
	}
	stage_mark(ctx, STAGE_END, st);
	if (!p.config.enable_svgf) for (int k = 0; k < count; k++) {
		const StreamSubmission & sub = subs[k];
		// the AOVs the accumulate kernel does not fold (DIRECT / INDIRECT exist for the filter only) are cleared the plain way
		for (int i : { RT_AOV_RADIANCE_DIRECT, RT_AOV_RADIANCE_INDIRECT }) if (p.aovs[i].framebuffer)
			RT_HIP(ctx, hipMemsetAsync(p.aovs[i].framebuffer + size_t(sub.slot_base) * ctx->frame_pixels, 0, ctx->frame_pixels * 16 * size_t(sub.sample_count), st)); // aovs_clear_to_zero
	}
	for (int k = 0; k < count; ) { // statistics rows: consecutive ring entries travel in one copy
		int run = 1;
		while (k + run < count && subs[k + run].ring == subs[k].ring + run) run++;
		RT_HIP(ctx, hipMemcpyAsync(s.stats_host + size_t(subs[k].ring) * RT_STREAM_STATS_ROW, &s.control->stats[subs[k].ring][0][0], sizeof(int) * RT_STREAM_STATS_ROW * size_t(run), hipMemcpyDeviceToHost, st));
		k += run;
	}
	for (int k = 0; k < count; k++) {
		const StreamSubmission & sub = subs[k];
		RT_HIP(ctx, hipEventRecord(s.ev_end[sub.ring], st));
		for (int j = 0; j < sub.sample_count; j++) s.slot_used[sub.slot_base + j] = false;
		s.last_completed_ring = sub.ring;
		s.submissions_completed++;
	}
	RT_HIP(ctx, hipEventRecord(s.ev_idle, st));
	return RT_OK;
}

// The fused traversal launch fetches every node from ONE array: the TLAS (which lives in a ring of versions, for the
// per-submission chains that keep frames of different scene versions in flight) is copied into the slots [0, node count)
// that the BLAS node array reserves for it -- node indices below the TLAS size never name BLAS nodes. Nothing of the merged
// wavefront is in flight when the TLAS changes (every upload completes it first), and the other kernels never read those slots.
static int stream_sync_tlas(rt_context * ctx) {
	if (ctx->tlas_version_in_nodes == ctx->tlas_version) return RT_OK;
	if (!ctx->params.tlas_nodes || ctx->params.tlas_node_count <= 0) {   // one BVH, no TLAS: node 0 is its root
		ctx->tlas_version_in_nodes = ctx->tlas_version; return RT_OK;
	}
	if (!ctx->bvh8_nodes || size_t(ctx->params.tlas_node_count) > ctx->bvh8_node_count) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_samples: the TLAS does not fit the node slots the geometry reserves for it");
	// the copy below overwrites node slots [0, tlas_node_count): they must be CWBVH nodes and must not hold BLAS nodes (the host
	// classes reserve 2 x meshes slots in front of the first BLAS; a C-API caller with another layout gets an error, not a
	// silently corrupted BVH -- the per-submission scheduler, which reads the TLAS from its own buffer, renders such layouts)
	if (ctx->tlas_node_bytes != 80) return fail(ctx, RT_ERROR_INVALID_ARG, "rt_render_samples: the merged wavefront needs the TLAS as 80-byte CWBVH nodes (rt_upload_tlas); select RT_SCHEDULER_SLOTS for other BVH types");
	if (ctx->params.tlas_node_count > ctx->lowest_blas_root) return fail(ctx, RT_ERROR_INVALID_ARG, "rt_render_samples: a BLAS root (node %d) lies inside the %d node slots the merged wavefront copies the TLAS into; reserve them in front of the BLAS nodes or select RT_SCHEDULER_SLOTS", ctx->lowest_blas_root, ctx->params.tlas_node_count);
	hipStream_t st = ctx->path_stream.stream;
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_scene, 0));
	RT_HIP(ctx, hipMemcpyAsync(ctx->bvh8_nodes, ctx->params.tlas_nodes, size_t(ctx->params.tlas_node_count) * 80, hipMemcpyDeviceToDevice, st));
	ctx->tlas_version_in_nodes = ctx->tlas_version;
	return RT_OK;
}

// The primary rays of a new submission: appended to the trace queue of the iteration that is enqueued next, behind the
// rays of the submissions already waiting for it.
static int stream_generate(rt_context * ctx, const StreamSubmission & sub) {
	PathStream & s = ctx->path_stream;
	hipStream_t st = s.stream;
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_scene, 0));  // asynchronous scene uploads (they flush the wavefront first)
	RtParams pg = stream_params(ctx, s.iteration);
	pg.batch_samples = sub.sample_count;
	pg.tile_pixels = sub.tile_pixels; pg.tile_first = sub.tile_first; pg.tile_stride = sub.tile_stride;
	RT_HIP(ctx, hipEventRecord(s.ev_begin[sub.ring], st));
	stage_mark(ctx, STAGE_GENERATE, st);
	span_mark(ctx, SPAN_GENERATE, st);
	// Queue order of the primary rays: 8 x 8 pixel patches along bands of 8 scan lines when the submission's pixel list is made of whole bands
	// (the whole frame, or tiles of whole bands: rt_map_pixel keeps a band together); scan lines otherwise. GRT_PRIMARY_ORDER=WxH / 0 for A / B runs.
	static const int order_width = [] { const char * e = getenv("GRT_PRIMARY_ORDER"); return e ? atoi(e) : 8; }();
	static const int order_rows  = [] { const char * e = getenv("GRT_PRIMARY_ORDER"); const char * x = e ? strchr(e, 'x') : nullptr; return e ? (x ? atoi(x + 1) : atoi(e)) : 8; }();
	const int frame = pg.screen_width * pg.screen_height;
	const bool whole_bands = order_width > 0 && order_rows > 0 && (sub.tile_pixels > 0 ? sub.tile_pixels % (order_rows * pg.screen_width) == 0 : (sub.range_offset == 0 && sub.range_count == frame));
	rt_launch_generate_stream(pg, sub.first_sample, sub.range_offset, sub.range_count, sub.slot_base, int(s.pending_paths), whole_bands ? order_width : 0, whole_bands ? order_rows : 0, st);
	span_mark(ctx, SPAN_GENERATE, st);
	s.pending++; s.pending_paths += sub.paths;
	return RT_OK;
}

// One iteration of the wavefront: advance (counts the rays generated since the last one in) -> fused trace -> sort ->
// shade, then the accumulate step of every submission that has just passed its last bounce.
static int stream_enqueue_iteration(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	const int i = s.iteration;
	hipStream_t st = s.stream;
	if (i - RT_STREAM_RUN_AHEAD >= 0) RT_HIP(ctx, hipEventSynchronize(s.iteration_done[(i - RT_STREAM_RUN_AHEAD) % RT_STREAM_PROGRESS_RING]));
	RtParams p = stream_params(ctx, i);
	const int generated = int(s.pending_paths);
	s.pending = 0; s.pending_paths = 0;
	if (s.table_dirty) {
		// the table travels in stream order: the kernels of earlier iterations have run when it lands, and they never look at the entries of free slots
		int snapshot = s.table_next; s.table_next = (s.table_next + 1) % RT_STREAM_TABLE_SNAPSHOTS;
		RT_HIP(ctx, hipEventSynchronize(s.table_copied[snapshot]));
		s.table_staging[snapshot] = s.table_host;
		RT_HIP(ctx, hipMemcpyAsync(s.table_device, &s.table_staging[snapshot], sizeof(RtStreamTable), hipMemcpyHostToDevice, st));
		RT_HIP(ctx, hipEventRecord(s.table_copied[snapshot], st));
		s.table_dirty = false;
	}
	rt_launch_stream_advance(s.control, i, generated, s.progress + 2 * (i % RT_STREAM_PROGRESS_RING), s.reset_ring_first, s.reset_ring_count, st);
	s.reset_ring_count = 0;
	s.generated[i % RT_STREAM_PROGRESS_RING] = generated;
	RT_HIP(ctx, hipEventRecord(s.iteration_done[i % RT_STREAM_PROGRESS_RING], st));
	stage_mark(ctx, STAGE_TRACE, st);
	span_mark(ctx, SPAN_TRACE, st);
	rt_launch_trace_stream(p, ctx->trace_statistics ? ctx->trace_stats : nullptr, st);
	span_mark(ctx, SPAN_TRACE, st);
	if (ctx->trace_statistics && ctx->stream_history_rows < RT_STREAM_HISTORY_ROWS)
		RT_HIP(ctx, hipMemcpyAsync(ctx->stream_history + size_t(10) * ctx->stream_history_rows++, ctx->trace_stats, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	stage_mark(ctx, STAGE_SORT, st);
	span_mark(ctx, SPAN_SORT, st);
	rt_launch_sort_stream(p, st);
	span_mark(ctx, SPAN_SORT, st);
	stage_mark(ctx, STAGE_SHADE, st);
	for (int m = 0; m < 4; m++) if (ctx->has_material[m]) { span_mark(ctx, SPAN_MATERIAL + m, st); rt_launch_material_stream(p, m, st); span_mark(ctx, SPAN_MATERIAL + m, st); }
	stage_mark(ctx, STAGE_END, st);
	s.iteration = i + 1;
	StreamSubmission done[RT_STREAM_MAX_BATCH]; int done_count = 0;
	while (!s.in_flight.empty() && s.in_flight.front().last <= i) {
		done[done_count++] = s.in_flight.front();
		s.in_flight.pop_front();
		if (done_count == RT_STREAM_MAX_BATCH || s.in_flight.empty() || s.in_flight.front().last > i) {
			int status = stream_complete(ctx, done, done_count, p); if (status) return status;
			done_count = 0;
		}
	}
	if (s.in_flight.empty()) s.base_iteration = s.iteration;
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

// Submissions that wait for company get their iteration now (before a setting they were made under changes).
static int stream_launch_pending(rt_context * ctx) {
	if (!ctx->path_stream.created || ctx->path_stream.pending == 0) return RT_OK;
	(void)hipSetDevice(ctx->device);
	return stream_enqueue_iteration(ctx);
}

// Runs the wavefront until nothing is in flight (the calls that read results or change state need that).
static hipError_t stream_flush(rt_context * ctx) {
	PathStream & s = ctx->path_stream;
	while (s.created && !s.in_flight.empty()) if (stream_enqueue_iteration(ctx) != RT_OK) return hipErrorUnknown;
	return hipSuccess;
}

// rt_render_samples under RT_SCHEDULER_MERGED
static int stream_submit(rt_context * ctx, int sample_index, int sample_count, int range_offset, int range_count) {
	int status = stream_create(ctx); if (status) return status;
	PathStream & s = ctx->path_stream;
	const int num_bounces = ctx->params.config.num_bounces;
	const long long paths = (long long)range_count * sample_count;
	if (paths <= 0) return RT_OK;
	// submissions per iteration: one, unless the application pipelines frames and they are small
	// (SVGF frames: never, a frame inherits the g-buffers its predecessor's bounce 0 has written)
	const long long batch_paths = ctx->stream_batch_paths > 0 ? ctx->stream_batch_paths : (long long)RT_STREAM_BATCH_PATHS;
	const int batch = (ctx->frame_pipelining && !ctx->params.config.enable_svgf) ? int(std::min<long long>(RT_STREAM_MAX_BATCH, (batch_paths + paths - 1) / paths)) : 1;
	status = stream_ensure_frames(ctx, sample_count * (num_bounces + 1) * batch); if (status) return status;
	if (sample_count > s.frame_slots) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_samples: %d samples of a %zu pixel frame exceed the %d sample slots of the merged wavefront", sample_count, ctx->frame_pixels, s.frame_slots);
	static const double factor = getenv("GRT_STREAM_CAPACITY_FACTOR") ? atof(getenv("GRT_STREAM_CAPACITY_FACTOR")) : 4.0;
	// room for four iterations' worth of new paths -- or, for a declared burst of whole frames (rt_set_stream_batch), for the burst and a quarter: its frames
	// enter together and only die from then on (412 bytes per path: 21 GB for five 4-sample frames at 1080p instead of 68)
	const double room = batch_paths > (long long)RT_STREAM_BATCH_PATHS && paths * batch > (long long)RT_STREAM_BATCH_PATHS ? 1.25 : (factor < 1.0 ? 1.0 : factor);
	if (size_t(paths * batch) > s.capacity || !s.queues_allocated) { status = stream_ensure_queues(ctx, size_t(double(paths * batch) * room) + 1024); if (status) return status; }

	// admission: sample slots, a statistics ring entry, and room in the queues
	int slot_base = -1;
	for (;;) {
		bool ring_free = s.in_flight.size() < RT_STREAM_SUBMISSIONS - 1;
		slot_base = -1;
		const int legal_bases = s.frame_slots - sample_count + 1;   // every base 0 .. frame_slots - sample_count is tried, starting at next_slot (round robin)
		for (int start = 0; start < legal_bases && slot_base < 0; start++) {
			int candidate = ((s.next_slot < legal_bases ? s.next_slot : 0) + start) % legal_bases;
			bool free_run = true;
			for (int k = 0; k < sample_count; k++) if (s.slot_used[candidate + k]) { free_run = false; break; }
			if (free_run) slot_base = candidate;
		}
		bool fits = stream_bound(s) + s.pending_paths + paths <= (long long)s.capacity;   // (with the submissions already waiting for the next iteration)
		if (ring_free && slot_base >= 0 && fits) break;
		if (!fits && s.known_iteration < s.iteration - 1 && s.iteration > s.base_iteration) { // the bound is stale: let the device catch up
			RT_HIP(ctx, hipEventSynchronize(s.iteration_done[(s.iteration - 1) % RT_STREAM_PROGRESS_RING]));
			continue;
		}
		if (s.in_flight.empty()) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_samples: the merged wavefront cannot take %lld paths (capacity %zu, %d sample slots)", paths, s.capacity, s.frame_slots);
		status = stream_enqueue_iteration(ctx); if (status) return status;   // advance (with the submissions waiting, if any): paths die, submissions complete
	}

	StreamSubmission sub;
	sub.first_sample = sample_index; sub.sample_count = sample_count; sub.slot_base = slot_base;
	sub.ring = s.next_ring; s.next_ring = (s.next_ring + 1) % RT_STREAM_SUBMISSIONS;
	sub.birth = s.iteration; sub.last = s.iteration + num_bounces - 1; sub.paths = int(paths);
	sub.range_offset = range_offset; sub.range_count = range_count;
	sub.tile_pixels = ctx->params.tile_pixels; sub.tile_first = ctx->params.tile_first; sub.tile_stride = ctx->params.tile_stride;
	for (int k = 0; k < sample_count; k++) {
		s.slot_used[slot_base + k] = true;
		s.table_host.slots[slot_base + k] = { sample_index + k, sub.birth, sub.ring, k };
	}
	s.next_slot = (slot_base + sample_count) % s.frame_slots;
	s.table_host.submission_birth[sub.ring] = sub.birth;
	// the table and the submission's (zeroed) statistics row travel with the iteration it joins (stream_enqueue_iteration): generate, the only kernel that runs
	// before that, reads neither
	if (s.reset_ring_count == 0) s.reset_ring_first = sub.ring;
	s.reset_ring_count++; s.table_dirty = true;
	if (ctx->trace_statistics && s.in_flight.empty()) { // statistics are per run of the wavefront
		RT_HIP(ctx, hipMemsetAsync(ctx->trace_stats, 0, 10 * sizeof(unsigned long long), s.stream));
		ctx->stream_history_rows = 0;
	}
	s.in_flight.push_back(sub);
	if (ctx->params.config.enable_svgf && s.gbuffers[0]) { // the frame starts from the g-buffers of the frame before it
		const void * from[3];
		for (int i = 0; i < 3; i++) {
			if (s.last_gbuffer_slot >= 0) from[i] = (char *)s.gbuffers[i] + ctx->frame_pixels * stream_gbuffer_elem[i] * size_t(s.last_gbuffer_slot);
			else from[i] = (ctx->last_slot > 0 && ctx->slots[ctx->last_slot].gbuffers[i]) ? ctx->slots[ctx->last_slot].gbuffers[i] : ctx->svgf_buffers[i]; // (the slot scheduler is drained)
			void * to = (char *)s.gbuffers[i] + ctx->frame_pixels * stream_gbuffer_elem[i] * size_t(slot_base);
			if (from[i] && from[i] != to) RT_HIP(ctx, hipMemcpyAsync(to, from[i], ctx->frame_pixels * stream_gbuffer_elem[i], hipMemcpyDeviceToDevice, s.stream));
		}
		s.last_gbuffer_slot = slot_base;
	}
	status = stream_sync_tlas(ctx); if (status) return status;
	status = stream_generate(ctx, sub); if (status) return status;
	if (s.pending < batch && s.pending_paths < batch_paths) return RT_OK;   // wait for more of the same size
	return stream_enqueue_iteration(ctx);
}

extern "C" {

int rt_render_sample(rt_context * ctx, int sample_index) { return rt_render_samples(ctx, sample_index, 1); }

int rt_render_samples(rt_context * ctx, int sample_index, int sample_count) {
	RT_REQUIRE(ctx, ctx, "rt_render_sample: NULL context");
	RT_REQUIRE(ctx, sample_count >= 1 && sample_count <= RT_MAX_BATCH_SAMPLES, "rt_render_samples: sample_count must be 1..16");
	(void)hipSetDevice(ctx->device);
	const RtParams & base = ctx->params;
	if (!base.triangles || !bvh_nodes_present(ctx)) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: geometry not uploaded");
	if (!base.mesh_bvh_root_indices)        return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: instances not uploaded");
	if (!base.materials)                    return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: materials not uploaded");
	if (!base.pmj_samples || !base.blue_noise) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: RNG tables not uploaded");
	if (!base.sky)                          return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: sky not set");
	if (ctx->frame_pixels == 0)          return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: rt_resize was not called");
	if (ctx->bvh_width != 8 && ctx->trace_statistics) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_sample: trace statistics exist for the CWBVH kernels only");
	// Slot choice: round-robin over the samples in flight. Profiling / statistics passes use one slot,
	// serialised. SVGF frames pipeline like plain samples under either scheduler: only their filter stage is ordered.
	// Scheduler (see rt_set_scheduler): the merged wavefront where its kernels exist
	const bool merged = ctx->scheduler == RT_SCHEDULER_MERGED && ctx->params.config.num_bounces > 0 && ctx->bvh_width == 8
	                    && !(ctx->params.config.enable_svgf && ctx->defer_filter) // the tile split of SVGF frames exchanges the filter's inputs per frame
	                    && !(ctx->batch_size_request > 0); // explicit pixel batches (a VRAM bound of the reference) are a slot-scheduler feature
	if (ctx->params.config.enable_svgf && sample_count != 1) return fail(ctx, RT_ERROR_INVALID_ARG, "rt_render_samples: SVGF frames are rendered one sample at a time");
	if (merged != ctx->last_render_merged) { RT_HIP(ctx, quiesce(ctx)); ctx->last_render_merged = merged; }
	if (merged) {
		if (ctx->has_material[2] || ctx->has_material[3]) { int ls = ensure_luts(ctx); if (ls) return ls; }
		if (size_t(sample_count) * ctx->frame_pixels >= (1u << 30)) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_samples: %d samples of a %zu pixel frame exceed the 30-bit path index", sample_count, ctx->frame_pixels);
		const RtParams & cp = ctx->params;
		int frame = cp.screen_width * cp.screen_height;
		int offset = ctx->pixel_offset, count = ctx->pixel_count < 0 ? frame - offset : ctx->pixel_count;
		if (cp.tile_pixels > 0) { // tile mode: local pixels 0..count-1 are mapped to this context's tiles by rt_map_pixel
			int tiles_total = (frame + cp.tile_pixels - 1) / cp.tile_pixels;
			int owned = cp.tile_first < tiles_total ? (tiles_total - cp.tile_first + cp.tile_stride - 1) / cp.tile_stride : 0;
			offset = 0; count = owned * cp.tile_pixels;
			int last_tile = cp.tile_first + (owned - 1) * cp.tile_stride;
			if (owned > 0 && last_tile == tiles_total - 1) count -= tiles_total * cp.tile_pixels - frame; // clipped last tile
		}
		if (cp.tile_pixels == 0 && offset + count > frame) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_sample: pixel range [%d,%d) exceeds the %d pixel frame", offset, offset + count, frame);
		ctx->time_this_sample = ctx->launch_timing;
		return stream_submit(ctx, sample_index, sample_count, offset, count);
	}
	bool exclusive = ctx->profiling || ctx->trace_statistics || ctx->defer_filter;
	int slot_index = exclusive ? 0 : int(ctx->render_counter++ % unsigned(ctx->samples_in_flight));
	int s = ensure_slot(ctx, slot_index); if (s) return s;
	if (ctx->has_material[2] || ctx->has_material[3]) { s = ensure_luts(ctx); if (s) return s; }
	SampleSlot & slot = ctx->slots[slot_index];
	if (size_t(sample_count) * ctx->frame_pixels >= (1u << 30)) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_samples: %d samples of a %zu pixel frame exceed the 30-bit path index", sample_count, ctx->frame_pixels);
	s = ensure_aov_batch(ctx, slot_index, sample_count); if (s) return s;
	RtParams p = slot_params(ctx, slot, slot_index);
	p.batch_samples = sample_count;
	RtParams p_shadow = p; // the shadow launch may be resident together with the next closest-hit launch

	int frame_pixels = p.screen_width * p.screen_height;
	int range_offset = ctx->pixel_offset;
	int range_count  = ctx->pixel_count < 0 ? frame_pixels - range_offset : ctx->pixel_count;
	if (p.tile_pixels > 0) { // tile mode: local pixels 0..count-1 are mapped to this context's tiles by rt_map_pixel
		int tiles_total = (frame_pixels + p.tile_pixels - 1) / p.tile_pixels;
		int owned = p.tile_first < tiles_total ? (tiles_total - p.tile_first + p.tile_stride - 1) / p.tile_stride : 0;
		range_offset = 0;
		range_count = owned * p.tile_pixels;
		int last_tile = p.tile_first + (owned - 1) * p.tile_stride;
		if (owned > 0 && last_tile == tiles_total - 1) range_count -= tiles_total * p.tile_pixels - frame_pixels; // clipped last tile
	}
	if (p.tile_pixels == 0 && range_offset + range_count > frame_pixels) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_sample: pixel range [%d,%d) exceeds the %d pixel frame", range_offset, range_offset + range_count, frame_pixels);
	int batch_limit = int(wanted_batch_size(ctx));
	int batch_size  = range_count < batch_limit ? range_count : batch_limit; // pixels per wavefront batch; each carries sample_count paths
	s = ensure_queues(ctx, slot_index, size_t(batch_size > 0 ? batch_size : 1) * sample_count); if (s) return s;
	memcpy(p.trace, slot.trace, sizeof(p.trace)); memcpy(p.material, slot.material, sizeof(p.material)); p.shadow = slot.shadow; // (re)allocated just now
	p_shadow = p; p_shadow.stack_spill = (uint2 *)slot.spill[1];

	ctx->time_this_sample = ctx->launch_timing;
	hipStream_t st = slot.stream;
	// The wavefront part depends on nothing the main stream does asynchronously (uploads and the LUT
	// integration synchronise); only the accumulate step below has to follow the main-stream work
	// submitted so far (rt_pack_pixels of the previous frame reads, rt_unpack_pixels writes the image).
	RT_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
	// ... and it reads the scene version that is current now (asynchronous TLAS / instance uploads)
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_scene, 0));

	if (exclusive) for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) if (ctx->slots[k].created) RT_HIP(ctx, hipStreamWaitEvent(st, ctx->slots[k].ev_done, 0));

	ctx->stage_used = 0;
	RT_HIP(ctx, hipEventRecord(slot.ev_frame_start, st));
	const bool svgf = p.config.enable_svgf != 0;
	if (svgf && ctx->path_stream.last_gbuffer_slot >= 0) { // the previous frame went through the merged wavefront (which is drained by now)
		stream_save_gbuffers(ctx);                            // ... into the set of slot 0 (a blocking copy)
	}
	if (svgf && ctx->last_slot >= 0 && ctx->last_slot != slot_index) { // inherit the g-buffers of the previous frame
		SampleSlot & prev = ctx->slots[ctx->last_slot];
		void * from[3] = { ctx->last_slot == 0 ? ctx->svgf_buffers[0] : prev.gbuffers[0], ctx->last_slot == 0 ? ctx->svgf_buffers[1] : prev.gbuffers[1], ctx->last_slot == 0 ? ctx->svgf_buffers[2] : prev.gbuffers[2] };
		void * to[3]   = { p.gbuffer_normal_and_depth, p.gbuffer_mesh_id_and_triangle_id, p.gbuffer_screen_position_prev };
		const size_t elem[3] = { 16, 8, 8 };
		if (prev.created) RT_HIP(ctx, hipStreamWaitEvent(st, prev.ev_gbuffers, 0));
		for (int i = 0; i < 3; i++) if (from[i] && to[i] && from[i] != to[i]) RT_HIP(ctx, hipMemcpyAsync(to[i], from[i], ctx->frame_pixels * elem[i], hipMemcpyDeviceToDevice, st));
	}

	bool trace_shadows = ctx->has_lights && p.config.enable_next_event_estimation && p.lights_total_weight > 0.0f;
	// Shadow rays of bounce b only feed the frame buffers, so they run on the side stream while the
	// main chain traces bounce b+1; they are joined before the next kernel that touches the frame
	// buffers (sort: sky / emissive hits), which keeps the order of the float additions per pixel.
	bool overlap = trace_shadows && ctx->overlap_shadows && !ctx->profiling && !ctx->trace_statistics;

	// generate -> (trace, sort, shade, shadow) x bounces for every batch of the range, on st (+ side)
	auto submit_wavefront = [&]() -> int {
		RT_HIP(ctx, hipMemsetAsync(slot.counter_totals, 0, 6 * RT_MAX_BOUNCES * sizeof(int), st));
		if (ctx->trace_statistics) RT_HIP(ctx, hipMemsetAsync(ctx->trace_stats, 0, 10 * sizeof(unsigned long long), st));
		int pixels_left = range_count;
		bool shadow_pending = false;
		while (pixels_left > 0) {
			int pixel_offset = range_offset + (range_count - pixels_left);
			int pixel_count  = batch_size < pixels_left ? batch_size : pixels_left;

			if (shadow_pending) { RT_HIP(ctx, hipStreamWaitEvent(st, slot.ev_shadowed, 0)); shadow_pending = false; } // the previous batch's queues are reused
			RT_HIP(ctx, hipMemsetAsync(slot.sizes, 0, sizeof(RtBufferSizes), st));
			RT_HIP(ctx, hipMemsetAsync(slot.xcd_counters, 0, RT_MAX_BOUNCES * 2 * 8 * sizeof(int), st));
			stage_mark(ctx, STAGE_GENERATE, st);
			rt_launch_generate(p, sample_index, pixel_offset, pixel_count, st);

			for (int bounce = 0; bounce < p.config.num_bounces; bounce++) {
				stage_mark(ctx, STAGE_TRACE, st);
				if (ctx->trace_statistics) rt_launch_trace_counting(p, bounce, ctx->trace_stats, st);
				else { span_mark(ctx, SPAN_TRACE, st); rt_launch_trace(p, bounce, st); span_mark(ctx, SPAN_TRACE, st); }
				if (shadow_pending) { RT_HIP(ctx, hipStreamWaitEvent(st, slot.ev_shadowed, 0)); shadow_pending = false; }
				stage_mark(ctx, STAGE_SORT, st);
				rt_launch_sort(p, bounce, sample_index, st);
				stage_mark(ctx, STAGE_SHADE, st);
				for (int m = 0; m < 4; m++) if (ctx->has_material[m]) rt_launch_material(p, m, bounce, sample_index, st);
				if (svgf && bounce == 0 && pixels_left <= batch_size) RT_HIP(ctx, hipEventRecord(slot.ev_gbuffers, st)); // last pixel batch: the g-buffers of this frame are complete
				if (trace_shadows) {
					stage_mark(ctx, STAGE_SHADOW, st);
					if (ctx->trace_statistics) rt_launch_trace_shadow_counting(p, bounce, ctx->trace_stats, st);
					else if (!overlap) rt_launch_trace_shadow(p, bounce, st);
					else {
						RT_HIP(ctx, hipEventRecord(slot.ev_shaded, st));
						RT_HIP(ctx, hipStreamWaitEvent(slot.side, slot.ev_shaded, 0));
						span_mark(ctx, SPAN_SHADOW, slot.side);
						rt_launch_trace_shadow(p_shadow, bounce, slot.side);
						span_mark(ctx, SPAN_SHADOW, slot.side);
						RT_HIP(ctx, hipEventRecord(slot.ev_shadowed, slot.side));
						shadow_pending = true;
					}
				}
			}
			hipLaunchKernelGGL(kernel_accumulate_counters, dim3(1), dim3(RT_MAX_BOUNCES), 0, st, slot.sizes, slot.counter_totals);
			pixels_left -= batch_size;
		}
		if (shadow_pending) RT_HIP(ctx, hipStreamWaitEvent(st, slot.ev_shadowed, 0));
		return RT_OK;
	};

	{ int status = submit_wavefront(); if (status) return status; }

	// The accumulate step folds this sample into the shared accumulators: strictly in sample order.
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_main, 0));
	if (ctx->last_slot >= 0 && ctx->last_slot != slot_index && ctx->slots[ctx->last_slot].created) RT_HIP(ctx, hipStreamWaitEvent(st, ctx->slots[ctx->last_slot].ev_done, 0));
	stage_mark(ctx, STAGE_POST, st);
	const bool deferred = p.config.enable_svgf && ctx->defer_filter; // rt_filter_frame does the rest once the ranks have exchanged their tiles
	if (deferred) { }
	else if (p.config.enable_svgf) { p.taa_frame_prev = ctx->params.taa_frame_prev; p.taa_frame_next = ctx->params.taa_frame_next; rt_launch_svgf_taa(p, sample_index, st); if (p.config.enable_taa) std::swap(ctx->params.taa_frame_prev, ctx->params.taa_frame_next); }
	else rt_launch_accumulate(p, float(sample_index), range_offset, range_count, st);
	stage_mark(ctx, STAGE_END, st);

	// aovs_clear_to_zero (Integrator.cpp:379-385)
	if (!deferred) for (int i = 0; i < RT_AOV_COUNT; i++) if (p.aovs[i].framebuffer) RT_HIP(ctx, hipMemsetAsync(p.aovs[i].framebuffer, 0, ctx->frame_pixels * 16 * sample_count, st));

	RT_HIP(ctx, hipMemcpyAsync(slot.pinned_counters, slot.counter_totals, 6 * RT_MAX_BOUNCES * sizeof(int), hipMemcpyDeviceToHost, st));
	RT_HIP(ctx, hipEventRecord(slot.ev_frame_end, st));
	RT_HIP(ctx, hipEventRecord(slot.ev_done, st));
	{ int status = mark_scene_versions_in_use(ctx, slot_index, st); if (status) return status; }
	ctx->last_slot = slot_index;
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

// Ambient-occlusion integrator (AO::render, Integrators/AO.cpp:148-200): generate -> trace ->
// kernel_ambient_occlusion -> shadow trace (sets RADIANCE) per batch, then accumulate. One sample at
// a time on slot 0; it shares the queues, the trace kernels and the accumulate kernel of the path tracer.
int rt_render_ao_sample(rt_context * ctx, int sample_index, float ao_radius) {
	RT_REQUIRE(ctx, ctx, "rt_render_ao_sample: NULL context");
	RT_REQUIRE(ctx, ao_radius > 0.0f, "rt_render_ao_sample: ao_radius must be positive");
	(void)hipSetDevice(ctx->device);
	const RtParams & base = ctx->params;
	if (!base.triangles || !bvh_nodes_present(ctx)) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_ao_sample: geometry not uploaded");
	if (!base.mesh_bvh_root_indices)           return fail(ctx, RT_ERROR_NOT_READY, "rt_render_ao_sample: instances not uploaded");
	if (!base.pmj_samples || !base.blue_noise) return fail(ctx, RT_ERROR_NOT_READY, "rt_render_ao_sample: RNG tables not uploaded");
	if (ctx->frame_pixels == 0)                return fail(ctx, RT_ERROR_NOT_READY, "rt_render_ao_sample: rt_resize was not called");
	if (base.tile_pixels > 0)                  return fail(ctx, RT_ERROR_INVALID_ARG, "rt_render_ao_sample: tile mode is not supported, use rt_set_pixel_range");

	int frame_pixels = base.screen_width * base.screen_height;
	int range_offset = ctx->pixel_offset;
	int range_count  = ctx->pixel_count < 0 ? frame_pixels - range_offset : ctx->pixel_count;
	if (range_offset + range_count > frame_pixels) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_render_ao_sample: pixel range [%d,%d) exceeds the %d pixel frame", range_offset, range_offset + range_count, frame_pixels);

	if (ctx->last_render_merged) { RT_HIP(ctx, quiesce(ctx)); ctx->last_render_merged = false; } // this integrator runs on slot 0
	int s = ensure_slot(ctx, 0); if (s) return s;
	int batch_limit = int(wanted_batch_size(ctx));
	int batch_size  = range_count < batch_limit ? range_count : batch_limit;
	s = ensure_queues(ctx, 0, size_t(batch_size > 0 ? batch_size : 1)); if (s) return s;
	SampleSlot & slot = ctx->slots[0];
	RtParams p = slot_params(ctx, slot, 0);
	p.batch_samples = 1;
	p.config.enable_svgf = 0; // the AO integrator has no SVGF path; the TAA jitter of kernel_generate stays off

	hipStream_t st = slot.stream;
	RT_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
	RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_main, 0));

	for (int k = 1; k < RT_MAX_SAMPLE_SLOTS; k++) if (ctx->slots[k].created) RT_HIP(ctx, hipStreamWaitEvent(st, ctx->slots[k].ev_done, 0));

	RT_HIP(ctx, hipEventRecord(slot.ev_frame_start, st));
	RT_HIP(ctx, hipMemsetAsync(slot.counter_totals, 0, 6 * RT_MAX_BOUNCES * sizeof(int), st));
	int pixels_left = range_count;
	while (pixels_left > 0) {
		int pixel_offset = range_offset + (range_count - pixels_left);
		int pixel_count  = batch_size < pixels_left ? batch_size : pixels_left;
		RT_HIP(ctx, hipMemsetAsync(slot.sizes, 0, sizeof(RtBufferSizes), st));
		RT_HIP(ctx, hipMemsetAsync(slot.xcd_counters, 0, RT_MAX_BOUNCES * 2 * 8 * sizeof(int), st));
		rt_launch_generate(p, sample_index, pixel_offset, pixel_count, st);
		rt_launch_trace(p, 0, st);
		rt_launch_ambient_occlusion(p, sample_index, ao_radius, st);
		rt_launch_trace_shadow_ao(p, st);
		hipLaunchKernelGGL(kernel_accumulate_counters, dim3(1), dim3(RT_MAX_BOUNCES), 0, st, slot.sizes, slot.counter_totals);
		pixels_left -= batch_size;
	}
	RtParams p_acc = p; // kernel_accumulate of AO.cu:161-183 folds RADIANCE, NORMAL and POSITION only
	p_acc.aovs[RT_AOV_ALBEDO].framebuffer = nullptr;
	rt_launch_accumulate(p_acc, float(sample_index), range_offset, range_count, st);
	for (int i = 0; i < RT_AOV_COUNT; i++) if (p.aovs[i].framebuffer) RT_HIP(ctx, hipMemsetAsync(p.aovs[i].framebuffer, 0, ctx->frame_pixels * 16, st));
	RT_HIP(ctx, hipMemcpyAsync(slot.pinned_counters, slot.counter_totals, 6 * RT_MAX_BOUNCES * sizeof(int), hipMemcpyDeviceToHost, st));
	RT_HIP(ctx, hipEventRecord(slot.ev_frame_end, st));
	RT_HIP(ctx, hipEventRecord(slot.ev_done, st));
	{ int status = mark_scene_versions_in_use(ctx, 0, st); if (status) return status; }
	ctx->last_slot = 0;
	RT_HIP(ctx, hipGetLastError());
	return RT_OK;
}

int rt_set_pixel_query(rt_context * ctx, int pixel_index) {
	RT_REQUIRE(ctx, ctx, "rt_set_pixel_query: NULL context");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemset(ctx->pixel_query_out, 0xff, 2 * sizeof(int))); // { INVALID, INVALID }
	ctx->params.pixel_query_pixel = pixel_index;
	return RT_OK;
}

int rt_get_pixel_query(rt_context * ctx, int * mesh_id, int * triangle_id) {
	RT_REQUIRE(ctx, ctx && mesh_id && triangle_id, "rt_get_pixel_query: NULL argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	int out[2] = { -1, -1 };
	RT_HIP(ctx, hipMemcpy(out, ctx->pixel_query_out, sizeof(out), hipMemcpyDeviceToHost));
	*mesh_id = out[0]; *triangle_id = out[1];
	return RT_OK;
}

int rt_set_samples_in_flight(rt_context * ctx, int count) {
	RT_REQUIRE(ctx, ctx && count >= 1 && count <= RT_MAX_SAMPLE_SLOTS, "rt_set_samples_in_flight: count must be 1..8");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	ctx->samples_in_flight = count;
	return RT_OK;
}

int rt_advance(rt_context * ctx) {
	RT_REQUIRE(ctx, ctx, "rt_advance: NULL context");
	(void)hipSetDevice(ctx->device);
	if (!ctx->path_stream.created || ctx->path_stream.in_flight.empty()) return RT_OK;
	ctx->time_this_sample = ctx->launch_timing;
	return stream_enqueue_iteration(ctx);
}

int rt_submissions_completed(rt_context * ctx, uint64_t * out_count) {
	RT_REQUIRE(ctx, ctx && out_count, "rt_submissions_completed: NULL argument");
	*out_count = ctx->path_stream.submissions_completed;
	return RT_OK;
}

int rt_synchronize(rt_context * ctx) {
	RT_REQUIRE(ctx, ctx, "rt_synchronize: NULL context");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	return RT_OK;
}

int rt_get_counters(rt_context * ctx, rt_counters * out) {
	RT_REQUIRE(ctx, ctx && out, "rt_get_counters: NULL argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx));
	rt_counters c; memset(&c, 0, sizeof(c));
	float ms = 0.0f;
	if (ctx->last_render_merged && ctx->path_stream.last_completed_ring >= 0) { // the last submission the merged wavefront completed
		const PathStream & s = ctx->path_stream;
		const int * row = s.stats_host + size_t(s.last_completed_ring) * RT_STREAM_STATS_ROW;
		memcpy(c.trace,      row + RT_STAT_TRACE      * RT_MAX_BOUNCES, sizeof(c.trace));
		memcpy(c.shadow,     row + RT_STAT_SHADOW     * RT_MAX_BOUNCES, sizeof(c.shadow));
		memcpy(c.diffuse,    row + RT_STAT_DIFFUSE    * RT_MAX_BOUNCES, sizeof(c.diffuse));
		memcpy(c.plastic,    row + RT_STAT_PLASTIC    * RT_MAX_BOUNCES, sizeof(c.plastic));
		memcpy(c.dielectric, row + RT_STAT_DIELECTRIC * RT_MAX_BOUNCES, sizeof(c.dielectric));
		memcpy(c.conductor,  row + RT_STAT_CONDUCTOR  * RT_MAX_BOUNCES, sizeof(c.conductor));
		if (hipEventElapsedTime(&ms, s.ev_begin[s.last_completed_ring], s.ev_end[s.last_completed_ring]) == hipSuccess) c.ms_total = ms;
	} else {
		const SampleSlot & slot = ctx->slots[ctx->last_slot < 0 ? 0 : ctx->last_slot];
		const int * totals = (const int *)slot.pinned_counters;
		memcpy(c.trace,      totals + 0 * RT_MAX_BOUNCES, sizeof(c.trace));
		memcpy(c.shadow,     totals + 1 * RT_MAX_BOUNCES, sizeof(c.shadow));
		memcpy(c.diffuse,    totals + 2 * RT_MAX_BOUNCES, sizeof(c.diffuse));
		memcpy(c.plastic,    totals + 3 * RT_MAX_BOUNCES, sizeof(c.plastic));
		memcpy(c.dielectric, totals + 4 * RT_MAX_BOUNCES, sizeof(c.dielectric));
		memcpy(c.conductor,  totals + 5 * RT_MAX_BOUNCES, sizeof(c.conductor));
		if (hipEventElapsedTime(&ms, slot.ev_frame_start, slot.ev_frame_end) == hipSuccess) c.ms_total = ms;
	}
	if (ctx->launch_timing) { // sums over every launch since the mode was enabled (rt_get_launch_timings returns each and starts over)
		for (size_t i = 0; i + 1 < ctx->span_used; i += 2) {
			float d = 0.0f;
			if (hipEventElapsedTime(&d, ctx->span_events[i], ctx->span_events[i + 1]) == hipSuccess) { if (ctx->span_kinds[i] == SPAN_TRACE) c.ms_trace += d; else if (ctx->span_kinds[i] == SPAN_SHADOW) c.ms_shadow += d; }
		}
	}
	if (ctx->profiling) {
		float * bucket[STAGE_END] = { &c.ms_generate, &c.ms_trace, &c.ms_sort, &c.ms_shade, &c.ms_shadow, &c.ms_post };
		static const bool print_stages = getenv("GRT_STAGE_TRACE") != nullptr; // one line per launch group, in submission order
		static const char * stage_names[STAGE_END] = { "generate", "trace", "sort", "shade", "shadow", "post" };
		for (size_t i = 0; i + 1 < ctx->stage_used; i++) {
			float d = 0.0f;
			if (hipEventElapsedTime(&d, ctx->stage_events[i], ctx->stage_events[i + 1]) == hipSuccess && ctx->stage_kinds[i] < STAGE_END) {
				*bucket[ctx->stage_kinds[i]] += d;
				if (print_stages) fprintf(stderr, "[grt] stage %-8s %8.4f ms\n", stage_names[ctx->stage_kinds[i]], d);
			}
		}
		if (ctx->last_render_merged) ctx->stage_used = 0; // the merged wavefront's stage events add up over its iterations until they are read
	}
	ctx->last_counters = c;
	*out = c;
	return RT_OK;
}

// ---- results ---------------------------------------------------------------------------------------------------

int rt_read_aov(rt_context * ctx, int aov_type, float * dst, int accumulated) {
	RT_REQUIRE(ctx, ctx && dst && aov_type >= 0 && aov_type < RT_AOV_COUNT, "rt_read_aov: invalid argument");
	(void)hipSetDevice(ctx->device);
	void * src = ctx->aov_buffers[aov_type][accumulated ? 1 : 0];
	if (!src) return fail(ctx, RT_ERROR_NOT_READY, "rt_read_aov: AOV %d is not enabled", aov_type);
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemcpy(dst, src, ctx->frame_pixels * 16, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_read_framebuffer(rt_context * ctx, float * dst) {
	RT_REQUIRE(ctx, ctx && dst, "rt_read_framebuffer: NULL argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->final_image) return fail(ctx, RT_ERROR_NOT_READY, "rt_read_framebuffer: rt_resize was not called");
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemcpy(dst, ctx->final_image, ctx->frame_pixels * 16, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_framebuffer_device_ptr(rt_context * ctx, void ** out_ptr, size_t * out_bytes) {
	RT_REQUIRE(ctx, ctx && out_ptr && out_bytes, "rt_framebuffer_device_ptr: NULL argument");
	if (!ctx->final_image) return fail(ctx, RT_ERROR_NOT_READY, "rt_framebuffer_device_ptr: rt_resize was not called");
	*out_ptr = ctx->final_image;
	*out_bytes = ctx->frame_pixels * 16;
	return RT_OK;
}

int rt_screen_pitch(rt_context * ctx) { return ctx ? ctx->params.screen_pitch : 0; }

int rt_read_luts(rt_context * ctx, float * dielectric_dir_enter, float * dielectric_dir_leave, float * dielectric_enter, float * dielectric_leave, float * conductor_dir, float * conductor) {
	RT_REQUIRE(ctx, ctx, "rt_read_luts: NULL context");
	(void)hipSetDevice(ctx->device);
	if (!ctx->params.pmj_samples) return fail(ctx, RT_ERROR_NOT_READY, "rt_read_luts: RNG tables not uploaded");
	int s = ensure_luts(ctx); if (s) return s;
	RT_HIP(ctx, quiesce(ctx));
	float * dst[6] = { dielectric_dir_enter, dielectric_dir_leave, dielectric_enter, dielectric_leave, conductor_dir, conductor };
	const size_t bytes[6] = { 4096 * 4, 4096 * 4, 256 * 4, 256 * 4, 1024 * 4, 32 * 4 };
	for (int i = 0; i < 6; i++) if (dst[i]) RT_HIP(ctx, hipMemcpy(dst[i], ctx->luts[i], bytes[i], hipMemcpyDeviceToHost));
	return RT_OK;
}

// ---- kernel-level entry points ------------------------------------------------------------------------------------

struct TempBuffers {
	rt_context * ctx; std::vector<void *> ptrs;
	explicit TempBuffers(rt_context * c) : ctx(c) { }
	~TempBuffers() { for (void * p : ptrs) (void)hipFree(p); }
	void * get(size_t bytes, const void * src) {
		void * p = nullptr;
		if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
		ptrs.push_back(p);
		if (src && bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
		return p;
	}
};



int rt_trace_rays(rt_context * ctx, const float * ox, const float * oy, const float * oz,
                  const float * dx, const float * dy, const float * dz, size_t ray_count,
                  uint32_t * hits, int repeat, float * out_ms) {
	RT_REQUIRE(ctx, ctx && ox && oy && oz && dx && dy && dz && hits, "rt_trace_rays: NULL argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx)); // slot 0's spill area and cursors are borrowed
	if (!ctx->params.triangles || !bvh_nodes_present(ctx) || !ctx->params.mesh_bvh_root_indices) return fail(ctx, RT_ERROR_NOT_READY, "rt_trace_rays: geometry / instances not uploaded");
	TempBuffers tmp(ctx);
	size_t bytes = ray_count * 4;
	RtVec3SoA o = { (float *)tmp.get(bytes, ox), (float *)tmp.get(bytes, oy), (float *)tmp.get(bytes, oz) };
	RtVec3SoA d = { (float *)tmp.get(bytes, dx), (float *)tmp.get(bytes, dy), (float *)tmp.get(bytes, dz) };
	uint4 * dev_hits = (uint4 *)tmp.get(ray_count * 16, nullptr);
	if (!o.x || !o.y || !o.z || !d.x || !d.y || !d.z || !dev_hits) return fail(ctx, RT_ERROR_HIP, "rt_trace_rays: device allocation failed");

	if (repeat < 1) repeat = 1;
	hipEvent_t e0, e1; RT_HIP(ctx, hipEventCreate(&e0)); RT_HIP(ctx, hipEventCreate(&e1));
	float total = 0.0f;
	for (int r = 0; r < repeat; r++) {
		RT_HIP(ctx, hipMemsetAsync(ctx->explicit_retired, 0, 8 * sizeof(int), ctx->stream));
		RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
		rt_launch_trace_explicit(ctx->params, o, d, dev_hits, int(ray_count), ctx->explicit_retired, ctx->stream);
		RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
		RT_HIP(ctx, quiesce(ctx));
		float ms = 0.0f; RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
		total += ms;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	RT_HIP(ctx, hipGetLastError());
	if (out_ms) *out_ms = total / float(repeat);
	RT_HIP(ctx, hipMemcpy(hits, dev_hits, ray_count * 16, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_trace_shadow_rays(rt_context * ctx, const float * ox, const float * oy, const float * oz,
                         const float * dx, const float * dy, const float * dz, const float * max_distance,
                         size_t ray_count, uint8_t * occluded, int repeat, float * out_ms) {
	RT_REQUIRE(ctx, ctx && ox && oy && oz && dx && dy && dz && max_distance && occluded, "rt_trace_shadow_rays: NULL argument");
	(void)hipSetDevice(ctx->device);
	RT_HIP(ctx, quiesce(ctx)); // slot 0's spill area and cursors are borrowed
	if (!ctx->params.triangles || !bvh_nodes_present(ctx) || !ctx->params.mesh_bvh_root_indices) return fail(ctx, RT_ERROR_NOT_READY, "rt_trace_shadow_rays: geometry / instances not uploaded");
	TempBuffers tmp(ctx);
	size_t bytes = ray_count * 4;
	RtVec3SoA o = { (float *)tmp.get(bytes, ox), (float *)tmp.get(bytes, oy), (float *)tmp.get(bytes, oz) };
	RtVec3SoA d = { (float *)tmp.get(bytes, dx), (float *)tmp.get(bytes, dy), (float *)tmp.get(bytes, dz) };
	float * dev_max = (float *)tmp.get(bytes, max_distance);
	uint8_t * dev_occ = (uint8_t *)tmp.get(ray_count, nullptr);
	if (!o.x || !o.y || !o.z || !d.x || !d.y || !d.z || !dev_max || !dev_occ) return fail(ctx, RT_ERROR_HIP, "rt_trace_shadow_rays: device allocation failed");

	if (repeat < 1) repeat = 1;
	hipEvent_t e0, e1; RT_HIP(ctx, hipEventCreate(&e0)); RT_HIP(ctx, hipEventCreate(&e1));
	float total = 0.0f;
	for (int r = 0; r < repeat; r++) {
		RT_HIP(ctx, hipMemsetAsync(ctx->explicit_retired, 0, 8 * sizeof(int), ctx->stream));
		RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
		rt_launch_trace_shadow_explicit(ctx->params, o, d, dev_max, dev_occ, int(ray_count), ctx->explicit_retired, ctx->stream);
		RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
		RT_HIP(ctx, quiesce(ctx));
		float ms = 0.0f; RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
		total += ms;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	RT_HIP(ctx, hipGetLastError());
	if (out_ms) *out_ms = total / float(repeat);
	RT_HIP(ctx, hipMemcpy(occluded, dev_occ, ray_count, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_generate_rays(rt_context * ctx, int sample_index, int pixel_offset, int pixel_count,
                     float * ox, float * oy, float * oz, float * dx, float * dy, float * dz, uint32_t * pixel_index_and_flags) {
	RT_REQUIRE(ctx, ctx && ox && oy && oz && dx && dy && dz && pixel_index_and_flags, "rt_generate_rays: NULL argument");
	RT_REQUIRE(ctx, pixel_count >= 0, "rt_generate_rays: negative pixel_count");
	(void)hipSetDevice(ctx->device);
	if (!ctx->params.pmj_samples || ctx->frame_pixels == 0) return fail(ctx, RT_ERROR_NOT_READY, "rt_generate_rays: RNG tables not uploaded or rt_resize not called");
	RT_HIP(ctx, quiesce(ctx)); // slot 0's queues are borrowed
	int s = ensure_queues(ctx); if (s) return s;
	if (size_t(pixel_count) > ctx->slots[0].queue_capacity) return fail(ctx, RT_ERROR_OUT_OF_RANGE, "rt_generate_rays: pixel_count %d exceeds the queue capacity %zu", pixel_count, ctx->slots[0].queue_capacity);
	rt_launch_generate(ctx->params, sample_index, pixel_offset, pixel_count, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	RT_HIP(ctx, quiesce(ctx));
	const RtTraceBuffer & t = ctx->params.trace[0];
	size_t bytes = size_t(pixel_count) * 4;
	RT_HIP(ctx, hipMemcpy(ox, t.origin.x, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(oy, t.origin.y, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(oz, t.origin.z, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(dx, t.direction.x, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(dy, t.direction.y, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(dz, t.direction.z, bytes, hipMemcpyDeviceToHost));
	RT_HIP(ctx, hipMemcpy(pixel_index_and_flags, t.pixel_index_and_flags, bytes, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_random_samples(rt_context * ctx, int dimension, const uint32_t * pixel_indices, size_t count, uint32_t bounce, uint32_t sample_index, float * out_xy) {
	RT_REQUIRE(ctx, ctx && pixel_indices && out_xy && dimension >= 0 && dimension < 7, "rt_random_samples: invalid argument");
	(void)hipSetDevice(ctx->device);
	if (!ctx->params.pmj_samples || ctx->params.screen_pitch == 0) return fail(ctx, RT_ERROR_NOT_READY, "rt_random_samples: RNG tables not uploaded or rt_resize not called");
	TempBuffers tmp(ctx);
	unsigned * dev_px = (unsigned *)tmp.get(count * 4, pixel_indices);
	float2 * dev_out = (float2 *)tmp.get(count * 8, nullptr);
	if (!dev_px || !dev_out) return fail(ctx, RT_ERROR_HIP, "rt_random_samples: device allocation failed");
	rt_launch_random(ctx->params, dimension, dev_px, int(count), bounce, sample_index, dev_out, ctx->stream);
	RT_HIP(ctx, hipGetLastError());
	RT_HIP(ctx, quiesce(ctx));
	RT_HIP(ctx, hipMemcpy(out_xy, dev_out, count * 8, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_measure_stream_bandwidth(rt_context * ctx, size_t bytes, int repeat, float * out_gbps) {
	RT_REQUIRE(ctx, ctx && out_gbps && bytes >= 1024, "rt_measure_stream_bandwidth: invalid argument");
	(void)hipSetDevice(ctx->device);
	TempBuffers tmp(ctx);
	size_t count = bytes / 16;
	float4 * src = (float4 *)tmp.get(count * 16, nullptr);
	float * sink = (float *)tmp.get(16, nullptr);
	if (!src || !sink) return fail(ctx, RT_ERROR_HIP, "rt_measure_stream_bandwidth: device allocation failed");
	RT_HIP(ctx, hipMemsetAsync(src, 0x3c, count * 16, ctx->stream));
	if (repeat < 1) repeat = 1;
	hipEvent_t e0, e1; RT_HIP(ctx, hipEventCreate(&e0)); RT_HIP(ctx, hipEventCreate(&e1));
	rt_launch_stream_read(src, count, sink, ctx->stream); // warm-up
	float best = 1e30f;
	for (int r = 0; r < repeat; r++) {
		RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
		rt_launch_stream_read(src, count, sink, ctx->stream);
		RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
		RT_HIP(ctx, quiesce(ctx));
		float ms = 0.0f; RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*out_gbps = float(double(count * 16) / (double(best) * 1e-3) / 1e9);
	return RT_OK;
}

} // extern "C"
