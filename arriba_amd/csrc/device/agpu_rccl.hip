// agpu_rccl.hip -- the two exchanges of "one sample over N GPUs" issued from inside the C ABI, for hosts that hold an RCCL communicator themselves
// (agpu_shard_merge(ctx, ncclComm_t, hipStream_t)).  Thin compositions of entry points that are tested on their own
// (agpu_shard_export / agpu_shard_merge, agpu_mismapper_jobs / _verdicts / agpu_filter_mismappers_apply) with ncclAllReduce / ncclAllGather on the context's
// stream in between.  librccl is looked up at run time: the library does not link against it, and nothing else in it depends on these two functions.
// Exercised on a communicator of one rank by tests/test_gpu_parity.py::test_rccl_compositions_with_one_rank (the GPU box has one GPU); the Python driver
// arriba_amd/one_sample.py issues the same collectives through torch.distributed.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <rccl/rccl.h>
#include "agpu_context.hpp"

using namespace agpu;

namespace {

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)
#define TRY(call) do { int s_ = (call); if (s_ != AGPU_OK) return s_; } while (0)

struct Rccl {
	ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*error_string)(ncclResult_t) = nullptr;
	ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
	ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
	bool load() {
		if (all_reduce != nullptr) return true;
		void* library = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (library == nullptr) library = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (library == nullptr) { set_last_error(std::string("librccl.so not found: ") + dlerror()); return false; }
		all_reduce = (decltype(all_reduce)) dlsym(library, "ncclAllReduce"); all_gather = (decltype(all_gather)) dlsym(library, "ncclAllGather"); error_string = (decltype(error_string)) dlsym(library, "ncclGetErrorString");
		get_unique_id = (decltype(get_unique_id)) dlsym(library, "ncclGetUniqueId"); comm_init_rank = (decltype(comm_init_rank)) dlsym(library, "ncclCommInitRank"); comm_destroy = (decltype(comm_destroy)) dlsym(library, "ncclCommDestroy");
		if (all_reduce == nullptr || all_gather == nullptr || get_unique_id == nullptr || comm_init_rank == nullptr || comm_destroy == nullptr) { set_last_error("librccl.so lacks ncclAllReduce / ncclAllGather / ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy"); all_reduce = nullptr; return false; }
		return true;
	}
	int check(ncclResult_t status, const char* what) const {
		if (status == ncclSuccess) return AGPU_OK;
		set_last_error(std::string(what) + ": " + (error_string ? error_string(status) : "RCCL error"));
		return AGPU_ERR_DEVICE;
	}
};
Rccl g_rccl;

}

// How the ranks fared with the local work in front of a collective: one all-reduce (min) of a status word, which every rank reaches whatever happened to it -- a rank that could
// not allocate its buffers (22 GB of parts at 10^8 fragments) must not leave the others waiting in the all-gather behind them (advisor, round 5).  Returns the local status if it
// is a failure, AGPU_ERR_DEVICE with "another rank failed" if another rank's is, AGPU_OK otherwise.
static int agree(agpu_ctx* ctx, ncclComm_t comm, int local_status, const char* what) {
	hipStream_t s = ctx->stream;
	const std::string local_error = local_status != AGPU_OK ? std::string(agpu_last_error()) : std::string();
	DeviceBuffer& word = ctx->scratch("rccl.status"); // (16 bytes, allocated with the communicator: agpu_rccl_join; here for callers that bring their own)
	if (word.ptr == nullptr && !word.allocate(16)) { set_last_error("hipMalloc failed (rccl.status)"); return AGPU_ERR_NO_MEMORY; }
	int64_t ok = local_status == AGPU_OK ? 1 : 0;
	HIP_CHECK(hipMemcpyAsync(word.ptr, &ok, 8, hipMemcpyHostToDevice, s));
	TRY(g_rccl.check(g_rccl.all_reduce(word.ptr, word.ptr, 1, ncclInt64, ncclMin, comm, s), "ncclAllReduce(status of the ranks)"));
	HIP_CHECK(hipMemcpyAsync(&ok, word.ptr, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (local_status != AGPU_OK) { set_last_error(local_error); return local_status; }
	if (ok == 0) { set_last_error(std::string("another rank of the sample failed in front of ") + what + " (its own message says why)"); return AGPU_ERR_DEVICE; }
	return AGPU_OK;
}

extern "C" int agpu_shard_merge_rccl(agpu_ctx* ctx, void* nccl_comm, uint32_t n_ranks, agpu_ingest_result* result) {
	if (!ctx || !nccl_comm || n_ranks == 0) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	ncclComm_t comm = (ncclComm_t) nccl_comm;
	// phase 1 (local): the size of this part
	uint64_t bytes = 0;
	DeviceBuffer& widest = ctx->scratch("rccl.widest");
	auto size_of_part = [&]() -> int { TRY(agpu_shard_export_size(ctx, &bytes)); ALLOC(widest, 8); return AGPU_OK; };
	TRY(agree(ctx, comm, size_of_part(), "the exchange of the sizes of the parts"));
	// the blocks travel at the stride of the largest part
	HIP_CHECK(hipMemcpyAsync(widest.ptr, &bytes, 8, hipMemcpyHostToDevice, s));
	TRY(g_rccl.check(g_rccl.all_reduce(widest.ptr, widest.ptr, 1, ncclUint64, ncclMax, comm, s), "ncclAllReduce(size of the parts)"));
	uint64_t stride = 0;
	HIP_CHECK(hipMemcpyAsync(&stride, widest.ptr, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	stride = (stride + 15) & ~(uint64_t) 15;
	// phase 2 (local): the buffers of the all-gather (n_ranks x stride: the likeliest allocation to fail) and this part exported into its send buffer
	DeviceBuffer& mine = ctx->scratch("rccl.part"); DeviceBuffer& all = ctx->scratch("rccl.parts");
	auto export_part = [&]() -> int { ALLOC(mine, stride); ALLOC(all, (size_t) n_ranks * stride); TRY(agpu_shard_export(ctx, mine.ptr, stride)); return AGPU_OK; };
	{ const int status = agree(ctx, comm, export_part(), "the all-gather of the parts"); if (status != AGPU_OK) { mine.release(); all.release(); return status; } }
	TRY(g_rccl.check(g_rccl.all_gather(mine.ptr, all.ptr, stride, ncclUint8, comm, s), "ncclAllGather(parts of the sample)"));
	HIP_CHECK(hipStreamSynchronize(s));
	const int status = agpu_shard_merge(ctx, all.ptr, stride, n_ranks, result);
	mine.release(); all.release();
	return status;
}

extern "C" int agpu_filter_mismappers_rccl(agpu_ctx* ctx, void* nccl_comm, int32_t max_mate_gap, uint32_t rank, uint32_t n_ranks, uint64_t* remaining, uint64_t* discarded_reads) {
	if (!ctx || !nccl_comm || n_ranks == 0 || rank >= n_ranks) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	uint64_t n_jobs = 0;
	DeviceBuffer& verdicts = ctx->scratch("rccl.verdicts");
	auto search = [&]() -> int { // (local: the jobs, the buffer of the verdicts, this rank's share of the re-alignments)
		TRY(agpu_mismapper_jobs(ctx, &n_jobs));
		ALLOC(verdicts, std::max<uint64_t>(n_jobs, 1));
		HIP_CHECK(hipMemsetAsync(verdicts.ptr, 0, std::max<uint64_t>(n_jobs, 1), s));
		HIP_CHECK(hipStreamSynchronize(s));
		TRY(agpu_mismapper_verdicts(ctx, max_mate_gap, rank, n_ranks, verdicts.as<uint8_t>()));
		return AGPU_OK;
	};
	TRY(agree(ctx, (ncclComm_t) nccl_comm, search(), "the all-reduce of the verdicts of filter_mismappers"));
	if (n_jobs > 0) TRY(g_rccl.check(g_rccl.all_reduce(verdicts.ptr, verdicts.ptr, n_jobs, ncclUint8, ncclMax, (ncclComm_t) nccl_comm, s), "ncclAllReduce(verdicts)"));
	HIP_CHECK(hipStreamSynchronize(s));
	return agpu_filter_mismappers_apply(ctx, verdicts.as<uint8_t>(), remaining, discarded_reads);
}

// ... and the collectives of the read-sharded split (agpu_sharded.hip; the C++ driver: workflow.cpp gather_bytes / sum over the ranks) over DEVICE memory: the emissions of
// find_fusions, the states of the reads, the winners of the duplicate keys and coverage_t stay in HBM from the kernel that wrote them to the kernel that reads them
extern "C" int agpu_rccl_all_gather_device(agpu_ctx* ctx, void* nccl_comm, const void* mine, void* all, uint64_t bytes) {
	if (!ctx || !nccl_comm || (bytes > 0 && (!mine || !all))) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (bytes == 0) return AGPU_OK;
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	TRY(g_rccl.check(g_rccl.all_gather(mine, all, bytes, ncclUint8, (ncclComm_t) nccl_comm, ctx->stream), "ncclAllGather(device bytes)"));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}
extern "C" int agpu_rccl_all_reduce_device(agpu_ctx* ctx, void* nccl_comm, void* values, uint64_t count, int kind) {
	if (!ctx || !nccl_comm || (count > 0 && !values) || kind < AGPU_REDUCE_MAX_INT64 || kind > AGPU_REDUCE_SUM_UINT32) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (count == 0) return AGPU_OK;
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	const ncclDataType_t type = kind == AGPU_REDUCE_MAX_BYTES ? ncclUint8 : kind == AGPU_REDUCE_SUM_UINT32 ? ncclUint32 : ncclInt64;
	const ncclRedOp_t operation = kind == AGPU_REDUCE_MIN_INT64 ? ncclMin : (kind == AGPU_REDUCE_SUM_INT64 || kind == AGPU_REDUCE_SUM_UINT32) ? ncclSum : ncclMax;
	TRY(g_rccl.check(g_rccl.all_reduce(values, values, count, type, operation, (ncclComm_t) nccl_comm, ctx->stream), "ncclAllReduce(device values)"));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}

// ---- a communicator of RCCL alone for the hosts that have none (the C++ driver: include/arriba_workflow.h, arriba_workflow_join_rccl): the id of rank 0 travels by whatever
// started the ranks (an environment variable, a file, torch.distributed's store), every rank joins with it; the small exchanges of the driver (sizes, status words, the texts
// of the rows) are host memory bounced through a buffer of the context

static_assert(sizeof(ncclUniqueId) == AGPU_RCCL_ID_BYTES, "include/arriba_gpu.h: AGPU_RCCL_ID_BYTES");

extern "C" int agpu_rccl_unique_id(uint8_t* id) {
	if (!id) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	ncclUniqueId unique;
	TRY(g_rccl.check(g_rccl.get_unique_id(&unique), "ncclGetUniqueId"));
	memcpy(id, &unique, sizeof(unique));
	return AGPU_OK;
}

extern "C" int agpu_rccl_join(agpu_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t n_ranks, void** nccl_comm) {
	if (!ctx || !id || !nccl_comm || n_ranks == 0 || rank >= n_ranks) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	ncclUniqueId unique;
	memcpy(&unique, id, sizeof(unique));
	ncclComm_t comm = nullptr;
	TRY(g_rccl.check(g_rccl.comm_init_rank(&comm, (int) n_ranks, unique, (int) rank), "ncclCommInitRank"));
	ALLOC(ctx->scratch("rccl.status"), 16); // (the word the ranks tell each other their status with: there before anything can run out of memory)
	*nccl_comm = comm;
	return AGPU_OK;
}

extern "C" int agpu_rccl_leave(void* nccl_comm) {
	if (!nccl_comm) return AGPU_OK;
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	return g_rccl.check(g_rccl.comm_destroy((ncclComm_t) nccl_comm), "ncclCommDestroy");
}

extern "C" int agpu_rccl_all_gather_host(agpu_ctx* ctx, void* nccl_comm, uint32_t n_ranks, const void* mine, void* all, uint64_t bytes) {
	if (!ctx || !nccl_comm || n_ranks == 0 || (bytes > 0 && (!mine || !all))) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (bytes == 0) return AGPU_OK;
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	DeviceBuffer& bounce = ctx->scratch("rccl.bounce");
	ALLOC(bounce, (size_t) (n_ranks + 1) * bytes);
	uint8_t* sent = bounce.as<uint8_t>() + (size_t) n_ranks * bytes;
	HIP_CHECK(hipMemcpyAsync(sent, mine, bytes, hipMemcpyHostToDevice, s));
	TRY(g_rccl.check(g_rccl.all_gather(sent, bounce.ptr, bytes, ncclUint8, (ncclComm_t) nccl_comm, s), "ncclAllGather(host bytes)"));
	HIP_CHECK(hipMemcpyAsync(all, bounce.ptr, (size_t) n_ranks * bytes, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if ((size_t) (n_ranks + 1) * bytes > ((size_t) 64 << 20)) bounce.release(); // (the texts of a large discarded.tsv: not kept)
	return AGPU_OK;
}

extern "C" int agpu_rccl_all_reduce_host(agpu_ctx* ctx, void* nccl_comm, void* values, uint64_t count, int kind) {
	if (!ctx || !nccl_comm || (count > 0 && !values) || kind < AGPU_REDUCE_MAX_INT64 || kind > AGPU_REDUCE_MAX_BYTES) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (count == 0) return AGPU_OK;
	if (!g_rccl.load()) return AGPU_ERR_DEVICE;
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const size_t bytes = (size_t) count * (kind == AGPU_REDUCE_MAX_BYTES ? 1 : 8);
	DeviceBuffer& bounce = ctx->scratch("rccl.bounce");
	ALLOC(bounce, bytes);
	HIP_CHECK(hipMemcpyAsync(bounce.ptr, values, bytes, hipMemcpyHostToDevice, s));
	const ncclDataType_t type = kind == AGPU_REDUCE_MAX_BYTES ? ncclUint8 : ncclInt64;
	const ncclRedOp_t operation = kind == AGPU_REDUCE_MIN_INT64 ? ncclMin : kind == AGPU_REDUCE_SUM_INT64 ? ncclSum : ncclMax;
	TRY(g_rccl.check(g_rccl.all_reduce(bounce.ptr, bounce.ptr, count, type, operation, (ncclComm_t) nccl_comm, s), "ncclAllReduce(host values)"));
	HIP_CHECK(hipMemcpyAsync(values, bounce.ptr, bytes, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	return AGPU_OK;
}
