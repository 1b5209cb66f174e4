// arriba_amd/csrc/device/agpu_sharded.hip -- one sample over the GPUs of a node with the READS SHARDED (include/arriba_gpu.h; SURVEY.md section 8 row e; BASELINE.json north_star:
// "BAM records shard naturally by read ... a single RCCL all-gather ... to merge breakpoint histograms before candidate scoring").
//
// A context keeps the fragments of the part of the file it ingested (agpu_shard_keep, agpu_ingest.hip).  find_fusions runs on every rank over the emissions of all ranks
// (agpu_find_fusions_from_emissions): the candidate table and every read list exist on every rank, the lists in global name ranks.  The stages behind it are of three kinds:
//   * they look at candidates alone (e-value, select_best, blacklist, ...): unchanged, the same on every rank;
//   * they judge a candidate by its read lists, asking of a read only its filter / multi-mapper flag / "an alignment is exonic" (filter_both_intronic, recover_both_spliced,
//     recover_internal_tandem_duplication, the recounts of filter_multimappers and filter_mismappers): they read the REPLICATED STATE of the reads, one byte per fragment of the
//     sample (this file: agpu_read_state_export -> one all-gather -> agpu_read_state_import; candidate_walk_batch hands the stages a view of it);
//   * they need the alignments of a read (scores of filter_multimappers, re-alignments of filter_mismappers, clipped mates of filter_in_vitro, gene sets for the expression proxy):
//     computed where the read is, over the reads [first_rank, first_rank + n) of the lists (the sharded forms in agpu_multimappers.hip, agpu_mismappers.hip, agpu_events.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include "agpu_context.hpp"
#include "device_utils.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

// what the walks over read lists ask of a read, in one byte per fragment (views.hpp: BatchView::walk)
__global__ void walk_byte_kernel(BatchView b, uint8_t* walk) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	uint8_t byte = b.filter[i] == FILTER_none ? WALK_UNFILTERED : 0;
	if (b.fbits[i] & FBIT_MULTIMAPPER) byte |= WALK_MULTIMAPPER;
	for (int slot = 0; slot < b.n_aln[i]; ++slot) if (b.abits[slot][i] & ABIT_EXONIC) byte |= WALK_EXONIC;
	walk[i] = byte;
}
// ... of the replicated states: the filter of the moment, the two bits that no stage changes
__global__ void walk_byte_of_state_kernel(const uint8_t* filter, const uint8_t* bits, uint64_t n, uint8_t* walk) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < n) walk[i] = (filter[i] == FILTER_none ? WALK_UNFILTERED : 0) | bits[i];
}
__global__ void read_state_export_kernel(BatchView b, uint8_t* state) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	uint8_t byte = b.filter[i] & 0x3F;
	if (b.fbits[i] & FBIT_MULTIMAPPER) byte |= AGPU_READ_STATE_MULTIMAPPER;
	for (int slot = 0; slot < b.n_aln[i]; ++slot) if (b.abits[slot][i] & ABIT_EXONIC) byte |= AGPU_READ_STATE_EXONIC;
	state[i] = byte;
}
// filters and bits of the sample from its states; *mismatch counts the own reads whose imported filter is not the one this context holds
__global__ void read_state_import_kernel(const uint8_t* state, uint64_t n, uint8_t* filter, uint8_t* bits, const uint8_t* own_filter, uint64_t first_rank, uint64_t own_n, unsigned int* mismatch) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint8_t byte = state[i];
	filter[i] = byte & 0x3F;
	bits[i] = ((byte & AGPU_READ_STATE_MULTIMAPPER) ? WALK_MULTIMAPPER : 0) | ((byte & AGPU_READ_STATE_EXONIC) ? WALK_EXONIC : 0);
	const uint64_t own = i - first_rank;
	if (own < own_n && own_filter[own] != (byte & 0x3F)) atomicAdd(mismatch, 1u);
}
__global__ void pull_filters_kernel(const uint8_t* global_filter, uint64_t first_rank, uint64_t n, uint8_t* filter) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < n) filter[i] = global_filter[first_rank + i];
}

}

int agpu::candidate_walk_batch(agpu_ctx* ctx, BatchView& batch, bool with_walk_bytes) {
	hipStream_t s = ctx->stream;
	if (!ctx->read_sharded) {
		batch = ctx->batch;
		if (!with_walk_bytes || ctx->n == 0) return AGPU_OK;
		DeviceBuffer& walk = ctx->scratch("events.walk_bytes"); // (the filters of the reads change from stage to stage: made anew by every stage that walks with them)
		ALLOC(walk, ctx->n);
		{ KernelTimer timer(ctx, "walk_byte_kernel", ctx->n * 7);
		  walk_byte_kernel<<<grid_for(ctx->n), BLOCK, 0, s>>>(ctx->batch, walk.as<uint8_t>()); }
		batch.walk = walk.as<uint8_t>();
		return AGPU_OK;
	}
	if (!ctx->state_imported) { set_last_error("the reads of the sample are sharded over the ranks: agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
	const uint64_t N = ctx->global_n;
	BatchView view;
	memset((void*) &view, 0, sizeof(view));
	view.n = N; view.first_rank = 0; view.filter = ctx->scratch("sharded.filter").as<uint8_t>();
	view.walk = nullptr;
	if (with_walk_bytes && N > 0) {
		DeviceBuffer& walk = ctx->scratch("events.walk_bytes");
		ALLOC(walk, N);
		{ KernelTimer timer(ctx, "walk_byte_kernel", N * 3);
		  walk_byte_of_state_kernel<<<grid_for(N), BLOCK, 0, s>>>(view.filter, ctx->scratch("sharded.bits").as<uint8_t>(), N, walk.as<uint8_t>()); }
		view.walk = walk.as<uint8_t>();
	}
	batch = view;
	return AGPU_OK;
}

int agpu::pull_filters_of_own_reads(agpu_ctx* ctx) {
	if (!ctx->read_sharded || ctx->n == 0) return AGPU_OK;
	pull_filters_kernel<<<grid_for(ctx->n), BLOCK, 0, ctx->stream>>>(ctx->scratch("sharded.filter").as<uint8_t>(), ctx->batch.first_rank, ctx->n, ctx->filter.as<uint8_t>());
	HIP_CHECK(hipGetLastError());
	return AGPU_OK;
}

extern "C" int agpu_read_state_export(agpu_ctx* ctx, uint8_t* state) {
	if (!ctx || !ctx->have_batch || !state) { set_last_error("no batch on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	if (ctx->n == 0) return AGPU_OK;
	DeviceBuffer& mine = ctx->scratch("sharded.state");
	ALLOC(mine, ctx->n);
	{ KernelTimer timer(ctx, "read_state_export_kernel", ctx->n * 7);
	  read_state_export_kernel<<<grid_for(ctx->n), BLOCK, 0, s>>>(ctx->batch, mine.as<uint8_t>()); }
	HIP_CHECK(hipMemcpyAsync(state, mine.ptr, ctx->n, hipMemcpyDefault, s)); // (host or device memory: the send buffer of the all-gather)
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

extern "C" int agpu_read_state_import(agpu_ctx* ctx, const uint8_t* state) {
	if (!ctx || !ctx->have_batch || !ctx->read_sharded) { set_last_error("agpu_shard_keep must run first"); return AGPU_ERR_INVALID; }
	const uint64_t N = ctx->global_n;
	if (N > 0 && !state) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	DeviceBuffer& all = ctx->scratch("sharded.states_of_all"); DeviceBuffer& filter = ctx->scratch("sharded.filter"); DeviceBuffer& bits = ctx->scratch("sharded.bits"); DeviceBuffer& counter = ctx->scratch("sharded.counter");
	ALLOC(all, std::max<uint64_t>(N, 1)); ALLOC(filter, std::max<uint64_t>(N, 1)); ALLOC(bits, std::max<uint64_t>(N, 1)); ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	unsigned int mismatch = 0;
	if (N > 0) {
		HIP_CHECK(hipMemcpyAsync(all.ptr, state, N, hipMemcpyDefault, s));
		{ KernelTimer timer(ctx, "read_state_import_kernel", N * 3 + ctx->n);
		  read_state_import_kernel<<<grid_for(N), BLOCK, 0, s>>>(all.as<uint8_t>(), N, filter.as<uint8_t>(), bits.as<uint8_t>(), ctx->filter.as<uint8_t>(), ctx->batch.first_rank, ctx->n, counter.as<unsigned int>()); }
		HIP_CHECK(hipMemcpyAsync(&mismatch, counter.ptr, 4, hipMemcpyDeviceToHost, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	if (mismatch != 0) { set_last_error("the imported read states do not hold the filters of this context's own reads (" + std::to_string(mismatch) + " differ)"); return AGPU_ERR_INVALID; }
	ctx->state_imported = true;
	return AGPU_OK;
}

extern "C" int agpu_scratch_buffer(agpu_ctx* ctx, const char* name, uint64_t bytes, void** pointer) {
	if (!ctx || !name || !pointer || strncmp(name, "exchange.", 9) != 0) { set_last_error("agpu_scratch_buffer: a context, a name that starts with \"exchange.\" and a place for the pointer"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	DeviceBuffer& buffer = ctx->scratch(name);
	ALLOC(buffer, std::max<uint64_t>(bytes, 16));
	*pointer = buffer.ptr;
	return AGPU_OK;
}
extern "C" int agpu_device_copy(agpu_ctx* ctx, void* destination, const void* source, uint64_t bytes) {
	if (!ctx || (bytes > 0 && (!destination || !source))) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (bytes > 0) { HIP_CHECK(hipMemcpyAsync(destination, source, bytes, hipMemcpyDefault, ctx->stream)); HIP_CHECK(hipStreamSynchronize(ctx->stream)); }
	return AGPU_OK;
}
