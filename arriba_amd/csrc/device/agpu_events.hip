// arriba_amd/csrc/device/agpu_events.hip -- event-level predicates behind filter_relative_support on the device:
// filter_both_intronic, filter_short_anchor, filter_end_to_end_fusions, filter_no_coverage (event_core.hpp), and the upload of coverage_t.
// One thread per candidate; the tallies go through one atomic per workgroup (device_utils.hpp).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include "agpu_context.hpp"
#include "event_core.hpp"
#include "device_utils.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_DEVICE; } } while (0)

__global__ void event_predicate_kernel(int stage, BatchView b, AnnotationView ann, GenomeView genome, CoverageView coverage, CandidateTable t, uint32_t min_anchor_length, unsigned int* remaining) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		if (t.filter[c] != FILTER_none) continue;
		const uint8_t verdict = event_predicate(stage, b, ann, genome, coverage, t, c, min_anchor_length);
		if (verdict == FILTER_none) ++kept; else if (verdict != EVENT_KEPT_UNCOUNTED) t.filter[c] = verdict;
	}
	block_tally(kept, remaining, &block_sum);
}

int run_event_stage(agpu_ctx* ctx, int stage, uint8_t filter_id, const char* kernel_name, uint32_t min_anchor_length, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (stage == EVENT_both_intronic && ctx->candidates_imported) { set_last_error("filter_both_intronic reads the read lists: not available on an imported (replicated) candidate table"); return AGPU_ERR_INVALID; }
	if ((stage == EVENT_no_coverage || stage == EVENT_marginal_read_through) && !ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter");
	ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		const int effective_stage = ctx->params.filter_enabled[filter_id] ? stage : EVENT_count_only; // a stage switched off with -f only counts
		KernelTimer timer(ctx, kernel_name, (uint64_t) C * 60 + (stage == EVENT_both_intronic ? (uint64_t) ctx->n_list_entries * 8 : 0));
		event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(effective_stage, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, min_anchor_length, counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 60;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

}

extern "C" int agpu_upload_coverage(agpu_ctx* ctx, const agpu_coverage_view* in) {
	if (!ctx || !in) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t windows = in->n_contigs ? in->window_offset[in->n_contigs] : 0;
	ALLOC(ctx->coverage_window_offset, ((size_t) in->n_contigs + 1) * 8); ALLOC(ctx->coverage_windows, std::max<uint64_t>(windows, 1) * 2);
	ALLOC(ctx->coverage_fragment_starts, std::max<uint64_t>(windows, 1)); ALLOC(ctx->coverage_fragment_ends, std::max<uint64_t>(windows, 1));
	HIP_CHECK(hipMemcpyAsync(ctx->coverage_window_offset.ptr, in->window_offset, ((size_t) in->n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
	if (windows > 0) {
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_windows.ptr, in->coverage, windows * 2, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_starts.ptr, in->fragment_starts, windows, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_ends.ptr, in->fragment_ends, windows, hipMemcpyHostToDevice, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->coverage.n_contigs = in->n_contigs; ctx->coverage.window_offset = ctx->coverage_window_offset.as<uint64_t>(); ctx->coverage.coverage = ctx->coverage_windows.as<uint16_t>();
	ctx->coverage.fragment_starts = ctx->coverage_fragment_starts.as<uint8_t>(); ctx->coverage.fragment_ends = ctx->coverage_fragment_ends.as<uint8_t>();
	ctx->have_coverage = true;
	return AGPU_OK;
}

extern "C" int agpu_filter_both_intronic(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_both_intronic, FILTER_intronic, "event_predicate_kernel(both_intronic)", 0, remaining); }
extern "C" int agpu_filter_short_anchor(agpu_ctx* ctx, uint32_t min_length, uint64_t* remaining) { return run_event_stage(ctx, EVENT_short_anchor, FILTER_short_anchor, "event_predicate_kernel(short_anchor)", min_length, remaining); }
extern "C" int agpu_filter_end_to_end(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_end_to_end, FILTER_end_to_end, "event_predicate_kernel(end_to_end)", 0, remaining); }
extern "C" int agpu_filter_no_coverage(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_no_coverage, FILTER_no_coverage, "event_predicate_kernel(no_coverage)", 0, remaining); }
extern "C" int agpu_filter_marginal_read_through(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_marginal_read_through, FILTER_marginal_read_through, "event_predicate_kernel(marginal_read_through)", 0, remaining); }
