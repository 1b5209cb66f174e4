// arriba_amd/csrc/device/agpu_context.hpp -- device context: owns every HBM buffer of the hot path.
#ifndef AGPU_CONTEXT_HPP
#define AGPU_CONTEXT_HPP 1

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <string>
#include <vector>
#include "../../../include/arriba_gpu.h"
#include "evalue_core.hpp"
#include "event_core.hpp"

namespace agpu {

void set_last_error(const std::string& message);

bool debug_allocation_fails(); // agpu_api.hip: a test asked for the next allocation inside agpu_ingest_finish to fail once (agpu_debug_fail_allocation_in_finish)
void note_failed_allocation(size_t bytes);
bool debug_finish_runs_out_of_memory(struct ::agpu_ctx* ctx); // agpu_api.hip: agpu_debug_exhaust_memory_in_finish // agpu_api.hip: the size asked for and what the device has free go into the next "hipMalloc failed" message of this thread
struct DeviceBuffer {
	void* ptr = nullptr;
	size_t bytes = 0;
	DeviceBuffer() {}
	DeviceBuffer(const DeviceBuffer&) = delete;
	DeviceBuffer& operator=(const DeviceBuffer&) = delete;
	~DeviceBuffer() { release(); }
	// grow-only: a buffer that is already large enough is reused, so that repeated passes over batches of the same size
	// (and the stage calls inside one pass) do not go back to hipMalloc/hipFree; `bytes` is the size asked for last
	size_t capacity = 0;
	bool allocate(size_t n) {
		if (n == 0) n = 16;
		if (ptr != nullptr && n <= capacity) { bytes = n; return true; }
		release();
		if (debug_allocation_fails() || hipMalloc(&ptr, n) != hipSuccess) {
			// out of memory: the contexts give back what they only keep for the next sample (the stream and the tables of the last ingest while the stages run, the
			// buffers of the stages while an ingest runs), then once more
			(void) hipGetLastError();
			ptr = nullptr;
			if (!release_idle_buffers() || hipMalloc(&ptr, n) != hipSuccess) { (void) hipGetLastError(); ptr = nullptr; note_failed_allocation(n); return false; }
		}
		bytes = n; capacity = n;
		return true;
	}
	static bool release_idle_buffers(); // agpu_api.hip; true if anything was given back
	// a view of a part of another buffer (find_fusions carves its working arrays out of a buffer that is idle at that point): nothing is freed with it
	bool borrowed = false;
	void borrow(void* memory, size_t n) { release(); ptr = memory; bytes = n; capacity = n; borrowed = true; }
	void release() {
		if (!ptr) return;
		if (!borrowed) (void) hipFree(ptr);
		ptr = nullptr; bytes = 0; capacity = 0; borrowed = false;
	}
	void swap(DeviceBuffer& other) { std::swap(ptr, other.ptr); std::swap(bytes, other.bytes); std::swap(capacity, other.capacity); std::swap(borrowed, other.borrowed); }
	template <class T> T* as() const { return (T*) ptr; }
};

// per-kernel timing with HIP events on the launch stream (agpu_set_profiling / agpu_get_kernel_profile)
struct KernelSample { const char* name; hipEvent_t start, stop; uint64_t bytes; float ms; unsigned int epoch; };

// The front of read_chimeric_alignments on the device while the pieces of the file are still arriving (agpu_ingest.hip, "the front as the pieces arrive"): the stream is
// cut into windows, one per pushed piece; a window goes through four steps on a stream of its own (record chain; offsets, record keys, active records; runs of one name;
// the loop body of the reference per name), each step behind the read-back of the few words the next one needs to size its launches.  The thread that pushes never waits
// for them: a step is enqueued when the words of the one before have arrived (hipEventQuery).
struct IngestWindow {
	uint64_t avail = 0;                     // bytes of the stream in HBM when the window was made: the size of the stream for its kernels
	uint64_t segment_begin = 0, segment_end = 0;
	bool last = false;                      // made by agpu_ingest_finish: reaches to the end of the stream
	int enqueued = 0, known = 0;            // steps enqueued / steps whose words were read
	hipEvent_t readback = nullptr;          // behind the read-back of the step enqueued last
	unsigned int slot = 0;                  // of the ring of read-back words
	uint64_t record_begin = 0, record_end = 0, active_begin = 0, active_end = 0, head_begin = 0, head_end = 0;
};
struct IngestProgress {
	bool on = false, abandoned = false, touched = false; // abandoned: something the windows cannot decide (a read name in two places, a record longer than the margin, ...): the whole stream is done again at the end
	hipStream_t work = nullptr;
	std::deque<IngestWindow> windows;
	std::vector<hipEvent_t> events;         // free ones
	uint32_t* host_words = nullptr;         // pinned, INGEST_WINDOW_RING x INGEST_WINDOW_WORDS
	unsigned int next_slot = 0;
	uint64_t window_bytes = 0;              // bytes of the stream the last window was made at
	uint64_t segments_done = 0, records = 0, active = 0, heads = 0, groups_done = 0, size_hint = 0;
	unsigned int windows_made = 0;
};

}

enum { AGPU_PIECE_SLOTS = 4 };
namespace agpu {
// The scratch buffers of a context, addressed by name, grow-only.  Two contexts of one device may share a pool (agpu_create_sibling): the lanes of a resident session that feeds
// the file of the next sample while the stages of the current one run (include/arriba_workflow.h: arriba_workflow_submit).  The two never use the same names at the same time -- one
// is between agpu_ingest_begin and agpu_ingest_finish ("ingest.*", the stream, the raw pieces), the other in its stages -- so ~150 GB of tables exist once, not twice.
struct ScratchPool {
	std::mutex mutex; // (of the map: buffers are looked up from the thread that feeds and from the thread that runs the stages)
	std::map<std::string, DeviceBuffer> buffers;
	DeviceBuffer& get(const char* name) { std::lock_guard<std::mutex> lock(mutex); return buffers[name]; }
};
struct PoolSlots { DeviceBuffer* slot[AGPU_PIECE_SLOTS]; DeviceBuffer& operator[](size_t k) { return *slot[k]; } };
}
struct agpu_ctx {
	explicit agpu_ctx(std::shared_ptr<agpu::ScratchPool> shared = std::shared_ptr<agpu::ScratchPool>()) : pool(shared ? shared : std::make_shared<agpu::ScratchPool>()),
		ingest_stream(pool->get("ingest.stream")), coverage_windows32(pool->get("ingest.coverage_windows32")) {
		static const char* const raw[AGPU_PIECE_SLOTS] = { "ingest.raw0", "ingest.raw1", "ingest.raw2", "ingest.raw3" }; static const char* const blocks[AGPU_PIECE_SLOTS] = { "ingest.blocks0", "ingest.blocks1", "ingest.blocks2", "ingest.blocks3" };
		for (int k = 0; k < AGPU_PIECE_SLOTS; ++k) { ingest_raw.slot[k] = &pool->get(raw[k]); ingest_blocks.slot[k] = &pool->get(blocks[k]); }
	}
	agpu_ctx(const agpu_ctx&) = delete; agpu_ctx& operator=(const agpu_ctx&) = delete;
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t event_start = nullptr, event_stop = nullptr;
	float last_ms = 0;
	uint64_t last_bytes = 0;
	agpu_params params;
	bool profiling = false;
	// launches are timed from the thread that runs the stages and, in a session with two lanes, from the thread that feeds this context: profile_mutex guards the three vectors;
	// agpu_set_profiling starts a new epoch, samples of an older one are dropped when their events have completed
	std::mutex profile_mutex;
	unsigned int profile_epoch = 0;
	std::vector<agpu::KernelSample> samples_pending, samples_done;
	std::vector<hipEvent_t> event_pool;
	// scratch buffers of the stage calls, kept between calls (grow-only) and addressed by name
	std::shared_ptr<agpu::ScratchPool> pool;
	agpu::DeviceBuffer& scratch(const char* name) { return pool->get(name); }
	bool keeps_batch_buffers = false; // agpu_keep_batch_buffers: the lanes of a session do not hand the batch over (agpu_api.hip: take_sample_buffers)
	agpu_ctx* sibling = nullptr; // agpu_create_sibling: the other lane of a session (shares `pool`; the buffers of a sample change hands in agpu_ingest_finish: take_sample_buffers)

	// annotation
	uint32_t n_genes = 0, n_exons = 0, n_dummy = 0;
	agpu::DeviceBuffer gene_contig, gene_start, gene_end, gene_bits, gene_exonic_length;
	agpu::DeviceBuffer exon_start, exon_end, exon_gene, exon_previous, exon_next, exon_cds_start, exon_cds_end;
	agpu::DeviceBuffer exon_index_contig_offset, exon_index_keys, exon_index_member_offset, exon_index_members, exon_index_bins;
	agpu::DeviceBuffer gene_index_contig_offset, gene_index_keys, gene_index_member_offset, gene_index_members, gene_index_bins;
	agpu::DeviceBuffer dummy_start_key, dummy_end_key;
	agpu::AnnotationView annotation;
	bool have_annotation = false;

	// genome
	agpu::DeviceBuffer genome_contig_offset, genome_contig_bits, genome_bases;
	agpu::GenomeView genome;
	std::vector<uint8_t> host_contig_bits;
	std::vector<uint64_t> host_contig_offset;
	bool have_genome = false;

	// batch
	uint64_t n = 0;
	agpu::DeviceBuffer n_aln, fbits, filter, group;
	agpu::DeviceBuffer pristine_fbits, pristine_abits[3]; // state as uploaded, restored by agpu_reset
	agpu::DeviceBuffer contig[3], start[3], end[3], abits[3], cigar_offset[3], cigar_count[3], cigar_pool;
	agpu::DeviceBuffer seq_offset[2], seq_length[2], seq_pool;
	agpu::DeviceBuffer gene_count[3], genes[3], gene_pool, counters;
	std::string failed_launch; // the first kernel launch of the sample that the runtime refused (KernelTimer), or empty
	agpu::BatchView batch;
	uint64_t batch_input_bytes = 0;
	uint32_t max_read_length = 0;
	bool have_batch = false, annotated = false, stage1_done = false, stage2_done = false, annotate_begun = false;
	uint32_t n_unmapped = 0;
	// the read lists of the candidates this rank built, kept when the replicated table is imported (agpu_import_candidates)
	agpu::DeviceBuffer owned_list_offset, owned_read_lists, owned_global_index;
	uint32_t n_owned = 0; uint64_t n_owned_list_entries = 0;
	bool candidates_imported = false, owned_index_set = false, multimappers_begun = false;
	uint64_t n_multimappers_global = 0;
	// coverage_t (agpu_upload_coverage)
	agpu::DeviceBuffer coverage_window_offset, coverage_windows, coverage_fragment_starts, coverage_fragment_ends;
	agpu::CoverageView coverage = { 0, nullptr, nullptr, nullptr, nullptr };
	bool have_coverage = false;
	// read_chimeric_alignments on the device (agpu_ingest.hip): the uncompressed BAM stream while it is being pushed, and what stays behind the pack
	agpu::DeviceBuffer& ingest_stream; agpu::PoolSlots ingest_raw, ingest_blocks; agpu::DeviceBuffer& coverage_windows32; // (in the pool: shared by the lanes of a session)
	agpu::DeviceBuffer ingest_tid_to_contig, ingest_viral_counts;
	agpu::DeviceBuffer names, name_offset; // "QNAME,HI" of every fragment of a batch built on the device
	uint64_t ingest_stream_size = 0, ingest_first_record = 0, names_size = 0;
	uint32_t ingest_n_targets = 0, ingest_max_itd_length = 100, ingest_pushes = 0, ingest_host_buffers = 2;
	uint8_t ingest_external_duplicate_marking = 0;
	bool ingest_active = false, ingest_finishing = false /* inside agpu_ingest_finish: its tables are in use although the feed is over */, batch_from_ingest = false, ingest_part_of_sample = false, ingest_verify_crc = false, ingest_deflated_pieces = false /* a piece of this ingest went through bgzf_inflate_kernel */;
	agpu::DeviceBuffer ingest_qname_keys; uint64_t ingest_qname_runs = 0; // a part of a sample: 128-bit keys of the runs of read names in its stream
	agpu_ingest_result ingest_result; uint64_t ingest_pool_sizes[2] = { 0, 0 }; // what the last ingest (or merge of parts) reported; CIGAR words and sequence bytes of its pools
	agpu::IngestProgress ingest_progress;
	// A pushed piece: copied on the context's stream (piece_copied: the caller's buffer is free), unwrapped and CRC-checked on a stream of its own (piece_stream; piece_ready: its
	// bytes are in the stream, piece_done: the raw bytes are not needed any more), so that the copy of the next piece never waits for a kernel; AGPU_PIECE_SLOTS raw buffers in turn
	hipStream_t piece_stream = nullptr, piece_stream2 = nullptr /* deflated pieces take the two in turn */; hipEvent_t piece_copied[AGPU_PIECE_SLOTS] = {}, piece_ready[AGPU_PIECE_SLOTS] = {}, piece_done[AGPU_PIECE_SLOTS] = {};
	std::vector<uint64_t> host_coverage_window_offset;
	agpu::DeviceBuffer gather_ids, gather_cigar_base, gather_seq_base, gather_name_base; // agpu_gather_rows_begin -> _copy
	uint64_t gather_n = 0, gather_sizes[3] = { 0, 0, 0 };
	bool gather_all = false;
	// closest genomic breakpoints per candidate (agpu_mark_genomic_support); not marked: every candidate -1
	agpu::DeviceBuffer cand_closest1, cand_closest2;
	bool genomic_support_marked = false;
	uint32_t n_selected = 0, selected_of_candidates = 0xFFFFFFFFu; // agpu_select_candidates: how many, and of which table (scratch "select.ids" holds them)
	uint32_t confidence_candidates = 0xFFFFFFFFu; // n_candidates of the last agpu_assign_confidence (its result stays in scratch "events.confidence")
	agpu::GenomicSupport genomic_support() { agpu::GenomicSupport wgs = { genomic_support_marked ? cand_closest1.as<int32_t>() : nullptr, genomic_support_marked ? cand_closest2.as<int32_t>() : nullptr }; return wgs; }
	uint64_t annotation_serial = 1, gene_read_counts_of_annotation = 0; // every annotate of a batch takes a new serial; scratch "events.gene_read_count" holds the counts of that one
	std::vector<uint32_t> host_gene_read_counts;
	uint64_t global_n = 0; // fragments of the whole sample when this context holds one shard of it (agpu_set_shard); 0 = not sharded
	// One sample over the GPUs of a node with the reads sharded (agpu_shard_keep, include/arriba_gpu.h): the candidate table and every read list are here (global name ranks), the reads
	// of the other ranks are not; what a walk over read lists asks of a read is replicated, one byte per fragment of the SAMPLE (agpu_read_state_import): scratch "sharded.filter"
	// (the filter ids) and "sharded.bits" (WALK_MULTIMAPPER | WALK_EXONIC) -- agpu_api.hip: candidate_walk_batch
	bool read_sharded = false, state_imported = false, sample_gene_read_counts_set = false;
	uint32_t n_clipped_entries = 0; // agpu_in_vitro_clipped_mates -> agpu_copy_in_vitro_clipped_mates (scratch "sharded.clipped_entries")

	// scratch
	agpu::DeviceBuffer unmapped_keys, sort_scratch, sorted_keys, scan_flags, scan_ids;
	agpu::DeviceBuffer viral_pairs;
	uint64_t viral_pair_capacity = 0;
	agpu::DeviceBuffer duplicate_keys, duplicate_slots, duplicate_entries;
	uint32_t n_duplicate_entries = 0;
	agpu::DeviceBuffer sample_flags, sample_values, samples;
	agpu::DeviceBuffer stage_counts;

	// find_fusions
	agpu::DeviceBuffer emissions, discordant_swapped;
	agpu::DeviceBuffer cand_gene1, cand_gene2, cand_contigs, cand_breakpoint1, cand_breakpoint2, cand_flags, cand_filter, cand_split_reads1, cand_split_reads2, cand_discordant_mates;
	agpu::DeviceBuffer cand_anchor1, cand_anchor2, cand_list_offset, cand_read_lists, cand_evalue, cand_iteration_rank, cand_votes, cand_first_occurrence, cand_extra_split_list;
	agpu::DeviceBuffer evalue_support_scale, evalue_intragenic_support, evalue_intergenic_support, evalue_distance_tables;
	agpu::EvalueGlobals evalue_globals;
	bool evalue_done = false, iteration_order_done = false;

	// k-mer index + splice sites (filter_mismappers)
	agpu::DeviceBuffer kmer_contig_table, kmer_offsets, kmer_positions, splice_offset, splice_sites, splice_bits;
	uint32_t kmer_positions_count = 0, splice_sites_for_dummy = 0, mismapper_jobs = 0, mismapper_heavy = 0, mismapper_leftover = 0;
	bool kmer_index_done = false, have_splice_sites = false, mismapper_jobs_ready = false;
	agpu::CandidateTable candidates;
	uint32_t n_emissions = 0, n_candidates = 0, n_queued_buckets = 0, n_discordant_emissions = 0; uint64_t n_list_entries = 0;
	// implicit discordant-mate lists (fusion_core.hpp: CandidateTable::discordant_before): where the windows of candidates are cut, what the expansion of a list needs again
	bool lists_implicit = false; std::vector<uint32_t> list_window_cuts; uint64_t list_window_entries = 0; int32_t lists_max_mate_gap = 0; uint32_t lists_n_bucket_rows = 0;
	bool fusions_done = false;

	// tables
	agpu::DeviceBuffer mismatch_verdict, kmer_threshold, filter_enabled, viral_verdict_top, viral_verdict_low;
	agpu::FilterTables tables;
	uint64_t genome_size = 0;
};

namespace agpu {

// agpu_ingest.hip: the stream and the per-record tables of the last ingest given back to the device (they are kept for the next sample as long as memory allows); true if there were any
bool release_ingest_buffers(agpu_ctx* ctx);
// agpu_api.hip: the buffers that hold a sample (batch, gene sets, candidates, read lists, k-mer index, ...) change hands between the lanes of a session: `ctx`, about to build
// its batch, takes what its sibling -- whose sample is done on the device -- holds wherever that is the larger buffer; the sibling's sample is gone afterwards
void take_sample_buffers(agpu_ctx* ctx, bool batch_group = true, bool stage_group = true);
// agpu_api.hip: what follows the columns of a batch, whoever filled them (agpu_upload_batch, or the ingest on the device: agpu_ingest.hip)
int finish_batch_setup(agpu_ctx* ctx);
// agpu_api.hip: the batch a stage sees that judges CANDIDATES by their read lists.  One context with all reads: its own batch (with walk bytes made now, if asked for).  The reads
// sharded over the ranks: a view of the global_n fragments of the sample that holds the replicated filters (and the walk bytes made from them) and NOTHING else -- a stage that
// wants more of a read runs where the read is.  pull_filters_of_own_reads: behind a stage that changed filters of reads through that view (the same on every rank).
int candidate_walk_batch(agpu_ctx* ctx, BatchView& batch, bool with_walk_bytes);
int pull_filters_of_own_reads(agpu_ctx* ctx);

// Brackets one kernel launch (or library call) with HIP events when profiling is on.  Usage:
//   { KernelTimer timer(ctx, "stage2_kernel", bytes); stage2_kernel<<<...>>>(...); }
struct KernelTimer {
	agpu_ctx* ctx; bool armed;
	hipStream_t stream;
	KernelSample sample;
	const char* launched;
	KernelTimer(agpu_ctx* c, const char* name, uint64_t bytes, hipStream_t on = nullptr) : ctx(c), armed(false), stream(on ? on : c->stream), launched(name) {
		if (!ctx->profiling) return;
		sample.name = name; sample.bytes = bytes; sample.ms = 0; sample.start = nullptr; sample.stop = nullptr;
		{ std::lock_guard<std::mutex> lock(ctx->profile_mutex);
		  sample.epoch = ctx->profile_epoch;
		  hipEvent_t* events[2] = { &sample.start, &sample.stop };
		  for (int k = 0; k < 2; ++k) if (!ctx->event_pool.empty()) { *events[k] = ctx->event_pool.back(); ctx->event_pool.pop_back(); } }
		if ((sample.start == nullptr && hipEventCreate(&sample.start) != hipSuccess) || (sample.stop == nullptr && hipEventCreate(&sample.stop) != hipSuccess)) return;
		(void) hipEventRecord(sample.start, stream);
		armed = true;
	}
	~KernelTimer() {
		// a launch that the runtime refused (a grid of more than 2^32 work-items, too much LDS) says so only here: noted with the context, and the sample fails where its results are
		// picked (agpu_select_candidates) instead of going on with whatever the kernel's output buffer held
		const hipError_t refused = hipGetLastError();
		if (refused != hipSuccess) { std::lock_guard<std::mutex> lock(ctx->profile_mutex); if (ctx->failed_launch.empty()) ctx->failed_launch = std::string(launched) + ": " + hipGetErrorString(refused); }
		if (!armed) return;
		(void) hipEventRecord(sample.stop, stream);
		std::lock_guard<std::mutex> lock(ctx->profile_mutex);
		ctx->samples_pending.push_back(sample);
	}
};
// the stages that walk the read lists of the candidates (agpu_fusions.hip: implicit discordant lists)
// whose lists a stage reads: an expansion of implicit lists skips the candidates the stage does not look at (their part of the window holds nothing, or zeros with zero_fill --
// for a stage that strides over the entries of a window without asking whose they are)
enum { LISTS_OF_ALL = 0, LISTS_OF_UNFILTERED = 1, LISTS_OF_IN_VITRO = 2 /* in_vitro_looks_at */, LISTS_OF_BOTH_SPLICED = 3 /* both_spliced_is_member */ };
int for_each_list_window(agpu_ctx* ctx, const std::function<int(const CandidateTable&, uint32_t, uint32_t)>& stage, int lists_of = LISTS_OF_ALL, bool zero_fill = false);
int recut_list_windows(agpu_ctx* ctx);
// resolve the pending samples whose events have completed (the caller has synchronised the streams it launched on; what another thread launched meanwhile stays pending)
inline void collect_kernel_samples(agpu_ctx* ctx) {
	std::lock_guard<std::mutex> lock(ctx->profile_mutex);
	size_t kept = 0;
	for (size_t k = 0; k < ctx->samples_pending.size(); ++k) {
		KernelSample sample = ctx->samples_pending[k];
		if (hipEventQuery(sample.stop) == hipErrorNotReady) { ctx->samples_pending[kept++] = sample; continue; }
		if (hipEventElapsedTime(&sample.ms, sample.start, sample.stop) != hipSuccess) sample.ms = 0;
		ctx->event_pool.push_back(sample.start); ctx->event_pool.push_back(sample.stop);
		if (sample.epoch == ctx->profile_epoch) ctx->samples_done.push_back(sample);
	}
	ctx->samples_pending.resize(kept);
}

}

#endif
