// arriba_amd/csrc/device/agpu_api.hip -- HIP kernels (gfx950) and the C ABI of include/arriba_gpu.h.
//
// Kernel inventory (all HBM-bandwidth / latency bound integer work, one thread per chimeric fragment,
// columns read with coalesced loads, no MFMA):
//   mark_multimappers_kernel   neighbour compare of the name-group column
//   annotate_stage1_kernel     strands, exon/gene interval lookups, splice-site disambiguation, unmapped positions
//   dummy_*_kernel             segmentation of the sorted unmapped positions into intergenic dummy genes
//   annotate_stage2_kernel     dummy-gene assignment, viral integration-site pairs
//   duplicate_*_kernel         duplicate keys, lock-free hash insert (min name rank per key), resolve
//   stage1_kernel              duplicates + contig filters
//   sample_*_kernel            ordered mate-gap samples for the fragment-length estimate
//   stage2_kernel              read_through ... low_entropy cascade (k-mer counters in LDS)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "device_utils.hpp"
#include "index_bins.hpp"

using namespace agpu;

namespace agpu {
static thread_local std::string g_last_error, g_allocation_note;
void set_last_error(const std::string& message) { g_last_error = message; if (message.compare(0, 16, "hipMalloc failed") == 0 && !g_allocation_note.empty()) { g_last_error += g_allocation_note; g_allocation_note.clear(); } }
// test hook: the first hipMalloc inside agpu_ingest_finish (on the thread that armed it) is treated as failed, so that the path behind a failure -- the idle buffers given
// back, the allocation tried again -- runs where a test can see what it does to an ingest that is finishing
static thread_local int g_fail_allocations_in_finish = 0;
thread_local bool g_inside_ingest_finish = false;
// test hook (agpu_debug_exhaust_memory_in_finish): the next calls of agpu_ingest_finish on this thread end as they do when the device is out of memory -- AGPU_ERR_NO_MEMORY --,
// n > 0: the next n of a context that has a sibling (two samples in flight on one device: what the session's retry is for); n < 0: every call (a device too small for the sample)
static thread_local int g_exhausted_finishes = 0;
bool debug_finish_runs_out_of_memory(agpu_ctx* ctx) {
	if (g_exhausted_finishes == 0) return false;
	if (g_exhausted_finishes < 0) return true;
	if (ctx->sibling == nullptr) return false;
	--g_exhausted_finishes;
	return true;
}
bool debug_allocation_fails() { if (g_fail_allocations_in_finish > 0 && g_inside_ingest_finish) { --g_fail_allocations_in_finish; return true; } return false; }
void note_failed_allocation(size_t bytes) {
	size_t free_bytes = 0, total_bytes = 0;
	(void) hipMemGetInfo(&free_bytes, &total_bytes);
	g_allocation_note = ": " + std::to_string(bytes >> 20) + " MB asked for, " + std::to_string(free_bytes >> 20) + " of " + std::to_string(total_bytes >> 20) + " MB free on the device";
}

// the live contexts of the process: an allocation that fails asks them to give back what they only keep for their next sample
static std::mutex g_contexts_mutex;
static std::vector<agpu_ctx*> g_contexts;
bool DeviceBuffer::release_idle_buffers() {
	std::lock_guard<std::mutex> lock(g_contexts_mutex);
	bool released = false;
	int device = -1;
	(void) hipGetDevice(&device);
	for (size_t k = 0; k < g_contexts.size(); ++k) {
		agpu_ctx* ctx = g_contexts[k];
		if (ctx->device != device) continue; // memory of another device does not help the allocation that failed (and its context may be at work on another thread)
		// the lanes of a session share their pool: while one feeds the other runs its stages, and nothing of the pool is idle.  A session of two lanes therefore does NOT get
		// memory back here: an allocation that fails in it fails the sample (message with what was asked for and what is free); the caller drains the queue
		// (arriba_workflow_cancel) and runs the sample alone, where this function does help ("Memory").  The memo tables and task lists of filter_mismappers,
		// the largest reservation, fall back to fewer workgroups by themselves (agpu_mismappers.hip)
		if (ctx->pool.use_count() > 1) continue;
		(void) hipStreamSynchronize(ctx->stream);
		if (!ctx->ingest_active && !ctx->ingest_finishing) { if (release_ingest_buffers(ctx)) released = true; }
		else { // an ingest runs: nothing of the stages of the sample before is needed any more
			std::lock_guard<std::mutex> pool_lock(ctx->pool->mutex);
			for (std::map<std::string, DeviceBuffer>::iterator buffer = ctx->pool->buffers.begin(); buffer != ctx->pool->buffers.end(); ++buffer)
				if (buffer->first.compare(0, 7, "ingest.") != 0 && buffer->second.ptr != nullptr) { buffer->second.release(); released = true; }
		}
	}
	return released;
}
// batch_group: the columns, pools and names of the batch (what agpu_ingest_finish writes); stage_group: everything the stages behind it fill.  A session whose lanes keep their
// batch buffers (agpu_keep_batch_buffers: the ingest of the next sample is FINISHED beside the stages of the current one) takes nothing at the finish and the stage group when
// its stages begin (agpu_mark_multimappers), when the sibling's sample is done on the device.
void take_sample_buffers(agpu_ctx* ctx, bool batch_group, bool stage_group) {
	agpu_ctx* from = ctx->sibling;
	if (from == nullptr || (!batch_group && !stage_group)) return;
	if (batch_group) (void) hipStreamSynchronize(from->stream); // (its last copies back to the host have left the buffers; the stage group is taken by the thread that ran the sibling's stages to their end)
	DeviceBuffer* mine[] = { &ctx->n_aln, &ctx->fbits, &ctx->filter, &ctx->group, &ctx->pristine_fbits, &ctx->pristine_abits[0], &ctx->pristine_abits[1], &ctx->pristine_abits[2],
		&ctx->contig[0], &ctx->contig[1], &ctx->contig[2], &ctx->start[0], &ctx->start[1], &ctx->start[2], &ctx->end[0], &ctx->end[1], &ctx->end[2], &ctx->abits[0], &ctx->abits[1], &ctx->abits[2],
		&ctx->cigar_offset[0], &ctx->cigar_offset[1], &ctx->cigar_offset[2], &ctx->cigar_count[0], &ctx->cigar_count[1], &ctx->cigar_count[2], &ctx->cigar_pool, &ctx->seq_offset[0], &ctx->seq_offset[1],
		&ctx->seq_length[0], &ctx->seq_length[1], &ctx->seq_pool, &ctx->gene_count[0], &ctx->gene_count[1], &ctx->gene_count[2], &ctx->genes[0], &ctx->genes[1], &ctx->genes[2], &ctx->gene_pool,
		&ctx->names, &ctx->name_offset, &ctx->ingest_qname_keys, &ctx->gather_ids, &ctx->gather_cigar_base, &ctx->gather_seq_base, &ctx->gather_name_base, &ctx->unmapped_keys, &ctx->viral_pairs,
		/* (up to here: what agpu_ingest_finish and finish_batch_setup fill -- the batch group of take_sample_buffers; from here on: the buffers of the stages) */ &ctx->cand_closest1, &ctx->cand_closest2,
		&ctx->sort_scratch, &ctx->sorted_keys, &ctx->scan_flags, &ctx->scan_ids, &ctx->duplicate_keys, &ctx->duplicate_slots, &ctx->duplicate_entries,
		&ctx->sample_flags, &ctx->sample_values, &ctx->samples, &ctx->emissions, &ctx->discordant_swapped,
		&ctx->cand_gene1, &ctx->cand_gene2, &ctx->cand_contigs, &ctx->cand_breakpoint1, &ctx->cand_breakpoint2, &ctx->cand_flags, &ctx->cand_filter, &ctx->cand_split_reads1, &ctx->cand_split_reads2, &ctx->cand_discordant_mates,
		&ctx->cand_anchor1, &ctx->cand_anchor2, &ctx->cand_list_offset, &ctx->cand_read_lists, &ctx->cand_evalue, &ctx->cand_iteration_rank, &ctx->cand_votes, &ctx->cand_first_occurrence, &ctx->cand_extra_split_list,
		&ctx->kmer_contig_table, &ctx->kmer_offsets, &ctx->kmer_positions, &ctx->splice_offset, &ctx->splice_sites, &ctx->splice_bits };
	DeviceBuffer* theirs[] = { &from->n_aln, &from->fbits, &from->filter, &from->group, &from->pristine_fbits, &from->pristine_abits[0], &from->pristine_abits[1], &from->pristine_abits[2],
		&from->contig[0], &from->contig[1], &from->contig[2], &from->start[0], &from->start[1], &from->start[2], &from->end[0], &from->end[1], &from->end[2], &from->abits[0], &from->abits[1], &from->abits[2],
		&from->cigar_offset[0], &from->cigar_offset[1], &from->cigar_offset[2], &from->cigar_count[0], &from->cigar_count[1], &from->cigar_count[2], &from->cigar_pool, &from->seq_offset[0], &from->seq_offset[1],
		&from->seq_length[0], &from->seq_length[1], &from->seq_pool, &from->gene_count[0], &from->gene_count[1], &from->gene_count[2], &from->genes[0], &from->genes[1], &from->genes[2], &from->gene_pool,
		&from->names, &from->name_offset, &from->ingest_qname_keys, &from->gather_ids, &from->gather_cigar_base, &from->gather_seq_base, &from->gather_name_base, &from->unmapped_keys, &from->viral_pairs,
		/* (up to here: what agpu_ingest_finish and finish_batch_setup fill -- the batch group of take_sample_buffers; from here on: the buffers of the stages) */ &from->cand_closest1, &from->cand_closest2,
		&from->sort_scratch, &from->sorted_keys, &from->scan_flags, &from->scan_ids, &from->duplicate_keys, &from->duplicate_slots, &from->duplicate_entries,
		&from->sample_flags, &from->sample_values, &from->samples, &from->emissions, &from->discordant_swapped,
		&from->cand_gene1, &from->cand_gene2, &from->cand_contigs, &from->cand_breakpoint1, &from->cand_breakpoint2, &from->cand_flags, &from->cand_filter, &from->cand_split_reads1, &from->cand_split_reads2, &from->cand_discordant_mates,
		&from->cand_anchor1, &from->cand_anchor2, &from->cand_list_offset, &from->cand_read_lists, &from->cand_evalue, &from->cand_iteration_rank, &from->cand_votes, &from->cand_first_occurrence, &from->cand_extra_split_list,
		&from->kmer_contig_table, &from->kmer_offsets, &from->kmer_positions, &from->splice_offset, &from->splice_sites, &from->splice_bits };
	static_assert(sizeof(mine) == sizeof(theirs), "the same buffers of both contexts");
	bool any = false;
	size_t first_of_stage_group = 0;
	while (mine[first_of_stage_group] != &ctx->cand_closest1) ++first_of_stage_group;
	for (size_t k = batch_group ? 0 : first_of_stage_group; k < (stage_group ? sizeof(mine) / sizeof(mine[0]) : first_of_stage_group); ++k) if (theirs[k]->capacity > mine[k]->capacity) { mine[k]->swap(*theirs[k]); any = true; }
	if (!any) return;
	// the views of the sibling point at what it gave away: it has no sample until its next ingest (the batch flags only when the batch went: with the stage group alone the sibling
	// may be finishing its next ingest on another thread at this very moment, and those flags are that thread's)
	if (batch_group) { from->have_batch = false; from->annotated = false; from->stage1_done = false; from->stage2_done = false; from->n = 0; }
	from->fusions_done = false; from->evalue_done = false; from->iteration_order_done = false;
	from->kmer_index_done = false; from->have_splice_sites = false; from->genomic_support_marked = false; from->mismapper_jobs_ready = false; from->n_candidates = 0;
	from->candidates = agpu::CandidateTable();
	if (batch_group) { from->viral_pair_capacity = 0; ctx->viral_pair_capacity = 0; } // (set again by finish_batch_setup)
	// ... and what this context kept of its own last sample went with the buffers
	ctx->have_splice_sites = false; ctx->kmer_index_done = false;
}

}

namespace {

const int BLOCK = 256;
enum { COUNTER_GENE_POOL = 0, COUNTER_UNMAPPED = 1, COUNTER_ERROR = 2, COUNTER_VIRAL_PAIRS = 3, COUNTER_MARKED = 4, COUNTER_SAMPLES = 5, COUNTER_VISITED_LO = 6, COUNTER_VISITED_HI = 7, COUNTER_COUNT = 16 };
enum { ERROR_GENE_SET_OVERFLOW = 1, ERROR_VIRAL_PAIR_OVERFLOW = 2 };
const uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
const uint32_t MAX_SAMPLES = 100001;

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)

inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

// ---- kernels ------------------------------------------------------------------------------------

__global__ void mark_multimappers_kernel(BatchView b, uint32_t* counters) {
	__shared__ uint32_t marked;
	uint32_t mine = 0;
	for (uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; i < b.n; i += gridDim.x * (uint64_t) BLOCK) {
		uint32_t group = b.group[i];
		bool same_as_previous = i > 0 && b.group[i - 1] == group;
		bool same_as_next = i + 1 < b.n && b.group[i + 1] == group;
		if (same_as_previous || same_as_next) b.fbits[i] |= FBIT_MULTIMAPPER;
		if (same_as_next) ++mine;
	}
	block_tally(mine, &counters[COUNTER_MARKED], &marked);
}

__global__ void __launch_bounds__(BLOCK) annotate_stage1_kernel(BatchView b, AnnotationView ann, uint32_t strandedness, uint64_t* unmapped_keys, uint32_t* counters) {
	__shared__ uint32_t wave_offset[BLOCK / 64];
	__shared__ uint32_t block_base;
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	uint64_t unmapped[2];
	uint32_t n_unmapped = 0;
	if (i < b.n && !annotate_fragment_stage1(b, ann, strandedness, i, unmapped, n_unmapped))
		atomicOr(&counters[COUNTER_ERROR], (uint32_t) ERROR_GENE_SET_OVERFLOW);
	const uint32_t at = block_append<BLOCK>(n_unmapped, &counters[COUNTER_UNMAPPED], wave_offset, &block_base); // one atomic per workgroup
	if (n_unmapped >= 1) unmapped_keys[at] = unmapped[0];
	if (n_unmapped >= 2) unmapped_keys[at + 1] = unmapped[1];
}

__global__ void dummy_flags_kernel(const uint64_t* sorted_keys, uint32_t n, FlatIndexView gene_index, uint32_t* flags) {
	uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	flags[i] = dummy_gene_starts_here(sorted_keys, i, gene_index) ? 1u : 0u;
}

// ids = inclusive scan of flags; dummy gene j = ids[i]-1 covers the run of positions with that id
__global__ void dummy_write_kernel(const uint64_t* sorted_keys, uint32_t n, const uint32_t* flags, const uint32_t* ids, uint64_t* dummy_start_key, uint64_t* dummy_end_key) {
	uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	uint32_t j = ids[i] - 1;
	if (flags[i]) dummy_start_key[j] = sorted_keys[i];
	if (i + 1 == n || flags[i + 1]) dummy_end_key[j] = sorted_keys[i];
}

__global__ void dummy_gene_table_kernel(uint32_t n_genes, uint32_t n_dummy, const uint64_t* dummy_start_key, const uint64_t* dummy_end_key,
                                        uint16_t* gene_contig, int32_t* gene_start, int32_t* gene_end, uint8_t* gene_bits, int32_t* gene_exonic_length) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= n_dummy) return;
	uint32_t g = n_genes + j;
	gene_contig[g] = (uint16_t) (dummy_start_key[j] >> 32);
	gene_start[g] = (int32_t) (uint32_t) dummy_start_key[j];
	gene_end[g] = (int32_t) (uint32_t) dummy_end_key[j];
	gene_bits[g] = GBIT_STRAND | GBIT_DUMMY;  // strand FORWARD, not protein coding (source/arriba.cpp:238-241)
	gene_exonic_length[g] = 10000;
}

__global__ void __launch_bounds__(BLOCK) annotate_stage2_kernel(BatchView b, AnnotationView ann, GenomeView genome, uint32_t* viral_pairs, uint32_t viral_pair_capacity, uint32_t* counters) {
	__shared__ uint32_t wave_offset[BLOCK / 64];
	__shared__ uint32_t block_base;
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	AGPU_IDSET(genes); genes.clear();
	uint32_t viral_contig = 0;
	if (i < b.n) {
		if (!annotate_fragment_stage2(b, ann, i))
			atomicOr(&counters[COUNTER_ERROR], (uint32_t) ERROR_GENE_SET_OVERFLOW);
		// virus-host chimeric fragments: host genes per viral contig (source/filter_top_expressed_viral_contigs.cpp:95-112)
		int mate2 = (b.n_aln[i] == 3) ? SUPPLEMENTARY : MATE2;
		int viral_slot = -1, host_slot = -1;
		uint8_t bits1 = genome.contig_bits[b.contig[MATE1][i]], bits2 = genome.contig_bits[b.contig[mate2][i]];
		if (bits1 & CBIT_VIRAL) viral_slot = MATE1; else if (bits1 & CBIT_INTERESTING) host_slot = MATE1;
		if (bits2 & CBIT_VIRAL) viral_slot = mate2; else if (bits2 & CBIT_INTERESTING) host_slot = mate2;
		if (viral_slot >= 0 && host_slot >= 0) { load_genes(b, host_slot, i, genes); viral_contig = b.contig[viral_slot][i]; }
	}
	const uint32_t at = block_reserve<BLOCK>(genes.n, &counters[COUNTER_VIRAL_PAIRS], wave_offset, &block_base); // one atomic per workgroup
	if (genes.n == 0) return;
	if (at + genes.n > viral_pair_capacity) { atomicOr(&counters[COUNTER_ERROR], (uint32_t) ERROR_VIRAL_PAIR_OVERFLOW); return; }
	for (uint32_t g = 0; g < genes.n; ++g) {
		viral_pairs[2 * (at + g)] = viral_contig;
		viral_pairs[2 * (at + g) + 1] = genes.get(g);
	}
}

__global__ void duplicate_keys_kernel(BatchView b, DuplicateKey* keys) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	keys[i] = duplicate_key(b, i);
}

// lock-free open addressing: every slot holds the smallest name rank seen for its key
__global__ void duplicate_insert_kernel(uint64_t n, const DuplicateKey* keys, uint32_t* slots, uint32_t mask) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	DuplicateKey key = keys[i];
	uint32_t h = (uint32_t) hash_duplicate_key(key) & mask;
	while (true) {
		uint32_t owner = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (owner == EMPTY_SLOT) {
			owner = atomicCAS(&slots[h], EMPTY_SLOT, (uint32_t) i);
			if (owner == EMPTY_SLOT) return;
		}
		if (keys_equal(keys[owner], key)) { if ((uint32_t) i < owner) atomicMin(&slots[h], (uint32_t) i); return; } // the slot only ever decreases
		h = (h + 1) & mask;
	}
}

// sharded samples: the local winners (key, global name rank) are exchanged between the shards; DuplicateEntry is the wire format
struct DuplicateEntry { DuplicateKey key; uint32_t rank; };
__global__ void duplicate_winner_flag_kernel(uint64_t n, const DuplicateKey* keys, const uint32_t* slots, uint32_t mask, uint8_t* flags) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	DuplicateKey key = keys[i];
	uint32_t h = (uint32_t) hash_duplicate_key(key) & mask;
	while (true) {
		uint32_t owner = slots[h];
		if (keys_equal(keys[owner], key)) { flags[i] = owner == (uint32_t) i; return; }
		h = (h + 1) & mask;
	}
}
__global__ void duplicate_entry_write_kernel(uint32_t n_winners, const uint32_t* winners, const DuplicateKey* keys, uint64_t first_rank, DuplicateEntry* entries) {
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k >= n_winners) return;
	DuplicateEntry entry; entry.key = keys[winners[k]]; entry.rank = (uint32_t) (first_rank + winners[k]);
	entries[k] = entry;
}
// table over the entries of all shards (ascending global rank): a slot holds the smallest entry index of its key
__global__ void duplicate_entry_insert_kernel(uint64_t n_entries, const DuplicateEntry* entries, uint32_t* slots, uint32_t mask) {
	uint64_t e = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (e >= n_entries) return;
	DuplicateKey key = entries[e].key;
	uint32_t h = (uint32_t) hash_duplicate_key(key) & mask;
	while (true) {
		uint32_t owner = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (owner == EMPTY_SLOT) {
			owner = atomicCAS(&slots[h], EMPTY_SLOT, (uint32_t) e);
			if (owner == EMPTY_SLOT) return;
		}
		if (keys_equal(entries[owner].key, key)) { if ((uint32_t) e < owner) atomicMin(&slots[h], (uint32_t) e); return; }
		h = (h + 1) & mask;
	}
}
__global__ void duplicate_global_verdict_kernel(BatchView b, const DuplicateKey* keys, const DuplicateEntry* entries, const uint32_t* slots, uint32_t mask, uint8_t* is_duplicate) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	DuplicateKey key = keys[i];
	uint32_t h = (uint32_t) hash_duplicate_key(key) & mask;
	while (true) {
		uint32_t owner = slots[h];
		if (keys_equal(entries[owner].key, key)) { is_duplicate[i] = entries[owner].rank != (uint32_t) (b.first_rank + i); return; }
		h = (h + 1) & mask;
	}
}

__global__ void stage1_kernel(BatchView b, GenomeView genome, FilterTables t, const uint8_t* enabled, const DuplicateKey* keys, const uint32_t* slots, uint32_t mask, const uint8_t* global_duplicate, unsigned long long* stage_counts) {
	__shared__ unsigned int hits[5];
	if (threadIdx.x < 5) hits[threadIdx.x] = 0;
	__syncthreads();
	// capped grid with a grid-stride loop: the five counters of a workgroup are flushed once (one cache line of counters takes ~10 ns per atomic)
	for (uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; i < b.n; i += gridDim.x * (uint64_t) BLOCK) {
		uint8_t filter = b.filter[i];
		if (filter == FILTER_none && enabled[FILTER_duplicates]) {
			if (t.external_duplicate_marking) {
				if (b.fbits[i] & FBIT_DUPLICATE) filter = FILTER_duplicates;
			} else if (global_duplicate != nullptr) { // verdict from the table over all shards
				if (global_duplicate[i]) filter = FILTER_duplicates;
			} else {
				DuplicateKey key = keys[i];
				uint32_t h = (uint32_t) hash_duplicate_key(key) & mask;
				while (true) {
					uint32_t owner = slots[h];
					if (keys_equal(keys[owner], key)) { if (owner != (uint32_t) i) filter = FILTER_duplicates; break; }
					h = (h + 1) & mask;
				}
			}
			if (filter != FILTER_none) atomicAdd(&hits[0], 1u);
		}
		if (filter == FILTER_none) {
			uint8_t hit = contig_filters(b, genome, t, i);
			if (hit != FILTER_none && enabled[hit]) {
				filter = hit;
				atomicAdd(&hits[hit == FILTER_uninteresting_contigs ? 1 : hit == FILTER_viral_contigs ? 2 : hit == FILTER_top_expressed_viral_contigs ? 3 : 4], 1u);
			}
		}
		b.filter[i] = filter;
	}
	__syncthreads();
	if (threadIdx.x < 5 && hits[threadIdx.x]) atomicAdd(&stage_counts[threadIdx.x], (unsigned long long) hits[threadIdx.x]);
}

__global__ void sample_flags_kernel(BatchView b, AnnotationView ann, uint64_t first, uint64_t count, uint8_t* flags, int32_t* values) {
	uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k >= count) return;
	uint64_t i = first + k;
	bool sample = b.filter[i] == FILTER_none && !(b.fbits[i] & FBIT_SINGLE_END) && b.n_aln[i] == 3;
	flags[k] = sample;
	if (sample) values[k] = mate_gap_sample(b, ann, i);
}

// one workgroup appends the flagged values in order until MAX_SAMPLES are collected
__global__ void sample_compact_kernel(uint64_t first, uint64_t count, const uint8_t* flags, const int32_t* values, int32_t* samples, uint32_t* counters, uint32_t limit) {
	__shared__ uint32_t wave_totals[16];
	__shared__ uint32_t base;
	__shared__ uint32_t done;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
	if (threadIdx.x == 0) { base = counters[COUNTER_SAMPLES]; done = 0; }
	__syncthreads();
	for (uint64_t offset = 0; offset < count; offset += blockDim.x) {
		uint64_t k = offset + threadIdx.x;
		bool flag = k < count && flags[k];
		unsigned long long ballot = __ballot(flag);
		uint32_t before = __popcll(ballot & ((1ull << lane) - 1));
		if (lane == 0) wave_totals[wave] = __popcll(ballot);
		__syncthreads();
		uint32_t wave_offset = 0, total = 0;
		for (uint32_t w = 0; w < waves; ++w) { if (w < wave) wave_offset += wave_totals[w]; total += wave_totals[w]; }
		uint32_t position = base + wave_offset + before;
		if (flag && position < limit) {
			samples[position] = values[k];
			if (position == limit - 1) { // the reference stops right after this fragment
				uint64_t visited = first + k + 1;
				counters[COUNTER_VISITED_LO] = (uint32_t) visited; counters[COUNTER_VISITED_HI] = (uint32_t) (visited >> 32);
				done = 1;
			}
		}
		__syncthreads();
		if (threadIdx.x == 0) base += total;
		__syncthreads();
		if (done) break;
	}
	if (threadIdx.x == 0) counters[COUNTER_SAMPLES] = base < limit ? base : limit;
}

// ---- stream compaction between the filter stages ---------------------------------------------------------------------------
// Most fragments are already discarded when the expensive stages run (30 % PCR duplicates alone); a wavefront of which half the lanes
// return at once still pays for the whole walk of the others.  So the survivors are compacted first: every wavefront ballots its
// survivors, lane 0 reserves the output range with one atomic, the lanes write their fragment index at the prefix popcount.  The
// order inside the list is arbitrary (the stages do not depend on it).
enum { SELECT_UNFILTERED = 0, SELECT_LOW_ENTROPY_TEST = 1 };
const int SELECT_BLOCK = 1024; // one atomic per 1024 fragments: ~10 k atomics on the list cursor per launch at 10 M fragments
__global__ void __launch_bounds__(SELECT_BLOCK) select_fragments_kernel(BatchView b, FilterTables t, int what, uint32_t* selected, uint32_t* count) {
	__shared__ uint32_t wave_offset[SELECT_BLOCK / 64];
	__shared__ uint32_t block_base;
	const uint64_t i = blockIdx.x * (uint64_t) SELECT_BLOCK + threadIdx.x;
	bool keep = false;
	if (i < b.n) {
		const uint8_t filter = b.filter[i];
		keep = (what == SELECT_UNFILTERED) ? filter == FILTER_none : needs_low_entropy_test(b, t, i, filter);
	}
	const unsigned long long ballot = __ballot(keep);
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_offset[wave] = (uint32_t) __popcll(ballot);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < SELECT_BLOCK / 64; ++w) { uint32_t c = wave_offset[w]; wave_offset[w] = total; total += c; }
		block_base = total ? atomicAdd(count, total) : 0;
	}
	__syncthreads();
	if (keep) selected[block_base + wave_offset[wave] + __popcll(ballot & ((1ull << lane) - 1))] = (uint32_t) i;
}

// read_through ... mismatches: one thread per fragment that is still unfiltered
__global__ void __launch_bounds__(BLOCK) stage2_kernel(BatchView b, AnnotationView ann, GenomeView genome, FilterTables t, const uint8_t* enabled, const uint32_t* selected, const uint32_t* n_selected, unsigned long long* stage_counts) {
	__shared__ unsigned int hits[10];
	if (threadIdx.x < 10) hits[threadIdx.x] = 0;
	__syncthreads();
	// (one fragment per thread: a capped grid with a grid-stride loop was measured 10 % slower here, the walks are too uneven)
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k < *n_selected) {
		const uint64_t i = selected[k];
		uint32_t first_hit;
		uint8_t before = b.filter[i];
		uint8_t filter = read_filters_stage2(b, ann, genome, t, enabled, i, before, no_stage(), first_hit);
		if (filter != before) b.filter[i] = filter;
		if (first_hit < 9) atomicAdd(&hits[first_hit], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 10 && hits[threadIdx.x]) atomicAdd(&stage_counts[5 + threadIdx.x], (unsigned long long) hits[threadIdx.x]);
}

// low_entropy: the 3-mer counters of a thread are bit-sliced registers (filter_core.hpp), no LDS; one thread per fragment that needs the test
__global__ void __launch_bounds__(BLOCK) low_entropy_kernel(BatchView b, FilterTables t, const uint32_t* selected, const uint32_t* n_selected, unsigned long long* stage_counts) {
	__shared__ unsigned int hits;
	if (threadIdx.x == 0) hits = 0;
	__syncthreads();
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	uint32_t mine = 0;
	if (k < *n_selected) {
		const uint64_t i = selected[k];
		if (has_low_entropy(b, t, i, no_stage())) {
			if (b.filter[i] == FILTER_none) ++mine;
			b.filter[i] = FILTER_low_entropy;
		}
	}
	if (mine) atomicAdd(&hits, mine);
	__syncthreads();
	if (threadIdx.x == 0 && hits) atomicAdd(&stage_counts[13], (unsigned long long) hits);
}

// ---- host helpers ---------------------------------------------------------------------------------

template <class T> int upload(DeviceBuffer& buffer, const T* host, size_t count, hipStream_t stream) {
	if (!buffer.allocate(count * sizeof(T))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (count > 0) HIP_CHECK(hipMemcpyAsync(buffer.ptr, host, count * sizeof(T), hipMemcpyHostToDevice, stream));
	return AGPU_OK;
}
#define TRY(call) do { int s_ = (call); if (s_ != AGPU_OK) return s_; } while (0)

int upload_index(const agpu_flat_index& in, DeviceBuffer& contig_offset, DeviceBuffer& keys, DeviceBuffer& member_offset, DeviceBuffer& members, DeviceBuffer& bins, FlatIndexView& out, hipStream_t stream) {
	std::vector<uint32_t> host_bins;
	build_index_bins(in.n_contigs, in.contig_offset, in.keys, host_bins);
	TRY(upload(bins, host_bins.data(), host_bins.size(), stream));
	HIP_CHECK(hipStreamSynchronize(stream)); // the host vector goes out of scope
	out.bins = bins.as<uint32_t>();
	TRY(upload(contig_offset, in.contig_offset, (size_t) in.n_contigs + 1, stream));
	TRY(upload(keys, in.keys, in.n_keys, stream));
	TRY(upload(member_offset, in.member_offset, (size_t) in.n_keys + 1, stream));
	TRY(upload(members, in.members, in.n_members, stream));
	out.n_contigs = in.n_contigs;
	out.contig_offset = contig_offset.as<uint32_t>(); out.keys = keys.as<int32_t>();
	out.member_offset = member_offset.as<uint32_t>(); out.members = members.as<uint32_t>();
	return AGPU_OK;
}

// make room for new_bytes while keeping the first keep_bytes (the GTF genes); with headroom, so that the next pass over the same batch does not reallocate
int grow_preserving(DeviceBuffer& buffer, size_t keep_bytes, size_t new_bytes, hipStream_t stream) {
	if (new_bytes <= buffer.capacity) { buffer.bytes = new_bytes; return AGPU_OK; }
	DeviceBuffer larger;
	if (!larger.allocate(new_bytes + (new_bytes >> 2))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (keep_bytes > 0) HIP_CHECK(hipMemcpyAsync(larger.ptr, buffer.ptr, keep_bytes, hipMemcpyDeviceToDevice, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	buffer.swap(larger);
	buffer.bytes = new_bytes;
	return AGPU_OK;
}

void begin_timing(agpu_ctx* ctx) { (void) hipEventRecord(ctx->event_start, ctx->stream); }
int end_timing(agpu_ctx* ctx, uint64_t bytes) {
	HIP_CHECK(hipEventRecord(ctx->event_stop, ctx->stream));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	ctx->last_bytes = bytes;
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

int read_counters(agpu_ctx* ctx, uint32_t* host) {
	HIP_CHECK(hipMemcpyAsync(host, ctx->counters.ptr, COUNTER_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}

// the verdict of the mismatch filter as a function of (mismatches, aligned length); arithmetic exactly as in
// source/filter_mismatches.cpp:55-99 (float / double / long double mix, hazard H5), evaluated once on the host
double binomial_coefficient(const unsigned int k, const unsigned int n) {
	double result = 1;
	for (unsigned int i = n - k + 1; i <= n; ++i) result *= i;
	for (unsigned int i = 1; i <= k; ++i) result /= i;
	return result;
}
float binomial_distribution(const unsigned int k, const unsigned int n, const float p) {
	return binomial_coefficient(k, n) * pow(p, k) * pow(1 - p, n - k);
}
bool mismatch_verdict(unsigned int mismatches, unsigned int alignment_length, const float mismatch_probability, unsigned long int genome_size, const float pvalue_cutoff) {
	if (binomial_distribution(mismatches, alignment_length, mismatch_probability) < pvalue_cutoff) {
		return true;
	} else if (mismatches > 0) {
		long double number_of_permutations_of_bases = pow(4, alignment_length - mismatches);
		if (genome_size >= number_of_permutations_of_bases)
			return true;
		return (1 - pow(1 - genome_size / number_of_permutations_of_bases, binomial_coefficient(mismatches, alignment_length))) > 0.01;
	}
	return false;
}

int build_tables(agpu_ctx* ctx) {
	const uint32_t max_length = std::max<uint32_t>(ctx->max_read_length, 1) + 8;
	std::vector<uint32_t> verdict(((size_t) (max_length + 1) * (max_length + 2) / 2 + 31) / 32, 0);
	const float mismatch_probability = 0.01; // source/arriba.cpp:403
	for (uint32_t n = 0; n <= max_length; ++n)
		for (uint32_t k = 0; k <= n; ++k)
			if (mismatch_verdict(k, n, mismatch_probability, ctx->genome_size, ctx->params.mismatch_pvalue_cutoff)) {
				uint32_t bit = n * (n + 1) / 2 + k;
				verdict[bit >> 5] |= 1u << (bit & 31);
			}
	std::vector<uint32_t> threshold(max_length + 1);
	const unsigned int kmer_length = 3;
	const float kmer_content = ctx->params.max_kmer_content;
	for (uint32_t length = 0; length <= max_length; ++length) {
		unsigned int value = length * kmer_content / kmer_length + 0.5; // source/filter_low_entropy.cpp:67-69
		threshold[length] = value;
	}
	TRY(upload(ctx->mismatch_verdict, verdict.data(), verdict.size(), ctx->stream));
	TRY(upload(ctx->kmer_threshold, threshold.data(), threshold.size(), ctx->stream));
	TRY(upload(ctx->filter_enabled, ctx->params.filter_enabled, AGPU_FILTER_COUNT, ctx->stream));
	FilterTables& t = ctx->tables;
	t.mismatch_verdict = ctx->mismatch_verdict.as<uint32_t>(); t.mismatch_max_length = max_length;
	t.kmer_threshold = ctx->kmer_threshold.as<uint32_t>(); t.kmer_threshold_size = max_length + 1;
	t.max_kmer_content = kmer_content;
	t.homopolymer_length = ctx->params.homopolymer_length;
	t.min_read_through_distance = (int32_t) ctx->params.min_read_through_distance;
	t.max_itd_length = ctx->params.max_itd_length;
	t.external_duplicate_marking = ctx->params.external_duplicate_marking;
	t.top_expressed_viral_verdict = nullptr; t.low_coverage_viral_verdict = nullptr;
	return AGPU_OK;
}

void refresh_annotation_view(agpu_ctx* ctx) {
	AnnotationView& a = ctx->annotation;
	a.n_genes = ctx->n_genes; a.n_dummy = ctx->n_dummy;
	a.gene_contig = ctx->gene_contig.as<uint16_t>(); a.gene_start = ctx->gene_start.as<int32_t>(); a.gene_end = ctx->gene_end.as<int32_t>();
	a.gene_bits = ctx->gene_bits.as<uint8_t>(); a.gene_exonic_length = ctx->gene_exonic_length.as<int32_t>();
	a.n_exons = ctx->n_exons;
	a.exon_start = ctx->exon_start.as<int32_t>(); a.exon_end = ctx->exon_end.as<int32_t>(); a.exon_gene = ctx->exon_gene.as<uint32_t>();
	a.exon_previous = ctx->exon_previous.as<int32_t>(); a.exon_next = ctx->exon_next.as<int32_t>();
	a.exon_cds_start = ctx->exon_cds_start.as<int32_t>(); a.exon_cds_end = ctx->exon_cds_end.as<int32_t>();
	a.dummy_start_key = ctx->dummy_start_key.as<uint64_t>(); a.dummy_end_key = ctx->dummy_end_key.as<uint64_t>();
}

// Algorithmic bytes per launch: every input column the kernel needs read once + every output written once.
uint64_t fragment_column_bytes(const agpu_ctx* ctx) { // n_aln, fbits + per slot contig/start/end/abits/cigar_offset/cigar_count
	return ctx->n * (1 + 1 + 3 * (2 + 4 + 4 + 1 + 4 + 2));
}
uint64_t annotate_stage1_bytes(const agpu_ctx* ctx) { // fragment columns + CIGARs in, abits + gene sets out
	return fragment_column_bytes(ctx) + ctx->cigar_pool.bytes + ctx->n * (3 * (1 + 1 + GENE_INLINE * 4));
}
uint64_t annotation_bytes(const agpu_ctx* ctx) { return annotate_stage1_bytes(ctx) + ctx->n * (1 + 3 * (2 + 1 + 1 + GENE_INLINE * 4)); }
uint64_t stage2_bytes(const agpu_ctx* ctx) { // fragment columns, CIGARs, gene sets, both sequences, the aligned reference bases (1 B per base), filter in/out
	return fragment_column_bytes(ctx) + ctx->cigar_pool.bytes + ctx->n * (3 * (1 + GENE_INLINE * 4) + 2 * 8 + 2) + ctx->seq_pool.bytes + 2 * ctx->seq_pool.bytes;
}
uint64_t low_entropy_bytes(const agpu_ctx* ctx) { // sequences + their offsets/lengths, clip lengths from the CIGARs, filter in/out
	return ctx->seq_pool.bytes + ctx->n * (2 * 8 + 3 * (4 + 2 + 8) + 1 + 3 + 2);
}

}

namespace agpu {
// everything behind the columns of a batch (uploaded from the host or built on the device by the ingest): result columns, pools, pristine copies, tables
int finish_batch_setup(agpu_ctx* ctx) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	if (ctx->max_read_length > 1024) { set_last_error("reads longer than 1024 nt are not supported by the low_entropy kernel's 8-plane k-mer counters"); return AGPU_ERR_INVALID; }
	if (!ctx->filter.allocate(n)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	HIP_CHECK(hipMemsetAsync(ctx->filter.ptr, 0, n ? n : 1, s));
	if (!ctx->pristine_fbits.allocate(n)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (n > 0) HIP_CHECK(hipMemcpyAsync(ctx->pristine_fbits.ptr, ctx->fbits.ptr, n, hipMemcpyDeviceToDevice, s));
	for (int k = 0; k < 3; ++k) {
		if (!ctx->pristine_abits[k].allocate(n)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		if (n > 0) HIP_CHECK(hipMemcpyAsync(ctx->pristine_abits[k].ptr, ctx->abits[k].ptr, n, hipMemcpyDeviceToDevice, s));
		if (!ctx->gene_count[k].allocate(n) || !ctx->genes[k].allocate(n * GENE_INLINE * sizeof(uint32_t))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		HIP_CHECK(hipMemsetAsync(ctx->gene_count[k].ptr, 0, n ? n : 1, s));
	}
	uint32_t pool_capacity = (uint32_t) std::min<uint64_t>(n / 2 + (1u << 20), 0x7FFFFFFFull);
	if (!ctx->gene_pool.allocate((size_t) pool_capacity * sizeof(uint32_t))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (!ctx->unmapped_keys.allocate((2 * n + 2) * sizeof(uint64_t))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	ctx->viral_pair_capacity = std::max<uint64_t>(1u << 20, n / 8);
	if (!ctx->viral_pairs.allocate(ctx->viral_pair_capacity * 2 * sizeof(uint32_t))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	HIP_CHECK(hipMemsetAsync(ctx->counters.ptr, 0, ctx->counters.bytes, s));
	HIP_CHECK(hipMemsetAsync(ctx->stage_counts.ptr, 0, ctx->stage_counts.bytes, s));

	BatchView& b = ctx->batch;
	b.n = n; b.first_rank = 0; b.n_aln = ctx->n_aln.as<uint8_t>(); b.fbits = ctx->fbits.as<uint8_t>(); b.filter = ctx->filter.as<uint8_t>(); b.group = ctx->group.as<uint32_t>();
	for (int k = 0; k < 3; ++k) {
		b.contig[k] = ctx->contig[k].as<uint16_t>(); b.start[k] = ctx->start[k].as<int32_t>(); b.end[k] = ctx->end[k].as<int32_t>(); b.abits[k] = ctx->abits[k].as<uint8_t>();
		b.cigar_offset[k] = ctx->cigar_offset[k].as<uint32_t>(); b.cigar_count[k] = ctx->cigar_count[k].as<uint16_t>();
		b.gene_count[k] = ctx->gene_count[k].as<uint8_t>(); b.genes[k] = ctx->genes[k].as<uint32_t>();
	}
	b.cigar_pool = ctx->cigar_pool.as<uint32_t>();
	for (int k = 0; k < 2; ++k) { b.seq_offset[k] = ctx->seq_offset[k].as<uint32_t>(); b.seq_length[k] = ctx->seq_length[k].as<uint32_t>(); }
	b.seq_pool = ctx->seq_pool.as<uint8_t>();
	b.gene_pool = ctx->gene_pool.as<uint32_t>(); b.gene_pool_used = ctx->counters.as<uint32_t>() + COUNTER_GENE_POOL; b.gene_pool_capacity = pool_capacity;
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->have_batch = true; ctx->annotated = false; ctx->stage1_done = false; ctx->stage2_done = false; ctx->fusions_done = false;
	ctx->evalue_done = false; ctx->iteration_order_done = false; ctx->kmer_index_done = false; ctx->genomic_support_marked = false; ctx->confidence_candidates = 0xFFFFFFFFu;
	ctx->n_dummy = 0; ctx->candidates_imported = false;
	ctx->global_n = 0; ctx->read_sharded = false; ctx->state_imported = false; ctx->sample_gene_read_counts_set = false; // (a batch of its own until agpu_set_shard / agpu_shard_keep say otherwise)
	refresh_annotation_view(ctx);
	if (ctx->have_genome) TRY(build_tables(ctx));
	return AGPU_OK;
}
}

// ---- C ABI ----------------------------------------------------------------------------------------

extern "C" {

const char* agpu_last_error(void) { return g_last_error.c_str(); }
int agpu_api_version(void) { return AGPU_API_VERSION; }

int agpu_device_count(void) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess) return 0;
	return count;
}

void agpu_default_params(agpu_params* p) { // source/options.cpp:71-107
	memset(p, 0, sizeof(*p));
	p->homopolymer_length = 6; p->min_read_through_distance = 10000; p->max_itd_length = 100; p->subsampling_threshold = 300;
	p->mismatch_pvalue_cutoff = 0.01; p->max_kmer_content = 0.6; p->evalue_cutoff = 0.3; p->max_mismapper_fraction = 0.8;
	p->fragment_length = 200; p->external_duplicate_marking = 0; p->strandedness = 0; p->exonic_fraction = 0.33; p->min_support = 2;
	for (int f = 1; f < AGPU_FILTER_COUNT; ++f) p->filter_enabled[f] = 1;
}

static agpu_ctx* create_context(int device, const agpu_params* params, std::shared_ptr<agpu::ScratchPool> pool);
agpu_ctx* agpu_create(int device, const agpu_params* params) { return create_context(device, params, std::shared_ptr<agpu::ScratchPool>()); }
agpu_ctx* agpu_create_sibling(agpu_ctx* of) {
	if (!of) { set_last_error("null argument"); return nullptr; }
	if (of->sibling) { set_last_error("the context has a sibling already"); return nullptr; }
	agpu_ctx* ctx = create_context(of->device, &of->params, of->pool);
	if (ctx) { ctx->profiling = of->profiling; ctx->sibling = of; of->sibling = ctx; ctx->keeps_batch_buffers = of->keeps_batch_buffers; }
	return ctx;
}
static agpu_ctx* create_context(int device, const agpu_params* params, std::shared_ptr<agpu::ScratchPool> pool) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_last_error("no HIP device visible: the hot path requires an MI355X (gfx950) and has no CPU fallback"); return nullptr; }
	if (device < 0 || device >= count) { set_last_error("invalid device ordinal"); return nullptr; }
	if (hipSetDevice(device) != hipSuccess) { set_last_error("hipSetDevice failed"); return nullptr; }
	agpu_ctx* ctx = new agpu_ctx(pool);
	ctx->device = device;
	if (params) ctx->params = *params; else agpu_default_params(&ctx->params);
	// Stream priorities: the pieces of a file (unwrap / inflate, CRC) in front, then the stages, the windows of the ingest last.  In a session with two lanes the stages of one sample
	// run beside the feed of the next.  Round 4a had the stages in front (with the pieces in front a 10^8-fragment step spent 1.33 s in its stages instead of 0.77:
	// profiles/r04c_bench100m.json); since the replay kernel of the windows takes 0.24 s instead of 0.8 the step is set by the feed, whose pushes wait for the kernels of the pieces:
	// 2.15 instead of 2.22 s per step, 2.03 instead of 2.08 with the faster writer (profiles/r04r_*).  ARRIBA_STREAM_PRIORITIES=stages: the stages in front, for measurements.
	int least_priority = 0, greatest_priority = 0;
	(void) hipDeviceGetStreamPriorityRange(&least_priority, &greatest_priority);
	const char* priorities = getenv("ARRIBA_STREAM_PRIORITIES");
	const bool pieces_first = priorities == nullptr || strcmp(priorities, "stages") != 0;
	if (hipStreamCreateWithPriority(&ctx->stream, hipStreamDefault, pieces_first ? (least_priority + greatest_priority) / 2 : greatest_priority) != hipSuccess || hipEventCreate(&ctx->event_start) != hipSuccess || hipEventCreate(&ctx->event_stop) != hipSuccess) {
		set_last_error("failed to create HIP stream/events"); delete ctx; return nullptr;
	}
	if (!ctx->counters.allocate(COUNTER_COUNT * sizeof(uint32_t)) || !ctx->stage_counts.allocate(16 * sizeof(unsigned long long))) { set_last_error("hipMalloc failed"); delete ctx; return nullptr; }
	(void) hipMemsetAsync(ctx->counters.ptr, 0, ctx->counters.bytes, ctx->stream);
	(void) hipMemsetAsync(ctx->stage_counts.ptr, 0, ctx->stage_counts.bytes, ctx->stream);
	{ std::lock_guard<std::mutex> lock(g_contexts_mutex); g_contexts.push_back(ctx); }
	return ctx;
}

void agpu_destroy(agpu_ctx* ctx) {
	if (!ctx) return;
	if (ctx->sibling) { ctx->sibling->sibling = nullptr; ctx->sibling = nullptr; }
	{ std::lock_guard<std::mutex> lock(g_contexts_mutex); g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), ctx), g_contexts.end()); }
	(void) hipSetDevice(ctx->device);
	(void) hipStreamSynchronize(ctx->stream);
	if (ctx->event_start) (void) hipEventDestroy(ctx->event_start);
	if (ctx->event_stop) (void) hipEventDestroy(ctx->event_stop);
	collect_kernel_samples(ctx);
	for (size_t k = 0; k < ctx->event_pool.size(); ++k) (void) hipEventDestroy(ctx->event_pool[k]);
	if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
	{ agpu::IngestProgress& progress = ctx->ingest_progress;
	  if (progress.work) { (void) hipStreamSynchronize(progress.work); (void) hipStreamDestroy(progress.work); }
	  for (size_t k = 0; k < progress.windows.size(); ++k) if (progress.windows[k].readback) (void) hipEventDestroy(progress.windows[k].readback);
	  for (size_t k = 0; k < progress.events.size(); ++k) (void) hipEventDestroy(progress.events[k]);
	  if (progress.host_words) (void) hipHostFree(progress.host_words); }
	if (ctx->piece_stream) { (void) hipStreamSynchronize(ctx->piece_stream); (void) hipStreamDestroy(ctx->piece_stream); }
	if (ctx->piece_stream2) { (void) hipStreamSynchronize(ctx->piece_stream2); (void) hipStreamDestroy(ctx->piece_stream2); }
	for (int k = 0; k < AGPU_PIECE_SLOTS; ++k) { if (ctx->piece_copied[k]) (void) hipEventDestroy(ctx->piece_copied[k]); if (ctx->piece_ready[k]) (void) hipEventDestroy(ctx->piece_ready[k]); if (ctx->piece_done[k]) (void) hipEventDestroy(ctx->piece_done[k]); }
	delete ctx;
}

int agpu_set_params(agpu_ctx* ctx, const agpu_params* params) {
	if (!ctx || !params) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	ctx->params = *params;
	if (ctx->have_batch && ctx->have_genome) TRY(build_tables(ctx));
	return AGPU_OK;
}

int agpu_upload_annotation(agpu_ctx* ctx, const agpu_annotation_view* in) {
	if (!ctx || !in) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	ctx->n_genes = in->n_genes; ctx->n_exons = in->n_exons; ctx->n_dummy = 0;
	TRY(upload(ctx->gene_contig, in->gene_contig, in->n_genes, s)); TRY(upload(ctx->gene_start, in->gene_start, in->n_genes, s)); TRY(upload(ctx->gene_end, in->gene_end, in->n_genes, s));
	TRY(upload(ctx->gene_bits, in->gene_bits, in->n_genes, s)); TRY(upload(ctx->gene_exonic_length, in->gene_exonic_length, in->n_genes, s));
	TRY(upload(ctx->exon_start, in->exon_start, in->n_exons, s)); TRY(upload(ctx->exon_end, in->exon_end, in->n_exons, s)); TRY(upload(ctx->exon_gene, in->exon_gene, in->n_exons, s));
	TRY(upload(ctx->exon_previous, in->exon_previous, in->n_exons, s)); TRY(upload(ctx->exon_next, in->exon_next, in->n_exons, s));
	TRY(upload(ctx->exon_cds_start, in->exon_cds_start, in->n_exons, s)); TRY(upload(ctx->exon_cds_end, in->exon_cds_end, in->n_exons, s));
	TRY(upload_index(in->exon_index, ctx->exon_index_contig_offset, ctx->exon_index_keys, ctx->exon_index_member_offset, ctx->exon_index_members, ctx->exon_index_bins, ctx->annotation.exon_index, s));
	TRY(upload_index(in->gene_index, ctx->gene_index_contig_offset, ctx->gene_index_keys, ctx->gene_index_member_offset, ctx->gene_index_members, ctx->gene_index_bins, ctx->annotation.gene_index, s));
	refresh_annotation_view(ctx);
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->have_annotation = true; ctx->annotated = false; ctx->have_splice_sites = false;
	return AGPU_OK;
}

int agpu_upload_genome(agpu_ctx* ctx, const agpu_genome_view* in) {
	if (!ctx || !in) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	TRY(upload(ctx->genome_contig_offset, in->contig_offset, (size_t) in->n_contigs + 1, s));
	TRY(upload(ctx->genome_contig_bits, in->contig_bits, in->n_contigs, s));
	{ // (16 bytes of padding behind the last base: the re-alignment of filter_mismappers reads the genome eight bases at a time)
		const size_t bases = in->contig_offset[in->n_contigs];
		if (!ctx->genome_bases.allocate(bases + 16)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		HIP_CHECK(hipMemsetAsync((char*) ctx->genome_bases.ptr + bases, 0, 16, s));
		if (bases > 0) HIP_CHECK(hipMemcpyAsync(ctx->genome_bases.ptr, in->bases, bases, hipMemcpyHostToDevice, s));
	}
	ctx->genome.n_contigs = in->n_contigs; ctx->genome.contig_offset = ctx->genome_contig_offset.as<uint64_t>();
	ctx->genome.contig_bits = ctx->genome_contig_bits.as<uint8_t>(); ctx->genome.bases = ctx->genome_bases.as<char>();
	ctx->host_contig_bits.assign(in->contig_bits, in->contig_bits + in->n_contigs);
	ctx->host_contig_offset.assign(in->contig_offset, in->contig_offset + in->n_contigs + 1);
	ctx->genome_size = 0; // source/filter_mismatches.cpp:103-108: total size of the interesting contigs
	for (uint32_t c = 0; c < in->n_contigs; ++c)
		if (in->contig_bits[c] & AGPU_CBIT_INTERESTING) ctx->genome_size += in->contig_offset[c + 1] - in->contig_offset[c];
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->have_genome = true;
	if (ctx->have_batch) TRY(build_tables(ctx));
	return AGPU_OK;
}

int agpu_upload_batch(agpu_ctx* ctx, const agpu_batch_view* in) {
	if (!ctx || !in) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	{ std::lock_guard<std::mutex> lock(ctx->profile_mutex); ctx->failed_launch.clear(); } // (a new sample)
	if (in->n >= 0xFFFFFFF0ull) { set_last_error("a batch holds at most 2^32-16 fragments; shard larger inputs"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = in->n;
	ctx->n = n;
	uint64_t bytes = 0;
	TRY(upload(ctx->n_aln, in->n_aln, n, s)); TRY(upload(ctx->fbits, in->fbits, n, s)); TRY(upload(ctx->group, in->group, n, s));
	bytes += n * (1 + 1 + 4);
	for (int k = 0; k < 3; ++k) {
		TRY(upload(ctx->contig[k], in->contig[k], n, s)); TRY(upload(ctx->start[k], in->start[k], n, s)); TRY(upload(ctx->end[k], in->end[k], n, s));
		TRY(upload(ctx->abits[k], in->abits[k], n, s)); TRY(upload(ctx->cigar_offset[k], in->cigar_offset[k], n, s)); TRY(upload(ctx->cigar_count[k], in->cigar_count[k], n, s));
		bytes += n * (2 + 4 + 4 + 1 + 4 + 2);
	}
	TRY(upload(ctx->cigar_pool, in->cigar_pool, in->cigar_pool_size, s));
	bytes += in->cigar_pool_size * 4;
	ctx->max_read_length = 0;
	for (int k = 0; k < 2; ++k) {
		TRY(upload(ctx->seq_offset[k], in->seq_offset[k], n, s)); TRY(upload(ctx->seq_length[k], in->seq_length[k], n, s));
		bytes += n * 8;
		for (uint64_t i = 0; i < n; ++i) if (in->seq_length[k][i] > ctx->max_read_length) ctx->max_read_length = in->seq_length[k][i];
	}
	TRY(upload(ctx->seq_pool, in->seq_pool, in->seq_pool_size, s));
	bytes += in->seq_pool_size;
	ctx->batch_input_bytes = bytes;
	TRY(agpu::finish_batch_setup(ctx));
	return AGPU_OK;
}

int agpu_reset(agpu_ctx* ctx) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	if (n > 0) {
		HIP_CHECK(hipMemcpyAsync(ctx->fbits.ptr, ctx->pristine_fbits.ptr, n, hipMemcpyDeviceToDevice, s));
		HIP_CHECK(hipMemsetAsync(ctx->filter.ptr, 0, n, s));
		for (int k = 0; k < 3; ++k) {
			HIP_CHECK(hipMemcpyAsync(ctx->abits[k].ptr, ctx->pristine_abits[k].ptr, n, hipMemcpyDeviceToDevice, s));
			HIP_CHECK(hipMemsetAsync(ctx->gene_count[k].ptr, 0, n, s));
		}
	}
	HIP_CHECK(hipMemsetAsync(ctx->counters.ptr, 0, ctx->counters.bytes, s));
	HIP_CHECK(hipMemsetAsync(ctx->stage_counts.ptr, 0, ctx->stage_counts.bytes, s));
	// the gene table shrinks back to the GTF genes; the dummy genes are re-created by agpu_annotate
	ctx->n_dummy = 0;
	refresh_annotation_view(ctx);
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->annotated = false; ctx->stage1_done = false; ctx->stage2_done = false; ctx->fusions_done = false; ctx->failed_launch.clear();
	ctx->evalue_done = false; ctx->iteration_order_done = false; ctx->kmer_index_done = false; ctx->genomic_support_marked = false; ctx->confidence_candidates = 0xFFFFFFFFu;
	return AGPU_OK;
}

int agpu_keep_batch_buffers(agpu_ctx* ctx, int on) {
	if (!ctx) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	ctx->keeps_batch_buffers = on != 0;
	if (ctx->sibling) ctx->sibling->keeps_batch_buffers = on != 0;
	return AGPU_OK;
}

int agpu_mark_multimappers(agpu_ctx* ctx, uint64_t* marked) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->keeps_batch_buffers) take_sample_buffers(ctx, false, true); // (the first stage of a sample: the buffers of the stages from the lane whose sample has just left the device)
	begin_timing(ctx);
	if (ctx->n > 0) { KernelTimer timer(ctx, "mark_multimappers_kernel", ctx->n * (4 + 1 + 1)); mark_multimappers_kernel<<<tally_grid(ctx->n, BLOCK), BLOCK, 0, ctx->stream>>>(ctx->batch, ctx->counters.as<uint32_t>()); }
	TRY(end_timing(ctx, ctx->n * (4 + 1 + 1)));
	uint32_t counters[COUNTER_COUNT];
	TRY(read_counters(ctx, counters));
	if (marked) *marked = counters[COUNTER_MARKED];
	return AGPU_OK;
}

int agpu_annotate_begin(agpu_ctx* ctx, uint64_t* n_unmapped) {
	if (!ctx || !ctx->have_batch || !ctx->have_annotation || !ctx->have_genome) { set_last_error("annotation, genome and batch must be uploaded first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	uint32_t* counters = ctx->counters.as<uint32_t>();
	begin_timing(ctx);
	if (n > 0) { KernelTimer timer(ctx, "annotate_stage1_kernel", annotate_stage1_bytes(ctx)); annotate_stage1_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->params.strandedness, ctx->unmapped_keys.as<uint64_t>(), counters); }
	uint32_t host_counters[COUNTER_COUNT];
	TRY(read_counters(ctx, host_counters));
	if (host_counters[COUNTER_ERROR] & ERROR_GENE_SET_OVERFLOW) { set_last_error("a gene set exceeded the device capacity"); return AGPU_ERR_CAPACITY; }
	ctx->n_unmapped = host_counters[COUNTER_UNMAPPED];
	ctx->annotate_begun = true;
	if (n_unmapped) *n_unmapped = ctx->n_unmapped;
	return AGPU_OK;
}

int agpu_copy_unmapped_positions(agpu_ctx* ctx, uint64_t* destination) {
	if (!ctx || !ctx->annotate_begun || !destination) { set_last_error("agpu_annotate_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->n_unmapped) HIP_CHECK(hipMemcpy(destination, ctx->unmapped_keys.ptr, (size_t) ctx->n_unmapped * 8, hipMemcpyDefault));
	return AGPU_OK;
}

int agpu_annotate_finish(agpu_ctx* ctx, const uint64_t* positions, uint64_t n_positions, uint32_t* n_dummy_genes) {
	if (!ctx || !ctx->annotate_begun) { set_last_error("agpu_annotate_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	uint32_t* counters = ctx->counters.as<uint32_t>();
	uint32_t host_counters[COUNTER_COUNT];
	if (positions != nullptr) { // the unmapped positions of all shards: dummy genes are cut from the sorted positions of the whole sample
		if (n_positions >= 0xFFFFFFF0ull) { set_last_error("too many unmapped positions"); return AGPU_ERR_CAPACITY; }
		if (!ctx->unmapped_keys.allocate((std::max<uint64_t>(n_positions, 2 * n + 2)) * 8)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		if (n_positions) HIP_CHECK(hipMemcpyAsync(ctx->unmapped_keys.ptr, positions, (size_t) n_positions * 8, hipMemcpyDefault, s));
		ctx->n_unmapped = (uint32_t) n_positions;
	}
	const uint32_t unmapped = ctx->n_unmapped;
	ctx->n_dummy = 0;
	if (unmapped > 0) {
		// sort the unmapped positions and cut them into dummy genes (source/arriba.cpp:232-260)
		if (!ctx->sorted_keys.allocate((size_t) unmapped * 8) || !ctx->scan_flags.allocate((size_t) unmapped * 4) || !ctx->scan_ids.allocate((size_t) unmapped * 4)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		size_t sort_bytes = 0, scan_bytes = 0;
		HIP_CHECK(rocprim::radix_sort_keys(nullptr, sort_bytes, ctx->unmapped_keys.as<uint64_t>(), ctx->sorted_keys.as<uint64_t>(), unmapped, 0, 48, s));
		HIP_CHECK(rocprim::inclusive_scan(nullptr, scan_bytes, ctx->scan_flags.as<uint32_t>(), ctx->scan_ids.as<uint32_t>(), unmapped, rocprim::plus<uint32_t>(), s));
		if (!ctx->sort_scratch.allocate(std::max(sort_bytes, scan_bytes))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		HIP_CHECK(rocprim::radix_sort_keys(ctx->sort_scratch.ptr, sort_bytes, ctx->unmapped_keys.as<uint64_t>(), ctx->sorted_keys.as<uint64_t>(), unmapped, 0, 48, s));
		dummy_flags_kernel<<<grid_for(unmapped), BLOCK, 0, s>>>(ctx->sorted_keys.as<uint64_t>(), unmapped, ctx->annotation.gene_index, ctx->scan_flags.as<uint32_t>());
		HIP_CHECK(rocprim::inclusive_scan(ctx->sort_scratch.ptr, scan_bytes, ctx->scan_flags.as<uint32_t>(), ctx->scan_ids.as<uint32_t>(), unmapped, rocprim::plus<uint32_t>(), s));
		uint32_t n_dummy = 0;
		HIP_CHECK(hipMemcpyAsync(&n_dummy, ctx->scan_ids.as<uint32_t>() + (unmapped - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		ctx->n_dummy = n_dummy;
		if (!ctx->dummy_start_key.allocate((size_t) n_dummy * 8) || !ctx->dummy_end_key.allocate((size_t) n_dummy * 8)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		dummy_write_kernel<<<grid_for(unmapped), BLOCK, 0, s>>>(ctx->sorted_keys.as<uint64_t>(), unmapped, ctx->scan_flags.as<uint32_t>(), ctx->scan_ids.as<uint32_t>(), ctx->dummy_start_key.as<uint64_t>(), ctx->dummy_end_key.as<uint64_t>());
		// extend the gene table by the dummy genes
		const uint32_t total = ctx->n_genes + n_dummy;
		TRY(grow_preserving(ctx->gene_contig, (size_t) ctx->n_genes * 2, (size_t) total * 2, s)); TRY(grow_preserving(ctx->gene_start, (size_t) ctx->n_genes * 4, (size_t) total * 4, s));
		TRY(grow_preserving(ctx->gene_end, (size_t) ctx->n_genes * 4, (size_t) total * 4, s)); TRY(grow_preserving(ctx->gene_bits, (size_t) ctx->n_genes, (size_t) total, s));
		TRY(grow_preserving(ctx->gene_exonic_length, (size_t) ctx->n_genes * 4, (size_t) total * 4, s));
		dummy_gene_table_kernel<<<grid_for(n_dummy), BLOCK, 0, s>>>(ctx->n_genes, n_dummy, ctx->dummy_start_key.as<uint64_t>(), ctx->dummy_end_key.as<uint64_t>(),
			ctx->gene_contig.as<uint16_t>(), ctx->gene_start.as<int32_t>(), ctx->gene_end.as<int32_t>(), ctx->gene_bits.as<uint8_t>(), ctx->gene_exonic_length.as<int32_t>());
	}
	refresh_annotation_view(ctx);
	if (n > 0) { KernelTimer timer(ctx, "annotate_stage2_kernel", n * (1 + 3 * (2 + 1 + 1 + GENE_INLINE * 4))); annotate_stage2_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, ctx->viral_pairs.as<uint32_t>(), (uint32_t) ctx->viral_pair_capacity, counters); }
	TRY(end_timing(ctx, annotation_bytes(ctx)));
	TRY(read_counters(ctx, host_counters));
	if (host_counters[COUNTER_ERROR] & ERROR_GENE_SET_OVERFLOW) { set_last_error("a gene set exceeded the device capacity"); return AGPU_ERR_CAPACITY; }
	if (host_counters[COUNTER_ERROR] & ERROR_VIRAL_PAIR_OVERFLOW) { set_last_error("too many virus-host fragments for the integration-site buffer"); return AGPU_ERR_CAPACITY; }
	if (n_dummy_genes) *n_dummy_genes = ctx->n_dummy;
	ctx->annotated = true; ctx->annotate_begun = false; ++ctx->annotation_serial;
	return AGPU_OK;
}

int agpu_annotate(agpu_ctx* ctx, uint32_t* n_dummy_genes) {
	TRY(agpu_annotate_begin(ctx, nullptr));
	float begin_ms = ctx->last_ms;
	(void) begin_ms;
	return agpu_annotate_finish(ctx, nullptr, 0, n_dummy_genes);
}

int agpu_set_shard(agpu_ctx* ctx, uint64_t first_rank, uint64_t global_n) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	if (first_rank + ctx->n > global_n || global_n >= 0xFFFFFFF0ull) { set_last_error("shard range out of bounds"); return AGPU_ERR_INVALID; }
	ctx->batch.first_rank = first_rank; ctx->global_n = global_n;
	return AGPU_OK;
}

int agpu_get_viral_integration_sites(agpu_ctx* ctx, uint32_t* pairs, uint64_t capacity, uint64_t* count) {
	if (!ctx || !ctx->annotated) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	uint32_t counters[COUNTER_COUNT];
	TRY(read_counters(ctx, counters));
	uint64_t available = counters[COUNTER_VIRAL_PAIRS];
	if (count) *count = available;
	if (pairs && available > 0) {
		uint64_t copy = std::min(available, capacity);
		HIP_CHECK(hipMemcpy(pairs, ctx->viral_pairs.ptr, copy * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
	}
	return AGPU_OK;
}

namespace {
// duplicate keys + the local table (smallest local index per key)
int build_local_duplicate_table(agpu_ctx* ctx, uint32_t& mask) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	uint64_t slots = 1024;
	while (slots < 2 * n) slots <<= 1;
	mask = (uint32_t) (slots - 1);
	if (!ctx->duplicate_keys.allocate((size_t) std::max<uint64_t>(n, 1) * sizeof(DuplicateKey)) || !ctx->duplicate_slots.allocate(slots * 4)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	HIP_CHECK(hipMemsetAsync(ctx->duplicate_slots.ptr, 0xFF, slots * 4, s));
	if (n > 0) {
		{ KernelTimer timer(ctx, "duplicate_keys_kernel", n * (1 + 2 * (2 + 4 + 4 + 1 + 4 + 2 + 8) + 12)); duplicate_keys_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->duplicate_keys.as<DuplicateKey>()); }
		{ KernelTimer timer(ctx, "duplicate_insert_kernel", n * (12 + 4)); duplicate_insert_kernel<<<grid_for(n), BLOCK, 0, s>>>(n, ctx->duplicate_keys.as<DuplicateKey>(), ctx->duplicate_slots.as<uint32_t>(), mask); }
	}
	return AGPU_OK;
}

int run_stage1(agpu_ctx* ctx, const uint8_t* top_verdict, const uint8_t* low_verdict, uint32_t mask, const uint8_t* global_duplicate) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	if (top_verdict) { TRY(upload(ctx->viral_verdict_top, top_verdict, ctx->genome.n_contigs, s)); ctx->tables.top_expressed_viral_verdict = ctx->viral_verdict_top.as<uint8_t>(); } else ctx->tables.top_expressed_viral_verdict = nullptr;
	if (low_verdict) { TRY(upload(ctx->viral_verdict_low, low_verdict, ctx->genome.n_contigs, s)); ctx->tables.low_coverage_viral_verdict = ctx->viral_verdict_low.as<uint8_t>(); } else ctx->tables.low_coverage_viral_verdict = nullptr;
	HIP_CHECK(hipMemsetAsync(ctx->stage_counts.ptr, 0, ctx->stage_counts.bytes, s));
	if (n > 0) {
		KernelTimer timer(ctx, "stage1_kernel", n * (1 + 1 + 12 + 4 + 3 * 2 + 1));
		stage1_kernel<<<tally_grid(n, BLOCK), BLOCK, 0, s>>>(ctx->batch, ctx->genome, ctx->tables, ctx->filter_enabled.as<uint8_t>(), ctx->duplicate_keys.as<DuplicateKey>(), ctx->duplicate_slots.as<uint32_t>(), mask, global_duplicate, ctx->stage_counts.as<unsigned long long>());
	}
	return AGPU_OK;
}
}

int agpu_read_filters_stage1(agpu_ctx* ctx, const uint8_t* top_verdict, const uint8_t* low_verdict) {
	if (!ctx || !ctx->annotated) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	uint32_t mask = 0;
	begin_timing(ctx);
	if (!ctx->params.external_duplicate_marking) TRY(build_local_duplicate_table(ctx, mask));
	TRY(run_stage1(ctx, top_verdict, low_verdict, mask, nullptr));
	TRY(end_timing(ctx, ctx->n * (3 * (2 + 4 + 4 + 1) + 2 * (4 + 2 + 8) + 12 * 2 + 8 + 2)));
	ctx->stage1_done = true;
	return AGPU_OK;
}

// ---- sharded samples: filter_duplicates keeps the first fragment in name order of the WHOLE sample (source/filter_duplicates.cpp:8-55)

int agpu_duplicates_begin(agpu_ctx* ctx, uint64_t* n_entries) {
	if (!ctx || !ctx->annotated) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	uint32_t mask = 0;
	begin_timing(ctx);
	TRY(build_local_duplicate_table(ctx, mask));
	DeviceBuffer& flags = ctx->scratch("duplicates.flags"); DeviceBuffer& winners = ctx->scratch("duplicates.winners"); DeviceBuffer& count = ctx->scratch("duplicates.count"); DeviceBuffer& scratch = ctx->scratch("duplicates.rocprim");
	if (!flags.allocate(std::max<uint64_t>(n, 1)) || !winners.allocate(std::max<uint64_t>(n, 1) * 4) || !count.allocate(16)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	uint32_t n_winners = 0;
	if (n > 0) {
		duplicate_winner_flag_kernel<<<grid_for(n), BLOCK, 0, s>>>(n, ctx->duplicate_keys.as<DuplicateKey>(), ctx->duplicate_slots.as<uint32_t>(), mask, flags.as<uint8_t>());
		size_t bytes = 0;
		HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), winners.as<uint32_t>(), count.as<uint32_t>(), n, s));
		if (bytes > scratch.capacity && !scratch.allocate(bytes)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
		HIP_CHECK(rocprim::select(scratch.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), winners.as<uint32_t>(), count.as<uint32_t>(), n, s)); // keeps name order
		HIP_CHECK(hipMemcpyAsync(&n_winners, count.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	if (!ctx->duplicate_entries.allocate((size_t) std::max<uint32_t>(n_winners, 1) * sizeof(DuplicateEntry))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (n_winners > 0) duplicate_entry_write_kernel<<<grid_for(n_winners), BLOCK, 0, s>>>(n_winners, winners.as<uint32_t>(), ctx->duplicate_keys.as<DuplicateKey>(), ctx->batch.first_rank, ctx->duplicate_entries.as<DuplicateEntry>());
	TRY(end_timing(ctx, n * 44 + (uint64_t) n_winners * 16));
	ctx->n_duplicate_entries = n_winners;
	if (n_entries) *n_entries = n_winners;
	return AGPU_OK;
}

int agpu_copy_duplicate_entries(agpu_ctx* ctx, void* destination) {
	if (!ctx || !destination) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->n_duplicate_entries) HIP_CHECK(hipMemcpy(destination, ctx->duplicate_entries.ptr, (size_t) ctx->n_duplicate_entries * sizeof(DuplicateEntry), hipMemcpyDefault));
	return AGPU_OK;
}

int agpu_read_filters_stage1_global(agpu_ctx* ctx, const void* entries, uint64_t n_entries, const uint8_t* top_verdict, const uint8_t* low_verdict) {
	if (!ctx || !ctx->annotated || (!entries && n_entries)) { set_last_error("agpu_annotate and agpu_duplicates_begin must run first"); return AGPU_ERR_INVALID; }
	if (n_entries >= 0xFFFFFFF0ull) { set_last_error("too many duplicate entries"); return AGPU_ERR_CAPACITY; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	DeviceBuffer& all_entries = ctx->scratch("duplicates.all_entries"); DeviceBuffer& global_slots = ctx->scratch("duplicates.global_slots"); DeviceBuffer& verdict = ctx->scratch("duplicates.verdict");
	uint64_t slots = 1024;
	while (slots < 2 * n_entries) slots <<= 1;
	const uint32_t mask = (uint32_t) (slots - 1);
	if (!all_entries.allocate(std::max<uint64_t>(n_entries, 1) * sizeof(DuplicateEntry)) || !global_slots.allocate(slots * 4) || !verdict.allocate(std::max<uint64_t>(n, 1))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	if (n_entries) HIP_CHECK(hipMemcpyAsync(all_entries.ptr, entries, (size_t) n_entries * sizeof(DuplicateEntry), hipMemcpyDefault, s));
	HIP_CHECK(hipMemsetAsync(global_slots.ptr, 0xFF, slots * 4, s));
	begin_timing(ctx);
	const uint8_t* global_duplicate = nullptr;
	if (!ctx->params.external_duplicate_marking && n > 0) {
		duplicate_entry_insert_kernel<<<grid_for(n_entries), BLOCK, 0, s>>>(n_entries, all_entries.as<DuplicateEntry>(), global_slots.as<uint32_t>(), mask);
		duplicate_global_verdict_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->duplicate_keys.as<DuplicateKey>(), all_entries.as<DuplicateEntry>(), global_slots.as<uint32_t>(), mask, verdict.as<uint8_t>());
		global_duplicate = verdict.as<uint8_t>();
	}
	TRY(run_stage1(ctx, top_verdict, low_verdict, 0, global_duplicate));
	TRY(end_timing(ctx, n * 60 + n_entries * 20));
	ctx->stage1_done = true;
	return AGPU_OK;
}

int agpu_fragment_length_samples_limited(agpu_ctx* ctx, uint32_t limit, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited);
int agpu_fragment_length_samples(agpu_ctx* ctx, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited) {
	return agpu_fragment_length_samples_limited(ctx, MAX_SAMPLES, mate_gaps, n_samples, fragments_visited);
}

int agpu_fragment_length_samples_limited(agpu_ctx* ctx, uint32_t limit, int32_t* mate_gaps, uint32_t* n_samples, uint64_t* fragments_visited) {
	if (limit == 0 || limit > MAX_SAMPLES) { set_last_error("the sample limit must be in 1..100001"); return AGPU_ERR_INVALID; }
	if (!ctx || !ctx->stage1_done) { set_last_error("agpu_read_filters_stage1 must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n, chunk = 1u << 20;
	if (!ctx->sample_flags.allocate(chunk) || !ctx->sample_values.allocate(chunk * 4) || !ctx->samples.allocate(MAX_SAMPLES * 4)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	uint32_t zero[3] = { 0, 0, 0 };
	HIP_CHECK(hipMemcpyAsync(ctx->counters.as<uint32_t>() + COUNTER_SAMPLES, zero, sizeof(zero), hipMemcpyHostToDevice, s));
	begin_timing(ctx);
	uint32_t counters[COUNTER_COUNT] = { 0 };
	for (uint64_t first = 0; first < n; first += chunk) {
		uint64_t count = std::min(chunk, n - first);
		sample_flags_kernel<<<grid_for(count), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, first, count, ctx->sample_flags.as<uint8_t>(), ctx->sample_values.as<int32_t>());
		sample_compact_kernel<<<1, 1024, 0, s>>>(first, count, ctx->sample_flags.as<uint8_t>(), ctx->sample_values.as<int32_t>(), ctx->samples.as<int32_t>(), ctx->counters.as<uint32_t>(), limit);
		TRY(read_counters(ctx, counters));
		if (counters[COUNTER_SAMPLES] >= limit) break;
	}
	TRY(end_timing(ctx, 0));
	uint32_t collected = counters[COUNTER_SAMPLES];
	if (n_samples) *n_samples = collected;
	if (fragments_visited) *fragments_visited = (collected >= limit) ? ((uint64_t) counters[COUNTER_VISITED_HI] << 32 | counters[COUNTER_VISITED_LO]) : n;
	if (mate_gaps && collected > 0) HIP_CHECK(hipMemcpy(mate_gaps, ctx->samples.ptr, (size_t) collected * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

int agpu_read_filters_stage2(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->stage1_done) { set_last_error("agpu_read_filters_stage1 must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	begin_timing(ctx);
	DeviceBuffer& selected = ctx->scratch("stage2.selected"); DeviceBuffer& selected_count = ctx->scratch("stage2.selected_count");
	if (!selected.allocate(std::max<uint64_t>(n, 1) * 4) || !selected_count.allocate(16)) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	HIP_CHECK(hipMemsetAsync(selected_count.ptr, 0, 16, s));
	if (n > 0) {
		// the grids are sized for all fragments (no round trip for the count); workgroups behind the end of the list return at once
		uint32_t* counts = selected_count.as<uint32_t>();
		{ KernelTimer timer(ctx, "select_fragments_kernel(unfiltered)", n * (1 + 4)); select_fragments_kernel<<<(unsigned int) ((n + SELECT_BLOCK - 1) / SELECT_BLOCK), SELECT_BLOCK, 0, s>>>(ctx->batch, ctx->tables, SELECT_UNFILTERED, selected.as<uint32_t>(), counts); }
		{ KernelTimer timer(ctx, "stage2_kernel", stage2_bytes(ctx)); stage2_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, ctx->tables, ctx->filter_enabled.as<uint8_t>(), selected.as<uint32_t>(), counts, ctx->stage_counts.as<unsigned long long>()); }
		if (ctx->params.filter_enabled[FILTER_low_entropy]) {
			{ KernelTimer timer(ctx, "select_fragments_kernel(low_entropy)", n * (1 + 4)); select_fragments_kernel<<<(unsigned int) ((n + SELECT_BLOCK - 1) / SELECT_BLOCK), SELECT_BLOCK, 0, s>>>(ctx->batch, ctx->tables, SELECT_LOW_ENTROPY_TEST, selected.as<uint32_t>(), counts + 1); }
			KernelTimer timer(ctx, "low_entropy_kernel", low_entropy_bytes(ctx));
			low_entropy_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->tables, selected.as<uint32_t>(), counts + 1, ctx->stage_counts.as<unsigned long long>());
		}
	}
	TRY(end_timing(ctx, stage2_bytes(ctx) + low_entropy_bytes(ctx)));
	unsigned long long counts[16];
	HIP_CHECK(hipMemcpy(counts, ctx->stage_counts.ptr, sizeof(counts), hipMemcpyDeviceToHost));
	{ // the two stages only touch the fragments that were selected for them: scale their algorithmic bytes accordingly
		uint32_t selected_counts[2] = { 0, 0 };
		HIP_CHECK(hipMemcpy(selected_counts, selected_count.ptr, sizeof(selected_counts), hipMemcpyDeviceToHost));
		for (size_t k = ctx->samples_done.size(); k > 0 && n > 0; --k) {
			KernelSample& sample = ctx->samples_done[k - 1];
			if (strcmp(sample.name, "low_entropy_kernel") == 0) sample.bytes = (uint64_t) ((double) sample.bytes * selected_counts[1] / n);
			else if (strcmp(sample.name, "stage2_kernel") == 0) { sample.bytes = (uint64_t) ((double) sample.bytes * selected_counts[0] / n); break; }
		}
		ctx->last_bytes = (uint64_t) ((double) stage2_bytes(ctx) * selected_counts[0] / std::max<uint64_t>(n, 1) + (double) low_entropy_bytes(ctx) * selected_counts[1] / std::max<uint64_t>(n, 1)) + n * 10;
	}
	if (remaining) {
		// order of execution (source/arriba.cpp:327-409) -> "(remaining=N)" after each stage
		static const int order[14] = { 1 /*duplicates*/, 30, 31, 32, 33, 4 /*read_through*/, 2, 3, 6, 7, 5, 8, 10, 36 };
		for (int f = 0; f < AGPU_FILTER_COUNT; ++f) remaining[f] = 0;
		uint64_t left = n;
		for (int k = 0; k < 14; ++k) { left -= counts[k]; remaining[order[k]] = left; }
	}
	ctx->stage2_done = true;
	return AGPU_OK;
}

int agpu_get_filters(agpu_ctx* ctx, uint8_t* filter) {
	if (!ctx || !ctx->have_batch || !filter) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->read_sharded) { // (the reads sharded over the ranks: the filters of the fragments of the SAMPLE, from the replicated states)
		if (!ctx->state_imported) { set_last_error("the reads of the sample are sharded: agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
		if (ctx->global_n) HIP_CHECK(hipMemcpy(filter, ctx->scratch("sharded.filter").ptr, ctx->global_n, hipMemcpyDeviceToHost));
		return AGPU_OK;
	}
	if (ctx->n) HIP_CHECK(hipMemcpy(filter, ctx->filter.ptr, ctx->n, hipMemcpyDeviceToHost));
	return AGPU_OK;
}
int agpu_get_alignment_bits(agpu_ctx* ctx, int slot, uint8_t* abits) {
	if (!ctx || !ctx->have_batch || !abits || slot < 0 || slot > 2) { set_last_error("bad argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->n) HIP_CHECK(hipMemcpy(abits, ctx->abits[slot].ptr, ctx->n, hipMemcpyDeviceToHost));
	return AGPU_OK;
}
int agpu_get_fragment_bits(agpu_ctx* ctx, uint8_t* fbits) {
	if (!ctx || !ctx->have_batch || !fbits) { set_last_error("bad argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->n) HIP_CHECK(hipMemcpy(fbits, ctx->fbits.ptr, ctx->n, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

int agpu_get_gene_sets(agpu_ctx* ctx, int slot, uint8_t* count, uint32_t* genes, uint64_t capacity, uint64_t* total) {
	if (!ctx || !ctx->annotated || slot < 0 || slot > 2) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	const uint64_t n = ctx->n;
	std::vector<uint8_t> counts(n);
	std::vector<uint32_t> inline_ids(n * GENE_INLINE);
	if (n) {
		HIP_CHECK(hipMemcpy(counts.data(), ctx->gene_count[slot].ptr, n, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(inline_ids.data(), ctx->genes[slot].ptr, n * GENE_INLINE * 4, hipMemcpyDeviceToHost));
	}
	uint32_t counters[COUNTER_COUNT];
	TRY(read_counters(ctx, counters));
	std::vector<uint32_t> pool(counters[COUNTER_GENE_POOL]);
	if (!pool.empty()) HIP_CHECK(hipMemcpy(pool.data(), ctx->gene_pool.ptr, pool.size() * 4, hipMemcpyDeviceToHost));
	uint64_t sum = 0;
	for (uint64_t i = 0; i < n; ++i) sum += counts[i];
	if (total) *total = sum;
	if (count && n) memcpy(count, counts.data(), n);
	if (genes) {
		if (capacity < sum) { set_last_error("gene buffer too small"); return AGPU_ERR_INVALID; }
		uint64_t at = 0;
		for (uint64_t i = 0; i < n; ++i) {
			const uint32_t* source = (counts[i] <= GENE_INLINE) ? &inline_ids[i * GENE_INLINE] : &pool[inline_ids[i * GENE_INLINE]];
			for (uint32_t k = 0; k < counts[i]; ++k) genes[at++] = source[k];
		}
	}
	return AGPU_OK;
}

int agpu_get_gene_table(agpu_ctx* ctx, uint32_t first, uint32_t count, uint16_t* contig, int32_t* start, int32_t* end, uint8_t* bits, int32_t* exonic_length) {
	if (!ctx || !ctx->have_annotation) { set_last_error("no annotation uploaded"); return AGPU_ERR_INVALID; }
	if ((uint64_t) first + count > (uint64_t) ctx->n_genes + ctx->n_dummy) { set_last_error("gene range out of bounds"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (count == 0) return AGPU_OK;
	if (contig) HIP_CHECK(hipMemcpy(contig, ctx->gene_contig.as<uint16_t>() + first, (size_t) count * 2, hipMemcpyDeviceToHost));
	if (start) HIP_CHECK(hipMemcpy(start, ctx->gene_start.as<int32_t>() + first, (size_t) count * 4, hipMemcpyDeviceToHost));
	if (end) HIP_CHECK(hipMemcpy(end, ctx->gene_end.as<int32_t>() + first, (size_t) count * 4, hipMemcpyDeviceToHost));
	if (bits) HIP_CHECK(hipMemcpy(bits, ctx->gene_bits.as<uint8_t>() + first, (size_t) count, hipMemcpyDeviceToHost));
	if (exonic_length) HIP_CHECK(hipMemcpy(exonic_length, ctx->gene_exonic_length.as<int32_t>() + first, (size_t) count * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

void agpu_debug_fail_allocation_in_finish(int count) { agpu::g_fail_allocations_in_finish = count; }
void agpu_debug_exhaust_memory_in_finish(int finishes) { agpu::g_exhausted_finishes = finishes; }
int agpu_set_profiling(agpu_ctx* ctx, int enabled) {
	if (!ctx) return AGPU_ERR_INVALID;
	std::lock_guard<std::mutex> lock(ctx->profile_mutex);
	ctx->profiling = enabled != 0;
	++ctx->profile_epoch; // (what was launched before this call and has not completed yet is dropped when it does)
	ctx->samples_done.clear();
	return AGPU_OK;
}
int agpu_get_kernel_profile(agpu_ctx* ctx, char* names, float* ms, uint64_t* bytes, uint32_t capacity, uint32_t* count) {
	if (!ctx || !count) return AGPU_ERR_INVALID;
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->piece_stream) { HIP_CHECK(hipStreamSynchronize(ctx->piece_stream)); HIP_CHECK(hipStreamSynchronize(ctx->piece_stream2)); }
	if (ctx->ingest_progress.work) HIP_CHECK(hipStreamSynchronize(ctx->ingest_progress.work));
	collect_kernel_samples(ctx);
	std::lock_guard<std::mutex> lock(ctx->profile_mutex);
	*count = (uint32_t) ctx->samples_done.size();
	for (uint32_t k = 0; k < *count && k < capacity; ++k) {
		if (names) { strncpy(names + (size_t) k * AGPU_KERNEL_NAME_LENGTH, ctx->samples_done[k].name, AGPU_KERNEL_NAME_LENGTH - 1); names[(size_t) k * AGPU_KERNEL_NAME_LENGTH + AGPU_KERNEL_NAME_LENGTH - 1] = 0; }
		if (ms) ms[k] = ctx->samples_done[k].ms;
		if (bytes) bytes[k] = ctx->samples_done[k].bytes;
	}
	return AGPU_OK;
}
int agpu_last_kernel_ms(agpu_ctx* ctx, float* ms) { if (!ctx || !ms) return AGPU_ERR_INVALID; *ms = ctx->last_ms; return AGPU_OK; }
int agpu_last_kernel_bytes(agpu_ctx* ctx, uint64_t* bytes) { if (!ctx || !bytes) return AGPU_ERR_INVALID; *bytes = ctx->last_bytes; return AGPU_OK; }

} // extern "C"
