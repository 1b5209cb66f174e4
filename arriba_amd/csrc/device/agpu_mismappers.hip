// arriba_amd/csrc/device/agpu_mismappers.hip -- make_kmer_index + filter_mismappers on the device
// (reference: source/filter_mismappers.cpp:16-359, called at source/arriba.cpp:547-565).
//
//   kmer_gene_flag_kernel       genes of unfiltered candidates with gene1 != gene2 (:50-58)
//   kmer_window_mark_kernel     bitmap over the genome: positions inside the padded window of an indexed gene (:65-74)
//   kmer_count/write_kernel     (table << 16 | 8-mer, position) for every marked position whose base is not N, in genome order
//   rocPRIM radix_sort_pairs    stable sort by (contig table, 8-mer): positions stay ascending inside a bucket (:77-84)
//   kmer_offsets_kernel         CSR offsets: 4^8 + 1 per indexed contig
//   splice_site_count/write     downstream splice sites of every gene (:16-31) as a second CSR
//   mismapper_flag/verdict      reads of unfiltered candidates -> greedy seed-and-extend re-alignment (mismapper_core.hpp), one wavefront per read,
//                               64 read positions of the seed search tried at once
//   mismapper_candidate_kernel  fraction of mis-mappers per candidate (:336-356)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "mismapper_core.hpp"
#include "device_utils.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
const int ALIGN_BLOCK = 128;
inline unsigned int grid_for(uint64_t n, int block = BLOCK) { return (unsigned int) ((n + block - 1) / block); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

__global__ void kmer_gene_flag_kernel(CandidateTable t, uint8_t* gene_flags) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n || t.filter[c] != FILTER_none) return;
	uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	if (gene1 == gene2) return; // sequence similarity only makes sense between different genes
	gene_flags[gene1] = 1; gene_flags[gene2] = 1;
}

// window of gene g: positions p with max(start - padding, 0) <= p and p + 8 < min(end + padding, contig size - 1)
__global__ void kmer_window_mark_kernel(AnnotationView ann, GenomeView genome, const uint8_t* gene_flags, int32_t padding, uint32_t* bitmap) {
	const uint32_t gene = blockIdx.x;
	if (!gene_flags[gene]) return;
	const uint32_t contig = ann.gene_contig[gene];
	const uint64_t contig_begin = genome.contig_offset[contig];
	const int64_t contig_size = (int64_t) (genome.contig_offset[contig + 1] - contig_begin);
	int64_t first = (int64_t) ann.gene_start[gene] - padding; if (first < 0) first = 0;
	int64_t gene_end = (int64_t) ann.gene_end[gene] + padding; if (gene_end > contig_size - 1) gene_end = contig_size - 1;
	const int64_t last = gene_end - KMER_LENGTH; // exclusive
	if (first >= last) return;
	const uint64_t bit_begin = contig_begin + (uint64_t) first, bit_end = contig_begin + (uint64_t) last;
	for (uint64_t word = (bit_begin >> 5) + threadIdx.x; word <= ((bit_end - 1) >> 5); word += blockDim.x) {
		uint32_t mask = 0xFFFFFFFFu;
		if (word == (bit_begin >> 5)) mask &= 0xFFFFFFFFu << (bit_begin & 31);
		if (word == ((bit_end - 1) >> 5)) mask &= 0xFFFFFFFFu >> (31 - ((bit_end - 1) & 31));
		atomicOr(&bitmap[word], mask);
	}
}

__device__ __forceinline__ uint32_t indexable_bits(const GenomeView& genome, const uint32_t* bitmap, uint64_t word, uint64_t genome_size) {
	uint32_t bits = bitmap[word];
	if (bits == 0) return 0;
	uint32_t valid = 0;
	for (uint32_t k = 0; k < 32; ++k) {
		uint64_t position = (word << 5) + k;
		if ((bits >> k & 1) && position < genome_size && genome.bases[position] != 'N') valid |= 1u << k; // masked regions are not indexed
	}
	return valid;
}
__global__ void kmer_count_kernel(GenomeView genome, const uint32_t* bitmap, uint64_t n_words, uint64_t genome_size, uint32_t* counts) {
	uint64_t word = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (word >= n_words) return;
	counts[word] = __popc(indexable_bits(genome, bitmap, word, genome_size));
}
__global__ void kmer_write_kernel(GenomeView genome, const uint32_t* bitmap, uint64_t n_words, uint64_t genome_size, const uint32_t* offsets, const uint32_t* contig_table, uint32_t* keys, int32_t* positions) {
	uint64_t word = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (word >= n_words) return;
	uint32_t valid = indexable_bits(genome, bitmap, word, genome_size);
	if (valid == 0) return;
	uint32_t at = offsets[word];
	// contig of the first position of this word
	uint32_t lo = 0, hi = genome.n_contigs;
	const uint64_t base = word << 5;
	while (lo + 1 < hi) { uint32_t mid = (lo + hi) >> 1; if (genome.contig_offset[mid] <= base) lo = mid; else hi = mid; }
	uint32_t contig = lo;
	for (uint32_t k = 0; k < 32; ++k) {
		if (!(valid >> k & 1)) continue;
		uint64_t position = base + k;
		while (contig + 1 < genome.n_contigs && position >= genome.contig_offset[contig + 1]) ++contig;
		uint32_t kmer = 0;
		for (int j = 0; j < KMER_LENGTH; ++j) kmer = kmer << 2 | kmer_digit_of_char(genome.bases[position + j]);
		keys[at] = contig_table[contig] << 16 | kmer;
		positions[at] = (int32_t) (position - genome.contig_offset[contig]);
		++at;
	}
}
__global__ void kmer_offsets_kernel(const uint32_t* sorted_keys, uint32_t n, uint32_t n_entries, uint32_t* offsets) {
	uint32_t entry = blockIdx.x * BLOCK + threadIdx.x;
	if (entry > n_entries) return;
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (sorted_keys[mid] < entry) lo = mid + 1; else hi = mid; }
	offsets[entry] = lo;
}

// reference: get_downstream_splice_sites (source/filter_mismappers.cpp:16-31); write == false counts
__global__ void splice_site_kernel(AnnotationView ann, GenomeView genome, uint32_t n_genes_total, const uint32_t* offsets, uint32_t* counts, int32_t* sites, uint32_t* bits, bool write) {
	uint32_t gene = blockIdx.x * BLOCK + threadIdx.x;
	if (gene >= n_genes_total) return;
	const FlatIndexView& index = ann.exon_index;
	uint32_t contig = ann.gene_contig[gene];
	uint32_t found = 0;
	if (contig < index.n_contigs && index.contig_offset[contig] != index.contig_offset[contig + 1]) {
		uint32_t k = index_lower_bound(index, contig, ann.gene_start[gene]);
		const uint32_t contig_end = index.contig_offset[contig + 1];
		const int32_t gene_end = ann.gene_end[gene];
		for (; k != contig_end && index.keys[k] <= gene_end; ++k)
			if (is_breakpoint_spliced(ann, gene, false, index.keys[k])) {
				if (write) { // ... and its bit in the map of all splice sites (SpliceSiteView::bits)
					sites[offsets[gene] + found] = index.keys[k];
					const uint64_t bit = genome.contig_offset[contig] + (uint64_t) index.keys[k];
					if (index.keys[k] >= 0 && contig < genome.n_contigs && bit < genome.contig_offset[genome.n_contigs]) atomicOr(&bits[bit >> 5], 1u << (bit & 31));
				}
				++found;
			}
	}
	if (!write) counts[gene] = found;
}

// (the candidates [first, end): all of them, or a window when the discordant lists are implicit -- for_each_list_window)
__global__ void mismapper_flag_kernel(BatchView b, CandidateTable t, uint8_t* read_flags, unsigned long long* first_entry, bool by_candidate, uint32_t first, uint32_t end) {
	uint32_t c = first + blockIdx.x * BLOCK + threadIdx.x;
	if (c >= end || t.filter[c] != FILTER_none) return;
	const uint64_t* offsets = t.list_offset + 3 * (uint64_t) c;
	for (uint64_t k = offsets[0]; k < offsets[3]; ++k) {
		const uint64_t read = (uint64_t) t.read_lists[k] - b.first_rank; // (the lists hold global name ranks: a context that holds one shard of the sample takes the reads it holds)
		if (read >= b.n || b.filter[read] != FILTER_none) continue;
		if (!read_flags[read]) read_flags[read] = 1;
		const unsigned long long key = by_candidate ? c : k;
		if (first_entry[read] > key) atomicMin(&first_entry[read], key); // where the read stands first in the lists: jobs in that order keep the reads of one candidate together
	}
}
__global__ void mismapper_job_key_kernel(const uint32_t* jobs, uint32_t n_jobs, const unsigned long long* first_entry, unsigned long long* keys) {
	const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j < n_jobs) keys[j] = first_entry[jobs[j]];
}

// The thread-per-read pass of round 2, off by default since the second pass sweeps every seed of a read once and is the cheaper place for every read (see the launch below;
// ARRIBA_MISMAPPER_FIRST_PASS=1 turns it on for measurements).  One thread per read: the reference's seed-and-extend search (mismapper_core.hpp) as a chain of dependent
// look-ups -- k-mer table, position list, genome bases; the frame of the call being worked on sits in registers (the stack in scratch memory only sees calls and returns), the
// bases of the segment sit in LDS (one column per lane), the stack is short (ALIGN_SHALLOW_DEPTH: enough for reads of 128 nt).  The jobs are ordered by the candidate that lists
// the read first: the lanes of a wavefront then work on reads of one gene pair.  A read that runs out of its step budget (or of stack) is put on the `heavy` list: the verdict
// of a read does not depend on who computes it.
const int SEGMENT_CACHE = 128; // bases of a segment kept in LDS per lane (longer segments are read from HBM)
// Steps of the search loop + bases compared that a read gets in that pass.  At 10^8 fragments first + second pass took 641 + 306 ms with 2048 steps, 235 + 471 ms with 512,
// 119 + 539 ms with 256 (profiles/r03e_mismapper_sweep.txt; ARRIBA_FIRST_PASS_STEPS for measurements) -- and 0 + 460 ms without the pass (profiles/r03r_mismapper_passes.txt).
const int64_t FIRST_PASS_STEPS = 256;
__global__ void __launch_bounds__(ALIGN_BLOCK, 5) mismapper_verdict_kernel(BatchView b, AnnotationView ann, GenomeView genome, KmerIndexView kmers, SpliceSiteView splice, const uint32_t* jobs, uint32_t n_jobs, int32_t max_mate_gap,
                                                                        int64_t first_pass_steps, uint32_t* heavy, unsigned int* counters /* [1] discarded, [3] heavy */) {
	__shared__ uint32_t block_sum;
	__shared__ uint8_t segment_bases[SEGMENT_CACHE * ALIGN_BLOCK];
	const uint32_t j = blockIdx.x * ALIGN_BLOCK + threadIdx.x;
	uint32_t mine = 0;
	if (j < n_jobs) {
		AlignFrame stack[ALIGN_SHALLOW_DEPTH];
		int64_t budget = first_pass_steps;
		AlignRunner runner; runner.stack = stack; runner.lane = 0; runner.lanes = 1; runner.budget = &budget; runner.max_depth = ALIGN_SHALLOW_DEPTH;
		runner.cache = segment_bases + threadIdx.x; runner.cache_stride = ALIGN_BLOCK; runner.cache_capacity = SEGMENT_CACHE;
		const uint32_t read = jobs[j];
		const bool verdict = is_mismapper(b, ann, genome, kmers, splice, read, max_mate_gap, runner);
		if (runner.exhausted()) heavy[atomicAdd(&counters[3], 1u)] = read;
		else if (verdict) { b.filter[read] = FILTER_mismappers; mine = 1; }
	}
	block_tally(mine, &counters[1], &block_sum);
}

// Second pass, one wavefront per heavy read: the read positions one after the other, the seeds of a position split among the 64 lanes (every seed of the outermost
// loop of align() is an independent attempt), full stack, no budget; the segment in LDS is shared by the lanes, and so is the memo of failed nested calls
// (AlignMemo, mismapper_core.hpp: it turns the exponential re-evaluation of the reference's recursion into one search per distinct call).  The workgroups are
// persistent: each owns one memo table in HBM and takes the next heavy read from a queue (counters[4]) when it is done with one -- the searches differ in
// length by orders of magnitude, a fixed share per workgroup would wait for the unluckiest one.  The kernel waits for dependent loads (k-mer table -> hit list ->
// genome bases -> memo slot), so the number of wavefronts in flight is what sets its speed: 5120 workgroups of one wavefront = 5 per SIMD.
const uint32_t MEMO_SLOTS_LOG2 = 16;    // 512 KB per workgroup, 2.7 GB for 5120 of them.  2^20 (8 MB, 41 GB in all, cleared per sample) until round 6, sized for the reads of long genes that make 10^5..10^6
                                        // distinct nested calls -- but the time of the kernel is the same from 2^11 to 2^20 slots at 10^8 fragments (profiles/r06e_heavy_ab.txt: 363 / 369 / 368 ms):
                                        // a full table costs repeated searches, never correctness, and the 38 GB are the difference between a rank that holds its share of a sample beside
                                        // the tables and one that does not.  ARRIBA_MEMO_SLOTS_LOG2 for measurements
const uint32_t TASK_CAPACITY_LOG2 = 15; // 2^15 listed calls of 16 bytes per workgroup, and as many for the calls of a block beyond the memory of the sweep: 5 GB for 5120 workgroups (2^17, 21 GB,
                                        // until round 6; the longest list of the 10^8-fragment sample has 1 600 calls).  A search that lists more is done by the recursion.  ARRIBA_TASK_CAPACITY_LOG2
const uint32_t HEAVY_WORKGROUPS = 5120; // five wavefronts per SIMD (launch bounds below)
// Tried and taken back in round 5 (profiles/r05o_waves6.json, r05p_*, r05q_*): (a) six wavefronts per SIMD -- 80 VGPRs, 177 spilled: 457 ms instead of 419 at 10^8 fragments;
// (b) TWO searches per wavefront, the two strands of a gene at once on 32 lanes each with a sweep, a memo and lists of their own (the calls of a block kept in LDS cut from 256 to
// 64 to make room): the same verdict bytes under every schedule of the GPU tier, but 496 ms -- the halves of a wavefront go separate ways through the walks and the look-ups of the
// read positions take twice the turns; with 64 calls in LDS and one search per wavefront: 432 ms.
// SWEEP_ONLY: the searches go through the sweep or are given up -- no recursion, no stack of frames in the kernel; the reads of the searches given up are appended to `leftover`
// (counters[5]) and done by the instantiation that holds everything.  queue: the index of the queue's counter (the second launch has a queue of its own).
template <int WAVES_PER_SIMD, bool SWEEP_ONLY> __global__ void __launch_bounds__(64, WAVES_PER_SIMD) mismapper_heavy_kernel(BatchView b, AnnotationView ann, GenomeView genome, KmerIndexView kmers, SpliceSiteView splice, const uint32_t* heavy, uint32_t n_heavy, int32_t max_mate_gap,
                                                             unsigned long long* memo_tables, uint32_t memo_slots, unsigned long long* task_lists, uint32_t task_capacity, bool by_sweep, unsigned long long* read_times, unsigned int* counters, uint32_t* leftover, int queue, bool lds_front, bool strands_together) {
	__shared__ uint8_t segment_bases[2 * 304]; // (the segment being searched and its reverse complement)
	__shared__ uint8_t segment_chars[2 * 312];  // (... as characters, eight zero bytes behind each)
	__shared__ AlignSweep sweep;
	__shared__ AlignMemo memo;
	__shared__ AlignWorklist worklist;
	__shared__ uint32_t worklist_state[4];
	__shared__ unsigned long long memo_front[ALIGN_MEMO_FRONT_SLOTS]; // (mismapper_core.hpp: the first keys of a search never leave the CU)
	__shared__ unsigned long long list_head[2 * ALIGN_LIST_HEAD_TASKS];
	__shared__ uint32_t memo_spilled;
	__shared__ uint32_t next_job;
	__shared__ uint32_t study[16];
	if (threadIdx.x == 0) {
		worklist.stats = read_times != nullptr ? study : nullptr;
		memo.slots = memo_tables + (size_t) blockIdx.x * memo_slots; memo.mask = memo_slots - 1; memo.epoch = 0;
		memo.front = lds_front ? memo_front : nullptr; memo.front_mask = lds_front ? ALIGN_MEMO_FRONT_SLOTS - 1 : 0; memo.spilled = lds_front ? &memo_spilled : nullptr; memo_spilled = 0;
		worklist.head = lds_front ? list_head : nullptr; worklist.head_capacity = lds_front ? ALIGN_LIST_HEAD_TASKS : 0;
		worklist.words = task_lists != nullptr ? task_lists + (size_t) blockIdx.x * task_capacity * 2 : nullptr; worklist.capacity = task_capacity; worklist.state = worklist_state;
		worklist.sweep = by_sweep ? &sweep : nullptr; // one sweep over the read positions, every seed walked once (mismapper_core.hpp: AlignSweep)
		worklist.relevant_words = task_lists != nullptr ? task_lists + ((size_t) gridDim.x + blockIdx.x) * task_capacity * 2 : nullptr; worklist.relevant_capacity = task_capacity; // (the second half of the buffer: the calls of a block beyond those kept in LDS)
	}
	for (uint32_t k = threadIdx.x; k < ALIGN_MEMO_FRONT_SLOTS; k += 64) memo_front[k] = 0; // (epoch 0 = free)
	__syncthreads();
	__shared__ int64_t given_up; // (SWEEP_ONLY: < 0 when a search of the read was not one for the sweep)
	AlignFrame stack[SWEEP_ONLY ? 1 : ALIGN_MAX_DEPTH];
	AlignRunnerT<SWEEP_ONLY> runner; runner.stack = stack; runner.lane = threadIdx.x; runner.lanes = 64; runner.budget = SWEEP_ONLY ? &given_up : nullptr; runner.max_depth = SWEEP_ONLY ? 1 : ALIGN_MAX_DEPTH;
	runner.lanes_share_seeds = true; // a read lands here because its search is long: the lanes split the seeds of every read position (when the task list is off or overflows)
	runner.memo = &memo;
	runner.worklist = task_lists != nullptr ? &worklist : nullptr; // the search as rounds of up to 64 tasks (mismapper_core.hpp: AlignWorklist)
	runner.cache = segment_bases; runner.cache_stride = 1; runner.cache_capacity = 304; runner.cache2 = segment_bases + 304; runner.strands_together = strands_together; runner.chars = segment_chars; runner.chars2 = segment_chars + 312;
	while (true) {
		__syncthreads(); // (every lane has read next_job of the previous round)
		if (threadIdx.x == 0) { next_job = atomicAdd(&counters[queue], 1u); given_up = 0; }
		__syncthreads();
		if (next_job >= n_heavy) break;
		const uint32_t read = heavy[next_job];
		const unsigned long long started = read_times != nullptr ? wall_clock64() : 0ull;
		if (read_times != nullptr && threadIdx.x == 0) for (int k = 0; k < 16; ++k) study[k] = 0;
		const bool verdict = is_mismapper(b, ann, genome, kmers, splice, read, max_mate_gap, runner);
		if (SWEEP_ONLY && runner.exhausted()) { if (threadIdx.x == 0) leftover[atomicAdd(&counters[5], 1u)] = read; continue; } // (the same for every lane: written behind a barrier)
		if (verdict && threadIdx.x == 0) { b.filter[read] = FILTER_mismappers; atomicAdd(&counters[1], 1u); }
		if (read_times != nullptr && threadIdx.x == 0) { // (ARRIBA_MISMAPPER_TIMES=1: ticks of the 100 MHz clock per read, and what the search of the read consisted of)
			read_times[4 * (size_t) next_job] = wall_clock64() - started; read_times[4 * (size_t) next_job + 1] = (unsigned long long) study[0] << 32 | study[1]; read_times[4 * (size_t) next_job + 2] = (unsigned long long) study[2] << 32 | study[3];
			read_times[4 * (size_t) next_job + 3] = (unsigned long long) read << 8 | b.n_aln[read];
			atomicAdd(&read_times[4 * (size_t) n_heavy], (unsigned long long) study[4]); atomicAdd(&read_times[4 * (size_t) n_heavy + 1], (unsigned long long) study[5]); atomicAdd(&read_times[4 * (size_t) n_heavy + 2], (unsigned long long) study[6]);
			for (int k = 8; k < 16; ++k) atomicAdd(&read_times[4 * (size_t) n_heavy + 4 + (k - 8)], (unsigned long long) study[k]);
		}
	}
}

// filter_mismappers shared out over several GPUs that hold the same batch: the jobs of one part, the verdicts of a part for the exchange, the verdicts of all parts applied
__global__ void mismapper_take_part_kernel(const uint32_t* jobs, uint32_t n_jobs, uint32_t part, uint32_t parts, uint32_t* mine) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if ((uint64_t) k * parts + part < n_jobs) mine[k] = jobs[(uint64_t) k * parts + part];
}
__global__ void mismapper_collect_kernel(BatchView b, const uint32_t* jobs, uint32_t n_jobs, uint32_t part, uint32_t parts, uint8_t* verdicts) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_jobs) verdicts[j] = (j % parts == part && b.filter[jobs[j]] == FILTER_mismappers) ? 1 : 0;
}
__global__ void mismapper_apply_kernel(BatchView b, const uint32_t* jobs, uint32_t n_jobs, const uint8_t* verdicts, unsigned int* counters) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_jobs && verdicts[j]) { b.filter[jobs[j]] = FILTER_mismappers; atomicAdd(&counters[1], 1u); }
}

__global__ void mismapper_candidate_kernel(BatchView b, CandidateTable t, float max_mismapper_fraction, bool count_only, unsigned int* remaining, uint32_t first, uint32_t end) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	for (uint32_t c = first + blockIdx.x * BLOCK + threadIdx.x; c < end; c += gridDim.x * BLOCK) {
		if (t.filter[c] != FILTER_none) continue;
		if (!count_only && count_candidate_mismappers(b, t, c, max_mismapper_fraction)) t.filter[c] = FILTER_mismappers; else ++kept;
	}
	block_tally(kept, remaining, &block_sum);
}

int build_splice_sites(agpu_ctx* ctx) {
	hipStream_t s = ctx->stream;
	const uint32_t n_genes_total = ctx->n_genes + ctx->n_dummy;
	DeviceBuffer& counts = ctx->scratch("mismappers.splice_counts"); DeviceBuffer& scratch = ctx->scratch("mismappers.rocprim");
	ALLOC(counts, ((size_t) n_genes_total + 1) * 4); ALLOC(ctx->splice_offset, ((size_t) n_genes_total + 1) * 4);
	HIP_CHECK(hipMemsetAsync(counts.ptr, 0, ((size_t) n_genes_total + 1) * 4, s));
	splice_site_kernel<<<grid_for(n_genes_total), BLOCK, 0, s>>>(ctx->annotation, ctx->genome, n_genes_total, nullptr, counts.as<uint32_t>(), nullptr, nullptr, false);
	size_t bytes = 0;
	HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, counts.as<uint32_t>(), ctx->splice_offset.as<uint32_t>(), 0u, (size_t) n_genes_total + 1, rocprim::plus<uint32_t>(), s));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, bytes, counts.as<uint32_t>(), ctx->splice_offset.as<uint32_t>(), 0u, (size_t) n_genes_total + 1, rocprim::plus<uint32_t>(), s));
	uint32_t total = 0;
	HIP_CHECK(hipMemcpyAsync(&total, ctx->splice_offset.as<uint32_t>() + n_genes_total, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	ALLOC(ctx->splice_sites, (size_t) std::max<uint32_t>(total, 1) * 4);
	const size_t bitmap_bytes = ((size_t) (ctx->host_contig_offset[ctx->genome.n_contigs] + 31) / 32 + 16) * 4; // (a walk looks up to the length of its segment behind a position, and segments of 300 bases and more are not re-aligned at all -- align_both_strands, as the reference: 16 words = 512 bits of room behind the last contig)
	ALLOC(ctx->splice_bits, bitmap_bytes);
	HIP_CHECK(hipMemsetAsync(ctx->splice_bits.ptr, 0, bitmap_bytes, s));
	splice_site_kernel<<<grid_for(n_genes_total), BLOCK, 0, s>>>(ctx->annotation, ctx->genome, n_genes_total, ctx->splice_offset.as<uint32_t>(), nullptr, ctx->splice_sites.as<int32_t>(), ctx->splice_bits.as<uint32_t>(), true);
	ctx->splice_sites_for_dummy = ctx->n_dummy;
	ctx->have_splice_sites = true;
	return AGPU_OK;
}

}

extern "C" int agpu_set_read_filters(agpu_ctx* ctx, const uint8_t* filter) {
	if (!ctx || !ctx->have_batch || !filter) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->n) HIP_CHECK(hipMemcpy(ctx->filter.ptr, filter, ctx->n, hipMemcpyHostToDevice));
	return AGPU_OK;
}

extern "C" int agpu_make_kmer_index(agpu_ctx* ctx, int32_t padding, uint64_t* n_positions) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint32_t n_genes_total = ctx->n_genes + ctx->n_dummy;
	const uint32_t n_contigs = ctx->genome.n_contigs;
	const uint64_t genome_size = ctx->host_contig_offset[n_contigs];
	const uint64_t n_words = (genome_size + 31) >> 5;
	if (padding < 0) padding = 0;
	ctx->kmer_index_done = false;
	DeviceBuffer& gene_flags = ctx->scratch("mismappers.gene_flags"); DeviceBuffer& bitmap = ctx->scratch("mismappers.bitmap"); DeviceBuffer& counts = ctx->scratch("mismappers.counts");
	DeviceBuffer& offsets = ctx->scratch("mismappers.offsets"); DeviceBuffer& keys = ctx->scratch("mismappers.keys"); DeviceBuffer& positions = ctx->scratch("mismappers.positions");
	DeviceBuffer& sorted_keys = ctx->scratch("mismappers.sorted_keys"); DeviceBuffer& scratch = ctx->scratch("mismappers.rocprim");
	ALLOC(gene_flags, std::max<uint32_t>(n_genes_total, 1)); ALLOC(bitmap, (n_words + 1) * 4); ALLOC(counts, (n_words + 1) * 4); ALLOC(offsets, (n_words + 1) * 4);
	HIP_CHECK(hipMemsetAsync(gene_flags.ptr, 0, std::max<uint32_t>(n_genes_total, 1), s));
	HIP_CHECK(hipMemsetAsync(bitmap.ptr, 0, (n_words + 1) * 4, s));
	HIP_CHECK(hipMemsetAsync(counts.ptr, 0, (n_words + 1) * 4, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) { KernelTimer timer(ctx, "kmer_gene_flag_kernel", (uint64_t) C * 9); kmer_gene_flag_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->candidates, gene_flags.as<uint8_t>()); }

	// contigs that hold an indexed gene get an offset table
	std::vector<uint8_t> host_flags(n_genes_total);
	std::vector<uint16_t> host_gene_contig(n_genes_total);
	if (n_genes_total) {
		HIP_CHECK(hipMemcpyAsync(host_flags.data(), gene_flags.ptr, n_genes_total, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(host_gene_contig.data(), ctx->gene_contig.ptr, (size_t) n_genes_total * 2, hipMemcpyDeviceToHost, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	std::vector<uint32_t> contig_table(n_contigs, NO_KMER_TABLE);
	uint32_t n_tables = 0, n_indexed_genes = 0;
	for (uint32_t g = 0; g < n_genes_total; ++g)
		if (host_flags[g]) {
			++n_indexed_genes;
			if (host_gene_contig[g] < n_contigs && contig_table[host_gene_contig[g]] == NO_KMER_TABLE) contig_table[host_gene_contig[g]] = 0;
		}
	for (uint32_t contig = 0; contig < n_contigs; ++contig) if (contig_table[contig] != NO_KMER_TABLE) contig_table[contig] = n_tables++;
	if (n_tables >= 65536) { set_last_error("too many contigs with indexed genes"); return AGPU_ERR_CAPACITY; }
	ALLOC(ctx->kmer_contig_table, (size_t) std::max<uint32_t>(n_contigs, 1) * 4);
	HIP_CHECK(hipMemcpyAsync(ctx->kmer_contig_table.ptr, contig_table.data(), (size_t) n_contigs * 4, hipMemcpyHostToDevice, s));

	uint32_t total = 0;
	if (n_indexed_genes > 0) {
		{ KernelTimer timer(ctx, "kmer_window_mark_kernel", 0); kmer_window_mark_kernel<<<n_genes_total, BLOCK, 0, s>>>(ctx->annotation, ctx->genome, gene_flags.as<uint8_t>(), padding, bitmap.as<uint32_t>()); }
		{ KernelTimer timer(ctx, "kmer_count_kernel", n_words * 8); kmer_count_kernel<<<grid_for(n_words), BLOCK, 0, s>>>(ctx->genome, bitmap.as<uint32_t>(), n_words, genome_size, counts.as<uint32_t>()); }
		size_t bytes = 0;
		HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
		HIP_CHECK(hipMemcpyAsync(&total, offsets.as<uint32_t>() + n_words, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	ALLOC(ctx->kmer_positions, (size_t) std::max<uint32_t>(total, 1) * 4);
	ALLOC(ctx->kmer_offsets, ((size_t) std::max<uint32_t>(n_tables, 1) * KMER_COUNT + 1) * 4);
	if (total > 0) {
		ALLOC(keys, (size_t) total * 4); ALLOC(positions, (size_t) total * 4); ALLOC(sorted_keys, (size_t) total * 4);
		{ KernelTimer timer(ctx, "kmer_write_kernel", (uint64_t) total * (8 + 8) + n_words * 8);
		  kmer_write_kernel<<<grid_for(n_words), BLOCK, 0, s>>>(ctx->genome, bitmap.as<uint32_t>(), n_words, genome_size, offsets.as<uint32_t>(), ctx->kmer_contig_table.as<uint32_t>(), keys.as<uint32_t>(), positions.as<int32_t>()); }
		uint32_t key_bits = 16;
		while ((1u << (key_bits - 16)) < n_tables) ++key_bits;
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys.as<uint32_t>(), sorted_keys.as<uint32_t>(), positions.as<int32_t>(), ctx->kmer_positions.as<int32_t>(), total, 0, key_bits, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		{ KernelTimer timer(ctx, "rocprim::radix_sort_pairs(kmer index)", (uint64_t) total * 16);
		  HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys.as<uint32_t>(), sorted_keys.as<uint32_t>(), positions.as<int32_t>(), ctx->kmer_positions.as<int32_t>(), total, 0, key_bits, s)); }
	}
	const uint32_t n_entries = std::max<uint32_t>(n_tables, 1) * KMER_COUNT;
	{ KernelTimer timer(ctx, "kmer_offsets_kernel", (uint64_t) n_entries * 4); kmer_offsets_kernel<<<grid_for((uint64_t) n_entries + 1), BLOCK, 0, s>>>(sorted_keys.as<uint32_t>(), total, n_entries, ctx->kmer_offsets.as<uint32_t>()); }
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) total * 24 + n_words * 16;
	ctx->kmer_positions_count = total;
	ctx->kmer_index_done = true;
	if (n_positions) *n_positions = total;
	return AGPU_OK;
}

namespace {
enum { PHASE_JOBS = 1, PHASE_SEARCH = 2, PHASE_FINISH = 4, PHASE_NO_JUDGE = 8 /* with PHASE_FINISH: the verdicts are with the reads, the candidates are judged later (the reads sharded over the ranks: behind the exchange of their states) */ };
// filter_mismappers in three phases -- the reads to look at (jobs), their re-alignment (search: all jobs, or every parts-th one from `part` on), the verdicts
// applied to the candidates (finish) -- so that ranks which hold the same batch can share out the search (agpu_mismapper_jobs / _verdicts / agpu_filter_mismappers_apply)
int filter_mismappers_phases(agpu_ctx* ctx, int phases, int32_t max_mate_gap, uint32_t part, uint32_t parts, uint8_t* verdicts_out, const uint8_t* verdicts_in, uint64_t* remaining, uint64_t* discarded_reads) {
	if (!ctx || !ctx->kmer_index_done) { set_last_error("agpu_make_kmer_index must run first"); return AGPU_ERR_INVALID; }
	if (parts == 0 || part >= parts) { set_last_error("part out of range"); return AGPU_ERR_INVALID; }
	if (!(phases & PHASE_JOBS) && !ctx->mismapper_jobs_ready) { set_last_error("agpu_mismapper_jobs must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t n = ctx->n;
	if (!ctx->have_splice_sites || ctx->splice_sites_for_dummy != ctx->n_dummy) { int status = build_splice_sites(ctx); if (status != AGPU_OK) return status; }
	DeviceBuffer& read_flags = ctx->scratch("mismappers.read_flags"); DeviceBuffer& jobs = ctx->scratch("mismappers.jobs"); DeviceBuffer& counters = ctx->scratch("mismappers.counters");
	DeviceBuffer& scratch = ctx->scratch("mismappers.rocprim"); DeviceBuffer& first_entry = ctx->scratch("mismappers.first_entry"); DeviceBuffer& job_keys = ctx->scratch("mismappers.job_keys");
	DeviceBuffer& job_keys_sorted = ctx->scratch("mismappers.job_keys_sorted"); DeviceBuffer& jobs_sorted = ctx->scratch("mismappers.jobs_sorted");
	ALLOC(counters, 32);
	if (phases & PHASE_JOBS) {
		ALLOC(read_flags, n ? n : 1); ALLOC(jobs, (n ? n : 1) * 4); ALLOC(first_entry, (n ? n : 1) * 8);
		HIP_CHECK(hipMemsetAsync(read_flags.ptr, 0, n ? n : 1, s));
		HIP_CHECK(hipMemsetAsync(first_entry.ptr, 0xFF, (n ? n : 1) * 8, s));
		HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 32, s));
		ctx->mismapper_jobs = 0; ctx->mismapper_jobs_ready = false;
	}
	unsigned int* device_counters = counters.as<unsigned int>(); // [0] jobs, [1] reads discarded, [2] candidates remaining, [3] reads left to the second pass, [4] queue of the second pass
	KmerIndexView kmers;
	kmers.contig_table = ctx->kmer_contig_table.as<uint32_t>(); kmers.offsets = ctx->kmer_offsets.as<uint32_t>(); kmers.positions = ctx->kmer_positions.as<int32_t>(); kmers.n_contigs = ctx->genome.n_contigs;
	SpliceSiteView splice;
	splice.offset = ctx->splice_offset.as<uint32_t>(); splice.sites = ctx->splice_sites.as<int32_t>(); splice.bits = ctx->splice_bits.as<uint32_t>();
	(void) hipEventRecord(ctx->event_start, s);
	const bool enabled = ctx->params.filter_enabled[FILTER_mismappers] != 0; // switched off with -f: the reference skips the stage (source/arriba.cpp:562); no read and no candidate is touched, the unfiltered candidates are counted
	uint32_t n_jobs = ctx->mismapper_jobs;
	if ((phases & PHASE_JOBS) && enabled && C > 0 && n > 0) {
		// (ARRIBA_MISMAPPER_JOB_ORDER=candidate: the jobs of a candidate in the order of the reads instead of the order of its lists -- the verdict of a read does not depend on the
		//  order of the jobs, and a test that runs both orders says so)
		const char* order_knob = getenv("ARRIBA_MISMAPPER_JOB_ORDER");
		const bool job_order_by_candidate = order_knob != nullptr && strcmp(order_knob, "candidate") == 0;
		{ const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			KernelTimer timer(ctx, "mismapper_flag_kernel", (uint64_t) ctx->n_list_entries * 5);
			mismapper_flag_kernel<<<grid_for(end - begin), BLOCK, 0, s>>>(ctx->batch, window, read_flags.as<uint8_t>(), first_entry.as<unsigned long long>(), job_order_by_candidate, begin, end);
			return AGPU_OK;
		  }, LISTS_OF_UNFILTERED);
		  if (status != AGPU_OK) return status; }
		size_t bytes = 0;
		HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), read_flags.as<uint8_t>(), jobs.as<uint32_t>(), device_counters, n, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::select(scratch.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), read_flags.as<uint8_t>(), jobs.as<uint32_t>(), device_counters, n, s));
		HIP_CHECK(hipMemcpyAsync(&n_jobs, device_counters, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (n_jobs > 0) { // ordered by candidate: reads that search the same genes sit next to each other
			ALLOC(job_keys, (size_t) n_jobs * 8); ALLOC(job_keys_sorted, (size_t) n_jobs * 8); ALLOC(jobs_sorted, (size_t) n_jobs * 4);
			mismapper_job_key_kernel<<<grid_for(n_jobs), BLOCK, 0, s>>>(jobs.as<uint32_t>(), n_jobs, first_entry.as<unsigned long long>(), job_keys.as<unsigned long long>());
			int key_bits = 1; // (the positions of the lists are 64-bit; only the bits a position of this sample can have are sorted)
			while (key_bits < 64 && (ctx->n_list_entries >> key_bits) != 0) ++key_bits;
			HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, job_keys.as<unsigned long long>(), job_keys_sorted.as<unsigned long long>(), jobs.as<uint32_t>(), jobs_sorted.as<uint32_t>(), n_jobs, 0, key_bits, s));
			if (bytes > scratch.capacity) ALLOC(scratch, bytes);
			HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, job_keys.as<unsigned long long>(), job_keys_sorted.as<unsigned long long>(), jobs.as<uint32_t>(), jobs_sorted.as<uint32_t>(), n_jobs, 0, key_bits, s));
		}
		ctx->mismapper_jobs = n_jobs;
	}
	if (phases & PHASE_JOBS) ctx->mismapper_jobs_ready = true;
	if ((phases & PHASE_SEARCH) && enabled && n_jobs > 0) {
		HIP_CHECK(hipMemsetAsync(device_counters + 3, 0, 8, s)); // the reads left to the second pass and its queue: a search that is run again (agpu_mismapper_verdicts called twice) starts with empty ones
		{
			// this context's share of the jobs: all of them, or every parts-th one (neighbours in the order of the candidates cost about the same: the shares are even)
			const uint32_t n_all = n_jobs;
			DeviceBuffer& my_jobs = ctx->scratch("mismappers.my_jobs");
			const uint32_t* job_list = jobs_sorted.as<uint32_t>();
			if (parts > 1) {
				n_jobs = (n_all + parts - 1 - part) / parts;
				ALLOC(my_jobs, (size_t) std::max<uint32_t>(n_jobs, 1) * 4);
				if (n_jobs > 0) mismapper_take_part_kernel<<<grid_for(n_jobs), BLOCK, 0, s>>>(jobs_sorted.as<uint32_t>(), n_all, part, parts, my_jobs.as<uint32_t>());
				job_list = my_jobs.as<uint32_t>();
			}
			DeviceBuffer& heavy = ctx->scratch("mismappers.heavy");
			ALLOC(heavy, (size_t) std::max<uint32_t>(n_jobs, 1) * 4);
			uint32_t n_heavy = 0;
			if (n_jobs > 0) {
			// Since the wavefront-per-read kernel sweeps every seed of a read once it is the cheaper place for every read: without the thread-per-read pass in front of it the
			// step is 0.04 s shorter at 10^7 fragments and 0.10 s at 10^8 (the 11 % of the reads that the first pass finished cost the second nothing measurable:
			// profiles/r03r_mismapper_passes.txt).  "1": with the first pass, for measurements.
			const char* first_pass = getenv("ARRIBA_MISMAPPER_FIRST_PASS");
			const char* steps_knob = getenv("ARRIBA_FIRST_PASS_STEPS");
			const int64_t first_pass_steps = steps_knob != nullptr && atoll(steps_knob) > 0 ? atoll(steps_knob) : FIRST_PASS_STEPS;
			if (!(first_pass != nullptr && first_pass[0] == '1')) {
				HIP_CHECK(hipMemcpyAsync(heavy.ptr, job_list, (size_t) n_jobs * 4, hipMemcpyDeviceToDevice, s));
				n_heavy = n_jobs;
			} else {
				{ KernelTimer timer(ctx, "mismapper_verdict_kernel", (uint64_t) n_jobs * 300);
				  mismapper_verdict_kernel<<<grid_for(n_jobs, ALIGN_BLOCK), ALIGN_BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, job_list, n_jobs, max_mate_gap, first_pass_steps, heavy.as<uint32_t>(), device_counters); }
				HIP_CHECK(hipMemcpyAsync(&n_heavy, device_counters + 3, 4, hipMemcpyDeviceToHost, s));
				HIP_CHECK(hipStreamSynchronize(s));
			}
			ctx->mismapper_heavy = n_heavy;
			if (n_heavy > 0) {
				// (tuning knobs for measurements: number of persistent workgroups, log2 of the memo slots of each)
				const char* knob = getenv("ARRIBA_HEAVY_WORKGROUPS");
				const uint32_t wanted = knob != nullptr && atoi(knob) > 0 ? (uint32_t) atoi(knob) : HEAVY_WORKGROUPS;
				knob = getenv("ARRIBA_MEMO_SLOTS_LOG2");
				const uint32_t memo_slots = 1u << (knob != nullptr && atoi(knob) >= 10 && atoi(knob) <= 24 ? (uint32_t) atoi(knob) : MEMO_SLOTS_LOG2);
				uint32_t workgroups = std::min<uint32_t>(n_heavy, wanted);
				DeviceBuffer& memo_tables = ctx->scratch("mismappers.memo_tables");
				DeviceBuffer& task_lists = ctx->scratch("mismappers.task_lists");
				knob = getenv("ARRIBA_TASK_CAPACITY_LOG2");
				const uint32_t task_capacity = 1u << (knob != nullptr && atoi(knob) >= 6 && atoi(knob) <= 20 ? (uint32_t) atoi(knob) : TASK_CAPACITY_LOG2);
				knob = getenv("ARRIBA_MISMAPPER_WORKLIST");
				const bool use_worklist = !(knob != nullptr && knob[0] == '0');
				// 512 KB of memo and 1 MB of task lists per persistent workgroup: 8 GB for 5120 of them (62 GB until round 6).  A device (or what other contexts have left of it) that does not hold them
				// runs the search with fewer workgroups -- slower, the same verdicts -- instead of failing
				while (!memo_tables.allocate((size_t) workgroups * memo_slots * 8) || (use_worklist && !task_lists.allocate((size_t) workgroups * task_capacity * 16 * 2))) {
					if (workgroups <= 64) { set_last_error("hipMalloc failed (the memo tables and task lists of filter_mismappers)"); return AGPU_ERR_NO_MEMORY; }
					workgroups /= 2;
				}
				HIP_CHECK(hipMemsetAsync(memo_tables.ptr, 0, (size_t) workgroups * memo_slots * 8, s));
				// the task lists of the workgroups: task_capacity tasks of 16 bytes each (and as much again for the calls of a block beyond the memory of the sweep) (a search that lists more is done by the recursion); ARRIBA_MISMAPPER_WORKLIST=0: recursion only
				knob = getenv("ARRIBA_MISMAPPER_SWEEP"); // "0": the lanes take whole listed calls in rounds (the schedule of round 2), for A/B measurements
				const bool by_sweep = !(knob != nullptr && knob[0] == '0');
				const bool want_times = getenv("ARRIBA_MISMAPPER_TIMES") != nullptr; // a study: how long the wavefronts worked on every read of the second pass, on stderr
				DeviceBuffer& read_times = ctx->scratch("mismappers.read_times");
				if (want_times) { ALLOC(read_times, (size_t) n_heavy * 32 + 96); HIP_CHECK(hipMemsetAsync(read_times.ptr, 0, (size_t) n_heavy * 32 + 96, s)); }
				// The searches go through the sweep in a kernel that holds nothing else (SWEEP_ONLY: no recursion, no stack of frames: fewer registers, no scratch memory behind them);
				// the few it gives up -- a gene of 2^24 bases and more, lists that ran over, ARRIBA_MISMAPPER_SWEEP=0 / ARRIBA_MISMAPPER_WORKLIST=0 -- are done by the kernel that holds
				// everything, which is the only one with ARRIBA_MISMAPPER_KERNELS=one (the way of round 3, for measurements).
				// (launch bounds of the full kernel: left alone the compiler takes 132 VGPRs, which fit 3 wavefronts per SIMD: 554 ms at 10^8 fragments; 4 per SIMD = 128 VGPRs, 3 spilled:
				//  459 ms with 4096 workgroups; 5 = 96 VGPRs, 61 spilled: 416 ms with 5120 workgroups -- profiles/r03p, r03r.  ARRIBA_HEAVY_WAVES=4 with ARRIBA_HEAVY_WORKGROUPS=4096 for measurements)
				knob = getenv("ARRIBA_MISMAPPER_KERNELS");
				const bool sweep_kernel_first = by_sweep && use_worklist && !(knob != nullptr && strcmp(knob, "one") == 0); // (the study of ARRIBA_MISMAPPER_TIMES looks at the first kernel when there are two)
				knob = getenv("ARRIBA_MEMO_FRONT"); // "0": every key of the memo and every listed call in HBM only, as until round 6 (for A/B measurements)
				const bool lds_front = !(knob != nullptr && knob[0] == '0');
				knob = getenv("ARRIBA_STRANDS_TOGETHER"); // "0": the two strands of a segment one after the other, as until round 6 (for A/B measurements)
				const bool strands_together = !(knob != nullptr && knob[0] == '0');
				const bool four_waves = getenv("ARRIBA_HEAVY_WAVES") != nullptr && atoi(getenv("ARRIBA_HEAVY_WAVES")) == 4;
				DeviceBuffer& leftover = ctx->scratch("mismappers.leftover");
				ALLOC(leftover, (size_t) n_heavy * 4);
				HIP_CHECK(hipMemsetAsync(device_counters + 5, 0, 8, s)); // [5] reads the sweep-only kernel gave up, [6] the queue of the kernel that does them
				const uint32_t* todo = heavy.as<uint32_t>(); uint32_t n_todo = n_heavy;
				if (sweep_kernel_first) {
					{ KernelTimer timer(ctx, "mismapper_heavy_kernel", (uint64_t) n_heavy * 300);
					  const int heavy_waves = getenv("ARRIBA_HEAVY_WAVES") != nullptr ? atoi(getenv("ARRIBA_HEAVY_WAVES")) : 5; // (6 with ARRIBA_HEAVY_WORKGROUPS=6144: 80 VGPRs, 177 spilled -- for measurements; launch bounds of 8 are not honoured: 122 VGPRs, four wavefronts)
					  if (heavy_waves == 6) mismapper_heavy_kernel<6, true><<<workgroups, 64, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, heavy.as<uint32_t>(), n_heavy, max_mate_gap, memo_tables.as<unsigned long long>(), memo_slots, task_lists.as<unsigned long long>(), task_capacity,
					                                                true, want_times ? read_times.as<unsigned long long>() : nullptr, device_counters, leftover.as<uint32_t>(), 4, lds_front, strands_together);
					  else if (!four_waves) mismapper_heavy_kernel<5, true><<<workgroups, 64, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, heavy.as<uint32_t>(), n_heavy, max_mate_gap, memo_tables.as<unsigned long long>(), memo_slots, task_lists.as<unsigned long long>(), task_capacity,
					                                                true, want_times ? read_times.as<unsigned long long>() : nullptr, device_counters, leftover.as<uint32_t>(), 4, lds_front, strands_together);
					  else mismapper_heavy_kernel<4, true><<<workgroups, 64, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, heavy.as<uint32_t>(), n_heavy, max_mate_gap, memo_tables.as<unsigned long long>(), memo_slots, task_lists.as<unsigned long long>(), task_capacity,
					                                                true, want_times ? read_times.as<unsigned long long>() : nullptr, device_counters, leftover.as<uint32_t>(), 4, lds_front, strands_together); }
					HIP_CHECK(hipMemcpyAsync(&n_todo, device_counters + 5, 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(hipStreamSynchronize(s));
					todo = leftover.as<uint32_t>();
					ctx->mismapper_leftover = n_todo;
				}
				if (n_todo > 0) {
					const uint32_t groups = std::min<uint32_t>(n_todo, workgroups);
					if (sweep_kernel_first) HIP_CHECK(hipMemsetAsync(memo_tables.ptr, 0, (size_t) groups * memo_slots * 8, s)); // (the epochs of the memo start again: nothing of the first kernel's searches may match)
					KernelTimer timer(ctx, sweep_kernel_first ? "mismapper_heavy_kernel(searches the sweep gave up)" : "mismapper_heavy_kernel", (uint64_t) n_todo * 300);
					if (!four_waves) mismapper_heavy_kernel<5, false><<<groups, 64, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, todo, n_todo, max_mate_gap, memo_tables.as<unsigned long long>(), memo_slots, use_worklist ? task_lists.as<unsigned long long>() : nullptr, task_capacity,
					                                                by_sweep, want_times && !sweep_kernel_first ? read_times.as<unsigned long long>() : nullptr, device_counters, nullptr, sweep_kernel_first ? 6 : 4, lds_front, strands_together);
					else mismapper_heavy_kernel<4, false><<<groups, 64, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, kmers, splice, todo, n_todo, max_mate_gap, memo_tables.as<unsigned long long>(), memo_slots, use_worklist ? task_lists.as<unsigned long long>() : nullptr, task_capacity,
					                                                by_sweep, want_times && !sweep_kernel_first ? read_times.as<unsigned long long>() : nullptr, device_counters, nullptr, sweep_kernel_first ? 6 : 4, lds_front, strands_together);
				}
				if (want_times) {
					std::vector<unsigned long long> ticks(4 * (size_t) n_heavy + 12);
					HIP_CHECK(hipMemcpyAsync(ticks.data(), read_times.ptr, (size_t) n_heavy * 32 + 96, hipMemcpyDeviceToHost, s));
					HIP_CHECK(hipStreamSynchronize(s));
					unsigned long long histogram[40] = { 0 }, total = 0, longest = 0, sums[4] = { 0, 0, 0, 0 };
					std::vector<std::pair<unsigned long long, uint32_t> > by_time(n_heavy);
					for (uint32_t k = 0; k < n_heavy; ++k) {
						const unsigned long long t = ticks[4 * (size_t) k];
						int bucket = 0; for (unsigned long long u = t / 100; u > 1; u >>= 1) ++bucket; histogram[bucket < 39 ? bucket : 39]++; total += t; if (t > longest) longest = t;
						sums[0] += ticks[4 * (size_t) k + 1] >> 32; sums[1] += ticks[4 * (size_t) k + 1] & 0xFFFFFFFFu; sums[2] += ticks[4 * (size_t) k + 2] >> 32; sums[3] += ticks[4 * (size_t) k + 2] & 0xFFFFFFFFu;
						by_time[k] = std::make_pair(t, k);
					}
					std::sort(by_time.begin(), by_time.end());
					fprintf(stderr, "[mismapper_heavy_kernel] %u reads of %u jobs, %u workgroups, sweep %d: %.1f ms of wavefront time in all, longest read %.2f ms; calls listed %llu, calls reaching into the blocks %llu, seeds %llu, walks %llu; reads by time (us):", n_heavy, n_jobs, workgroups, (int) by_sweep, total / 1e5, longest / 1e5, sums[0], sums[1], sums[2], sums[3]);
					for (int bucket = 0; bucket < 40; ++bucket) if (histogram[bucket]) fprintf(stderr, " 2^%d:%llu", bucket, histogram[bucket]);
					fprintf(stderr, "\n[mismapper_heavy_kernel] of the wavefront time: look-ups of the seeds %.1f ms, calls of the blocks collected %.1f ms, seeds %.1f ms\n", ticks[4 * (size_t) n_heavy] / 1e5, ticks[4 * (size_t) n_heavy + 1] / 1e5, ticks[4 * (size_t) n_heavy + 2] / 1e5);
					{ const unsigned long long* parts = &ticks[4 * (size_t) n_heavy + 4];
					  fprintf(stderr, "[mismapper_heavy_kernel] of the seeds: bases ahead + to the left %.1f ms, arrivals of the calls %.1f ms, the walk %.1f ms (%.1f); numbering the seeds of a block %.1f ms; rounds of seeds %llu, blocks with calls %llu of %llu\n",
					          parts[0] / 1e5, parts[1] / 1e5, parts[2] / 1e5, parts[3] / 1e5, parts[4] / 1e5, parts[5], parts[6], parts[7]); }
					for (uint32_t rank = 0; rank < 24 && rank < n_heavy; ++rank) { // the slowest reads, and every 1/8 quantile below them
						const uint32_t k = by_time[rank < 16 ? n_heavy - 1 - rank : (size_t) (n_heavy - 1) * (24 - rank) / 9].second;
						fprintf(stderr, "[mismapper_heavy_kernel]   read %llu (%llu alignments): %.2f ms, calls listed %llu, reaching into blocks %llu, seeds %llu, walks %llu\n", ticks[4 * (size_t) k + 3] >> 8, ticks[4 * (size_t) k + 3] & 255, ticks[4 * (size_t) k] / 1e5,
						        ticks[4 * (size_t) k + 1] >> 32, ticks[4 * (size_t) k + 1] & 0xFFFFFFFFu, ticks[4 * (size_t) k + 2] >> 32, ticks[4 * (size_t) k + 2] & 0xFFFFFFFFu);
					}
				}
			}
			}
			n_jobs = n_all;
		}
		if (verdicts_out != nullptr) { // for the exchange: one byte per job of the whole list, set where a job of this share is a mis-mapper
			DeviceBuffer& verdict_bytes = ctx->scratch("mismappers.verdicts");
			ALLOC(verdict_bytes, n_jobs);
			mismapper_collect_kernel<<<grid_for(n_jobs), BLOCK, 0, s>>>(ctx->batch, jobs_sorted.as<uint32_t>(), n_jobs, part, parts, verdict_bytes.as<uint8_t>());
			HIP_CHECK(hipMemcpyAsync(verdicts_out, verdict_bytes.ptr, n_jobs, hipMemcpyDefault, s));
		}
	}
	if (phases & PHASE_FINISH) {
		if (verdicts_in != nullptr && enabled && n_jobs > 0) { // the verdicts of all shares: the reads of the others are discarded here, too, and counted
			DeviceBuffer& verdict_bytes = ctx->scratch("mismappers.verdicts");
			ALLOC(verdict_bytes, n_jobs);
			HIP_CHECK(hipMemcpyAsync(verdict_bytes.ptr, verdicts_in, n_jobs, hipMemcpyDefault, s));
			HIP_CHECK(hipMemsetAsync(device_counters + 1, 0, 4, s));
			mismapper_apply_kernel<<<grid_for(n_jobs), BLOCK, 0, s>>>(ctx->batch, jobs_sorted.as<uint32_t>(), n_jobs, verdict_bytes.as<uint8_t>(), device_counters);
		}
		BatchView reads; // (the batch the candidates are judged by: the own one, or the replicated states of the reads of the sample -- agpu_sharded.hip)
		if (!(phases & PHASE_NO_JUDGE)) { const int status = candidate_walk_batch(ctx, reads, false); if (status != AGPU_OK) return status; }
		if (!(phases & PHASE_NO_JUDGE) && C > 0 && (!enabled || reads.n > 0)) {
			const bool count_only = !enabled;
			const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
				KernelTimer timer(ctx, "mismapper_candidate_kernel", (uint64_t) ctx->n_list_entries * 5 + (uint64_t) C * 14);
				mismapper_candidate_kernel<<<tally_grid(end - begin, BLOCK) * 4, BLOCK, 0, s>>>(reads, window, ctx->params.max_mismapper_fraction, count_only, device_counters + 2, begin, end);
				return AGPU_OK;
			}, LISTS_OF_UNFILTERED);
			if (status != AGPU_OK) return status;
		}
		if (!(phases & PHASE_NO_JUDGE)) ctx->mismapper_jobs_ready = false;
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) n_jobs * 300 + (uint64_t) ctx->n_list_entries * 10;
	unsigned int host_counters[4];
	HIP_CHECK(hipMemcpy(host_counters, counters.ptr, sizeof(host_counters), hipMemcpyDeviceToHost));
	if (remaining) *remaining = host_counters[2];
	if (discarded_reads) *discarded_reads = host_counters[1];
	return AGPU_OK;
}
}

extern "C" int agpu_filter_mismappers(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* remaining, uint64_t* discarded_reads) {
	return filter_mismappers_phases(ctx, PHASE_JOBS | PHASE_SEARCH | PHASE_FINISH, max_mate_gap, 0, 1, nullptr, nullptr, remaining, discarded_reads);
}
extern "C" int agpu_mismapper_jobs(agpu_ctx* ctx, uint64_t* n_jobs) {
	const int status = filter_mismappers_phases(ctx, PHASE_JOBS, 0, 0, 1, nullptr, nullptr, nullptr, nullptr);
	if (status == AGPU_OK && n_jobs) *n_jobs = ctx->mismapper_jobs;
	return status;
}
extern "C" int agpu_mismapper_verdicts(agpu_ctx* ctx, int32_t max_mate_gap, uint32_t part, uint32_t parts, uint8_t* verdicts) {
	if (!verdicts) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	return filter_mismappers_phases(ctx, PHASE_SEARCH, max_mate_gap, part, parts, verdicts, nullptr, nullptr, nullptr);
}
extern "C" int agpu_filter_mismappers_apply(agpu_ctx* ctx, const uint8_t* verdicts, uint64_t* remaining, uint64_t* discarded_reads) {
	if (!verdicts) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	return filter_mismappers_phases(ctx, PHASE_FINISH, 0, 0, 1, nullptr, verdicts, remaining, discarded_reads);
}

// the reads sharded over the ranks (include/arriba_gpu.h: agpu_shard_keep): every rank re-aligns the reads it holds -- the lists, which are all here, say which --, the mis-mappers
// travel as states of the reads (agpu_read_state_export / _import), and the candidates are judged on every rank from the replicated filters
extern "C" int agpu_filter_mismappers_search(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* discarded_reads) {
	if (!ctx || !ctx->read_sharded) { set_last_error("agpu_shard_keep must run first"); return AGPU_ERR_INVALID; }
	const int status = filter_mismappers_phases(ctx, PHASE_JOBS | PHASE_SEARCH | PHASE_FINISH | PHASE_NO_JUDGE, max_mate_gap, 0, 1, nullptr, nullptr, nullptr, discarded_reads);
	ctx->state_imported = false;
	return status;
}
extern "C" int agpu_filter_mismappers_judge(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->read_sharded || !ctx->state_imported) { set_last_error("agpu_filter_mismappers_search and agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
	// (the counter of the candidates that remain starts over; the reads discarded were counted by the search)
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipMemsetAsync(ctx->scratch("mismappers.counters").as<unsigned int>() + 2, 0, 4, ctx->stream));
	return filter_mismappers_phases(ctx, PHASE_FINISH, 0, 0, 1, nullptr, nullptr, remaining, nullptr);
}
